// dtof_shade.h -- the shade kernel of the wavefront path tracer (one bounce of DopplerToFPathIntegrator::sample, src/integrators/dopplertofpath.cpp:130-277,
// and, as the fused first-bounce kernel, lane generation + primary ray + up to four bounces with the path state in registers) and the launch templates
// the dtof_shade_*.hip files instantiate.
#pragma once
#include "dtof_device.h"
#include <atomic>

namespace dtof {


// Block-wide exclusive prefix of a predicate (ballot + popcount per wave, 4 wave totals through LDS).
// Returns this lane's slot relative to `running` and advances `running` by the block total.
DTOF_D uint32_t block_append(bool pred, uint32_t *s_cnt, uint32_t &running) {
    uint64_t mask = __ballot(pred);
    uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    if (kShadeBlock == 64) {   // single-wave block: the wave-level prefix is the block-level prefix
        uint32_t slot1 = running + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
        running += (uint32_t) __popcll(mask);
        return slot1;
    }
    if (lane == 0) s_cnt[wave] = (uint32_t) __popcll(mask);
    __syncthreads();
    uint32_t c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3];
    __syncthreads();
    uint32_t before = wave == 0 ? 0 : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
    uint32_t slot = running + before + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
    running += c0 + c1 + c2 + c3;
    return slot;
}

// BSDF::eval_pdf_sample of the shape's BSDF (src/render/bsdf.cpp:20-29) with every adapter around it -- what one vertex of the bounce loop asks of the material:
// value and density for the emitter direction `wo` (local frame; only when active_em) and the sampled continuation.  SPEC = 0: the diffuse-only scenes' kernels
// (diffuse under an optional twosided); 1: every BSDF / texture / adapter; 2: ... and blendbsdf / twosided with two BSDFs (the chain in a loop over two records).
// One function for k_shade and for the known-answer entry dtof_bsdf_eval (the reference's BSDF unit tests run against it on the GPU).
struct BsdfOut { V3 val, weight, wo; float pdf, bs_pdf, bs_eta; bool bs_delta, bs_null; };   // bs_null: has_flag(bs.sampled_type, BSDFFlags::Null)
template <int SPEC>
DTOF_D void bsdf_eval_pdf_sample(const SceneView &sv, const DShape *sh, Surface &si, V3 wo, bool active_em, float sample_1, float s2x, float s2y, BsdfOut &out) {
    // ---- the shape's BSDF.  Outermost a `mask`, if any: MaskBSDF (src/bsdfs/mask.cpp:125-163) -- with probability 1 - opacity the path goes straight on (a null
    // interaction: wo = -wi, weight 1, pdf 1 - opacity), otherwise the nested BSDF is sampled with sample1 / opacity; eval and pdf of the nested BSDF are scaled by it.
    const bool masked = SPEC && (sh->flags & SF_MASK);
    bool null_pick = false; float opacity = 1.f;
    if (masked) { opacity = mask_opacity_at(sv, sh, si.u, si.v); null_pick = !(sample_1 < opacity); sample_1 = sample_1 / opacity; }
    // Then a `blendbsdf`, if any (src/bsdfs/blendbsdf.cpp:114-213): eval and pdf are the weighted sums of both nested BSDFs; sample1 <= weight samples bsdf_1 with
    // sample1 / weight, otherwise bsdf_0 with (sample1 - weight) / (1 - weight), and the nested sample goes back as it is.  The chain below runs once per nested BSDF
    // (record 0 = the shape's own, record 1 = the material-only record DShape::blend_other points at).
    // (its own instantiations, SPEC == 2: with the chain inside a loop of run-time trip count the compiler's code for the K = 4 fused kernels gave wrong films
    // on scenes WITHOUT any blend -- found by the random scene sweep; where SPEC != 2 the loop below has one iteration at compile time)
    const bool blend = SPEC == 2 && (sh->flags & SF_BLEND);
    float blend_w = 0.f; bool pick_1 = false;
    if (blend) {
        blend_w = sh->tex_blend ? texture_eval_1(sv, sh->tex_blend << 4, si.u, si.v) : sh->blend_weight;
        blend_w = fmin_(fmax_(blend_w, 0.f), 1.f);   // eval_weight (:213-215)
        pick_1 = sample_1 <= blend_w;
    }
    // `twosided` with two nested BSDFs (twosided.cpp:75-86,111-148): the back side (wi.z < 0) has a record of its own; the flip itself is the chain's
    const DShape *side_sh = (SPEC == 2 && (sh->flags & SF_TWOSIDED2) && si.wi.z < 0.f) ? &sv.shapes[sh->blend_other] : sh;
    const V3 wi_plain = si.wi, wo_plain = wo;
    V3 bsdf_val = mk(0, 0, 0), bsdf_weight = mk(0, 0, 0), bs_wo = mk(0, 0, 0);
    float bsdf_pdf = 0.f, bs_pdf = 0.f, bs_eta = 0.f; bool bs_delta = false, bs_null = false;   // bs_null: has_flag(bs.sampled_type, BSDFFlags::Null)
    V3 val_0 = mk(0, 0, 0), keep_weight = mk(0, 0, 0), keep_wo = mk(0, 0, 0); float pdf_0 = 0.f, keep_pdf = 0.f, keep_eta = 0.f; bool keep_delta = false, keep_null = false;
    for (int pass = 0; pass < (SPEC == 2 && blend ? 2 : 1); ++pass) {
        const DShape *bsh = pass ? &sv.shapes[sh->blend_other] : side_sh;
        const float s1 = !blend ? sample_1 : (pass ? sample_1 / blend_w : (sample_1 - blend_w) / (1.f - blend_w));
        si.wi = wi_plain; wo = wo_plain;
        // ---- BSDF eval_pdf + sample (twosided.cpp:111-148,219-258; diffuse.cpp:101-125,160-180)
        bool twosided = bsh->flags & SF_TWOSIDED;
        // NormalMap (src/bsdfs/normalmap.cpp:110-179) around the plain BSDF, itself inside the two-sided adapter if there is one: the adapter's flip of
        // wi.z / wo.z comes first (twosided.cpp:111-148), then wi and wo move into the frame of the normal map; the sampled direction comes back the
        // same way.  A direction that changes sides between the two frames is a light leak: no value, no density, no weight.
        const bool nmap = SPEC && (bsh->flags & (SF_NORMALMAP | SF_BUMPMAP));   // BumpMap (src/bsdfs/bumpmap.cpp:114-197) wraps its nested BSDF the same way
        LocalFrame nf; V3 wo_flipped = wo; bool nm_back = false;
        if (nmap) {
            nf = (bsh->flags & SF_BUMPMAP) ? bumpmap_frame(sv, bsh, si) : normalmap_frame(sv, bsh, si);
            nm_back = twosided && si.wi.z < 0.f;
            V3 wi_f = si.wi;
            if (nm_back) { wi_f.z = -wi_f.z; wo_flipped.z = -wo_flipped.z; }
            si.wi = frame_to_local(nf, wi_f);
            wo = frame_to_local(nf, wo_flipped);
            twosided = false;
        }
        float wiz = si.wi.z, woz = wo.z;
        if (twosided) { woz = mulsign(woz, wiz); wiz = fabsf(wiz); }
        V3 refl = mk(bsh->refl[0], bsh->refl[1], bsh->refl[2]);
        if (SPEC && (bsh->nonlinear >> 1)) refl = texture_eval(sv, (bsh->nonlinear >> 1) << 4, si.u, si.v);   // m_reflectance->eval(si)
        HitMaterial hm;   // specular colours and roughness of this hit: constants, or the textures on those slots (SPEC instantiations only)
        if (SPEC) hm = material_at(sv, bsh, si.u, si.v);
        bsdf_val = mk(0, 0, 0); bsdf_weight = mk(0, 0, 0); bs_wo = mk(0, 0, 0);
        bsdf_pdf = 0.f; bs_pdf = 0.f; bs_eta = 0.f; bs_delta = false; bs_null = false;
        if (SPEC && bsh->bsdf == BSDF_CONDUCTOR) {
            // SmoothConductor::sample (conductor.cpp:226-277) under TwoSidedBRDF::sample; eval / pdf of a delta lobe are zero
            const float cos_theta_i = twosided ? fabsf(si.wi.z) : si.wi.z;
            if (cos_theta_i > 0.f) {
                bs_wo = mk(-si.wi.x, -si.wi.y, si.wi.z);   // reflect(wi); the two-sided flips of wi.z and wo.z cancel
                bs_eta = 1.f; bs_pdf = 1.f; bs_delta = true;
                bsdf_weight = mk(hm.spec_refl[0] * fresnel_conductor(cos_theta_i, bsh->cond_eta[0], bsh->cond_k[0]),
                                 hm.spec_refl[1] * fresnel_conductor(cos_theta_i, bsh->cond_eta[1], bsh->cond_k[1]),
                                 hm.spec_refl[2] * fresnel_conductor(cos_theta_i, bsh->cond_eta[2], bsh->cond_k[2]));
            }
        } else if (SPEC && bsh->bsdf == BSDF_DIELECTRIC) {
            // SmoothDielectric::sample (dielectric.cpp:231-338), TransportMode::Radiance
            float r_i, cos_theta_t, eta_it, eta_ti;
            fresnel_dielectric(si.wi.z, bsh->diel_eta, r_i, cos_theta_t, eta_it, eta_ti);
            const float t_i = 1.f - r_i;
            const bool selected_r = s1 <= r_i;
            bs_pdf = selected_r ? r_i : t_i; bs_delta = true;
            bs_wo = selected_r ? mk(-si.wi.x, -si.wi.y, si.wi.z) : mk(-eta_ti * si.wi.x, -eta_ti * si.wi.y, cos_theta_t);
            bs_eta = selected_r ? 1.f : eta_it;
            const float f2 = sqr(eta_ti);
            bsdf_weight = selected_r ? mk(hm.spec_refl[0], hm.spec_refl[1], hm.spec_refl[2])
                                     : mk(hm.spec_trans[0] * f2, hm.spec_trans[1] * f2, hm.spec_trans[2] * f2);
        } else if (SPEC && bsh->bsdf == BSDF_NULL) {
            // Null::sample (null.cpp:42-66): straight on, weight 1, pdf 1, a null (and with it a delta) lobe; eval and pdf are zero (:68-79)
            bs_wo = mk(-si.wi.x, -si.wi.y, -si.wi.z); bs_eta = 1.f; bs_pdf = 1.f; bs_delta = true; bs_null = true; bsdf_weight = mk(1.f, 1.f, 1.f);
        } else if (SPEC && bsh->bsdf == BSDF_THINDIELECTRIC) {
            // ThinDielectric::sample (thindielectric.cpp:173-226): the reflectance of the slab with all internal bounces, wo = -wi
            float r, t1, t2, t3;
            fresnel_dielectric(fabsf(si.wi.z), bsh->diel_eta, r, t1, t2, t3);
            r *= 2.f / (1.f + r);
            const bool selected_r = s1 <= r;
            bs_pdf = selected_r ? r : 1.f - r; bs_delta = true; bs_eta = 1.f;
            bs_null = !selected_r;   // bs.sampled_type = select(selected_r, DeltaReflection, Null) (:179)
            bs_wo = selected_r ? mk(-si.wi.x, -si.wi.y, si.wi.z) : mk(-si.wi.x, -si.wi.y, -si.wi.z);
            bsdf_weight = selected_r ? mk(hm.spec_refl[0], hm.spec_refl[1], hm.spec_refl[2]) : mk(hm.spec_trans[0], hm.spec_trans[1], hm.spec_trans[2]);
        } else if (SPEC && bsh->bsdf == BSDF_ROUGHDIELECTRIC) {
            // RoughDielectric::eval_pdf / sample (roughdielectric.cpp:240-346,503-611): glossy reflection and transmission lobes
            const Ggx g = mf_make((bsh->flags & SF_BECKMANN) ? MF_BECKMANN : MF_GGX, hm.alpha_u, hm.alpha_v, !(bsh->flags & SF_SAMPLE_ALL));
            const V3 wi = si.wi;
            if (active_em) rough_dielectric_eval_pdf(g, bsh, hm, wi, wo, bsdf_val, bsdf_pdf);
            if (wi.z != 0.f) {
                float mpdf;
                Ggx gs = g;   // sample_distr (:266-269)
                if (!g.visible) { const float sc = 1.2f - .2f * sqrtf(fabsf(wi.z)); gs.au *= sc; gs.av *= sc; }
                const V3 m = ggx_sample(gs, mk(mulsign(wi.x, wi.z), mulsign(wi.y, wi.z), mulsign(wi.z, wi.z)), s2x, s2y, mpdf);
                const float dwm = dot(wi, m);
                float F, cos_theta_t, eta_it, eta_ti; fresnel_dielectric(dwm, bsh->diel_eta, F, cos_theta_t, eta_it, eta_ti);
                const bool selected_r = s1 <= F;
                bs_pdf = mpdf * (selected_r ? F : 1.f - F);
                bs_eta = selected_r ? 1.f : eta_it;
                float dwh_dwo; V3 w;
                if (selected_r) {
                    bs_wo = mk(fmaf(m.x, 2.f * dwm, -wi.x), fmaf(m.y, 2.f * dwm, -wi.y), fmaf(m.z, 2.f * dwm, -wi.z));   // reflect(wi, m)
                    w = mk(hm.spec_refl[0], hm.spec_refl[1], hm.spec_refl[2]);
                    dwh_dwo = rcp(4.f * dot(bs_wo, m));
                } else {
                    const float k = fmaf(dwm, eta_ti, cos_theta_t);                                                         // refract(wi, m, cos_theta_t, eta_ti)
                    bs_wo = mk(fmaf(m.x, k, -(wi.x * eta_ti)), fmaf(m.y, k, -(wi.y * eta_ti)), fmaf(m.z, k, -(wi.z * eta_ti)));
                    const float f2 = sqr(eta_ti);
                    w = mk(f2 * hm.spec_trans[0], f2 * hm.spec_trans[1], f2 * hm.spec_trans[2]);
                    const float dom = dot(bs_wo, m);
                    dwh_dwo = (sqr(bs_eta) * dom) / sqr(dwm + bs_eta * dom);
                }
                // :345-349: smith_g1(wo, m) with visible normals, else G(wi, wo, m) dot(wi, m) / (cos_theta_i cos_theta(m))
                const float g1 = g.visible ? ggx_smith_g1(g, bs_wo, m) : ggx_smith_g1(g, wi, m) * ggx_smith_g1(g, bs_wo, m) * dwm / (wi.z * m.z);
                bs_pdf *= fabsf(dwh_dwo);
                if (mpdf != 0.f) bsdf_weight = w * g1;
            }
        } else if (SPEC && bsh->bsdf == BSDF_ROUGHCONDUCTOR) {
            // RoughConductor::eval / pdf / sample (roughconductor.cpp:229-415), GGX + visible normals, under TwoSidedBRDF
            V3 wi = si.wi, wo_l = wo;
            if (twosided && wi.z < 0.f) { wi.z = -wi.z; wo_l.z = -wo_l.z; }
            const Ggx g = mf_make((bsh->flags & SF_BECKMANN) ? MF_BECKMANN : MF_GGX, hm.alpha_u, hm.alpha_v, !(bsh->flags & SF_SAMPLE_ALL));
            if (wi.z > 0.f && wo_l.z > 0.f) {
                const V3 H = normalize(wo_l + wi);
                const float D = ggx_eval(g, H);
                if (D != 0.f) {
                    const float G = ggx_smith_g1(g, wi, H) * ggx_smith_g1(g, wo_l, H);
                    const float result = D * G / (4.f * wi.z), c = dot(wi, H);
                    bsdf_val = mk(fresnel_conductor(c, bsh->cond_eta[0], bsh->cond_k[0]) * (result * hm.spec_refl[0]),
                                  fresnel_conductor(c, bsh->cond_eta[1], bsh->cond_k[1]) * (result * hm.spec_refl[1]),
                                  fresnel_conductor(c, bsh->cond_eta[2], bsh->cond_k[2]) * (result * hm.spec_refl[2]));
                }
                if (dot(wi, H) > 0.f && dot(wo_l, H) > 0.f) bsdf_pdf = g.visible ? ggx_eval(g, H) * ggx_smith_g1(g, wi, H) / (4.f * wi.z) : ggx_pdf(g, wi, H) / (4.f * dot(wo_l, H));   // :405-409
            }
            if (wi.z > 0.f) {
                float mpdf;
                const V3 m = ggx_sample(g, wi, s2x, s2y, mpdf);
                const float dwm = dot(wi, m);
                const V3 r = mk(fmaf(m.x, 2.f * dwm, -wi.x), fmaf(m.y, 2.f * dwm, -wi.y), fmaf(m.z, 2.f * dwm, -wi.z));   // reflect(wi, m)
                bs_wo = r; bs_eta = 1.f;
                const bool ok = mpdf != 0.f && r.z > 0.f;
                const float weight = g.visible ? ggx_smith_g1(g, r, m) : ggx_smith_g1(g, wi, m) * ggx_smith_g1(g, r, m) * dwm / (wi.z * m.z);   // :260-265
                bs_pdf = mpdf / (4.f * dot(r, m));
                if (ok) bsdf_weight = mk(fresnel_conductor(dwm, bsh->cond_eta[0], bsh->cond_k[0]) * (weight * hm.spec_refl[0]),
                                         fresnel_conductor(dwm, bsh->cond_eta[1], bsh->cond_k[1]) * (weight * hm.spec_refl[1]),
                                         fresnel_conductor(dwm, bsh->cond_eta[2], bsh->cond_k[2]) * (weight * hm.spec_refl[2]));
                if (twosided && si.wi.z < 0.f) bs_wo.z = -bs_wo.z;
            }
        } else if (SPEC && bsh->bsdf == BSDF_ROUGHPLASTIC) {
            // RoughPlastic::eval / pdf / sample (roughplastic.cpp:259-421), GGX + visible normals, under TwoSidedBRDF
            V3 wi = si.wi, wo_l = wo;
            if (twosided && wi.z < 0.f) { wi.z = -wi.z; wo_l.z = -wo_l.z; }
            const Ggx g = mf_make((bsh->flags & SF_BECKMANN) ? MF_BECKMANN : MF_GGX, hm.alpha_u, hm.alpha_u, !(bsh->flags & SF_SAMPLE_ALL));
            const float *table = (const float *) (sv.base + bsh->rough_table);
            const float w = bsh->spec_sampling_weight, ir = bsh->fdr_int;
            const V3 diff = (bsh->nonlinear & 1u) ? mk(refl.x / (1.f - refl.x * ir), refl.y / (1.f - refl.y * ir), refl.z / (1.f - refl.z * ir))
                                          : mk(refl.x / (1.f - ir), refl.y / (1.f - ir), refl.z / (1.f - ir));
            if (wi.z > 0.f) {
                const float t_i = lerp_gather64(table, wi.z);
                float prob_specular = (1.f - t_i) * w, prob_diffuse = t_i * (1.f - w);
                prob_specular = prob_specular / (prob_specular + prob_diffuse);
                prob_diffuse = 1.f - prob_specular;
                if (wo_l.z > 0.f) rough_plastic_eval_pdf(g, bsh, hm, table, diff, wi, wo_l, t_i, prob_specular, prob_diffuse, bsdf_val, bsdf_pdf);
                if (s1 < prob_specular) {
                    float mpdf; const V3 m = ggx_sample(g, wi, s2x, s2y, mpdf);
                    const float dwm = dot(wi, m);
                    bs_wo = mk(fmaf(m.x, 2.f * dwm, -wi.x), fmaf(m.y, 2.f * dwm, -wi.y), fmaf(m.z, 2.f * dwm, -wi.z));   // reflect(wi, m)
                } else bs_wo = cosine_hemisphere(s2x, s2y);
                bs_eta = 1.f;
                V3 value = mk(0, 0, 0);
                if (bs_wo.z > 0.f) rough_plastic_eval_pdf(g, bsh, hm, table, diff, wi, bs_wo, t_i, prob_specular, prob_diffuse, value, bs_pdf);
                if (bs_pdf > 0.f) bsdf_weight = value * rcp(bs_pdf);                  // Spectrum / Float = multiplication by the reciprocal
                if (twosided && si.wi.z < 0.f) bs_wo.z = -bs_wo.z;
            }
        } else if (SPEC && bsh->bsdf == BSDF_PLASTIC) {
            // SmoothPlastic::eval / pdf / sample (plastic.cpp:219-360) under TwoSidedBRDF; wiz / woz are already flipped
            float f_i, t1, t2, t3;
            fresnel_dielectric(wiz, bsh->diel_eta, f_i, t1, t2, t3);
            const float w = bsh->spec_sampling_weight, fdr = bsh->fdr_int;
            const V3 diff = (bsh->nonlinear & 1u) ? mk(refl.x / (1.f - refl.x * fdr), refl.y / (1.f - refl.y * fdr), refl.z / (1.f - refl.z * fdr))
                                          : mk(refl.x / (1.f - fdr), refl.y / (1.f - fdr), refl.z / (1.f - fdr));
            if (wiz > 0.f && woz > 0.f) {
                float f_o; fresnel_dielectric(woz, bsh->diel_eta, f_o, t1, t2, t3);
                const float k = kInvPi * woz * bsh->inv_eta_2 * (1.f - f_i) * (1.f - f_o);
                bsdf_val = mk(diff.x * k, diff.y * k, diff.z * k);
                const float prob_specular = f_i * w; float prob_diffuse = (1.f - f_i) * (1.f - w);
                prob_diffuse = prob_diffuse / (prob_specular + prob_diffuse);
                bsdf_pdf = kInvPi * woz * prob_diffuse;
            }
            if (wiz > 0.f) {
                float prob_specular = f_i * w, prob_diffuse = (1.f - f_i) * (1.f - w);
                prob_specular = prob_specular / (prob_specular + prob_diffuse);
                prob_diffuse = 1.f - prob_specular;
                bs_eta = 1.f;
                if (s1 < prob_specular) {
                    bs_wo = mk(-si.wi.x, -si.wi.y, wiz);
                    bs_pdf = prob_specular; bs_delta = true;
                    const float value = f_i / bs_pdf;
                    bsdf_weight = mk(value * hm.spec_refl[0], value * hm.spec_refl[1], value * hm.spec_refl[2]);
                } else {
                    bs_wo = cosine_hemisphere(s2x, s2y);
                    bs_pdf = prob_diffuse * (kInvPi * bs_wo.z);
                    float f_o; fresnel_dielectric(bs_wo.z, bsh->diel_eta, f_o, t1, t2, t3);
                    const float k = bsh->inv_eta_2 * (1.f - f_i) * (1.f - f_o) / prob_diffuse;
                    bsdf_weight = mk(diff.x * k, diff.y * k, diff.z * k);
                }
                if (twosided) bs_wo.z = mulsign(bs_wo.z, si.wi.z);
            }
        } else {
            if (wiz > 0.f && woz > 0.f) { bsdf_val = mk(refl.x * kInvPi * woz, refl.y * kInvPi * woz, refl.z * kInvPi * woz); bsdf_pdf = kInvPi * woz; }
            if (wiz > 0.f) {
                bs_wo = cosine_hemisphere(s2x, s2y);
                bs_pdf = kInvPi * bs_wo.z;
                bs_eta = 1.f;
                if (bs_pdf > 0.f) bsdf_weight = refl;
                if (twosided) bs_wo.z = mulsign(bs_wo.z, si.wi.z);
            }
        }
        if (nmap) {
            if (!(wo_flipped.z * wo.z > 0.f)) { bsdf_val = mk(0, 0, 0); bsdf_pdf = 0.f; }
            if (bsdf_weight.x != 0.f || bsdf_weight.y != 0.f || bsdf_weight.z != 0.f) {   // active &= any(weight != 0): a zero sample goes back as it is
                const V3 pw = frame_to_world(nf, bs_wo);
                if (!(bs_wo.z * pw.z > 0.f)) bsdf_weight = mk(0, 0, 0);
                bs_wo = pw;
            }
            if (nm_back) bs_wo.z = -bs_wo.z;
        }
        if (blend) {
            if ((pass == 1) == pick_1) { keep_weight = bsdf_weight; keep_wo = bs_wo; keep_pdf = bs_pdf; keep_eta = bs_eta; keep_delta = bs_delta; keep_null = bs_null; }
            if (pass == 0) { val_0 = bsdf_val; pdf_0 = bsdf_pdf; }
            else {
                const float w0 = 1.f - blend_w;
                bsdf_val = mk(val_0.x * w0 + bsdf_val.x * blend_w, val_0.y * w0 + bsdf_val.y * blend_w, val_0.z * w0 + bsdf_val.z * blend_w);
                bsdf_pdf = pdf_0 * w0 + bsdf_pdf * blend_w;
                bsdf_weight = keep_weight; bs_wo = keep_wo; bs_pdf = keep_pdf; bs_eta = keep_eta; bs_delta = keep_delta; bs_null = keep_null;
            }
        }
    }
    si.wi = wi_plain;
    if (masked) {
        bsdf_val = bsdf_val * opacity; bsdf_pdf *= opacity;
        if (null_pick) { bs_wo = mk(-si.wi.x, -si.wi.y, -si.wi.z); bs_eta = 1.f; bs_pdf = 1.f - opacity; bs_delta = true; bs_null = true; bsdf_weight = mk(1.f, 1.f, 1.f); }
    }
    out.val = bsdf_val; out.pdf = bsdf_pdf; out.weight = bsdf_weight; out.wo = bs_wo; out.bs_pdf = bs_pdf; out.bs_eta = bs_eta; out.bs_delta = bs_delta; out.bs_null = bs_null;
}

// FUSED = false: the "split" pipeline -- shadow rays go to the shadow queue (k_shadow commits them) and the
//                 continuation ray is traced by the next k_trace launch.
// FUSED = true : one kernel per bounce -- the occlusion query and the closest-hit query of the continuation ray
//                 run inline, so neither the shadow queue nor a separate trace launch exists; the state streams
//                 through HBM once per bounce (this is the default: the split kernels are latency-bound on small
//                 scenes and the shadow records alone cost 96 B per path-bounce).
// AREA: the scene has area emitters (emitter-hit term, prev_si state).  KMAX: compile-time bound of the batched offsets (1 or 4);
// both keep the common case -- point lights, one offset -- free of the extra registers.
// MODE 0 = split, 1 = fused, 2 = fused AND first bounce: the lane is generated (sampler seeding, camera ray) and its primary
// ray traced right here, so the 96-byte primary state never makes the round trip through HBM and neither k_generate nor the
// primary k_trace launch exists (`dbg`, if given, receives the camera ray for the lane-dump entry point).
// SPEC: the scene has delta BSDFs (conductor / dielectric): the relative index of refraction along the path and the
// "previous lobe was a delta" flag travel in st_c; instantiated together with AREA and MESH only.
// The arguments of k_shade travel as ONE by-value block and are read through the kernarg segment pointer, re-based once per chunk
// on an offset the compiler cannot see through: the ~170 dwords of parameters (camera matrices, sampler and modulation constants, 17
// queue pointers) are then fetched by scalar loads where they are used, instead of being loaded once ahead of the chunk loop and
// kept alive across it -- which, in the first-bounce instantiation, spilled ~120 SGPRs to VGPR lanes (350 v_writelane / v_readlane).
// RESW != 0: the RESIDENT form of the fused first-bounce kernel for scenes too large to stage whole (Domino: 64 KB of TLAS nodes + 128 KB of
// instance records).  ONE block of RESW waves per CU stays for the whole launch; it copies the TLAS nodes (as four planes, see kResNodes) and
// the block of small records (groups, shapes, emitters, triangles, shading data) into LDS once, then every wave takes 512-lane segments from a
// global counter until none is left -- a wave owns its segment exactly as a one-wave block does, there is no barrier after the stage.  What
// still comes through the vector L1 is the 128-byte instance record of a leaf visit and the queue traffic: the unstaged kernel keeps the CU's
// vector memory path busy 75 - 85 % of the time (TA / TD busy counters, profiles/r03_pmc_domino_fused*.txt) with the four 16-byte node loads
// per step per lane, and waits for it.
template <bool LDS, int MODE, bool AREA, int KMAX, bool MESH, int SPEC, int RESW = 0, bool RH16 = false>   // RH16: the resident stage holds HALF-FLOAT node records (DNode16): a TLAS of up to 2 * kResNodes nodes in the LDS of kResNodes float ones; SPEC: 0 diffuse-only scenes, 1 every BSDF / emitter / texture, 2 = 1 + blendbsdf (the BSDF chain in a loop over two records)
#ifndef DTOF_MESH_WAVES
#define DTOF_MESH_WAVES 3   // waves / SIMD the fused kernels with triangle code are compiled for (A/B: make variant DEFS=-DDTOF_MESH_WAVES=4)
#endif
// History of the every-BSDF kernels (SPEC) with four offset films (KMAX == 4): at three waves per SIMD (168 VGPRs, 240 - 390 spilled registers) their fused instantiations
// produced wrong films on scenes of the random sweep whenever the kernel grew, while the K = 1 kernels, the split pipeline and the same source at two waves stayed exact.
// The cause was not the spill code: the films depended on the INITIAL value of the path-state registers declared without one (`main` / `path` below; right with
// -ftrivial-auto-var-init=zero, NaN with =pattern, profiles/r03_k4_uninitialised.txt).  They are initialised now, and both wave counts are correct (sweeps of 240 .. 1 200
// scenes each).  Two waves stay for these instantiations because they are FASTER there: with four films in registers the 168-VGPR build spills 240 - 390 registers, and
// the K = 4 frames of the every-BSDF scenes take 1 - 10 % longer at three waves (cornell_specular 9.13 -> 9.60 ms, cornell_spot 7.80 -> 8.58; profiles/r03_k4_waves_ab.txt)
// -- the opposite of the K = 1 kernels, which lose 20 - 27 % at two (profiles/r03_spec_waves_ab.txt).
// DTOF_K4_RES_MEM (default 1): the fused first-bounce kernels of SEVERAL films (KMAX > 1) keep neither the K running results nor the K NEE candidates in registers.
// Round 4's four-film Domino kernel carried 12 + 12 of them across both traversals of every iteration, in scratch: 54 GB of HBM traffic per launch against 10.7 GB of
// outputs, half its wave-cycles waiting (profiles/r04_pmc_c4_c5.txt).  What is pending across the shadow ray is K-INDEPENDENT -- the throughput, the unweighted
// contribution bsdf_val * em_weight * mis_em and the path length to the emitter (dopplertofpath.cpp:214-226: only eval_modulation_weight, :60-77, depends on the offset)
// = 7 registers -- and the K weights are applied when the sample is committed, in the same fmaf order.  The running results live where they have to end up anyway, in
// q.res: a commit is a read-modify-write of the lane's K records (the first one of a path writes without reading, `res_live`), lines the same wave wrote a few
// microseconds earlier.  MEASURED (profiles/r05_k4_film_state.txt): on the four-film Domino frame that form is no faster than the registers (181.9 against 178.9 ms) --
// the kernel's scratch stays above the L2 either way -- so it is OFF (=1 builds it for A/B).
// DTOF_K4_RES_LDS (default 1): the RESIDENT several-film kernels keep the running results in LDS instead: Domino's stage leaves 44 KiB of the CU's 160 free at 16 waves,
// kParkWords = 11 words per thread, which hold 11 of the 12 floats of four RGB results (the twelfth stays a register); the pending sample is the K-independent one
// described above.  Film state then costs the K = 4 kernel ONE register more than the K = 1 kernel's, and a commit is 12 ds_read + 12 ds_write.
#ifndef DTOF_K4_RES_MEM
#define DTOF_K4_RES_MEM 0
#endif
#ifndef DTOF_K4_RES_LDS
#define DTOF_K4_RES_LDS 1
#endif
__global__ __launch_bounds__(RESW ? RESW * 64 : kShadeBlock, RESW ? RESW / 4 : (MODE == 2 && !MESH && !SPEC && KMAX == 1) ? 4 : (MODE != 0 && MESH) ? ((SPEC && KMAX > 1) ? 2 : DTOF_MESH_WAVES) : 1) void k_shade(ShadeArgs args_by_value) {
    constexpr bool FUSED = MODE != 0, FIRST = MODE == 2;
    constexpr bool RES_LDS = RESW != 0 && KMAX > 1 && DTOF_K4_RES_LDS;   // several films, resident stage: running results in LDS columns
    constexpr bool S16 = RESW != 0 && KMAX > 1;                           // ... whose LDS comes from 16-bit traversal stacks (dtof_traverse.h: encode_child16)
    constexpr bool RES_MEM = FIRST && KMAX > 1 && (RES_LDS || DTOF_K4_RES_MEM);   // several films: running results outside the registers (LDS, else q.res), the pending NEE sample K-independent (see above)
    static_assert(!RES_LDS || 3 * KMAX == (int) kParkWords + 1, "the film-state columns hold all but the last float of the K results");
    constexpr bool PARK_ST = DTOF_PARK && RESW == 16 && KMAX == 1;   // the path state no traversal reads waits in LDS columns while the rays are traced: throughput / path length (one film only) ...
    constexpr bool PARK = DTOF_PARK && RESW == 16 && (KMAX == 1 || RES_LDS);   // ... and both PCG streams (several films: behind the film words)
    constexpr uint32_t kRngAt = RES_LDS ? kParkWords : 0u;
    constexpr int KREG = RES_MEM ? 1 : KMAX;                         // film-state registers the lane carries
    constexpr uint32_t kStackStride = RESW ? RESW * 64 : kShadeBlock;   // the block size = the stride of the traversal-stack columns
    static_assert(RESW == 0 || (MODE == 2 && !LDS && MESH), "the resident stage exists for the unstaged fused first-bounce kernel with mesh code");
    static_assert(!RH16 || RESW != 0, "half-float LDS planes belong to the resident stage");
    extern __shared__ uint4 lds[];
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_inline_all[(RESW ? RESW : 1) * 2 * kMaxInline];   // FIRST: lanes alive after / shadow rays of every inline iteration but the last (statistics), per wave
    typedef const char __attribute__((address_space(4))) *KernargBytes;
    const KernargBytes kernarg = (KernargBytes) __builtin_amdgcn_kernarg_segment_ptr();
    const ShadeArgs &A0 = *(const ShadeArgs *) kernarg;
    const uint32_t lane_id = RESW ? threadIdx.x & 63u : threadIdx.x, wave_id = RESW ? threadIdx.x >> 6 : 0u;   // one wave per block otherwise
    uint32_t *const s_inline = s_inline_all + wave_id * 2 * kMaxInline;
    const uint32_t stage_words = RESW ? 4u * kResNodes + A0.res_small_words : A0.stage_words;
    // dynamic LDS: [staged scene | resident stage: node planes, record block][fused: instance memo, kMemoWords x 64 words per wave][traversal stack columns]
    // flat scenes with ONE instance (the moving wall of C2): the column holds the instance matrix as well (launch_shade sizes the LDS accordingly) -- re-deriving it per
    // iteration cost every lane 51 instructions, a hit on the wall now reads it back
    const bool memo_m_lds = FUSED && !MESH && !RESW && A0.rp.flat_objects != 0u && A0.rp.memo_obj != 0xffffffffu;
    const uint32_t memo_words = FUSED ? (RESW ? (A0.res_memo ? RESW * kMemoWords * kMemoStride : 0u) : (memo_m_lds ? 2u : 1u) * kMemoWords * kMemoStride) : 0u;
    // per-thread traversal stack column; the resident kernels' columns are 16-bit (halfword address in a uint32_t *: only trace_scene<SOA> / node_step<SOA> touch it)
    uint32_t *stack = S16 ? (uint32_t *) ((uint16_t *) ((uint32_t *) (lds + stage_words) + memo_words) + threadIdx.x) : (uint32_t *) (lds + stage_words) + memo_words + threadIdx.x;
    // One block per 512-lane segment -- or, for a small frame whose whole path runs inline (rp.chunk_blocks = 8: nothing is compacted for a
    // later launch), one block per 64-lane chunk, so that a 1 M-lane frame is 16 384 waves instead of 2 048; the per-segment statistics are
    // then accumulated with atomics into slots the host has zeroed.
    const uint32_t sub = FIRST ? (RESW ? A0.rp.res_units : A0.rp.chunk_blocks) : 1u;   // work units per segment: blocks (1 or kSeg / kShadeBlock), resident: what a wave takes from the counter at a time
    const uint32_t unit_lanes = kSeg / sub;
    SceneView sv_res;
    if (RESW) {   // the resident stage: every thread of the block copies, ONE barrier, then the waves go their own ways
        const BlobHeader *gh = (const BlobHeader *) A0.scene;
        if (RH16) {   // two planes of 2 * kResNodes 16-byte pieces: (left box, left child) | (right box, right child)
            const uint4 *gn = (const uint4 *) (A0.scene + gh->off_nodes16);
            const uint32_t n_pieces = gh->n_nodes * 2u;
            for (uint32_t i = threadIdx.x; i < n_pieces; i += blockDim.x) {
                uint4 piece = gn[i];
                if (S16) piece.w = encode_child16(piece.w);
                lds[(i & 1u) * (2u * kResNodes) + (i >> 1)] = piece;
            }
        } else {
        const uint4 *gn = (const uint4 *) (A0.scene + gh->off_nodes);
        const uint32_t n_pieces = gh->n_nodes * 4u;
        for (uint32_t i = threadIdx.x; i < n_pieces; i += blockDim.x) {
            uint4 piece = gn[i];
            if (S16 && (i & 3u) < 2u) piece.w = encode_child16(piece.w);   // left / right child: 16-bit references (dtof_traverse.h), so that the stack columns are 16-bit
            lds[(i & 3u) * kResNodes + (i >> 2)] = piece;
        }
        }
        const uint4 *gs = (const uint4 *) (A0.scene + A0.res_small_off);
        for (uint32_t i = threadIdx.x; i < A0.res_small_words; i += blockDim.x) lds[4u * kResNodes + i] = gs[i];
        __syncthreads();
        const uint8_t *small = (const uint8_t *) (lds + 4u * kResNodes) - A0.res_small_off;   // blob offsets of the copied block resolve into LDS
        sv_res = make_view(A0.scene);
        sv_res.nodes = (const DNode *) lds; sv_res.nodes16 = (const DNode16 *) lds;
        sv_res.groups = (const DGroup *) (small + gh->off_groups); sv_res.shapes = (const DShape *) (small + gh->off_shapes);
        sv_res.tris = (const DTri *) (small + gh->off_tris); sv_res.shading = (const DTriShade *) (small + gh->off_shading);
        sv_res.isect = (const DTriIsect *) (small + gh->off_isect);
        sv_res.emitters = (const DEmitter *) (small + gh->off_emitters);
    }
    for (uint32_t seg_first = 1;; seg_first = 0) {   // resident: until the segment counter runs out; otherwise once
    uint32_t seg, sub_index = 0;
    if (RESW) {
        uint32_t taken = 0;
        if (lane_id == 0) taken = atomicAdd(A0.q.seg_counter, 1u);
        uint32_t unit = (uint32_t) __builtin_amdgcn_readfirstlane((int) taken);
        if (unit >= A0.n_seg * sub) break;
        // the units are handed out from the LAST pixel rows to the first: a launch ends with a tail in which the waves run out of work one after the other, and the
        // tail is as long as the last units are expensive -- the top rows of a frame tend to see sky (Domino C4: 34.05 -> 33.77 ms, profiles/r04_domino_waves_batch.txt)
        unit = A0.n_seg * sub - 1u - unit;
        seg = sub > 1 ? unit / sub : unit; sub_index = sub > 1 ? unit - seg * sub : 0u;
    } else {
        seg = sub > 1 ? blockIdx.x / sub : blockIdx.x; sub_index = sub > 1 ? blockIdx.x - seg * sub : 0u;
    }
    (void) seg_first;
    const uint32_t count = seg_count(A0.count_in, seg, A0.rp.n_lanes);
    uint32_t n_alive = 0, n_shadow = 0;
    if (FIRST && lane_id < 2 * kMaxInline) s_inline[lane_id] = 0;   // a wave's own slots: no barrier needed
    if (count != 0) {
    const uint8_t *base = LDS ? stage_scene(A0.scene, A0.scene_bytes, lds) : A0.scene;
    SceneView sv = RESW ? sv_res : make_view(base);
    if (FUSED) { sv.memo_obj = A0.rp.memo_obj; sv.memo = (float *) (lds + stage_words) + (RESW ? wave_id * kMemoWords * kMemoStride + lane_id : threadIdx.x); sv.memo_m = memo_m_lds; }
    const bool have_memo = FUSED && sv.memo_obj != 0xffffffffu;
    for (uint32_t cbase = sub > 1 ? sub_index * unit_lanes : 0u; cbase < (sub > 1 ? (sub_index + 1) * unit_lanes < count ? (sub_index + 1) * unit_lanes : count : count); cbase += kShadeBlock) {
    uint32_t rebase = 0;
    asm volatile("" : "+s"(rebase));
    const ShadeArgs &A = *(const ShadeArgs *) (kernarg + rebase);
    const RenderParams &rp = A.rp; const Queues &q = A.q;
    const uint32_t *const qin = A.qin; uint32_t *const qout = A.qout; const uint32_t depth0 = A.depth, trace_next_last = A.trace_next; LaneDebug *const dbg = A.dbg;
    const uint32_t flat = FUSED && !MESH ? rp.flat_objects : 0u;   // != 0: the scene's object count, every ray tests them all (trace_flat)
    uint32_t j = cbase + lane_id;
    bool in_range = j < count;
    bool alive = false, want_shadow = false;
    uint32_t l = 0;
    // every register of the path state starts defined: a build of these kernels whose K = 4 every-BSDF instantiations ran at three waves per SIMD produced films that
    // depended on the initial value of `main` / `path` below (wrong with the registers' garbage, NaN with -ftrivial-auto-var-init=pattern, right with =zero;
    // profiles/r03_k4_uninitialised.txt) although no source path reads them before they are assigned
    float4 sha = make_float4(0.f, 0.f, 0.f, 0.f), shb = sha, nra = sha, nrb = sha; float3 cand[KREG];
#pragma unroll
    for (int k = 0; k < KREG; ++k) cand[k] = make_float3(0.f, 0.f, 0.f);
    float3 rbase[KREG];   // FIRST: the result a lane ends this launch with if its NEE candidate is not committed
#pragma unroll
    for (int k = 0; k < KREG; ++k) rbase[k] = make_float3(0.f, 0.f, 0.f);
    // RES_MEM: the pending emitter sample without its modulation weights -- throughput before the bounce, bsdf_val * em_weight * mis_em, path length to the emitter --
    // and whether q.res holds the lane's running result yet (it is 0 until something was added: the first write needs no read, a path that adds nothing writes zeros at the end)
    V3 pend_thr = mk(0, 0, 0), pend_v = mk(0, 0, 0); float pend_len = 0.f; bool res_live = false;
    // the lane's running result of film k: a register (FIRST), its q.res record (the bounce kernels; RES_MEM once res_live -- before that it is 0 and nothing is read)
    float *const park = RES_LDS || PARK ? (float *) (lds + stage_words) + memo_words + A0.res_park_off + threadIdx.x : nullptr;   // word f of this thread at park[f * kStackStride]
    float res_last = 0.f;   // RES_LDS: the one float of the K results that has no LDS word (film KMAX - 1, blue)
    if (RES_LDS) {
#pragma unroll
        for (uint32_t f = 0; f < kParkWords; ++f) park[f * kStackStride] = 0.f;
    }
    auto res_get = [&](int k) -> float4 {
        if (RES_LDS) return make_float4(park[(3 * k) * kStackStride], park[(3 * k + 1) * kStackStride], 3 * k + 2 < (int) kParkWords ? park[(3 * k + 2) * kStackStride] : res_last, 0.f);
        if (RES_MEM) return res_live ? q.res[(size_t) k * q.capacity + l] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (FIRST) { const float3 r = rbase[RES_MEM ? 0 : k]; return make_float4(r.x, r.y, r.z, 0.f); }
        return q.res[(size_t) k * q.capacity + l];
    };
    auto res_put = [&](int k, float4 v) {
        if (RES_LDS) {
            park[(3 * k) * kStackStride] = v.x; park[(3 * k + 1) * kStackStride] = v.y;
            if (3 * k + 2 < (int) kParkWords) park[(3 * k + 2) * kStackStride] = v.z; else res_last = v.z;
        }
        else if (FIRST && !RES_MEM) rbase[RES_MEM ? 0 : k] = make_float3(v.x, v.y, v.z);
        else q.res[(size_t) k * q.capacity + l] = v;
    };
    // "the lane's results are zero again" (a path of null interactions that ends invalid): RES_LDS rewrites its columns, the q.res form only drops its flag
    auto res_clear = [&]() {
        if (RES_LDS) {
#pragma unroll
            for (uint32_t f = 0; f < kParkWords; ++f) park[f * kStackStride] = 0.f;
            res_last = 0.f;
        }
        res_live = false;
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // Path state of the lane.  MODE 2 runs rp.inline_iters iterations of the bounce loop right here ("megakernel" head): between them the
    // state stays in these registers instead of making the round trip through the queues in HBM; after the last one the survivors are
    // written out and compacted exactly as before, for the bounce kernels (MODE 1) to continue with.
    uint32_t hid = 0xffffffffu; float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = make_float4(0.f, 0.f, 1.f, 0.f), st = ra; uint4 hh = make_uint4(0u, 0u, 0u, 0u); Rng main = { 0ull, 1ull }, path = { 0ull, 1ull };
    float4 stb_reg = make_float4(0.f, 0.f, 0.f, 1.f); float2 stc_reg = make_float2(1.f, 1.f);   // prev_si.p | prev_bsdf_pdf, eta | prev_bsdf_delta
    uint32_t *const parku = (uint32_t *) park;
    auto park_store = [&]() {   // PARK: both streams (state + 32-bit selector: inc = sel << 1 | 1) and throughput / path length leave the registers
        parku[kRngAt * kStackStride] = (uint32_t) main.state; parku[(kRngAt + 1) * kStackStride] = (uint32_t) (main.state >> 32); parku[(kRngAt + 2) * kStackStride] = (uint32_t) (main.inc >> 1);
        parku[(kRngAt + 3) * kStackStride] = (uint32_t) path.state; parku[(kRngAt + 4) * kStackStride] = (uint32_t) (path.state >> 32); parku[(kRngAt + 5) * kStackStride] = (uint32_t) (path.inc >> 1);
        if (PARK_ST) { park[6 * kStackStride] = st.x; park[7 * kStackStride] = st.y; park[8 * kStackStride] = st.z; park[9 * kStackStride] = st.w; }
    };
    auto park_load = [&]() {
        main.state = (uint64_t) parku[kRngAt * kStackStride] | ((uint64_t) parku[(kRngAt + 1) * kStackStride] << 32); main.inc = ((uint64_t) parku[(kRngAt + 2) * kStackStride] << 1) | 1u;
        path.state = (uint64_t) parku[(kRngAt + 3) * kStackStride] | ((uint64_t) parku[(kRngAt + 4) * kStackStride] << 32); path.inc = ((uint64_t) parku[(kRngAt + 5) * kStackStride] << 1) | 1u;
        if (PARK_ST) st = make_float4(park[6 * kStackStride], park[7 * kStackStride], park[8 * kStackStride], park[9 * kStackStride]);
    };
    // valid_ray (dopplertofpath.cpp:101-102,252-253,279-282): starts as "the environment is visible", becomes true at the first vertex whose sampled lobe is not
    // BSDFFlags::Null (a `mask` that lets the path through, the transmission of a `thindielectric`); a path that ends without it returns 0 (and alpha 0).  Only the
    // every-BSDF kernels (SPEC) can sample a null lobe: elsewhere a lane is valid as soon as its primary ray hits something.  FIRST: this register; the bounce
    // kernels carry it as bit 1 of st_c.y (bit 0 = prev_bsdf_delta).
    const bool valid_start = SPEC && rp.has_env && !rp.hide_emitters;
    bool valid_reg = valid_start;
    float memo_m[12], memo_inv[12];   // instance memo: the one instance's matrix and inverse at this lane's ray time
    bool lane_on = in_range;
#ifdef DTOF_TRAVERSAL_STATS   // development builds: lanes without a path carry a poison pattern; DTOF_POISON_CHECK counts it wherever path state is written out (statistic [14])
    constexpr uint64_t kPoisonState = 0xdeadbeefcafef00dull;
    if (!in_range) { main.state = path.state = kPoisonState; const float pz = u2f(0x7fc0dead); ra = rb = st = make_float4(pz, pz, pz, pz); }
#define DTOF_POISON_CHECK() do { if (main.state == kPoisonState || path.state == kPoisonState || f2u(st.x) == 0x7fc0deadu || f2u(ra.x) == 0x7fc0deadu) DTOF_STAT(14); } while (0)
#else
#define DTOF_POISON_CHECK() ((void) 0)
#endif
    // the wave's 64 lanes are the 64-aligned lanes [lane_base + seg * 512 + cbase, + 64): samples of one pixel if spp is a multiple of 64
    // ... AND the whole wave is in range: the pair swap of the correlated seeding (generate_lane) reads the partner lane, which a ragged tail
    // (dtof_sample_lanes with an odd count) would leave inactive
    const bool wave_pixel = FIRST && rp.spp_log2 != 0xffffffffu && rp.spp_log2 >= 6 && (rp.lane_base & 63u) == 0 && count - cbase >= (uint32_t) kShadeBlock;
    // Fused splat (uniform): this launch runs the whole path of its lanes (nobody continues) and the host handed in the film -- the wave reduces the footprint
    // values of its 64 samples itself and issues the film atomics, the result never goes through q.res / q.pos and the splat kernel's round trip through HBM
    const bool fuse_splat = FIRST && A.film != nullptr;
    float2 pos_reg = make_float2(0.f, 0.f);
    if (in_range) {
        l = qin ? qin[seg * kSeg + j] : seg * kSeg + j;
        if (FIRST) {
            const PrimaryLane pl = generate_lane<SPEC == 0>(rp, global_lane(rp, rp.lane_base + l), wave_pixel, rp.lane_base + l);
            ra = pl.ray_a; rb = pl.ray_b; main = pl.main; path = pl.path; st = make_float4(1.f, 1.f, 1.f, 0.f);
            if (PARK) park_store();
            pos_reg = pl.pos;
            if (!fuse_splat) {   // what the later launches (stream selectors) and the splat kernels (sample position) read
                q.pos[l] = pl.pos;
                q.rng_b[l] = make_uint2((uint32_t) (main.inc >> 1), (uint32_t) (path.inc >> 1));
            }
            if (dbg) {
                LaneDebug &o = dbg[l];
                o.time = ra.w; o.ray_o[0] = ra.x; o.ray_o[1] = ra.y; o.ray_o[2] = ra.z; o.ray_d[0] = rb.x; o.ray_d[1] = rb.y; o.ray_d[2] = rb.z;
            }
            if (have_memo) instance_memo_fill(sv, ra.w, memo_m, memo_inv);
            Hit h;
            bool found = flat ? trace_flat<false, true>(sv, (ConstBytes) A.scene + rp.flat_off, rp.flat_off, flat, stack, mk(ra.x, ra.y, ra.z), mk(rb.x, rb.y, rb.z), ra.w, rb.w, h)
                              : trace_scene<false, MESH, FUSED, RESW != 0, kStackStride, S16, 0u, false, RH16>(sv, stack, mk(ra.x, ra.y, ra.z), mk(rb.x, rb.y, rb.z), ra.w, rb.w, h);
            hh = make_uint4(f2u(h.t), f2u(h.u), f2u(h.v), h.prim);
            hid = found ? (h.obj | (h.shape << q.id_shift)) : 0xffffffffu;
        }
    }
    const uint32_t n_inline = FIRST ? rp.inline_iters : 1u;
    for (uint32_t it = 0; ; ++it) {
    const uint32_t depth = depth0 + it;
    const bool last = it + 1 >= n_inline;                 // uniform
    const uint32_t trace_next = last ? trace_next_last : 1u;
    alive = false; want_shadow = false;
    if (lane_on) {
        if (PARK) park_load();
        if (!FIRST) {
            hid = q.hit_id[l];
            if (hid != 0xffffffffu) {
                ra = q.ray_a[l]; rb = q.ray_b[l]; hh = load_hit<MESH>(q, l); st = q.st_a[l];
                const uint4 rs = q.rng_a[l]; const uint2 ri = q.rng_b[l];
                main.state = (uint64_t) rs.x | ((uint64_t) rs.y << 32); main.inc = ((uint64_t) ri.x << 1) | 1u;
                path.state = (uint64_t) rs.z | ((uint64_t) rs.w << 32); path.inc = ((uint64_t) ri.y << 1) | 1u;
            }
        }
        if (hid == 0xffffffffu && rp.n_passes > 1) {
            // several passes: the streams are carried into the next pass, so the six draws the reference makes for EVERY lane that is
            // active at the entry of an iteration (App. A step 5; both streams advance on each, correlated.cpp:156-161) also happen
            // for the lanes whose ray misses (single-pass renders drop the state of a finished path instead)
            if (!FIRST) {
                const uint4 rs = q.rng_a[l]; const uint2 ri = q.rng_b[l];
                main.state = (uint64_t) rs.x | ((uint64_t) rs.y << 32); main.inc = ((uint64_t) ri.x << 1) | 1u;
                path.state = (uint64_t) rs.z | ((uint64_t) rs.w << 32); path.inc = ((uint64_t) ri.y << 1) | 1u;
            }
            const bool one = rp.integrator != 0 || rp.sampler_kind != SAMPLER_CORRELATED;
            main.state = pcg_jump6(main.state, main.inc); if (!one) path.state = pcg_jump6(path.state, path.inc);   // six steps at once: the same integers
            q.rng_a[l] = make_uint4((uint32_t) main.state, (uint32_t) (main.state >> 32), (uint32_t) path.state, (uint32_t) (path.state >> 32));
        }
        // valid_ray of the lane as it enters this iteration
        const bool valid_in = FIRST ? valid_reg : (SPEC ? (depth > 0 ? q.st_c[l].y >= 2.f : valid_start) : depth > 0);
        if (hid == 0xffffffffu) {   // the path ends here: nothing validates it any more
            if (SPEC && !valid_in && depth > 0) {   // select(valid_ray, result, 0) (:279-282): what the path gathered behind null interactions does not count
                if (RES_MEM) res_clear();
                else {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) res_put(k, zero4);
                }
            }
            if (rp.want_valid) q.valid_out[l] = make_float4(valid_in ? 1.f : 0.f, 0.f, 0.f, 0.f);
        }
        if (SPEC && hid == 0xffffffffu && rp.has_env && valid_in) {
            // The ray left the scene: si.emitter(scene) is the environment (dopplertofpath.cpp:150-168).  DirectionSample(scene, si, prev_si)
            // points along the ray; ConstantBackgroundEmitter::pdf_direction is the uniform-sphere density (constant.cpp:150-155).  A path that
            // is not valid here (hidden emitters, and nothing but null interactions so far -- a primary ray above all) returns 0 whatever it adds
            // (valid_ray, :101-102,279-282).
            const float4 stv = FIRST ? st : q.st_a[l];
            const float time_ = FIRST ? ra.w : q.ray_a[l].w;
            float prev_pdf = 1.f; bool pdelta = true;
            if (depth > 0) { prev_pdf = FIRST ? stb_reg.w : q.st_b[l].w; pdelta = FIRST ? stc_reg.y != 0.f : ((uint32_t) q.st_c[l].y & 1u) != 0u; }
            const DEmitter &env = sv.emitters[rp.env_index];
            const bool is_map = env.kind == EMITTER_ENVMAP;   // EnvironmentMapEmitter::pdf_direction / eval (envmap.cpp:408-425,299-310) with ds.d = -si.wi = the ray direction
            const V3 rd = FIRST ? mk(rb.x, rb.y, rb.z) : [&] { const float4 b4 = q.ray_b[l]; return mk(b4.x, b4.y, b4.z); }();
            const float em_pdf = pdelta ? 0.f : (is_map ? env_pdf_direction(sv.base, env, rd) : kInvFourPi) * (1.f / (float) sv.n_emitters);
            const float mis_bsdf = mis_weight(prev_pdf, em_pdf);
            const V3 le = prev_pdf > 0.f ? (is_map ? env_eval(sv.base, env, rd) : mk(env.intensity[0], env.intensity[1], env.intensity[2])) : mk(0, 0, 0);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) {
                V3 v = le * mis_bsdf;
                if (rp.integrator == 0) v = v * modulation_weight(rp, rp.phase[k], time_, stv.w);
                const float4 r = res_get(k);
                res_put(k, make_float4(fmaf(stv.x, v.x, r.x), fmaf(stv.y, v.y, r.y), fmaf(stv.z, v.z, r.z), 0.f));
            }
            if (RES_MEM) res_live = true;
        }
        if (hid != 0xffffffffu) {   // a miss ends the path (active_next = false, dopplertofpath.cpp:171)
            V3 o = mk(ra.x, ra.y, ra.z), d = mk(rb.x, rb.y, rb.z); float time = ra.w;
            V3 thr = mk(st.x, st.y, st.z); float path_length = st.w;
            float eta_path = 1.f; bool prev_delta = depth == 0;   // dopplertofpath.cpp:103-108: eta = 1, prev_bsdf_delta = true
            if (SPEC && depth > 0) { const float2 sc = FIRST ? stc_reg : q.st_c[l]; eta_path = sc.x; prev_delta = FIRST ? sc.y != 0.f : ((uint32_t) sc.y & 1u) != 0u; }
            bool correlate = (depth + 1) < rp.path_correlation_depth;
            const bool plain = rp.integrator != 0;   // `path`: no modulation weight
            const bool single = plain || rp.sampler_kind != SAMPLER_CORRELATED;   // main stream only (path.cpp:197,213-214,273; sampler.h:141-144)
            float t = u2f(hh.x);
            path_length += t * eta_path;   // dopplertofpath.cpp:141 (eta stays 1 without dielectrics)
            bool active_next = depth + 1 < rp.max_depth;

            Surface si;
            if (!FIRST && have_memo) instance_memo_fill(sv, time, memo_m, memo_inv);
            if (FIRST && it > 0 && have_memo && !sv.memo_m) {   // a ray's time does not change along its path: the inverse filled at generation still sits in the LDS column
                instance_matrix(sv.objects[sv.memo_obj], time, memo_m); instance_memo_load(sv, memo_inv);
            }
            compute_surface<MESH>(sv, hid & ((1u << q.id_shift) - 1u), hid >> q.id_shift, hh.w, t, u2f(hh.y), u2f(hh.z), o, d, time, si, have_memo, memo_m, memo_inv);
            const DShape *sh = si.shape;

            const float pmf = sv.n_emitters ? 1.f / (float) sv.n_emitters : 0.f;   // m_emitter_pmf (scene.cpp:96)
            // ---- direct emission (dopplertofpath.cpp:150-168 / path.cpp): the hit shape carries an area emitter
            bool res_dirty = false;
            float4 rcur[KREG];   // (RES_MEM: the emitter-hit term goes straight to q.res)
            if (AREA) {
                if (!RES_MEM) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) rcur[RES_MEM ? 0 : k] = res_get(k);
                }
                if (sh->flags & SF_EMITTER) {
                    float4 pb = depth > 0 ? (FIRST ? stb_reg : q.st_b[l]) : make_float4(0.f, 0.f, 0.f, 1.f);   // prev_si.p, prev_bsdf_pdf
                    V3 rel = si.p - mk(pb.x, pb.y, pb.z);                      // DirectionSample(scene, si, prev_si), records.h:173-180
                    float dist = norm(rel);
                    V3 dsd = rel * rcp(dist);
                    float em_pdf = 0.f;
                    if (!prev_delta) {                                          // !prev_bsdf_delta: AreaLight::pdf_direction (area.cpp:161-180)
                        float dp = dot(dsd, si.sh_n);   // ds.n = si.sh_frame.n (PositionSample(si), records.h:63-65)
                        if (SPEC && dp < 0.f && sh->tex_radiance) {   // area.cpp:170-176: pdf_position of the texture at ds.uv = si.uv, through the parameterisation's |dp_du x dp_dv|
                            V3 pp, pn; float su, sv_, area_norm;
                            if (rect_eval_parameterization(*sh, si.u, si.v, pp, pn, su, sv_, area_norm))
                                em_pdf = texture_pdf_position(sv, sh->tex_radiance << 4, si.u, si.v) * sqr(dist) / (area_norm * -dp) * pmf;
                        } else
                        if (dp < 0.f) {
                            const float adp = fabsf(dp);
                            const float pdf = MESH && sh->kind == SHAPE_SPHERE ? sphere_pdf_direction(*sh, mk(pb.x, pb.y, pb.z), dsd, si.sh_n, dist)
                                                                               : sh->inv_area * (adp != 0.f ? (dist * dist) / adp : 0.f);
                            em_pdf = pdf * pmf;
                        }
                    }
                    float mis_bsdf = mis_weight(pb.w, em_pdf);
                    bool on = si.wi.z > 0.f && pb.w > 0.f;                       // AreaLight::eval (area.cpp:82-89), mask prev_bsdf_pdf > 0
                    V3 le = on ? mk(sh->radiance[0], sh->radiance[1], sh->radiance[2]) : mk(0, 0, 0);
                    if (SPEC && on && sh->tex_radiance) le = texture_eval(sv, sh->tex_radiance << 4, si.u, si.v);   // m_radiance->eval(si)
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) {
                        V3 v = le * mis_bsdf;
                        if (!plain) v = v * modulation_weight(rp, rp.phase[k], time, path_length);
                        const float4 r = RES_MEM ? res_get(k) : rcur[RES_MEM ? 0 : k];
                        const float4 acc = make_float4(fmaf(thr.x, v.x, r.x), fmaf(thr.y, v.y, r.y), fmaf(thr.z, v.z, r.z), 0.f);
                        if (RES_MEM) res_put(k, acc); else rcur[RES_MEM ? 0 : k] = acc;
                    }
                    if (RES_MEM) res_live = true; else res_dirty = true;
                }
            }

            // ---- emitter sampling (scene.cpp:235-291; point.cpp:118-147; area.cpp:116-159 + shape.cpp:370-384 + rectangle.cpp:152-166)
            // The six draws of this iteration (App. A step 5) come from ONE stream: the main one (`path`, other samplers, correlate = false) or
            // the path-correlated one; next_1d_correlate advances both on every draw (correlated.cpp:156-161), so the stream that is not read
            // is moved six steps at once at the end (pcg_jump6: the same integers as six single steps).
            const bool use_path = !single && correlate;
            Rng sel = use_path ? path : main;
            float e1 = next_f32(sel), e2 = next_f32(sel);
            // has_flag(bsdf->flags(), BSDFFlags::Smooth) (:178): diffuse, (rough)plastic and roughconductor have a smooth lobe
            bool active_em = active_next && sv.n_emitters > 0 && (!SPEC || bsdf_is_smooth(sh->bsdf) || ((sh->flags & (SF_BLEND | SF_TWOSIDED2)) && bsdf_is_smooth(sv.shapes[sh->blend_other].bsdf)));   // a blend (a twosided of two BSDFs) has the flags of both
            V3 em_weight = mk(0, 0, 0), wo = mk(0, 0, 0); float ds_dist = 0.f, ds_pdf = 0.f; bool ds_delta = true;
            if (active_em) {
                uint32_t ne = sv.n_emitters, idx = 0; float em_w = 1.f, sx = e1;
                if (ne > 1) { float scaled = e1 * (float) ne; idx = (uint32_t) scaled; if (idx > ne - 1) idx = ne - 1; em_w = (float) ne; sx = scaled - (float) idx; }
                const DEmitter &em = sv.emitters[idx];
                V3 dsp, dd; bool em_active = true;
                if ((!AREA && !SPEC) || em.kind == EMITTER_POINT) {   // scenes without surface emitters that run the diffuse-only kernels have point lights only (render_rows)
                    dsp = mk(em.pos[0], em.pos[1], em.pos[2]);
                    dd = dsp - si.p;
                    float dist2 = dot(dd, dd), inv_dist = rsqrt_(dist2);
                    ds_dist = sqrtf(dist2);
                    dd = dd * inv_dist;
                    float id2 = sqr(inv_dist);
                    em_weight = mk(em.intensity[0] * id2, em.intensity[1] * id2, em.intensity[2] * id2);
                    ds_pdf = 1.f;
                } else if (SPEC && em.kind == EMITTER_CONSTANT) {   // ConstantBackgroundEmitter::sample_direction (constant.cpp:118-148)
                    dd = uniform_sphere(sx, e2);
                    const float radius = fmax_(em.cutoff_angle, norm(si.p - mk(em.pos[0], em.pos[1], em.pos[2])));   // m_bsphere, enlarged to hold the reference point
                    ds_dist = 2.f * radius;
                    dsp = vfma(dd, ds_dist, si.p);
                    ds_pdf = kInvFourPi; ds_delta = false;
                    const float ip = rcp(ds_pdf);
                    em_weight = mk(em.intensity[0] * ip, em.intensity[1] * ip, em.intensity[2] * ip);
                } else if (SPEC && em.kind == EMITTER_DIRECTIONAL) {   // DirectionalEmitter::sample_direction (directional.cpp:148-176)
                    const V3 dir = mk(em.to_local[0], em.to_local[1], em.to_local[2]);
                    const float radius = fmax_(em.cutoff_angle, norm(si.p - mk(em.pos[0], em.pos[1], em.pos[2])));
                    ds_dist = 2.f * radius;
                    dsp = si.p - dir * ds_dist;
                    dd = -dir;
                    ds_pdf = 1.f;
                    em_weight = mk(em.intensity[0], em.intensity[1], em.intensity[2]);
                } else if (SPEC && em.kind == EMITTER_ENVMAP) {   // EnvironmentMapEmitter::sample_direction (envmap.cpp:363-406)
                    env_sample_direction(sv.base, em, si.p, sx, e2, dd, ds_dist, ds_pdf, em_weight, em_active);
                    dsp = si.p + dd * ds_dist;
                    ds_delta = false;
                } else if (SPEC && em.kind == EMITTER_SPOT) {   // SpotLight::sample_direction (spot.cpp:152-187), falloff_curve (:116-126)
                    dsp = mk(em.pos[0], em.pos[1], em.pos[2]);
                    dd = dsp - si.p;
                    ds_dist = norm(dd);
                    const float inv_dist = rcp(ds_dist);
                    dd = dd * inv_dist;
                    const V3 local = normalize(xf_vector(em.to_local, -dd));
                    const float cos_theta = local.z;
                    const float beam = cos_theta >= em.cos_beam ? 1.f : (em.cutoff_angle - acos_(cos_theta)) * em.inv_transition;
                    const float falloff = cos_theta > em.cos_cutoff ? beam : 0.f;
                    const float k = falloff * sqr(inv_dist);
                    em_weight = falloff > 0.f ? mk(em.intensity[0] * k, em.intensity[1] * k, em.intensity[2] * k) : mk(0, 0, 0);
                    ds_pdf = 1.f;
                } else {
                    const DShape &es = sv.shapes[em.shape];
                    V3 en;
                    if (SPEC && es.tex_radiance) {
                        // AreaLight::sample_direction with a spatially varying radiance (area.cpp:129-153): the TEXTURE is sampled (Texture::sample_position), the shape maps the
                        // uv to a point (Rectangle::eval_parameterization), the density goes from uv space to solid angle with |dp_du x dp_dv|
                        float tu, tv, tpdf, su = 0.f, sv_ = 0.f, area_norm = 1.f;
                        texture_sample_position(sv, es.tex_radiance << 4, sx, e2, tu, tv, tpdf);
                        V3 pp = si.p; en = mk(0.f, 0.f, 1.f);
                        const bool valid = tpdf != 0.f && rect_eval_parameterization(es, tu, tv, pp, en, su, sv_, area_norm);
                        dsp = valid ? pp : si.p;
                        dd = dsp - si.p;
                        const float dist2 = dot(dd, dd);
                        ds_dist = sqrtf(dist2);
                        dd = dd * rcp(ds_dist);
                        const float dp = dot(dd, en);
                        em_active = valid && dp < 0.f;
                        ds_pdf = em_active ? tpdf / area_norm * dist2 / -dp : 0.f;
                        ds_delta = false;
                        const V3 c = em_active ? texture_eval(sv, es.tex_radiance << 4, su, sv_) : mk(0, 0, 0);   // m_radiance->eval(si) / ds.pdf
                        em_weight = em_active ? mk(c.x / ds_pdf, c.y / ds_pdf, c.z / ds_pdf) : mk(0, 0, 0);
                    } else {
                    if (MESH && es.kind == SHAPE_SPHERE) {   // Sphere overrides Shape::sample_direction
                        sphere_sample_direction(es, si.p, sx, e2, dsp, en, dd, ds_dist, ds_pdf);
                    } else {
                        if (!MESH || es.kind == SHAPE_RECT) {
                            dsp = xf_point(es.to_world, mk(sx * 2.f - 1.f, e2 * 2.f - 1.f, 0.f));
                            en = mk(es.n[0], es.n[1], es.n[2]);
                        } else if (es.kind == SHAPE_DISK) {   // Disk::sample_position (disk.cpp:158-177)
                            float px, py; concentric_disk(sx, e2, px, py);
                            dsp = xf_point(es.to_world, mk(px, py, 0.f));
                            en = mk(es.n[0], es.n[1], es.n[2]);
                        } else mesh_sample_position(sv, es, sx, e2, dsp, en);
                        dd = dsp - si.p;
                        float dist2 = dot(dd, dd);
                        ds_dist = sqrtf(dist2);
                        dd = dd * rcp(ds_dist);
                        float dp = fabsf(dot(dd, en)), x = dist2 / dp;
                        ds_pdf = es.inv_area * (isfinite(x) ? x : 0.f);
                    }
                    ds_delta = false;
                    em_active = dot(dd, en) < 0.f && ds_pdf != 0.f;
                    float ip = rcp(ds_pdf);
                    em_weight = em_active ? mk(em.intensity[0] * ip, em.intensity[1] * ip, em.intensity[2] * ip) : mk(0, 0, 0);
                    }
                }
                ds_pdf *= pmf; em_weight = em_weight * em_w;
                active_em = ds_pdf != 0.f && em_active;
                // Interaction::spawn_ray_to (interaction.h:141-149)
                V3 so = offset_p(si, dsp - si.p);
                V3 sd = dsp - so;
                float sdist = norm(sd);
                sd = sd * rcp(sdist);
                sha = make_float4(so.x, so.y, so.z, sdist * (1.f - kShadowEps));
                shb = make_float4(sd.x, sd.y, sd.z, time);
                wo = mk(dot(dd, si.sh_s), dot(dd, si.sh_t), dot(dd, si.sh_n));
            }
            float sample_1 = next_f32(sel); (void) sample_1;
            float s2x = next_f32(sel), s2y = next_f32(sel);

            // ---- the shape's BSDF with its adapters (mask, blendbsdf, twosided, normalmap / bumpmap): bsdf_eval_pdf_sample above
            BsdfOut bo;
            bsdf_eval_pdf_sample<SPEC>(sv, sh, si, wo, active_em, sample_1, s2x, s2y, bo);
            const V3 bsdf_val = bo.val, bs_wo = bo.wo; V3 bsdf_weight = bo.weight;
            const float bsdf_pdf = bo.pdf, bs_pdf = bo.bs_pdf, bs_eta = bo.bs_eta; const bool bs_delta = bo.bs_delta, bs_null = bo.bs_null;
            // ---- emitter contribution candidate (dopplertofpath.cpp:214-226); committed by k_shadow if unoccluded
            if (active_em) {
                const float mis_em = ds_delta ? 1.f : mis_weight(ds_pdf, bsdf_pdf);   // dopplertofpath.cpp:218-219
                bool nonzero = false;
                if (RES_MEM) {   // the sample waits for its visibility test WITHOUT the K modulation weights; they are applied at the commit (below), in the order of the loop that follows
                    pend_thr = thr; pend_len = path_length + ds_dist;
                    pend_v = mk(bsdf_val.x * em_weight.x * mis_em, bsdf_val.y * em_weight.y * mis_em, bsdf_val.z * em_weight.z * mis_em);
                    // a sample whose three products are zero adds exactly zero to every film whatever its weights: no visibility test (the register form compares the
                    // candidate with the current result, which also catches a zero weight and a term below the result's last bit -- a few more shadow rays here, same films)
                    nonzero = (int) (pend_thr.x * pend_v.x != 0.f) | (int) (pend_thr.y * pend_v.y != 0.f) | (int) (pend_thr.z * pend_v.z != 0.f);
                } else {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) {
                    float4 r = AREA ? rcur[RES_MEM ? 0 : k] : res_get(k);
                    V3 v = mk(bsdf_val.x * em_weight.x * mis_em, bsdf_val.y * em_weight.y * mis_em, bsdf_val.z * em_weight.z * mis_em);
                    if (!plain) { float lw = modulation_weight(rp, rp.phase[k], time, path_length + ds_dist); v = v * lw; }
                    float3 c = make_float3(fmaf(thr.x, v.x, r.x), fmaf(thr.y, v.y, r.y), fmaf(thr.z, v.z, r.z));
                    cand[RES_MEM ? 0 : k] = c;
                    nonzero |= f2u(c.x) != f2u(r.x) || f2u(c.y) != f2u(r.y) || f2u(c.z) != f2u(r.z);
                }
                }
                want_shadow = nonzero;   // a candidate identical to the current result needs no visibility test
            }
            if (res_dirty) {   // the emitter-hit term stands whether or not the NEE candidate is later committed
#pragma unroll
                for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) res_put(k, rcur[RES_MEM ? 0 : k]);
            }
            // ---- continuation (dopplertofpath.cpp:232-276)
            V3 nd = vfma(si.sh_n, bs_wo.z, vfma(si.sh_t, bs_wo.y, si.sh_s * bs_wo.x));   // Frame::to_world
            V3 no = offset_p(si, nd);
            thr = mk(thr.x * bsdf_weight.x, thr.y * bsdf_weight.y, thr.z * bsdf_weight.z);
            const float eta = eta_path * bs_eta;   // eta *= bs.eta (:252); bs.eta = 0 for the zero-initialised sample when cos_theta_i <= 0
            uint32_t ndepth = depth + 1;
            float thr_max = fmax_(fmax_(thr.x, thr.y), thr.z);
            float rr_prob = fmin_(thr_max * sqr(eta), .95f);
            bool rr_active = ndepth >= rp.rr_depth;
            bool rr_continue = next_f32(sel) < rr_prob;
            if (use_path) { path.state = sel.state; main.state = pcg_jump6(main.state, main.inc); }
            else { main.state = sel.state; if (!single) path.state = pcg_jump6(path.state, path.inc); }
            if (rr_active) thr = thr * rcp(rr_prob);
            alive = active_next && (!rr_active || rr_continue) && thr_max != 0.f;
            // valid_ray |= active && si.is_valid() && !has_flag(bsdf_sample.sampled_type, BSDFFlags::Null) (:252-253)
            const bool valid_now = !SPEC || valid_in || !bs_null;
            const bool ends = !alive || (last && !trace_next);   // nobody continues this path: what it returns is decided here
            if (SPEC && ends && !valid_now) {   // select(valid_ray, result, 0) (:279-282): neither the emitter-hit term nor this vertex's NEE candidate survives
                want_shadow = false;
                if (RES_MEM) res_clear();
                else {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) res_put(k, zero4);
                }
            }
            if (ends && rp.want_valid) q.valid_out[l] = make_float4(valid_now ? 1.f : 0.f, 0.f, 0.f, 0.f);
            if (FIRST) valid_reg = valid_now;
            if (alive) {
                nra = make_float4(no.x, no.y, no.z, time); nrb = make_float4(nd.x, nd.y, nd.z, kLargest);
                const float4 sta = make_float4(thr.x, thr.y, thr.z, path_length), stb = make_float4(si.p.x, si.p.y, si.p.z, bs_pdf);
                const float2 stc = make_float2(eta, bs_delta ? 1.f : 0.f);
                if ((!FIRST || last) && trace_next) {   // the state leaves for the queues (an inline iteration keeps it in registers; after the last iteration of the loop nobody reads it)
                    q.ray_a[l] = nra;
                    q.ray_b[l] = nrb;
                    q.st_a[l] = sta;
                    if (AREA) q.st_b[l] = stb;   // prev_si, prev_bsdf_pdf (:256-257)
                    if (SPEC) q.st_c[l] = make_float2(stc.x, stc.y + (valid_now ? 2.f : 0.f));   // eta, prev_bsdf_delta (:252,258) | valid_ray << 1
                }
                if (FIRST) { st = sta; stb_reg = stb; stc_reg = stc; }
                if (PARK) park_store();
                DTOF_POISON_CHECK();   // the state of a path that continues
            }
            if ((alive && (!FIRST || last) && trace_next) || rp.n_passes > 1)   // several passes: the state of a finished path is what its lane starts the next pass with
                q.rng_a[l] = make_uint4((uint32_t) main.state, (uint32_t) (main.state >> 32), (uint32_t) path.state, (uint32_t) (path.state >> 32));
        }
    }
    if (last) {
        uint32_t slot = block_append(alive, s_cnt, n_alive);
        if (alive && trace_next) qout[seg * kSeg + slot] = l;
    } else {   // FIRST, one wave per block
        const uint32_t n_on = (uint32_t) __popcll(__ballot(alive));
        if (lane_id == 0) s_inline[2 * it] += n_on;
    }
    if (FUSED) {
        bool commit = false;
        // (Both rays in ONE traversal loop -- a lane going on with its continuation ray while its neighbours are still in their shadow rays -- was built and measured in
        // round 5 and lost by a third: tools/experiments/r05_pair_traversal.patch, profiles/r05_pair_traversal.txt.)
        if (want_shadow) {   // test_visibility (scene.cpp:266-271): an unoccluded sample commits its candidate result
            Hit hs;
#if defined(DTOF_ABLATE) && (DTOF_ABLATE & 1)
            commit = sha.w > 0.f;
#else
            commit = flat ? !trace_flat<true, true>(sv, (ConstBytes) A.scene + rp.flat_off, rp.flat_off, flat, stack, mk(sha.x, sha.y, sha.z), mk(shb.x, shb.y, shb.z), shb.w, sha.w, hs)
                          : !trace_scene<true, MESH, true, RESW != 0, kStackStride, S16, 0u, false, RH16>(sv, stack, mk(sha.x, sha.y, sha.z), mk(shb.x, shb.y, shb.z), shb.w, sha.w, hs);
#endif
        }
        if (RES_MEM) {   // the committed sample gets its K modulation weights now (dopplertofpath.cpp:221-226) and is added to the films' records in q.res
            if (commit) {
                const bool plain_ = rp.integrator != 0;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) if (k < rp.n_offsets) {
                    V3 v = pend_v;
                    if (!plain_) v = v * modulation_weight(rp, rp.phase[k], shb.w, pend_len);   // shb.w: the ray time (the shadow ray carries it)
                    const float4 r = res_get(k);
                    res_put(k, make_float4(fmaf(pend_thr.x, v.x, r.x), fmaf(pend_thr.y, v.y, r.y), fmaf(pend_thr.z, v.z, r.z), 0.f));
                }
                res_live = true;
            }
            if (RES_LDS) {   // the results leave the LDS columns for q.res (what the splat kernels read)
                if (last && in_range) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) if (k < rp.n_offsets) q.res[(size_t) k * q.capacity + l] = res_get(k);
                }
            } else if (last && in_range && !res_live) {   // nothing was ever added (or a path of null interactions was zeroed): the records still hold an earlier batch's values
#pragma unroll
                for (int k = 0; k < KMAX; ++k) if (k < rp.n_offsets) res_put(k, zero4);
            }
        } else if (FIRST) {   // the running result stays in rbase over the inline iterations; every lane's result is defined after the last (nothing zeroed it)
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) {
                if (commit) rbase[RES_MEM ? 0 : k] = cand[RES_MEM ? 0 : k];
                if (last && in_range && !fuse_splat) q.res[(size_t) k * q.capacity + l] = make_float4(rbase[RES_MEM ? 0 : k].x, rbase[RES_MEM ? 0 : k].y, rbase[RES_MEM ? 0 : k].z, 0.f);
            }
        } else if (commit) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) q.res[(size_t) k * q.capacity + l] = make_float4(cand[RES_MEM ? 0 : k].x, cand[RES_MEM ? 0 : k].y, cand[RES_MEM ? 0 : k].z, 0.f);
        }
        if (alive && trace_next) {   // closest hit of the continuation ray, consumed by the next bounce (the Hit lives inside this block: no half-defined registers across the commit above)
            Hit h;
#if defined(DTOF_ABLATE) && (DTOF_ABLATE & 2)
            bool found = nra.x < 1e30f; h.t = 0.5f + 0.1f * nrb.x; h.u = nrb.y; h.v = nrb.z; h.obj = nrb.x > 0.3f ? 3 : nrb.y > 0.f ? 1 : 0; h.shape = 0; h.prim = 0;
#else
            bool found = flat ? trace_flat<false, true>(sv, (ConstBytes) A.scene + rp.flat_off, rp.flat_off, flat, stack, mk(nra.x, nra.y, nra.z), mk(nrb.x, nrb.y, nrb.z), nra.w, nrb.w, h)
                              : trace_scene<false, MESH, true, RESW != 0, kStackStride, S16, 0u, false, RH16>(sv, stack, mk(nra.x, nra.y, nra.z), mk(nrb.x, nrb.y, nrb.z), nra.w, nrb.w, h);
#endif
            if (!FIRST || last) store_hit<MESH>(q, l, h, found);
            if (FIRST) { hh = make_uint4(f2u(h.t), f2u(h.u), f2u(h.v), h.prim); hid = found ? (h.obj | (h.shape << q.id_shift)) : 0xffffffffu; }
        }
        const uint32_t n_sh = (uint32_t) __popcll(__ballot(want_shadow)) * ((threadIdx.x & 63) == 0 ? 1u : 0u);   // per-wave partial (stats only)
        if (last) n_shadow += n_sh; else if (lane_id == 0) s_inline[2 * it + 1] += n_sh;
    } else {
        uint32_t sslot = seg * kSeg + block_append(want_shadow, s_cnt, n_shadow);
        if (want_shadow) {
            q.sh_a[sslot] = sha; q.sh_b[sslot] = shb;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets)
                q.sh_c[(size_t) k * q.capacity + sslot] = make_float4(cand[RES_MEM ? 0 : k].x, cand[RES_MEM ? 0 : k].y, cand[RES_MEM ? 0 : k].z, u2f(l));
        }
    }
    if (last) break;
    // next inline iteration: the continuation ray and its closest hit become the lane's current ray (st / stb / stc were set where the
    // bounce computed them); a lane whose path ended sits out the remaining iterations
    lane_on = alive;
    if (alive) { ra = nra; rb = nrb; DTOF_POISON_CHECK(); }
    }   // inline iterations
    if constexpr (!RES_MEM) if (FIRST && fuse_splat) {   // ---- ImageBlock::put (imageblock.cpp:414-531) of the wave's samples: tent filter of radius <= 1, a 3 x 3 footprint anchored at the sample's pixel
        const uint32_t W = (uint32_t) rp.crop_w;
        const uint32_t pix = fdiv(global_lane(rp, rp.lane_base + l), rp.d_spp);
        const int py = (int) fdiv(pix, rp.d_w), px = (int) (pix - W * (uint32_t) py);
        const float sx = pos_reg.x, sy = pos_reg.y;
        const bool regular = in_range && (int) floorf(sx) - rp.crop_x == px && (int) floorf(sy) - rp.crop_y == py;   // (rarely a float position rounds up into the next pixel)
        if (wave_pixel) {   // one pixel per wave: 36 footprint sums over the 64 lanes, lane L < 36 ends up with sum L and adds it to the film
            const int upx = __builtin_amdgcn_readfirstlane(px), upy = __builtin_amdgcn_readfirstlane(py);
            const float relx = (float) (upx + rp.crop_x - 1) + .5f - sx, rely = (float) (upy + rp.crop_y - 1) + .5f - sy;
            float wx[3], wy[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { wx[a] = regular ? tent(relx + (float) a, rp.inv_radius) : 0.f; wy[a] = tent(rely + (float) a, rp.inv_radius); }
            const uint32_t sum_id = __brev(lane_id) >> 26, tap = sum_id >> 2, ch = sum_id & 3u;   // which of the 36 totals this lane ends up with (wave_totals_36)
            const int fx = upx - 1 + (int) (tap % 3u), fy = upy - 1 + (int) (tap / 3u);
            const bool store = sum_id < 36u && (unsigned) fx < W && (unsigned) fy < (unsigned) rp.crop_h;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) {
                float *fk = A.film + (size_t) k * A.film_stride;
                if (in_range && !regular) splat_lane(rp, fk, sx, sy, px, py, rbase[k].x, rbase[k].y, rbase[k].z);
                float v[36];
#pragma unroll
                for (int ys = 0; ys < 3; ++ys)
#pragma unroll
                    for (int xs = 0; xs < 3; ++xs) {
                        const float w = wx[xs] * wy[ys]; const int c = 4 * (3 * ys + xs);
                        v[c] = rbase[k].x * w; v[c + 1] = rbase[k].y * w; v[c + 2] = rbase[k].z * w; v[c + 3] = w;
                    }
                const float total = wave_totals_36(v, lane_id);
                if (store && total != 0.f) atomicAdd(fk + 4 * ((size_t) fy * W + (size_t) fx) + ch, total);
            }
        } else if (in_range) {   // a ragged chunk: every lane by itself
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets)
                splat_lane(rp, A.film + (size_t) k * A.film_stride, sx, sy, px, py, rbase[k].x, rbase[k].y, rbase[k].z);
        }
    }
    }   // chunk loop
    }   // count != 0
    if (FUSED && kShadeBlock > 64) {   // shadow-ray count for the statistics: sum the four per-wave partials
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = n_shadow;
        __syncthreads();
        n_shadow = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
    if (lane_id == 0) {
        if (sub > 1) { atomicAdd(&A0.alive_out[seg], n_alive); atomicAdd(&A0.shadow_out[seg], n_shadow); }
        else { A0.alive_out[seg] = n_alive; A0.shadow_out[seg] = n_shadow; }
        if (FIRST) {   // the count slots of the inline iterations before the last lie 2 * n_seg words apart below the last one's (render_rows)
            const uint32_t n_inl = A0.rp.inline_iters, n_seg = RESW ? A0.n_seg : gridDim.x / sub;
            for (uint32_t i = 0; i + 1 < n_inl; ++i) {
                uint32_t *slot = A0.alive_out - (size_t) 2 * (n_inl - 1 - i) * n_seg;
                if (sub > 1) { atomicAdd(&slot[seg], s_inline[2 * i]); atomicAdd(&slot[n_seg + seg], s_inline[2 * i + 1]); }
                else { slot[seg] = s_inline[2 * i]; slot[n_seg + seg] = s_inline[2 * i + 1]; }
            }
        }
    }
    if (!RESW) break;
    }   // segments of a resident wave
#undef DTOF_POISON_CHECK
}

// ---------------------------------------------------------------------------- launch templates (one group of instantiations per dtof_shade_*.hip)
// the six kernels of one (AREA, KMAX, MESH, SPEC) variant: staged / unstaged x split / fused / fused first bounce
template <bool A, int K, bool M, int S>
static void launch_shade_variant(const ShadeLaunch &L) {
#define DTOF_LAUNCH_ONE(LDS_, MODE_) hipLaunchKernelGGL((k_shade<LDS_, MODE_, A, K, M, S>), dim3(L.grid), dim3(kShadeBlock), L.lds, L.stream, L.args)
    if (L.staged) { if (L.mode == 2) DTOF_LAUNCH_ONE(true, 2); else if (L.mode == 1) DTOF_LAUNCH_ONE(true, 1); else DTOF_LAUNCH_ONE(true, 0); }
    else          { if (L.mode == 2) DTOF_LAUNCH_ONE(false, 2); else if (L.mode == 1) DTOF_LAUNCH_ONE(false, 1); else DTOF_LAUNCH_ONE(false, 0); }
#undef DTOF_LAUNCH_ONE
}
// the resident form of the fused first-bounce kernel: `waves` waves per block, one block per CU.  Its dynamic LDS lies above the 64 KiB a kernel gets
// without asking: hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE, so the high-water mark is kept per device ordinal (dtof-render drives
// one host thread per GPU in one process).
template <bool A, int K, int S, int W>
static void launch_resident_waves(const ShadeLaunch &L) {
    static std::atomic<uint32_t> attr_lds[64];
    int dev = 0; (void) hipGetDevice(&dev);
    std::atomic<uint32_t> &mark = attr_lds[(unsigned) dev & 63u];
    if (L.lds > mark.load()) {
        if (hipFuncSetAttribute((const void *) k_shade<false, 2, A, K, true, S, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) L.lds) != hipSuccess ||
            hipFuncSetAttribute((const void *) k_shade<false, 2, A, K, true, S, W, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) L.lds) != hipSuccess)
            throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        mark.store(L.lds);
    }
    if (L.args.rp.res_half) hipLaunchKernelGGL((k_shade<false, 2, A, K, true, S, W, true>), dim3(L.grid), dim3(W * 64), L.lds, L.stream, L.args);   // a TLAS of 1 025 .. 2 048 nodes: half-float planes
    else hipLaunchKernelGGL((k_shade<false, 2, A, K, true, S, W>), dim3(L.grid), dim3(W * 64), L.lds, L.stream, L.args);
}
template <bool A, int K, int S>
static void launch_resident_variant(const ShadeLaunch &L) {
    if (L.waves == 16) launch_resident_waves<A, K, S, 16>(L);
    else if (L.waves == 12) launch_resident_waves<A, K, S, 12>(L);
    else if (L.waves == 8) launch_resident_waves<A, K, S, 8>(L);
    else throw std::runtime_error("resident stage: 8, 12 or 16 waves per block");
}

}  // namespace dtof
