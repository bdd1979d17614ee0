#pragma once
// dtof_sampling.h -- device side of the samplers and of the integrator's modulation: PCG streams, correlated / time draws
// (src/samplers/correlated.cpp), waveforms (include/mitsuba/render/waveform_utils.h), modulation weight
// (src/integrators/dopplertofpath.cpp:60-77) and the generation of a lane (sampler seeding, jitter, time, camera ray).
#include "dtof_kernels.h"
#include "dtof_scene.h"
#include "dtof_math.h"

#ifndef DTOF_D
#define DTOF_D __device__ __forceinline__
#endif

namespace dtof {

// ---------------------------------------------------------------------------- sampler
struct Rng { uint64_t state, inc; };
DTOF_D float next_f32(Rng &r) { return pcg_next_f32(r.state, r.inc); }
// PCG32Sampler::seed / CorrelatedSampler::seed -- sampler.cpp:115-134, correlated.cpp:38-64
DTOF_D Rng seed_stream(uint32_t seed_value, uint32_t index) {
    uint32_t v0, v1; tea32(seed_value, index, v0, v1);
    Rng r; pcg_seed(v0, v1, r.state, r.inc); return r;
}
DTOF_D uint64_t stream_inc(uint32_t seed_value, uint32_t index) {
    uint32_t v0, v1; tea32(seed_value, index, v0, v1);
    return ((uint64_t) v1 << 1) | 1u;
}
// next_1d_correlate -- correlated.cpp:156-161: both streams advance, the draw of one of them is returned.  The output permutation
// of PCG (xorshift, rotation, float conversion) is evaluated once, on the state of the selected stream.
DTOF_D float next_correlate(Rng &main, Rng &path, bool correlate) {
    uint64_t old = correlate ? path.state : main.state;
    path.state = path.state * kPcgMult + path.inc;
    main.state = main.state * kPcgMult + main.inc;
    return pcg_output_f32(old);
}
// next_1d_time -- correlated.cpp:92-153; si = current_sample_index (sampler.cpp:94-103)
DTOF_D float next_time(const RenderParams &rp, Rng &main, Rng &tm, uint32_t si, uint32_t perm_seed, uint32_t &dim) {
    int strategy = rp.time_sampling; uint32_t tcn = rp.tcn;
    if (strategy == TIME_UNIFORM) return next_f32(main);
    float r = strategy == TIME_STRATIFIED ? next_f32(main) : next_f32(tm);
    const uint32_t quo = fdiv(si, rp.d_tcn), rem = si - quo * tcn;   // si / tcn, si % tcn
    if (rp.stratify) {
        if (strategy == TIME_STRATIFIED) {
            // the reference evaluates p1 (seed + dim) and p2 (seed + dim + 1) and selects; the permutation is a pure function,
            // so only the selected one is computed
            const uint32_t ps = perm_seed + dim + ((rem != 0) ? 0u : 1u);
            dim += 2;
            const uint32_t p = permute_kensler(quo, rp.n_stratum, ps, rp.d_stratum);
            r = ((float) p + r) * rp.inv_n_stratum;
        } else {
            r = ((float) quo + r) * rp.inv_n_stratum;
        }
    }
    if (strategy == TIME_STRATIFIED) return ((float) rem + r) * rp.inv_tcn;
    if (strategy == TIME_ANTITHETIC) {
        if (tcn == 2) { float r2 = r + rp.antithetic_shift; return rem != 1 ? r : r2; }
        return r + (float) rem / (float) tcn;
    }
    if (strategy == TIME_ANTITHETIC_MIRROR) {   // Assert(m_time_correlate_number == 2) (:142): checked on the host
        float r2 = 1.0f - r + rp.antithetic_shift;
        return rem != 1 ? r : r2;
    }
    if (strategy == TIME_PERIODIC) return r + (float) rem / (float) tcn;   // correlated.cpp:147-150
    return r;   // TIME_REGULAR falls through every branch (:152)
}

// ---------------------------------------------------------------------------- modulation
constexpr float kInvTwoPiF = 0.15915494309189533577f;   // quotient estimate of the exact fmod (dtof_math.h: fmod_pos)
// waveform_utils.h:24-33
DTOF_D float waveform(float _t, int type) {
    float t = fmod_pos(_t, 2.f * kPi, kInvTwoPiF);
    if (type == WAVE_RECT) return fabsf(t - kPi) > 0.5f * kPi ? 1.f : -1.f;
    if (type == WAVE_TRI) return t < kPi ? 1.f - 2.f * t * (1.0f / kPi) : -3.f + 2.f * t * (1.0f / kPi);
    return cos_(t);
}
// waveform_utils.h:36-62
DTOF_D float waveform_low_pass(float _t, int type) {
    float t = fmod_pos(_t, 2.f * kPi, kInvTwoPiF);
    if (type == WAVE_SIN) return cos_(t);
    float a = t * (1.0f / kPi), b = 2.f - a, c = a < b ? a : b;
    if (type == WAVE_RECT) return 2.f - 4.f * c;
    if (type == WAVE_TRI) return (4.f * c * c * c - 6.f * c * c + 1.f) * 2.0f * (1.0f / 3.0f);
    float r = 2.f - 4.f * c;
    return fmin_(fmax_(2.0f * r, -2.0f), 2.0f);
}
// eval_modulation_weight -- dopplertofpath.cpp:60-77
DTOF_D float modulation_weight(const RenderParams &rp, float phase, float ray_time, float path_length) {
    float phi = rp.phi_coef * path_length;
    if (rp.low_pass) {
        float t = rp.w_d * ray_time + phase + phi;
        return rp.amp * waveform_low_pass(t, rp.wave_type);
    }
    float t1 = rp.w_g * ray_time - phi;
    float t2 = (rp.w_g + rp.w_d) * ray_time + phase;
    float g_t = rp.g_1 * waveform(t1, rp.wave_type) + rp.g_0;
    float s_t = waveform(t2, rp.wave_type);
    return s_t * g_t;
}

// ---------------------------------------------------------------------------- generate
// One lane of render_sample's head (integrator.cpp:476-495 / :416-431): sampler seeding, pixel jitter, time sample, camera ray.
struct PrimaryLane { float4 ray_a, ray_b; Rng main, path; float2 pos; };
// global lane index (pixel-major, the index every stream of the sampler is seeded with) of a lane of this launch
DTOF_D uint32_t global_lane(const RenderParams &rp, uint32_t virtual_lane) {
    if (rp.stripe_rows == 0) return virtual_lane;
    const uint32_t v = fdiv(virtual_lane, rp.d_lanes_per_row), in_row = virtual_lane - v * rp.lanes_per_row;
    const uint32_t s = fdiv(v, rp.d_stripe_rows), y = rp.stripe_first + s * rp.stripe_period + (v - s * rp.stripe_rows);
    return y * rp.lanes_per_row + in_row;
}
// Per-pixel part of a lane: the permutation seed of its sample sequence (compute_per_sequence_seed, sampler.cpp:85-92) and the
// pixel position (integrator.cpp:278-285).  UNIFORM: the 64 lanes of the wave hold samples of ONE pixel -- the TEA evaluation and the
// coordinates are then computed once per wave on the scalar unit instead of 64 times on the vector unit (same integers).
struct PixelInfo { uint32_t perm_seed; float posx, posy; };
template <bool UNIFORM>
DTOF_D PixelInfo pixel_info(const RenderParams &rp, uint32_t pix) {
    if (UNIFORM) pix = __builtin_amdgcn_readfirstlane(pix);
    PixelInfo pi; uint32_t tmp;
    tea32(rp.base_seed, rp.spp * pix + rp.seed, pi.perm_seed, tmp);
    const uint32_t W = (uint32_t) rp.crop_w, py = fdiv(pix, rp.d_w), px = pix - W * py;
    pi.posx = (float) (px + (uint32_t) rp.crop_x); pi.posy = (float) (py + (uint32_t) rp.crop_y);
    return pi;
}
// wave_pixel (uniform): see pixel_info; true only if spp is a multiple of 64 and the wave's lanes are 64 consecutive, 64-aligned lanes
// `vlane`: the lane's position in the rendered range (what the between-pass stream states are indexed by)
// Sensor::sample_ray_differential of the three sensors, without the time: (ax, ay) = the position sample in [0, 1]^2 of the crop window ("adjusted_pos",
// integrator.cpp:486-488), (apx, apy) = the aperture sample.  PERSPECTIVE_ONLY: see generate_lane.
template <bool PERSPECTIVE_ONLY = false>
DTOF_D void camera_ray(const RenderParams &rp, float ax, float ay, float apx, float apy, V3 &o, V3 &dw, float &maxt) {
    // PerspectiveCamera::sample_ray_differential (perspective.cpp:238-279)
    const float *m = rp.s2c;
    float r0 = fmaf(m[2], 0.f, fmaf(m[1], ay, fmaf(m[0], ax, m[3])));
    float r1 = fmaf(m[6], 0.f, fmaf(m[5], ay, fmaf(m[4], ax, m[7])));
    float r2 = fmaf(m[10], 0.f, fmaf(m[9], ay, fmaf(m[8], ax, m[11])));
    float r3 = fmaf(m[14], 0.f, fmaf(m[13], ay, fmaf(m[12], ax, m[15])));
    float iw = rcp(r3);
    V3 near_p = mk(r0 * iw, r1 * iw, r2 * iw);
    if (!PERSPECTIVE_ONLY && rp.orthographic) {   // OrthographicCamera::sample_ray_differential (orthographic.cpp:169-196): parallel rays from the near plane
        o = xf_point(rp.cam_to_world, near_p);
        dw = normalize(xf_vector(rp.cam_to_world, mk(0.f, 0.f, 1.f)));
        maxt = rp.far_clip - rp.near_clip;
        return;
    }
    const bool lens = !PERSPECTIVE_ONLY && rp.aperture_radius != 0.f;
    V3 d;
    if (lens) {   // ThinLensCamera::sample_ray_differential_impl (thinlens.cpp:257-305)
        float tx, ty; concentric_disk(apx, apy, tx, ty);
        const V3 aperture_p = mk(rp.aperture_radius * tx, rp.aperture_radius * ty, 0.f);
        const float f_dist = rp.focus_distance / near_p.z;
        d = normalize(near_p * f_dist - aperture_p);
        o = xf_point(rp.cam_to_world, aperture_p);
    } else {
        d = normalize(near_p);
        o = mk(rp.cam_to_world[3], rp.cam_to_world[7], rp.cam_to_world[11]);
    }
    dw = xf_vector(rp.cam_to_world, d);
    float inv_z = rcp(d.z), near_t = rp.near_clip * inv_z, far_t = rp.far_clip * inv_z;
    o = o + dw * near_t;
    maxt = far_t - near_t;
}

// PERSPECTIVE_ONLY: the sensor is known to be the plain perspective camera (the diffuse-only kernels: scenes with a thinlens / orthographic sensor run the every-BSDF
// instantiations) -- the aperture draw and the two other ray constructions are not compiled in
template <bool PERSPECTIVE_ONLY = false>
DTOF_D PrimaryLane generate_lane(const RenderParams &rp, uint32_t lane, bool wave_pixel = false, uint32_t vlane = 0) {
    // m_rng_time is drawn from by every strategy of the correlated sampler but uniform and stratified (correlated.cpp:96-106)
    const bool needs_tm = rp.integrator == 0 && rp.sampler_kind == SAMPLER_CORRELATED && rp.time_sampling >= TIME_ANTITHETIC;   // every strategy but uniform / stratified (:103-107)
    Rng main, tm, path; tm.state = 0; tm.inc = 1;
    uint2 *const carried = rp.n_passes > 1 ? rp.pass_rng + (size_t) (vlane - rp.pass_first) * 3 : nullptr;
    if (rp.pass == 0 && wave_pixel && needs_tm && rp.tcn == 2 && rp.pcn == 2) {
        // Correlated pairs (the default time_correlate_number = path_correlate_number = 2): lanes 2k and 2k + 1 share their time stream
        // TEA(seed + 1, k) and their path stream TEA(seed + 2, k) (correlated.cpp:54-63).  The even lane evaluates the first, the odd lane the
        // second, and the two swap results (quad_perm [1, 0, 3, 2]): one TEA evaluation per lane instead of two, same integers.
        // wave_pixel: the wave's lanes are consecutive, 64-aligned and all active or all inactive in pairs.
        main = seed_stream(rp.seed_value, lane);
        const bool odd = lane & 1u;
        uint32_t a0, a1; tea32(rp.seed_value + (odd ? 2u : 1u), lane >> 1, a0, a1);
        const uint32_t b0 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) a0, 0xb1, 0xf, 0xf, false), b1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) a1, 0xb1, 0xf, 0xf, false);
        pcg_seed(odd ? b0 : a0, odd ? b1 : a1, tm.state, tm.inc);
        pcg_seed(odd ? a0 : b0, odd ? a1 : b1, path.state, path.inc);
    } else if (rp.pass == 0) {
        main = seed_stream(rp.seed_value, lane);
        if (needs_tm) tm = seed_stream(rp.seed_value + 1, fdiv(lane, rp.d_tcn));
        path = seed_stream(rp.seed_value + 2, fdiv(lane, rp.d_pcn));
    } else {   // later passes: the sampler was seeded once (integrator.cpp:265); its streams run on where the previous pass left them
        const uint2 a = carried[0], b = carried[1], c = carried[2];
        main.state = (uint64_t) a.x | ((uint64_t) a.y << 32); main.inc = stream_inc(rp.seed_value, lane);
        if (needs_tm) { tm.state = (uint64_t) b.x | ((uint64_t) b.y << 32); tm.inc = stream_inc(rp.seed_value + 1, fdiv(lane, rp.d_tcn)); }
        path.state = (uint64_t) c.x | ((uint64_t) c.y << 32); path.inc = stream_inc(rp.seed_value + 2, fdiv(lane, rp.d_pcn));
    }
    const uint32_t pix = fdiv(lane, rp.d_spp);
    // current_sample_index = m_sample_index * samples_per_wavefront + lane % samples_per_wavefront (sampler.cpp:94-103); Sampler::advance
    // bumps m_sample_index once per pass (sampler.cpp:52-55)
    uint32_t si = (rp.spp > 1 ? lane - pix * rp.spp : 0) + rp.pass * rp.spp;
    const PixelInfo pi = wave_pixel ? pixel_info<true>(rp, pix) : pixel_info<false>(rp, pix);
    const uint32_t perm_seed = pi.perm_seed; const float posx = pi.posx, posy = pi.posy;
    uint32_t dim = 0;

    bool cp = rp.path_correlation_depth > 0;
    const bool doppler = rp.integrator == 0;
    // one stream only: the plain branch of render_sample (integrator.cpp:416-431: next_2d / next_1d), and every sampler but
    // `correlated` (Sampler::next_*_correlate default to next_1d / next_2d, include/mitsuba/render/sampler.h:141-144)
    const bool single = !doppler || rp.sampler_kind != SAMPLER_CORRELATED;
    float jx = single ? next_f32(main) : next_correlate(main, path, cp), jy = single ? next_f32(main) : next_correlate(main, path, cp);
    float spx = posx + jx, spy = posy + jy;
    float ax = fmaf(spx, rp.scale_x, rp.offset_x), ay = fmaf(spy, rp.scale_y, rp.offset_y);
    // needs_aperture_sample() (thinlens.cpp:155): a second 2-D draw of the same kind (integrator.cpp:421-423,490-492)
    const bool lens = !PERSPECTIVE_ONLY && rp.aperture_radius != 0.f;
    float apx = .5f, apy = .5f;
    if (lens) { apx = single ? next_f32(main) : next_correlate(main, path, cp); apy = single ? next_f32(main) : next_correlate(main, path, cp); }
    float time = rp.shutter_open;
    if (rp.shutter_open_time > 0.f) {
        float u;
        if (!doppler || rp.sampler_kind == SAMPLER_INDEPENDENT) u = next_f32(main);   // Sampler::next_1d_time -> next_1d (sampler.h:131-132)
        else if (rp.sampler_kind == SAMPLER_CORRELATED) u = next_time(rp, main, tm, si, perm_seed, dim);
        else {   // TimeStratifiedSampler::next_1d_time (timestratified.cpp:117-129): the strategy arguments are ignored
            uint32_t p = permute_kensler(si, rp.sample_count, perm_seed + dim++, rp.d_sample_count);
            float j = rp.jitter ? next_f32(main) : .5f;
            u = ((float) p + j) * rp.inv_spp;
        }
        time += u * rp.shutter_open_time;
    }
    if (carried && needs_tm) carried[1] = make_uint2((uint32_t) tm.state, (uint32_t) (tm.state >> 32));

    V3 o, dw; float maxt;
    camera_ray<PERSPECTIVE_ONLY>(rp, ax, ay, apx, apy, o, dw, maxt);
    if (doppler) time = time < rp.T ? time : time - rp.T;   // dopplertofpath.cpp:93

    PrimaryLane pl;
    pl.ray_a = make_float4(o.x, o.y, o.z, time);
    pl.ray_b = make_float4(dw.x, dw.y, dw.z, maxt);
    pl.main = main; pl.path = path; pl.pos = make_float2(spx, spy);
    return pl;
}

}  // namespace dtof
