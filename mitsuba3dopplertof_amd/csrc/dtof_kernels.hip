// dtof_kernels.hip -- hand-written HIP kernels (gfx950) of the wavefront Doppler-ToF path tracer.
//
// Stages (one kernel each, SoA queues in HBM, see dtof_kernels.h):
//   generate : sampler seeding + pixel jitter + time sample + camera ray
//              (src/render/integrator.cpp:273-290, 476-502; src/samplers/correlated.cpp:38-64,92-167;
//               src/sensors/perspective.cpp:238-279)
//   trace    : closest hit through the TLAS, motion-blur instances re-lerped per ray
//              (src/render/scene_embree.inl:202-333 semantics; src/shapes/instance.cpp:295-311)
//   shade    : surface interaction, point-light NEE set-up, diffuse BSDF eval+sample, modulation
//              weight, Russian roulette, wave-ballot compaction of survivors and shadow rays
//              (src/integrators/dopplertofpath.cpp:130-277)
//   shadow   : occlusion query; visible lanes commit their candidate result
//              (src/render/scene.cpp:235-291 test_visibility branch)
//   splat    : reconstruction-filter splat with per-pixel wave reduction, then float atomics
//              (src/render/imageblock.cpp:414-531)
//   develop  : RGB / W (src/films/hdrfilm.cpp:305-406)
//
// Compiled with -ffp-contract=off: an fma is issued exactly where fmaf() is written, so a lane's
// arithmetic is bit-identical to the scalar restatement in oracle/ (same helper algebra in dtof_math.h).
#include "dtof_kernels.h"
#include "dtof_scene.h"
#include "dtof_math.h"

#include <stdexcept>

namespace dtof {

#define DTOF_D __device__ __forceinline__
constexpr int kBlock = 256;

// ---------------------------------------------------------------------------- scene view
struct SceneView {
    const BvhNode *nodes; const DObject *objects; const DGroup *groups; const DShape *shapes;
    const DTri *tris; const DTriShade *shading; const DEmitter *emitters; const uint8_t *base;
    uint32_t n_nodes, n_emitters;
};
DTOF_D SceneView make_view(const uint8_t *base) {
    const BlobHeader *h = (const BlobHeader *) base;
    SceneView v;
    v.nodes = (const BvhNode *) (base + h->off_nodes);
    v.objects = (const DObject *) (base + h->off_objects);
    v.groups = (const DGroup *) (base + h->off_groups);
    v.shapes = (const DShape *) (base + h->off_shapes);
    v.tris = (const DTri *) (base + h->off_tris);
    v.shading = (const DTriShade *) (base + h->off_shading);
    v.emitters = (const DEmitter *) (base + h->off_emitters);
    v.base = base;
    v.n_nodes = h->n_nodes; v.n_emitters = h->n_emitters;
    return v;
}
// Stage the whole scene blob into LDS (small scenes: the Cornell blob is ~5 KB).
DTOF_D const uint8_t *stage_scene(const uint8_t *g, uint32_t bytes, uint4 *lds) {
    const uint4 *src = (const uint4 *) g;
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    return (const uint8_t *) lds;
}

struct Hit { float t, u, v; uint32_t obj, shape, prim; };

// ---------------------------------------------------------------------------- primitives
// Rectangle::ray_intersect_preliminary_impl, src/shapes/rectangle.cpp:201-224
DTOF_D bool rect_hit(const DShape &sh, V3 o, V3 d, float maxt, float &t, float &u, float &v) {
    V3 lo = xf_point(sh.to_object, o), ld = xf_vector(sh.to_object, d);
    t = -lo.z / ld.z;
    u = fmaf(ld.x, t, lo.x); v = fmaf(ld.y, t, lo.y);
    return t >= 0.f && t <= maxt && fabsf(u) <= 1.f && fabsf(v) <= 1.f;
}
// Moeller-Trumbore as in Embree 3's triangle intersector (tnear < t <= tfar; u,v weight vertices 1,2).
// The 48-byte record is fetched with three 16-byte loads issued together; `face` rides in p0.w.
DTOF_D bool tri_hit(const DTri &tr, V3 o, V3 d, float maxt, float &t, float &u, float &v, uint32_t &face) {
    const uint4 *tp = (const uint4 *) &tr;
    const uint4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
    face = q0.w;
    V3 p0 = mk(u2f(q0.x), u2f(q0.y), u2f(q0.z)), p1 = mk(u2f(q1.x), u2f(q1.y), u2f(q1.z)), p2 = mk(u2f(q2.x), u2f(q2.y), u2f(q2.z));
    V3 e1 = p0 - p1, e2 = p2 - p0, ng = cross(e2, e1);
    V3 c = p0 - o, r = cross(c, d);
    float den = dot(ng, d), aden = fabsf(den);
    uint32_t sgn = f2u(den) & 0x80000000u;
    float U = u2f(f2u(dot(r, e2)) ^ sgn), Vv = u2f(f2u(dot(r, e1)) ^ sgn);
    if (!(den != 0.f && U >= 0.f && Vv >= 0.f && U + Vv <= aden)) return false;
    float T = u2f(f2u(dot(ng, c)) ^ sgn);
    if (!(0.f < T && T <= aden * maxt)) return false;
    float rc = 1.0f / aden;
    u = U * rc; v = Vv * rc; t = T * rc;
    return true;
}
// math::solve_quadratic (include/mitsuba/core/math.h:357-401), float64
DTOF_D bool solve_quadratic_d(double a, double b, double c, double &x0, double &x1) {
    const bool linear = a == 0.0, valid_linear = linear && b != 0.0;
    x0 = x1 = -c / b;
    const double discrim = fma(b, b, -(4.0 * a * c));
    const bool valid_quadratic = !linear && discrim >= 0.0;
    if (valid_quadratic) {
        const double sq = sqrt(discrim), temp = -0.5 * (b + copysign(sq, b));
        const double x0p = temp / a, x1p = c / temp;
        x0 = x0p < x1p ? x0p : x1p; x1 = x0p < x1p ? x1p : x0p;
    }
    return valid_linear || valid_quadratic;
}
DTOF_D double dot3d(double ax, double ay, double az, double bx, double by, double bz) { return fma(az, bz, fma(ay, by, ax * bx)); }
// Sphere::ray_intersect_preliminary_impl (src/shapes/sphere.cpp:338-394) / ray_test_impl (:396-431): float64 on the llvm back
// end; the point of the ray closest to the centre is evaluated with the FLOAT ray (Ray::operator() takes a Float, ray.h:61).
template <bool ANY>
DTOF_D bool sphere_hit(const DShape &sh, V3 o, V3 d, float maxt_f, float &t_out) {
    const double radius = sh.dp_du[0], cx = sh.n[0], cy = sh.n[1], cz = sh.n[2], maxt = maxt_f;
    const double dx = d.x, dy = d.y, dz = d.z;
    double near_t, far_t;
    if (ANY) {
        const double ox = (double) o.x - cx, oy = (double) o.y - cy, oz = (double) o.z - cz;
        const double A = dot3d(dx, dy, dz, dx, dy, dz), B = 2.0 * dot3d(ox, oy, oz, dx, dy, dz), C = dot3d(ox, oy, oz, ox, oy, oz) - radius * radius;
        const bool found = solve_quadratic_d(A, B, C, near_t, far_t);
        const bool out_bounds = !(near_t <= maxt && far_t >= 0.0), in_bounds = near_t < 0.0 && far_t > maxt;
        return found && !out_bounds && !in_bounds;
    }
    const double lx = (double) o.x - cx, ly = (double) o.y - cy, lz = (double) o.z - cz;
    const double plane_t = dot3d(-lx, -ly, -lz, dx, dy, dz) / sqrt(dot3d(dx, dy, dz, dx, dy, dz));
    bool no_hit = plane_t == 0.0 && (o.x != sh.n[0] && o.y != sh.n[1] && o.z != sh.n[2]);
    const V3 pp = vfma(d, (float) plane_t, o);
    const double ox = (double) pp.x - cx, oy = (double) pp.y - cy, oz = (double) pp.z - cz;
    no_hit = no_hit && sqrt(dot3d(ox, oy, oz, ox, oy, oz)) > radius;
    const double A = dot3d(dx, dy, dz, dx, dy, dz), B = 2.0 * dot3d(ox, oy, oz, dx, dy, dz), C = dot3d(ox, oy, oz, ox, oy, oz) - radius * radius;
    const bool found = solve_quadratic_d(A, B, C, near_t, far_t);
    near_t += plane_t; far_t += plane_t;
    const bool out_bounds = !(near_t <= maxt && far_t >= 0.0), in_bounds = near_t < 0.0 && far_t > maxt;
    if (!(found && !no_hit && !out_bounds && !in_bounds)) return false;
    t_out = near_t < 0.0 ? (float) far_t : (float) near_t;
    return true;
}
// AnimatedTransform::eval, include/mitsuba/core/transform.h:439-466
DTOF_D void instance_matrix(const DObject &ob, float time, float *m) {
    if (ob.n_keys <= 1) {
#pragma unroll
        for (int i = 0; i < 12; ++i) m[i] = ob.key0[i];
        return;
    }
    float t = fmin_(fmax_((time - ob.t0) / (ob.t1 - ob.t0), 0.f), 1.f), omt = 1 - t;
#pragma unroll
    for (int i = 0; i < 12; ++i) m[i] = ob.key0[i] * omt + ob.key1[i] * t;
}

// Slab test against a padded box; NaNs (0*inf) fall out of the min/max chain conservatively.
DTOF_D float box_entry(const float *bmin, const float *bmax, V3 o, V3 id, float tbest) {
    float tx0 = (bmin[0] - o.x) * id.x, tx1 = (bmax[0] - o.x) * id.x;
    float ty0 = (bmin[1] - o.y) * id.y, ty1 = (bmax[1] - o.y) * id.y;
    float tz0 = (bmin[2] - o.z) * id.z, tz1 = (bmax[2] - o.z) * id.z;
    float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.f));
    float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tbest));
    return tn <= tf ? tn : INFINITY;
}
// One BVH node = four 16-byte loads issued together (no load depends on a field of the node); entry distances of both
// children, INFINITY = missed / absent.
DTOF_D void node_test(const BvhNode *node, V3 o, V3 id, float tbest, float &tl, float &tr, uint32_t &left, uint32_t &right) {
    const uint4 *np = (const uint4 *) node;
    const uint4 a = np[0], b = np[1], c = np[2], d = np[3];
    const float lmin[3] = { u2f(a.x), u2f(a.y), u2f(a.z) }, lmax[3] = { u2f(b.x), u2f(b.y), u2f(b.z) };
    const float rmin[3] = { u2f(c.x), u2f(c.y), u2f(c.z) }, rmax[3] = { u2f(d.x), u2f(d.y), u2f(d.z) };
    left = a.w; right = b.w;
    tl = box_entry(lmin, lmax, o, id, tbest);
    tr = box_entry(rmin, rmax, o, id, tbest);
    if (right == kNoChild) tr = INFINITY;
}

// Closest hit (ANY=false) or occlusion (ANY=true) of one top-level object.  Candidates are every
// primitive hit with t <= maxt; the winner is the smallest t, ties going to the lowest
// (object, shape, face) -- the rule the oracle uses, independent of traversal order.
// `stack + sp * stride` onwards is free for the BLAS traversal of a mesh.
template <bool ANY, bool MESH>
DTOF_D bool intersect_object(const SceneView &sv, uint32_t oi, V3 o, V3 d, float time, float maxt, Hit &best,
                             uint32_t *stack, int sp, uint32_t stride) {
    const DObject &ob = sv.objects[oi];
    uint32_t first = ob.index, count = 1;
    V3 lo = o, ld = d;
    if (ob.kind == OBJ_INSTANCE) {
        float m[12], inv[12];
        instance_matrix(ob, time, m);
        affine_inverse(m, inv);
        lo = xf_point(inv, o); ld = xf_vector(inv, d);
        const DGroup &g = sv.groups[ob.index];
        first = g.first_shape; count = g.n_shapes;
    }
    bool found = false;
    for (uint32_t k = 0; k < count; ++k) {
        const DShape &sh = sv.shapes[first + k];
        float t, u, v;
        if (sh.kind == SHAPE_RECT) {
            if (rect_hit(sh, lo, ld, maxt, t, u, v)) {
                if (ANY) return true;
                if (t < best.t || (t == best.t && !found && best.obj != 0xffffffffu && oi < best.obj)) {
                    best.t = t; best.u = u; best.v = v; best.obj = oi; best.shape = k; best.prim = 0; found = true;
                }
            }
            continue;
        }
        if (!MESH) continue;   // instantiations for rectangle-only scenes carry no triangle / sphere code at all
        if (sh.kind == SHAPE_SPHERE) {
            if (sphere_hit<ANY>(sh, lo, ld, maxt, t)) {
                if (ANY) return true;
                if (t < best.t || (t == best.t && !found && best.obj != 0xffffffffu && oi < best.obj)) {
                    best.t = t; best.u = 0.f; best.v = 0.f; best.obj = oi; best.shape = k; best.prim = 0; found = true;
                }
            }
            continue;
        }
        // cull with the mesh's own (padded) bounds: TLAS boxes of moving instances are the union over the whole
        // motion and let many rays through that miss the mesh at their time
        V3 lid = mk(__builtin_amdgcn_rcpf(ld.x == 0.f ? 1e-30f : ld.x), __builtin_amdgcn_rcpf(ld.y == 0.f ? 1e-30f : ld.y), __builtin_amdgcn_rcpf(ld.z == 0.f ? 1e-30f : ld.z));
        if (!(box_entry(sh.bmin, sh.bmax, lo, lid, ANY ? maxt : best.t) < INFINITY)) continue;
        // `face` of the best hit so far IF it lies on this very mesh (ties between two of its triangles go to the lower face)
        uint32_t best_face = 0xffffffffu;
        auto test = [&](uint32_t f) -> bool {
            uint32_t face;
            if (!tri_hit(sv.tris[sh.first_tri + f], lo, ld, maxt, t, u, v, face)) return false;
            if (ANY) return true;
            bool take = t < best.t;
            if (t == best.t) take = best_face != 0xffffffffu ? face < best_face : (!found && best.obj != 0xffffffffu && oi < best.obj);
            if (take) { best.t = t; best.u = u; best.v = v; best.obj = oi; best.shape = k; best.prim = f; best_face = face; found = true; }
            return false;
        };
        if (sh.blas_root == kNoChild) {
            for (uint32_t f = 0; f < sh.n_tris; ++f) if (test(f)) return true;
            continue;
        }
        // BLAS: same node format and while-while shape as the TLAS loop below
        constexpr uint32_t kDone = 0x7fffffffu;
        uint32_t cur = sh.blas_root; int bsp = sp;
        for (;;) {
            while (!(cur & kLeafFlag) && cur != kDone) {
                float tl, tr; uint32_t left, right;
                node_test(sv.nodes + cur, lo, lid, ANY ? maxt : best.t, tl, tr, left, right);
                bool hl = tl < INFINITY, hr = tr < INFINITY;
                if (hl && hr) {
                    uint32_t nearc = tl <= tr ? left : right, farc = tl <= tr ? right : left;
                    stack[bsp * stride] = farc; ++bsp;
                    cur = nearc;
                } else if (hl) cur = left;
                else if (hr) cur = right;
                else if (bsp == sp) cur = kDone;
                else { --bsp; cur = stack[bsp * stride]; }
            }
            if (cur == kDone) break;
            uint32_t f0 = (cur & ~kLeafFlag) >> kBlasLeafBits, fn = (cur & ((1u << kBlasLeafBits) - 1u)) + 1u;
            for (uint32_t f = f0; f < f0 + fn; ++f) if (test(f)) return true;
            if (bsp == sp) break;
            --bsp; cur = stack[bsp * stride];
        }
    }
    return found;
}

// TLAS traversal; `stack` is a per-thread LDS column (stride blockDim.x).
template <bool ANY, bool MESH>
DTOF_D bool trace_scene(const SceneView &sv, uint32_t *stack, V3 o, V3 d, float time, float maxt, Hit &best) {
    best.t = maxt; best.u = best.v = 0.f; best.obj = 0xffffffffu; best.shape = 0; best.prim = 0;
    if (sv.n_nodes == 0) return false;
    // direction reciprocal for the slab test only (exact zero components are nudged)
    // v_rcp_f32 (1 ulp) is enough here: the boxes are padded by 1e-5 relative on the host
    V3 id = mk(__builtin_amdgcn_rcpf(d.x == 0.f ? 1e-30f : d.x), __builtin_amdgcn_rcpf(d.y == 0.f ? 1e-30f : d.y), __builtin_amdgcn_rcpf(d.z == 0.f ? 1e-30f : d.z));
    // "while-while" traversal: every lane first descends inner nodes until it holds a leaf (or is done), THEN the
    // lanes that hold a leaf run the expensive object intersection together -- the wave does not pay the leaf
    // body once per node step of its slowest lane.
    constexpr uint32_t kDone = 0x7fffffffu;
    int sp = 0;
    uint32_t cur = 0;
    const uint32_t stride = blockDim.x;
    for (;;) {
        while (!(cur & kLeafFlag) && cur != kDone) {
            float tl, tr; uint32_t left, right;
            node_test(sv.nodes + cur, o, id, best.t, tl, tr, left, right);
            bool hl = tl < INFINITY, hr = tr < INFINITY;
            if (hl && hr) {
                uint32_t nearc = tl <= tr ? left : right, farc = tl <= tr ? right : left;
                stack[sp * stride] = farc; ++sp;
                cur = nearc;
            } else if (hl) cur = left;
            else if (hr) cur = right;
            else if (sp == 0) cur = kDone;
            else { --sp; cur = stack[sp * stride]; }
        }
        if (cur == kDone) break;
        if (intersect_object<ANY, MESH>(sv, cur & ~kLeafFlag, o, d, time, maxt, best, stack, sp, stride) && ANY) return true;
        if (sp == 0) break;
        --sp; cur = stack[sp * stride];
    }
    return best.obj != 0xffffffffu;
}

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
// next_1d_correlate -- correlated.cpp:156-161
DTOF_D float next_correlate(Rng &main, Rng &path, bool correlate) {
    float r1 = next_f32(path), r2 = next_f32(main);
    return correlate ? r1 : r2;
}
// next_1d_time -- correlated.cpp:92-153; si = current_sample_index (sampler.cpp:94-103)
DTOF_D float next_time(const RenderParams &rp, Rng &main, Rng &tm, uint32_t si, uint32_t perm_seed, uint32_t &dim) {
    int strategy = rp.time_sampling; uint32_t tcn = rp.tcn;
    if (strategy == TIME_UNIFORM) return next_f32(main);
    float r = strategy == TIME_STRATIFIED ? next_f32(main) : next_f32(tm);
    if (rp.stratify) {
        if (strategy == TIME_STRATIFIED) {
            // the reference evaluates p1 (seed + dim) and p2 (seed + dim + 1) and selects; the permutation is a pure function,
            // so only the selected one is computed
            const uint32_t ps = perm_seed + dim + ((si % tcn != 0) ? 0u : 1u);
            dim += 2;
            const uint32_t p = permute_kensler(si / tcn, rp.n_stratum, ps);
            r = ((float) p + r) * rp.inv_n_stratum;
        } else {
            r = ((float) (si / tcn) + r) * rp.inv_n_stratum;
        }
    }
    if (strategy == TIME_STRATIFIED) return ((float) (si % tcn) + r) * rp.inv_tcn;
    if (strategy == TIME_ANTITHETIC) {
        uint32_t rem = si % tcn;
        if (tcn == 2) { float r2 = r + rp.antithetic_shift; return rem != 1 ? r : r2; }
        return r + (float) rem / (float) tcn;
    }
    // TIME_ANTITHETIC_MIRROR
    float r2 = 1.0f - r + rp.antithetic_shift;
    return (si % tcn) != 1 ? r : r2;
}

// ---------------------------------------------------------------------------- modulation
// waveform_utils.h:24-33
DTOF_D float waveform(float _t, int type) {
    float t = fmodf(_t, 2.f * kPi);
    if (type == WAVE_RECT) return fabsf(t - kPi) > 0.5f * kPi ? 1.f : -1.f;
    if (type == WAVE_TRI) return t < kPi ? 1.f - 2.f * t * (1.0f / kPi) : -3.f + 2.f * t * (1.0f / kPi);
    return cos_(t);
}
// waveform_utils.h:36-62
DTOF_D float waveform_low_pass(float _t, int type) {
    float t = fmodf(_t, 2.f * kPi);
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
    const uint32_t v = virtual_lane / rp.lanes_per_row, in_row = virtual_lane - v * rp.lanes_per_row;
    const uint32_t s = v / rp.stripe_rows, y = rp.stripe_first + s * rp.stripe_period + (v - s * rp.stripe_rows);
    return y * rp.lanes_per_row + in_row;
}
DTOF_D PrimaryLane generate_lane(const RenderParams &rp, uint32_t lane) {
    Rng main = seed_stream(rp.seed_value, lane);
    // m_rng_time is only drawn from by the antithetic strategies of the correlated sampler (correlated.cpp:96-106)
    const bool needs_tm = rp.integrator == 0 && rp.sampler_kind == SAMPLER_CORRELATED && (rp.time_sampling == TIME_ANTITHETIC || rp.time_sampling == TIME_ANTITHETIC_MIRROR);
    Rng tm; tm.state = 0; tm.inc = 1;
    if (needs_tm) tm = seed_stream(rp.seed_value + 1, lane / rp.tcn);
    Rng path = seed_stream(rp.seed_value + 2, lane / rp.pcn);
    uint32_t pix = rp.spp_log2 != 0xffffffffu ? lane >> rp.spp_log2 : lane / rp.spp;
    uint32_t si = rp.spp > 1 ? lane - pix * rp.spp : 0;
    uint32_t perm_seed, tmp; tea32(rp.base_seed, rp.spp * pix + rp.seed, perm_seed, tmp);
    uint32_t dim = 0;

    uint32_t W = (uint32_t) rp.crop_w;
    uint32_t py = pix / W, px = pix - W * py;
    float posx = (float) (px + (uint32_t) rp.crop_x), posy = (float) (py + (uint32_t) rp.crop_y);
    bool cp = rp.path_correlation_depth > 0;
    const bool doppler = rp.integrator == 0;
    // one stream only: the plain branch of render_sample (integrator.cpp:416-431: next_2d / next_1d), and every sampler but
    // `correlated` (Sampler::next_*_correlate default to next_1d / next_2d, include/mitsuba/render/sampler.h:141-144)
    const bool single = !doppler || rp.sampler_kind != SAMPLER_CORRELATED;
    float jx = single ? next_f32(main) : next_correlate(main, path, cp), jy = single ? next_f32(main) : next_correlate(main, path, cp);
    float spx = posx + jx, spy = posy + jy;
    float ax = fmaf(spx, rp.scale_x, rp.offset_x), ay = fmaf(spy, rp.scale_y, rp.offset_y);
    float time = rp.shutter_open;
    if (rp.shutter_open_time > 0.f) {
        float u;
        if (!doppler || rp.sampler_kind == SAMPLER_INDEPENDENT) u = next_f32(main);   // Sampler::next_1d_time -> next_1d (sampler.h:131-132)
        else if (rp.sampler_kind == SAMPLER_CORRELATED) u = next_time(rp, main, tm, si, perm_seed, dim);
        else {   // TimeStratifiedSampler::next_1d_time (timestratified.cpp:117-129): the strategy arguments are ignored
            uint32_t p = permute_kensler(si, rp.spp, perm_seed + dim++);
            float j = rp.jitter ? next_f32(main) : .5f;
            u = ((float) p + j) * rp.inv_spp;
        }
        time += u * rp.shutter_open_time;
    }

    // PerspectiveCamera::sample_ray_differential (perspective.cpp:238-279)
    const float *m = rp.s2c;
    float r0 = fmaf(m[2], 0.f, fmaf(m[1], ay, fmaf(m[0], ax, m[3])));
    float r1 = fmaf(m[6], 0.f, fmaf(m[5], ay, fmaf(m[4], ax, m[7])));
    float r2 = fmaf(m[10], 0.f, fmaf(m[9], ay, fmaf(m[8], ax, m[11])));
    float r3 = fmaf(m[14], 0.f, fmaf(m[13], ay, fmaf(m[12], ax, m[15])));
    float iw = rcp(r3);
    V3 d = normalize(mk(r0 * iw, r1 * iw, r2 * iw));
    V3 o = mk(rp.cam_to_world[3], rp.cam_to_world[7], rp.cam_to_world[11]);
    V3 dw = xf_vector(rp.cam_to_world, d);
    float inv_z = rcp(d.z), near_t = rp.near_clip * inv_z, far_t = rp.far_clip * inv_z;
    o = o + dw * near_t;
    float maxt = far_t - near_t;
    if (doppler) time = time < rp.T ? time : time - rp.T;   // dopplertofpath.cpp:93

    PrimaryLane pl;
    pl.ray_a = make_float4(o.x, o.y, o.z, time);
    pl.ray_b = make_float4(dw.x, dw.y, dw.z, maxt);
    pl.main = main; pl.path = path; pl.pos = make_float2(spx, spy);
    return pl;
}
__global__ __launch_bounds__(kBlock) void k_generate(RenderParams rp, Queues q) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= rp.n_lanes) return;
    const PrimaryLane pl = generate_lane(rp, global_lane(rp, rp.lane_base + i));
    q.ray_a[i] = pl.ray_a;
    q.ray_b[i] = pl.ray_b;
    q.st_a[i] = make_float4(1.f, 1.f, 1.f, 0.f);
    q.rng_a[i] = make_uint4((uint32_t) pl.main.state, (uint32_t) (pl.main.state >> 32), (uint32_t) pl.path.state, (uint32_t) (pl.path.state >> 32));
    q.rng_b[i] = make_uint2((uint32_t) (pl.main.inc >> 1), (uint32_t) (pl.path.inc >> 1));
    q.pos[i] = pl.pos;
    for (int k = 0; k < rp.n_offsets; ++k) q.res[(size_t) k * q.capacity + i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------- trace
// Segmented queues: the wavefront is cut into segments of kSeg lanes.  A shade block owns one
// segment: it compacts the survivors (and the shadow rays) of its segment to the front of the same
// segment of the output queue and records the count -- order preserving, deterministic and without a
// single global atomic (a shared counter serialises at ~88 returning atomics/us on MI355X, which
// made the first version of this kernel 10x slower than its memory traffic).
constexpr uint32_t kSeg = 512;
constexpr int kShadeBlock = 64;   // k_shade runs ONE wave per block: compaction is ballot+popcount only, no barrier in the chunk loop
DTOF_D uint32_t seg_count(const uint32_t *counts, uint32_t seg, uint32_t n_lanes) {
    return counts ? counts[seg] : min(kSeg, n_lanes - seg * kSeg);
}

// BLOCK: 256 threads when the scene is staged into LDS (the staging is shared by four waves), ONE wave otherwise -- the waves of a
// block share nothing then, and a block only frees its LDS and wave slots when its slowest wave is done, which costs occupancy
// on divergent traversals (large scenes).
template <bool LDS, bool MESH, int BLOCK>
__global__ __launch_bounds__(BLOCK, BLOCK == 64 ? 6 : 1) void k_trace(const uint8_t *scene, uint32_t scene_bytes, uint32_t stage_words,
                                                 Queues q, const uint32_t *qin, const uint32_t *count_in, uint32_t n_lanes) {
    constexpr uint32_t kBlock = BLOCK, kSub = kSeg / BLOCK;
    extern __shared__ uint4 lds[];
    uint32_t seg = blockIdx.x / kSub, sub = blockIdx.x % kSub;
    uint32_t count = seg_count(count_in, seg, n_lanes);
    if (sub * kBlock >= count) return;
    const uint8_t *base = LDS ? stage_scene(scene, scene_bytes, lds) : scene;
    uint32_t *stack = (uint32_t *) (lds + stage_words) + threadIdx.x;
    SceneView sv = make_view(base);
    uint32_t j = sub * kBlock + threadIdx.x;
    if (j >= count) return;
    uint32_t l = qin ? qin[seg * kSeg + j] : seg * kSeg + j;
    float4 a = q.ray_a[l], b = q.ray_b[l];
    Hit h;
    bool found = trace_scene<false, MESH>(sv, stack, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), a.w, b.w, h);
    q.hit[l] = make_uint4(f2u(h.t), f2u(h.u), f2u(h.v), h.prim);
    q.hit_id[l] = found ? (h.obj | (h.shape << 24)) : 0xffffffffu;
}

// ---------------------------------------------------------------------------- shade
struct Surface { V3 p, n, sh_n, sh_s, sh_t, wi; const DShape *shape; };

// Shape::compute_surface_interaction for rectangle (rectangle.cpp:250-323) / mesh (mesh.cpp:632-864),
// Instance::compute_surface_interaction (instance.cpp:155-250), finalize (interaction.h:493-513)
template <bool MESH>
DTOF_D void compute_surface(const SceneView &sv, uint32_t oi, uint32_t shape_k, uint32_t prim, float t, float b1, float b2,
                            V3 o, V3 d, float time, Surface &si) {
    const DObject &ob = sv.objects[oi];
    bool inst = ob.kind == OBJ_INSTANCE;
    float m[12], inv[12];
    V3 lo = o, ld = d;
    const DShape *sh;
    if (inst) {
        instance_matrix(ob, time, m);
        affine_inverse(m, inv);
        lo = xf_point(inv, o); ld = xf_vector(inv, d);
        sh = &sv.shapes[sv.groups[ob.index].first_shape + shape_k];
    } else sh = &sv.shapes[ob.index];
    si.shape = sh;
    V3 dp_du, dp_dv;
    if (!MESH || sh->kind == SHAPE_RECT) {
        V3 n = mk(sh->n[0], sh->n[1], sh->n[2]);
        V3 p = vfma(ld, t, lo);
        V3 tr = mk(sh->to_world[3], sh->to_world[7], sh->to_world[11]);
        float dist = dot(tr - p, n);
        si.p = p + n * dist; si.n = n; si.sh_n = n;
        dp_du = mk(sh->dp_du[0], sh->dp_du[1], sh->dp_du[2]);
        dp_dv = mk(sh->dp_dv[0], sh->dp_dv[1], sh->dp_dv[2]);
    } else if (sh->kind == SHAPE_SPHERE) {   // Sphere::compute_surface_interaction (sphere.cpp:509-513, 527-551)
        const V3 c = mk(sh->n[0], sh->n[1], sh->n[2]); const float radius = sh->dp_du[0];
        V3 n = normalize(vfma(ld, t, lo) - c);
        si.p = vfma(n, radius, c);
        const V3 local = xf_point(sh->to_object, si.p);
        const float rd = sqrtf(sqr(local.x) + sqr(local.y)), inv_rd = rcp(rd);
        V3 dpv = mk(local.z * (local.x * inv_rd), local.z * (local.y * inv_rd), -rd);
        if (rd == 0.f) dpv = mk(1.f, 0.f, 0.f);
        dp_du = xf_vector(sh->to_world, mk(-local.y, local.x, 0.f)) * (2.f * kPi);
        dp_dv = xf_vector(sh->to_world, dpv) * kPi;
        if (sh->flags & SF_FLIP_NORMALS) n = -n;
        si.n = n; si.sh_n = n;
    } else {
        const DTri &tr = sv.tris[sh->first_tri + prim];
        const DTriShade &ts = sv.shading[sh->first_tri + prim];
        V3 p0 = mk(tr.p0[0], tr.p0[1], tr.p0[2]), p1 = mk(tr.p1[0], tr.p1[1], tr.p1[2]), p2 = mk(tr.p2[0], tr.p2[1], tr.p2[2]);
        float b0 = 1.f - b1 - b2;
        V3 dp0 = p1 - p0, dp1 = p2 - p0;
        si.p = vfma(p0, b0, vfma(p1, b1, p2 * b2));
        si.n = normalize(cross(dp0, dp1));
        coordinate_system(si.n, dp_du, dp_dv);
        float d0x = ts.uv1[0] - ts.uv0[0], d0y = ts.uv1[1] - ts.uv0[1], d1x = ts.uv2[0] - ts.uv0[0], d1y = ts.uv2[1] - ts.uv0[1];
        float det = fmaf(d0x, d1y, -(d0y * d1x)), inv_det = rcp(det);
        if (det != 0.f) {
            dp_du = mk(fmaf(d1y, dp0.x, -(d0y * dp1.x)), fmaf(d1y, dp0.y, -(d0y * dp1.y)), fmaf(d1y, dp0.z, -(d0y * dp1.z))) * inv_det;
            dp_dv = mk(fmaf(-d1x, dp0.x, d0x * dp1.x), fmaf(-d1x, dp0.y, d0x * dp1.y), fmaf(-d1x, dp0.z, d0x * dp1.z)) * inv_det;
        }
        if (!(sh->flags & SF_FACE_NORMALS)) {
            V3 n0 = mk(ts.n0[0], ts.n0[1], ts.n0[2]), n1 = mk(ts.n1[0], ts.n1[1], ts.n1[2]), n2 = mk(ts.n2[0], ts.n2[1], ts.n2[2]);
            V3 n = vfma(n2, b2, vfma(n1, b1, n0 * b0));
            si.sh_n = n * rsqrt_(dot(n, n));
        } else si.sh_n = si.n;
        if (sh->flags & SF_FLIP_NORMALS) { si.n = -si.n; si.sh_n = -si.sh_n; }
    }
    if (inst) {
        si.p = xf_point(m, si.p);
        si.n = normalize(xf_normal(inv, si.n));
        si.sh_n = normalize(xf_normal(inv, si.sh_n));
        dp_du = xf_vector(m, dp_du);
    }
    // initialize_sh_frame (interaction.h:258-268)
    V3 s = normalize(vfma(si.sh_n, -dot(si.sh_n, dp_du), dp_du));
    if (dp_du.x == 0.f && dp_du.y == 0.f && dp_du.z == 0.f) { V3 tt; coordinate_system(si.sh_n, s, tt); }
    si.sh_s = s; si.sh_t = cross(si.sh_n, s);
    V3 md = -d;
    si.wi = mk(dot(md, si.sh_s), dot(md, si.sh_t), dot(md, si.sh_n));
}
// Interaction::offset_p (interaction.h:161-165)
DTOF_D V3 offset_p(const Surface &si, V3 d) {
    float mag = (1.f + fmax_(fmax_(fabsf(si.p.x), fabsf(si.p.y)), fabsf(si.p.z))) * kRayEps;
    mag = mulsign(mag, dot(si.n, d));
    return vfma(si.n, mag, si.p);
}
// warp::square_to_cosine_hemisphere (warp.h:54-86, 320-344)
DTOF_D V3 cosine_hemisphere(float sx, float sy) {
    float x = fmaf(2.f, sx, -1.f), y = fmaf(2.f, sy, -1.f);
    bool is_zero = x == 0.f && y == 0.f, q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * kPi * rp / r;
    if (q13) phi = 0.5f * kPi - phi;
    if (is_zero) phi = 0.f;
    float s, c; sincos_(phi, s, c);
    float px = r * c, py = r * s;
    return mk(px, py, sqrtf(fmax_(1.f - fmaf(py, py, px * px), 0.f)));
}
// Mesh::sample_position (mesh.cpp:513-568): face by DiscreteDistribution::sample_reuse on sample.y (distr_1d.h:113-160,
// dr::binary_search over [m_valid.x, m_valid.y]), point by warp::square_to_uniform_triangle (warp.h:153-156), normal from
// the vertex normals if the mesh has them.
DTOF_D void mesh_sample_position(const SceneView &sv, const DShape &es, float s_x, float s_y, V3 &p, V3 &n) {
    const float *cdf = (const float *) (sv.base + es.emit_table), *pmf = cdf + es.n_tris;
    const uint32_t *slot = (const uint32_t *) (pmf + es.n_tris);
    const float v = s_y * es.emit_sum;
    uint32_t lo = es.emit_lo, hi = es.emit_hi;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] < v) lo = mid + 1 < hi ? mid + 1 : hi; else hi = mid;
    }
    const float pm = pmf[lo] * es.inv_area, cd = lo > 0 ? cdf[lo - 1] * es.inv_area : 0.f;
    const float y = (s_y - cd) / pm;
    const uint32_t k = es.first_tri + slot[lo];
    const DTri &tr = sv.tris[k];
    V3 p0 = mk(tr.p0[0], tr.p0[1], tr.p0[2]), p1 = mk(tr.p1[0], tr.p1[1], tr.p1[2]), p2 = mk(tr.p2[0], tr.p2[1], tr.p2[2]);
    V3 e0 = p1 - p0, e1 = p2 - p0;
    const float t = sqrtf(fmax_(1.f - s_x, 0.f)), bx = 1.f - t, by = t * y;
    p = vfma(e0, bx, vfma(e1, by, p0));
    if (!(es.flags & SF_FACE_NORMALS)) {
        const DTriShade &ts = sv.shading[k];
        V3 n0 = mk(ts.n0[0], ts.n0[1], ts.n0[2]), n1 = mk(ts.n1[0], ts.n1[1], ts.n1[2]), n2 = mk(ts.n2[0], ts.n2[1], ts.n2[2]);
        n = vfma(n0, 1.f - bx - by, vfma(n1, bx, n2 * by));
    } else n = cross(e0, e1);
    n = normalize(n);
    if (es.flags & SF_FLIP_NORMALS) n = -n;
}

constexpr float kInvTwoPi = 0.15915494309189533577f;
DTOF_D float uniform_cone_pdf(float cos_cutoff) { return kInvTwoPi / (1.f - cos_cutoff); }   // warp::square_to_uniform_cone_pdf (warp.h:475-485)
// Sphere::sample_direction (sphere.cpp:222-296): cone sampling of the visible cap from outside, uniform sphere from inside
DTOF_D void sphere_sample_direction(const DShape &sh, V3 ref, float s_x, float s_y, V3 &p, V3 &n, V3 &dd, float &dist, float &pdf) {
    const V3 center = mk(sh.n[0], sh.n[1], sh.n[2]); const float radius = sh.dp_du[0];
    const bool flip = sh.flags & SF_FLIP_NORMALS;
    const V3 dc_v = center - ref;
    const float dc_2 = dot(dc_v, dc_v), radius_adj = radius * (flip ? (1.f + kRayEps) : (1.f - kRayEps));
    const bool outside = dc_2 > sqr(radius_adj);
    V3 dloc;
    if (outside) {
        const float inv_dc = rsqrt_(dc_2), sin_theta_max = radius * inv_dc, sin_theta_max_2 = sqr(sin_theta_max),
                    inv_sin_theta_max = rcp(sin_theta_max), cos_theta_max = safe_sqrt(1.f - sin_theta_max_2);
        const float sin_theta_2 = sin_theta_max_2 > 0.00068523f ? 1.f - sqr(fmaf(cos_theta_max - 1.f, s_x, 1.f)) : sin_theta_max_2 * s_x;
        const float cos_theta = safe_sqrt(1.f - sin_theta_2);
        const float cos_alpha = sin_theta_2 * inv_sin_theta_max + cos_theta * safe_sqrt(fmaf(-sin_theta_2, sqr(inv_sin_theta_max), 1.f));
        const float sin_alpha = safe_sqrt(fmaf(-cos_alpha, cos_alpha, 1.f));
        float sin_phi, cos_phi; sincos_(s_y * (2.f * kPi), sin_phi, cos_phi);
        const V3 fn = dc_v * -inv_dc; V3 fs, ft;
        coordinate_system(fn, fs, ft);
        dloc = vfma(fn, cos_alpha, vfma(ft, sin_phi * sin_alpha, fs * (cos_phi * sin_alpha)));
        pdf = uniform_cone_pdf(cos_theta_max);
    } else {   // warp::square_to_uniform_sphere (warp.h:250-255)
        const float z = fmaf(-2.f, s_y, 1.f), r = safe_sqrt(fmaf(-z, z, 1.f)); float sn, cs;
        sincos_(2.f * kPi * s_x, sn, cs);
        dloc = mk(r * cs, r * sn, z);
        pdf = 0.f;
    }
    p = vfma(dloc, radius, center); dd = p - ref;
    const float dist2 = dot(dd, dd);
    dist = sqrtf(dist2);
    dd = dd * rcp(dist);
    if (outside) { if (dist == 0.f) pdf = 0.f; }
    else pdf = sh.inv_area * dist2 / fabsf(dot(dd, dloc));
    n = flip ? -dloc : dloc;
}
// Sphere::pdf_direction (sphere.cpp:298-310)
DTOF_D float sphere_pdf_direction(const DShape &sh, V3 ref, V3 ds_d, V3 ds_n, float ds_dist) {
    const V3 center = mk(sh.n[0], sh.n[1], sh.n[2]);
    const float sin_alpha = sh.dp_du[0] * rcp(norm(center - ref)), cos_alpha = safe_sqrt(1.f - sin_alpha * sin_alpha);
    return sin_alpha < 0.99999994f ? uniform_cone_pdf(cos_alpha) : sh.inv_area * sqr(ds_dist) / fabsf(dot(ds_d, ds_n));
}
// RoughPlastic::lerp_gather (roughplastic.cpp:373-383) on the 64-entry transmittance table
DTOF_D float lerp_gather64(const float *data, float x) {
    x *= 63.f;
    uint32_t index = (uint32_t) x; if (index > 62u) index = 62u;
    const float v0 = data[index], v1 = data[index + 1], t = x - (float) index;
    return fmaf(v1, t, fmaf(-v0, t, v0));                        // dr::lerp(v0, v1, t)
}
// RoughPlastic::eval (:333-371) and pdf (:385-421) for wi.z > 0 and wo.z > 0
DTOF_D void rough_plastic_eval_pdf(Ggx g, const DShape *sh, const float *table, V3 diff, V3 wi, V3 wo, float t_i, float prob_specular,
                                   float prob_diffuse, V3 &value, float &pdf) {
    const V3 H = normalize(wo + wi);
    const float D = ggx_eval(g, H);
    float F, t1, t2, t3; fresnel_dielectric(dot(wi, H), sh->diel_eta, F, t1, t2, t3);
    const float G = ggx_smith_g1(g, wi, H) * ggx_smith_g1(g, wo, H);
    const float spec = F * D * G / (4.f * wi.z);
    const float t_o = lerp_gather64(table, wo.z);
    const float k = kInvPi * sh->inv_eta_2 * wo.z * t_i * t_o;
    value = mk(spec * sh->spec_refl[0] + diff.x * k, spec * sh->spec_refl[1] + diff.y * k, spec * sh->spec_refl[2] + diff.z * k);
    float result = D * ggx_smith_g1(g, wi, H) / (4.f * wi.z);
    result *= prob_specular;
    pdf = result + prob_diffuse * (kInvPi * wo.z);
}
// RoughDielectric::eval_pdf (roughdielectric.cpp:503-611), GGX + visible normals, TransportMode::Radiance
DTOF_D void rough_dielectric_eval_pdf(Ggx g, const DShape *sh, V3 wi, V3 wo, V3 &value, float &pdf) {
    const float cti = wi.z, cto = wo.z, m_eta = sh->diel_eta, m_inv_eta = rcp(m_eta);
    const bool reflect = cti * cto > 0.f;
    const float eta = cti > 0.f ? m_eta : m_inv_eta, inv_eta = cti > 0.f ? m_inv_eta : m_eta;
    V3 m = normalize(wi + wo * (reflect ? 1.f : eta));
    m = mk(mulsign(m.x, m.z), mulsign(m.y, m.z), mulsign(m.z, m.z));
    const float dwm = dot(wi, m), dom = dot(wo, m);
    const bool active = cti != 0.f && dwm * cti > 0.f && dom * cto > 0.f;
    const float D = ggx_eval(g, m);
    float F, t1, t2, t3; fresnel_dielectric(dwm, m_eta, F, t1, t2, t3);
    const float G = ggx_smith_g1(g, wi, m) * ggx_smith_g1(g, wo, m);
    value = mk(0, 0, 0); pdf = 0.f;
    if (!active) return;
    if (reflect) {
        const float v = F * D * G / (4.f * fabsf(cti));
        value = mk(v * sh->spec_refl[0], v * sh->spec_refl[1], v * sh->spec_refl[2]);
    } else {
        const float scale = sqr(inv_eta);
        const float v = fabsf((scale * (1.f - F) * D * G * eta * eta * dwm * dom) / (cti * sqr(dwm + eta * dom)));
        value = mk(v * sh->spec_trans[0], v * sh->spec_trans[1], v * sh->spec_trans[2]);
    }
    float p = ggx_pdf(g, mk(mulsign(wi.x, cti), mulsign(wi.y, cti), mulsign(wi.z, cti)), m);
    p *= reflect ? F : 1.f - F;
    const float dwh_dwo = reflect ? rcp(4.f * dom) : (eta * eta * dom) / sqr(dwm + eta * dom);
    pdf = p * fabsf(dwh_dwo);
}
// fresnel_conductor -- include/mitsuba/render/fresnel.h:93-117 (one colour channel)
DTOF_D float fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) {
    const float cos_theta_i_2 = cos_theta_i * cos_theta_i, sin_theta_i_2 = 1.f - cos_theta_i_2, sin_theta_i_4 = sin_theta_i_2 * sin_theta_i_2;
    const float temp_1 = eta_r * eta_r - eta_i * eta_i - sin_theta_i_2,
                a_2_pb_2 = safe_sqrt(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r),
                a = safe_sqrt(.5f * (a_2_pb_2 + temp_1));
    const float term_1 = a_2_pb_2 + cos_theta_i_2, term_2 = 2.f * cos_theta_i * a;
    const float r_s = (term_1 - term_2) / (term_1 + term_2);
    const float term_3 = a_2_pb_2 * cos_theta_i_2 + sin_theta_i_4, term_4 = term_2 * sin_theta_i_2;
    const float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return 0.5f * (r_s + r_p);
}
DTOF_D float mis_weight(float a, float b) { a *= a; b *= b; float w = a / (a + b); return isfinite(w) ? w : 0.f; }

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
template <bool LDS, int MODE, bool AREA, int KMAX, bool MESH, bool SPEC>
__global__ __launch_bounds__(kShadeBlock) void k_shade(const uint8_t *scene, uint32_t scene_bytes, uint32_t stage_words, RenderParams rp, Queues q,
                                                  const uint32_t *qin, const uint32_t *count_in,
                                                  uint32_t *qout, uint32_t *alive_out, uint32_t *shadow_out, uint32_t depth,
                                                  uint32_t trace_next, LaneDebug *dbg) {
    constexpr bool FUSED = MODE != 0, FIRST = MODE == 2;
    extern __shared__ uint4 lds[];
    __shared__ uint32_t s_cnt[4];
    uint32_t *stack = (uint32_t *) (lds + stage_words) + threadIdx.x;
    const uint32_t seg = blockIdx.x;
    const uint32_t count = seg_count(count_in, seg, rp.n_lanes);
    uint32_t n_alive = 0, n_shadow = 0;
    if (count != 0) {
    const uint8_t *base = LDS ? stage_scene(scene, scene_bytes, lds) : scene;
    SceneView sv = make_view(base);
    for (uint32_t cbase = 0; cbase < count; cbase += kShadeBlock) {
    uint32_t j = cbase + threadIdx.x;
    bool in_range = j < count;
    bool alive = false, want_shadow = false;
    uint32_t l = 0;
    float4 sha, shb, nra, nrb; float3 cand[KMAX];
    float3 rbase[KMAX];   // FIRST: the result a lane ends this launch with if its NEE candidate is not committed
#pragma unroll
    for (int k = 0; k < KMAX; ++k) rbase[k] = make_float3(0.f, 0.f, 0.f);
    if (in_range) {
        l = qin ? qin[seg * kSeg + j] : seg * kSeg + j;
        uint32_t hid; float4 ra, rb, st; uint4 hh; Rng main, path;
        if (FIRST) {
            const PrimaryLane pl = generate_lane(rp, global_lane(rp, rp.lane_base + l));
            ra = pl.ray_a; rb = pl.ray_b; main = pl.main; path = pl.path; st = make_float4(1.f, 1.f, 1.f, 0.f);
            q.pos[l] = pl.pos;
            q.rng_b[l] = make_uint2((uint32_t) (main.inc >> 1), (uint32_t) (path.inc >> 1));
            if (dbg) {
                LaneDebug &o = dbg[l];
                o.time = ra.w; o.ray_o[0] = ra.x; o.ray_o[1] = ra.y; o.ray_o[2] = ra.z; o.ray_d[0] = rb.x; o.ray_d[1] = rb.y; o.ray_d[2] = rb.z;
            }
            Hit h;
            bool found = trace_scene<false, MESH>(sv, stack, mk(ra.x, ra.y, ra.z), mk(rb.x, rb.y, rb.z), ra.w, rb.w, h);
            hh = make_uint4(f2u(h.t), f2u(h.u), f2u(h.v), h.prim);
            hid = found ? (h.obj | (h.shape << 24)) : 0xffffffffu;
        } else {
            hid = q.hit_id[l];
            if (hid != 0xffffffffu) {
                ra = q.ray_a[l]; rb = q.ray_b[l]; hh = q.hit[l]; st = q.st_a[l];
                const uint4 rs = q.rng_a[l]; const uint2 ri = q.rng_b[l];
                main.state = (uint64_t) rs.x | ((uint64_t) rs.y << 32); main.inc = ((uint64_t) ri.x << 1) | 1u;
                path.state = (uint64_t) rs.z | ((uint64_t) rs.w << 32); path.inc = ((uint64_t) ri.y << 1) | 1u;
            }
        }
        if (hid != 0xffffffffu) {   // a miss ends the path (active_next = false, dopplertofpath.cpp:171)
            V3 o = mk(ra.x, ra.y, ra.z), d = mk(rb.x, rb.y, rb.z); float time = ra.w;
            V3 thr = mk(st.x, st.y, st.z); float path_length = st.w;
            float eta_path = 1.f; bool prev_delta = depth == 0;   // dopplertofpath.cpp:103-108: eta = 1, prev_bsdf_delta = true
            if (SPEC && depth > 0) { const float2 sc = q.st_c[l]; eta_path = sc.x; prev_delta = sc.y != 0.f; }
            bool correlate = (depth + 1) < rp.path_correlation_depth;
            const bool plain = rp.integrator != 0;   // `path`: no modulation weight
            const bool single = plain || rp.sampler_kind != SAMPLER_CORRELATED;   // main stream only (path.cpp:197,213-214,273; sampler.h:141-144)
            float t = u2f(hh.x);
            path_length += t * eta_path;   // dopplertofpath.cpp:141 (eta stays 1 without dielectrics)
            bool active_next = depth + 1 < rp.max_depth;

            Surface si;
            compute_surface<MESH>(sv, hid & 0xffffffu, hid >> 24, hh.w, t, u2f(hh.y), u2f(hh.z), o, d, time, si);
            const DShape *sh = si.shape;

            const float pmf = sv.n_emitters ? 1.f / (float) sv.n_emitters : 0.f;   // m_emitter_pmf (scene.cpp:96)
            // ---- direct emission (dopplertofpath.cpp:150-168 / path.cpp): the hit shape carries an area emitter
            bool res_dirty = false;
            float4 rcur[KMAX];
            if (AREA) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) rcur[k] = FIRST ? make_float4(0.f, 0.f, 0.f, 0.f) : q.res[(size_t) k * q.capacity + l];
                if (sh->flags & SF_EMITTER) {
                    float4 pb = depth > 0 ? q.st_b[l] : make_float4(0.f, 0.f, 0.f, 1.f);   // prev_si.p, prev_bsdf_pdf
                    V3 rel = si.p - mk(pb.x, pb.y, pb.z);                      // DirectionSample(scene, si, prev_si), records.h:173-180
                    float dist = norm(rel);
                    V3 dsd = rel * rcp(dist);
                    float em_pdf = 0.f;
                    if (!prev_delta) {                                          // !prev_bsdf_delta: AreaLight::pdf_direction (area.cpp:161-180)
                        float dp = dot(dsd, si.sh_n);   // ds.n = si.sh_frame.n (PositionSample(si), records.h:63-65)
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
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) {
                        V3 v = le * mis_bsdf;
                        if (!plain) v = v * modulation_weight(rp, rp.phase[k], time, path_length);
                        rcur[k] = make_float4(fmaf(thr.x, v.x, rcur[k].x), fmaf(thr.y, v.y, rcur[k].y), fmaf(thr.z, v.z, rcur[k].z), 0.f);
                    }
                    res_dirty = true;
                }
            }

            // ---- emitter sampling (scene.cpp:235-291; point.cpp:118-147; area.cpp:116-159 + shape.cpp:370-384 + rectangle.cpp:152-166)
            float e1 = single ? next_f32(main) : next_correlate(main, path, correlate), e2 = single ? next_f32(main) : next_correlate(main, path, correlate);
            // has_flag(bsdf->flags(), BSDFFlags::Smooth) (:178): diffuse, (rough)plastic and roughconductor have a smooth lobe
            bool active_em = active_next && sv.n_emitters > 0 && (!SPEC || sh->bsdf == BSDF_DIFFUSE || sh->bsdf == BSDF_PLASTIC || sh->bsdf == BSDF_ROUGHCONDUCTOR || sh->bsdf == BSDF_ROUGHPLASTIC || sh->bsdf == BSDF_ROUGHDIELECTRIC);
            V3 em_weight = mk(0, 0, 0), wo = mk(0, 0, 0); float ds_dist = 0.f, ds_pdf = 0.f; bool ds_delta = true;
            if (active_em) {
                uint32_t ne = sv.n_emitters, idx = 0; float em_w = 1.f, sx = e1;
                if (ne > 1) { float scaled = e1 * (float) ne; idx = (uint32_t) scaled; if (idx > ne - 1) idx = ne - 1; em_w = (float) ne; sx = scaled - (float) idx; }
                const DEmitter &em = sv.emitters[idx];
                V3 dsp, dd; bool em_active = true;
                if (em.kind == EMITTER_POINT) {
                    dsp = mk(em.pos[0], em.pos[1], em.pos[2]);
                    dd = dsp - si.p;
                    float dist2 = dot(dd, dd), inv_dist = rsqrt_(dist2);
                    ds_dist = sqrtf(dist2);
                    dd = dd * inv_dist;
                    float id2 = sqr(inv_dist);
                    em_weight = mk(em.intensity[0] * id2, em.intensity[1] * id2, em.intensity[2] * id2);
                    ds_pdf = 1.f;
                } else {
                    const DShape &es = sv.shapes[em.shape];
                    V3 en;
                    if (MESH && es.kind == SHAPE_SPHERE) {   // Sphere overrides Shape::sample_direction
                        sphere_sample_direction(es, si.p, sx, e2, dsp, en, dd, ds_dist, ds_pdf);
                    } else {
                        if (!MESH || es.kind == SHAPE_RECT) {
                            dsp = xf_point(es.to_world, mk(sx * 2.f - 1.f, e2 * 2.f - 1.f, 0.f));
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
            const float sample_1 = single ? next_f32(main) : next_correlate(main, path, correlate); (void) sample_1;
            float s2x = single ? next_f32(main) : next_correlate(main, path, correlate), s2y = single ? next_f32(main) : next_correlate(main, path, correlate);

            // ---- BSDF eval_pdf + sample (twosided.cpp:111-148,219-258; diffuse.cpp:101-125,160-180)
            bool twosided = sh->flags & SF_TWOSIDED;
            float wiz = si.wi.z, woz = wo.z;
            if (twosided) { woz = mulsign(woz, wiz); wiz = fabsf(wiz); }
            V3 refl = mk(sh->refl[0], sh->refl[1], sh->refl[2]);
            V3 bsdf_val = mk(0, 0, 0), bsdf_weight = mk(0, 0, 0), bs_wo = mk(0, 0, 0);
            float bsdf_pdf = 0.f, bs_pdf = 0.f, bs_eta = 0.f; bool bs_delta = false;
            if (SPEC && sh->bsdf == BSDF_CONDUCTOR) {
                // SmoothConductor::sample (conductor.cpp:226-277) under TwoSidedBRDF::sample; eval / pdf of a delta lobe are zero
                const float cos_theta_i = twosided ? fabsf(si.wi.z) : si.wi.z;
                if (cos_theta_i > 0.f) {
                    bs_wo = mk(-si.wi.x, -si.wi.y, si.wi.z);   // reflect(wi); the two-sided flips of wi.z and wo.z cancel
                    bs_eta = 1.f; bs_pdf = 1.f; bs_delta = true;
                    bsdf_weight = mk(sh->spec_refl[0] * fresnel_conductor(cos_theta_i, sh->cond_eta[0], sh->cond_k[0]),
                                     sh->spec_refl[1] * fresnel_conductor(cos_theta_i, sh->cond_eta[1], sh->cond_k[1]),
                                     sh->spec_refl[2] * fresnel_conductor(cos_theta_i, sh->cond_eta[2], sh->cond_k[2]));
                }
            } else if (SPEC && sh->bsdf == BSDF_DIELECTRIC) {
                // SmoothDielectric::sample (dielectric.cpp:231-338), TransportMode::Radiance
                float r_i, cos_theta_t, eta_it, eta_ti;
                fresnel_dielectric(si.wi.z, sh->diel_eta, r_i, cos_theta_t, eta_it, eta_ti);
                const float t_i = 1.f - r_i;
                const bool selected_r = sample_1 <= r_i;
                bs_pdf = selected_r ? r_i : t_i; bs_delta = true;
                bs_wo = selected_r ? mk(-si.wi.x, -si.wi.y, si.wi.z) : mk(-eta_ti * si.wi.x, -eta_ti * si.wi.y, cos_theta_t);
                bs_eta = selected_r ? 1.f : eta_it;
                const float f2 = sqr(eta_ti);
                bsdf_weight = selected_r ? mk(sh->spec_refl[0], sh->spec_refl[1], sh->spec_refl[2])
                                         : mk(sh->spec_trans[0] * f2, sh->spec_trans[1] * f2, sh->spec_trans[2] * f2);
            } else if (SPEC && sh->bsdf == BSDF_THINDIELECTRIC) {
                // ThinDielectric::sample (thindielectric.cpp:173-226): the reflectance of the slab with all internal bounces, wo = -wi
                float r, t1, t2, t3;
                fresnel_dielectric(fabsf(si.wi.z), sh->diel_eta, r, t1, t2, t3);
                r *= 2.f / (1.f + r);
                const bool selected_r = sample_1 <= r;
                bs_pdf = selected_r ? r : 1.f - r; bs_delta = true; bs_eta = 1.f;
                bs_wo = selected_r ? mk(-si.wi.x, -si.wi.y, si.wi.z) : mk(-si.wi.x, -si.wi.y, -si.wi.z);
                bsdf_weight = selected_r ? mk(sh->spec_refl[0], sh->spec_refl[1], sh->spec_refl[2]) : mk(sh->spec_trans[0], sh->spec_trans[1], sh->spec_trans[2]);
            } else if (SPEC && sh->bsdf == BSDF_ROUGHDIELECTRIC) {
                // RoughDielectric::eval_pdf / sample (roughdielectric.cpp:240-346,503-611): glossy reflection and transmission lobes
                const Ggx g = ggx_make(sh->alpha_u, sh->alpha_v);
                const V3 wi = si.wi;
                if (active_em) rough_dielectric_eval_pdf(g, sh, wi, wo, bsdf_val, bsdf_pdf);
                if (wi.z != 0.f) {
                    float mpdf;
                    const V3 m = ggx_sample(g, mk(mulsign(wi.x, wi.z), mulsign(wi.y, wi.z), mulsign(wi.z, wi.z)), s2x, s2y, mpdf);
                    const float dwm = dot(wi, m);
                    float F, cos_theta_t, eta_it, eta_ti; fresnel_dielectric(dwm, sh->diel_eta, F, cos_theta_t, eta_it, eta_ti);
                    const bool selected_r = sample_1 <= F;
                    bs_pdf = mpdf * (selected_r ? F : 1.f - F);
                    bs_eta = selected_r ? 1.f : eta_it;
                    float dwh_dwo; V3 w;
                    if (selected_r) {
                        bs_wo = mk(fmaf(m.x, 2.f * dwm, -wi.x), fmaf(m.y, 2.f * dwm, -wi.y), fmaf(m.z, 2.f * dwm, -wi.z));   // reflect(wi, m)
                        w = mk(sh->spec_refl[0], sh->spec_refl[1], sh->spec_refl[2]);
                        dwh_dwo = rcp(4.f * dot(bs_wo, m));
                    } else {
                        const float k = fmaf(dwm, eta_ti, cos_theta_t);                                                         // refract(wi, m, cos_theta_t, eta_ti)
                        bs_wo = mk(fmaf(m.x, k, -(wi.x * eta_ti)), fmaf(m.y, k, -(wi.y * eta_ti)), fmaf(m.z, k, -(wi.z * eta_ti)));
                        const float f2 = sqr(eta_ti);
                        w = mk(f2 * sh->spec_trans[0], f2 * sh->spec_trans[1], f2 * sh->spec_trans[2]);
                        const float dom = dot(bs_wo, m);
                        dwh_dwo = (sqr(bs_eta) * dom) / sqr(dwm + bs_eta * dom);
                    }
                    const float g1 = ggx_smith_g1(g, bs_wo, m);
                    bs_pdf *= fabsf(dwh_dwo);
                    if (mpdf != 0.f) bsdf_weight = w * g1;
                }
            } else if (SPEC && sh->bsdf == BSDF_ROUGHCONDUCTOR) {
                // RoughConductor::eval / pdf / sample (roughconductor.cpp:229-415), GGX + visible normals, under TwoSidedBRDF
                V3 wi = si.wi, wo_l = wo;
                if (twosided && wi.z < 0.f) { wi.z = -wi.z; wo_l.z = -wo_l.z; }
                const Ggx g = ggx_make(sh->alpha_u, sh->alpha_v);
                if (wi.z > 0.f && wo_l.z > 0.f) {
                    const V3 H = normalize(wo_l + wi);
                    const float D = ggx_eval(g, H);
                    if (D != 0.f) {
                        const float G = ggx_smith_g1(g, wi, H) * ggx_smith_g1(g, wo_l, H);
                        const float result = D * G / (4.f * wi.z), c = dot(wi, H);
                        bsdf_val = mk(fresnel_conductor(c, sh->cond_eta[0], sh->cond_k[0]) * (result * sh->spec_refl[0]),
                                      fresnel_conductor(c, sh->cond_eta[1], sh->cond_k[1]) * (result * sh->spec_refl[1]),
                                      fresnel_conductor(c, sh->cond_eta[2], sh->cond_k[2]) * (result * sh->spec_refl[2]));
                    }
                    if (dot(wi, H) > 0.f && dot(wo_l, H) > 0.f) bsdf_pdf = ggx_eval(g, H) * ggx_smith_g1(g, wi, H) / (4.f * wi.z);
                }
                if (wi.z > 0.f) {
                    float mpdf;
                    const V3 m = ggx_sample(g, wi, s2x, s2y, mpdf);
                    const float dwm = dot(wi, m);
                    const V3 r = mk(fmaf(m.x, 2.f * dwm, -wi.x), fmaf(m.y, 2.f * dwm, -wi.y), fmaf(m.z, 2.f * dwm, -wi.z));   // reflect(wi, m)
                    bs_wo = r; bs_eta = 1.f;
                    const bool ok = mpdf != 0.f && r.z > 0.f;
                    const float weight = ggx_smith_g1(g, r, m);
                    bs_pdf = mpdf / (4.f * dot(r, m));
                    if (ok) bsdf_weight = mk(fresnel_conductor(dwm, sh->cond_eta[0], sh->cond_k[0]) * (weight * sh->spec_refl[0]),
                                             fresnel_conductor(dwm, sh->cond_eta[1], sh->cond_k[1]) * (weight * sh->spec_refl[1]),
                                             fresnel_conductor(dwm, sh->cond_eta[2], sh->cond_k[2]) * (weight * sh->spec_refl[2]));
                    if (twosided && si.wi.z < 0.f) bs_wo.z = -bs_wo.z;
                }
            } else if (SPEC && sh->bsdf == BSDF_ROUGHPLASTIC) {
                // RoughPlastic::eval / pdf / sample (roughplastic.cpp:259-421), GGX + visible normals, under TwoSidedBRDF
                V3 wi = si.wi, wo_l = wo;
                if (twosided && wi.z < 0.f) { wi.z = -wi.z; wo_l.z = -wo_l.z; }
                const Ggx g = ggx_make(sh->alpha_u, sh->alpha_u);
                const float *table = (const float *) (sv.base + sh->rough_table);
                const float w = sh->spec_sampling_weight, ir = sh->fdr_int;
                const V3 diff = sh->nonlinear ? mk(refl.x / (1.f - refl.x * ir), refl.y / (1.f - refl.y * ir), refl.z / (1.f - refl.z * ir))
                                              : mk(refl.x / (1.f - ir), refl.y / (1.f - ir), refl.z / (1.f - ir));
                if (wi.z > 0.f) {
                    const float t_i = lerp_gather64(table, wi.z);
                    float prob_specular = (1.f - t_i) * w, prob_diffuse = t_i * (1.f - w);
                    prob_specular = prob_specular / (prob_specular + prob_diffuse);
                    prob_diffuse = 1.f - prob_specular;
                    if (wo_l.z > 0.f) rough_plastic_eval_pdf(g, sh, table, diff, wi, wo_l, t_i, prob_specular, prob_diffuse, bsdf_val, bsdf_pdf);
                    if (sample_1 < prob_specular) {
                        float mpdf; const V3 m = ggx_sample(g, wi, s2x, s2y, mpdf);
                        const float dwm = dot(wi, m);
                        bs_wo = mk(fmaf(m.x, 2.f * dwm, -wi.x), fmaf(m.y, 2.f * dwm, -wi.y), fmaf(m.z, 2.f * dwm, -wi.z));   // reflect(wi, m)
                    } else bs_wo = cosine_hemisphere(s2x, s2y);
                    bs_eta = 1.f;
                    V3 value = mk(0, 0, 0);
                    if (bs_wo.z > 0.f) rough_plastic_eval_pdf(g, sh, table, diff, wi, bs_wo, t_i, prob_specular, prob_diffuse, value, bs_pdf);
                    if (bs_pdf > 0.f) bsdf_weight = value * rcp(bs_pdf);                  // Spectrum / Float = multiplication by the reciprocal
                    if (twosided && si.wi.z < 0.f) bs_wo.z = -bs_wo.z;
                }
            } else if (SPEC && sh->bsdf == BSDF_PLASTIC) {
                // SmoothPlastic::eval / pdf / sample (plastic.cpp:219-360) under TwoSidedBRDF; wiz / woz are already flipped
                float f_i, t1, t2, t3;
                fresnel_dielectric(wiz, sh->diel_eta, f_i, t1, t2, t3);
                const float w = sh->spec_sampling_weight, fdr = sh->fdr_int;
                const V3 diff = sh->nonlinear ? mk(refl.x / (1.f - refl.x * fdr), refl.y / (1.f - refl.y * fdr), refl.z / (1.f - refl.z * fdr))
                                              : mk(refl.x / (1.f - fdr), refl.y / (1.f - fdr), refl.z / (1.f - fdr));
                if (wiz > 0.f && woz > 0.f) {
                    float f_o; fresnel_dielectric(woz, sh->diel_eta, f_o, t1, t2, t3);
                    const float k = kInvPi * woz * sh->inv_eta_2 * (1.f - f_i) * (1.f - f_o);
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
                    if (sample_1 < prob_specular) {
                        bs_wo = mk(-si.wi.x, -si.wi.y, wiz);
                        bs_pdf = prob_specular; bs_delta = true;
                        const float value = f_i / bs_pdf;
                        bsdf_weight = mk(value * sh->spec_refl[0], value * sh->spec_refl[1], value * sh->spec_refl[2]);
                    } else {
                        bs_wo = cosine_hemisphere(s2x, s2y);
                        bs_pdf = prob_diffuse * (kInvPi * bs_wo.z);
                        float f_o; fresnel_dielectric(bs_wo.z, sh->diel_eta, f_o, t1, t2, t3);
                        const float k = sh->inv_eta_2 * (1.f - f_i) * (1.f - f_o) / prob_diffuse;
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
            // ---- emitter contribution candidate (dopplertofpath.cpp:214-226); committed by k_shadow if unoccluded
            if (active_em) {
                const float mis_em = ds_delta ? 1.f : mis_weight(ds_pdf, bsdf_pdf);   // dopplertofpath.cpp:218-219
                bool nonzero = false;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) {
                    float4 r = AREA ? rcur[k] : (FIRST ? make_float4(0.f, 0.f, 0.f, 0.f) : q.res[(size_t) k * q.capacity + l]);
                    V3 v = mk(bsdf_val.x * em_weight.x * mis_em, bsdf_val.y * em_weight.y * mis_em, bsdf_val.z * em_weight.z * mis_em);
                    if (!plain) { float lw = modulation_weight(rp, rp.phase[k], time, path_length + ds_dist); v = v * lw; }
                    float3 c = make_float3(fmaf(thr.x, v.x, r.x), fmaf(thr.y, v.y, r.y), fmaf(thr.z, v.z, r.z));
                    cand[k] = c;
                    nonzero |= f2u(c.x) != f2u(r.x) || f2u(c.y) != f2u(r.y) || f2u(c.z) != f2u(r.z);
                }
                want_shadow = nonzero;   // a candidate identical to the current result needs no visibility test
            }
            if (res_dirty) {   // the emitter-hit term stands whether or not the NEE candidate is later committed
#pragma unroll
                for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) {
                    if (FIRST) rbase[k] = make_float3(rcur[k].x, rcur[k].y, rcur[k].z);
                    else q.res[(size_t) k * q.capacity + l] = rcur[k];
                }
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
            bool rr_continue = (single ? next_f32(main) : next_correlate(main, path, correlate)) < rr_prob;
            if (rr_active) thr = thr * rcp(rr_prob);
            alive = active_next && (!rr_active || rr_continue) && thr_max != 0.f;
            if (alive) {
                nra = make_float4(no.x, no.y, no.z, time); nrb = make_float4(nd.x, nd.y, nd.z, kLargest);
                q.ray_a[l] = nra;
                q.ray_b[l] = nrb;
                q.st_a[l] = make_float4(thr.x, thr.y, thr.z, path_length);
                if (AREA) q.st_b[l] = make_float4(si.p.x, si.p.y, si.p.z, bs_pdf);   // prev_si, prev_bsdf_pdf (:256-257)
                if (SPEC) q.st_c[l] = make_float2(eta, bs_delta ? 1.f : 0.f);         // eta, prev_bsdf_delta (:252,258)
                q.rng_a[l] = make_uint4((uint32_t) main.state, (uint32_t) (main.state >> 32), (uint32_t) path.state, (uint32_t) (path.state >> 32));
            }
        }
    }
    uint32_t slot = block_append(alive, s_cnt, n_alive);
    if (alive) qout[seg * kSeg + slot] = l;
    if (FUSED) {
        bool commit = false;
        if (want_shadow) {   // test_visibility (scene.cpp:266-271): an unoccluded sample commits its candidate result
            Hit hs;
            commit = !trace_scene<true, MESH>(sv, stack, mk(sha.x, sha.y, sha.z), mk(shb.x, shb.y, shb.z), shb.w, sha.w, hs);
        }
        if (FIRST ? in_range : commit) {   // FIRST: every lane's result is defined here (nothing zeroed it beforehand)
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) {
                const float3 v = commit ? cand[k] : rbase[k];
                q.res[(size_t) k * q.capacity + l] = make_float4(v.x, v.y, v.z, 0.f);
            }
        }
        if (alive && trace_next) {   // closest hit of the continuation ray, consumed by the next bounce
            Hit h;
            bool found = trace_scene<false, MESH>(sv, stack, mk(nra.x, nra.y, nra.z), mk(nrb.x, nrb.y, nrb.z), nra.w, nrb.w, h);
            q.hit[l] = make_uint4(f2u(h.t), f2u(h.u), f2u(h.v), h.prim);
            q.hit_id[l] = found ? (h.obj | (h.shape << 24)) : 0xffffffffu;
        }
        n_shadow += (uint32_t) __popcll(__ballot(want_shadow)) * ((threadIdx.x & 63) == 0 ? 1u : 0u);   // per-wave partial (stats only)
    } else {
        uint32_t sslot = seg * kSeg + block_append(want_shadow, s_cnt, n_shadow);
        if (want_shadow) {
            q.sh_a[sslot] = sha; q.sh_b[sslot] = shb;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets)
                q.sh_c[(size_t) k * q.capacity + sslot] = make_float4(cand[k].x, cand[k].y, cand[k].z, u2f(l));
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
    if (threadIdx.x == 0) { alive_out[seg] = n_alive; shadow_out[seg] = n_shadow; }
}

// ---------------------------------------------------------------------------- shadow
template <bool LDS, bool MESH, int BLOCK>
__global__ __launch_bounds__(BLOCK, BLOCK == 64 ? 6 : 1) void k_shadow(const uint8_t *scene, uint32_t scene_bytes, uint32_t stage_words, RenderParams rp,
                                                  Queues q, const uint32_t *count_in) {
    constexpr uint32_t kBlock = BLOCK, kSub = kSeg / BLOCK;
    extern __shared__ uint4 lds[];
    uint32_t seg = blockIdx.x / kSub, sub = blockIdx.x % kSub;
    uint32_t count = count_in[seg];
    if (sub * kBlock >= count) return;
    const uint8_t *base = LDS ? stage_scene(scene, scene_bytes, lds) : scene;
    uint32_t *stack = (uint32_t *) (lds + stage_words) + threadIdx.x;
    SceneView sv = make_view(base);
    uint32_t j = sub * kBlock + threadIdx.x;
    if (j >= count) return;
    uint32_t i = seg * kSeg + j;
    float4 a = q.sh_a[i], b = q.sh_b[i];
    Hit h;
    bool occluded = trace_scene<true, MESH>(sv, stack, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), b.w, a.w, h);
    if (!occluded) {
#pragma unroll
        for (int k = 0; k < kMaxOffsets; ++k) if (k < rp.n_offsets) {
            float4 c = q.sh_c[(size_t) k * q.capacity + i];
            uint32_t l = f2u(c.w);
            q.res[(size_t) k * q.capacity + l] = make_float4(c.x, c.y, c.z, 0.f);
        }
    }
}

// ---------------------------------------------------------------------------- velocity
// VelocityIntegrator::sample (src/integrators/velocity.cpp:125-142): the primary ray is intersected at time 0 and at
// time T; radial velocity = (t2 - t1) / T where both hit, 0 otherwise, in all three channels.
template <bool LDS>
__global__ __launch_bounds__(kBlock) void k_velocity(const uint8_t *scene, uint32_t scene_bytes, uint32_t stage_words, RenderParams rp, Queues q) {
    extern __shared__ uint4 lds[];
    const uint8_t *base = LDS ? stage_scene(scene, scene_bytes, lds) : scene;
    uint32_t *stack = (uint32_t *) (lds + stage_words) + threadIdx.x;
    SceneView sv = make_view(base);
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= rp.n_lanes) return;
    float4 a = q.ray_a[i], b = q.ray_b[i];
    V3 o = mk(a.x, a.y, a.z), d = mk(b.x, b.y, b.z);
    Hit h1, h2;
    bool v1 = trace_scene<false, true>(sv, stack, o, d, 0.f, b.w, h1);
    bool v2 = trace_scene<false, true>(sv, stack, o, d, rp.T, b.w, h2);
    float vel = ((v2 ? h2.t : 0.f) - (v1 ? h1.t : 0.f)) * (1.0f / rp.T);
    vel = (v1 && v2) ? vel : 0.f;
    q.res[i] = make_float4(vel, vel, vel, 0.f);
}

// ---------------------------------------------------------------------------- splat
DTOF_D float tent(float x, float inv_r) { return fmax_(0.f, 1.f - fabsf(x * inv_r)); }
// ReconstructionFilter::eval: tent (tent.cpp:53-55) or gaussian (gaussian.cpp:94-96, polynomial branch)
DTOF_D float filter_weight(const RenderParams &rp, float x) {
    if (rp.filter == FILTER_GAUSSIAN) return fmax_(estrin10(sqr(x), rp.gauss_coeff), 0.f);
    if (rp.filter == FILTER_MITCHELL) {   // MitchellNetravaliFilter::eval (mitchell.cpp:47-67): coefficients in ScalarFloat, Horner with fmadd
        x = fabsf(x);
        const float x2 = x * x, x3 = x2 * x, B = rp.filter_b, C = rp.filter_c;
        const float a3 = (12.f - 9.f * B - 6.f * C), a2 = (-18.f + 12.f * B + 6.f * C), a0 = (6.f - 2.f * B),
                    b3 = (-B - 6.f * C), b2 = (6.f * B + 30.f * C), b1 = (-12.f * B - 48.f * C), b0 = (8.f * B + 24.f * C);
        const float r = (1.f / 6.f) * (x < 1.f ? fmaf(a3, x3, fmaf(a2, x2, a0)) : fmaf(b3, x3, fmaf(b2, x2, fmaf(b1, x, b0))));
        return x < 2.f ? r : 0.f;
    }
    if (rp.filter == FILTER_CATMULLROM) {   // CatmullRomFilter::eval (catmullrom.cpp:38-53): B = 0, C = 1/2, plain multiplies and adds
        x = fabsf(x);
        const float x2 = x * x, x3 = x2 * x, B = 0.f, C = .5f;
        const float r = (1.f / 6.f) * (x < 1.f ? (12.f - 9.f * B - 6.f * C) * x3 + (-18.f + 12.f * B + 6.f * C) * x2 + (6.f - 2.f * B)
                                               : (-B - 6.f * C) * x3 + (6.f * B + 30.f * C) * x2 + (-12.f * B - 48.f * C) * x + (8.f * B + 24.f * C));
        return x < 2.f ? r : 0.f;
    }
    return tent(x, rp.inv_radius);
}
// v + (v moved by the DPP control); lanes without a valid source (or in rows masked off) add 0
template <int CTRL, int ROW_MASK = 0xf>
DTOF_D float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

// Generic per-lane splat (any filter radius / any spp): direct float atomics.
DTOF_D void splat_lane(const RenderParams &rp, float *film, float spx, float spy, int pixel_x, int pixel_y, float r, float g, float b) {
    int W = rp.crop_w, H = rp.crop_h;
    if (rp.filter == FILTER_BOX) {
        // block->put(box_filter ? pos : sample_pos) (integrator.cpp:540-541): the box filter splats at the lane's own pixel
        int x = pixel_x, y = pixel_y;
        if ((unsigned) x < (unsigned) W && (unsigned) y < (unsigned) H) {
            float *p = film + 4 * ((size_t) y * W + x);
            atomicAdd(p, r); atomicAdd(p + 1, g); atomicAdd(p + 2, b); atomicAdd(p + 3, 1.f);
        }
        return;
    }
    int n = (int) ceilf(rp.filter_radius - .5f), cnt = 2 * n + 1;
    int pix = (int) floorf(spx) - n, piy = (int) floorf(spy) - n;
    float relx = (float) pix + .5f - spx, rely = (float) piy + .5f - spy;
    int lx = pix - rp.crop_x, ly = piy - rp.crop_y;
    for (int ys = 0; ys < cnt; ++ys) {
        float wy = filter_weight(rp, rely + (float) ys);
        for (int xs = 0; xs < cnt; ++xs) {
            float w = filter_weight(rp, relx + (float) xs) * wy;
            int x = lx + xs, y = ly + ys;
            if ((unsigned) x < (unsigned) W && (unsigned) y < (unsigned) H) {
                float *p = film + 4 * ((size_t) y * W + x);
                atomicAdd(p, r * w); atomicAdd(p + 1, g * w); atomicAdd(p + 2, b * w); atomicAdd(p + 3, w);
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_splat_generic(RenderParams rp, Queues q, float *film, size_t film_stride) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= rp.n_lanes) return;
    float2 p = q.pos[i];
    uint32_t lane = global_lane(rp, rp.lane_base + i);
    uint32_t pix = rp.spp_log2 != 0xffffffffu ? lane >> rp.spp_log2 : lane / rp.spp, W = (uint32_t) rp.crop_w;
    int py = (int) (pix / W), px = (int) (pix - W * (uint32_t) py);
    for (int k = 0; k < rp.n_offsets; ++k) {
        float4 r = q.res[(size_t) k * q.capacity + i];
        splat_lane(rp, film + (size_t) k * film_stride, p.x, p.y, px, py, r.x, r.y, r.z);
    }
}

// Fast path: tent filter with radius <= 1 (3x3 footprint) and power-of-two spp.  All samples of a
// pixel are SEG = min(spp,64) consecutive lanes of one wave and (almost always) share the footprint
// anchored at the pixel, so the 36 footprint values are reduced across the segment with DPP/shuffles
// and one lane issues the 36 atomics.  The rare sample whose float position rounds up to the next
// pixel splats by itself.
__global__ __launch_bounds__(kBlock) void k_splat_tent3(RenderParams rp, Queues q, float *film, size_t film_stride, uint32_t seg) {
    __shared__ float4 s_acc4[(kBlock / 2) * 9];
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    bool in_range = i < rp.n_lanes;
    uint32_t lane = global_lane(rp, rp.lane_base + (in_range ? i : 0));
    uint32_t pix = lane >> rp.spp_log2;
    uint32_t W = (uint32_t) rp.crop_w;
    int py = (int) (pix / W), px = (int) (pix - W * (uint32_t) py);
    float2 p = in_range ? q.pos[i] : make_float2(0.f, 0.f);
    int fx = (int) floorf(p.x) - rp.crop_x, fy = (int) floorf(p.y) - rp.crop_y;
    bool regular = in_range && fx == px && fy == py;
    float wx[3], wy[3];
    {
        float relx = (float) (px + rp.crop_x - 1) + .5f - p.x, rely = (float) (py + rp.crop_y - 1) + .5f - p.y;
#pragma unroll
        for (int a = 0; a < 3; ++a) { wx[a] = tent(relx + (float) a, rp.inv_radius); wy[a] = tent(rely + (float) a, rp.inv_radius); }
    }
    for (int k = 0; k < rp.n_offsets; ++k) {
        float4 r = in_range ? q.res[(size_t) k * q.capacity + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float *fk = film + (size_t) k * film_stride;
        if (in_range && !regular) splat_lane(rp, fk, p.x, p.y, px, py, r.x, r.y, r.z);
        float acc[36];
#pragma unroll
        for (int ys = 0; ys < 3; ++ys)
#pragma unroll
            for (int xs = 0; xs < 3; ++xs) {
                float w = regular ? wx[xs] * wy[ys] : 0.f;
                acc[4 * (3 * ys + xs) + 0] = r.x * w; acc[4 * (3 * ys + xs) + 1] = r.y * w;
                acc[4 * (3 * ys + xs) + 2] = r.z * w; acc[4 * (3 * ys + xs) + 3] = w;
            }
        // segment sums with DPP adds (one v_add_f32_dpp each, no LDS traffic): pairs, quads, row_ror 4/8 give every
        // lane of a 16-lane row the row total; row_bcast:15 / :31 carry it into the last lane of 32 / 64 lanes.
        // The total of a segment therefore ends up in the segment's LAST lane.
#pragma unroll
        for (int c = 0; c < 36; ++c) {
            float v = acc[c];
            v = dpp_add<0xb1>(v);
            if (seg >= 4) v = dpp_add<0x4e>(v);
            if (seg >= 8) v = dpp_add<0x124>(v);
            if (seg >= 16) v = dpp_add<0x128>(v);
            if (seg >= 32) v = dpp_add<0x142, 0xa>(v);
            if (seg >= 64) v = dpp_add<0x143, 0xc>(v);
            acc[c] = v;
        }
        // Stage the per-segment sums in LDS (9 ds_write_b128 by the segment's last lane) and let the whole block issue
        // the atomics: 36 values per segment become lanes of a few full wave-instructions instead of 36 single-lane
        // atomics that stall their wave once 16 are outstanding.
        if (k > 0) __syncthreads();
        const uint32_t sidx_mine = threadIdx.x / seg;
        if ((threadIdx.x & (seg - 1)) == seg - 1) {
#pragma unroll
            for (int c = 0; c < 9; ++c) s_acc4[sidx_mine * 9 + c] = make_float4(acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]);
        }
        __syncthreads();
        const uint32_t total = (kBlock / seg) * 36;
        const float *s_acc = (const float *) s_acc4;
        for (uint32_t idx = threadIdx.x; idx < total; idx += kBlock) {
            uint32_t sidx = idx / 36, c = idx - sidx * 36;
            uint32_t first_lane = blockIdx.x * kBlock + sidx * seg;
            if (first_lane >= rp.n_lanes) continue;
            uint32_t spix = global_lane(rp, rp.lane_base + first_lane) >> rp.spp_log2;
            int sy = (int) (spix / W), sx = (int) (spix - W * (uint32_t) sy);
            int x = sx - 1 + (int) ((c % 12) >> 2), y = sy - 1 + (int) (c / 12);
            float v = s_acc[idx];
            if ((unsigned) x < W && (unsigned) y < (unsigned) rp.crop_h && v != 0.f)
                atomicAdd(fk + 4 * ((size_t) y * W + (size_t) x) + (c & 3), v);
        }
    }
}

__global__ void k_develop(const float *film, float *rgb, int64_t n) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 f = ((const float4 *) film)[i];
    float w = f.w == 0.f ? 1.f : f.w;
    rgb[3 * i] = f.x / w; rgb[3 * i + 1] = f.y / w; rgb[3 * i + 2] = f.z / w;
}

__global__ void k_lane_dump(RenderParams rp, Queues q, LaneDebug *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rp.n_lanes) return;
    float2 p = q.pos[i]; float4 r = q.res[i];
    LaneDebug &o = out[i];
    o.sample_pos[0] = p.x; o.sample_pos[1] = p.y;
    o.rgb[0] = r.x; o.rgb[1] = r.y; o.rgb[2] = r.z;
}
// primary-ray snapshot taken right after generate (ray buffers are overwritten by the first shade)
__global__ void k_lane_dump_rays(RenderParams rp, Queues q, LaneDebug *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rp.n_lanes) return;
    float4 a = q.ray_a[i], b = q.ray_b[i];
    LaneDebug &o = out[i];
    o.time = a.w; o.ray_o[0] = a.x; o.ray_o[1] = a.y; o.ray_o[2] = a.z; o.ray_d[0] = b.x; o.ray_d[1] = b.y; o.ray_d[2] = b.z;
}

// ---------------------------------------------------------------------------- launchers
static inline uint32_t nblk(uint32_t n) { return (n + kBlock - 1) / kBlock; }
constexpr uint32_t kLdsSceneLimit = 48 * 1024;
// Block size of the unstaged k_trace / k_shadow instantiations.  One wave per block when some mesh has its own BLAS: those
// traversals are long and divergent, and a 256-thread block keeps its LDS and wave slots until its slowest wave is done (mesh room,
// 522 k triangles: 20.1 -> 16.6 ms per frame).  Scenes of many small objects (Domino: 1 025 instances of a 12-triangle cube) are
// faster with four waves sharing a CU's L1 on the same TLAS / object records (71.3 vs 75.3 ms).  DTOF_TRACE_BLOCK = 64 | 128 | 256
// overrides (experiments).
static inline uint32_t unstaged_block(const RenderParams &rp) {
    static const uint32_t env = [] { const char *e = getenv("DTOF_TRACE_BLOCK"); int b = e ? atoi(e) : 0; return (uint32_t) (b == 64 || b == 128 || b == 256 ? b : 0); }();
    return env ? env : (rp.has_blas ? 64u : (uint32_t) kBlock);
}

static inline uint32_t stack_bytes(uint32_t depth, uint32_t block = kBlock) { return (depth < 2 ? 2 : depth) * block * 4; }

void launch_generate(const RenderParams &rp, const Queues &q, hipStream_t s) {
    if (rp.n_lanes == 0) return;
    hipLaunchKernelGGL(k_generate, dim3(nblk(rp.n_lanes)), dim3(kBlock), 0, s, rp, q);
}
// the scene is staged into LDS if it is small AND leaves room for the traversal stacks within the 64 KiB a block may ask for
constexpr uint32_t kLdsBlockLimit = 64 * 1024;
static inline uint32_t stage_words_for(uint32_t scene_bytes, uint32_t stack = 0) {
    const uint32_t w = (scene_bytes + 15) / 16;
    return scene_bytes <= kLdsSceneLimit && w * 16 + stack + 64 <= kLdsBlockLimit ? w : 0;
}
static inline void check_lds(uint32_t lds) {
    if (lds + 64 > kLdsBlockLimit) throw std::runtime_error("the BVH of this scene is too deep for the LDS traversal stack");
}

// out[row] = sum of counts[row][0..n_seg): the statistics' per-iteration totals (kSumSlices blocks per row, one atomic each)
constexpr uint32_t kSumSlices = 32;
__global__ void k_sum_counts(const uint32_t *counts, uint32_t n_seg, unsigned long long *out) {
    __shared__ unsigned long long s_part[4];
    const uint32_t *row = counts + (size_t) blockIdx.x * n_seg;
    unsigned long long acc = 0;
    for (uint32_t i = blockIdx.y * blockDim.x + threadIdx.x; i < n_seg; i += blockDim.x * kSumSlices) acc += row[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[blockIdx.x], s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}
void launch_sum_counts(const uint32_t *counts, uint32_t n_seg, uint32_t n_rows, unsigned long long *out, hipStream_t s) {
    if (!n_rows) return;
    (void) hipMemsetAsync(out, 0, (size_t) n_rows * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_sum_counts, dim3(n_rows, kSumSlices), dim3(256), 0, s, counts, n_seg, out);
}
static inline uint32_t nseg(uint32_t n) { return (n + kSeg - 1) / kSeg; }
uint32_t segments_for(uint32_t n_lanes) { return nseg(n_lanes); }

void launch_trace(const uint8_t *scene, uint32_t scene_bytes, const RenderParams &rp, const Queues &q,
                  const uint32_t *qin, const uint32_t *count_in, uint32_t stack_depth, hipStream_t s) {
    if (rp.n_lanes == 0) return;
    const uint32_t sw = stage_words_for(scene_bytes, stack_bytes(stack_depth)), block = sw ? kBlock : unstaged_block(rp);
    const uint32_t lds = sw * 16 + stack_bytes(stack_depth, block), grid = nseg(rp.n_lanes) * (kSeg / block);
    check_lds(lds);
#define DTOF_LAUNCH_TRACE(L, M, B) hipLaunchKernelGGL((k_trace<L, M, B>), dim3(grid), dim3(B), lds, s, scene, scene_bytes, sw, q, qin, count_in, rp.n_lanes)
    if (sw) { if (rp.has_tris) DTOF_LAUNCH_TRACE(true, true, kBlock); else DTOF_LAUNCH_TRACE(true, false, kBlock); }
    else if (block == 64)  { if (rp.has_tris) DTOF_LAUNCH_TRACE(false, true, 64); else DTOF_LAUNCH_TRACE(false, false, 64); }
    else if (block == 128) { if (rp.has_tris) DTOF_LAUNCH_TRACE(false, true, 128); else DTOF_LAUNCH_TRACE(false, false, 128); }
    else                   { if (rp.has_tris) DTOF_LAUNCH_TRACE(false, true, kBlock); else DTOF_LAUNCH_TRACE(false, false, kBlock); }
#undef DTOF_LAUNCH_TRACE
}
void launch_shade(const uint8_t *scene, uint32_t scene_bytes, const RenderParams &rp, const Queues &q,
                  const uint32_t *qin, const uint32_t *count_in, uint32_t *qout,
                  uint32_t *alive_out, uint32_t *shadow_out, uint32_t depth, bool fused, bool trace_next,
                  uint32_t stack_depth, hipStream_t s, bool first, LaneDebug *dbg) {
    if (rp.n_lanes == 0) return;
    const uint32_t shade_stack = fused ? stack_bytes(stack_depth, kShadeBlock) : 0;
    uint32_t sw = stage_words_for(scene_bytes, shade_stack), grid = nseg(rp.n_lanes), lds = sw * 16 + shade_stack;
    check_lds(lds);
    uint32_t tn = trace_next ? 1u : 0u;
#define DTOF_LAUNCH_SHADE(L, F, A, K) do { if (rp.has_tris) DTOF_LAUNCH_SHADE_M(L, F, A, K, true, false); else DTOF_LAUNCH_SHADE_M(L, F, A, K, false, false); } while (0)
#define DTOF_LAUNCH_SHADE_M(L, F, A, K, M, S) hipLaunchKernelGGL((k_shade<L, F, A, K, M, S>), dim3(grid), dim3(kShadeBlock), lds, s, scene, scene_bytes, sw, rp, q, qin, \
                                                         count_in, qout, alive_out, shadow_out, depth, tn, dbg)
#define DTOF_SHADE_AK(L, F) do { if (rp.has_spec) { if (rp.n_offsets == 1) DTOF_LAUNCH_SHADE_M(L, F, true, 1, true, true); else DTOF_LAUNCH_SHADE_M(L, F, true, kMaxOffsets, true, true); } \
                                 else if (rp.has_area) { if (rp.n_offsets == 1) DTOF_LAUNCH_SHADE(L, F, true, 1); else DTOF_LAUNCH_SHADE(L, F, true, kMaxOffsets); } \
                                 else { if (rp.n_offsets == 1) DTOF_LAUNCH_SHADE(L, F, false, 1); else DTOF_LAUNCH_SHADE(L, F, false, kMaxOffsets); } } while (0)
    if (first && !fused) throw std::runtime_error("the first-bounce kernel exists in the fused pipeline only");
    if (sw) { if (first) DTOF_SHADE_AK(true, 2); else if (fused) DTOF_SHADE_AK(true, 1); else DTOF_SHADE_AK(true, 0); }
    else    { if (first) DTOF_SHADE_AK(false, 2); else if (fused) DTOF_SHADE_AK(false, 1); else DTOF_SHADE_AK(false, 0); }
#undef DTOF_SHADE_AK
#undef DTOF_LAUNCH_SHADE
#undef DTOF_LAUNCH_SHADE_M
}
void launch_shadow(const uint8_t *scene, uint32_t scene_bytes, const RenderParams &rp, const Queues &q,
                   const uint32_t *count_in, uint32_t stack_depth, hipStream_t s) {
    if (rp.n_lanes == 0) return;
    const uint32_t sw = stage_words_for(scene_bytes, stack_bytes(stack_depth)), block = sw ? kBlock : unstaged_block(rp);
    const uint32_t lds = sw * 16 + stack_bytes(stack_depth, block), grid = nseg(rp.n_lanes) * (kSeg / block);
    check_lds(lds);
#define DTOF_LAUNCH_SHADOW(L, M, B) hipLaunchKernelGGL((k_shadow<L, M, B>), dim3(grid), dim3(B), lds, s, scene, scene_bytes, sw, rp, q, count_in)
    if (sw) { if (rp.has_tris) DTOF_LAUNCH_SHADOW(true, true, kBlock); else DTOF_LAUNCH_SHADOW(true, false, kBlock); }
    else if (block == 64)  { if (rp.has_tris) DTOF_LAUNCH_SHADOW(false, true, 64); else DTOF_LAUNCH_SHADOW(false, false, 64); }
    else if (block == 128) { if (rp.has_tris) DTOF_LAUNCH_SHADOW(false, true, 128); else DTOF_LAUNCH_SHADOW(false, false, 128); }
    else                   { if (rp.has_tris) DTOF_LAUNCH_SHADOW(false, true, kBlock); else DTOF_LAUNCH_SHADOW(false, false, kBlock); }
#undef DTOF_LAUNCH_SHADOW
}
void launch_velocity(const uint8_t *scene, uint32_t scene_bytes, const RenderParams &rp, const Queues &q, uint32_t stack_depth, hipStream_t s) {
    if (rp.n_lanes == 0) return;
    uint32_t sw = stage_words_for(scene_bytes, stack_bytes(stack_depth)), lds = sw * 16 + stack_bytes(stack_depth);
    check_lds(lds);
    if (sw) hipLaunchKernelGGL(k_velocity<true>, dim3(nblk(rp.n_lanes)), dim3(kBlock), lds, s, scene, scene_bytes, sw, rp, q);
    else hipLaunchKernelGGL(k_velocity<false>, dim3(nblk(rp.n_lanes)), dim3(kBlock), lds, s, scene, scene_bytes, sw, rp, q);
}
void launch_splat(const RenderParams &rp, const Queues &q, float *film, int32_t film_w, int32_t film_h, hipStream_t s) {
    if (rp.n_lanes == 0) return;
    size_t stride = (size_t) film_w * film_h * 4;
    bool fast = rp.filter == FILTER_TENT && rp.filter_radius <= 1.f && rp.filter_radius > .5f && rp.spp_log2 != 0xffffffffu && rp.spp >= 2;
    if (fast) {
        uint32_t seg = rp.spp < 64 ? rp.spp : 64;
        hipLaunchKernelGGL(k_splat_tent3, dim3(nblk(rp.n_lanes)), dim3(kBlock), 0, s, rp, q, film, stride, seg);
    } else {
        hipLaunchKernelGGL(k_splat_generic, dim3(nblk(rp.n_lanes)), dim3(kBlock), 0, s, rp, q, film, stride);
    }
}
void launch_develop(const float *film, float *rgb, int64_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_develop, dim3((uint32_t) ((n + 255) / 256)), dim3(256), 0, s, film, rgb, n);
}
void launch_lane_dump(const RenderParams &rp, const Queues &q, LaneDebug *out, hipStream_t s) {
    if (rp.n_lanes == 0) return;
    hipLaunchKernelGGL(k_lane_dump, dim3(nblk(rp.n_lanes)), dim3(kBlock), 0, s, rp, q, out);
}
void launch_lane_dump_rays(const RenderParams &rp, const Queues &q, LaneDebug *out, hipStream_t s) {
    if (rp.n_lanes == 0) return;
    hipLaunchKernelGGL(k_lane_dump_rays, dim3(nblk(rp.n_lanes)), dim3(kBlock), 0, s, rp, q, out);
}

// ---------------------------------------------------------------------------- sampler / waveform KAT kernels
__global__ void k_sampler_seed(RenderParams rp, SamplerState st) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= st.n) return;
    Rng a = seed_stream(rp.seed_value, i), b = seed_stream(rp.seed_value + 1, i / rp.tcn), c = seed_stream(rp.seed_value + 2, i / rp.pcn);
    st.rng[i] = make_uint2((uint32_t) a.state, (uint32_t) (a.state >> 32));
    st.rng_time[i] = make_uint2((uint32_t) b.state, (uint32_t) (b.state >> 32));
    st.rng_path[i] = make_uint2((uint32_t) c.state, (uint32_t) (c.state >> 32));
    uint32_t ps, tmp; tea32(rp.base_seed, rp.spp * (i / rp.spp) + rp.seed, ps, tmp);   // compute_per_sequence_seed, sampler.cpp:85-92
    st.perm_seed[i] = ps; st.dim[i] = 0;
}
DTOF_D Rng load_rng(const uint2 *arr, uint32_t i, uint64_t inc) { Rng r; uint2 v = arr[i]; r.state = (uint64_t) v.x | ((uint64_t) v.y << 32); r.inc = inc; return r; }
DTOF_D void store_rng(uint2 *arr, uint32_t i, const Rng &r) { arr[i] = make_uint2((uint32_t) r.state, (uint32_t) (r.state >> 32)); }

__global__ void k_sampler_next_correlate(RenderParams rp, SamplerState st, const uint8_t *correlate, int correlate_all, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= st.n) return;
    Rng m = load_rng(st.rng, i, stream_inc(rp.seed_value, i)), p = load_rng(st.rng_path, i, stream_inc(rp.seed_value + 2, i / rp.pcn));
    out[i] = next_correlate(m, p, correlate ? correlate[i] != 0 : correlate_all != 0);
    store_rng(st.rng, i, m); store_rng(st.rng_path, i, p);
}
__global__ void k_sampler_next_1d(RenderParams rp, SamplerState st, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= st.n) return;
    Rng m = load_rng(st.rng, i, stream_inc(rp.seed_value, i));
    out[i] = next_f32(m);
    store_rng(st.rng, i, m);
}
__global__ void k_sampler_next_time(RenderParams rp, SamplerState st, uint32_t sample_index_base, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= st.n) return;
    Rng m = load_rng(st.rng, i, stream_inc(rp.seed_value, i)), t = load_rng(st.rng_time, i, stream_inc(rp.seed_value + 1, i / rp.tcn));
    uint32_t si = sample_index_base + (rp.spp > 1 ? i % rp.spp : 0), dim = st.dim[i];
    out[i] = next_time(rp, m, t, si, st.perm_seed[i], dim);
    st.dim[i] = dim;
    store_rng(st.rng, i, m); store_rng(st.rng_time, i, t);
}
// mode 0: eval_modulation_weight(t, len) ; 1: waveform(t) ; 2: waveform_low_pass(t)
__global__ void k_waveform_eval(RenderParams rp, const float *t, const float *len, float *out, int mode, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = mode == 0 ? modulation_weight(rp, rp.phase[0], t[i], len[i]) : mode == 1 ? waveform(t[i], rp.wave_type) : waveform_low_pass(t[i], rp.wave_type);
}
void launch_sampler_seed(const RenderParams &rp, const SamplerState &st, hipStream_t s) {
    if (st.n) hipLaunchKernelGGL(k_sampler_seed, dim3(nblk(st.n)), dim3(kBlock), 0, s, rp, st);
}
void launch_sampler_next_correlate(const RenderParams &rp, const SamplerState &st, const uint8_t *correlate, int correlate_all, float *out, hipStream_t s) {
    if (st.n) hipLaunchKernelGGL(k_sampler_next_correlate, dim3(nblk(st.n)), dim3(kBlock), 0, s, rp, st, correlate, correlate_all, out);
}
void launch_sampler_next_1d(const RenderParams &rp, const SamplerState &st, float *out, hipStream_t s) {
    if (st.n) hipLaunchKernelGGL(k_sampler_next_1d, dim3(nblk(st.n)), dim3(kBlock), 0, s, rp, st, out);
}
void launch_sampler_next_time(const RenderParams &rp, const SamplerState &st, uint32_t sample_index_base, float *out, hipStream_t s) {
    if (st.n) hipLaunchKernelGGL(k_sampler_next_time, dim3(nblk(st.n)), dim3(kBlock), 0, s, rp, st, sample_index_base, out);
}
void launch_waveform_eval(const RenderParams &rp, const float *t, const float *len, float *out, int mode, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_waveform_eval, dim3(nblk(n)), dim3(kBlock), 0, s, rp, t, len, out, mode, n);
}

}  // namespace dtof
