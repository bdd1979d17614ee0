#pragma once
// dtof_traverse.h -- device side of the scene: blob view, LDS staging, primitive tests (rectangle, triangle, sphere), motion-blur
// instance matrices and the TLAS / BLAS traversal (closest hit and occlusion).  Included by dtof_kernels.hip only.
#include "dtof_kernels.h"
#include "dtof_scene.h"
#include "dtof_math.h"

#ifndef DTOF_D
#define DTOF_D __device__ __forceinline__
#endif

namespace dtof {

// ---------------------------------------------------------------------------- scene view
struct SceneView {
    const DNode *nodes; const DNode16 *nodes16; const DObject *objects; const DGroup *groups; const DShape *shapes;
    const DTri *tris; const DTriIsect *isect; const DTriShade *shading; const DEmitter *emitters; const uint8_t *base;
    uint32_t n_nodes, n_emitters;
    // Fused shade kernels, scenes with ONE instance object: the world -> object matrix of that instance at the lane's ray time is
    // computed once per path vertex (instance_memo_fill) and kept in a per-thread LDS column; the closest-hit and occlusion queries of
    // the vertex and its surface interaction read it back instead of re-deriving it (a ray's time does not change along a path,
    // dopplertofpath.cpp:93,240-244).  memo_obj = 0xffffffff: no memo.
    uint32_t memo_obj; float *memo;
    bool memo_m;   // the column also holds the instance matrix itself (words 12 .. 23; the flat scenes' kernels, whose LDS has the room): compute_surface reads both back
};
// Resident scene stage (k_shade<..., RESW>): the TLAS nodes live in LDS as FOUR PLANES of 16-byte pieces -- piece k of node i at
// uint4 index k * kResNodes + i -- so that the 16 lanes of a ds_read_b128 lane group, which read the same piece of 16 different nodes,
// spread over all 64 banks (node-major 64-byte records would put piece k of every node on the same 16 banks: 4-way conflicts on
// average); the plane stride is a compile-time constant, so the four reads of a node step are one address plus immediate offsets.
constexpr uint32_t kResNodes = 1024;
// ... and in the kernels of SEVERAL films their child references are re-encoded to 16 bits while they are copied (at most 1 024 nodes, and a TLAS of that size has at
// most 1 025 objects): leaf flag = bit 15, "no child" = 0xffff.  The traversal stack of those kernels then holds 16-BIT entries -- half the LDS of the 32-bit columns
// (Domino: 24 instead of 48 KiB at 16 waves), which pays for the film-state AND the path-state columns of k_shade (round 5: C5 164.9 -> 156.4 ms).  The one-film kernels
// keep 32-bit entries: their columns fit as they are, and the 16-bit form costs them 0.8 % (C4 33.74 -> 34.02 ms; profiles/r05_k4_film_state.txt).
constexpr uint32_t kLeafFlag16 = 0x8000u, kNoChild16 = 0xffffu, kDone16 = 0x7fffu;
DTOF_D uint32_t encode_child16(uint32_t w) { return w == kNoChild ? kNoChild16 : ((w & 0x7fffu) | ((w >> 16) & 0x8000u)); }
DTOF_D SceneView make_view(const uint8_t *base) {
    const BlobHeader *h = (const BlobHeader *) base;
    SceneView v;
    v.nodes = (const DNode *) (base + h->off_nodes);
    v.nodes16 = (const DNode16 *) (base + h->off_nodes16);   // (meaningful where off_nodes16 != 0: the launch picks the kernels that read it)
    v.objects = (const DObject *) (base + h->off_objects);
    v.groups = (const DGroup *) (base + h->off_groups);
    v.shapes = (const DShape *) (base + h->off_shapes);
    v.tris = (const DTri *) (base + h->off_tris);
    v.isect = (const DTriIsect *) (base + h->off_isect);
    v.shading = (const DTriShade *) (base + h->off_shading);
    v.emitters = (const DEmitter *) (base + h->off_emitters);
    v.base = base;
    v.n_nodes = h->n_nodes; v.n_emitters = h->n_emitters;
    v.memo_obj = 0xffffffffu; v.memo = nullptr; v.memo_m = false;
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

// Traversal statistics (development builds only: make STATS=1 -> libdtof_stats.so, read by tools/traversal_stats.py).
// [0] rays  [1] node steps (lane)  [2] node iterations (wave)  [3] leaf visits (lane)  [4] leaf rounds (wave)  [5] mesh loops entered (lane)
// [6] triangle tests (lane)  [7] BLAS node steps (lane)  [8] instance transforms (lane)  [9] instance transforms (wave)  [10] mesh loops entered (wave)
// [11] triangle tests (wave)  [12] rectangle tests (lane)  [13] rectangle tests (wave)
// [14] POISON HITS: in these builds k_shade gives the path state of every lane that has no path (beyond the end of its segment) a recognisable pattern (kPoisonState) and
//      counts it wherever path state leaves the registers -- queue stores, the LDS columns, the inline iterations' hand-over.  Must stay 0: the K = 4 incident of round 3
//      (profiles/r03_k4_uninitialised.txt, profiles/r05_k4_root_cause.txt) was a film that depended on the INITIAL value of those registers.
#ifdef DTOF_TRAVERSAL_STATS
// every translation unit with kernels counts into its own copy (no relocatable device code); each registers a reader, read_traversal_stats sums and resets them all
static __device__ unsigned long long g_trav_stats[16];
static bool read_tu_traversal_stats(unsigned long long *acc8) {
    unsigned long long v[16], zero[16] = { 0 };
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_trav_stats), 128) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_trav_stats), zero, 128) != hipSuccess) return false;
    for (int i = 0; i < 16; ++i) acc8[i] += v[i];
    return true;
}
static const int g_trav_stats_registered = (register_traversal_stats_reader(read_tu_traversal_stats), 0);
#define DTOF_STAT(i) atomicAdd(&g_trav_stats[i], 1ull)
#define DTOF_STAT_WAVE(i) do { if (__lane_id() == (uint32_t) __ffsll((long long) __ballot(1)) - 1u) atomicAdd(&g_trav_stats[i], 1ull); } while (0)
#else
#define DTOF_STAT(i) ((void) 0)
#define DTOF_STAT_WAVE(i) ((void) 0)
#endif

// ---------------------------------------------------------------------------- primitives
// Rectangle::ray_intersect_preliminary_impl, src/shapes/rectangle.cpp:201-224
DTOF_D bool rect_hit(const DShape &sh, V3 o, V3 d, float maxt, float &t, float &u, float &v) {
    V3 lo = xf_point(sh.to_object, o), ld = xf_vector(sh.to_object, d);
    t = -lo.z / ld.z;
    u = fmaf(ld.x, t, lo.x); v = fmaf(ld.y, t, lo.y);
    return t >= 0.f && t <= maxt && fabsf(u) <= 1.f && fabsf(v) <= 1.f;
}
// Disk::ray_intersect_preliminary_impl, src/shapes/disk.cpp:216-232: the rectangle's plane test with a circular bound
DTOF_D bool disk_hit(const DShape &sh, V3 o, V3 d, float maxt, float &t, float &u, float &v) {
    V3 lo = xf_point(sh.to_object, o), ld = xf_vector(sh.to_object, d);
    t = -lo.z / ld.z;
    u = fmaf(ld.x, t, lo.x); v = fmaf(ld.y, t, lo.y);
    return t >= 0.f && t <= maxt && u * u + v * v <= 1.f;
}
// Moeller-Trumbore as in Embree 3's triangle intersector (tnear < t <= tfar; u,v weight vertices 1,2).  The 48-byte record (DTriIsect: first vertex, edges and
// geometric normal, precomputed on the host with these operations) is fetched with three 16-byte loads issued together.
DTOF_D bool tri_hit(const DTriIsect &tr, V3 o, V3 d, float maxt, float &t, float &u, float &v) {
    const uint4 *tp = (const uint4 *) &tr;
    const uint4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
    const V3 p0 = mk(u2f(q0.x), u2f(q0.y), u2f(q0.z)), e1 = mk(u2f(q1.x), u2f(q1.y), u2f(q1.z)), e2 = mk(u2f(q2.x), u2f(q2.y), u2f(q2.z));
    const V3 ng = mk(u2f(q0.w), u2f(q1.w), u2f(q2.w));
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
// Cylinder::ray_intersect_preliminary_impl / ray_test_impl (src/shapes/cylinder.cpp:300-391): the unit cylinder in object space, float64 on the
// llvm back end (the ray is transformed in float32, then widened)
DTOF_D bool cylinder_hit(const DShape &sh, V3 o, V3 d, float maxt_f, float &t_out) {
    const V3 lo = xf_point(sh.to_object, o), ld = xf_vector(sh.to_object, d);
    const double ox = lo.x, oy = lo.y, oz = lo.z, dx = ld.x, dy = ld.y, dz = ld.z, maxt = maxt_f;
    const double A = dx * dx + dy * dy, B = 2.0 * (dx * ox + dy * oy), C = ox * ox + oy * oy - 1.0;
    double near_t, far_t;
    const bool found = solve_quadratic_d(A, B, C, near_t, far_t);
    const bool out_bounds = !(near_t <= maxt && far_t >= 0.0), in_bounds = near_t < 0.0 && far_t > maxt;
    const double z_near = oz + dz * near_t, z_far = oz + dz * far_t;
    const bool near_ok = z_near >= 0.0 && z_near <= 1.0 && near_t >= 0.0, far_ok = z_far >= 0.0 && z_far <= 1.0 && far_t <= maxt;
    if (!(found && !out_bounds && !in_bounds && (near_ok || far_ok))) return false;
    t_out = near_ok ? (float) near_t : (float) far_t;
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

// instance memo (see SceneView): 12 words per thread, word k of thread t at memo[k * kMemoStride + t]
constexpr uint32_t kMemoStride = 64, kMemoWords = 12;
DTOF_D void instance_memo_fill(const SceneView &sv, float time, float *m, float *inv) {
    instance_matrix(sv.objects[sv.memo_obj], time, m);
    affine_inverse(m, inv);
#pragma unroll
    for (uint32_t k = 0; k < kMemoWords; ++k) sv.memo[k * kMemoStride] = inv[k];
    if (sv.memo_m) {
#pragma unroll
        for (uint32_t k = 0; k < kMemoWords; ++k) sv.memo[(kMemoWords + k) * kMemoStride] = m[k];
    }
}
DTOF_D void instance_memo_load_matrix(const SceneView &sv, float *m) {   // only where sv.memo_m
#pragma unroll
    for (uint32_t k = 0; k < kMemoWords; ++k) m[k] = sv.memo[(kMemoWords + k) * kMemoStride];
}
DTOF_D void instance_memo_load(const SceneView &sv, float *inv) {
#pragma unroll
    for (uint32_t k = 0; k < kMemoWords; ++k) inv[k] = sv.memo[k * kMemoStride];
}

// Slab test of a padded box.  A plane at coordinate b is crossed at t = b * id - o * id: ONE multiply-add per plane with `noid` = -(o * id) computed once per ray (the
// difference-then-product form costs two instructions per plane, 12 more per node step of an issue-bound traversal).  The test only culls -- which primitive is hit, and
// where, is decided by the primitive tests -- so it has to be conservative, not equal to anything: its rounding error is 2^-24 (|o| + |t / id|) |id| per plane against
// the padding of (1e-5 max(|b|, extent) + 1e-6) |id| the host puts around every box (scene_build.cpp: Box::pad), the same order as the other form's 2^-23 |b - o| |id|
// since |o| <= |b - o| + |b|.  NaNs fall out of the min / max chain (fminf / fmaxf return the other operand); a box whose test has no number left is missed.
struct SlabRay { V3 id, noid; };
DTOF_D SlabRay slab_ray(V3 o, V3 d) {
    // direction reciprocal for the slab test only (exact zero components are nudged); v_rcp_f32 (1 ulp) is enough here, see the padding above
    SlabRay r;
    r.id = mk(__builtin_amdgcn_rcpf(d.x == 0.f ? 1e-30f : d.x), __builtin_amdgcn_rcpf(d.y == 0.f ? 1e-30f : d.y), __builtin_amdgcn_rcpf(d.z == 0.f ? 1e-30f : d.z));
    r.noid = mk(-(o.x * r.id.x), -(o.y * r.id.y), -(o.z * r.id.z));
    return r;
}
DTOF_D bool box_hit(const float *bmin, const float *bmax, const SlabRay &r, float tbest, float &tn) {
    const float tx0 = fmaf(bmin[0], r.id.x, r.noid.x), tx1 = fmaf(bmax[0], r.id.x, r.noid.x);
    const float ty0 = fmaf(bmin[1], r.id.y, r.noid.y), ty1 = fmaf(bmax[1], r.id.y, r.noid.y);
    const float tz0 = fmaf(bmin[2], r.id.z, r.noid.z), tz1 = fmaf(bmax[2], r.id.z, r.noid.z);
    tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.f));
    const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tbest));
    return tn <= tf;
}
// Traversal stack entry `sp` of this thread: a per-thread LDS column (32-bit, or 16-bit in the resident kernels of several films) -- and, in the eight-waves-per-SIMD
// ray kernels of large meshes (k_trace / k_shadow<..., W8>), only its first LDSN entries: deeper ones overflow into a private array `ovf` (scratch memory; a BLAS of half a
// million triangles is 20-odd levels deep, a traversal rarely holds more than a dozen entries), so that 32 one-wave blocks fit a CU's LDS instead of 23.
template <bool S16, uint32_t LDSN>
DTOF_D void stack_put(uint32_t *stack, uint32_t *ovf, int sp, uint32_t stride, uint32_t x) {
    // (LDSN: the empty asm keeps the two stores in their branches -- merged, they become ONE flat_store through a select of the scratch and the LDS address, and a flat
    //  access goes through the texture-address unit these kernels saturate; round 5, read in the ISA of k_trace<false, true, 64, true>)
    if (LDSN != 0 && (uint32_t) sp >= LDSN) { ovf[(uint32_t) sp - LDSN] = x; return; }
    if (S16) ((uint16_t *) stack)[sp * stride] = (uint16_t) x;
    else stack[sp * stride] = x;
    if (LDSN != 0) asm volatile("" : : "v"(x));   // BEHIND the store: the last instructions of the two branches differ, the stores are not sunk into one
}
template <bool S16, uint32_t LDSN>
DTOF_D uint32_t stack_get(const uint32_t *stack, const uint32_t *ovf, int sp, uint32_t stride) {
    if (LDSN != 0 && (uint32_t) sp >= LDSN) return ovf[(uint32_t) sp - LDSN];
    uint32_t x = S16 ? (uint32_t) ((const uint16_t *) stack)[sp * stride] : stack[sp * stride];
    if (LDSN != 0) asm volatile("" : "+v"(x));   // (see stack_put: a ds_read in its own branch, not a flat_load)
    return x;
}
// One traversal step at inner node `cur` = four 16-byte loads issued together (no load depends on a field of the node): continue with the nearest child that is
// hit, push the other, pop when nothing is hit.  STRIDE: the stride of the per-thread stack columns when the kernel knows its block size (a shift instead of v_mul_lo_u32).
DTOF_D float half_lo(uint32_t w) { return (float) __builtin_bit_cast(_Float16, (uint16_t) (w & 0xffffu)); }
DTOF_D float half_hi(uint32_t w) { return (float) __builtin_bit_cast(_Float16, (uint16_t) (w >> 16)); }
template <bool SOA = false, uint32_t STRIDE = 0, bool S16 = false, uint32_t LDSN = 0, bool H16 = false>   // S16: 16-bit child references and stack entries (the resident kernels of several films); LDSN: LDS entries before the overflow; H16: `nodes` points at DNode16 records (half-float boxes: two loads)
DTOF_D uint32_t node_step(const BvhNode *nodes, uint32_t cur, const SlabRay &r, float tbest, uint32_t *stack, int &sp, int sp_floor, uint32_t stride_rt, uint32_t done, uint32_t *ovf = nullptr) {
    const uint32_t stride = STRIDE ? STRIDE : stride_rt;
    if (H16) {
        // (SOA: the resident stage's LDS copy -- TWO planes of 2 * kResNodes pieces, piece k of node i at k * 2 * kResNodes + i: 2 048 half-float nodes in the 64 KiB of 1 024 float ones)
        const uint4 *np = SOA ? (const uint4 *) nodes + cur : (const uint4 *) ((const DNode16 *) nodes + cur);
        const uint4 a = np[0], b = np[SOA ? 2 * kResNodes : 1];
        const float lmin[3] = { half_lo(a.x), half_hi(a.x), half_lo(a.y) }, lmax[3] = { half_hi(a.y), half_lo(a.z), half_hi(a.z) };
        const float rmin[3] = { half_lo(b.x), half_hi(b.x), half_lo(b.y) }, rmax[3] = { half_hi(b.y), half_lo(b.z), half_hi(b.z) };
        const uint32_t left = a.w, right = b.w;
        float tl, tr;
        const bool hl = box_hit(lmin, lmax, r, tbest, tl);
        const bool hr = (int) box_hit(rmin, rmax, r, tbest, tr) & (int) (right != (S16 ? kNoChild16 : kNoChild));
        if (hl && hr) {
            const bool left_first = tl <= tr;
            stack_put<S16, LDSN>(stack, ovf, sp, stride, left_first ? right : left);
            ++sp;
            return left_first ? left : right;
        }
        if (hl) return left;
        if (hr) return right;
        if (sp == sp_floor) return done;
        --sp; return stack_get<S16, LDSN>(stack, ovf, sp, stride);
    }
    const uint4 *np = SOA ? (const uint4 *) nodes + cur : (const uint4 *) (nodes + cur);
    const uint4 a = np[0], b = np[SOA ? kResNodes : 1], c = np[SOA ? 2 * kResNodes : 2], d = np[SOA ? 3 * kResNodes : 3];
    const float lmin[3] = { u2f(a.x), u2f(a.y), u2f(a.z) }, lmax[3] = { u2f(b.x), u2f(b.y), u2f(b.z) };
    const float rmin[3] = { u2f(c.x), u2f(c.y), u2f(c.z) }, rmax[3] = { u2f(d.x), u2f(d.y), u2f(d.z) };
    const uint32_t left = a.w, right = b.w;
    float tl, tr;
    const bool hl = box_hit(lmin, lmax, r, tbest, tl);
    const bool hr = (int) box_hit(rmin, rmax, r, tbest, tr) & (int) (right != (S16 ? kNoChild16 : kNoChild));
    if (hl && hr) {   // (S16: the caller hands in the halfword address of this thread's column)
        const bool left_first = tl <= tr;
        stack_put<S16, LDSN>(stack, ovf, sp, stride, left_first ? right : left);
        ++sp;
        return left_first ? left : right;
    }
    if (hl) return left;
    if (hr) return right;
    if (sp == sp_floor) return done;
    --sp; return stack_get<S16, LDSN>(stack, ovf, sp, stride);
}

// Closest hit (ANY=false) or occlusion (ANY=true) of one top-level object.  Candidates are every
// primitive hit with t <= maxt; the winner is the smallest t, ties going to the lowest
// (object, shape, face) -- the rule the oracle uses, independent of traversal order.
// `stack + sp * stride` onwards is free for the BLAS traversal of a mesh.
// MESH: 0 = rectangles only, 1 = every shape, 2 = rectangles and triangle meshes (scenes without analytic shapes: no float64 sphere / cylinder code, fewer registers).
template <bool ANY, int MESH, bool MEMO = false, uint32_t STRIDE = 0, bool NOBLAS = false, uint32_t LDSN = 0, bool H16 = false>   // H16: the BLAS are walked through sv.nodes16; NOBLAS: the resident kernels (no mesh behind a BLAS; their stack columns are 16-bit)
DTOF_D bool intersect_object(const SceneView &sv, uint32_t oi, V3 o, V3 d, float time, float maxt, Hit &best,
                             uint32_t *stack, int sp, uint32_t stride, uint32_t *ovf = nullptr) {
    const DObject &ob = sv.objects[oi];
    uint32_t first = ob.index, count = 1;
    V3 lo = o, ld = d;
    if (ob.kind == OBJ_INSTANCE) {
        float m[12], inv[12];
        if (MEMO && oi == sv.memo_obj) instance_memo_load(sv, inv);
        else {
            DTOF_STAT(8); DTOF_STAT_WAVE(9);
            instance_matrix(ob, time, m);
            affine_inverse(m, inv);
        }
        lo = xf_point(inv, o); ld = xf_vector(inv, d);
        const DGroup &g = sv.groups[ob.index];
        first = g.first_shape; count = g.n_shapes;
    }
    bool found = false;
    for (uint32_t k = 0; k < count; ++k) {
        const DShape &sh = sv.shapes[first + k];
        float t, u, v;
        if (sh.kind == SHAPE_RECT) {
            DTOF_STAT(12); DTOF_STAT_WAVE(13);
            if (rect_hit(sh, lo, ld, maxt, t, u, v)) {
                if (ANY) return true;
                if (t < best.t || (t == best.t && !found && best.obj != 0xffffffffu && oi < best.obj)) {
                    best.t = t; best.u = u; best.v = v; best.obj = oi; best.shape = k; best.prim = 0; found = true;
                }
            }
            continue;
        }
        if (!MESH) continue;   // instantiations for rectangle-only scenes carry no triangle / sphere code at all
        if (MESH == 1 && sh.kind == SHAPE_DISK) {
            if (disk_hit(sh, lo, ld, maxt, t, u, v)) {
                if (ANY) return true;
                if (t < best.t || (t == best.t && !found && best.obj != 0xffffffffu && oi < best.obj)) {
                    best.t = t; best.u = u; best.v = v; best.obj = oi; best.shape = k; best.prim = 0; found = true;
                }
            }
            continue;
        }
        if (MESH == 1 && sh.kind == SHAPE_CYLINDER) {
            if (cylinder_hit(sh, lo, ld, maxt, t)) {
                if (ANY) return true;
                if (t < best.t || (t == best.t && !found && best.obj != 0xffffffffu && oi < best.obj)) {
                    best.t = t; best.u = 0.f; best.v = 0.f; best.obj = oi; best.shape = k; best.prim = 0; found = true;
                }
            }
            continue;
        }
        if (MESH == 1 && sh.kind == SHAPE_SPHERE) {
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
        const SlabRay lr = slab_ray(lo, ld);
        float t_entry;
        if (!box_hit(sh.bmin, sh.bmax, lr, ANY ? maxt : best.t, t_entry)) continue;
        // `face` of the best hit so far IF it lies on this very mesh (ties between two of its triangles go to the lower face)
        uint32_t best_face = 0xffffffffu;
        DTOF_STAT(5); DTOF_STAT_WAVE(10);
        auto test = [&](uint32_t f) -> bool {
            DTOF_STAT(6); DTOF_STAT_WAVE(11);
            if (!tri_hit(sv.isect[sh.first_tri + f], lo, ld, maxt, t, u, v)) return false;
            if (ANY) return true;
            const uint32_t face = sv.tris[sh.first_tri + f].face;   // the triangle's index in the mesh's own order (the BLAS permutes them): decides ties
            bool take = t < best.t;
            if (t == best.t) take = best_face != 0xffffffffu ? face < best_face : (!found && best.obj != 0xffffffffu && oi < best.obj);
            if (take) { best.t = t; best.u = u; best.v = v; best.obj = oi; best.shape = k; best.prim = f; best_face = face; found = true; }
            return false;
        };
        if (sh.blas_root == kNoChild) {
            for (uint32_t f = 0; f < sh.n_tris; ++f) if (test(f)) return true;
            continue;
        }
        if (NOBLAS) continue;
        // BLAS: same node format and while-while shape as the TLAS loop below
        constexpr uint32_t kDone = 0x7fffffffu;
        uint32_t cur = sh.blas_root; int bsp = sp;
        for (;;) {
            while (!(cur & kLeafFlag) && cur != kDone) {
                DTOF_STAT(7); DTOF_STAT_WAVE(15);
                cur = node_step<false, STRIDE, false, LDSN, H16>(H16 ? (const BvhNode *) sv.nodes16 : sv.nodes, cur, lr, ANY ? maxt : best.t, stack, bsp, sp, stride, kDone, ovf);
            }
            if (cur == kDone) break;
            uint32_t f0 = (cur & ~kLeafFlag) >> kBlasLeafBits, fn = (cur & ((1u << kBlasLeafBits) - 1u)) + 1u;
            for (uint32_t f = f0; f < f0 + fn; ++f) if (test(f)) return true;
            if (bsp == sp) break;
            --bsp; cur = stack_get<false, LDSN>(stack, ovf, bsp, stride);
        }
    }
    return found;
}

// TLAS traversal; `stack` is a per-thread LDS column (stride blockDim.x).
template <bool ANY, int MESH, bool MEMO = false, bool SOA = false, uint32_t STRIDE = 0, bool S16 = false, uint32_t LDSN = 0, bool TL = false, bool H16 = false, bool DEFER = false>   // DEFER: objects whose TLAS leaf carries kLeafBlas are not entered -- their indices go to `cand` (up to four; a fifth is entered on the spot) for trace_deferred in a second, dense launch; H16: nodes read from sv.nodes16 (the TLAS too, unless TL); TL: `tlas` = a copy of the TLAS nodes in LDS (the BLAS stay where sv.nodes points); SOA: the TLAS nodes are the LDS planes of the resident stage (binary nodes only); STRIDE: the block size, if the kernel knows it; S16: 16-bit references / stack; LDSN: see stack_put
DTOF_D bool trace_scene(const SceneView &sv, uint32_t *stack, V3 o, V3 d, float time, float maxt, Hit &best, uint32_t *ovf = nullptr, const BvhNode *tlas = nullptr, uint4 *cand = nullptr) {
    best.t = maxt; best.u = best.v = 0.f; best.obj = 0xffffffffu; best.shape = 0; best.prim = 0;
    if (sv.n_nodes == 0) return false;
    const SlabRay r = slab_ray(o, d);
    // "while-while" traversal: every lane first descends inner nodes until it holds a leaf (or is done), THEN the
    // lanes that hold a leaf run the expensive object intersection together -- the wave does not pay the leaf
    // body once per node step of its slowest lane.
    constexpr uint32_t kDone = S16 ? kDone16 : 0x7fffffffu, kLeaf = S16 ? kLeafFlag16 : kLeafFlag;   // S16: 16-bit child references and stack entries (encode_child16)
    int sp = 0;
    uint32_t cur = 0;
    const uint32_t stride = STRIDE ? STRIDE : blockDim.x;
    DTOF_STAT(0);
    for (;;) {
        while (!(cur & kLeaf) && cur != kDone) {
            DTOF_STAT(1); DTOF_STAT_WAVE(2);
            cur = TL ? node_step<SOA, STRIDE, S16, LDSN>(tlas, cur, r, best.t, stack, sp, 0, stride, kDone, ovf)
                     : node_step<SOA, STRIDE, S16, LDSN, H16>(H16 ? (const BvhNode *) sv.nodes16 : sv.nodes, cur, r, best.t, stack, sp, 0, stride, kDone, ovf);
        }
        if (cur == kDone) break;
        DTOF_STAT(3); DTOF_STAT_WAVE(4);
        bool put_aside = false;
        if (DEFER && (cur & kLeafBlas)) {   // a mesh behind a BLAS: later, with the other rays that reached one
            const uint32_t oi = cur & kLeafObjMask;
            put_aside = true;
            if (cand->x == 0xffffffffu) cand->x = oi; else if (cand->y == 0xffffffffu) cand->y = oi; else if (cand->z == 0xffffffffu) cand->z = oi; else if (cand->w == 0xffffffffu) cand->w = oi;
            else put_aside = false;
        }
        if (!put_aside && intersect_object<ANY, MESH, MEMO, STRIDE, S16, LDSN, H16>(sv, cur & ~kLeaf & (S16 ? 0xffffu : kLeafObjMask), o, d, time, maxt, best, stack, sp, stride, ovf) && ANY) return true;
        if (sp == 0) break;
        --sp; cur = stack_get<S16, LDSN>(stack, ovf, sp, stride);
    }
    return best.obj != 0xffffffffu;
}


// (A cooperative form of the small-mesh triangle loops -- one wave-uniform traversal whose 12-triangle loops are shared out over idle lanes -- was built and
// measured slower in round 3; it is parked as tools/experiments/r03_coop_triangles.patch with its numbers in profiles/r03_coop_triangles_ab.txt.)

// Scene query of a kernel whose call site is wave-uniform; `active` says whether this lane has a ray.
template <bool ANY, int MESH, bool MEMO = false, bool SOA = false, uint32_t STRIDE = 0, uint32_t LDSN = 0, bool TL = false, bool H16 = false, bool DEFER = false>
DTOF_D bool trace_rays(const SceneView &sv, uint32_t *stack, bool active, V3 o, V3 d, float time, float maxt, Hit &best, uint32_t *ovf = nullptr, const BvhNode *tlas = nullptr, uint4 *cand = nullptr) {
    bool r = false;
    if (active) r = trace_scene<ANY, MESH, MEMO, SOA, STRIDE, false, LDSN, TL, H16, DEFER>(sv, stack, o, d, time, maxt, best, ovf, tlas, cand);
    return r;
}
// The second launch of a DEFER pair: the objects a ray's TLAS walk put aside, entered one after the other with the hit the first launch found (closest hit: `best` as it
// stands, its distance culls; occlusion: any hit ends the ray).  Every lane of the launch has at least one such object: the BLAS walks run with full waves.
template <bool ANY, int MESH, uint32_t STRIDE, uint32_t LDSN, bool H16>
DTOF_D bool trace_deferred(const SceneView &sv, uint32_t *stack, uint4 cand, V3 o, V3 d, float time, float maxt, Hit &best, uint32_t *ovf) {
    const uint32_t stride = STRIDE ? STRIDE : blockDim.x;
    const uint32_t c[4] = { cand.x, cand.y, cand.z, cand.w };
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        if (c[i] == 0xffffffffu) break;
        if (intersect_object<ANY, MESH, false, STRIDE, false, LDSN, H16>(sv, c[i], o, d, time, maxt, best, stack, 0, stride, ovf) && ANY) return true;
    }
    return best.obj != 0xffffffffu;
}


// Rectangle-only scenes of a handful of top-level objects (the five walls of the Cornell room of C2): no TLAS walk.  Every lane tests
// every object in index order -- a loop whose trip count and addresses are wave-uniform, so the object and shape records come in by
// SCALAR loads from the blob in global memory (constant address space; the copy staged in LDS serves the per-lane reads of
// compute_surface) and their matrix entries are SGPR operands of the lane arithmetic.  No stack, no divergence: for <= 8 rectangles the
// ~40 instructions per rectangle cost less than the ~55 per node step + ~60 per leaf visit of the binary tree at 0.5 lane utilisation.
// Hits are those of trace_scene bit for bit: same rect_hit arithmetic, smallest t wins, ties go to the lowest object index
// (ascending order + strict <).  Instances take intersect_object (one uniform branch).
typedef const uint8_t __attribute__((address_space(4))) *ConstBytes;
// DTOF_FLAT_PK=1: the rectangle transform of trace_flat as packed multiply-adds (v_pk_fma_f32 with the matrix entries as SGPR-pair operands: two
// multiply-adds in the 4 cycles ONE scalar-operand v_fma_f32 takes, profiles/r03_ubench_valu_rate.txt).  Bit-exact; measured on C2: 3 % fewer VALU
// instructions but the pairs cost registers in a kernel at its cap -- spill loads / stores per wave 157 -> 580, frame 1.53 -> 1.65 ms
// (profiles/r03_flat_packed_ab.txt).  Off.
#ifndef DTOF_FLAT_PK
#define DTOF_FLAT_PK 0
#endif
#ifndef DTOF_FLAT_LDS
#define DTOF_FLAT_LDS 1
#endif
typedef float F2 __attribute__((ext_vector_type(2)));
struct FlatRecord { uint32_t instance; F2 c0, c1, c2, c3; float z0, z1, z2, z3; };   // (x, y) entries of the four columns as pairs, the z row apart
DTOF_D FlatRecord flat_load(const DFlatObject __attribute__((address_space(4))) *f) {
    FlatRecord r; r.instance = f->instance;
    r.c0 = F2{ f->c0[0], f->c0[1] }; r.c1 = F2{ f->c1[0], f->c1[1] }; r.c2 = F2{ f->c2[0], f->c2[1] }; r.c3 = F2{ f->c3[0], f->c3[1] };
    r.z0 = f->c0[2]; r.z1 = f->c1[2]; r.z2 = f->c2[2]; r.z3 = f->c3[2];
    return r;
}
template <bool ANY, bool MEMO>
DTOF_D bool trace_flat(const SceneView &sv, ConstBytes flat_table, uint32_t flat_off, uint32_t n_objects, uint32_t *stack, V3 o, V3 d, float time, float maxt, Hit &best) {
    typedef const DFlatObject __attribute__((address_space(4))) *ConstFlat;
    const ConstFlat table = (ConstFlat) flat_table;
    best.t = maxt; best.u = best.v = 0.f; best.obj = 0xffffffffu; best.shape = 0; best.prim = 0;
    bool occluded = false;
    DTOF_STAT(0);
    auto test = [&](const FlatRecord &rec, uint32_t oi) {   // rect_hit on a plain rectangle's record
        V3 ro = o, rd = d;
        if (rec.instance == 2) {   // (uniform) the one memoised instance: into its space first, as intersect_object does (instance.cpp:101-114)
            float inv[12]; instance_memo_load(sv, inv);
            ro = xf_point(inv, o); rd = xf_vector(inv, d);
        }
        // xf_point / xf_vector with the matrix entries as scalar operands, two multiply-adds per instruction: the x and y rows of the point as one pair,
        // those of the direction as another, the z rows of point AND direction as the third (same entries, different vectors).  Each half is the IEEE
        // operation of the scalar form, in its order; the direction's leading product becomes fma(m, d, -0), which equals m * d for every input.
#if DTOF_FLAT_PK
        const F2 lo_xy = __builtin_elementwise_fma(rec.c2, F2{ ro.z, ro.z }, __builtin_elementwise_fma(rec.c1, F2{ ro.y, ro.y }, __builtin_elementwise_fma(rec.c0, F2{ ro.x, ro.x }, rec.c3)));
        const F2 ld_xy = __builtin_elementwise_fma(rec.c2, F2{ rd.z, rd.z }, __builtin_elementwise_fma(rec.c1, F2{ rd.y, rd.y }, rec.c0 * F2{ rd.x, rd.x }));
        const F2 z = __builtin_elementwise_fma(F2{ rec.z2, rec.z2 }, F2{ ro.z, rd.z }, __builtin_elementwise_fma(F2{ rec.z1, rec.z1 }, F2{ ro.y, rd.y },
                                               __builtin_elementwise_fma(F2{ rec.z0, rec.z0 }, F2{ ro.x, rd.x }, F2{ rec.z3, -0.f })));
#else   // the scalar form (one multiply-add per instruction, matrix entries as SGPR operands: 4 cycles each), kept for A/B timing
        const F2 lo_xy = F2{ fmaf(rec.c2.x, ro.z, fmaf(rec.c1.x, ro.y, fmaf(rec.c0.x, ro.x, rec.c3.x))), fmaf(rec.c2.y, ro.z, fmaf(rec.c1.y, ro.y, fmaf(rec.c0.y, ro.x, rec.c3.y))) };
        const F2 ld_xy = F2{ fmaf(rec.c2.x, rd.z, fmaf(rec.c1.x, rd.y, rec.c0.x * rd.x)), fmaf(rec.c2.y, rd.z, fmaf(rec.c1.y, rd.y, rec.c0.y * rd.x)) };
        const F2 z = F2{ fmaf(rec.z2, ro.z, fmaf(rec.z1, ro.y, fmaf(rec.z0, ro.x, rec.z3))), fmaf(rec.z2, rd.z, fmaf(rec.z1, rd.y, rec.z0 * rd.x)) };
#endif
        const float t = -z.x / z.y;
        const float u = fmaf(ld_xy.x, t, lo_xy.x), v = fmaf(ld_xy.y, t, lo_xy.y);
        // no short-circuit: four compares and three mask ANDs instead of three exec-mask branches per rectangle (the scalar unit is as busy as the vector units here)
        const bool hit = (int) (t >= 0.f) & (int) (t <= maxt) & (int) (fabsf(u) <= 1.f) & (int) (fabsf(v) <= 1.f);
        if (ANY) occluded |= hit;
        else {
            const bool take = (int) hit & (int) (t < best.t);
            best.t = take ? t : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v; best.obj = take ? oi : best.obj;   // best.shape stays 0: the instances, which may set it, come after the rectangles
        }
    };
    // Two record buffers take turns (the loop is unrolled by two), so the next record's scalar load flies while the current one is tested
    // and no register is copied from one iteration to the next.  Instances are noted in a mask and intersected after the rectangles: the
    // tie rule of intersect_object (equal t goes to the lower object index) does not depend on the order of the visits.
    uint32_t instances = 0, oi = 0;
#if DTOF_FLAT_LDS
    // The records come from the scene copy staged in LDS, every lane reading the same address (a broadcast, four ds_read_b128 per record): the matrix entries are then
    // VGPR operands of the multiply-adds, which issue at full rate -- as SGPR operands (scalar loads from the blob) each of the 21 costs two issue slots
    // (profiles/r03_ubench_valu_rate.txt).
    const DFlatObject *lt = (const DFlatObject *) (sv.base + flat_off);
    for (; oi < n_objects; ++oi) {
        const uint4 *rp4 = (const uint4 *) (lt + oi);
        const uint4 r0 = rp4[0], r1 = rp4[1], r2 = rp4[2], r3 = rp4[3];
        FlatRecord a; a.instance = (uint32_t) __builtin_amdgcn_readfirstlane((int) r0.w);
        a.c0 = F2{ u2f(r0.x), u2f(r0.y) }; a.c1 = F2{ u2f(r1.x), u2f(r1.y) }; a.c2 = F2{ u2f(r2.x), u2f(r2.y) }; a.c3 = F2{ u2f(r3.x), u2f(r3.y) };
        a.z0 = u2f(r0.z); a.z1 = u2f(r1.z); a.z2 = u2f(r2.z); a.z3 = u2f(r3.z);
        if (a.instance == 1 || (a.instance == 2 && !(MEMO && sv.memo_obj == oi))) instances |= 1u << oi; else test(a, oi);
    }
#else
    FlatRecord a = flat_load(table);
    for (;;) {
        FlatRecord b;
        const bool more_b = oi + 1 < n_objects;
        if (more_b) b = flat_load(table + oi + 1);
        if (a.instance == 1 || (a.instance == 2 && !(MEMO && sv.memo_obj == oi))) instances |= 1u << oi; else test(a, oi);
        if (!more_b) break;
        ++oi;
        const bool more_a = oi + 1 < n_objects;
        if (more_a) a = flat_load(table + oi + 1);
        if (b.instance == 1 || (b.instance == 2 && !(MEMO && sv.memo_obj == oi))) instances |= 1u << oi; else test(b, oi);
        if (!more_a) break;
        ++oi;
    }
#endif
    while (instances) {   // uniform
        const uint32_t k = (uint32_t) __builtin_ctz(instances); instances &= instances - 1u;
        const bool hit = intersect_object<ANY, false, MEMO>(sv, k, o, d, time, maxt, best, stack, 0, blockDim.x);
        if (ANY) occluded |= hit;
    }
    return ANY ? occluded : best.obj != 0xffffffffu;
}

}  // namespace dtof
