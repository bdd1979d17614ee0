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
#include "dtof_device.h"
#include <algorithm>
#include <string>

namespace dtof {


// ---------------------------------------------------------------------------- generate
__global__ __launch_bounds__(kBlock) void k_generate(RenderParams rp, Queues q) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= rp.n_lanes) return;
    const PrimaryLane pl = generate_lane(rp, global_lane(rp, rp.lane_base + i), false, rp.lane_base + i);
    q.ray_a[i] = pl.ray_a;
    q.ray_b[i] = pl.ray_b;
    q.st_a[i] = make_float4(1.f, 1.f, 1.f, 0.f);
    q.rng_a[i] = make_uint4((uint32_t) pl.main.state, (uint32_t) (pl.main.state >> 32), (uint32_t) pl.path.state, (uint32_t) (pl.path.state >> 32));
    q.rng_b[i] = make_uint2((uint32_t) (pl.main.inc >> 1), (uint32_t) (pl.path.inc >> 1));
    q.pos[i] = pl.pos;
    for (int k = 0; k < rp.n_offsets; ++k) q.res[(size_t) k * q.capacity + i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------- trace
// BLOCK: 256 threads when the scene is staged into LDS (the staging is shared by four waves), ONE wave otherwise -- the waves of a
// block share nothing then, and a block only frees its LDS and wave slots when its slowest wave is done, which costs occupancy
// on divergent traversals (large scenes).
// W8: the ray kernels of scenes with LARGE triangle meshes and no analytic shapes (the mesh room: 522 k triangles behind two BLAS, bound by the latency of its L2 misses).
// Eight waves per SIMD instead of six -- more misses in flight per CU -- need (a) at most 64 VGPRs: the instantiation carries triangle and rectangle code only (MESH = 2: no
// float64 sphere / cylinder arithmetic), and (b) 32 one-wave blocks in a CU's LDS: the stack column holds kLdsStack8 entries, deeper ones overflow into a private array.
constexpr uint32_t kLdsStack8 = 16, kOvfStack8 = 48;
// TL (round 5): the eight-wave kernels were measured with their texture-address / data units 87 - 90 % busy (profiles/r05_mesh_room.txt): every node step is four 16-byte
// loads per lane, and a third of the loads a ray issues walk the TLAS -- a handful of nodes in such scenes (the mesh room: walls, a light, two blobs).  A TLAS of at most
// kTlasLds8 nodes is copied behind the block's stack column (32 blocks x (4 KiB + 1 KiB) = the CU's 160 KiB) and walked with ds_read_b128; the BLAS stay in global memory.
constexpr uint32_t kTlasLds8 = 16;
// XCD-aware block order of the unstaged ray kernels (guide: cdna_hip_programming.md T1).  Blocks are dealt round-robin over the 8 XCDs, each with an L2 of its own (4 MiB):
// with block b tracing queue segment b, every XCD sees rays from all over the image and all eight L2s fight over the same 35 MB of nodes and triangle records.  The remap gives
// the blocks that share an XCD a CONTIGUOUS run of segments (= a band of the image for the primary and shadow rays), so each L2 mostly holds the geometry of its band.
// Which block traces which segment changes nothing in the results.  MEASURED on the mesh room (512 x 512 x 64 spp, 522 k triangles; profiles/r05_mesh_room.txt), run = blocks
// an XCD gets in a row: off 6.90 G rays/s | 8 (one segment) 6.90 | 64 6.05 | 512 (ONE PIXEL ROW) 7.23 | 1 024 7.15 | 2 048 7.12 | 4 096 6.91 | a whole band per XCD 5.68 (the
// bands cost different amounts: the XCDs finish one after the other).  Default: one pixel row per run for scenes with a BLAS (xcd_run below); DTOF_XCD_REMAP=<run> | 0 overrides.
// `run` = consecutive segments one XCD gets before the next XCD's run starts (a whole-image band per XCD -- run = n / 8 -- LOST 18 % on the mesh room: the bands cost
// different amounts and the XCDs finish one after the other); blocks beyond the last full group of 8 runs keep their index.
DTOF_D uint32_t xcd_remap(uint32_t orig, uint32_t n, uint32_t run) {
    const uint32_t group = 8u * run, full = n - n % group;
    if (orig >= full) return orig;
    const uint32_t g = orig / group, w = orig - g * group, xcd = w & 7u, k = w >> 3;   // within a group: block w runs on XCD w % 8 and is that XCD's k-th block
    return g * group + xcd * run + k;
}
// DEFER: a one-wave block adds its rays that put objects aside to their segment's list (the order inside a segment's list is the order the blocks get there: it decides
// which lane of the second launch traces which ray, nothing else)
DTOF_D void defer_append(const Queues &q, uint32_t seg, bool put, uint32_t index, uint4 cand) {
    const unsigned long long m = __ballot(put);
    if (m == 0ull) return;
    uint32_t base = 0;
    if (threadIdx.x == 0) base = atomicAdd(&q.defer_cnt[seg], (uint32_t) __popcll(m));
    base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
    if (put) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
        q.defer_idx[seg * kSeg + base + rank] = index;
        q.cand[index] = cand;
    }
}
template <bool LDS, bool MESH, int BLOCK, bool W8 = false, bool TL = false, bool H16 = false, bool DEFER = false>
__global__ __launch_bounds__(BLOCK, W8 ? 8 : BLOCK == 64 ? 6 : 1) void k_trace(const uint8_t *scene, uint32_t scene_bytes, uint32_t stage_words,
                                                 Queues q, const uint32_t *qin, const uint32_t *count_in, uint32_t n_lanes, uint32_t n_tlas) {
    static_assert(!TL || W8, "the TLAS copy sits behind the eight-wave kernels' stack column");
    static_assert(!DEFER || (W8 && H16), "the two-launch form belongs to the eight-wave kernels");
    static_assert(!H16 || (!LDS && MESH && BLOCK == 64), "half-float nodes: the unstaged one-wave kernels of scenes with a BLAS");
    static_assert(!W8 || (!LDS && MESH && BLOCK == 64), "the eight-wave form exists for the unstaged one-wave kernels with triangle code");
    constexpr uint32_t kBlock = BLOCK, kSub = kSeg / BLOCK;
    extern __shared__ uint4 lds[];
    const uint32_t bid = !LDS && q.xcd_remap ? xcd_remap(blockIdx.x, gridDim.x, q.xcd_remap) : blockIdx.x;
    uint32_t seg = bid / kSub, sub = bid % kSub;
    uint32_t count = seg_count(count_in, seg, n_lanes);
    if (sub * kBlock >= count) return;
    const uint8_t *base = LDS ? stage_scene(scene, scene_bytes, lds) : scene;
    uint32_t *stack = (uint32_t *) (lds + stage_words) + threadIdx.x;
    SceneView sv = make_view(base);
    const BvhNode *tlas = nullptr;
    if (TL) {   // the block's own copy of the TLAS nodes, behind its stack column (one wave: the barrier is a wave barrier)
        uint4 *const t = lds + stage_words + kLdsStack8 * kBlock / 4u;
        for (uint32_t i = threadIdx.x; i < n_tlas * 4u; i += kBlock) t[i] = ((const uint4 *) sv.nodes)[i];
        __syncthreads();
        tlas = (const BvhNode *) t;
    }
    uint32_t j = sub * kBlock + threadIdx.x;
    const bool active = j < count;   // lanes past the end of the segment stay as helpers of the shared triangle loops (trace_rays)
    uint32_t l = 0; float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 1.f, 0.f);
    if (active) { l = qin ? qin[seg * kSeg + j] : seg * kSeg + j; a = q.ray_a[l]; b = q.ray_b[l]; }
    Hit h;
    uint32_t ovf[W8 ? kOvfStack8 : 1];
    uint4 cand = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    bool found = trace_rays<false, W8 ? 2 : (MESH ? 1 : 0), false, false, BLOCK, W8 ? kLdsStack8 : 0u, TL, H16, DEFER>(sv, stack, active, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), a.w, b.w, h, ovf, tlas, &cand);
    if (active) store_hit<MESH>(q, l, h, found);
    if (DEFER) defer_append(q, seg, active && cand.x != 0xffffffffu, l, cand);   // the hit so far is stored: k_trace_deferred goes on from it
}
// The second launch of the pair: the rays of each segment that reached at least one mesh behind a BLAS, packed -- block `sub` of a segment takes entries [64 sub, 64 sub + 64)
// of its list.  Every lane enters a BLAS, where the first launch's waves did so with a tenth of theirs (profiles/r05_mesh_room.txt, section 8).
template <bool H16>
__global__ __launch_bounds__(64, 8) void k_trace_deferred(const uint8_t *scene, Queues q, uint32_t n_seg) {
    extern __shared__ uint4 lds[];
    const uint32_t seg = blockIdx.x % n_seg, sub = blockIdx.x / n_seg;   // the blocks of one sub-range side by side: all but the first two or three ranges of a segment are empty
    const uint32_t count = q.defer_cnt[seg];
    if (sub * 64u >= count) return;
    uint32_t *stack = (uint32_t *) lds + threadIdx.x;
    const SceneView sv = make_view(scene);
    const uint32_t j = sub * 64u + threadIdx.x;
    if (j >= count) return;
    const uint32_t l = q.defer_idx[seg * kSeg + j];
    const float4 a = q.ray_a[l], b = q.ray_b[l];
    const uint4 cand = q.cand[l], hh = q.hit[l];
    const uint32_t hid = q.hit_id[l];
    Hit h;
    if (hid == 0xffffffffu) { h.t = b.w; h.u = h.v = 0.f; h.obj = 0xffffffffu; h.shape = 0; h.prim = 0; }
    else { h.t = u2f(hh.x); h.u = u2f(hh.y); h.v = u2f(hh.z); h.prim = hh.w; h.obj = hid & ((1u << q.id_shift) - 1u); h.shape = hid >> q.id_shift; }
    uint32_t ovf[kOvfStack8];
    const bool found = trace_deferred<false, 2, 64, kLdsStack8, H16>(sv, stack, cand, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), a.w, b.w, h, ovf);
    store_hit<true>(q, l, h, found);
}

// ---------------------------------------------------------------------------- shadow
// an unoccluded sample: its candidate results become the lane's results (the K films of a batch)
DTOF_D void shadow_commit(const RenderParams &rp, const Queues &q, uint32_t i) {
#pragma unroll
    for (int k = 0; k < kMaxOffsets; ++k) if (k < rp.n_offsets) {
        float4 c = q.sh_c[(size_t) k * q.capacity + i];
        uint32_t l = f2u(c.w);
        q.res[(size_t) k * q.capacity + l] = make_float4(c.x, c.y, c.z, 0.f);
    }
}
template <bool LDS, bool MESH, int BLOCK, bool W8 = false, bool TL = false, bool H16 = false, bool DEFER = false>
__global__ __launch_bounds__(BLOCK, W8 ? 8 : BLOCK == 64 ? 6 : 1) void k_shadow(const uint8_t *scene, uint32_t scene_bytes, uint32_t stage_words, RenderParams rp,
                                                  Queues q, const uint32_t *count_in, uint32_t n_tlas) {
    constexpr uint32_t kBlock = BLOCK, kSub = kSeg / BLOCK;
    extern __shared__ uint4 lds[];
    const uint32_t bid = !LDS && q.xcd_remap ? xcd_remap(blockIdx.x, gridDim.x, q.xcd_remap) : blockIdx.x;
    uint32_t seg = bid / kSub, sub = bid % kSub;
    uint32_t count = count_in[seg];
    if (sub * kBlock >= count) return;
    const uint8_t *base = LDS ? stage_scene(scene, scene_bytes, lds) : scene;
    uint32_t *stack = (uint32_t *) (lds + stage_words) + threadIdx.x;
    SceneView sv = make_view(base);
    const BvhNode *tlas = nullptr;
    if (TL) {   // the block's own copy of the TLAS nodes, behind its stack column (one wave: the barrier is a wave barrier)
        uint4 *const t = lds + stage_words + kLdsStack8 * kBlock / 4u;
        for (uint32_t i = threadIdx.x; i < n_tlas * 4u; i += kBlock) t[i] = ((const uint4 *) sv.nodes)[i];
        __syncthreads();
        tlas = (const BvhNode *) t;
    }
    uint32_t j = sub * kBlock + threadIdx.x;
    const bool active = j < count;
    uint32_t i = seg * kSeg + j;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 1.f, 0.f);
    if (active) { a = q.sh_a[i]; b = q.sh_b[i]; }
    Hit h;
    uint32_t ovf[W8 ? kOvfStack8 : 1];
    uint4 cand = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    bool occluded = trace_rays<true, W8 ? 2 : (MESH ? 1 : 0), false, false, BLOCK, W8 ? kLdsStack8 : 0u, TL, H16, DEFER>(sv, stack, active, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), b.w, a.w, h, ovf, tlas, &cand);
    const bool later = DEFER && active && !occluded && cand.x != 0xffffffffu;   // nothing else is in the way: the meshes it put aside decide (k_shadow_deferred)
    if (active && !occluded && !later) shadow_commit(rp, q, i);
    if (DEFER) defer_append(q, seg, later, i, cand);
}
template <bool H16>
__global__ __launch_bounds__(64, 8) void k_shadow_deferred(const uint8_t *scene, RenderParams rp, Queues q, uint32_t n_seg) {
    extern __shared__ uint4 lds[];
    const uint32_t seg = blockIdx.x % n_seg, sub = blockIdx.x / n_seg;
    const uint32_t count = q.defer_cnt[seg];
    if (sub * 64u >= count) return;
    uint32_t *stack = (uint32_t *) lds + threadIdx.x;
    const SceneView sv = make_view(scene);
    const uint32_t j = sub * 64u + threadIdx.x;
    if (j >= count) return;
    const uint32_t i = q.defer_idx[seg * kSeg + j];
    const float4 a = q.sh_a[i], b = q.sh_b[i];
    const uint4 cand = q.cand[i];
    Hit h; h.t = a.w; h.u = h.v = 0.f; h.obj = 0xffffffffu; h.shape = 0; h.prim = 0;
    uint32_t ovf[kOvfStack8];
    if (!trace_deferred<true, 2, 64, kLdsStack8, H16>(sv, stack, cand, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), b.w, a.w, h, ovf)) shadow_commit(rp, q, i);
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
    const bool active = i < rp.n_lanes;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 1.f, 0.f);
    if (active) { a = q.ray_a[i]; b = q.ray_b[i]; }
    V3 o = mk(a.x, a.y, a.z), d = mk(b.x, b.y, b.z);
    Hit h1, h2;
    bool v1 = trace_rays<false, true>(sv, stack, active, o, d, 0.f, b.w, h1);
    bool v2 = trace_rays<false, true>(sv, stack, active, o, d, rp.T, b.w, h2);
    float vel = ((v2 ? h2.t : 0.f) - (v1 ? h1.t : 0.f)) * (1.0f / rp.T);
    vel = (v1 && v2) ? vel : 0.f;
    if (active) q.res[i] = make_float4(vel, vel, vel, 0.f);
    if (active && rp.want_valid) q.valid_out[i] = make_float4(v1 && v2 ? 1.f : 0.f, 0.f, 0.f, 0.f);   // valid_ray = si2.is_valid() && si1.is_valid() (velocity.cpp:137)
}

// ---------------------------------------------------------------------------- splat
// (tent, filter_weight, dpp_add, splat_lane: dtof_device.h -- the fused first-bounce kernel splats too)
__global__ __launch_bounds__(kBlock) void k_splat_generic(RenderParams rp, Queues q, float *film, size_t film_stride) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= rp.n_lanes) return;
    float2 p = q.pos[i];
    uint32_t lane = global_lane(rp, rp.lane_base + i);
    uint32_t pix = fdiv(lane, rp.d_spp), W = (uint32_t) rp.crop_w;
    int py = (int) fdiv(pix, rp.d_w), px = (int) (pix - W * (uint32_t) py);
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
    int py = (int) fdiv(pix, rp.d_w), px = (int) (pix - W * (uint32_t) py);
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
            int sy = (int) fdiv(spix, rp.d_w), sx = (int) (spix - W * (uint32_t) sy);
            int x = sx - 1 + (int) ((c % 12) >> 2), y = sy - 1 + (int) (c / 12);
            float v = s_acc[idx];
            if ((unsigned) x < W && (unsigned) y < (unsigned) rp.crop_h && v != 0.f)
                atomicAdd(fk + 4 * ((size_t) y * W + (size_t) x) + (c & 3), v);
        }
    }
}

// The same sums with EIGHT SAMPLES PER LANE (spp a power of two >= 16), for every filter whose footprint is N x N pixels with N = 1 (box: the
// sample's own pixel), 3 (tent of radius <= 1) or 5 (radius <= 2: the default gaussian of hdrfilm, mitchell, catmullrom, wider tents): a lane
// accumulates the 4 N^2 footprint values of eight samples of its pixel serially in registers, and only then are the seg8 = min(spp / 8, 64)
// lanes that share a pixel reduced with DPP adds -- log2(seg8) steps of 4 N^2 adds per EIGHT samples instead of log2(min(spp, 64)) steps per
// sample (C2, 64 spp, tent: 13.5 instead of 216 DPP adds per sample).  Which samples a lane takes does not matter for the sums: lane `sub` of a
// segment takes the samples sub, sub + seg8, sub + 2 seg8, ... of the segment's 8 seg8 consecutive ones, so each load instruction reads seg8
// consecutive records per segment (64 spp: whole 128-byte lines of q.res).  Segment totals leave through LDS ((kBlock / seg8) x N^2 float4) and
// block-wide atomics as in k_splat_tent3.  The per-sample atomics of k_splat_generic cost 58x the rest of the frame on C2 with a gaussian.
constexpr uint32_t kSplatPer = 8;
template <int N, int F>
__global__ __launch_bounds__(kBlock) void k_splat_x8(RenderParams rp, Queues q, float *film, size_t film_stride, uint32_t seg8) {
    constexpr int NN = N * N, HALF = N / 2;
    extern __shared__ float4 s_acc4[];   // (kBlock / seg8) * NN
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x, sub = i & (seg8 - 1);
    const uint32_t first = (i - sub) * kSplatPer;           // first sample of this lane's segment (8 seg8 consecutive samples of one pixel)
    const bool in_range = first < rp.n_lanes;               // n_lanes is a multiple of spp, spp of 8 seg8
    const uint32_t lane = global_lane(rp, rp.lane_base + (in_range ? first : 0));
    const uint32_t pix = lane >> rp.spp_log2, W = (uint32_t) rp.crop_w;
    const int py = (int) fdiv(pix, rp.d_w), px = (int) (pix - W * (uint32_t) py);
    const float bx = (float) (px + rp.crop_x - HALF) + .5f, by = (float) (py + rp.crop_y - HALF) + .5f;
    for (int k = 0; k < rp.n_offsets; ++k) {
        float *fk = film + (size_t) k * film_stride;
        const float4 *res = q.res + (size_t) k * q.capacity + first + sub;
        const float2 *pos = q.pos + first + sub;
        float acc[4 * NN];
#pragma unroll
        for (int c = 0; c < 4 * NN; ++c) acc[c] = 0.f;
        uint32_t irregular = 0;
#pragma unroll 1
        for (uint32_t h = 0; h < kSplatPer; h += 4) {   // four samples at a time: their eight loads are issued together
            float4 r[4]; float2 pp[4];
#pragma unroll
            for (uint32_t m = 0; m < 4; ++m) {
                r[m] = in_range ? res[(size_t) (h + m) * seg8] : make_float4(0.f, 0.f, 0.f, 0.f);
                pp[m] = in_range && N > 1 ? pos[(size_t) (h + m) * seg8] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (uint32_t m = 0; m < 4; ++m) {
                if (N == 1) {   // box: block->put(pos) -- the lane's own pixel, weight 1 (integrator.cpp:540-541)
                    const float w = in_range ? 1.f : 0.f;
                    acc[0] += r[m].x; acc[1] += r[m].y; acc[2] += r[m].z; acc[3] += w;
                    continue;
                }
                const float sx = pp[m].x, sy = pp[m].y;
                const int fx = (int) floorf(sx) - rp.crop_x, fy = (int) floorf(sy) - rp.crop_y;
                const bool regular = in_range && fx == px && fy == py;
                if (in_range && !regular) irregular |= 1u << (h + m);
                const float relx = bx - sx, rely = by - sy;
                float wx[N], wy[N];
#pragma unroll
                for (int a = 0; a < N; ++a) { wx[a] = regular ? filter_weight<F>(rp, relx + (float) a) : 0.f; wy[a] = filter_weight<F>(rp, rely + (float) a); }
#pragma unroll
                for (int ys = 0; ys < N; ++ys)
#pragma unroll
                    for (int xs = 0; xs < N; ++xs) {
                        const float w = wx[xs] * wy[ys]; const int c = 4 * (N * ys + xs);
                        acc[c] = fmaf(r[m].x, w, acc[c]); acc[c + 1] = fmaf(r[m].y, w, acc[c + 1]); acc[c + 2] = fmaf(r[m].z, w, acc[c + 2]); acc[c + 3] += w;
                    }
            }
        }
#pragma unroll 1
        for (uint32_t m = 0; irregular >> m; ++m) if ((irregular >> m) & 1u) {   // positions that rounded into the next pixel (rare): splat by themselves
            const float2 sp = pos[(size_t) m * seg8]; const float4 sr = res[(size_t) m * seg8];
            splat_lane(rp, fk, sp.x, sp.y, px, py, sr.x, sr.y, sr.z);
        }
#pragma unroll
        for (int c = 0; c < 4 * NN; ++c) {   // segment totals end up in the segment's LAST lane (see k_splat_tent3)
            float v = acc[c];
            v = dpp_add<0xb1>(v);
            if (seg8 >= 4) v = dpp_add<0x4e>(v);
            if (seg8 >= 8) v = dpp_add<0x124>(v);
            if (seg8 >= 16) v = dpp_add<0x128>(v);
            if (seg8 >= 32) v = dpp_add<0x142, 0xa>(v);
            if (seg8 >= 64) v = dpp_add<0x143, 0xc>(v);
            acc[c] = v;
        }
        if (k > 0) __syncthreads();
        const uint32_t sidx_mine = threadIdx.x / seg8;
        if ((threadIdx.x & (seg8 - 1)) == seg8 - 1) {
#pragma unroll
            for (int c = 0; c < NN; ++c) s_acc4[sidx_mine * NN + c] = make_float4(acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]);
        }
        __syncthreads();
        const uint32_t total = (kBlock / seg8) * 4 * NN;
        const float *s_acc = (const float *) s_acc4;
        for (uint32_t idx = threadIdx.x; idx < total; idx += kBlock) {
            const uint32_t sidx = idx / (4 * NN), c = idx - sidx * (4 * NN), tap = c >> 2;
            const uint32_t seg_first = (blockIdx.x * kBlock + sidx * seg8) * kSplatPer;
            if (seg_first >= rp.n_lanes) continue;
            const uint32_t spix = global_lane(rp, rp.lane_base + seg_first) >> rp.spp_log2;
            const int sy = (int) fdiv(spix, rp.d_w), sx = (int) (spix - W * (uint32_t) sy);
            const int x = sx - HALF + (int) (tap % N), y = sy - HALF + (int) (tap / N);
            const float v = s_acc[idx];
            if ((unsigned) x < W && (unsigned) y < (unsigned) rp.crop_h && v != 0.f)
                atomicAdd(fk + 4 * ((size_t) y * W + (size_t) x) + (c & 3), v);
        }
    }
}

// Any sample count (not a power of two, below 16): ONE THREAD PER PIXEL (or per `parts`-th of a pixel's samples when the frame has too few
// pixels to fill the chip), one wave per block.  A thread accumulates the N x N footprint x (r, g, b, weight) of its own samples in registers,
// no cross-lane reduction.  The samples of a pixel are consecutive records of q.pos / q.res, so the wave fetches them COALESCED, eight samples of
// its 64 pixels per step (lane l loads sample l % 8 of pixel 8 i + l / 8: 128-byte pieces), and transposes them through LDS (row stride 9
// records: conflict-free column reads).  The sums leave through LDS one footprint row at a time: atomic wave-instructions whose lanes are
// (16 pixels) x (r, g, b, w), so the channels of one film record are updated by one instruction.  k_splat_generic's per-sample atomics on the
// 9 - 25 addresses of a pixel serialise: 34 ms instead of 1.8 for C2's frame at 48 spp.
constexpr uint32_t kSplatStep = 8, kSplatRow = kSplatStep + 1;
template <int N, int F>
__global__ __launch_bounds__(64) void k_splat_pixel(RenderParams rp, Queues q, float *film, size_t film_stride, uint32_t parts, uint32_t chunk, uint32_t n_threads) {
    constexpr int NN = N * N, HALF = N / 2;
    constexpr uint32_t kAccRow = 4 * N + 1;                                // one footprint row of one thread, padded
    constexpr uint32_t kTileF4 = 64 * kSplatRow + 32 * kSplatRow, kRowF4 = (64 * kAccRow + 3) / 4;
    __shared__ float4 s_mem[kTileF4 > kRowF4 ? kTileF4 : kRowF4];         // sample tiles; the row sums reuse the space once the samples are consumed
    __shared__ int2 s_anchor[64];
    float4 *const s_res = s_mem; float2 *const s_pos = (float2 *) (s_mem + 64 * kSplatRow); float *const s_acc = (float *) s_mem;
    const uint32_t l = threadIdx.x, t0 = blockIdx.x * 64;
    // thread t: pixel t / parts, samples [part * chunk, part * chunk + count) of it; first_of(t) = its first record, count_of(t) = how many
    auto first_of = [&](uint32_t t) { const uint32_t pixel = parts == 1 ? t : t / parts; return pixel * rp.spp + (t - pixel * parts) * chunk; };
    auto count_of = [&](uint32_t t) -> uint32_t {
        if (t >= n_threads) return 0u;
        const uint32_t pixel = parts == 1 ? t : t / parts, c0 = (t - pixel * parts) * chunk;
        return c0 >= rp.spp ? 0u : (c0 + chunk < rp.spp ? chunk : rp.spp - c0);
    };
    const uint32_t t = t0 + l, mine = count_of(t);
    const uint32_t pix = fdiv(global_lane(rp, rp.lane_base + (mine ? first_of(t) : 0u)), rp.d_spp), W = (uint32_t) rp.crop_w;
    const int py = (int) fdiv(pix, rp.d_w), px = (int) (pix - W * (uint32_t) py);
    const float bx = (float) (px + rp.crop_x - HALF) + .5f, by = (float) (py + rp.crop_y - HALF) + .5f;
    const uint32_t lm = l % kSplatStep, lj = l / kSplatStep;   // loader role: sample lm of the threads lj, lj + 8, ...
    for (int k = 0; k < rp.n_offsets; ++k) {
        float *fk = film + (size_t) k * film_stride;
        const float4 *res = q.res + (size_t) k * q.capacity;
        float acc[4 * NN];
#pragma unroll
        for (int c = 0; c < 4 * NN; ++c) acc[c] = 0.f;
        for (uint32_t s = 0; s < chunk; s += kSplatStep) {
            __syncthreads();   // one wave: orders the LDS reads of the previous step (or of the previous offset's epilogue) before these writes
#pragma unroll
            for (uint32_t i = 0; i < 64 / kSplatStep; ++i) {
                const uint32_t j = i * (64 / kSplatStep) + lj, tj = t0 + j;
                if (s + lm < count_of(tj)) {
                    const uint32_t rec = first_of(tj) + s + lm;
                    s_res[j * kSplatRow + lm] = res[rec];
                    if (N > 1) s_pos[j * kSplatRow + lm] = q.pos[rec];
                }
            }
            __syncthreads();
            const uint32_t n = s < mine ? (mine - s < kSplatStep ? mine - s : kSplatStep) : 0u;
#pragma unroll 2
            for (uint32_t m = 0; m < n; ++m) {
                const float4 r = s_res[l * kSplatRow + m];
                if (N == 1) { acc[0] += r.x; acc[1] += r.y; acc[2] += r.z; acc[3] += 1.f; continue; }   // box: the lane's own pixel, weight 1
                const float2 p = s_pos[l * kSplatRow + m];
                const int fx = (int) floorf(p.x) - rp.crop_x, fy = (int) floorf(p.y) - rp.crop_y;
                if (fx != px || fy != py) { splat_lane(rp, fk, p.x, p.y, px, py, r.x, r.y, r.z); continue; }   // the position rounded into the next pixel (rare)
                const float relx = bx - p.x, rely = by - p.y;
                float wx[N], wy[N];
#pragma unroll
                for (int a = 0; a < N; ++a) { wx[a] = filter_weight<F>(rp, relx + (float) a); wy[a] = filter_weight<F>(rp, rely + (float) a); }
#pragma unroll
                for (int ys = 0; ys < N; ++ys)
#pragma unroll
                    for (int xs = 0; xs < N; ++xs) {
                        const float w = wx[xs] * wy[ys]; const int c = 4 * (N * ys + xs);
                        acc[c] = fmaf(r.x, w, acc[c]); acc[c + 1] = fmaf(r.y, w, acc[c + 1]); acc[c + 2] = fmaf(r.z, w, acc[c + 2]); acc[c + 3] += w;
                    }
            }
        }
        s_anchor[l] = make_int2(mine ? px : -0x40000000, py);   // where thread l's footprint is anchored
#pragma unroll
        for (int ys = 0; ys < N; ++ys) {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 4 * N; ++c) s_acc[l * kAccRow + c] = acc[4 * N * ys + c];
            __syncthreads();
#pragma unroll 1
            for (uint32_t it = 0; it < 4 * N; ++it) {   // lanes: channel l % 4 of thread 16 * (it % 4) + l / 4, column it / 4 of this footprint row
                const uint32_t j = 16 * (it & 3) + l / 4, xs = it >> 2, ch = l & 3;
                const int2 anchor = s_anchor[j];
                const int x = anchor.x - HALF + (int) xs, y = anchor.y - HALF + ys;
                const float v = s_acc[j * kAccRow + 4 * xs + ch];
                if ((unsigned) x < W && (unsigned) y < (unsigned) rp.crop_h && v != 0.f)
                    atomicAdd(fk + 4 * ((size_t) y * W + (size_t) x) + ch, v);
            }
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

// pixel_format = rgba (hdrfilm.cpp:339-400): the block holds R, G, B, A, W and every channel is divided by W; here the alpha channel was accumulated into a
// film of its own, (A, 0, 0, W) with the same weights
__global__ void k_develop_rgba(const float *film, const float *alpha_film, float *rgba, int64_t n) {
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 f = ((const float4 *) film)[i], a = ((const float4 *) alpha_film)[i];
    const float w = f.w == 0.f ? 1.f : f.w, wa = a.w == 0.f ? 1.f : a.w;
    ((float4 *) rgba)[i] = make_float4(f.x / w, f.y / w, f.z / w, a.x / wa);
}

__global__ void k_lane_dump(RenderParams rp, Queues q, LaneDebug *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rp.n_lanes) return;
    float2 p = q.pos[i]; float4 r = q.res[i];
    LaneDebug &o = out[i];
    o.sample_pos[0] = p.x; o.sample_pos[1] = p.y;
    o.rgb[0] = r.x; o.rgb[1] = r.y; o.rgb[2] = r.z;
    o.valid = rp.want_valid ? q.valid_out[i].x : 0.f;
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
// Whole-blob staging pays while the blob is small against what a block's lanes read from it: measured on Domino fields (split pipeline,
// 1024^2 x 32 spp, one box): 4 KB blob staged 3.84 ms vs 4.28 ms unstaged, 9 KB 4.63 vs 5.16, 30 KB 10.39 vs 7.90 (every 64-lane shade
// block and every 256-lane trace block copies the blob; the L1 hit rate of the unstaged kernels is 99 % on such scenes).
constexpr uint32_t kLdsSceneLimit = 16 * 1024;
// Block size of the unstaged k_trace / k_shadow instantiations.  One wave per block when some mesh has its own BLAS: those
// traversals are long and divergent, and a 256-thread block keeps its LDS and wave slots until its slowest wave is done (mesh room,
// 522 k triangles: 20.1 -> 16.6 ms per frame).  Scenes of many small objects (Domino: 1 025 instances of a 12-triangle cube) are
// faster with four waves sharing a CU's L1 on the same TLAS / object records (71.3 vs 75.3 ms).  DTOF_TRACE_BLOCK = 64 | 128 | 256
// overrides (experiments).
// the eight-waves-per-SIMD ray kernels (k_trace / k_shadow<false, true, 64, true>): one-wave blocks of an unstaged scene whose meshes sit behind a BLAS and that has no
// analytic shape (their float64 code does not fit 64 VGPRs), stacks no deeper than the LDS part + the overflow array.  DTOF_TRACE8=0 keeps the six-wave kernels (A/B).
static inline bool eight_wave_rays(const RenderParams &rp, uint32_t stage_words, uint32_t block, uint32_t stack_depth);
static inline uint32_t unstaged_block(const RenderParams &rp) {
    static const uint32_t env = [] { const char *e = getenv("DTOF_TRACE_BLOCK"); int b = e ? atoi(e) : 0; return (uint32_t) (b == 64 || b == 128 || b == 256 ? b : 0); }();
    return env ? env : (rp.has_blas ? 64u : (uint32_t) kBlock);
}


static inline uint32_t xcd_run(const RenderParams &rp, uint32_t stage_words, uint32_t block) {
    if (const char *e = getenv("DTOF_XCD_REMAP")) return (uint32_t) atoi(e);   // read per call: A/B runs
    if (stage_words != 0 || !rp.has_blas) return 0;                              // scenes staged in LDS and scenes of small objects: L2 locality is not what they wait for
    const uint64_t row_blocks = (uint64_t) rp.crop_w * rp.spp / block;
    return (uint32_t) (row_blocks >= 8 && row_blocks <= (1u << 20) ? row_blocks : 0);
}
static inline int defer_rays(const Queues &q) {   // DTOF_DEFER=0: one launch per ray kernel (A/B, tests); 1: every ray kernel as a pair; 2: all but the primary rays'; 3: the shadow rays' only.  The workspace has the lists only for scenes with a BLAS (render_rows)
    const char *e = getenv("DTOF_DEFER"); const int v = e ? atoi(e) : 2;   // (measured: profiles/r05_mesh_room.txt section 8 -- primary rays that reach a blob already fill their waves)
    return q.cand != nullptr ? v : 0;
}
static inline bool half_nodes(const RenderParams &rp) {   // DTOF_NODES16=0: the 64-byte float nodes (A/B, tests)
    const char *e = getenv("DTOF_NODES16"); const bool off = e && e[0] == '0';
    return !off && rp.has_nodes16;
}
static inline bool tlas_in_lds(const RenderParams &rp) {   // DTOF_TLAS_LDS=0: the TLAS is walked in global memory like the BLAS (A/B, tests)
    const char *e = getenv("DTOF_TLAS_LDS"); const bool off = e && e[0] == '0';
    return !off && rp.n_tlas_nodes != 0 && rp.n_tlas_nodes <= kTlasLds8;
}
static inline bool eight_wave_rays(const RenderParams &rp, uint32_t stage_words, uint32_t block, uint32_t stack_depth) {
    const char *e = getenv("DTOF_TRACE8"); const bool off = e && e[0] == '0';   // read per call (a few launches per frame): tests and A/B runs switch it inside one process
    return !off && stage_words == 0 && block == 64 && rp.has_tris && rp.has_blas && !rp.has_analytic && stack_depth <= kLdsStack8 + kOvfStack8;   // (whatever the materials: the ray kernels only intersect)
}
void launch_generate(const RenderParams &rp, const Queues &q, hipStream_t s) {
    if (rp.n_lanes == 0) return;
    hipLaunchKernelGGL(k_generate, dim3(nblk(rp.n_lanes)), dim3(kBlock), 0, s, rp, q);
}
// the scene is staged into LDS if it is small AND leaves room for the traversal stacks within the 64 KiB a block may ask for
constexpr uint32_t kLdsBlockLimit = 64 * 1024;
static inline uint32_t stage_words_for(uint32_t scene_bytes, uint32_t stack = 0) {
    static const bool off = [] { const char *e = getenv("DTOF_STAGE"); return e && e[0] == '0'; }();   // DTOF_STAGE=0: never stage the scene into LDS (measurement)
    if (off) return 0;
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
uint32_t segments_for(uint32_t n_lanes) { return nseg(n_lanes); }

void launch_trace(const uint8_t *scene, uint32_t scene_bytes, const RenderParams &rp, const Queues &q,
                  const uint32_t *qin, const uint32_t *count_in, uint32_t stack_depth, hipStream_t s) {
    if (rp.n_lanes == 0) return;
    const uint32_t sw = stage_words_for(scene_bytes, stack_bytes(stack_depth)), block = sw ? kBlock : unstaged_block(rp);
    const bool w8 = eight_wave_rays(rp, sw, block, stack_depth), tl = w8 && tlas_in_lds(rp), h16 = w8 && half_nodes(rp);
    const uint32_t lds = sw * 16 + stack_bytes(w8 ? kLdsStack8 : stack_depth, block) + (tl ? kTlasLds8 * 64u : 0u), grid = nseg(rp.n_lanes) * (kSeg / block);
    check_lds(lds);
    Queues qx = q; qx.xcd_remap = xcd_run(rp, sw, block);
#define DTOF_LAUNCH_TRACE(L, M, B) hipLaunchKernelGGL((k_trace<L, M, B>), dim3(grid), dim3(B), lds, s, scene, scene_bytes, sw, qx, qin, count_in, rp.n_lanes, 0u)
    if (w8 && h16 && (defer_rays(q) == 1 || (defer_rays(q) == 2 && qin != nullptr))) {   // two launches: the TLAS walk of every ray, then the BLAS walks of the rays that reached a mesh, packed (k_trace_deferred)
        const uint32_t n_seg = nseg(rp.n_lanes);
        (void) hipMemsetAsync(q.defer_cnt, 0, (size_t) n_seg * 4, s);
        if (tl) hipLaunchKernelGGL((k_trace<false, true, 64, true, true, true, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, qx, qin, count_in, rp.n_lanes, rp.n_tlas_nodes);
        else hipLaunchKernelGGL((k_trace<false, true, 64, true, false, true, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, qx, qin, count_in, rp.n_lanes, 0u);
        hipLaunchKernelGGL((k_trace_deferred<true>), dim3(grid), dim3(64), stack_bytes(kLdsStack8, 64), s, scene, qx, n_seg);
    }
    else if (w8 && h16) { if (tl) hipLaunchKernelGGL((k_trace<false, true, 64, true, true, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, qx, qin, count_in, rp.n_lanes, rp.n_tlas_nodes);
                     else hipLaunchKernelGGL((k_trace<false, true, 64, true, false, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, qx, qin, count_in, rp.n_lanes, 0u); }
    else if (tl) hipLaunchKernelGGL((k_trace<false, true, 64, true, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, qx, qin, count_in, rp.n_lanes, rp.n_tlas_nodes);
    else if (w8) hipLaunchKernelGGL((k_trace<false, true, 64, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, qx, qin, count_in, rp.n_lanes, 0u);
    else if (sw) { if (rp.has_tris) DTOF_LAUNCH_TRACE(true, true, kBlock); else DTOF_LAUNCH_TRACE(true, false, kBlock); }
    else if (block == 64 && rp.has_tris && rp.has_blas && half_nodes(rp))   // (a scene with a BLAS AND analytic shapes: six waves, every shape's code, half-float nodes)
        hipLaunchKernelGGL((k_trace<false, true, 64, false, false, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, qx, qin, count_in, rp.n_lanes, 0u);
    else if (block == 64)  { if (rp.has_tris) DTOF_LAUNCH_TRACE(false, true, 64); else DTOF_LAUNCH_TRACE(false, false, 64); }
    else if (block == 128) { if (rp.has_tris) DTOF_LAUNCH_TRACE(false, true, 128); else DTOF_LAUNCH_TRACE(false, false, 128); }
    else                   { if (rp.has_tris) DTOF_LAUNCH_TRACE(false, true, kBlock); else DTOF_LAUNCH_TRACE(false, false, kBlock); }
#undef DTOF_LAUNCH_TRACE
}
uint32_t resident_lds_bytes(const RenderParams &rp, const ResidentStage &resident, uint32_t stack_depth, uint32_t waves) {
    static_assert(kResidentNodes == kResNodes, "resident stage size");
    const uint32_t memo = rp.memo_obj != 0xffffffffu ? 1u : 0u;
    // film-state columns of the several-film kernels (k_shade: RES_LDS) + parked path state (PARK: 16 waves only; the several-film kernels have room for the two streams)
    const uint32_t park_words = rp.n_offsets != 1 ? kParkWords + (DTOF_PARK && waves == 16 ? kParkRng : 0u) : (DTOF_PARK && waves == 16 ? kParkState : 0u);
    return (4u * kResNodes + resident.small_words) * 16u + memo * waves * kMemoWords * kMemoStride * 4u + resident_stack_bytes(stack_depth, waves, rp.n_offsets != 1) + park_words * waves * 64u * 4u;
}
uint32_t device_lds_limit() {
    if (const char *e = getenv("DTOF_LDS_LIMIT")) return (uint32_t) strtoul(e, nullptr, 10);   // tests: a smaller budget than the device's (the step-down / fallback paths)
    int dev = 0, bytes = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || bytes <= 0) return 64u * 1024u;
    return (uint32_t) bytes > 1024u ? (uint32_t) bytes - 1024u : 0u;   // k_shade's static LDS (count slots) comes on top of the dynamic size
}
void launch_shade(const uint8_t *scene, uint32_t scene_bytes, const RenderParams &rp, const Queues &q,
                  const uint32_t *qin, const uint32_t *count_in, uint32_t *qout,
                  uint32_t *alive_out, uint32_t *shadow_out, uint32_t depth, bool fused, bool trace_next,
                  uint32_t stack_depth, hipStream_t s, bool first, LaneDebug *dbg, const ResidentStage *resident, float *film, uint64_t film_stride) {
    if (rp.n_lanes == 0) return;
    if (first && !fused) throw std::runtime_error("the first-bounce kernel exists in the fused pipeline only");
    const bool k4 = rp.n_offsets != 1;
    if (resident && resident->waves && first && fused && rp.has_tris && rp.chunk_blocks <= 1) {
        // one block of `waves` waves per CU; LDS = node planes + record block + (instance memo) + stack columns, well above the 64 KiB default limit.
        // render_rows only offers the stage with a wave count that fits; a scene that still does not (a deeper stack than it assumed) takes the classic launch below.
        const uint32_t waves = resident->waves, lds = resident_lds_bytes(rp, *resident, stack_depth, waves);
        if (lds <= device_lds_limit()) {
            int dev = 0, n_cu = 0;
            (void) hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
            const uint32_t n_seg = nseg(rp.n_lanes), grid = std::min<uint32_t>((uint32_t) n_cu, (n_seg + waves - 1) / waves);
            const uint32_t memo = rp.memo_obj != 0xffffffffu ? 1u : 0u;
            const ShadeLaunch L = { false, 2, waves, grid, lds, s,
                                    { scene, scene_bytes, 0u, rp, q, qin, count_in, qout, alive_out, shadow_out, depth, trace_next ? 1u : 0u, dbg, film, film_stride, n_seg, resident->small_off, resident->small_words, memo, resident_stack_bytes(stack_depth, waves, rp.n_offsets != 1) / 4u } };
            if (hipMemsetAsync(q.seg_counter, 0, 4, s) != hipSuccess) throw std::runtime_error("hipMemsetAsync(seg_counter) failed");
            if (rp.has_spec == 2) launch_shade_resident2(k4, L);
            else if (rp.has_spec) launch_shade_resident1(k4, L);
            else launch_shade_resident0(rp.has_area != 0, k4, L);
            return;
        }
    }
    // + the instance memo; a flat scene with one instance keeps the instance matrix there as well (k_shade: memo_m_lds) and needs no traversal stack beyond one entry
    const bool memo_m = fused && !rp.has_tris && !rp.has_spec && rp.flat_objects != 0 && rp.memo_obj != 0xffffffffu;   // = the condition of k_shade's memo_m_lds in the instantiations launch_shade_plain picks
    const uint32_t shade_stack = fused ? stack_bytes(memo_m ? 1u : stack_depth, kShadeBlock) + (memo_m ? 2u : 1u) * kMemoWords * kMemoStride * 4 : 0;
    const uint32_t sw = stage_words_for(scene_bytes, shade_stack), grid = nseg(rp.n_lanes) * (first && rp.chunk_blocks > 1 ? rp.chunk_blocks : 1u), lds = sw * 16 + shade_stack;
    check_lds(lds);
    const ShadeLaunch L = { sw != 0, first ? 2 : fused ? 1 : 0, 0u, grid, lds, s,
                            { scene, scene_bytes, sw, rp, q, qin, count_in, qout, alive_out, shadow_out, depth, trace_next ? 1u : 0u, dbg, film, film_stride, nseg(rp.n_lanes), 0u, 0u, 0u, 0u } };
    if (rp.has_spec == 2) launch_shade_spec2(k4, L);
    else if (rp.has_spec) launch_shade_spec1(k4, L);
    else if (rp.has_tris) launch_shade_mesh(rp.has_area != 0, k4, L);
    else launch_shade_plain(rp.has_area != 0, k4, L);
}
void launch_shadow(const uint8_t *scene, uint32_t scene_bytes, const RenderParams &rp, const Queues &q,
                   const uint32_t *count_in, uint32_t stack_depth, hipStream_t s) {
    if (rp.n_lanes == 0) return;
    const uint32_t sw = stage_words_for(scene_bytes, stack_bytes(stack_depth)), block = sw ? kBlock : unstaged_block(rp);
    const bool w8 = eight_wave_rays(rp, sw, block, stack_depth), tl = w8 && tlas_in_lds(rp), h16 = w8 && half_nodes(rp);
    const uint32_t lds = sw * 16 + stack_bytes(w8 ? kLdsStack8 : stack_depth, block) + (tl ? kTlasLds8 * 64u : 0u), grid = nseg(rp.n_lanes) * (kSeg / block);
    check_lds(lds);
    Queues qx = q; qx.xcd_remap = xcd_run(rp, sw, block);
#define DTOF_LAUNCH_SHADOW(L, M, B) hipLaunchKernelGGL((k_shadow<L, M, B>), dim3(grid), dim3(B), lds, s, scene, scene_bytes, sw, rp, qx, count_in, 0u)
    if (w8 && h16 && defer_rays(q)) {
        const uint32_t n_seg = nseg(rp.n_lanes);
        (void) hipMemsetAsync(q.defer_cnt, 0, (size_t) n_seg * 4, s);
        if (tl) hipLaunchKernelGGL((k_shadow<false, true, 64, true, true, true, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, rp, qx, count_in, rp.n_tlas_nodes);
        else hipLaunchKernelGGL((k_shadow<false, true, 64, true, false, true, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, rp, qx, count_in, 0u);
        hipLaunchKernelGGL((k_shadow_deferred<true>), dim3(grid), dim3(64), stack_bytes(kLdsStack8, 64), s, scene, rp, qx, n_seg);
    }
    else if (w8 && h16) { if (tl) hipLaunchKernelGGL((k_shadow<false, true, 64, true, true, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, rp, qx, count_in, rp.n_tlas_nodes);
                     else hipLaunchKernelGGL((k_shadow<false, true, 64, true, false, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, rp, qx, count_in, 0u); }
    else if (tl) hipLaunchKernelGGL((k_shadow<false, true, 64, true, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, rp, qx, count_in, rp.n_tlas_nodes);
    else if (w8) hipLaunchKernelGGL((k_shadow<false, true, 64, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, rp, qx, count_in, 0u);
    else if (sw) { if (rp.has_tris) DTOF_LAUNCH_SHADOW(true, true, kBlock); else DTOF_LAUNCH_SHADOW(true, false, kBlock); }
    else if (block == 64 && rp.has_tris && rp.has_blas && half_nodes(rp))
        hipLaunchKernelGGL((k_shadow<false, true, 64, false, false, true>), dim3(grid), dim3(64), lds, s, scene, scene_bytes, sw, rp, qx, count_in, 0u);
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
void launch_splat(const RenderParams &rp, const Queues &q, float *film, uint64_t plane_stride, hipStream_t s) {
    if (rp.n_lanes == 0) return;
    size_t stride = (size_t) plane_stride;   // floats between the films of the batched offsets
    bool fast = rp.filter == FILTER_TENT && rp.filter_radius <= 1.f && rp.filter_radius > .5f && rp.spp_log2 != 0xffffffffu && rp.spp >= 2;
    static const int env_splat = [] { const char *e = getenv("DTOF_SPLAT"); std::string v = e ? e : ""; return v == "dpp" ? 1 : v == "generic" ? 2 : 0; }();   // A/B switches
    // footprint of the filter in pixels (ImageBlock::put: the pixels within ceil(radius - 0.5) of the sample's): 1 (box), 3 or 5 take the
    // eight-samples-per-lane kernel when spp is a power of two >= 16
    const int reach = rp.filter == FILTER_BOX ? 0 : (int) ceilf(rp.filter_radius - .5f);
    // the Lanczos filter (default radius 3: a 7 x 7 footprint) always takes the per-lane kernel
    const bool lanczos = rp.filter == FILTER_LANCZOS;
    if ((rp.filter == FILTER_BOX || (reach >= 1 && reach <= 2)) && !lanczos && rp.spp_log2 != 0xffffffffu && rp.spp >= 2 * kSplatPer && env_splat == 0) {
        const uint32_t groups = rp.n_lanes / kSplatPer, seg8 = rp.spp / kSplatPer < 64 ? rp.spp / kSplatPer : 64;
        const int n = 2 * reach + 1;
        const uint32_t lds = (kBlock / seg8) * n * n * 16u;
        if (lds <= 64u * 1024u) {
#define DTOF_SPLAT_X8(NN_, F_) hipLaunchKernelGGL((k_splat_x8<NN_, F_>), dim3(nblk(groups)), dim3(kBlock), lds, s, rp, q, film, stride, seg8)
            if (n == 1) DTOF_SPLAT_X8(1, FILTER_BOX);
            else if (rp.filter == FILTER_TENT) { if (n == 3) DTOF_SPLAT_X8(3, FILTER_TENT); else DTOF_SPLAT_X8(5, FILTER_TENT); }
            else if (rp.filter == FILTER_GAUSSIAN) { if (n == 3) DTOF_SPLAT_X8(3, FILTER_GAUSSIAN); else DTOF_SPLAT_X8(5, FILTER_GAUSSIAN); }
            else if (rp.filter == FILTER_MITCHELL) { if (n == 3) DTOF_SPLAT_X8(3, FILTER_MITCHELL); else DTOF_SPLAT_X8(5, FILTER_MITCHELL); }
            else { if (n == 3) DTOF_SPLAT_X8(3, FILTER_CATMULLROM); else DTOF_SPLAT_X8(5, FILTER_CATMULLROM); }
#undef DTOF_SPLAT_X8
            return;
        }
    }
    const bool small_pow2_tent = fast && rp.spp < 2 * kSplatPer;   // 2, 4, 8 spp under the radius-1 tent: k_splat_tent3 (one DPP segment per pixel)
    if ((rp.filter == FILTER_BOX || (reach >= 1 && reach <= 2)) && !lanczos && !small_pow2_tent && env_splat == 0) {
        // any other sample count: one thread per pixel; enough threads to fill the chip (parts of a pixel's samples, each >= 8, when the frame is small)
        const uint32_t n_pixels = rp.n_lanes / rp.spp, n = 2 * reach + 1;
        uint32_t parts = 1;
        while ((uint64_t) n_pixels * parts < 131072u && rp.spp / (parts * 2) >= 8) parts *= 2;
        const uint32_t chunk = (rp.spp + parts - 1) / parts, threads = n_pixels * parts;
#define DTOF_SPLAT_PIXEL(NN_, F_) hipLaunchKernelGGL((k_splat_pixel<NN_, F_>), dim3((threads + 63) / 64), dim3(64), 0, s, rp, q, film, stride, parts, chunk, threads)
        if (n == 1) DTOF_SPLAT_PIXEL(1, FILTER_BOX);
        else if (rp.filter == FILTER_TENT) { if (n == 3) DTOF_SPLAT_PIXEL(3, FILTER_TENT); else DTOF_SPLAT_PIXEL(5, FILTER_TENT); }
        else if (rp.filter == FILTER_GAUSSIAN) { if (n == 3) DTOF_SPLAT_PIXEL(3, FILTER_GAUSSIAN); else DTOF_SPLAT_PIXEL(5, FILTER_GAUSSIAN); }
        else if (rp.filter == FILTER_MITCHELL) { if (n == 3) DTOF_SPLAT_PIXEL(3, FILTER_MITCHELL); else DTOF_SPLAT_PIXEL(5, FILTER_MITCHELL); }
        else { if (n == 3) DTOF_SPLAT_PIXEL(3, FILTER_CATMULLROM); else DTOF_SPLAT_PIXEL(5, FILTER_CATMULLROM); }
#undef DTOF_SPLAT_PIXEL
        return;
    }
    if (fast && env_splat != 2) {
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
void launch_develop_rgba(const float *film, const float *alpha_film, float *rgba, int64_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_develop_rgba, dim3((uint32_t) ((n + 255) / 256)), dim3(256), 0, s, film, alpha_film, rgba, n);
}
// multi-pass renders: the main / path stream states the lanes of this batch ended the pass with (generate_lane wrote the time stream's)
__global__ void k_pass_save(RenderParams rp, Queues q) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rp.n_lanes) return;
    const uint4 rs = q.rng_a[i];
    uint2 *dst = rp.pass_rng + (size_t) (rp.lane_base + i - rp.pass_first) * 3;
    dst[0] = make_uint2(rs.x, rs.y); dst[2] = make_uint2(rs.z, rs.w);
}
void launch_pass_save(const RenderParams &rp, const Queues &q, hipStream_t s) {
    if (rp.n_lanes && rp.pass_rng) hipLaunchKernelGGL(k_pass_save, dim3(nblk(rp.n_lanes)), dim3(kBlock), 0, s, rp, q);
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
    Rng a = seed_stream(rp.seed_value, i), b = seed_stream(rp.seed_value + 1, fdiv(i, rp.d_tcn)), c = seed_stream(rp.seed_value + 2, fdiv(i, rp.d_pcn));
    st.rng[i] = make_uint2((uint32_t) a.state, (uint32_t) (a.state >> 32));
    st.rng_time[i] = make_uint2((uint32_t) b.state, (uint32_t) (b.state >> 32));
    st.rng_path[i] = make_uint2((uint32_t) c.state, (uint32_t) (c.state >> 32));
    uint32_t ps, tmp; tea32(rp.base_seed, rp.spp * fdiv(i, rp.d_spp) + rp.seed, ps, tmp);   // compute_per_sequence_seed, sampler.cpp:85-92
    st.perm_seed[i] = ps; st.dim[i] = 0;
}
DTOF_D Rng load_rng(const uint2 *arr, uint32_t i, uint64_t inc) { Rng r; uint2 v = arr[i]; r.state = (uint64_t) v.x | ((uint64_t) v.y << 32); r.inc = inc; return r; }
DTOF_D void store_rng(uint2 *arr, uint32_t i, const Rng &r) { arr[i] = make_uint2((uint32_t) r.state, (uint32_t) (r.state >> 32)); }

__global__ void k_sampler_next_correlate(RenderParams rp, SamplerState st, const uint8_t *correlate, int correlate_all, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= st.n) return;
    Rng m = load_rng(st.rng, i, stream_inc(rp.seed_value, i)), p = load_rng(st.rng_path, i, stream_inc(rp.seed_value + 2, fdiv(i, rp.d_pcn)));
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
    Rng m = load_rng(st.rng, i, stream_inc(rp.seed_value, i)), t = load_rng(st.rng_time, i, stream_inc(rp.seed_value + 1, fdiv(i, rp.d_tcn)));
    uint32_t si = sample_index_base + (rp.spp > 1 ? i - rp.spp * fdiv(i, rp.d_spp) : 0), dim = st.dim[i];
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

// ---------------------------------------------------------------------------- component evaluation (known-answer entry points)
// dtof_eval_component: the device functions the shade / splat kernels are made of, over arrays.  One thread per element.
__global__ void k_component(ComponentArgs a, RenderParams rp) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float *in = a.in + (size_t) i * a.in_stride; float *out = a.out + (size_t) i * a.out_stride;
    switch (a.component) {
        case COMP_MICROFACET_EVAL: case COMP_MICROFACET_PDF: case COMP_MICROFACET_G1: case COMP_MICROFACET_SAMPLE: {
            const Ggx g = mf_make((int) a.p[0], a.p[1], a.p[2], (int) a.p[3]);
            if (a.component == COMP_MICROFACET_EVAL) out[0] = ggx_eval(g, mk(in[0], in[1], in[2]));
            else if (a.component == COMP_MICROFACET_PDF) out[0] = ggx_pdf(g, mk(in[0], in[1], in[2]), mk(in[3], in[4], in[5]));
            else if (a.component == COMP_MICROFACET_G1) out[0] = ggx_smith_g1(g, mk(in[0], in[1], in[2]), mk(in[3], in[4], in[5]));
            else { float pdf; const V3 m = ggx_sample(g, mk(in[0], in[1], in[2]), in[3], in[4], pdf); out[0] = m.x; out[1] = m.y; out[2] = m.z; out[3] = pdf; }
        } break;
        case COMP_FRESNEL: fresnel_dielectric(in[0], a.p[0], out[0], out[1], out[2], out[3]); break;
        case COMP_FRESNEL_CONDUCTOR: out[0] = fresnel_conductor(in[0], a.p[0], a.p[1]); break;
        case COMP_RFILTER: {
            const float x = in[0], r = rp.filter_radius;
            if (rp.filter == FILTER_BOX) out[0] = (x >= -r && x < r) ? 1.f : 0.f;     // BoxFilter::eval (src/rfilters/box.cpp)
            else out[0] = fabsf(x) < r ? filter_weight(rp, x) : 0.f;
        } break;
        case COMP_WARP_COSINE_HEMISPHERE: { const V3 d = cosine_hemisphere(in[0], in[1]); out[0] = d.x; out[1] = d.y; out[2] = d.z; } break;
        case COMP_WARP_DISK_CONCENTRIC: concentric_disk(in[0], in[1], out[0], out[1]); break;
        case COMP_WARP_UNIFORM_TRIANGLE: uniform_triangle(in[0], in[1], out[0], out[1]); break;
        case COMP_WARP_UNIFORM_SPHERE: { const V3 d = uniform_sphere(in[0], in[1]); out[0] = d.x; out[1] = d.y; out[2] = d.z; } break;
        case COMP_COORDINATE_SYSTEM: { V3 s, t; coordinate_system(mk(in[0], in[1], in[2]), s, t); out[0] = s.x; out[1] = s.y; out[2] = s.z; out[3] = t.x; out[4] = t.y; out[5] = t.z; } break;
        case COMP_TEA_FLOAT32: { uint32_t v0, v1; tea32(f2u(in[0]), f2u(in[1]), v0, v1); out[0] = u2f((v1 >> 9) | 0x3f800000u) - 1.f; } break;   // sample_tea_float32 (random.h:63-67): the second word
        case COMP_MATH: {
            const float x = in[0]; const int fn = (int) a.p[0]; float s_, c_;
            out[0] = fn == 0 ? exp_(x) : fn == 1 ? log_(x) : fn == 2 ? tan_(x) : fn == 3 ? erf_(x) : fn == 4 ? erfinv_(x)
                   : fn == 5 ? (sincos_(x, s_, c_), s_) : fn == 6 ? cos_(x) : acos_(x);
        } break;
        default: break;
    }
}
// Sensor::sample_ray over arrays (dtof_camera_rays): in = position sample x, y in [0, 1]^2 of the crop window, aperture sample x, y; out = o[3], d[3], maxt
__global__ void k_camera_rays(RenderParams rp, const float *in, float *out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o, d; float maxt;
    camera_ray<false>(rp, in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3], o, d, maxt);
    float *w = out + (size_t) i * 7;
    w[0] = o.x; w[1] = o.y; w[2] = o.z; w[3] = d.x; w[4] = d.y; w[5] = d.z; w[6] = maxt;
}
void launch_camera_rays(const RenderParams &rp, const float *in, float *out, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_camera_rays, dim3(nblk(n)), dim3(kBlock), 0, s, rp, in, out, n);
}
void launch_component(const ComponentArgs &a, const RenderParams &rp, hipStream_t s) {
    if (a.n) hipLaunchKernelGGL(k_component, dim3(nblk(a.n)), dim3(kBlock), 0, s, a, rp);
}

// Scene::ray_intersect / ray_test over arrays (dtof_ray_intersect): closest hit + surface interaction, or occlusion only.
// rays: o[3], d[3], time, maxt (8 floats); out: t, p[3], n[3], sh_n[3], sh_s[3], sh_t[3], wi[3] (19 floats); ids: object, shape, prim
template <bool ANY>
__global__ __launch_bounds__(64) void k_ray_query(const uint8_t *scene, const float *rays, float *out, int32_t *ids, float *uv4, uint32_t n) {
    extern __shared__ uint4 lds[];
    uint32_t *stack = (uint32_t *) lds + threadIdx.x;
    const SceneView sv = make_view(scene);
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    float r[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f };
    if (active) for (int k = 0; k < 8; ++k) r[k] = rays[(size_t) i * 8 + k];
    const V3 o = mk(r[0], r[1], r[2]), d = mk(r[3], r[4], r[5]);
    Hit h;
    const bool found = trace_rays<ANY, true>(sv, stack, active, o, d, r[6], r[7], h);
    if (!active) return;
    if (ANY) { ids[i] = found ? 1 : 0; return; }
    float *w = out + (size_t) i * 19;
    for (int k = 0; k < 19; ++k) w[k] = 0.f;
    ids[3 * i] = found ? (int32_t) h.obj : -1; ids[3 * i + 1] = found ? (int32_t) h.shape : -1; ids[3 * i + 2] = found ? (int32_t) h.prim : -1;
    if (uv4) for (int k = 0; k < 4; ++k) uv4[(size_t) i * 4 + k] = 0.f;
    if (!found) { w[0] = u2f(0x7f800000u); return; }
    Surface si;
    compute_surface<true>(sv, h.obj, h.shape, h.prim, h.t, h.u, h.v, o, d, r[6], si);
    w[0] = h.t;
    const V3 f[6] = { si.p, si.n, si.sh_n, si.sh_s, si.sh_t, si.wi };
    for (int k = 0; k < 6; ++k) { w[1 + 3 * k] = f[k].x; w[2 + 3 * k] = f[k].y; w[3 + 3 * k] = f[k].z; }
    if (uv4) { uv4[(size_t) i * 4] = si.u; uv4[(size_t) i * 4 + 1] = si.v; uv4[(size_t) i * 4 + 2] = h.u; uv4[(size_t) i * 4 + 3] = h.v; }   // si.uv, pi.prim_uv
}
void launch_ray_query(const uint8_t *scene, const float *rays, float *out, int32_t *ids, float *uv4, uint32_t n, bool any, uint32_t stack_depth, hipStream_t s) {
    if (!n) return;
    const uint32_t lds = stack_bytes(stack_depth, 64);
    check_lds(lds);
    if (any) hipLaunchKernelGGL(k_ray_query<true>, dim3((n + 63) / 64), dim3(64), lds, s, scene, rays, out, ids, uv4, n);
    else hipLaunchKernelGGL(k_ray_query<false>, dim3((n + 63) / 64), dim3(64), lds, s, scene, rays, out, ids, uv4, n);
}

#ifdef DTOF_TRAVERSAL_STATS
static bool (*g_stats_readers[16])(unsigned long long *); static int g_n_stats_readers;   // zero-initialised before any dynamic initialiser runs
void register_traversal_stats_reader(bool (*reader)(unsigned long long *acc8)) { if (g_n_stats_readers < 16) g_stats_readers[g_n_stats_readers++] = reader; }
bool read_traversal_stats(unsigned long long *out8) {
    for (int i = 0; i < 16; ++i) out8[i] = 0;
    for (int k = 0; k < g_n_stats_readers; ++k) if (!g_stats_readers[k](out8)) return false;
    return true;
}
#endif

}  // namespace dtof
