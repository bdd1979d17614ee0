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

#define DTOF_D __device__ __forceinline__
namespace dtof { constexpr int kBlock = 256; }
#include "dtof_traverse.h"
#include "dtof_sampling.h"
#include "dtof_shading.h"

#include <stdexcept>

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
// Segmented queues: the wavefront is cut into segments of kSeg lanes.  A shade block owns one
// segment: it compacts the survivors (and the shadow rays) of its segment to the front of the same
// segment of the output queue and records the count -- order preserving, deterministic and without a
// single global atomic (a shared counter serialises at ~88 returning atomics/us on MI355X, which
// made the first version of this kernel 10x slower than its memory traffic).
constexpr uint32_t kSeg = 512;
static_assert(kSeg / 64 == kChunkBlocks, "chunks per segment");
constexpr int kShadeBlock = 64;   // k_shade runs ONE wave per block: compaction is ballot+popcount only, no barrier in the chunk loop
// The closest-hit record between the trace and the shade of a bounce: (t, u, v, primitive) + the object / shape id.  Rectangle-only
// instantiations (MESH = false) keep the distance alone: a rectangle's surface interaction is rebuilt from the ray and t
// (rectangle.cpp:250-323 recomputes the local hit point), its primitive index is 0 -- 12 bytes less to write and to read per path vertex.
template <bool MESH> DTOF_D void store_hit(const Queues &q, uint32_t l, const Hit &h, bool found) {
    if (MESH) q.hit[l] = make_uint4(f2u(h.t), f2u(h.u), f2u(h.v), h.prim);
    else q.hit_t[l] = h.t;
    q.hit_id[l] = found ? (h.obj | (h.shape << q.id_shift)) : 0xffffffffu;
}
template <bool MESH> DTOF_D uint4 load_hit(const Queues &q, uint32_t l) {
    if (MESH) return q.hit[l];
    return make_uint4(f2u(q.hit_t[l]), 0u, 0u, 0u);
}
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
    const bool active = j < count;   // lanes past the end of the segment stay as helpers of the shared triangle loops (trace_rays)
    uint32_t l = 0; float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 1.f, 0.f);
    if (active) { l = qin ? qin[seg * kSeg + j] : seg * kSeg + j; a = q.ray_a[l]; b = q.ray_b[l]; }
    Hit h;
    bool found = trace_rays<false, MESH>(sv, stack, active, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), a.w, b.w, h);
    if (active) store_hit<MESH>(q, l, h, found);
}

// ---------------------------------------------------------------------------- shade

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
// The arguments of k_shade travel as ONE by-value block and are read through the kernarg segment pointer, re-based once per chunk
// on an offset the compiler cannot see through: the ~170 dwords of parameters (camera matrices, sampler and modulation constants, 17
// queue pointers) are then fetched by scalar loads where they are used, instead of being loaded once ahead of the chunk loop and
// kept alive across it -- which, in the first-bounce instantiation, spilled ~120 SGPRs to VGPR lanes (350 v_writelane / v_readlane).
struct ShadeArgs {
    const uint8_t *scene; uint32_t scene_bytes, stage_words; RenderParams rp; Queues q;
    const uint32_t *qin, *count_in; uint32_t *qout, *alive_out, *shadow_out; uint32_t depth, trace_next; LaneDebug *dbg;
    uint32_t n_seg, res_small_off, res_small_words, res_memo;   // resident stage (RESW != 0): segments of the batch; byte offset / uint4 count of the record block copied to LDS; 1 = the instance memo has LDS
};
// RESW != 0: the RESIDENT form of the fused first-bounce kernel for scenes too large to stage whole (Domino: 64 KB of TLAS nodes + 128 KB of
// instance records).  ONE block of RESW waves per CU stays for the whole launch; it copies the TLAS nodes (as four planes, see kResNodes) and
// the block of small records (groups, shapes, emitters, triangles, shading data) into LDS once, then every wave takes 512-lane segments from a
// global counter until none is left -- a wave owns its segment exactly as a one-wave block does, there is no barrier after the stage.  What
// still comes through the vector L1 is the 128-byte instance record of a leaf visit and the queue traffic: the unstaged kernel keeps the CU's
// vector memory path busy 75 - 85 % of the time (TA / TD busy counters, profiles/r03_pmc_domino_fused*.txt) with the four 16-byte node loads
// per step per lane, and waits for it.
template <bool LDS, int MODE, bool AREA, int KMAX, bool MESH, int SPEC, int RESW = 0>   // SPEC: 0 diffuse-only scenes, 1 every BSDF / emitter / texture, 2 = 1 + blendbsdf (the BSDF chain in a loop over two records)
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
__global__ __launch_bounds__(RESW ? RESW * 64 : kShadeBlock, RESW ? RESW / 4 : (MODE == 2 && !MESH && !SPEC && KMAX == 1) ? 4 : (MODE != 0 && MESH) ? ((SPEC && KMAX > 1) ? 2 : DTOF_MESH_WAVES) : 1) void k_shade(ShadeArgs args_by_value) {
    constexpr bool FUSED = MODE != 0, FIRST = MODE == 2;
    static_assert(RESW == 0 || (MODE == 2 && !LDS && MESH), "the resident stage exists for the unstaged fused first-bounce kernel with mesh code");
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
    const uint32_t memo_words = FUSED ? (RESW ? (A0.res_memo ? RESW * kMemoWords * kMemoStride : 0u) : kMemoWords * kMemoStride) : 0u;
    uint32_t *stack = (uint32_t *) (lds + stage_words) + memo_words + threadIdx.x;
    // One block per 512-lane segment -- or, for a small frame whose whole path runs inline (rp.chunk_blocks = 8: nothing is compacted for a
    // later launch), one block per 64-lane chunk, so that a 1 M-lane frame is 16 384 waves instead of 2 048; the per-segment statistics are
    // then accumulated with atomics into slots the host has zeroed.
    const uint32_t sub = FIRST && !RESW ? A0.rp.chunk_blocks : 1u;        // blocks per segment: 1 or kSeg / kShadeBlock
    SceneView sv_res;
    if (RESW) {   // the resident stage: every thread of the block copies, ONE barrier, then the waves go their own ways
        const BlobHeader *gh = (const BlobHeader *) A0.scene;
        const uint4 *gn = (const uint4 *) (A0.scene + gh->off_nodes);
        const uint32_t n_pieces = gh->n_nodes * 4u;
        for (uint32_t i = threadIdx.x; i < n_pieces; i += blockDim.x) lds[(i & 3u) * kResNodes + (i >> 2)] = gn[i];
        const uint4 *gs = (const uint4 *) (A0.scene + A0.res_small_off);
        for (uint32_t i = threadIdx.x; i < A0.res_small_words; i += blockDim.x) lds[4u * kResNodes + i] = gs[i];
        __syncthreads();
        const uint8_t *small = (const uint8_t *) (lds + 4u * kResNodes) - A0.res_small_off;   // blob offsets of the copied block resolve into LDS
        sv_res = make_view(A0.scene);
        sv_res.nodes = (const DNode *) lds;
        sv_res.groups = (const DGroup *) (small + gh->off_groups); sv_res.shapes = (const DShape *) (small + gh->off_shapes);
        sv_res.tris = (const DTri *) (small + gh->off_tris); sv_res.shading = (const DTriShade *) (small + gh->off_shading);
        sv_res.emitters = (const DEmitter *) (small + gh->off_emitters);
    }
    for (uint32_t seg_first = 1;; seg_first = 0) {   // resident: until the segment counter runs out; otherwise once
    uint32_t seg, sub_index = 0;
    if (RESW) {
        uint32_t taken = 0;
        if (lane_id == 0) taken = atomicAdd(A0.q.seg_counter, 1u);
        seg = (uint32_t) __builtin_amdgcn_readfirstlane((int) taken);
        if (seg >= A0.n_seg) break;
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
    if (FUSED) { sv.memo_obj = A0.rp.memo_obj; sv.memo = (float *) (lds + stage_words) + (RESW ? wave_id * kMemoWords * kMemoStride + lane_id : threadIdx.x); }
    const bool have_memo = FUSED && sv.memo_obj != 0xffffffffu;
    for (uint32_t cbase = sub > 1 ? sub_index * kShadeBlock : 0u; cbase < (sub > 1 ? (sub_index + 1) * kShadeBlock < count ? (sub_index + 1) * kShadeBlock : count : count); cbase += kShadeBlock) {
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
#if DTOF_COOP
    float4 sha = make_float4(0.f, 0.f, 0.f, 0.f), shb = make_float4(0.f, 0.f, 1.f, 0.f), nra = sha, nrb = shb; float3 cand[KMAX];
#else
    // every register of the path state starts defined: a build of these kernels whose K = 4 every-BSDF instantiations ran at three waves per SIMD produced films that
    // depended on the initial value of `main` / `path` below (wrong with the registers' garbage, NaN with -ftrivial-auto-var-init=pattern, right with =zero;
    // profiles/r03_k4_uninitialised.txt) although no source path reads them before they are assigned
    float4 sha = make_float4(0.f, 0.f, 0.f, 0.f), shb = sha, nra = sha, nrb = sha; float3 cand[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) cand[k] = make_float3(0.f, 0.f, 0.f);
#endif
    float3 rbase[KMAX];   // FIRST: the result a lane ends this launch with if its NEE candidate is not committed
#pragma unroll
    for (int k = 0; k < KMAX; ++k) rbase[k] = make_float3(0.f, 0.f, 0.f);
    // Path state of the lane.  MODE 2 runs rp.inline_iters iterations of the bounce loop right here ("megakernel" head): between them the
    // state stays in these registers instead of making the round trip through the queues in HBM; after the last one the survivors are
    // written out and compacted exactly as before, for the bounce kernels (MODE 1) to continue with.
#if DTOF_COOP
    uint32_t hid = 0xffffffffu; float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = make_float4(0.f, 0.f, 1.f, 0.f), st; uint4 hh; Rng main, path;
#else
    uint32_t hid = 0xffffffffu; float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = make_float4(0.f, 0.f, 1.f, 0.f), st = ra; uint4 hh = make_uint4(0u, 0u, 0u, 0u); Rng main = { 0ull, 1ull }, path = { 0ull, 1ull };
#endif
    float4 stb_reg = make_float4(0.f, 0.f, 0.f, 1.f); float2 stc_reg = make_float2(1.f, 1.f);   // prev_si.p | prev_bsdf_pdf, eta | prev_bsdf_delta
    float memo_m[12], memo_inv[12];   // instance memo: the one instance's matrix and inverse at this lane's ray time
    bool lane_on = in_range;
    if (in_range) {
        l = qin ? qin[seg * kSeg + j] : seg * kSeg + j;
        if (FIRST) {
            // the wave's 64 lanes are the 64-aligned lanes [lane_base + seg * 512 + cbase, + 64): samples of one pixel if spp is a multiple of 64
            // ... AND the whole wave is in range: the pair swap of the correlated seeding (generate_lane) reads the partner lane, which a ragged tail
            // (dtof_sample_lanes with an odd count) would leave inactive
            const bool wave_pixel = rp.spp_log2 != 0xffffffffu && rp.spp_log2 >= 6 && (rp.lane_base & 63u) == 0 && count - cbase >= (uint32_t) kShadeBlock;
            const PrimaryLane pl = generate_lane(rp, global_lane(rp, rp.lane_base + l), wave_pixel, rp.lane_base + l);
            ra = pl.ray_a; rb = pl.ray_b; main = pl.main; path = pl.path; st = make_float4(1.f, 1.f, 1.f, 0.f);
            q.pos[l] = pl.pos;
            q.rng_b[l] = make_uint2((uint32_t) (main.inc >> 1), (uint32_t) (path.inc >> 1));
            if (dbg) {
                LaneDebug &o = dbg[l];
                o.time = ra.w; o.ray_o[0] = ra.x; o.ray_o[1] = ra.y; o.ray_o[2] = ra.z; o.ray_d[0] = rb.x; o.ray_d[1] = rb.y; o.ray_d[2] = rb.z;
            }
#if DTOF_COOP   // the shared triangle loops need wave-uniform call sites (dtof_traverse.h: trace_rays); measured slower, not the default
            if (have_memo) instance_memo_fill(sv, ra.w, memo_m, memo_inv);
        }
    }
    if (FIRST) {   // the primary rays: a wave-uniform call (lanes past the end of the segment help with the shared triangle loops, trace_rays)
        Hit h; bool found = false;
        if (flat) { if (in_range) found = trace_flat<false, true>(sv, (ConstBytes) A.scene + rp.flat_off, flat, stack, mk(ra.x, ra.y, ra.z), mk(rb.x, rb.y, rb.z), ra.w, rb.w, h); }
        else found = trace_rays<false, MESH, FUSED, RESW != 0>(sv, stack, in_range, mk(ra.x, ra.y, ra.z), mk(rb.x, rb.y, rb.z), ra.w, rb.w, h);
        if (in_range) {
            hh = make_uint4(f2u(h.t), f2u(h.u), f2u(h.v), h.prim);
            hid = found ? (h.obj | (h.shape << q.id_shift)) : 0xffffffffu;
        }
    }
#else
            if (have_memo) instance_memo_fill(sv, ra.w, memo_m, memo_inv);
            Hit h;
            bool found = flat ? trace_flat<false, true>(sv, (ConstBytes) A.scene + rp.flat_off, flat, stack, mk(ra.x, ra.y, ra.z), mk(rb.x, rb.y, rb.z), ra.w, rb.w, h)
                              : trace_scene<false, MESH, FUSED, RESW != 0>(sv, stack, mk(ra.x, ra.y, ra.z), mk(rb.x, rb.y, rb.z), ra.w, rb.w, h);
            hh = make_uint4(f2u(h.t), f2u(h.u), f2u(h.v), h.prim);
            hid = found ? (h.obj | (h.shape << q.id_shift)) : 0xffffffffu;
        }
    }
#endif
    const uint32_t n_inline = FIRST ? rp.inline_iters : 1u;
    for (uint32_t it = 0; ; ++it) {
    const uint32_t depth = depth0 + it;
    const bool last = it + 1 >= n_inline;                 // uniform
    const uint32_t trace_next = last ? trace_next_last : 1u;
    alive = false; want_shadow = false;
    if (lane_on) {
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
            for (int k = 0; k < 6; ++k) { main.state = main.state * kPcgMult + main.inc; if (!one) path.state = path.state * kPcgMult + path.inc; }
            q.rng_a[l] = make_uint4((uint32_t) main.state, (uint32_t) (main.state >> 32), (uint32_t) path.state, (uint32_t) (path.state >> 32));
        }
        if (SPEC && hid == 0xffffffffu && rp.has_env && !(depth == 0 && rp.hide_emitters)) {
            // The ray left the scene: si.emitter(scene) is the environment (dopplertofpath.cpp:150-168).  DirectionSample(scene, si, prev_si)
            // points along the ray; ConstantBackgroundEmitter::pdf_direction is the uniform-sphere density (constant.cpp:150-155).  A primary
            // ray that sees the environment directly only counts if emitters are not hidden (valid_ray, :101-102,279-282).
            const float4 stv = FIRST ? st : q.st_a[l];
            const float time_ = FIRST ? ra.w : q.ray_a[l].w;
            float prev_pdf = 1.f; bool pdelta = true;
            if (depth > 0) { prev_pdf = FIRST ? stb_reg.w : q.st_b[l].w; pdelta = (FIRST ? stc_reg.y : q.st_c[l].y) != 0.f; }
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
                const float4 r = FIRST ? make_float4(rbase[k].x, rbase[k].y, rbase[k].z, 0.f) : q.res[(size_t) k * q.capacity + l];
                const float4 acc = make_float4(fmaf(stv.x, v.x, r.x), fmaf(stv.y, v.y, r.y), fmaf(stv.z, v.z, r.z), 0.f);
                if (FIRST) rbase[k] = make_float3(acc.x, acc.y, acc.z); else q.res[(size_t) k * q.capacity + l] = acc;
            }
        }
        if (hid != 0xffffffffu) {   // a miss ends the path (active_next = false, dopplertofpath.cpp:171)
            V3 o = mk(ra.x, ra.y, ra.z), d = mk(rb.x, rb.y, rb.z); float time = ra.w;
            V3 thr = mk(st.x, st.y, st.z); float path_length = st.w;
            float eta_path = 1.f; bool prev_delta = depth == 0;   // dopplertofpath.cpp:103-108: eta = 1, prev_bsdf_delta = true
            if (SPEC && depth > 0) { const float2 sc = FIRST ? stc_reg : q.st_c[l]; eta_path = sc.x; prev_delta = sc.y != 0.f; }
            bool correlate = (depth + 1) < rp.path_correlation_depth;
            const bool plain = rp.integrator != 0;   // `path`: no modulation weight
            const bool single = plain || rp.sampler_kind != SAMPLER_CORRELATED;   // main stream only (path.cpp:197,213-214,273; sampler.h:141-144)
            float t = u2f(hh.x);
            path_length += t * eta_path;   // dopplertofpath.cpp:141 (eta stays 1 without dielectrics)
            bool active_next = depth + 1 < rp.max_depth;

            Surface si;
            if (!FIRST && have_memo) instance_memo_fill(sv, time, memo_m, memo_inv);
            if (FIRST && it > 0 && have_memo) {   // a ray's time does not change along its path: the inverse filled at generation still sits in the LDS column
                instance_matrix(sv.objects[sv.memo_obj], time, memo_m); instance_memo_load(sv, memo_inv);
            }
            compute_surface<MESH>(sv, hid & ((1u << q.id_shift) - 1u), hid >> q.id_shift, hh.w, t, u2f(hh.y), u2f(hh.z), o, d, time, si, have_memo, memo_m, memo_inv);
            const DShape *sh = si.shape;

            const float pmf = sv.n_emitters ? 1.f / (float) sv.n_emitters : 0.f;   // m_emitter_pmf (scene.cpp:96)
            // ---- direct emission (dopplertofpath.cpp:150-168 / path.cpp): the hit shape carries an area emitter
            bool res_dirty = false;
            float4 rcur[KMAX];
            if (AREA) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) rcur[k] = FIRST ? make_float4(rbase[k].x, rbase[k].y, rbase[k].z, 0.f) : q.res[(size_t) k * q.capacity + l];
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
                        rcur[k] = make_float4(fmaf(thr.x, v.x, rcur[k].x), fmaf(thr.y, v.y, rcur[k].y), fmaf(thr.z, v.z, rcur[k].z), 0.f);
                    }
                    res_dirty = true;
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
                if (em.kind == EMITTER_POINT) {
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
            float bsdf_pdf = 0.f, bs_pdf = 0.f, bs_eta = 0.f; bool bs_delta = false;
            V3 val_0 = mk(0, 0, 0), keep_weight = mk(0, 0, 0), keep_wo = mk(0, 0, 0); float pdf_0 = 0.f, keep_pdf = 0.f, keep_eta = 0.f; bool keep_delta = false;
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
                bsdf_pdf = 0.f; bs_pdf = 0.f; bs_eta = 0.f; bs_delta = false;
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
                } else if (SPEC && bsh->bsdf == BSDF_THINDIELECTRIC) {
                    // ThinDielectric::sample (thindielectric.cpp:173-226): the reflectance of the slab with all internal bounces, wo = -wi
                    float r, t1, t2, t3;
                    fresnel_dielectric(fabsf(si.wi.z), bsh->diel_eta, r, t1, t2, t3);
                    r *= 2.f / (1.f + r);
                    const bool selected_r = s1 <= r;
                    bs_pdf = selected_r ? r : 1.f - r; bs_delta = true; bs_eta = 1.f;
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
                    if ((pass == 1) == pick_1) { keep_weight = bsdf_weight; keep_wo = bs_wo; keep_pdf = bs_pdf; keep_eta = bs_eta; keep_delta = bs_delta; }
                    if (pass == 0) { val_0 = bsdf_val; pdf_0 = bsdf_pdf; }
                    else {
                        const float w0 = 1.f - blend_w;
                        bsdf_val = mk(val_0.x * w0 + bsdf_val.x * blend_w, val_0.y * w0 + bsdf_val.y * blend_w, val_0.z * w0 + bsdf_val.z * blend_w);
                        bsdf_pdf = pdf_0 * w0 + bsdf_pdf * blend_w;
                        bsdf_weight = keep_weight; bs_wo = keep_wo; bs_pdf = keep_pdf; bs_eta = keep_eta; bs_delta = keep_delta;
                    }
                }
            }
            si.wi = wi_plain;
            if (masked) {
                bsdf_val = bsdf_val * opacity; bsdf_pdf *= opacity;
                if (null_pick) { bs_wo = mk(-si.wi.x, -si.wi.y, -si.wi.z); bs_eta = 1.f; bs_pdf = 1.f - opacity; bs_delta = true; bsdf_weight = mk(1.f, 1.f, 1.f); }
            }
            // ---- emitter contribution candidate (dopplertofpath.cpp:214-226); committed by k_shadow if unoccluded
            if (active_em) {
                const float mis_em = ds_delta ? 1.f : mis_weight(ds_pdf, bsdf_pdf);   // dopplertofpath.cpp:218-219
                bool nonzero = false;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) {
                    float4 r = AREA ? rcur[k] : (FIRST ? make_float4(rbase[k].x, rbase[k].y, rbase[k].z, 0.f) : q.res[(size_t) k * q.capacity + l]);
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
            bool rr_continue = next_f32(sel) < rr_prob;
            if (use_path) { path.state = sel.state; main.state = pcg_jump6(main.state, main.inc); }
            else { main.state = sel.state; if (!single) path.state = pcg_jump6(path.state, path.inc); }
            if (rr_active) thr = thr * rcp(rr_prob);
            alive = active_next && (!rr_active || rr_continue) && thr_max != 0.f;
            if (alive) {
                nra = make_float4(no.x, no.y, no.z, time); nrb = make_float4(nd.x, nd.y, nd.z, kLargest);
                const float4 sta = make_float4(thr.x, thr.y, thr.z, path_length), stb = make_float4(si.p.x, si.p.y, si.p.z, bs_pdf);
                const float2 stc = make_float2(eta, bs_delta ? 1.f : 0.f);
                if ((!FIRST || last) && trace_next) {   // the state leaves for the queues (an inline iteration keeps it in registers; after the last iteration of the loop nobody reads it)
                    q.ray_a[l] = nra;
                    q.ray_b[l] = nrb;
                    q.st_a[l] = sta;
                    if (AREA) q.st_b[l] = stb;   // prev_si, prev_bsdf_pdf (:256-257)
                    if (SPEC) q.st_c[l] = stc;   // eta, prev_bsdf_delta (:252,258)
                }
                if (FIRST) { st = sta; stb_reg = stb; stc_reg = stc; }
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
#if DTOF_COOP
        bool commit = false;
        {   // test_visibility (scene.cpp:266-271): an unoccluded sample commits its candidate result
            Hit hs;
#if defined(DTOF_ABLATE) && (DTOF_ABLATE & 1)
            commit = want_shadow && sha.w > 0.f;
#else
            if (flat) { if (want_shadow) commit = !trace_flat<true, true>(sv, (ConstBytes) A.scene + rp.flat_off, flat, stack, mk(sha.x, sha.y, sha.z), mk(shb.x, shb.y, shb.z), shb.w, sha.w, hs); }
            else if (__ballot(want_shadow)) commit = !trace_rays<true, MESH, true, RESW != 0>(sv, stack, want_shadow, mk(sha.x, sha.y, sha.z), mk(shb.x, shb.y, shb.z), shb.w, sha.w, hs) && want_shadow;
#endif
        }
#else
        bool commit = false;
        if (want_shadow) {   // test_visibility (scene.cpp:266-271): an unoccluded sample commits its candidate result
            Hit hs;
#if defined(DTOF_ABLATE) && (DTOF_ABLATE & 1)
            commit = sha.w > 0.f;
#else
            commit = flat ? !trace_flat<true, true>(sv, (ConstBytes) A.scene + rp.flat_off, flat, stack, mk(sha.x, sha.y, sha.z), mk(shb.x, shb.y, shb.z), shb.w, sha.w, hs)
                          : !trace_scene<true, MESH, true, RESW != 0>(sv, stack, mk(sha.x, sha.y, sha.z), mk(shb.x, shb.y, shb.z), shb.w, sha.w, hs);
#endif
        }
#endif
        if (FIRST) {   // the running result stays in rbase over the inline iterations; every lane's result is defined after the last (nothing zeroed it)
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) {
                if (commit) rbase[k] = cand[k];
                if (last && in_range) q.res[(size_t) k * q.capacity + l] = make_float4(rbase[k].x, rbase[k].y, rbase[k].z, 0.f);
            }
        } else if (commit) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets) q.res[(size_t) k * q.capacity + l] = make_float4(cand[k].x, cand[k].y, cand[k].z, 0.f);
        }
#if DTOF_COOP
        const bool trace_now = alive && trace_next;
        if (__ballot(trace_now)) {   // closest hit of the continuation ray, consumed by the next bounce (wave-uniform call, see trace_rays)
            Hit h; bool found = false;
#if defined(DTOF_ABLATE) && (DTOF_ABLATE & 2)
            found = nra.x < 1e30f; h.t = 0.5f + 0.1f * nrb.x; h.u = nrb.y; h.v = nrb.z; h.obj = nrb.x > 0.3f ? 3 : nrb.y > 0.f ? 1 : 0; h.shape = 0; h.prim = 0;
#else
            if (flat) { if (trace_now) found = trace_flat<false, true>(sv, (ConstBytes) A.scene + rp.flat_off, flat, stack, mk(nra.x, nra.y, nra.z), mk(nrb.x, nrb.y, nrb.z), nra.w, nrb.w, h); }
            else found = trace_rays<false, MESH, true, RESW != 0>(sv, stack, trace_now, mk(nra.x, nra.y, nra.z), mk(nrb.x, nrb.y, nrb.z), nra.w, nrb.w, h);
#endif
            if (trace_now) {
                if (!FIRST || last) store_hit<MESH>(q, l, h, found);
                if (FIRST) { hh = make_uint4(f2u(h.t), f2u(h.u), f2u(h.v), h.prim); hid = found ? (h.obj | (h.shape << q.id_shift)) : 0xffffffffu; }
            }
        }
#else
        if (alive && trace_next) {   // closest hit of the continuation ray, consumed by the next bounce
            Hit h;
#if defined(DTOF_ABLATE) && (DTOF_ABLATE & 2)
            bool found = nra.x < 1e30f; h.t = 0.5f + 0.1f * nrb.x; h.u = nrb.y; h.v = nrb.z; h.obj = nrb.x > 0.3f ? 3 : nrb.y > 0.f ? 1 : 0; h.shape = 0; h.prim = 0;
#else
            bool found = flat ? trace_flat<false, true>(sv, (ConstBytes) A.scene + rp.flat_off, flat, stack, mk(nra.x, nra.y, nra.z), mk(nrb.x, nrb.y, nrb.z), nra.w, nrb.w, h)
                              : trace_scene<false, MESH, true, RESW != 0>(sv, stack, mk(nra.x, nra.y, nra.z), mk(nrb.x, nrb.y, nrb.z), nra.w, nrb.w, h);
#endif
            if (!FIRST || last) store_hit<MESH>(q, l, h, found);
            if (FIRST) { hh = make_uint4(f2u(h.t), f2u(h.u), f2u(h.v), h.prim); hid = found ? (h.obj | (h.shape << q.id_shift)) : 0xffffffffu; }
        }
#endif
        const uint32_t n_sh = (uint32_t) __popcll(__ballot(want_shadow)) * ((threadIdx.x & 63) == 0 ? 1u : 0u);   // per-wave partial (stats only)
        if (last) n_shadow += n_sh; else if (lane_id == 0) s_inline[2 * it + 1] += n_sh;
    } else {
        uint32_t sslot = seg * kSeg + block_append(want_shadow, s_cnt, n_shadow);
        if (want_shadow) {
            q.sh_a[sslot] = sha; q.sh_b[sslot] = shb;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (KMAX == 1 || k < rp.n_offsets)
                q.sh_c[(size_t) k * q.capacity + sslot] = make_float4(cand[k].x, cand[k].y, cand[k].z, u2f(l));
        }
    }
    if (last) break;
    // next inline iteration: the continuation ray and its closest hit become the lane's current ray (st / stb / stc were set where the
    // bounce computed them); a lane whose path ended sits out the remaining iterations
    lane_on = alive;
    if (alive) { ra = nra; rb = nrb; }
    }   // inline iterations
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
    const bool active = j < count;
    uint32_t i = seg * kSeg + j;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 1.f, 0.f);
    if (active) { a = q.sh_a[i]; b = q.sh_b[i]; }
    Hit h;
    bool occluded = trace_rays<true, MESH>(sv, stack, active, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), b.w, a.w, h);
    if (active && !occluded) {
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
}

// ---------------------------------------------------------------------------- splat
DTOF_D float tent(float x, float inv_r) { return fmax_(0.f, 1.f - fabsf(x * inv_r)); }
// ReconstructionFilter::eval: tent (tent.cpp:53-55) or gaussian (gaussian.cpp:94-96, polynomial branch)
template <int F = -1>   // F >= 0: the filter is known at compile time (the branches fold away)
DTOF_D float filter_weight(const RenderParams &rp, float x) {
    const int filter = F >= 0 ? F : (int) rp.filter;
    if (filter == FILTER_GAUSSIAN) return fmax_(estrin10(sqr(x), rp.gauss_coeff), 0.f);
    if (filter == FILTER_MITCHELL) {   // MitchellNetravaliFilter::eval (mitchell.cpp:47-67): coefficients in ScalarFloat, Horner with fmadd
        x = fabsf(x);
        const float x2 = x * x, x3 = x2 * x, B = rp.filter_b, C = rp.filter_c;
        const float a3 = (12.f - 9.f * B - 6.f * C), a2 = (-18.f + 12.f * B + 6.f * C), a0 = (6.f - 2.f * B),
                    b3 = (-B - 6.f * C), b2 = (6.f * B + 30.f * C), b1 = (-12.f * B - 48.f * C), b0 = (8.f * B + 24.f * C);
        const float r = (1.f / 6.f) * (x < 1.f ? fmaf(a3, x3, fmaf(a2, x2, a0)) : fmaf(b3, x3, fmaf(b2, x2, fmaf(b1, x, b0))));
        return x < 2.f ? r : 0.f;
    }
    if (filter == FILTER_CATMULLROM) {   // CatmullRomFilter::eval (catmullrom.cpp:38-53): B = 0, C = 1/2, plain multiplies and adds
        x = fabsf(x);
        const float x2 = x * x, x3 = x2 * x, B = 0.f, C = .5f;
        const float r = (1.f / 6.f) * (x < 1.f ? (12.f - 9.f * B - 6.f * C) * x3 + (-18.f + 12.f * B + 6.f * C) * x2 + (6.f - 2.f * B)
                                               : (-B - 6.f * C) * x3 + (6.f * B + 30.f * C) * x2 + (-12.f * B - 48.f * C) * x + (8.f * B + 24.f * C));
        return x < 2.f ? r : 0.f;
    }
    if (filter == FILTER_LANCZOS) {   // LanczosSincFilter::eval (lanczos.cpp:52-63): radius = lobes
        x = fabsf(x);
        const float x1 = kPi * x, x2 = x1 / rp.filter_radius;
        float s1, s2, c; sincos_(x1, s1, c); sincos_(x2, s2, c);
        const float result = (s1 * s2) / (x1 * x2);
        return x < 5.9604644775390625e-8f ? 1.f : (x > rp.filter_radius ? 0.f : result);
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
// Whole-blob staging pays while the blob is small against what a block's lanes read from it: measured on Domino fields (split pipeline,
// 1024^2 x 32 spp, one box): 4 KB blob staged 3.84 ms vs 4.28 ms unstaged, 9 KB 4.63 vs 5.16, 30 KB 10.39 vs 7.90 (every 64-lane shade
// block and every 256-lane trace block copies the blob; the L1 hit rate of the unstaged kernels is 99 % on such scenes).
constexpr uint32_t kLdsSceneLimit = 16 * 1024;
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
                  uint32_t stack_depth, hipStream_t s, bool first, LaneDebug *dbg, const ResidentStage *resident) {
    if (rp.n_lanes == 0) return;
    static_assert(kResidentNodes == kResNodes, "resident stage size");
    if (resident && resident->waves && first && fused && rp.has_tris && rp.chunk_blocks <= 1) {
        // one block of `waves` waves per CU; LDS = node planes + record block + (instance memo) + stack columns, well above the 64 KiB default limit
        const uint32_t waves = resident->waves, block = waves * 64;
        const uint32_t memo = rp.memo_obj != 0xffffffffu ? 1u : 0u;
        const uint32_t lds = (4u * kResNodes + resident->small_words) * 16u + memo * waves * kMemoWords * kMemoStride * 4u + stack_bytes(stack_depth, block);
        if (lds + 1024 > 160 * 1024) throw std::runtime_error("resident stage: the scene does not fit the 160 KiB of LDS");
        static const int n_cu = [] { int dev = 0, n = 0; (void) hipGetDevice(&dev); (void) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        const uint32_t n_seg = nseg(rp.n_lanes), grid = std::min<uint32_t>((uint32_t) n_cu, (n_seg + waves - 1) / waves);
        const ShadeArgs sa = { scene, scene_bytes, 0u, rp, q, qin, count_in, qout, alive_out, shadow_out, depth, trace_next ? 1u : 0u, dbg, n_seg, resident->small_off, resident->small_words, memo };
        (void) hipMemsetAsync(q.seg_counter, 0, 4, s);
#define DTOF_LAUNCH_RES(A, K, S, W) do { static uint32_t attr_lds = 0;   /* the kernel's static LDS (count slots) comes on top of the dynamic size */ \
            if (lds > attr_lds) { if (hipFuncSetAttribute((const void *) k_shade<false, 2, A, K, true, S, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); attr_lds = lds; } \
            hipLaunchKernelGGL((k_shade<false, 2, A, K, true, S, W>), dim3(grid), dim3(block), lds, s, sa); } while (0)
#define DTOF_RES_AKS(W) do { if (rp.has_spec == 2) { if (rp.n_offsets == 1) DTOF_LAUNCH_RES(true, 1, 2, W); else DTOF_LAUNCH_RES(true, kMaxOffsets, 2, W); } \
                             else if (rp.has_spec) { if (rp.n_offsets == 1) DTOF_LAUNCH_RES(true, 1, 1, W); else DTOF_LAUNCH_RES(true, kMaxOffsets, 1, W); } \
                             else if (rp.has_area) { if (rp.n_offsets == 1) DTOF_LAUNCH_RES(true, 1, false, W); else DTOF_LAUNCH_RES(true, kMaxOffsets, false, W); } \
                             else { if (rp.n_offsets == 1) DTOF_LAUNCH_RES(false, 1, false, W); else DTOF_LAUNCH_RES(false, kMaxOffsets, false, W); } } while (0)
        if (waves == 12) DTOF_RES_AKS(12); else if (waves == 8) DTOF_RES_AKS(8); else if (waves == 16) DTOF_RES_AKS(16); else throw std::runtime_error("resident stage: 8, 12 or 16 waves per block");
#undef DTOF_RES_AKS
#undef DTOF_LAUNCH_RES
        return;
    }
    const uint32_t shade_stack = fused ? stack_bytes(stack_depth, kShadeBlock) + kMemoWords * kMemoStride * 4 : 0;   // + the instance memo
    uint32_t sw = stage_words_for(scene_bytes, shade_stack), grid = nseg(rp.n_lanes) * (first && rp.chunk_blocks > 1 ? rp.chunk_blocks : 1u), lds = sw * 16 + shade_stack;
    check_lds(lds);
    uint32_t tn = trace_next ? 1u : 0u;
#define DTOF_LAUNCH_SHADE(L, F, A, K) do { if (rp.has_tris) DTOF_LAUNCH_SHADE_M(L, F, A, K, true, false); else DTOF_LAUNCH_SHADE_M(L, F, A, K, false, false); } while (0)
    const ShadeArgs sa = { scene, scene_bytes, sw, rp, q, qin, count_in, qout, alive_out, shadow_out, depth, tn, dbg, nseg(rp.n_lanes), 0u, 0u, 0u };
#define DTOF_LAUNCH_SHADE_M(L, F, A, K, M, S) hipLaunchKernelGGL((k_shade<L, F, A, K, M, S>), dim3(grid), dim3(kShadeBlock), lds, s, sa)
#define DTOF_SHADE_AK(L, F) do { if (rp.has_spec == 2) { if (rp.n_offsets == 1) DTOF_LAUNCH_SHADE_M(L, F, true, 1, true, 2); else DTOF_LAUNCH_SHADE_M(L, F, true, kMaxOffsets, true, 2); } \
                                 else if (rp.has_spec) { if (rp.n_offsets == 1) DTOF_LAUNCH_SHADE_M(L, F, true, 1, true, 1); else DTOF_LAUNCH_SHADE_M(L, F, true, kMaxOffsets, true, 1); } \
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
bool read_traversal_stats(unsigned long long *out8) {
    unsigned long long zero[8] = { 0 };
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_trav_stats), 64) == hipSuccess && hipMemcpyToSymbol(HIP_SYMBOL(g_trav_stats), zero, 64) == hipSuccess;
}
#endif

}  // namespace dtof
