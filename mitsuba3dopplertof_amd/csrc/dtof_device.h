// dtof_device.h -- what every translation unit with kernels includes: the device function headers (scene view and traversal, samplers and
// waveforms, surface interaction and BSDF helpers) and the few helpers the kernels and their launchers share (queue segments, hit records,
// LDS budgeting).  The kernels themselves live in dtof_kernels.hip (generate, trace, shadow, velocity, splat, develop, known-answer kernels)
// and dtof_shade.h (k_shade, instantiated by the dtof_shade_*.hip files -- one group of instantiations per file so that they compile in parallel).
#pragma once
#include "dtof_kernels.h"
#include "dtof_scene.h"
#include "dtof_math.h"

#define DTOF_D __device__ __forceinline__
namespace dtof { constexpr int kBlock = 256; }
#include "dtof_traverse.h"
#include "dtof_sampling.h"
#include "dtof_shading.h"

#include <stdexcept>
#include <cstdlib>

namespace dtof {

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

// ---------------------------------------------------------------------------- reconstruction filter and per-lane splat (shared by the splat kernels and k_shade)
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

// ---------------------------------------------------------------------------- wave-wide sums of MANY values (the fused splat of k_shade)
// xor_lane<S>(v): v of lane (id ^ (1 << S)).  S = 0, 1: one DPP quad permutation; 2, 3: two DPP moves (xor 4 = reverse the 8 lanes of a half row, then the 4 of
// each quad; xor 8 = reverse the 16 lanes of a row, then each half); 4: ds_swizzle in bit mode; 5: ds_bpermute.
template <int S> DTOF_D float xor_lane(float v) {
    const int x = __float_as_int(v);
    if (S == 0) return __int_as_float(__builtin_amdgcn_mov_dpp(x, 0xb1, 0xf, 0xf, false));
    if (S == 1) return __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x4e, 0xf, 0xf, false));
    if (S == 2) return __int_as_float(__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x141, 0xf, 0xf, false), 0x1b, 0xf, 0xf, false));
    if (S == 3) return __int_as_float(__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x140, 0xf, 0xf, false), 0x141, 0xf, 0xf, false));
    if (S == 4) return __int_as_float(__builtin_amdgcn_ds_swizzle(x, 0x401f));
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int) ((__lane_id() ^ 32u) << 2), x));   // (4 and 5: not used by wave_totals_36, which swaps)
}
// One stage of the butterfly: lanes pair up across bit S; of every two neighbouring values the lane whose bit is 0 keeps the first, its partner the second, each adding
// what the other holds of it.  N values become (N + 1) / 2.  Across bit 5 and bit 4 gfx950's v_permlane32_swap / v_permlane16_swap do the exchange AND the selection
// in one instruction (swap(a, b) = { [a.lo, b.lo], [a.hi, b.hi] }: their sum is a's total in the lower half and b's in the upper; tools/ubench/permlane_swap.hip):
// two instructions per result, against two selects and a DPP add elsewhere -- so the two stages with the most values come first.
template <int N, int S> DTOF_D void butterfly_stage(float *v, uint32_t lane) {
    const bool bit = (lane >> S) & 1u;
#pragma unroll
    for (int j = 0; j < (N + 1) / 2; ++j) {
        const float a = v[2 * j], b = 2 * j + 1 < N ? v[2 * j + 1] : 0.f;
        if (S == 5 || S == 4) {
            const auto r = S == 5 ? __builtin_amdgcn_permlane32_swap(f2u(a), f2u(b), false, false) : __builtin_amdgcn_permlane16_swap(f2u(a), f2u(b), false, false);
            v[j] = u2f(r[0]) + u2f(r[1]);
        } else v[j] = (bit ? b : a) + xor_lane<S>(bit ? a : b);
    }
}
// -> the lane whose bit-reversed id (6 bits) is L < 36 holds the sum of v[L] over the 64 lanes of the wave (all of them active): value bit k is decided by lane bit 5 - k
DTOF_D float wave_totals_36(float *v, uint32_t lane) {
    butterfly_stage<36, 5>(v, lane); butterfly_stage<18, 4>(v, lane); butterfly_stage<9, 3>(v, lane);
    butterfly_stage<5, 2>(v, lane); butterfly_stage<3, 1>(v, lane); butterfly_stage<2, 0>(v, lane);
    return v[0];
}

static inline uint32_t nblk(uint32_t n) { return (n + kBlock - 1) / kBlock; }
static inline uint32_t nseg(uint32_t n) { return (n + kSeg - 1) / kSeg; }
static inline uint32_t stack_bytes(uint32_t depth, uint32_t block = kBlock) { return (depth < 2 ? 2 : depth) * block * 4; }

}  // namespace dtof
