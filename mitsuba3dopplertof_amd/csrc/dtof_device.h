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

static inline uint32_t nblk(uint32_t n) { return (n + kBlock - 1) / kBlock; }
static inline uint32_t nseg(uint32_t n) { return (n + kSeg - 1) / kSeg; }
static inline uint32_t stack_bytes(uint32_t depth, uint32_t block = kBlock) { return (depth < 2 ? 2 : depth) * block * 4; }

}  // namespace dtof
