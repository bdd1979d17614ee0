// dtof_kernels.h -- kernel parameter blocks and queue layout shared by the host
// orchestration (dtof_render.hip) and the kernels (dtof_kernels.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "dtof_math.h"

namespace dtof {

constexpr uint32_t kChunkBlocks = 8; // 64-lane chunks of a 512-lane queue segment (RenderParams::chunk_blocks)
constexpr uint32_t kMaxInline = 4;  // iterations of the bounce loop the fused first-bounce kernel may run itself (RenderParams::inline_iters)
constexpr int kMaxOffsets = 4;      // modulation offsets evaluated per traversal (K)

// Everything a kernel needs besides the scene blob and the queues; passed by value.
struct RenderParams {
    // ---- camera (PerspectiveCamera, src/sensors/perspective.cpp:172-279)
    float s2c[16];                  // sample_to_camera
    float cam_to_world[12];
    float near_clip, far_clip, shutter_open, shutter_open_time;
    int32_t orthographic;                    // OrthographicCamera (src/sensors/orthographic.cpp:169-196)
    float aperture_radius, focus_distance;   // ThinLensCamera (src/sensors/thinlens.cpp:257-305); aperture_radius == 0: perspective
    // ---- film (lane -> pixel mapping src/render/integrator.cpp:273-290; splat imageblock.cpp:414-531)
    int32_t crop_x, crop_y, crop_w, crop_h;
    float scale_x, scale_y, offset_x, offset_y;   // render_sample: scale = 1/crop_size, offset = -crop_offset*scale
    int32_t filter; float filter_radius, inv_radius;
    float gauss_coeff[10];                        // GaussianFilter's Remez fit, scaled and shifted like gaussian.cpp:60-89
    float filter_b, filter_c;       // mitchell: B, C (src/rfilters/mitchell.cpp)
    // ---- sampler (src/samplers/correlated.cpp, src/render/sampler.cpp)
    uint32_t base_seed, seed, seed_value;         // seed_value = base_seed + seed
    uint32_t spp, spp_log2;                       // spp_log2 = 0xffffffff when spp is not a power of two
    uint32_t tcn, pcn;
    int32_t time_sampling; float antithetic_shift; int32_t stratify;
    uint32_t n_stratum; float inv_n_stratum, inv_tcn;
    // ---- integrator (src/integrators/dopplertofpath.cpp:19-77)
    float T, w_d, w_g, phi_coef, amp, g_1, g_0;
    float phase[kMaxOffsets]; int32_t n_offsets;
    int32_t wave_type, low_pass;
    uint32_t path_correlation_depth, max_depth, rr_depth;
    int32_t has_area;                             // scene has area emitters: emitter-hit term + prev_si / prev_bsdf_pdf state
    int32_t sampler_kind, jitter; float inv_spp;  // SamplerKind; timestratified: jitter, 1 / sample_count (timestratified.cpp:78-82)
    int32_t has_spec;                             // scene has delta BSDFs (conductor / dielectric): eta and prev_bsdf_delta become per-lane state; 2: ... and blendbsdf
    int32_t has_tris;                             // scene has triangle meshes (selects the kernel instantiations that carry mesh code)
    int32_t has_analytic;                         // the scene has spheres / disks / cylinders (the eight-wave ray kernels carry triangle and rectangle code only)
    uint32_t n_tlas_nodes;                        // nodes of the top-level BVH (the eight-wave ray kernels keep up to kTlasLds8 of them in LDS)
    uint32_t res_half;                            // resident stage: the LDS planes hold half-float node records (a TLAS of kResidentNodes + 1 .. 2 * kResidentNodes nodes)
    int32_t has_nodes16;                          // the blob carries the half-float copy of the node array (BlobHeader::off_nodes16)
    int32_t has_blas;                             // some mesh is traversed through its own BLAS: the unstaged k_trace / k_shadow run one wave per block
    int32_t integrator;                           // 0 dopplertofpath, 1 path (src/integrators/path.cpp), 2 velocity (velocity.cpp)
    // ---- batch
    uint32_t lane_base, n_lanes;                  // this batch covers (virtual) lanes [lane_base, lane_base + n_lanes)
    // striped shards (dtof_render_stripes): virtual row v of this shard is film row stripe_first + (v / stripe_rows) * stripe_period
    // + v % stripe_rows, and the GLOBAL lane index (what every RNG stream is a function of) follows from it.  stripe_rows == 0:
    // virtual = global (contiguous rows).
    uint32_t stripe_rows, stripe_period, stripe_first, lanes_per_row;
    // exact division by the launch-invariant divisors of the lane mappings (dtof_math.h: FastDiv)
    FastDiv d_spp, d_w, d_tcn, d_pcn, d_stratum, d_lanes_per_row, d_stripe_rows;
    // wavefronts of more than 2^32 - 1 lanes / samples_per_pass (integrator.cpp:121-124,227-245): pass `pass` of `n_passes`, each of
    // `spp` samples per pixel (= samples per wavefront); the sampler's streams are seeded in pass 0 and carried across the passes
    uint32_t pass, n_passes;
    uint32_t sample_count; FastDiv d_sample_count;   // Sampler::sample_count() of the whole render (spp = samples per wavefront = per pass)
    uint2 *pass_rng;                              // n_passes > 1: [lane - pass_first][3] = states of the main / time / path streams between passes
    uint32_t pass_first;                          // virtual lane the pass_rng array starts at
    int32_t has_env, hide_emitters; uint32_t env_index;   // `constant` environment emitter (scene.cpp:53-57), SamplingIntegrator::m_hide_emitters
    uint32_t chunk_blocks;                        // first-bounce kernel: blocks per 512-lane segment, 1 or 8 (small frames whose whole path runs inline)
    uint32_t res_units;                           // resident first-bounce kernel whose launch covers the whole path: work units a wave takes from the counter per 512-lane segment (1, 2, 4, 8)
    uint32_t inline_iters;                        // fused first-bounce kernel: iterations of the bounce loop it runs back to back with the path state in registers (1 .. kMaxInline)
    uint32_t flat_objects, flat_off;                        // fused pipeline, rectangle-only scenes of at most kFlatObjects objects: their number (trace_flat), else 0
    uint32_t memo_obj;                            // fused pipeline: the scene's only instance object (instance memo, dtof_traverse.h) or 0xffffffff
    int32_t want_valid;                           // the kernel that ends a path also writes its valid_ray flag to Queues::valid_out (alpha channel of an rgba film, lane dumps)
};

// SoA wavefront state for one batch (device pointers; all arrays have `capacity` entries and are
// indexed by the lane's position inside the batch).
struct Queues {
    float4 *ray_a;       // o.xyz, time
    float4 *ray_b;       // d.xyz, maxt
    uint4  *hit;         // t, u, v (float bits), prim
    float  *hit_t;       // rectangle-only scenes: t alone replaces `hit`
    uint32_t *hit_id;    // object (low id_shift bits) | shape-in-group (the bits above); 0xffffffff = miss
    float4 *st_a;        // throughput.xyz, path_length
    float4 *st_b;        // prev_si.p, prev_bsdf_pdf (only touched when the scene has area emitters)
    uint4  *rng_a;       // rng.state (lo,hi), rng_path.state (lo,hi)
    float2 *st_c;        // SPEC: (eta along the path, prev_bsdf_delta | valid_ray << 1 as a float 0 .. 3)
    float4 *valid_out;   // RenderParams::want_valid: (valid_ray ? 1 : 0, 0, 0, 0) of every finished path -- laid out like `res`, so the splat kernels accumulate the alpha film from it
    uint2  *rng_b;       // (main, path) stream selectors v1 of the TEA seeding: inc = (v1 << 1) | 1, constant per lane
    float4 *res;         // [K][capacity] accumulated result rgb (w unused)
    float2 *pos;         // sample position on the film
    float4 *sh_a;        // shadow ray o.xyz, maxt
    float4 *sh_b;        // shadow ray d.xyz, time
    float4 *sh_c;        // [K][capacity] candidate result rgb, w = as_float(lane position)
    uint32_t *q[2];      // active-lane index queues (ping-pong), segmented: entry j of segment S at S*kSeg + j
    uint32_t *counts;    // [iteration][2][n_segments]: survivors / shadow rays per segment
    uint4 *cand;         // DEFER (ray kernels of large meshes): up to four objects a ray's TLAS walk put aside, by lane (k_trace) / by shadow slot (k_shadow); nullptr = no second launch
    uint32_t *defer_idx; // ... the lanes / shadow slots of a segment that have some, entry j of segment S at S*kSeg + j, and
    uint32_t *defer_cnt; // ... how many (zeroed before each first launch)
    uint32_t *seg_counter;   // resident first-bounce kernel: next segment to hand out (zeroed before the launch)
    uint32_t capacity;
    uint32_t xcd_remap;  // unstaged ray kernels: XCD-aware block order (dtof_kernels.hip: xcd_remap); 0 = block b traces segment b
    uint32_t id_shift;   // bits of hit_id that hold the object index: 24 unless the scene needs more shapes per group than 8 bits hold (render_rows)
};

struct LaneDebug {       // mirrors orc_lane's comparable fields
    float sample_pos[2]; float time; float ray_o[3]; float ray_d[3]; float rgb[3]; float valid;
};

// The arguments of k_shade (dtof_shade.h) travel as ONE by-value block, read through the kernarg segment pointer.
struct ShadeArgs {
    const uint8_t *scene; uint32_t scene_bytes, stage_words; RenderParams rp; Queues q;
    const uint32_t *qin, *count_in; uint32_t *qout, *alive_out, *shadow_out; uint32_t depth, trace_next; LaneDebug *dbg;
    float *film; uint64_t film_stride;   // first-bounce kernel whose launch covers the whole path: it splats its lanes itself (k_shade, "fused splat"); nullptr: the splat kernels do
    uint32_t n_seg, res_small_off, res_small_words, res_memo;   // resident stage (RESW != 0): segments of the batch; byte offset / uint4 count of the record block copied to LDS; 1 = the instance memo has LDS
    uint32_t res_park_off;   // resident stage with several films: word offset (behind the memo) of the film-state columns, kParkWords words per thread behind the stack columns
};
// Resident kernels of several films keep the films' running results in LDS instead of registers (k_shade: RES_LDS): kParkWords of the 3 * kMaxOffsets floats per
// thread -- what Domino's stage leaves free of the CU's 160 KiB at 16 waves (64 KiB node planes + 3 KiB records + 48 KiB stack columns) -- the last one stays a register
constexpr uint32_t kParkWords = 11;
// DTOF_PARK (default 1): the resident ONE-film kernels at 16 waves (128 VGPRs, ~150 spilled) park the path state no traversal reads -- both PCG states with their
// stream selectors and throughput / path length, kParkState words -- in the same columns across the two traversals of an iteration instead of leaving them to the register
// allocator's spill code: scratch 200 -> 168 B per lane, C4 33.69 -> 33.26 ms (profiles/r05_k4_film_state.txt).  A scene whose stage no longer fits the CU's LDS with
// the columns at 16 waves takes 12 (resident_lds_bytes / render_rows' step-down), where the kernels have 168 VGPRs and park nothing.
#ifndef DTOF_PARK
#define DTOF_PARK 1
#endif
constexpr uint32_t kParkState = 10, kParkRng = 6;   // one film: both streams + throughput / path length; several films (behind the kParkWords film words): the streams only
// the stack columns of the resident kernels of several films hold 16-bit entries (dtof_traverse.h: encode_child16), the one-film kernels' 32-bit ones
static inline uint32_t resident_stack_bytes(uint32_t depth, uint32_t waves, bool several_films) { return (depth < 2 ? 2 : depth) * waves * 64u * (several_films ? 2u : 4u); }

// One launch of k_shade as launch_shade hands it to the translation unit that holds the instantiation (dtof_shade_*.hip: the ~100 instantiations of the
// kernel compile in parallel, one group per file): staged = the scene blob is copied to LDS by every block; mode 0 split, 1 fused, 2 fused first bounce;
// waves != 0: the resident form (`waves` waves per block, one block per CU).
struct ShadeLaunch { bool staged; int mode; uint32_t waves, grid, lds; hipStream_t stream; ShadeArgs args; };
void launch_shade_plain(bool area, bool k4, const ShadeLaunch &L);      // rectangle-only diffuse scenes          (dtof_shade_plain.hip)
void launch_shade_mesh(bool area, bool k4, const ShadeLaunch &L);       // + triangles / analytic shapes          (dtof_shade_mesh.hip)
void launch_shade_spec1(bool k4, const ShadeLaunch &L);                 // every BSDF / emitter / texture         (dtof_shade_spec1.hip)
void launch_shade_spec2(bool k4, const ShadeLaunch &L);                 // ... and blendbsdf                      (dtof_shade_spec2.hip)
void launch_shade_resident0(bool area, bool k4, const ShadeLaunch &L);  // resident first bounce, diffuse scenes  (dtof_shade_res0.hip)
void launch_shade_resident1(bool k4, const ShadeLaunch &L);             // resident first bounce, every BSDF      (dtof_shade_res1.hip)
void launch_shade_resident2(bool k4, const ShadeLaunch &L);             // ... and blendbsdf                      (dtof_shade_res2.hip)

// Resident stage of the fused first-bounce kernel (k_shade<..., RESW>, dtof_kernels.hip): `waves` waves per block (0 = off), one block per CU; the block
// [small_off, small_off + 16 * small_words) of the blob (groups, shapes, emitters, triangles, shading data) and the TLAS nodes live in LDS.
struct ResidentStage { uint32_t small_off = 0, small_words = 0, waves = 0; };
constexpr uint32_t kResidentNodes = 1024;   // TLAS nodes the stage holds (= kResNodes of dtof_traverse.h)
// dynamic LDS one block of the resident kernel needs with `waves` waves: node planes + record block + (instance memo) + stack columns
uint32_t resident_lds_bytes(const RenderParams &rp, const ResidentStage &resident, uint32_t stack_depth, uint32_t waves);
uint32_t device_lds_limit();   // hipDeviceAttributeMaxSharedMemoryPerBlock of the current device (160 KiB on gfx950), minus the kernels' static LDS

// kernels (dtof_kernels.hip)
void launch_generate(const RenderParams &rp, const Queues &q, hipStream_t s);
void launch_sum_counts(const uint32_t *counts, uint32_t n_seg, uint32_t n_rows, unsigned long long *out, hipStream_t s);
uint32_t segments_for(uint32_t n_lanes);   // number of queue segments (count slots) for a batch
void launch_trace(const uint8_t *scene, uint32_t scene_bytes, const RenderParams &rp, const Queues &q,
                  const uint32_t *qin, const uint32_t *count_in, uint32_t stack_depth, hipStream_t s);
void launch_shade(const uint8_t *scene, uint32_t scene_bytes, const RenderParams &rp, const Queues &q,
                  const uint32_t *qin, const uint32_t *count_in, uint32_t *qout,
                  uint32_t *alive_out, uint32_t *shadow_out, uint32_t depth, bool fused, bool trace_next,
                  uint32_t stack_depth, hipStream_t s, bool first = false, LaneDebug *dbg = nullptr,   // first: generate + primary trace inline (fused only)
                  const ResidentStage *resident = nullptr, float *film = nullptr, uint64_t film_stride = 0);   // film: the launch covers the whole path and splats its lanes itself
void launch_shadow(const uint8_t *scene, uint32_t scene_bytes, const RenderParams &rp, const Queues &q,
                   const uint32_t *count_in, uint32_t stack_depth, hipStream_t s);
void launch_velocity(const uint8_t *scene, uint32_t scene_bytes, const RenderParams &rp, const Queues &q, uint32_t stack_depth, hipStream_t s);
void launch_splat(const RenderParams &rp, const Queues &q, float *film, uint64_t plane_stride, hipStream_t s);
void launch_develop(const float *film, float *rgb, int64_t n_pixels, hipStream_t s);
void launch_develop_rgba(const float *film, const float *alpha_film, float *rgba, int64_t n_pixels, hipStream_t s);   // pixel_format = rgba (hdrfilm.cpp:339-400)
void launch_lane_dump(const RenderParams &rp, const Queues &q, LaneDebug *out, hipStream_t s);
void launch_pass_save(const RenderParams &rp, const Queues &q, hipStream_t s);   // multi-pass: main / path stream states of the batch -> rp.pass_rng
void launch_lane_dump_rays(const RenderParams &rp, const Queues &q, LaneDebug *out, hipStream_t s);

// sampler KAT kernels
struct SamplerState { uint2 *rng, *rng_time, *rng_path; uint32_t *perm_seed, *dim; uint32_t n; };
void launch_sampler_seed(const RenderParams &rp, const SamplerState &st, hipStream_t s);
void launch_sampler_next_correlate(const RenderParams &rp, const SamplerState &st, const uint8_t *correlate, int correlate_all,
                                   float *out, hipStream_t s);
void launch_sampler_next_1d(const RenderParams &rp, const SamplerState &st, float *out, hipStream_t s);
void launch_sampler_next_time(const RenderParams &rp, const SamplerState &st, uint32_t sample_index_base, float *out, hipStream_t s);
void launch_waveform_eval(const RenderParams &rp, const float *t, const float *len, float *out, int mode, uint32_t n, hipStream_t s);

// component evaluation (dtof_eval_component; the ids are the DTOF_COMP_* values of include/dtof.h)
enum { COMP_MICROFACET_EVAL = 0, COMP_MICROFACET_PDF = 1, COMP_MICROFACET_G1 = 2, COMP_MICROFACET_SAMPLE = 3, COMP_FRESNEL = 4,
       COMP_FRESNEL_CONDUCTOR = 5, COMP_RFILTER = 6, COMP_WARP_COSINE_HEMISPHERE = 7, COMP_WARP_DISK_CONCENTRIC = 8,
       COMP_WARP_UNIFORM_TRIANGLE = 9, COMP_WARP_UNIFORM_SPHERE = 10, COMP_COORDINATE_SYSTEM = 11, COMP_TEA_FLOAT32 = 12, COMP_MATH = 13,
       COMP_COUNT = 14 };
struct ComponentArgs { int component; float p[8]; const float *in; int in_stride; float *out; int out_stride; uint32_t n; };
void launch_component(const ComponentArgs &a, const RenderParams &rp, hipStream_t s);
void launch_bsdf_eval(const uint8_t *scene, uint32_t shape_index, const float *in, float *out, uint32_t n, hipStream_t s);   // BSDF::eval_pdf_sample over arrays (dtof_shade_spec2.hip)
void launch_camera_rays(const RenderParams &rp, const float *in, float *out, uint32_t n, hipStream_t s);   // Sensor::sample_ray over arrays (known-answer entry)
// Scene::ray_intersect / ray_test over arrays
void launch_ray_query(const uint8_t *scene, const float *rays, float *out, int32_t *ids, float *uv4, uint32_t n, bool any, uint32_t stack_depth, hipStream_t s);

#ifdef DTOF_TRAVERSAL_STATS
bool read_traversal_stats(unsigned long long *out8);
void register_traversal_stats_reader(bool (*reader)(unsigned long long *acc8));
#endif

}  // namespace dtof
