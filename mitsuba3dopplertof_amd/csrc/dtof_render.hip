// dtof_render.hip -- host orchestration of the wavefront renderer and the C ABI (include/dtof.h).
//
// Replaces, for the `dopplertofpath` + `correlated` path only, SamplingIntegrator::render
// (src/render/integrator.cpp:104-347, JIT branch :226-340): wavefront set-up, sampler seeding,
// lane->pixel mapping, the bounce loop and the film develop.  One host thread drives one HIP
// stream; the wavefront of W*H*spp lanes is cut into row-band batches (results are invariant to
// the cut because every lane's RNG streams are pure functions of its global lane index,
// sampler.cpp:115-134 / correlated.cpp:38-64).
#include "../../include/dtof.h"
#include "dtof_kernels.h"
#include "dtof_scene.h"
#include "dtof_math.h"
#include <atomic>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <string>
#include <memory>
#include <vector>

using namespace dtof;

namespace {

thread_local std::string g_last_error;

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    throw HipError(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

template <typename F> int guarded(F &&f) {
    try { f(); return DTOF_OK; }
    catch (const HipError &e) { g_last_error = e.what(); return DTOF_ERR_HIP; }
    catch (const std::exception &e) { g_last_error = e.what(); return DTOF_ERR_INVALID; }
    catch (...) { g_last_error = "unknown error"; return DTOF_ERR_INVALID; }
}

template <typename T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    void ensure(size_t count) {
        if (count <= n) return;
        release();
        HIP_CHECK(hipMalloc((void **) &p, count * sizeof(T))); n = count;
    }
    void release() { if (p) { (void) hipFree(p); p = nullptr; n = 0; } }
    ~DevBuf() { release(); }
};

// per-iteration count slots of a batch (statistics + the queue counts of the previous iteration); reused cyclically beyond that.
// DTOF_STAT_SLOTS shrinks it so that the tests can exercise the wrap-around with short paths.
static const uint32_t kMaxIter = [] { const char *e = getenv("DTOF_STAT_SLOTS"); int v = e ? atoi(e) : 0; return (uint32_t) (v >= 2 ? v : 256); }();
static uint64_t target_batch_lanes() {   // lanes per wavefront batch (DTOF_BATCH_LANES overrides)
    // 2^27 lanes (26 GB of workspace at 200 B per lane, 40 GB with four offset films -- of 288): every launch ends with a tail in which the CUs run dry one after the
    // other, and a Domino frame in 32 launches of 2^24 lanes lost 7 % to it (C5 206 -> 193 ms, C4 44.8 -> 41.1; profiles/r03_batch_lanes.txt); one launch per C4 frame
    // instead of two is another 2 % (34.86 -> 34.14 ms, profiles/r04_domino_waves_batch.txt).  render_range halves the batch until its workspace fits the free device memory.
    const char *e = getenv("DTOF_BATCH_LANES"); const uint64_t x = e ? strtoull(e, nullptr, 10) : 0; const uint64_t v = x ? x : (1ull << 27);   // read per call: tests of the batch seams set it
    return v;
}

struct Workspace {
    DevBuf<float4> ray_a, ray_b, st_a, st_b, res, sh_a, sh_b, sh_c;
    DevBuf<uint4> hit, rng_a;
    DevBuf<uint32_t> hit_id, q0, q1, counts;
    DevBuf<float> hit_t;
    DevBuf<float2> pos, st_c;
    DevBuf<uint2> rng_b;
    DevBuf<LaneDebug> dbg;
    DevBuf<float4> valid;   // Queues::valid_out, only when asked for (ensure_valid)
    DevBuf<uint4> cand; DevBuf<uint32_t> defer_idx, defer_cnt;   // Queues::cand / defer_idx / defer_cnt, only for scenes whose ray kernels run as a pair of launches (ensure_defer)
    uint32_t capacity = 0; int k = 0;
    void ensure_valid() { valid.ensure(capacity); }
    void ensure_defer() { cand.ensure(capacity); defer_idx.ensure(capacity); defer_cnt.ensure(segments_for(capacity)); }
    void ensure(uint32_t cap, int n_offsets) {
        if (cap <= capacity && n_offsets <= k) return;
        capacity = std::max(cap, capacity); k = std::max(n_offsets, k);
        ray_a.ensure(capacity); ray_b.ensure(capacity); st_a.ensure(capacity); st_b.ensure(capacity);
        res.ensure((size_t) capacity * k); sh_a.ensure(capacity); sh_b.ensure(capacity); sh_c.ensure((size_t) capacity * k);
        hit.ensure(capacity); hit_t.ensure(capacity); rng_a.ensure(capacity); hit_id.ensure(capacity); q0.ensure(capacity); q1.ensure(capacity);
        counts.ensure(2 * (size_t) kMaxIter * segments_for(capacity) + 16); pos.ensure(capacity); rng_b.ensure(capacity); st_c.ensure(capacity);   // + the segment counter of the resident kernel
    }
    Queues queues() {
        Queues q; memset(&q, 0, sizeof q);
        q.ray_a = ray_a.p; q.ray_b = ray_b.p; q.hit = hit.p; q.hit_t = hit_t.p; q.hit_id = hit_id.p; q.st_a = st_a.p; q.st_b = st_b.p; q.rng_a = rng_a.p; q.rng_b = rng_b.p; q.st_c = st_c.p;
        q.res = res.p; q.pos = pos.p; q.sh_a = sh_a.p; q.sh_b = sh_b.p; q.sh_c = sh_c.p; q.q[0] = q0.p; q.q[1] = q1.p;
        q.counts = counts.p; q.capacity = capacity; q.valid_out = valid.p;
        q.seg_counter = counts.p + 2 * (size_t) kMaxIter * segments_for(capacity);
        return q;
    }
};

}  // namespace

struct dtof_scene {
    HostScene host;
    PluginParams pp;
    std::vector<uint8_t> blob;
    DevBuf<uint8_t> d_blob; bool uploaded = false;
    Workspace ws, ws2;                       // one per in-flight batch
    DevBuf<float> d_film, d_rgb;
    // the caller's device film as declared with dtof_scene_set_film_layout (0 = not declared: colour planes only, W * H * 4 apart), and the plane distance of the running call
    int32_t film_planes = 0; uint64_t film_plane_stride = 0, film_stride_call = 0;
    DevBuf<unsigned long long> d_sums;       // [batch][2*kMaxIter] per-iteration totals (survivors, shadow rays)
    DevBuf<uint2> d_pass_rng;                // multi-pass renders: [lane][3] stream states between the passes
    uint32_t id_shift = 24;                  // Queues::id_shift of this scene
    hipStream_t stream = nullptr, stream2 = nullptr;   // stream: the library's own, or the caller's (dtof_scene_set_stream)
    hipStream_t own_stream = nullptr;                    // what ensure_device created and the destructor destroys
    std::atomic<bool> stop { false };
    // reusable statistics plumbing (creating events / pinned memory per call costs ~0.3 ms)
    std::vector<hipEvent_t> event_pool; size_t events_used = 0;
    // frames enqueued by dtof_render_rows_async and not collected yet: their events (frame, stages) and launch counters; no host synchronisation until dtof_async_collect
    struct DeferredFrame { hipEvent_t ev0, ev1; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[6]; dtof_render_stats counters; };
    std::vector<DeferredFrame> deferred; bool defer_next = false;
    uint32_t *pinned_counts = nullptr; size_t pinned_words = 0;
    hipEvent_t take_event() {
        if (events_used == event_pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) throw std::runtime_error("hipEventCreate failed"); event_pool.push_back(e); }
        return event_pool[events_used++];
    }
    uint32_t *pinned(size_t words) {
        if (words > pinned_words) {
            if (pinned_counts) (void) hipHostFree(pinned_counts);
            pinned_counts = nullptr; pinned_words = 0;
            if (hipHostMalloc((void **) &pinned_counts, words * 4, hipHostMallocDefault) != hipSuccess) throw std::runtime_error("hipHostMalloc failed");
            pinned_words = words;
        }
        return pinned_counts;
    }
    ~dtof_scene() {
        if (own_stream) (void) hipStreamDestroy(own_stream); if (stream2) (void) hipStreamDestroy(stream2);
        for (auto e : event_pool) (void) hipEventDestroy(e);
        if (pinned_counts) (void) hipHostFree(pinned_counts);
    }
};

struct dtof_sampler {
    uint32_t sample_count = 4, base_seed = 0; int32_t tcn = 2, pcn = 2;
    uint32_t seed = 0, wavefront = 0, spw = 1, sample_index = 0; bool seeded = false;
    DevBuf<uint2> rng, rng_time, rng_path; DevBuf<uint32_t> perm, dim; DevBuf<float> out; DevBuf<uint8_t> flags;
};

namespace {

void ensure_device(dtof_scene *sc) {
    if (!sc->own_stream) HIP_CHECK(hipStreamCreate(&sc->own_stream));
    if (!sc->stream) sc->stream = sc->own_stream;
    if (!sc->stream2) HIP_CHECK(hipStreamCreate(&sc->stream2));
    if (!sc->uploaded) {
        sc->d_blob.ensure(sc->blob.size());
        HIP_CHECK(hipMemcpy(sc->d_blob.p, sc->blob.data(), sc->blob.size(), hipMemcpyHostToDevice));
        sc->uploaded = true;
    }
}

// 4x4 float product with the fmadd chain of Dr.Jit's column-major matrix product
void m4_mul(const float *a, const float *b, float *out) {
    float r[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float s = a[4 * i] * b[j];
        for (int k = 1; k < 4; ++k) s = fmaf(a[4 * i + k], b[4 * k + j], s);
        r[4 * i + j] = s;
    }
    memcpy(out, r, sizeof r);
}
void m4_identity(float *m) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.f; }

// sample_to_camera: inverse of perspective_projection (include/mitsuba/render/sensor.h:226-262) as
// PerspectiveCamera::update_camera_transforms builds it (src/sensors/perspective.cpp:172-198); the
// Transform class carries analytic inverses, so this is the reversed product of the factor inverses.
void sample_to_camera(const HostSensor &s, float *inv_out) {
    float fw = (float) s.film_w, fh = (float) s.film_h;
    float rel_sx = (float) s.crop_w / fw, rel_sy = (float) s.crop_h / fh;
    float rel_ox = (float) s.crop_x / fw, rel_oy = (float) s.crop_y / fh;
    float aspect = fw / fh, near_ = s.near_clip, far_ = s.far_clip;
    float tanv = (float) std::tan((double) (s.x_fov * .5f) * (M_PI / 180.0));
    float S1i[16], T1i[16], S2i[16], T2i[16], Pi[16], tmp[16];
    m4_identity(S1i); S1i[0] = rcp(1.f / rel_sx); S1i[5] = rcp(1.f / rel_sy);
    m4_identity(T1i); T1i[3] = rel_ox; T1i[7] = rel_oy;
    m4_identity(S2i); S2i[0] = rcp(-0.5f); S2i[5] = rcp(-0.5f * aspect);
    m4_identity(T2i); T2i[3] = 1.f; T2i[7] = 1.f / aspect;
    memset(Pi, 0, 64); Pi[0] = tanv; Pi[5] = tanv; Pi[15] = rcp(near_); Pi[11] = 1.f; Pi[14] = (near_ - far_) / (far_ * near_);
    if (s.orthographic) {   // orthographic_projection (sensor.h:266-299): the last factor is scale(1, 1, 1 / (far - near)) * translate(0, 0, -near) (transform.h:242-245)
        float OT[16], OS[16];
        m4_identity(OT); OT[11] = near_;
        m4_identity(OS); OS[0] = rcp(1.f); OS[5] = rcp(1.f); OS[10] = rcp(1.f / (far_ - near_));
        m4_mul(OT, OS, Pi);
    }
    m4_mul(T1i, S1i, tmp); m4_mul(S2i, tmp, tmp); m4_mul(T2i, tmp, tmp); m4_mul(Pi, tmp, inv_out);
}

// the film's reconstruction filter as the splat kernels see it
void set_filter(RenderParams &rp, int32_t filter, float radius, float stddev, float B, float C) {
    rp.filter = filter; rp.filter_radius = radius; rp.inv_radius = 1.f / radius;
    rp.filter_b = B; rp.filter_c = C;
    if (filter == FILTER_GAUSSIAN) {   // GaussianFilter ctor (src/rfilters/gaussian.cpp:60-89), non-CUDA branch
        static const double coeff[10] = { 9.992604880e-1, -4.977025247e-1, 1.222248550e-1, -1.932406282e-2, 2.136713061e-3,
                                          -1.679873860e-4, 9.202145248e-6, -3.329417433e-7, 7.128382794e-9, -6.821193280e-11 };
        double scale = 1;
        for (int i = 0; i < 10; ++i) { rp.gauss_coeff[i] = (float) (coeff[i] * scale); scale /= (double) stddev * (double) stddev; }
        rp.gauss_coeff[0] -= estrin10(radius * radius, rp.gauss_coeff);
    }
}
// spp = samples per wavefront (per pass); sample_count = Sampler::sample_count() of the whole render (0: the same)
RenderParams make_params(const dtof_scene *sc, uint32_t seed, uint32_t spp, const float *offsets, int n_offsets, uint32_t sample_count = 0) {
    const HostSensor &se = sc->host.sensor; const PluginParams &pp = sc->pp;
    RenderParams rp; memset(&rp, 0, sizeof rp);
    sample_to_camera(se, rp.s2c);
    memcpy(rp.cam_to_world, se.to_world, 48);
    rp.near_clip = se.near_clip; rp.far_clip = se.far_clip; rp.shutter_open = se.shutter_open;
    rp.shutter_open_time = se.shutter_close - se.shutter_open;
    rp.orthographic = se.orthographic ? 1 : 0;
    rp.aperture_radius = se.thinlens ? se.aperture_radius : 0.f; rp.focus_distance = se.focus_distance;
    rp.crop_x = se.crop_x; rp.crop_y = se.crop_y; rp.crop_w = se.crop_w; rp.crop_h = se.crop_h;
    rp.scale_x = 1.f / (float) se.crop_w; rp.scale_y = 1.f / (float) se.crop_h;
    rp.offset_x = -(float) se.crop_x * rp.scale_x; rp.offset_y = -(float) se.crop_y * rp.scale_y;
    set_filter(rp, se.filter, se.filter_radius, se.filter_stddev, se.filter_b, se.filter_c);
    rp.base_seed = pp.base_seed; rp.seed = seed; rp.seed_value = pp.base_seed + seed;
    rp.spp = spp; rp.spp_log2 = 0xffffffffu;
    for (uint32_t b = 0; b < 32; ++b) if ((1u << b) == spp) rp.spp_log2 = b;
    rp.tcn = (uint32_t) pp.time_correlate_number; rp.pcn = (uint32_t) pp.path_correlate_number;
    rp.time_sampling = pp.time_sampling; rp.antithetic_shift = pp.antithetic_shift; rp.stratify = pp.stratify_each_interval;
    if (sample_count == 0) sample_count = spp;
    rp.sample_count = sample_count; rp.d_sample_count = make_fastdiv(sample_count);
    rp.n_stratum = sample_count / rp.tcn;                          // int n_stratum = m_sample_count / tcn (correlated.cpp:112)
    rp.inv_n_stratum = rp.n_stratum ? 1.0f / (float) (int) rp.n_stratum : 0.f;
    rp.inv_tcn = 1.0f / (float) pp.time_correlate_number;
    rp.d_spp = make_fastdiv(spp); rp.d_w = make_fastdiv((uint32_t) se.crop_w); rp.d_tcn = make_fastdiv(rp.tcn); rp.d_pcn = make_fastdiv(rp.pcn);
    rp.d_stratum = make_fastdiv(rp.n_stratum);
    for (uint32_t d : { spp, sample_count, (uint32_t) se.crop_w, rp.tcn, rp.pcn, rp.n_stratum })   // the kernels have no other division: fail loudly
        for (uint32_t n : { 0u, 1u, d - 1, d, d + 1, 2 * d - 1, 0x7fffffffu, 0xfffffffeu, 0xffffffffu })
            if (d && fdiv(n, make_fastdiv(d)) != n / d) throw std::runtime_error("internal error: fast division self-check failed");
    rp.n_passes = 1;
    // eval_modulation_weight's scalar prefactors are folded in double and rounded to float32 once
    // (they multiply JIT float32 arrays), dopplertofpath.cpp:62-69
    rp.T = pp.time;
    rp.w_g = (float) (2 * M_PI * (double) pp.w_g_mhz * 1e6);
    rp.w_d = (float) (2 * M_PI / (double) pp.time * (double) pp.hetero_frequency);
    rp.phi_coef = (float) ((2 * M_PI * (double) pp.w_g_mhz) / 300);
    rp.amp = (float) (0.5 * (double) pp.g_1);
    rp.g_1 = pp.g_1; rp.g_0 = pp.g_0;
    rp.wave_type = pp.wave_type; rp.low_pass = pp.low_frequency_component_only;
    if (n_offsets <= 0) { rp.n_offsets = 1; rp.phase[0] = pp.phase_offset; }
    else {
        if (n_offsets > kMaxOffsets) throw std::runtime_error("at most 4 modulation offsets can be batched per traversal");
        rp.n_offsets = n_offsets;
        for (int k = 0; k < n_offsets; ++k) rp.phase[k] = (float) ((double) (offsets[k] * 2) * M_PI);   // dopplertofpath.cpp:30-32
    }
    rp.path_correlation_depth = pp.path_correlation_depth; rp.max_depth = pp.max_depth; rp.rr_depth = pp.rr_depth;
    rp.integrator = pp.integrator;
    rp.sampler_kind = pp.sampler_kind; rp.jitter = pp.jitter; rp.inv_spp = 1.0f / (float) sample_count;   // dr::rcp(ScalarFloat(m_sample_count))
    if (pp.integrator != INTEGRATOR_DOPPLER && n_offsets > 0) throw std::runtime_error("modulation offsets only apply to the dopplertofpath integrator");
    return rp;
}

// Optional roctx ranges around the stage launches (the counterpart of the reference's ScopedPhase / NVTX ranges,
// include/mitsuba/core/profiler.h): DTOF_ROCTX=1 loads the roctx library at run time, `rocprofv3 --marker-trace` then shows
// "dtof:generate|trace|shade|shadow|splat|first" ranges on the host timeline.  No link-time dependency.
struct Roctx {
    int (*push)(const char *) = nullptr; int (*pop)() = nullptr;
    Roctx() {
        const char *e = getenv("DTOF_ROCTX");
        if (!e || e[0] == '0') return;
        void *h = nullptr;   // rocprofv3 listens to the SDK's roctx; the older libroctx64 serves rocprof v1 / v2
        for (const char *name : { "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/libroctx64.so" })
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return;
        push = (int (*)(const char *)) dlsym(h, "roctxRangePushA"); pop = (int (*)()) dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr;
    }
};
static const Roctx &roctx() { static const Roctx r; return r; }
static const char *const kStageNames[6] = { "dtof:generate", "dtof:trace", "dtof:shade", "dtof:shadow", "dtof:splat", "dtof:first" };

struct StageTimer {
    bool on; dtof_scene *sc; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[6];
    StageTimer(bool enabled, dtof_scene *scene) : on(enabled), sc(scene) { if (sc->deferred.empty()) sc->events_used = 0; }   // the events of uncollected frames stay taken
    int begin(int stage, hipStream_t s) {
        if (roctx().push) roctx().push(kStageNames[stage]);
        if (!on) return -1;
        hipEvent_t a = sc->take_event(), b = sc->take_event();
        ev[stage].emplace_back(a, b); HIP_CHECK(hipEventRecord(a, s));
        return (int) ev[stage].size() - 1;
    }
    void end(int stage, int idx, hipStream_t s) {
        if (on) HIP_CHECK(hipEventRecord(ev[stage][idx].second, s));
        if (roctx().push) roctx().pop();
        static const bool sync_each = [] { const char *e = getenv("DTOF_SYNC_LAUNCHES"); return e && e[0] == '1'; }();   // debugging: which stage of a frame never finishes?
        if (sync_each) { fprintf(stderr, "[dtof] %s enqueued ...", kStageNames[stage]); fflush(stderr); HIP_CHECK(hipStreamSynchronize(s)); fprintf(stderr, " done\n"); fflush(stderr); }
    }
    double total(int stage) {
        double ms = 0;
        for (auto &p : ev[stage]) { float t = 0; HIP_CHECK(hipEventElapsedTime(&t, p.first, p.second)); ms += t; }
        return ms;
    }
};

// The device-film entry points write K colour planes and, for an rgba film, the alpha plane behind them into memory whose size only the caller knows: an rgba scene is
// refused until the caller has declared a film of K + 1 planes (a caller written for rgb films would have its buffer overrun), and a declared count is checked either way.
static uint64_t caller_film_stride(const dtof_scene *sc, int n_offsets) {
    const HostSensor &se = sc->host.sensor;
    const int need = (n_offsets <= 0 ? 1 : n_offsets) + (se.alpha ? 1 : 0);
    if (se.alpha && sc->film_planes < need)
        throw std::runtime_error("rgba film: the device film needs " + std::to_string(need) + " RGBW planes (the alpha film lies behind the colour films); declare them with dtof_scene_set_film_layout");
    if (sc->film_planes != 0 && sc->film_planes < need)
        throw std::runtime_error("the device film was declared with " + std::to_string(sc->film_planes) + " planes, this call writes " + std::to_string(need));
    const uint64_t full = (uint64_t) se.crop_w * se.crop_h * 4;
    return sc->film_plane_stride ? sc->film_plane_stride : full;
}

// The wavefront loop over pixel rows [row_begin,row_end); accumulates into d_film (K films, sc->film_stride_call floats apart).
// lane_dump != nullptr: evaluate only lanes [dump_begin, dump_begin + dump_n) and copy their records out.
// stripe_rows > 0: the rows are the stripes [row_begin + k * stripe_period, ... + stripe_rows) below row_end (interleaved shards).
void render_rows(dtof_scene *sc, uint32_t seed, uint32_t spp, int32_t row_begin, int32_t row_end,
                 const float *offsets, int n_offsets, float *d_film, dtof_render_stats *stats,
                 LaneDebug *lane_dump = nullptr, uint64_t dump_begin = 0, uint64_t dump_n = 0,
                 uint32_t stripe_rows = 0, uint32_t stripe_period = 0) {
    if (!sc->host.has_sensor) throw std::runtime_error("the scene does not contain a sensor");
    ensure_device(sc);
    const HostSensor &se = sc->host.sensor;
    const uint64_t film_stride = sc->film_stride_call ? sc->film_stride_call : (uint64_t) se.crop_w * se.crop_h * 4;
    if (spp == 0) spp = sc->pp.sample_count;
    if (spp == 0) throw std::runtime_error("sample count must be positive");
    // SamplingIntegrator::render (integrator.cpp:121-135,227-245): spp_per_pass = min(samples_per_pass, spp) must divide spp; a wavefront
    // of more than 2^32 - 1 lanes is split into more passes (integer division, as written there), and Sampler::set_samples_per_wavefront
    // (sampler.cpp:75-83) insists that the sample count is a multiple of the samples per pass.  `spp` below is the samples per PASS.
    const uint32_t sample_count = spp;
    uint32_t n_passes = 1;
    {
        uint32_t per_pass = sc->pp.samples_per_pass == 0xffffffffu || sc->pp.samples_per_pass == 0 ? spp : std::min(sc->pp.samples_per_pass, spp);
        if (spp % per_pass != 0) throw std::runtime_error("sample_count (" + std::to_string(spp) + ") must be a multiple of spp_per_pass (" + std::to_string(per_pass) + ").");
        const uint64_t wavefront = (uint64_t) se.crop_w * se.crop_h * per_pass, limit = 0xffffffffull;
        if (wavefront > limit) {
            per_pass /= (uint32_t) ((wavefront + limit - 1) / limit);
            if (per_pass == 0 || spp % per_pass != 0) throw std::runtime_error("sample_count should be a multiple of samples_per_wavefront!");
        }
        n_passes = spp / per_pass; spp = per_pass;
    }
    uint64_t total_lanes = (uint64_t) se.crop_w * se.crop_h * spp;   // lanes of one pass (the wavefront)
    if (sc->pp.time_sampling != TIME_UNIFORM && sc->pp.stratify_each_interval && sample_count < (uint32_t) sc->pp.time_correlate_number)
        throw std::runtime_error("sample count must be at least time_correlate_number when per-interval stratification is on");
    if (sc->pp.integrator == 0 && sc->pp.sampler_kind == SAMPLER_CORRELATED && sc->pp.time_sampling == TIME_ANTITHETIC_MIRROR && sc->pp.time_correlate_number != 2)
        throw std::runtime_error("antithetic_mirror time sampling needs time_correlate_number == 2");   // Assert(m_time_correlate_number == 2), correlated.cpp:142
    RenderParams rp = make_params(sc, seed, spp, offsets, n_offsets, sample_count);
    rp.n_passes = n_passes;
    // lane dumps address (pass, lane) as pass * wavefront + lane and must stay inside one pass
    uint32_t dump_pass = 0;
    if (lane_dump) {
        dump_pass = (uint32_t) (dump_begin / total_lanes); dump_begin %= total_lanes;
        if (dump_pass >= n_passes || dump_begin + dump_n > total_lanes) throw std::runtime_error("lane range exceeds the wavefront");
    }
    row_begin = std::max(row_begin, 0); row_end = std::min(row_end, se.crop_h);
    uint64_t lanes_per_row = (uint64_t) se.crop_w * spp;
    uint64_t first = lane_dump ? dump_begin : lanes_per_row * (uint64_t) row_begin;
    uint64_t last = lane_dump ? dump_begin + dump_n : lanes_per_row * (uint64_t) std::max(row_end, row_begin);
    if (last > total_lanes) throw std::runtime_error("lane range exceeds the wavefront");
    if (stripe_rows) {   // virtual rows [0, V): the rows of this shard's stripes in ascending order
        if (stripe_period < stripe_rows) throw std::runtime_error("stripe period must be at least the stripe height");
        const uint64_t span = (uint64_t) std::max(row_end - row_begin, 0), full = span / stripe_period, rest = span % stripe_period;
        const uint64_t v_rows = full * stripe_rows + std::min<uint64_t>(rest, stripe_rows);
        rp.stripe_rows = stripe_rows; rp.stripe_period = stripe_period; rp.stripe_first = (uint32_t) row_begin; rp.lanes_per_row = (uint32_t) lanes_per_row;
        rp.d_lanes_per_row = make_fastdiv(rp.lanes_per_row); rp.d_stripe_rows = make_fastdiv(stripe_rows);
        first = 0; last = v_rows * lanes_per_row;
    }
    uint64_t batch = lane_dump ? std::min<uint64_t>(target_batch_lanes(), std::max<uint64_t>(dump_n, 1))
                               : std::max<uint64_t>(1, target_batch_lanes() / lanes_per_row) * lanes_per_row;
    batch = std::min<uint64_t>(batch, std::max<uint64_t>(last - first, 1));
    if (batch > sc->ws.capacity) {   // a workspace that has to grow: keep it within the free device memory (168 B + 32 B per offset film per lane, two copies with two streams)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t per_lane = 168 + 32ull * (uint64_t) std::max<int32_t>(rp.n_offsets, 1);
            while (batch > (1ull << 22) && batch * per_lane * 2 > (uint64_t) free_b + (uint64_t) sc->ws.capacity * per_lane) batch = std::max<uint64_t>(1, (batch / 2) / lanes_per_row) * lanes_per_row;
        }
    }
    // Two batches are kept in flight on two HIP streams (each with its own workspace): the VALU-bound
    // trace/shadow kernels of one batch overlap the HBM-bound shade kernel of the other.
    static const int env_streams = [] { const char *e = getenv("DTOF_STREAMS"); int v = e ? atoi(e) : 1; return v == 2 ? 2 : 1; }();   // default 1: measured gain of 2 is 0% (Cornell) .. 7% (Domino) and it blurs per-stage timing
    const int n_streams = (lane_dump || n_passes > 1) ? 1 : env_streams;   // the passes of a lane follow each other on one stream
    // Pipeline choice.  "fused" runs occlusion + continuation traversal inside the shade kernel (one kernel per bounce), "split" runs
    // k_trace -> k_shade -> k_shadow per bounce.
    // DTOF_PIPELINE=split|fused overrides the automatic choice below.
    const int env_pipeline = [] { const char *e = getenv("DTOF_PIPELINE"); std::string v = e ? e : ""; return v == "split" ? 0 : v == "fused" ? 1 : 2; }();   // read per call: tests switch it
    const BlobHeader *bh = (const BlobHeader *) sc->blob.data();
    static const bool env_fuse_first = [] { const char *e = getenv("DTOF_FUSE_FIRST"); return !(e && e[0] == '0'); }();
    bool only_rectangles = bh->n_tris == 0;
    for (auto &sh : sc->host.shapes) only_rectangles &= sh.kind == SHAPE_RECT;
    // auto: fused (one kernel per bounce, and the first-bounce kernel running up to four iterations with the path state in registers) unless large
    // meshes sit behind their own BLAS -- deep per-mesh traversals diverge inside the fat shade kernel (mesh room, 522 k triangles: 19.8 ms fused
    // vs 16.1 ms split) -- or reflectances are textured (6.7 vs 5.8 ms).  Everything else measured faster fused once the fused kernels were
    // capped at 168 VGPRs = 3 waves / SIMD (512 x 512 x 64: Cornell boxes 5.26 -> 4.60 ms, area light 7.12 -> 6.13, sphere light 6.42 -> 4.83,
    // disk 5.30 -> 3.90, Domino 1024 x 1024 x 128 with its 1 025 instances 69.5 -> 62.2 ms; profiles/r02_pipeline_choice.txt).
    uint64_t blas_triangles = 0;   // triangles behind per-mesh BLASes: 18 k still run faster fused (11.2 vs 12.1 ms), 132 k do not (16.1 vs 13.7 ms)
    {
        const DShape *dshapes = (const DShape *) (sc->blob.data() + bh->off_shapes);
        for (uint32_t i = 0; i < bh->n_shapes; ++i) if (dshapes[i].kind == SHAPE_MESH && dshapes[i].blas_root != kNoChild) blas_triangles += dshapes[i].n_tris;
    }
    (void) only_rectangles;
    const bool fused = env_pipeline == 2 ? (blas_triangles <= 32768 && sc->host.textures.empty()) : env_pipeline == 1;
    if (n_streams == 2 && !lane_dump && last - first <= batch && last - first >= 2 * lanes_per_row) {
        uint64_t rows = (last - first) / lanes_per_row;
        batch = ((rows + 1) / 2) * lanes_per_row;                // one batch would serialise: cut it in two row bands
    }
    sc->ws.ensure((uint32_t) batch, rp.n_offsets);
    if (n_streams == 2) sc->ws2.ensure((uint32_t) batch, rp.n_offsets);
    if (lane_dump) sc->ws.dbg.ensure(batch);
    Queues qs[2] = { sc->ws.queues(), n_streams == 2 ? sc->ws2.queues() : sc->ws.queues() };
    qs[0].id_shift = qs[1].id_shift = sc->id_shift;

    hipStream_t ss[2] = { sc->stream, n_streams == 2 ? sc->stream2 : sc->stream };
    const uint8_t *blob = sc->d_blob.p; uint32_t blob_bytes = (uint32_t) sc->blob.size();
    const uint32_t stack_depth = ((const BlobHeader *) sc->blob.data())->tlas_depth;
    bool has_surface_emitters = false;          // area emitters make the emitter-hit term (and the last iteration) live
    for (auto &e : sc->host.emitters) has_surface_emitters |= e.kind == EMITTER_AREA || e.kind == EMITTER_CONSTANT || e.kind == EMITTER_ENVMAP;   // the environment is "hit" by the rays that leave the scene
    rp.has_area = has_surface_emitters;
    for (auto &sh : sc->host.shapes) rp.has_spec |= sh.bsdf != BSDF_DIFFUSE || sh.masked || sh.tex_normal >= 0 || sh.blend_other;
    for (auto &e : sc->host.emitters) rp.has_spec |= e.kind == EMITTER_SPOT || e.kind == EMITTER_DIRECTIONAL;
    rp.has_spec |= !sc->host.textures.empty();
    rp.has_spec |= se.thinlens || se.orthographic;   // the diffuse-only kernels generate perspective rays only (generate_lane<PERSPECTIVE_ONLY>)
    for (size_t ei = 0; ei < sc->host.emitters.size(); ++ei) if (sc->host.emitters[ei].kind == EMITTER_CONSTANT || sc->host.emitters[ei].kind == EMITTER_ENVMAP) { rp.has_env = 1; rp.env_index = (uint32_t) ei; rp.has_spec = 1; }
    // valid_ray leaves the kernels only when somebody reads it: the alpha channel of an rgba film (integrator.cpp:528-533) and the lane dumps
    rp.want_valid = (lane_dump || se.alpha) ? 1 : 0;
    if (rp.want_valid) { sc->ws.ensure_valid(); if (n_streams == 2) sc->ws2.ensure_valid(); qs[0].valid_out = sc->ws.valid.p; qs[1].valid_out = n_streams == 2 ? sc->ws2.valid.p : sc->ws.valid.p; }
    rp.hide_emitters = sc->pp.hide_emitters;   // textured reflectances are looked up in the SPEC instantiations only   // the spot branch lives in the SPEC instantiations (keeps the common kernels lean)
    bool has_spheres = false;
    for (auto &sh : sc->host.shapes) has_spheres |= sh.kind == SHAPE_SPHERE || sh.kind == SHAPE_DISK || sh.kind == SHAPE_CYLINDER;   // analytic shapes of the MESH instantiations
    // anything but rectangles: the instantiations with triangle / sphere code.  The SPEC shade kernels are MESH instantiations (full 16-byte hit
    // record), so the trace kernels of the split pipeline must write that record for them too: a rectangle-only scene with textures (or any other
    // SPEC feature) counts as "has_tris" -- the compact 4-byte record is for the plain rectangle-only kernels
    for (auto &sh : sc->host.shapes) if (sh.blend_other) rp.has_spec = 2;   // blendbsdf: the instantiations whose BSDF chain loops over two records
    rp.has_tris = bh->n_tris != 0 || has_spheres || rp.has_spec;
    rp.has_analytic = has_spheres ? 1 : 0;
    {   // deep per-mesh traversals diverge: see unstaged_block() in dtof_kernels.hip
        const DShape *dshapes = (const DShape *) (sc->blob.data() + bh->off_shapes);
        rp.n_tlas_nodes = bh->n_tlas_nodes; rp.has_nodes16 = bh->off_nodes16 != 0;
        for (uint32_t i = 0; i < bh->n_shapes; ++i) rp.has_blas |= dshapes[i].kind == SHAPE_MESH && dshapes[i].blas_root != kNoChild;
        if (rp.has_blas && rp.has_nodes16 && !rp.has_analytic) {   // the eight-wave ray kernels of such scenes run as a pair of launches (dtof_kernels.hip: DEFER): 20 bytes per lane of lists
            sc->ws.ensure_defer(); if (n_streams == 2) sc->ws2.ensure_defer();
            Workspace &w1 = n_streams == 2 ? sc->ws2 : sc->ws;
            qs[0].cand = sc->ws.cand.p; qs[0].defer_idx = sc->ws.defer_idx.p; qs[0].defer_cnt = sc->ws.defer_cnt.p;
            qs[1].cand = w1.cand.p; qs[1].defer_idx = w1.defer_idx.p; qs[1].defer_cnt = w1.defer_cnt.p;
        }
    }
    rp.memo_obj = 0xffffffffu;
    {   // instance memo (dtof_traverse.h): pays when there is exactly one instance object, which then nearly every ray visits
        const DObject *dobj = (const DObject *) (sc->blob.data() + bh->off_objects);
        uint32_t n_inst = 0, last_inst = 0;
        for (uint32_t i = 0; i < bh->n_objects; ++i) if (dobj[i].kind == OBJ_INSTANCE) { ++n_inst; last_inst = i; }
        static const bool env_memo = [] { const char *e = getenv("DTOF_INSTANCE_MEMO"); return !(e && e[0] == '0'); }();
        if (fused && n_inst == 1 && env_memo) rp.memo_obj = last_inst;
    }
    {   // a handful of rectangles: test them all instead of walking a tree (trace_flat in dtof_traverse.h; DTOF_FLAT=0 keeps the TLAS)
        static const bool env_flat = [] { const char *e = getenv("DTOF_FLAT"); return !(e && e[0] == '0'); }();
        rp.flat_objects = fused && !rp.has_tris && bh->off_flat != 0 && env_flat ? bh->n_objects : 0u;
        rp.flat_off = bh->off_flat;
    }
    // Resident stage of the fused first-bounce kernel (dtof_kernels.hip, k_shade<..., RESW>): scenes whose blob is too large to stage whole but whose
    // TLAS (at most kResidentNodes nodes, no per-mesh BLAS) and small records fit one CU's LDS beside the stack columns -- Domino: 1 024 nodes, one
    // shared 12-triangle cube, 1 025 instance records that stay in global memory.  DTOF_RESIDENT=0 switches it off, =8 / =12 / =16 set the waves per block.
    ResidentStage resident;
    {
        // 16 waves per CU (4 per SIMD, 128 VGPRs) beat 12 (168 VGPRs) once the nodes come from LDS: 44.2 vs 47.8 ms on Domino (profiles/r03_resident_stage_ab.txt)
        // (round 3: the K = 4 kernels spilled too much at 128 VGPRs -- C5 218 ms with 12 waves, 232 ms with 16; with the shorter traversal code of round 4 it is the other way
        //  round: 192.7 ms with 12 waves, 181.0 with 16, profiles/r04_domino_waves_batch.txt)
        // (the every-BSDF kernels hold more state: 12 waves for one film, 8 for four -- 3.64 ms against 4.07 at 16, 5.15 ms against 5.82 at 12 on a Domino field of rough
        //  plastic cubes; profiles/r03_resident_spec_waves.txt)
        int env_res = [&] { const char *e = getenv("DTOF_RESIDENT"); return e ? atoi(e) : (rp.has_spec ? (rp.n_offsets > 1 ? 8 : 12) : 16); }();   // read per call: tests and A/B runs switch it
        const bool half_env = [] { const char *e = getenv("DTOF_RESIDENT_HALF"); return !(e && e[0] == '0'); }();   // a TLAS of 1 025 .. 2 048 nodes as half-float LDS planes (k_shade: RH16); =0: such scenes take the classic launch
        const uint32_t small_off = bh->off_groups, small_bytes = bh->off_tables - bh->off_groups;          // groups | shapes | emitters | triangles | shading data | intersection records
        if (fused && (env_res == 8 || env_res == 12 || env_res == 16) && rp.has_tris && !rp.has_blas && bh->n_nodes > 0 && (bh->n_nodes <= kResidentNodes || (half_env && bh->off_nodes16 != 0 && bh->n_nodes <= 2 * kResidentNodes)) && blob_bytes > 16 * 1024 &&
            bh->off_shapes > bh->off_groups && bh->off_emitters > bh->off_groups && bh->off_tris > bh->off_groups && bh->off_shading >= bh->off_tris && bh->off_isect >= bh->off_shading && bh->off_tables >= bh->off_isect && small_bytes <= 24 * 1024) {
            resident.small_off = small_off; resident.small_words = (small_bytes + 15) / 16; resident.waves = (uint32_t) env_res;
            rp.res_half = bh->n_nodes > kResidentNodes ? 1u : 0u;
            // the stage must fit the CU's LDS beside the stack columns (a deep TLAS needs many): fewer waves per block while it does not, none if 8 do not either
            const uint32_t limit = device_lds_limit();
            while (resident.waves && resident_lds_bytes(rp, resident, stack_depth, resident.waves) > limit) resident.waves = resident.waves > 8 ? resident.waves - 4 : 0;
        }
    }
    StageTimer tm(stats != nullptr, sc);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    struct EventGuard { hipEvent_t e = nullptr; ~EventGuard() { if (e) (void) hipEventDestroy(e); } } g_fork, g_join;   // released on every exit path
    if (n_streams == 2) { HIP_CHECK(hipEventCreateWithFlags(&g_fork.e, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&g_join.e, hipEventDisableTiming)); }
    hipEvent_t ev_fork = g_fork.e, ev_join = g_join.e;
    if (stats) { memset(stats, 0, sizeof *stats); ev0 = sc->take_event(); ev1 = sc->take_event(); HIP_CHECK(hipEventRecord(ev0, ss[0])); }
    if (n_streams == 2) { HIP_CHECK(hipEventRecord(ev_fork, ss[0])); HIP_CHECK(hipStreamWaitEvent(ss[1], ev_fork, 0)); }
    std::vector<uint64_t> h_counts; std::vector<uint32_t> batch_lanes, batch_iters, batch_inline;   // batch_inline: iterations the first-bounce launch of the batch covered (fused pipeline)
    // per-iteration totals of every batch: sized ONCE (DevBuf::ensure reallocates without copying, and a hipFree in the middle of the
    // frame would also synchronise the device)
    const uint32_t run_passes = lane_dump ? dump_pass + 1 : n_passes;   // a lane dump of pass k needs the stream states passes 0 .. k-1 leave
    const bool deferred = sc->defer_next; sc->defer_next = false;   // dtof_render_rows_async: timings by events, no counters read back, no synchronisation at the end
    if (stats && !deferred && last > first) sc->d_sums.ensure((size_t) ((last - first + batch - 1) / batch) * run_passes * 2 * kMaxIter);
    if (n_passes > 1 && last > first) {   // stream states carried from pass to pass (Sampler::advance keeps the RNGs running, sampler.cpp:52-55)
        sc->d_pass_rng.ensure((size_t) (last - first) * 3);
        rp.pass_rng = sc->d_pass_rng.p; rp.pass_first = (uint32_t) first;
    }
    // the reference's last iteration only looks for emitter hits; without surface emitters it contributes nothing and is skipped --
    // unless further passes follow, whose streams depend on the six draws every active lane makes in it
    // -- and unless a path can still be invalid when it gets there: a BSDF with a null lobe (`mask`, `thindielectric`) leaves valid_ray unset, and a non-null
    // vertex of the last iteration sets it (dopplertofpath.cpp:252-253), which decides whether the path returns what it gathered or 0 (:279-282); the alpha
    // channel / the lane dump's `valid` likewise depend on the hit of that iteration when max_depth is 1
    // (over the whole chain of material records behind a shape -- a blend's partner, the back side of a two-BSDF twosided, and whatever those carry in turn -- so that a
    // nesting the loader learns to accept later cannot slip a null lobe past this test)
    bool has_null_lobe = false;
    for (auto &sh : sc->host.shapes)
        for (const HostShape *m = &sh; m; m = m->blend_other.get())
            has_null_lobe |= m->masked || m->bsdf == BSDF_THINDIELECTRIC || m->bsdf == BSDF_NULL;
    const bool env_fuse_splat = [] { const char *e = getenv("DTOF_FUSE_SPLAT"); return !(e && e[0] == '0'); }();   // read per call: the parity test of the two splat paths switches it
    const bool fuse_splat_ok = env_fuse_splat && fused && !lane_dump && n_passes == 1 && !se.alpha && rp.filter == FILTER_TENT && rp.filter_radius <= 1.f && rp.filter_radius > .5f &&
                               rp.spp_log2 == 6 && rp.n_offsets == 1 && d_film != nullptr;   // exactly one wave per pixel, one film: measured (profiles/r04_fused_splat_ab.txt) -- with more
                               // waves per pixel (C3: 256 spp) or four films (C5) the separate splat kernel, which sums 8 samples per lane before it reduces, is faster
    const bool skip_tail = !has_surface_emitters && n_passes == 1 && !has_null_lobe && !rp.want_valid;

    // The host runs at most two batches ahead of the device: dtof_cancel (Integrator::cancel, integrator.h:96-109) is looked at when a
    // batch is enqueued, so an unbounded run-ahead would leave nothing to cancel once the launches of a long render are queued.
    hipEvent_t batch_done[2] = { sc->take_event(), sc->take_event() };
    uint32_t batch_index = 0;
    for (uint32_t pass = 0; pass < run_passes; ++pass)
    for (uint64_t b0 = first; b0 < last; b0 += batch, ++batch_index) {
        rp.pass = pass;
        const bool dump_now = lane_dump && pass == dump_pass;
        if (batch_index >= 2) HIP_CHECK(hipEventSynchronize(batch_done[batch_index & 1]));
        if (sc->stop.load()) break;
        const Queues &q = qs[batch_index & 1]; hipStream_t s = ss[batch_index & 1];
        rp.lane_base = (uint32_t) b0; rp.n_lanes = (uint32_t) std::min<uint64_t>(batch, last - b0);
        const uint32_t n_seg = segments_for(rp.n_lanes);
        // does iteration 0 of the bounce loop run at all?  (same conditions as the loop head below)
        const bool loop_runs = rp.integrator != INTEGRATOR_VELOCITY && rp.max_depth > 0 && !(1 >= rp.max_depth && skip_tail);
        // fused pipeline: the first bounce kernel generates the lanes and traces the primary rays itself (DTOF_FUSE_FIRST=0 keeps
        // the separate k_generate + k_trace launches)
        const bool first_inline = fused && loop_runs && env_fuse_first;
        int t = -1;
        if (rp.want_valid && !loop_runs && rp.integrator != INTEGRATOR_VELOCITY) HIP_CHECK(hipMemsetAsync(q.valid_out, 0, (size_t) rp.n_lanes * sizeof(float4), s));   // max_depth == 0: { 0, false } (dopplertofpath.cpp:87-88)
        if (!first_inline) {
            t = tm.begin(0, s); launch_generate(rp, q, s); tm.end(0, t, s);
            if (dump_now) launch_lane_dump_rays(rp, q, sc->ws.dbg.p, s);
        }
        const uint32_t *qin = nullptr, *count_in = nullptr; uint32_t it = 0;
        bool fused_splat_done = false;   // the first-bounce kernel of this batch splatted its lanes itself
        if (rp.integrator == INTEGRATOR_VELOCITY) { t = tm.begin(1, s); launch_velocity(blob, blob_bytes, rp, q, stack_depth, s); tm.end(1, t, s); }
        for (;; ++it) {
            if (rp.integrator == INTEGRATOR_VELOCITY) break;
            if (it >= rp.max_depth) break;
            // the last iteration of the reference only looks for emitter hits (dopplertofpath.cpp:136-171);
            // without surface emitters it cannot contribute and is skipped
            if (it + 1 >= rp.max_depth && skip_tail) break;
            if (it >= 8 && (it & 3) == 0) {   // unbounded depth: stop once every segment has drained
                std::vector<uint32_t> alive(n_seg);
                HIP_CHECK(hipMemcpyAsync(alive.data(), count_in, (size_t) n_seg * 4, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                uint64_t sum = 0; for (uint32_t v : alive) sum += v;
                if (sum == 0) break;
            }
            const bool first = first_inline && it == 0;
            // The first-bounce kernel of the fused pipeline runs up to kMaxInline iterations of the loop itself, the path state in registers
            // (RenderParams::inline_iters; DTOF_INLINE_ITERS=1 keeps one launch per iteration).  Multi-pass renders, whose stream states must be
            // in memory between the passes, take one iteration per launch; lane dumps (dtof_sample_lanes) run the same inline kernel as renders.
            // (One compacted launch per iteration instead -- for open scenes, whose paths mostly leave after a bounce -- was built and measured in round 4 and lost:
            // profiles/r04_compaction_ab.txt, tools/experiments/r04_after_hit_compaction.patch.)
            uint32_t span = 1;
            if (first && n_passes == 1) {
                const char *e = getenv("DTOF_INLINE_ITERS");   // read per call: tests switch it
                const int v = e ? atoi(e) : (int) kMaxInline;
                const uint32_t max_inline = (uint32_t) (v < 1 ? 1 : v > (int) kMaxInline ? (int) kMaxInline : v);
                while (span < max_inline && (it + span) < rp.max_depth && !(it + span + 1 >= rp.max_depth && skip_tail)) ++span;   // the loop head's conditions for iteration it + span
            }
            rp.inline_iters = span;
            it += span - 1;   // `it` is now the last iteration this launch covers
            {   // small frames whose whole path runs inline: one block per 64-lane chunk (8 x the waves); the count slots it adds into are zeroed first
                const bool whole_path = first && !((it + 1 < rp.max_depth) && !(it + 2 >= rp.max_depth && skip_tail));
                const uint32_t env_chunk_segs = [] { const char *e = getenv("DTOF_CHUNK_SEGS"); return e ? (uint32_t) atoi(e) : 8192u; }();   // frames up to this many segments (A/B switch; read per call: tests switch it)
                rp.chunk_blocks = whole_path && n_seg <= env_chunk_segs ? kChunkBlocks : 1u;
                // The resident kernel's waves take their work from a counter; a launch ends with a tail in which they run out one after the other, as long as the last unit
                // they started (a 512-lane segment = eight chunks of three bounces: ~0.7 ms on Domino).  When nothing is compacted for a later launch the unit can be a part
                // of a segment (DTOF_RES_UNITS = units per segment, 1 / 2 / 4 / 8; the statistics slots are then added into, like those of the chunked small frames).
                // MEASURED and OFF (profiles/r05_resident_units.txt): the shorter tail does not pay for the units' overhead on one GPU -- C4 33.20 ms at one unit per segment,
                // 34.13 at two, 35.98 at four; C5 170.8 against 179.3 ms at four.  Kept as a switch for launches a fraction of this size (a frame sharded over many GPUs).
                const uint32_t env_units = [] { const char *e = getenv("DTOF_RES_UNITS"); const int v = e ? atoi(e) : 1; return (uint32_t) (v == 1 || v == 2 || v == 4 || v == 8 ? v : 1); }();
                rp.res_units = whole_path ? env_units : 1u;
                if (rp.chunk_blocks > 1 || (rp.res_units > 1 && resident.waves)) HIP_CHECK(hipMemsetAsync(q.counts, 0, (size_t) 2 * (it + 1) * n_seg * 4, s));
            }
            // does iteration it+1 run?  (same conditions as the loop head)
            const bool next_runs = (it + 1 < rp.max_depth) && !(it + 2 >= rp.max_depth && skip_tail);
            // Fused splat: the first-bounce launch covers the whole path and every wave holds the 64 samples of one pixel -- it reduces their footprint values itself
            // and adds them to the film (k_shade; tent filter with a 3 x 3 footprint, 64 samples per pixel, one film, one pass, no alpha film, no lane dump).  The result then
            // never goes through q.res / q.pos and k_splat_x8's round trip through HBM.  DTOF_FUSE_SPLAT=0 keeps the splat kernel (A/B).
            const bool splat_here = first && !next_runs && fuse_splat_ok && rp.chunk_blocks <= kChunkBlocks;
            if (!fused || (it == 0 && !first)) { t = tm.begin(1, s); launch_trace(blob, blob_bytes, rp, q, qin, count_in, stack_depth, s); tm.end(1, t, s); if (stats) stats->n_launches_trace++; }
            // per-iteration count slots; beyond kMaxIter iterations (unbounded depth, paths that russian roulette keeps alive that long)
            // the slots are reused -- only the statistics lose those iterations, no path is cut short
            uint32_t *qout = q.q[it & 1], *alive_out = q.counts + (size_t) (2 * (it % kMaxIter)) * n_seg, *shadow_out = alive_out + n_seg;
            const int st_shade = first ? 5 : 2;
            t = tm.begin(st_shade, s); launch_shade(blob, blob_bytes, rp, q, qin, count_in, qout, alive_out, shadow_out, it + 1 - span, fused, next_runs, stack_depth, s, first, first && dump_now ? sc->ws.dbg.p : nullptr, &resident, splat_here ? d_film : nullptr, film_stride); tm.end(st_shade, t, s);
            fused_splat_done |= splat_here;
            if (stats && splat_here) stats->n_fused_splat_launches++;
            if (stats && first) { stats->n_launches_first++; stats->n_inline_iterations += span; batch_inline.push_back(span); }
            if (!fused) { t = tm.begin(3, s); launch_shadow(blob, blob_bytes, rp, q, shadow_out, stack_depth, s); tm.end(3, t, s); if (stats) stats->n_launches_shadow++; }
            if (stats) stats->n_launches_shade++;
            qin = qout; count_in = alive_out;
        }
        if (n_passes > 1 && pass + 1 < run_passes) launch_pass_save(rp, q, s);
        if (dump_now) {
            launch_lane_dump(rp, q, sc->ws.dbg.p, s);
            HIP_CHECK(hipMemcpyAsync(lane_dump + (b0 - first), sc->ws.dbg.p, (size_t) rp.n_lanes * sizeof(LaneDebug), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
        } else if (!lane_dump && !fused_splat_done) {
            t = tm.begin(4, s); launch_splat(rp, q, d_film, film_stride, s);
            if (se.alpha) {   // the alpha film (plane K behind the K offset films): the same splat over (valid, 0, 0) -- ImageBlock::put of aovs[3] (integrator.cpp:528-533)
                RenderParams ra = rp; ra.n_offsets = 1;
                Queues qa = q; qa.res = q.valid_out;
                launch_splat(ra, qa, d_film + (size_t) rp.n_offsets * film_stride, film_stride, s);
            }
            tm.end(4, t, s);
        }
        HIP_CHECK(hipGetLastError());   // a rejected launch (LDS size, launch bounds, grid) must not pass for an empty film
        HIP_CHECK(hipEventRecord(batch_done[batch_index & 1], s));
        if (stats) {   // per-iteration totals of this batch are reduced on the device; one small copy after the last batch
            const uint32_t it_counted = std::min<uint32_t>(it, kMaxIter);
            if (it_counted && !deferred) launch_sum_counts(q.counts, n_seg, 2 * it_counted, sc->d_sums.p + (size_t) batch_index * 2 * kMaxIter, s);
            batch_lanes.push_back(rp.n_lanes); batch_iters.push_back(it_counted);
            stats->n_batches++;
        }
    }
    if (n_streams == 2) { HIP_CHECK(hipEventRecord(ev_join, ss[1])); HIP_CHECK(hipStreamWaitEvent(ss[0], ev_join, 0)); }
    hipStream_t s = ss[0];
    if (stats && deferred) {
        HIP_CHECK(hipEventRecord(ev1, s));
        dtof_scene::DeferredFrame f; f.ev0 = ev0; f.ev1 = ev1; f.counters = *stats;
        for (int k = 0; k < 6; ++k) f.ev[k] = tm.ev[k];
        for (uint32_t b : batch_lanes) f.counters.n_paths += b;
        sc->deferred.push_back(std::move(f));
        if (sc->stop.load()) throw std::runtime_error("cancelled");
        return;
    }
    if (stats) {
        HIP_CHECK(hipEventRecord(ev1, s)); HIP_CHECK(hipEventSynchronize(ev1));
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1)); stats->ms_total = ms;
        std::vector<unsigned long long> sums(batch_lanes.size() * 2 * (size_t) kMaxIter);
        if (!sums.empty()) HIP_CHECK(hipMemcpy(sums.data(), sc->d_sums.p, sums.size() * 8, hipMemcpyDeviceToHost));
        for (size_t b = 0; b < batch_lanes.size(); ++b)
            for (uint32_t i = 0; i < 2 * batch_iters[b]; ++i) h_counts.push_back(sums[b * 2 * kMaxIter + i]);
        stats->ms_generate = tm.total(0); stats->ms_trace = tm.total(1); stats->ms_first = tm.total(5); stats->ms_shade = tm.total(2) + stats->ms_first;
        stats->ms_shadow = tm.total(3); stats->ms_splat = tm.total(4);
        size_t off = 0;
        for (size_t b = 0; b < batch_lanes.size(); ++b) {
            stats->n_paths += batch_lanes[b];
            uint64_t in = batch_lanes[b];
            for (uint32_t i = 0; i < batch_iters[b]; ++i) {
                stats->n_bounces += in; stats->n_shadow_rays += h_counts[off + 2 * i + 1];
                if (b < batch_inline.size() && i < batch_inline[b]) stats->n_bounces_inline += in;
                in = h_counts[off + 2 * i];
            }
            off += 2 * (size_t) batch_iters[b];
        }
    } else {
        HIP_CHECK(hipStreamSynchronize(s));
    }
    if (sc->stop.load()) throw std::runtime_error("cancelled");
}

std::map<std::string, std::string> to_map(const char *const *names, const char *const *values, int n) {
    std::map<std::string, std::string> m;
    for (int i = 0; i < n; ++i) m[names[i]] = values[i];
    return m;
}

PropBag make_bag(const char *plugin, const char *const *names, const char *types, const char *const *values, int n) {
    PropBag b; b.plugin = plugin ? plugin : "";
    for (int i = 0; i < n; ++i) {
        PropValue v; std::string val = values[i];
        switch (types[i]) {
            case 'f': v.type = PropValue::Float; v.f = std::stod(val); break;
            case 'i': v.type = PropValue::Int; v.i = std::stoll(val); break;
            case 'b': v.type = PropValue::Bool; if (val != "true" && val != "false") throw std::runtime_error("could not parse boolean value \"" + val + "\""); v.b = val == "true"; break;
            case 's': v.type = PropValue::String; v.s = val; break;
            default: throw std::runtime_error(std::string("unknown property type '") + types[i] + "'");
        }
        b.values[names[i]] = v;
    }
    return b;
}

dtof_scene *finish_scene(HostScene &&hs) {
    auto sc = new dtof_scene();
    try {
        sc->host = std::move(hs);
        if (!sc->host.has_sensor && sc->host.sampler.plugin.empty()) sc->host.sampler.plugin = "independent";   // no sensor, no sampler: Sensor's default (sensor.cpp:63-66)
        sc->pp = make_plugin_params(sc->host.integrator, sc->host.sampler);
        {   // the hit record packs (object, shape in its group) into 32 bits (Queues::hit_id): the object index gets 24 bits unless a shapegroup
            // needs more than the remaining 8 for its shapes; 0xffffffff stays free as the "miss" value
            uint32_t max_shapes = 1; for (auto &g : sc->host.groups) max_shapes = std::max(max_shapes, g.n_shapes);
            uint32_t shape_bits = 0; while ((1ull << shape_bits) < max_shapes) ++shape_bits;
            uint32_t obj_bits = 1; while ((1ull << obj_bits) < sc->host.objects.size() + 1ull) ++obj_bits;
            if (obj_bits + shape_bits > 31) throw std::runtime_error("too many scene objects / shapes per shapegroup: object index and shape index must fit 31 bits together");
            sc->id_shift = shape_bits <= 8 && obj_bits <= 24 ? 24 : 31 - shape_bits;
        }
        sc->blob = build_scene_blob(sc->host);
    } catch (...) { delete sc; throw; }
    return sc;
}

}  // namespace

// ================================================================================ C ABI
extern "C" {

const char *dtof_version(void) { return "dtof 0.1 (HIP, gfx950; dopplertofpath + correlated)"; }
const char *dtof_last_error(void) { return g_last_error.c_str(); }

int dtof_scene_load_string(const char *xml, const char *const *pn, const char *const *pv, int n, dtof_scene **out) {
    return guarded([&] {
        if (!xml || !out) throw std::runtime_error("null argument");
        *out = finish_scene(load_scene_xml(xml, to_map(pn, pv, n)));
    });
}
int dtof_scene_load_file(const char *path, const char *const *pn, const char *const *pv, int n, dtof_scene **out) {
    return guarded([&] {
        if (!path || !out) throw std::runtime_error("null argument");
        std::string p = path, dir = ".";
        size_t k = p.find_last_of('/'); if (k != std::string::npos) dir = k ? p.substr(0, k) : "/";
        *out = finish_scene(load_scene_xml(read_file(path), to_map(pn, pv, n), dir));
    });
}
void dtof_scene_destroy(dtof_scene *scene) { delete scene; }

int dtof_scene_set_integrator(dtof_scene *sc, const char *plugin, const char *const *names, const char *types, const char *const *values, int n) {
    return guarded([&] {
        if (!sc) throw std::runtime_error("null scene");
        PropBag b = make_bag(plugin, names, types, values, n);
        PluginParams p = make_plugin_params(b, sc->host.sampler);
        sc->host.integrator = b; sc->pp = p;
    });
}
int dtof_scene_set_sampler(dtof_scene *sc, const char *plugin, const char *const *names, const char *types, const char *const *values, int n) {
    return guarded([&] {
        if (!sc) throw std::runtime_error("null scene");
        PropBag b = make_bag(plugin, names, types, values, n);
        PluginParams p = make_plugin_params(sc->host.integrator, b);
        sc->host.sampler = b; sc->pp = p;
    });
}

struct dtof_integrator { PropBag bag; };
struct dtof_sampler_plugin { PropBag bag; };
static PropBag default_sampler_bag() { PropBag b; b.plugin = "correlated"; return b; }
static PropBag default_integrator_bag() { PropBag b; b.plugin = "dopplertofpath"; return b; }
int dtof_integrator_create(const char *plugin, const char *const *names, const char *types, const char *const *values, int n, dtof_integrator **out) {
    return guarded([&] {
        if (!out) throw std::runtime_error("null argument");
        PropBag b = make_bag(plugin, names, types, values, n);
        (void) make_plugin_params(b, default_sampler_bag());   // the constructor's checks (names, types, value ranges)
        *out = new dtof_integrator { b };
    });
}
void dtof_integrator_destroy(dtof_integrator *i) { delete i; }
int dtof_sampler_plugin_create(const char *plugin, const char *const *names, const char *types, const char *const *values, int n, dtof_sampler_plugin **out) {
    return guarded([&] {
        if (!out) throw std::runtime_error("null argument");
        PropBag b = make_bag(plugin, names, types, values, n);
        (void) make_plugin_params(default_integrator_bag(), b);
        *out = new dtof_sampler_plugin { b };
    });
}
void dtof_sampler_plugin_destroy(dtof_sampler_plugin *s) { delete s; }
int dtof_integrator_render(const dtof_integrator *integ, const dtof_sampler_plugin *smp, dtof_scene *sc, uint32_t sensor_index,
                           uint32_t seed, uint32_t spp, float *out_rgb, dtof_render_stats *stats) {
    int rc = guarded([&] {
        if (!integ || !sc) throw std::runtime_error("null argument");
        const PropBag &sb = smp ? smp->bag : sc->host.sampler;
        PluginParams p = make_plugin_params(integ->bag, sb);
        sc->host.integrator = integ->bag; sc->host.sampler = sb; sc->pp = p;
    });
    return rc ? rc : dtof_render(sc, sensor_index, seed, spp, out_rgb, stats);
}

int dtof_scene_get_info(const dtof_scene *sc, dtof_scene_info *info) {
    return guarded([&] {
        if (!sc || !info) throw std::runtime_error("null argument");
        const HostSensor &se = sc->host.sensor; const PluginParams &p = sc->pp;
        const BlobHeader *h = (const BlobHeader *) sc->blob.data();
        memset(info, 0, sizeof *info);
        info->film_width = se.film_w; info->film_height = se.film_h; info->crop_x = se.crop_x; info->crop_y = se.crop_y;
        info->crop_width = se.crop_w; info->crop_height = se.crop_h; info->sample_count = p.sample_count;
        info->n_shapes = h->n_shapes; info->n_groups = h->n_groups; info->n_objects = h->n_objects; info->n_emitters = h->n_emitters;
        info->n_triangles = h->n_tris; info->n_bvh_nodes = h->n_nodes; info->scene_blob_bytes = h->total_bytes;
        info->time = p.time; info->w_g = p.w_g_mhz; info->g_1 = p.g_1; info->g_0 = p.g_0; info->w_s = p.w_s_mhz;
        info->phase_offset = p.phase_offset; info->hetero_frequency = p.hetero_frequency; info->antithetic_shift = p.antithetic_shift;
        info->wave_type = p.wave_type; info->low_frequency_component_only = p.low_frequency_component_only;
        info->time_sampling = p.time_sampling; info->stratify_each_interval = p.stratify_each_interval;
        info->path_correlation_depth = p.path_correlation_depth; info->max_depth = p.max_depth; info->rr_depth = p.rr_depth;
        info->base_seed = p.base_seed; info->time_correlate_number = p.time_correlate_number; info->path_correlate_number = p.path_correlate_number;
        info->bvh_stack_depth = h->tlas_depth;
        info->filter_radius = se.filter_radius;
        info->filter_halo = se.filter == FILTER_BOX ? 0 : (int32_t) std::ceil(se.filter_radius - .5f);
        info->has_alpha = se.alpha ? 1 : 0;
    });
}

int dtof_scene_export(const dtof_scene *sc, int kind, float *out, size_t cap, size_t *n_written) {
    return guarded([&] {
        if (!sc || !n_written) throw std::runtime_error("null argument");
        std::vector<float> v;
        if (kind == 0) for (auto &o : sc->host.objects) {
            v.push_back(o.key_time[0]); v.push_back(o.key_time[1]);
            v.insert(v.end(), o.key[0], o.key[0] + 16); v.insert(v.end(), o.key[1], o.key[1] + 16);
        } else if (kind == 1) for (auto &s : sc->host.shapes) {
            v.insert(v.end(), s.to_world, s.to_world + 16); v.insert(v.end(), s.to_object, s.to_object + 16);
        } else if (kind == 2) {
            const HostSensor &s = sc->host.sensor;
            v.insert(v.end(), s.to_world, s.to_world + 16);
            v.push_back(s.x_fov); v.push_back(s.near_clip); v.push_back(s.far_clip); v.push_back(s.shutter_open); v.push_back(s.shutter_close);
            v.push_back(s.orthographic ? 2.f : s.thinlens ? 1.f : 0.f); v.push_back(s.aperture_radius); v.push_back(s.focus_distance);
        } else if (kind == 3) for (auto &e : sc->host.emitters) {
            v.insert(v.end(), e.pos, e.pos + 3); v.insert(v.end(), e.intensity, e.intensity + 3);
        } else if (kind >= 4 && kind <= 7) for (auto &s : sc->host.shapes) {
            if (s.kind != SHAPE_MESH) continue;
            if (kind == 4) v.insert(v.end(), s.positions.begin(), s.positions.end());
            else if (kind == 5) v.insert(v.end(), s.normals.begin(), s.normals.end());
            else if (kind == 6) v.insert(v.end(), s.texcoords.begin(), s.texcoords.end());
            else for (uint32_t f : s.faces) { float b; memcpy(&b, &f, 4); v.push_back(b); }
        } else if (kind == 8) for (auto &s : sc->host.shapes) {
            if (s.kind != SHAPE_SPHERE) continue;
            v.insert(v.end(), s.center, s.center + 3); v.push_back(s.radius); v.push_back(s.sphere_inv_area); v.push_back(s.flip_normals ? 1.f : 0.f);
        } else if (kind == 9) for (auto &s : sc->host.shapes) {
            v.push_back((float) s.bsdf); v.push_back(s.twosided ? 1.f : 0.f); v.push_back(s.diel_eta); v.push_back(s.nonlinear ? 1.f : 0.f);
            v.push_back(s.inv_eta_2); v.push_back(s.fdr_int); v.push_back(s.spec_sampling_weight);
            v.insert(v.end(), s.refl, s.refl + 3); v.insert(v.end(), s.spec_refl, s.spec_refl + 3); v.insert(v.end(), s.spec_trans, s.spec_trans + 3);
            v.insert(v.end(), s.cond_eta, s.cond_eta + 3); v.insert(v.end(), s.cond_k, s.cond_k + 3); v.push_back(s.alpha_u); v.push_back(s.alpha_v);
        } else if (kind == 13) for (auto &t : sc->host.textures) {
            v.push_back((float) t.kind); v.push_back((float) t.filter); v.push_back((float) t.wrap); v.push_back((float) t.channels);
            v.push_back((float) t.width); v.push_back((float) t.height);
            v.insert(v.end(), t.to_uv, t.to_uv + 4); v.insert(v.end(), t.color0, t.color0 + 3); v.insert(v.end(), t.color1, t.color1 + 3); v.push_back(t.mean);
        } else if (kind == 14) for (auto &t : sc->host.textures) v.insert(v.end(), t.data.begin(), t.data.end());
        else if (kind == 15) for (auto &s : sc->host.shapes) v.push_back((float) s.tex_refl);
        else if (kind == 17) for (auto &s : sc->host.shapes) v.push_back(s.sample_all ? 1.f : 0.f);
        else if (kind == 19) for (auto &s : sc->host.shapes) { v.push_back((float) s.tex_spec); v.push_back((float) s.tex_trans); v.push_back((float) s.tex_alpha_u); v.push_back((float) s.tex_alpha_v); }   // textures on the other slots: indices into the texture table, -1 = none
        else if (kind == 20) for (auto &s : sc->host.shapes) { v.push_back(s.masked ? 1.f : 0.f); v.push_back(s.opacity); v.push_back((float) s.tex_opacity); }   // mask: masked, opacity, its texture
        else if (kind == 21) for (auto &s : sc->host.shapes) v.push_back((float) s.tex_normal);   // normalmap / bumpmap: its texture, -1 = none
        else if (kind == 22) for (auto &s : sc->host.shapes) { v.push_back(s.bumpmap ? 1.f : 0.f); v.push_back(s.bump_scale); }   // bumpmap: is one, scale
        else if (kind == 24) for (auto &s : sc->host.shapes) v.push_back((float) s.tex_radiance);   // texture on the area emitter's radiance, -1 = a constant
        else if (kind == 23) for (auto &s : sc->host.shapes) {   // blendbsdf: is one, weight, its texture, kind and two-sidedness of bsdf_1
            v.push_back(s.blend_other ? (s.two_bsdfs ? 2.f : 1.f) : 0.f); v.push_back(s.blend_weight); v.push_back((float) s.tex_blend);   // 1 blendbsdf, 2 twosided with two BSDFs
            v.push_back(s.blend_other ? (float) s.blend_other->bsdf : -1.f); v.push_back(s.blend_other && s.blend_other->twosided ? 1.f : 0.f);
        }
        else if (kind == 18) for (auto &e : sc->host.emitters) {   // every emitter: kind, pos, intensity, first row of to_local (directional: its direction)
            v.push_back((float) e.kind); v.insert(v.end(), e.pos, e.pos + 3); v.insert(v.end(), e.intensity, e.intensity + 3); v.insert(v.end(), e.to_local, e.to_local + 3);
        }
        else if (kind == 16) {   // the environment map as packed into the blob: header words, m_data, then every level of the hierarchical warp
            const BlobHeader *bh = (const BlobHeader *) sc->blob.data();
            const DEmitter *de = (const DEmitter *) (sc->blob.data() + bh->off_emitters);
            for (uint32_t i = 0; i < bh->n_emitters; ++i) if (de[i].kind == EMITTER_ENVMAP) {
                const DEnvmap *e = (const DEnvmap *) (sc->blob.data() + de[i].shape);
                v.push_back((float) e->w); v.push_back((float) e->h); v.push_back((float) e->n_levels); v.push_back(e->scale);
                v.insert(v.end(), de[i].pos, de[i].pos + 3); v.push_back(de[i].cutoff_angle);
                v.insert(v.end(), e->to_world, e->to_world + 12); v.insert(v.end(), de[i].to_local, de[i].to_local + 12);
                const float *d = (const float *) (sc->blob.data() + e->data_off);
                v.insert(v.end(), d, d + (size_t) e->w * e->h * 3);
                for (uint32_t k = 0; k < e->n_levels; ++k) {
                    const uint32_t end = k + 1 < e->n_levels ? e->level_off[k + 1] : bh->total_bytes;
                    const float *lv = (const float *) (sc->blob.data() + e->level_off[k]);
                    size_t count = k == 0 ? (size_t) e->w * e->h : 0;
                    if (k > 0) { uint32_t lx = e->w - 1, ly = e->h - 1; for (uint32_t j = 1; j <= k; ++j) { lx += lx & 1u; ly += ly & 1u; if (j < k) { lx >>= 1; ly >>= 1; } } count = (size_t) lx * ly; }
                    (void) end;
                    v.push_back((float) e->level_w[k]); v.push_back((float) count);
                    v.insert(v.end(), lv, lv + count);
                }
            }
        }
        else if (kind == 12) for (auto &s : sc->host.shapes) {
            v.push_back(s.beckmann ? 0.f : 1.f);
        } else if (kind == 10) for (auto &s : sc->host.shapes) {
            if (s.bsdf == BSDF_ROUGHPLASTIC) v.insert(v.end(), s.rough_table.begin(), s.rough_table.end());
        } else if (kind == 11) for (auto &e : sc->host.emitters) {
            if (e.kind != EMITTER_SPOT) continue;
            v.insert(v.end(), e.pos, e.pos + 3); v.insert(v.end(), e.intensity, e.intensity + 3); v.insert(v.end(), e.to_local, e.to_local + 12);
            v.push_back(e.cutoff_angle); v.push_back(e.cos_cutoff); v.push_back(e.cos_beam); v.push_back(e.inv_transition);
        } else throw std::runtime_error("unknown export kind");
        *n_written = v.size();
        if (out) { if (v.size() > cap) throw std::runtime_error("export buffer too small"); memcpy(out, v.data(), v.size() * 4); }
    });
}

int dtof_render_rows(dtof_scene *sc, uint32_t seed, uint32_t spp, int32_t row_begin, int32_t row_end,
                     const float *offsets, int n_offsets, float *d_film, dtof_render_stats *stats) {
    return guarded([&] {
        if (!sc || !d_film) throw std::runtime_error("null argument");
        sc->stop = false;
        sc->film_stride_call = caller_film_stride(sc, n_offsets);
        render_rows(sc, seed, spp, row_begin, row_end, offsets, n_offsets, d_film, stats);
    });
}

int dtof_render_rows_async(dtof_scene *sc, uint32_t seed, uint32_t spp, int32_t row_begin, int32_t row_end, const float *offsets, int n_offsets, float *d_film) {
    return guarded([&] {
        if (!sc || !d_film) throw std::runtime_error("null argument");
        sc->stop = false;
        dtof_render_stats local;
        sc->defer_next = true;
        sc->film_stride_call = caller_film_stride(sc, n_offsets);
        try { render_rows(sc, seed, spp, row_begin, row_end, offsets, n_offsets, d_film, &local); }
        catch (...) { sc->defer_next = false; if (sc->deferred.empty()) sc->events_used = 0; throw; }   // the events the failed frame took go back to the pool
    });
}
int dtof_scene_set_film_layout(dtof_scene *sc, int32_t planes, uint64_t plane_stride_floats) {
    return guarded([&] {
        if (!sc) throw std::runtime_error("null scene");
        if (planes < 0) throw std::runtime_error("negative plane count");
        const HostSensor &se = sc->host.sensor;
        if (plane_stride_floats != 0 && (plane_stride_floats % 4 != 0 || plane_stride_floats < (uint64_t) se.crop_w * 4))
            throw std::runtime_error("plane stride must be a multiple of 4 floats and at least one film row");
        sc->film_planes = planes; sc->film_plane_stride = plane_stride_floats;
    });
}
int dtof_scene_set_stream(dtof_scene *sc, void *hip_stream) {
    return guarded([&] {
        if (!sc) throw std::runtime_error("null scene");
        if (!sc->deferred.empty()) throw std::runtime_error("frames are still in flight on the current stream: call dtof_async_collect first");
        ensure_device(sc);
        // nothing of ours is left behind on the stream we leave.  A FOREIGN stream may already be gone (the caller's torch stream was collected before it handed the stream
        // back): no frame is in flight on it (checked above), so a failing wait there means a dead handle, not lost work -- fall through to the new stream
        if (hipStreamSynchronize(sc->stream) != hipSuccess) {
            (void) hipGetLastError();
            if (sc->stream == sc->own_stream) throw std::runtime_error("hipStreamSynchronize failed on the scene's own stream");
        }
        sc->stream = hip_stream ? (hipStream_t) hip_stream : sc->own_stream;
    });
}
int dtof_render_stripes_async(dtof_scene *sc, uint32_t seed, uint32_t spp, int32_t first_row, int32_t stripe_rows, int32_t stripe_period,
                              const float *offsets, int n_offsets, float *d_film) {
    return guarded([&] {
        if (!sc || !d_film) throw std::runtime_error("null argument");
        if (first_row < 0 || stripe_rows <= 0 || stripe_period < stripe_rows) throw std::runtime_error("invalid stripe layout");
        sc->stop = false;
        dtof_render_stats local;
        sc->defer_next = true;
        sc->film_stride_call = caller_film_stride(sc, n_offsets);
        try { render_rows(sc, seed, spp, first_row, sc->host.sensor.crop_h, offsets, n_offsets, d_film, &local, nullptr, 0, 0, (uint32_t) stripe_rows, (uint32_t) stripe_period); }
        catch (...) { sc->defer_next = false; if (sc->deferred.empty()) sc->events_used = 0; throw; }
    });
}
int dtof_clear_async(dtof_scene *sc, void *d_ptr, size_t bytes) {
    return guarded([&] {
        if (!sc || !d_ptr) throw std::runtime_error("null argument");
        ensure_device(sc);
        HIP_CHECK(hipMemsetAsync(d_ptr, 0, bytes, sc->stream));
    });
}
int dtof_develop_async(dtof_scene *sc, const float *d_film, float *d_rgb, int64_t n_pixels) {
    return guarded([&] {
        if (!sc || !d_film || !d_rgb) throw std::runtime_error("null argument");
        ensure_device(sc);
        launch_develop(d_film, d_rgb, n_pixels, sc->stream);
        HIP_CHECK(hipGetLastError());
    });
}
int dtof_async_collect(dtof_scene *sc, dtof_render_stats *sum, double *frame_ms, uint32_t capacity, uint32_t *n_frames) {
    return guarded([&] {
        if (!sc || !sum || !n_frames) throw std::runtime_error("null argument");
        ensure_device(sc);
        HIP_CHECK(hipStreamSynchronize(sc->stream));
        memset(sum, 0, sizeof *sum);
        *n_frames = (uint32_t) sc->deferred.size();
        auto total = [](const std::vector<std::pair<hipEvent_t, hipEvent_t>> &v) { double ms = 0; for (auto &p : v) { float t = 0; HIP_CHECK(hipEventElapsedTime(&t, p.first, p.second)); ms += t; } return ms; };
        uint32_t i = 0;
        for (auto &f : sc->deferred) {
            float t = 0; HIP_CHECK(hipEventElapsedTime(&t, f.ev0, f.ev1));
            if (frame_ms && i < capacity) frame_ms[i] = t;
            ++i;
            sum->ms_total += t;
            sum->ms_generate += total(f.ev[0]); sum->ms_trace += total(f.ev[1]); sum->ms_first += total(f.ev[5]); sum->ms_shade += total(f.ev[2]) + total(f.ev[5]);
            sum->ms_shadow += total(f.ev[3]); sum->ms_splat += total(f.ev[4]);
            sum->n_paths += f.counters.n_paths; sum->n_batches += f.counters.n_batches;
            sum->n_launches_trace += f.counters.n_launches_trace; sum->n_launches_shade += f.counters.n_launches_shade; sum->n_launches_shadow += f.counters.n_launches_shadow;
            sum->n_launches_first += f.counters.n_launches_first; sum->n_inline_iterations += f.counters.n_inline_iterations; sum->n_fused_splat_launches += f.counters.n_fused_splat_launches;
        }
        sc->deferred.clear(); sc->events_used = 0;
    });
}

int dtof_render_stripes(dtof_scene *sc, uint32_t seed, uint32_t spp, int32_t first_row, int32_t stripe_rows, int32_t stripe_period,
                        const float *offsets, int n_offsets, float *d_film, dtof_render_stats *stats) {
    return guarded([&] {
        if (!sc || !d_film) throw std::runtime_error("null argument");
        if (first_row < 0 || stripe_rows <= 0 || stripe_period < stripe_rows) throw std::runtime_error("invalid stripe layout");
        sc->stop = false;
        sc->film_stride_call = caller_film_stride(sc, n_offsets);
        render_rows(sc, seed, spp, first_row, sc->host.sensor.crop_h, offsets, n_offsets, d_film, stats, nullptr, 0, 0, (uint32_t) stripe_rows, (uint32_t) stripe_period);
    });
}

int dtof_develop(const float *d_film, float *d_rgb, int64_t n_pixels) {
    return guarded([&] {
        if (!d_film || !d_rgb) throw std::runtime_error("null argument");
        launch_develop(d_film, d_rgb, n_pixels, nullptr);
        HIP_CHECK(hipGetLastError()); HIP_CHECK(hipStreamSynchronize(nullptr));
    });
}

int dtof_render_offsets(dtof_scene *sc, uint32_t seed, uint32_t spp, const float *offsets, int n_offsets, float *out_rgb, dtof_render_stats *stats) {
    return guarded([&] {
        if (!sc || !out_rgb) throw std::runtime_error("null argument");
        ensure_device(sc);
        sc->stop = false;
        int k = n_offsets <= 0 ? 1 : n_offsets;
        const HostSensor &se = sc->host.sensor;
        size_t px = (size_t) se.crop_w * se.crop_h;
        const int planes = k + (se.alpha ? 1 : 0), ch = se.alpha ? 4 : 3;   // rgba: one more film plane for the alpha channel, four channels out
        sc->d_film.ensure(px * 4 * planes); sc->d_rgb.ensure(px * ch * k);
        HIP_CHECK(hipMemsetAsync(sc->d_film.p, 0, px * 4 * planes * sizeof(float), sc->stream));
        sc->film_stride_call = px * 4;   // the library's own film
        render_rows(sc, seed, spp, 0, se.crop_h, offsets, n_offsets, sc->d_film.p, stats);
        if (se.alpha) for (int i = 0; i < k; ++i) launch_develop_rgba(sc->d_film.p + px * 4 * i, sc->d_film.p + px * 4 * k, sc->d_rgb.p + px * 4 * i, (int64_t) px, sc->stream);
        else launch_develop(sc->d_film.p, sc->d_rgb.p, (int64_t) px * k, sc->stream);
        HIP_CHECK(hipMemcpyAsync(out_rgb, sc->d_rgb.p, px * ch * k * sizeof(float), hipMemcpyDeviceToHost, sc->stream));
        HIP_CHECK(hipStreamSynchronize(sc->stream));
    });
}

int dtof_render(dtof_scene *sc, uint32_t sensor_index, uint32_t seed, uint32_t spp, float *out_rgb, dtof_render_stats *stats) {
    if (sensor_index != 0) { g_last_error = "Scene::render(): sensor index " + std::to_string(sensor_index) + " is out of bounds!"; return DTOF_ERR_INVALID; }
    return dtof_render_offsets(sc, seed, spp, nullptr, 0, out_rgb, stats);
}

void dtof_cancel(dtof_scene *sc) { if (sc) sc->stop = true; }

int dtof_sample_lanes_valid(dtof_scene *sc, uint32_t seed, uint32_t spp, uint64_t lane_begin, uint64_t n, float *out, uint32_t *valid) {
    return guarded([&] {
        if (!sc || !out) throw std::runtime_error("null argument");
        static_assert(sizeof(LaneDebug) == 52, "LaneDebug is 13 floats");
        sc->stop = false;
        if (n == 0) return;
        std::vector<LaneDebug> lanes(n);
        render_rows(sc, seed, spp, 0, 0, nullptr, 0, nullptr, nullptr, lanes.data(), lane_begin, n);
        for (uint64_t i = 0; i < n; ++i) { memcpy(out + 12 * i, &lanes[i], 48); if (valid) valid[i] = lanes[i].valid != 0.f ? 1u : 0u; }
    });
}
int dtof_sample_lanes(dtof_scene *sc, uint32_t seed, uint32_t spp, uint64_t lane_begin, uint64_t n, float *out) {
    return dtof_sample_lanes_valid(sc, seed, spp, lane_begin, n, out, nullptr);
}
int dtof_develop_on_stream(const float *d_film, float *d_rgb, int64_t n_pixels, void *hip_stream) {
    return guarded([&] {
        if (!d_film || !d_rgb) throw std::runtime_error("null argument");
        launch_develop(d_film, d_rgb, n_pixels, (hipStream_t) hip_stream);
        HIP_CHECK(hipGetLastError());
    });
}
int dtof_develop_rgba(const float *d_film, const float *d_alpha_film, float *d_rgba, int64_t n_pixels) {
    return guarded([&] {
        if (!d_film || !d_alpha_film || !d_rgba) throw std::runtime_error("null argument");
        launch_develop_rgba(d_film, d_alpha_film, d_rgba, n_pixels, nullptr);
        HIP_CHECK(hipGetLastError()); HIP_CHECK(hipStreamSynchronize(nullptr));
    });
}

// ---------------------------------------------------------------- sampler
static RenderParams sampler_params(const dtof_sampler *s) {
    RenderParams rp; memset(&rp, 0, sizeof rp);
    rp.base_seed = s->base_seed; rp.seed = s->seed; rp.seed_value = s->base_seed + s->seed;
    rp.spp = s->spw; rp.tcn = (uint32_t) s->tcn; rp.pcn = (uint32_t) s->pcn;
    rp.n_stratum = s->sample_count / (uint32_t) s->tcn;
    rp.inv_n_stratum = rp.n_stratum ? 1.0f / (float) (int) rp.n_stratum : 0.f;
    rp.inv_tcn = 1.0f / (float) s->tcn;
    rp.d_spp = make_fastdiv(rp.spp); rp.d_tcn = make_fastdiv(rp.tcn); rp.d_pcn = make_fastdiv(rp.pcn); rp.d_stratum = make_fastdiv(rp.n_stratum);
    rp.d_w = make_fastdiv(1); rp.n_passes = 1;
    return rp;
}
static SamplerState sampler_state(dtof_sampler *s) {
    SamplerState st; st.rng = s->rng.p; st.rng_time = s->rng_time.p; st.rng_path = s->rng_path.p; st.perm_seed = s->perm.p; st.dim = s->dim.p; st.n = s->wavefront;
    return st;
}
static void need_seeded(const dtof_sampler *s) { if (!s) throw std::runtime_error("null sampler"); if (!s->seeded) throw std::runtime_error("sampler is not seeded"); }

int dtof_sampler_create(uint32_t sample_count, uint32_t base_seed, int32_t tcn, int32_t pcn, dtof_sampler **out) {
    return guarded([&] {
        if (!out) throw std::runtime_error("null argument");
        if (tcn <= 0 || pcn <= 0) throw std::runtime_error("correlate numbers must be positive");
        auto s = new dtof_sampler(); s->sample_count = sample_count; s->base_seed = base_seed; s->tcn = tcn; s->pcn = pcn;
        *out = s;
    });
}
void dtof_sampler_destroy(dtof_sampler *s) { delete s; }
int dtof_sampler_set_samples_per_wavefront(dtof_sampler *s, uint32_t spw) {
    return guarded([&] {
        if (!s) throw std::runtime_error("null sampler");
        if (spw == 0 || s->sample_count % spw != 0) throw std::runtime_error("sample_count should be a multiple of samples_per_wavefront!");
        s->spw = spw;
    });
}
int dtof_sampler_seed(dtof_sampler *s, uint32_t seed, uint32_t wavefront_size) {
    return guarded([&] {
        if (!s) throw std::runtime_error("null sampler");
        if (wavefront_size == 0xffffffffu) { if (s->wavefront == 0) throw std::runtime_error("Sampler::seed(): wavefront_size should be specified!"); }
        else s->wavefront = wavefront_size;
        s->seed = seed; s->sample_index = 0;
        uint32_t n = s->wavefront;
        s->rng.ensure(n); s->rng_time.ensure(n); s->rng_path.ensure(n); s->perm.ensure(n); s->dim.ensure(n); s->out.ensure(2 * (size_t) n); s->flags.ensure(n);
        launch_sampler_seed(sampler_params(s), sampler_state(s), nullptr);
        HIP_CHECK(hipGetLastError()); HIP_CHECK(hipDeviceSynchronize());
        s->seeded = true;
    });
}
int dtof_sampler_advance(dtof_sampler *s) {
    return guarded([&] { need_seeded(s); s->sample_index++; HIP_CHECK(hipMemset(s->dim.p, 0, (size_t) s->wavefront * 4)); });
}
static void sampler_draw(dtof_sampler *s, const uint8_t *correlate, int all, int mode, float *out, int stride, int offset) {
    uint32_t n = s->wavefront;
    if (correlate) HIP_CHECK(hipMemcpy(s->flags.p, correlate, n, hipMemcpyHostToDevice));
    if (mode == 0) launch_sampler_next_1d(sampler_params(s), sampler_state(s), s->out.p, nullptr);
    else launch_sampler_next_correlate(sampler_params(s), sampler_state(s), correlate ? s->flags.p : nullptr, all, s->out.p, nullptr);
    HIP_CHECK(hipGetLastError());
    std::vector<float> tmp(n);
    HIP_CHECK(hipMemcpy(tmp.data(), s->out.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) out[(size_t) i * stride + offset] = tmp[i];
}
int dtof_sampler_next_1d(dtof_sampler *s, float *out) { return guarded([&] { need_seeded(s); sampler_draw(s, nullptr, 0, 0, out, 1, 0); }); }
int dtof_sampler_next_2d(dtof_sampler *s, float *out) {
    return guarded([&] { need_seeded(s); sampler_draw(s, nullptr, 0, 0, out, 2, 0); sampler_draw(s, nullptr, 0, 0, out, 2, 1); });
}
int dtof_sampler_next_1d_correlate(dtof_sampler *s, const uint8_t *c, int all, float *out) {
    return guarded([&] { need_seeded(s); sampler_draw(s, c, all, 1, out, 1, 0); });
}
int dtof_sampler_next_2d_correlate(dtof_sampler *s, const uint8_t *c, int all, float *out) {
    return guarded([&] { need_seeded(s); sampler_draw(s, c, all, 1, out, 2, 0); sampler_draw(s, c, all, 1, out, 2, 1); });
}
int dtof_sampler_next_1d_time(dtof_sampler *s, int strategy, float shift, int stratify, float *out) {
    return guarded([&] {
        need_seeded(s);
        if (strategy < 0 || strategy > TIME_REGULAR) throw std::runtime_error("unknown time sampling strategy");
        if (strategy == TIME_ANTITHETIC_MIRROR && s->tcn != 2)   // Assert(m_time_correlate_number == 2), correlated.cpp:142
            throw std::runtime_error("antithetic_mirror time sampling needs time_correlate_number == 2");
        if (strategy != TIME_UNIFORM && stratify && s->sample_count < (uint32_t) s->tcn)
            throw std::runtime_error("sample count must be at least time_correlate_number when per-interval stratification is on");
        RenderParams rp = sampler_params(s); rp.time_sampling = strategy; rp.antithetic_shift = shift; rp.stratify = stratify;
        launch_sampler_next_time(rp, sampler_state(s), s->sample_index * s->spw, s->out.p, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out, s->out.p, (size_t) s->wavefront * 4, hipMemcpyDeviceToHost));
    });
}
int dtof_sampler_get_state(dtof_sampler *s, uint32_t *out7) {
    return guarded([&] {
        need_seeded(s);
        uint32_t n = s->wavefront; std::vector<uint2> a(n), b(n), c(n); std::vector<uint32_t> p(n);
        HIP_CHECK(hipMemcpy(a.data(), s->rng.p, (size_t) n * 8, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(b.data(), s->rng_time.p, (size_t) n * 8, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(c.data(), s->rng_path.p, (size_t) n * 8, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(p.data(), s->perm.p, (size_t) n * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t *o = out7 + 7 * (size_t) i;
            o[0] = a[i].x; o[1] = a[i].y; o[2] = b[i].x; o[3] = b[i].y; o[4] = c[i].x; o[5] = c[i].y; o[6] = p[i];
        }
    });
}
// Sampler::fork (correlated.cpp:25-32: same configuration, fresh unseeded state) and Sampler::clone (:34-36: same configuration AND
// the current per-lane state: three PCG streams, permutation seed, dimension / sample index)
int dtof_sampler_fork(const dtof_sampler *s, dtof_sampler **out) {
    return guarded([&] {
        if (!s || !out) throw std::runtime_error("null argument");
        auto f = new dtof_sampler(); f->sample_count = s->sample_count; f->base_seed = s->base_seed; f->tcn = s->tcn; f->pcn = s->pcn;
        *out = f;
    });
}
int dtof_sampler_clone(const dtof_sampler *s, dtof_sampler **out) {
    return guarded([&] {
        if (!s || !out) throw std::runtime_error("null argument");
        std::unique_ptr<dtof_sampler> c(new dtof_sampler());
        c->sample_count = s->sample_count; c->base_seed = s->base_seed; c->tcn = s->tcn; c->pcn = s->pcn;
        c->seed = s->seed; c->wavefront = s->wavefront; c->spw = s->spw; c->sample_index = s->sample_index; c->seeded = s->seeded;
        if (s->seeded) {
            const size_t n = s->wavefront;
            c->rng.ensure(n); c->rng_time.ensure(n); c->rng_path.ensure(n); c->perm.ensure(n); c->dim.ensure(n); c->out.ensure(2 * n); c->flags.ensure(n);
            HIP_CHECK(hipMemcpy(c->rng.p, s->rng.p, n * 8, hipMemcpyDeviceToDevice));
            HIP_CHECK(hipMemcpy(c->rng_time.p, s->rng_time.p, n * 8, hipMemcpyDeviceToDevice));
            HIP_CHECK(hipMemcpy(c->rng_path.p, s->rng_path.p, n * 8, hipMemcpyDeviceToDevice));
            HIP_CHECK(hipMemcpy(c->perm.p, s->perm.p, n * 4, hipMemcpyDeviceToDevice));
            HIP_CHECK(hipMemcpy(c->dim.p, s->dim.p, n * 4, hipMemcpyDeviceToDevice));
        }
        *out = c.release();
    });
}
int dtof_sampler_set_sample_count(dtof_sampler *s, uint32_t spp) {   // Sampler::set_sample_count (sampler.h:129)
    return guarded([&] {
        if (!s) throw std::runtime_error("null sampler");
        if (spp == 0 || spp % s->spw != 0) throw std::runtime_error("sample_count should be a multiple of samples_per_wavefront!");
        s->sample_count = spp;
    });
}
int dtof_sampler_seeded(const dtof_sampler *s) { return s && s->seeded ? 1 : 0; }   // Sampler::seeded (sampler.h:141)
uint32_t dtof_sampler_wavefront_size(const dtof_sampler *s) { return s ? s->wavefront : 0; }
uint32_t dtof_sampler_sample_count(const dtof_sampler *s) { return s ? s->sample_count : 0; }

int dtof_eval_modulation(dtof_scene *sc, int mode, const float *t, const float *len, float *out, uint32_t n) {
    return guarded([&] {
        if (!sc || !t || !out || (mode == 0 && !len)) throw std::runtime_error("null argument");
        if (mode < 0 || mode > 2) throw std::runtime_error("unknown mode");
        RenderParams rp = make_params(sc, 0, sc->pp.sample_count ? sc->pp.sample_count : 1, nullptr, 0);
        DevBuf<float> dt, dl, dout; dt.ensure(n); dl.ensure(n); dout.ensure(n);
        HIP_CHECK(hipMemcpy(dt.p, t, (size_t) n * 4, hipMemcpyHostToDevice));
        if (len) HIP_CHECK(hipMemcpy(dl.p, len, (size_t) n * 4, hipMemcpyHostToDevice));
        launch_waveform_eval(rp, dt.p, dl.p, dout.p, mode, n, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out, dout.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    });
}

int dtof_eval_component(int component, const float *params, int n_params, const float *in, int in_stride, float *out, int out_stride, uint32_t n) {
    return guarded([&] {
        if ((!in || !out) && n) throw std::runtime_error("null argument");
        if (component < 0 || component >= COMP_COUNT) throw std::runtime_error("unknown component");
        if (n_params < 0 || n_params > 8 || (n_params && !params)) throw std::runtime_error("a component takes at most 8 parameters");
        static const int need_in[COMP_COUNT] = { 3, 6, 6, 5, 1, 1, 1, 2, 2, 2, 2, 3, 2, 1 }, need_out[COMP_COUNT] = { 1, 1, 1, 4, 4, 1, 1, 3, 2, 2, 3, 6, 1, 1 };
        static const int need_par[COMP_COUNT] = { 4, 4, 4, 4, 1, 2, 5, 0, 0, 0, 0, 0, 0, 1 };
        if (in_stride < need_in[component] || out_stride < need_out[component] || n_params < need_par[component])
            throw std::runtime_error("strides / parameter count too small for this component");
        int dev_count = 0;
        if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count == 0) { (void) hipGetLastError(); throw HipError("hipGetDeviceCount: no ROCm-capable device is detected"); }
        ComponentArgs a; memset(&a, 0, sizeof a);
        a.component = component; a.in_stride = in_stride; a.out_stride = out_stride; a.n = n;
        for (int i = 0; i < n_params; ++i) a.p[i] = params[i];
        RenderParams rp; memset(&rp, 0, sizeof rp);
        if (component == COMP_RFILTER) {
            const int kind = (int) a.p[0];
            if (kind < FILTER_BOX || kind > FILTER_LANCZOS || !(a.p[1] > 0.f)) throw std::runtime_error("unknown filter / non-positive radius");
            set_filter(rp, kind, a.p[1], a.p[2], a.p[3], a.p[4]);
        }
        DevBuf<float> din, dout; din.ensure((size_t) n * in_stride); dout.ensure((size_t) n * out_stride);
        HIP_CHECK(hipMemcpy(din.p, in, (size_t) n * in_stride * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemset(dout.p, 0, (size_t) n * out_stride * 4));
        a.in = din.p; a.out = dout.p;
        launch_component(a, rp, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out, dout.p, (size_t) n * out_stride * 4, hipMemcpyDeviceToHost));
    });
}

int dtof_bsdf_eval(dtof_scene *sc, uint32_t shape_index, uint32_t n, const float *in11, float *out14) {
    return guarded([&] {
        if (!sc || (n && (!in11 || !out14))) throw std::runtime_error("null argument");
        const BlobHeader *bh = (const BlobHeader *) sc->blob.data();
        if (shape_index >= bh->n_shapes) throw std::runtime_error("shape index out of range");
        ensure_device(sc);
        DevBuf<float> din, dout; din.ensure((size_t) n * 11); dout.ensure((size_t) n * 14);
        HIP_CHECK(hipMemcpy(din.p, in11, (size_t) n * 44, hipMemcpyHostToDevice));
        launch_bsdf_eval(sc->d_blob.p, shape_index, din.p, dout.p, n, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out14, dout.p, (size_t) n * 56, hipMemcpyDeviceToHost));
    });
}
int dtof_camera_rays(dtof_scene *sc, uint32_t n, const float *samples4, float *out7) {
    return guarded([&] {
        if (!sc || (n && (!samples4 || !out7))) throw std::runtime_error("null argument");
        if (!sc->host.has_sensor) throw std::runtime_error("the scene does not contain a sensor");
        ensure_device(sc);
        const RenderParams rp = make_params(sc, 0, sc->pp.sample_count ? sc->pp.sample_count : 1, nullptr, 0);
        DevBuf<float> din, dout; din.ensure((size_t) n * 4); dout.ensure((size_t) n * 7);
        HIP_CHECK(hipMemcpy(din.p, samples4, (size_t) n * 16, hipMemcpyHostToDevice));
        launch_camera_rays(rp, din.p, dout.p, n, nullptr);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out7, dout.p, (size_t) n * 28, hipMemcpyDeviceToHost));
    });
}

static int ray_query(dtof_scene *sc, uint32_t n, const float *rays8, float *out19, int32_t *ids, bool any, float *uv4 = nullptr) {
    return guarded([&] {
        if (!sc || (n && (!rays8 || !ids || (!any && !out19)))) throw std::runtime_error("null argument");
        ensure_device(sc);
        const BlobHeader *bh = (const BlobHeader *) sc->blob.data();
        DevBuf<float> dr, dout, duv; DevBuf<int32_t> dids;
        dr.ensure((size_t) n * 8); dout.ensure(any ? 1 : (size_t) n * 19); dids.ensure((size_t) n * (any ? 1 : 3));
        if (uv4) duv.ensure((size_t) n * 4);
        HIP_CHECK(hipMemcpy(dr.p, rays8, (size_t) n * 32, hipMemcpyHostToDevice));
        launch_ray_query(sc->d_blob.p, dr.p, dout.p, dids.p, uv4 ? duv.p : nullptr, n, any, bh->tlas_depth, nullptr);
        HIP_CHECK(hipGetLastError());
        if (!any) HIP_CHECK(hipMemcpy(out19, dout.p, (size_t) n * 19 * 4, hipMemcpyDeviceToHost));
        if (uv4 && n) HIP_CHECK(hipMemcpy(uv4, duv.p, (size_t) n * 16, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(ids, dids.p, (size_t) n * (any ? 1 : 3) * 4, hipMemcpyDeviceToHost));
    });
}
int dtof_ray_intersect(dtof_scene *sc, uint32_t n, const float *rays8, float *out19, int32_t *ids3) { return ray_query(sc, n, rays8, out19, ids3, false); }
int dtof_ray_intersect_uv(dtof_scene *sc, uint32_t n, const float *rays8, float *out19, int32_t *ids3, float *uv4) { return ray_query(sc, n, rays8, out19, ids3, false, uv4); }
int dtof_ray_test(dtof_scene *sc, uint32_t n, const float *rays8, int32_t *occluded) { return ray_query(sc, n, rays8, nullptr, occluded, true); }

#ifdef DTOF_TRAVERSAL_STATS
// development builds (make STATS=1): read and reset the traversal counters of dtof_traverse.h
int dtof_debug_traversal_stats(unsigned long long *out8) {
    return guarded([&] {
        HIP_CHECK(hipDeviceSynchronize());
        if (!read_traversal_stats(out8)) throw HipError("hipMemcpyFromSymbol(g_trav_stats) failed");
    });
}
#endif

}  // extern "C"
