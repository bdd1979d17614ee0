/*
 * dtof.h -- C ABI of libdtof: the MI355X-native `dopplertofpath` integrator and
 * `correlated` sampler (drop-in for that hot path of juhyeonkim95/Mitsuba3DopplerToF).
 *
 * Plain C, opaque handles, caller-owned buffers, integer status codes (0 = ok) with a
 * thread-local message in dtof_last_error().  No C++ or torch types cross this boundary.
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository root).  All compute runs in hand-written HIP kernels on the
 * current HIP device; there is no CPU fallback -- without a GPU every compute entry
 * point returns DTOF_ERR_HIP.
 */
#ifndef DTOF_H
#define DTOF_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTOF_OK            0
#define DTOF_ERR_INVALID   1   /* bad argument / unsupported scene feature / parse error (reference: Throw(...)) */
#define DTOF_ERR_HIP       2   /* HIP runtime failure (no device, out of memory, launch failure) */
#define DTOF_ERR_CANCELLED 3   /* Integrator::cancel() was called */

typedef struct dtof_scene   dtof_scene;     /* Scene + Sensor + Film + the two plugins' parameters */
typedef struct dtof_sampler dtof_sampler;   /* CorrelatedSampler state for n lanes (device resident) */

/* Version / capability string, e.g. "dtof 0.1 (gfx950)". */
const char *dtof_version(void);
/* Message of the last failing call on this thread (reference: the what() of the C++ exception). */
const char *dtof_last_error(void);

/* ---------------------------------------------------------------- scene loading
 * Replaces xml::load_file / xml::load_string (src/core/xml.cpp:1348-1429, called from
 * src/mitsuba/mitsuba.cpp:355-357 and mi.load_file in doppler_tutorials/src/program_runner.py:142).
 * `param_names/values` are the -Dname=value substitutions (src/mitsuba/mitsuba.cpp:241-248). */
int dtof_scene_load_file(const char *path, const char *const *param_names, const char *const *param_values,
                         int n_params, dtof_scene **out);
int dtof_scene_load_string(const char *xml, const char *const *param_names, const char *const *param_values,
                           int n_params, dtof_scene **out);
void dtof_scene_destroy(dtof_scene *scene);

/* Plugin construction: replaces PluginManager::create_object -> new DopplerToFPathIntegrator(props)
 * (src/integrators/dopplertofpath.cpp:19-57 + bases src/render/integrator.cpp:22-28,54-100,568-585) and
 * new CorrelatedSampler(props) (src/samplers/correlated.cpp:17-23, src/render/sampler.cpp:11-20), and
 * mi.load_dict({'type':'dopplertofpath', ...}) of doppler_tutorials/src/program_runner.py:127-141.
 * Properties are given as parallel arrays; `types[i]` is one of 'f' (float), 'i' (integer), 'b' (boolean,
 * value "true"/"false"), 's' (string).  The plugin name goes in `plugin`: integrators "dopplertofpath", "path"
 * (src/integrators/path.cpp), "velocity" (src/integrators/velocity.cpp:125-142); samplers "correlated", "independent"
 * (src/samplers/independent.cpp) and "timestratified" (src/samplers/timestratified.cpp:67-129).
 * Unknown plugin names, wrong types and unreferenced properties fail like the reference's loader. */
int dtof_scene_set_integrator(dtof_scene *scene, const char *plugin, const char *const *names,
                              const char *types, const char *const *values, int n);
int dtof_scene_set_sampler(dtof_scene *scene, const char *plugin, const char *const *names,
                           const char *types, const char *const *values, int n);

/* Scene-independent plugin objects, as the reference constructs them: PluginManager::create_object -> `new T(props)`
 * (src/core/plugin.cpp:174-208; mi.load_dict({...}), program_runner.py:142).  The constructor validates the properties exactly
 * like dtof_scene_set_integrator / _sampler; dtof_integrator_render is Integrator::render(scene, sensor, seed, spp)
 * (include/mitsuba/render/integrator.h:74-79): it installs the integrator (and, if given, the sampler -- NULL keeps the one of
 * the scene file) on `scene` and renders.  Destroy with the matching *_destroy. */
typedef struct dtof_integrator dtof_integrator;
typedef struct dtof_sampler_plugin dtof_sampler_plugin;
int  dtof_integrator_create(const char *plugin, const char *const *names, const char *types, const char *const *values, int n,
                            dtof_integrator **out);
void dtof_integrator_destroy(dtof_integrator *integrator);
int  dtof_sampler_plugin_create(const char *plugin, const char *const *names, const char *types, const char *const *values, int n,
                                dtof_sampler_plugin **out);
void dtof_sampler_plugin_destroy(dtof_sampler_plugin *sampler);

typedef struct {
    int32_t  film_width, film_height, crop_x, crop_y, crop_width, crop_height;
    uint32_t sample_count;          /* Sampler::sample_count() */
    uint32_t n_shapes, n_groups, n_objects, n_emitters, n_triangles, n_bvh_nodes;
    uint32_t scene_blob_bytes;
    /* constructor-rounded plugin parameters (for parity checks of the constructors) */
    float    time, w_g, g_1, g_0, w_s, phase_offset, hetero_frequency, antithetic_shift;
    int32_t  wave_type, low_frequency_component_only, time_sampling, stratify_each_interval;
    uint32_t path_correlation_depth, max_depth, rr_depth, base_seed;
    int32_t  time_correlate_number, path_correlate_number;
    uint32_t bvh_stack_depth;       /* entries a traversal stack can need: TLAS depth + deepest per-mesh BLAS */
    /* reconstruction filter (ReconstructionFilter::radius(), include/mitsuba/render/rfilter.h) and the rows a splat can reach beyond
     * the pixel of its sample: ceil(radius - 0.5) (ImageBlock::put, src/render/imageblock.cpp:423-426; 0 for the box filter, which
     * splats at the lane's own pixel, integrator.cpp:540-541).  A row-band shard must carry `filter_halo` padding rows on each side. */
    float    filter_radius;
    int32_t  filter_halo;
    /* hdrfilm pixel_format = rgba (FilmFlags::Alpha, src/films/hdrfilm.cpp:172-177): develop() returns R, G, B, A.  The host-buffer render calls then write
     * FOUR floats per pixel, and the device-film calls (dtof_render_rows / _stripes) accumulate the alpha film -- (A, 0, 0, W) -- as one more RGBW plane
     * behind the n_offsets colour films of `d_film_rgbw`: they REFUSE an rgba scene until the caller has declared a film of n_offsets + 1 planes
     * (dtof_scene_set_film_layout). */
    int32_t  has_alpha;
} dtof_scene_info;
/* What Film::crop_size / Sampler::sample_count / the plugins' to_string() report (src/films/hdrfilm.cpp:235-279, src/render/sampler.cpp:13-14,
 * src/integrators/dopplertofpath.cpp:315-328), plus the sizes of the packed scene. */
int dtof_scene_get_info(const dtof_scene *scene, dtof_scene_info *info);

/* Flat float32 export of what the loader produced (parity of the XML semantics, row X1):
 * kind 0: object keyframes  -> per object 2+32 floats (t0,t1, key0[16], key1[16]) , rows of `out`
 * kind 1: shape transforms  -> per shape 32 floats (to_world[16], to_object[16])
 * kind 2: sensor            -> to_world[16], x_fov, near, far, shutter_open, shutter_close, sensor kind (0 perspective, 1 thinlens, 2 orthographic), aperture_radius, focus_distance
 * kind 3: emitters          -> per emitter position[3], intensity[3]
 * kind 4..7: baked mesh data -> positions / vertex normals / texcoords / faces (uint32 bit patterns) of all mesh
 *                             shapes (cube, obj, ply) concatenated in shape order (cube.cpp:114-160, obj.cpp, ply.cpp)
 * kind 8: spheres           -> per sphere m_center[3], m_radius, m_inv_surface_area, flip_normals (sphere.cpp:138-160);
 *                             their composed to_world / to_object are in kind 1
 * kind 9: BSDF records      -> per shape 24 floats: kind (0 diffuse, 1 conductor, 2 dielectric, 3 plastic, 4 roughconductor), twosided, eta,
 *                             nonlinear, 1/eta^2, fdr_int, specular sampling weight, reflectance[3], specular_reflectance[3],
 *                             specular_transmittance[3], conductor eta[3], k[3], alpha_u, alpha_v
 *                             (src/bsdfs/{diffuse,conductor,dielectric,plastic,roughconductor,roughplastic}.cpp; 5 = roughplastic, 6 = thindielectric, 7 = roughdielectric,
 *                             whose fdr_int slot carries m_internal_reflectance)
 * kind 11: spot emitters   -> per spot 22 floats: position[3], intensity[3], world-to-local[12], cutoff angle (rad), cos(cutoff), cos(beam width),
 *                             1 / (cutoff - beam width) (src/emitters/spot.cpp:75-100)
 * kind 12: microfacet distribution of the rough BSDFs -> per shape 1 float: 0 beckmann, 1 ggx (MicrofacetType, include/mitsuba/render/microfacet.h:30-36)
 * kind 13: textures        -> per texture (in order of appearance) 17 floats: kind (0 checkerboard, 1 bitmap), filter (0 nearest, 1 bilinear), wrap (0 repeat,
 *                             1 mirror, 2 clamp), channels, width, height, to_uv 2x2, color0[3], color1[3], mean (src/textures/{checkerboard,bitmap}.cpp)
 * kind 18: per emitter 10 floats: kind, position[3], intensity | radiance | irradiance[3], direction of travel[3] (directional emitters)
 * kind 17: per shape 1 float: 1 if its rough BSDF samples all normals (sample_visible = false), else 0
 * kind 16: the environment map (src/emitters/envmap.cpp) as packed: w, h, levels, scale, bounding sphere[4], to_world[12], to_local[12], m_data[h*w*3], then
 *          per level of the Hierarchical2D warp (distr_2d.h:376-482): width, count, values[count]
 * kind 14: texels          -> the linear float32 texels of all bitmap textures, concatenated; kind 15: per shape the index of the texture on its
 *                             reflectance / diffuse_reflectance, -1 = a colour
 * kind 19: per shape 4 floats: indices of the textures on specular_reflectance, specular_transmittance, alpha_u, alpha_v (-1 = a constant)
 * kind 20: per shape 3 floats: inside a `mask` (0 / 1), its opacity (the constant, or the texture's mean), index of the opacity texture (src/bsdfs/mask.cpp)
 * kind 21: per shape 1 float: index of the texture of its `normalmap` / `bumpmap`, -1 = none; kind 22: per shape 2 floats: is a `bumpmap`, its scale
 * kind 23: per shape 5 floats: is a `blendbsdf`, its weight, index of the weight texture, BSDF kind and two-sidedness of bsdf_1 (bsdf_0 is what kind 9 reports)
 * kind 24: per shape 1 float: index of the texture on its area emitter's radiance, -1 = a constant colour
 * kind 10: roughplastic tables -> per roughplastic shape the 64 values of m_external_transmittance (roughplastic.cpp:222-257)
 * Returns the number of floats written (<= capacity) through *n_written. */
int dtof_scene_export(const dtof_scene *scene, int kind, float *out, size_t capacity, size_t *n_written);
/* Sensor::sample_ray of the scene's sensor over arrays (src/sensors/perspective.cpp:238-279, thinlens.cpp:257-305, orthographic.cpp:169-196), through the device
 * function the first-bounce kernel generates its primary rays with: per sample the position sample x, y in [0, 1]^2 of the crop window and the aperture sample
 * x, y (4 floats) -> ray origin[3], direction[3], maxt (7 floats).  The ray's time is the sampler's business (render_sample, integrator.cpp:494-496). */
int dtof_camera_rays(dtof_scene *scene, uint32_t n, const float *samples4, float *out7);
/* BSDF::eval_pdf_sample of a shape's BSDF over arrays (src/render/bsdf.cpp:20-29; src/bsdfs/{diffuse,twosided,plastic,conductor,dielectric,thindielectric,rough*,
 * mask,blendbsdf,normalmap,bumpmap,null}.cpp), through the very device function the shade kernels call at every path vertex.  The interaction has the flat local
 * frame (n = sh_n = +z, dp_du = s = +x): per query wi[3], wo[3] (local), sample1, sample2[2], uv[2] (11 floats) -> value * cos(theta_o)[3], pdf(wo), the sampled
 * direction[3], its pdf, eta, 1 if the sampled lobe is a delta lobe, weight[3], 1 if it is a null lobe (14 floats).  `shape_index` counts the scene's shapes in
 * file order (the shapes of a shapegroup included). */
int dtof_bsdf_eval(dtof_scene *scene, uint32_t shape_index, uint32_t n, const float *in11, float *out14);

/* ---------------------------------------------------------------- rendering
 * Replaces Integrator::render(Scene*, uint32_t sensor_index, uint32_t seed, uint32_t spp, bool develop,
 * bool evaluate) (include/mitsuba/render/integrator.h:74-79; src/render/integrator.cpp:104-347), i.e.
 * integrator.render(scene, seed=i, spp=n) of program_runner.py:15,23.  spp == 0 uses the sampler's
 * sample_count (integrator.cpp:121-124).  Passes: the integrator's `samples_per_pass` property, or a wavefront of more than
 * 2^32 - 1 lanes, splits the render into spp / spp_per_pass passes exactly as integrator.cpp:121-135,227-245 does -- the sampler is
 * seeded once, its streams run on from pass to pass (Sampler::advance, sampler.cpp:52-55), the film accumulates all passes. */
typedef struct {
    uint64_t n_paths;            /* W*H*spp lanes evaluated by this call */
    uint64_t n_bounces;          /* closest-hit rays traced (path-bounces through the trace+shade loop) */
    uint64_t n_shadow_rays;      /* occlusion rays traced */
    double   ms_total;           /* generate .. develop, HIP events on the library's stream */
    double   ms_generate, ms_trace, ms_shade, ms_shadow, ms_splat;   /* per-stage sums (HIP events) */
    uint32_t n_launches_trace, n_launches_shade, n_launches_shadow;
    uint32_t n_batches;
    /* fused pipeline: the first-bounce launches (lane generation + primary ray + bounce 0 in one kernel) are also counted
     * in ms_shade / n_launches_shade; these two fields single them out */
    uint32_t n_launches_first;
    double   ms_first;
    /* the first-bounce kernel runs up to four iterations of the bounce loop itself, the path state in registers: iterations covered by
     * the first-bounce launches (summed over the batches) and the path-bounces among n_bounces that ran there */
    uint32_t n_inline_iterations;
    uint64_t n_bounces_inline;
    /* first-bounce launches that also splatted their samples (the wave reduces the footprint values of its 64 samples and issues the film atomics itself: 64 spp, one
     * film, radius-1 tent, the whole path inline).  The sums of a pixel's samples are then taken in another ORDER than the splat kernels': films of the two paths
     * differ in the last bits, so checksums compare only between runs with the same value here */
    uint32_t n_fused_splat_launches;
} dtof_render_stats;

/* out_rgb: caller-owned host buffer, crop_height*crop_width*3 float32, developed (RGB / W). */
int dtof_render(dtof_scene *scene, uint32_t sensor_index, uint32_t seed, uint32_t spp,
                float *out_rgb, dtof_render_stats *stats);

/* Integrator::render on a scene-independent plugin object (include/mitsuba/render/integrator.h:74-79): the integrator created by
 * dtof_integrator_create renders `scene` with its own parameters and, if given, the sampler plugin object's (else the scene's). */
int dtof_integrator_render(const dtof_integrator *integrator, const dtof_sampler_plugin *sampler_or_null, dtof_scene *scene,
                           uint32_t sensor_index, uint32_t seed, uint32_t spp, float *out_rgb, dtof_render_stats *stats);

/* Tile / shard entry point (no reference counterpart: the reference is single-device).
 * Renders pixel rows [row_begin,row_end) of the crop window and ACCUMULATES the undeveloped R,G,B,W
 * film (hdrfilm.cpp:235-279 channel layout) into `d_film_rgbw`, a DEVICE buffer of
 * crop_height*crop_width*4 float32 the caller zeroed (rows row_begin-r..row_end+r receive splats,
 * r = filter footprint).  n_offsets > 1 evaluates several `hetero_offset` values (in units of 2*pi
 * like the plugin property) in ONE traversal; film k lives at d_film_rgbw + k*crop_h*crop_w*4.
 * offsets == NULL / n_offsets == 0 uses the integrator's own phase offset. */
int dtof_render_rows(dtof_scene *scene, uint32_t seed, uint32_t spp, int32_t row_begin, int32_t row_end,
                     const float *offsets, int n_offsets, float *d_film_rgbw, dtof_render_stats *stats);
/* Frames WITHOUT a host synchronisation per frame (a caller that renders frame after frame -- bench.py, an animation -- otherwise leaves the GPU idle while the host
 * reads counters and sets up the next call): dtof_render_rows_async enqueues what dtof_render_rows enqueues on the scene's stream and returns; HIP events around the
 * frame and its stages are kept; the bounce / shadow-ray counters are not read back.  dtof_clear_async / dtof_develop_async put a memset / the film development
 * (HDRFilm::develop, hdrfilm.cpp:305-406) on the same stream.  dtof_async_collect waits for the stream, sums the event times and launch counters of the frames
 * enqueued since the last collect into `sum` (n_bounces / n_shadow_rays stay 0), writes the duration of each frame to frame_ms[0 .. capacity) and their number to
 * *n_frames.  Frames whose pipeline needs counters on the host between bounces (paths that leave the fused first-bounce kernel) still wait there. */
int dtof_render_rows_async(dtof_scene *scene, uint32_t seed, uint32_t spp, int32_t row_begin, int32_t row_end, const float *offsets, int n_offsets, float *d_film);
int dtof_clear_async(dtof_scene *scene, void *d_ptr, size_t bytes);
int dtof_develop_async(dtof_scene *scene, const float *d_film, float *d_rgb, int64_t n_pixels);
int dtof_async_collect(dtof_scene *scene, dtof_render_stats *sum, double *frame_ms, uint32_t capacity, uint32_t *n_frames);

/* Interleaved shards (load balance when the cost of a row depends on what it sees): renders the stripes of rows
 * [first_row + k * stripe_period, first_row + k * stripe_period + stripe_rows), k = 0, 1, ..., below crop_height and accumulates
 * like dtof_render_rows.  Rank r of N uses first_row = r * stripe_rows, stripe_period = N * stripe_rows; the union over the
 * ranks is the full frame, lane for lane what a single device renders. */
int dtof_render_stripes(dtof_scene *scene, uint32_t seed, uint32_t spp, int32_t first_row, int32_t stripe_rows, int32_t stripe_period,
                        const float *offsets, int n_offsets, float *d_film_rgbw, dtof_render_stats *stats);
/* The stream the library enqueues on.  By default every scene owns one; a caller that has work of its own to order against the renders -- the film exchange of
 * a multi-GPU frame (torch.distributed / RCCL), torch operations on the films -- hands in ITS stream (hipStream_t; torch.cuda.current_stream().cuda_stream) and can
 * then queue clear -> render -> exchange -> develop for many frames without a host wait in between.  NULL goes back to the scene's own stream.  LIFETIME: the library
 * keeps the handle, not the stream -- the caller keeps the stream alive until it has handed it back (another dtof_scene_set_stream call; frames in flight are collected
 * first); handing back a stream that no longer exists is tolerated, every other call on a dead handle fails with DTOF_ERR_HIP.  The reference has
 * no counterpart (Dr.Jit owns the one CUDA stream of a thread, drjit-core/src/init.cpp). */
int dtof_scene_set_stream(dtof_scene *scene, void *hip_stream);
/* dtof_render_stripes without the host synchronisation at its end (see dtof_render_rows_async). */
int dtof_render_stripes_async(dtof_scene *scene, uint32_t seed, uint32_t spp, int32_t first_row, int32_t stripe_rows, int32_t stripe_period,
                              const float *offsets, int n_offsets, float *d_film_rgbw);
/* The layout of the caller's device film for the device-film calls (dtof_render_rows / _stripes and their _async forms): `planes` RGBW planes of crop_height * crop_width * 4
 * floats, `plane_stride_floats` apart (0 = dense).  Film k of the batched offsets is plane k; the alpha film of an rgba scene (dtof_scene_info::has_alpha,
 * FilmFlags::Alpha, src/films/hdrfilm.cpp:172-177; ImageBlock::put of aovs[3], integrator.cpp:528-533) is plane n_offsets.  A call that would write more planes than were
 * declared fails with DTOF_ERR_INVALID instead of writing past the buffer; planes = 0 (the default) declares nothing: rgb scenes write their n_offsets planes, rgba scenes
 * are refused.  A band shard that renders into a padded slab (pointer = slab + halo rows) passes the slab's size as the stride. */
int dtof_scene_set_film_layout(dtof_scene *scene, int32_t planes, uint64_t plane_stride_floats);
/* HDRFilm::develop (src/films/hdrfilm.cpp:305-406) on device buffers: rgb = RGB / (W == 0 ? 1 : W). */
int dtof_develop(const float *d_film_rgbw, float *d_rgb, int64_t n_pixels);
/* ... enqueued on the caller's stream without a host wait (hipStream_t; NULL = the null stream): the develop of a frame whose film exchange runs on a stream of its
 * own, beside the next frame's render (bench.py: the exchange of frame i overlaps the render of frame i + 1). */
int dtof_develop_on_stream(const float *d_film_rgbw, float *d_rgb, int64_t n_pixels, void *hip_stream);
/* ... of an rgba film: rgba = (R, G, B) / W of the colour film and A / W of the alpha film (the plane behind the colour films, see dtof_scene_info::has_alpha). */
int dtof_develop_rgba(const float *d_film_rgbw, const float *d_alpha_film, float *d_rgba, int64_t n_pixels);
/* Same as dtof_render but with n_offsets batched modulation offsets; out_rgb holds n_offsets images. */
int dtof_render_offsets(dtof_scene *scene, uint32_t seed, uint32_t spp, const float *offsets, int n_offsets,
                        float *out_rgb, dtof_render_stats *stats);

/* Integrator::cancel / should_stop (include/mitsuba/render/integrator.h:96-109). */
void dtof_cancel(dtof_scene *scene);

/* Per-lane debugging entry (SURVEY 8b "dtof_sample_lanes"): evaluates wavefront lanes
 * [lane_begin, lane_begin+n) exactly as dtof_render would (multi-pass renders: index = pass * wavefront_size + lane, a range must stay
 * inside one pass) and returns, per lane,
 * sample_pos[2], time, ray_o[3], ray_d[3], rgb[3] (12 floats) -- the (Spectrum, position) pair that
 * render_sample hands to ImageBlock::put (src/render/integrator.cpp:509-541). */
int dtof_sample_lanes(dtof_scene *scene, uint32_t seed, uint32_t spp, uint64_t lane_begin, uint64_t n, float *out_lanes12);
/* The same, plus the `valid` half of the (Spectrum, Mask) pair DopplerToFPathIntegrator::sample returns (src/integrators/dopplertofpath.cpp:279-282:
 * valid_ray -- the path met a vertex whose sampled lobe was not BSDFFlags::Null, or the environment is visible): 1 / 0 per lane. */
int dtof_sample_lanes_valid(dtof_scene *scene, uint32_t seed, uint32_t spp, uint64_t lane_begin, uint64_t n, float *out_lanes12, uint32_t *out_valid);

/* ---------------------------------------------------------------- sampler surface
 * Array-of-lanes form of the Sampler interface (include/mitsuba/render/sampler.h:99-168) for the
 * correlated sampler; state lives on the GPU, results are copied to caller-owned host arrays of n floats. */
int  dtof_sampler_create(uint32_t sample_count, uint32_t base_seed, int32_t time_correlate_number,
                         int32_t path_correlate_number, dtof_sampler **out);          /* correlated.cpp:17-23 */
void dtof_sampler_destroy(dtof_sampler *s);
int  dtof_sampler_seed(dtof_sampler *s, uint32_t seed, uint32_t wavefront_size);       /* correlated.cpp:38-64 */
int  dtof_sampler_set_samples_per_wavefront(dtof_sampler *s, uint32_t spw);            /* sampler.cpp:75-83 */
int  dtof_sampler_advance(dtof_sampler *s);                                            /* sampler.cpp:52-55 */
int  dtof_sampler_next_1d(dtof_sampler *s, float *out);                                /* correlated.cpp:79-84 */
int  dtof_sampler_next_2d(dtof_sampler *s, float *out_xy);                             /* correlated.cpp:86-90, n*2 */
/* correlate: per-lane flags (n bytes) or NULL to use `correlate_all` for every lane. */
int  dtof_sampler_next_1d_correlate(dtof_sampler *s, const uint8_t *correlate, int correlate_all, float *out);  /* :156-161 */
int  dtof_sampler_next_2d_correlate(dtof_sampler *s, const uint8_t *correlate, int correlate_all, float *out_xy); /* :163-167 */
/* strategy: ETimeSampling (sampler.h:27-34): 0 uniform, 1 stratified, 2 antithetic, 3 antithetic_mirror (time_correlate_number must be 2, correlated.cpp:142),
 * 4 periodic (correlated.cpp:147-150), 5 regular (falls through to `return r`, :152) */
int  dtof_sampler_next_1d_time(dtof_sampler *s, int strategy, float antithetic_shift, int stratify_each_interval, float *out); /* :92-153 */
/* state readback: 7 uint32 per lane = rng.state lo,hi, rng_time.state lo,hi, rng_path.state lo,hi, permutation seed */
int  dtof_sampler_get_state(dtof_sampler *s, uint32_t *out7);
/* Sampler::fork (src/samplers/correlated.cpp:25-32): same configuration, unseeded.  Sampler::clone (:34-36): same configuration and
 * the same per-lane state, so both produce the same numbers from here on.  set_sample_count / seeded: include/mitsuba/render/sampler.h:129,141.
 * (schedule_state / loop_put of the reference are Dr.Jit loop plumbing and have no counterpart here.) */
int  dtof_sampler_fork(const dtof_sampler *s, dtof_sampler **out);
int  dtof_sampler_clone(const dtof_sampler *s, dtof_sampler **out);
int  dtof_sampler_set_sample_count(dtof_sampler *s, uint32_t sample_count);
int  dtof_sampler_seeded(const dtof_sampler *s);
uint32_t dtof_sampler_wavefront_size(const dtof_sampler *s);
uint32_t dtof_sampler_sample_count(const dtof_sampler *s);

/* ---------------------------------------------------------------- modulation functions
 * eval_modulation_weight (dopplertofpath.cpp:60-77) and the waveform library
 * (include/mitsuba/render/waveform_utils.h:24-62) over arrays, evaluated by the same device functions the
 * shade kernel uses.  mode 0: weight(ray_time=t[i], path_length=len[i]) with the scene's integrator;
 * mode 1: eval_modulation_function_value(t[i]); mode 2: ..._low_pass(t[i]). */
int dtof_eval_modulation(dtof_scene *scene, int mode, const float *t, const float *len, float *out, uint32_t n);

/* ---------------------------------------------------------------- ray queries
 * Scene::ray_intersect / Scene::ray_test (include/mitsuba/render/scene.h; src/render/scene.cpp -> scene_embree.inl:202-333,349-426) over
 * arrays, through the same TLAS / BLAS traversal and surface-interaction code the render kernels use.  rays8: per ray o[3], d[3], time,
 * maxt.  out19: t (inf on a miss), p[3], n[3], sh_frame.n[3], sh_frame.s[3], sh_frame.t[3], wi[3]; ids3: object, shape-in-group,
 * primitive (-1 on a miss).  dtof_ray_test writes 1 / 0 per ray. */
int dtof_ray_intersect(dtof_scene *scene, uint32_t n, const float *rays8, float *out19, int32_t *ids3);
/* the same, plus uv4 per ray: si.uv[2] (the surface parameterisation, interaction.h) and pi.prim_uv[2] (PreliminaryIntersection::prim_uv: the barycentric
 * coordinates on a triangle, the local position on a rectangle or disk), as the reference's tests of meshes assert them (src/render/tests/test_mesh.py:258-292) */
int dtof_ray_intersect_uv(dtof_scene *scene, uint32_t n, const float *rays8, float *out19, int32_t *ids3, float *uv4);
int dtof_ray_test(dtof_scene *scene, uint32_t n, const float *rays8, int32_t *occluded);

/* ---------------------------------------------------------------- component evaluation
 * The device functions the shade and splat kernels are built from, evaluated over arrays on the GPU: the counterpart of the
 * free functions / small classes the reference exposes to its unit tests (mi.fresnel, mi.MicrofacetDistribution, mi.warp.*,
 * ReconstructionFilter::eval, ...), so that the reference's own known answers (tests/golden/reference_kats.json.gz) can be held
 * against the GPU code.  Element i reads in[i * in_stride ...] and writes out[i * out_stride ...]; `params` are per-call scalars. */
#define DTOF_COMP_MICROFACET_EVAL          0   /* MicrofacetDistribution::eval (microfacet.h:176-196). params: type (0 beckmann, 1 ggx), alpha_u, alpha_v, sample_visible; in m[3]; out 1 */
#define DTOF_COMP_MICROFACET_PDF           1   /* ::pdf (:219-228); in wi[3], m[3]; out 1 */
#define DTOF_COMP_MICROFACET_G1            2   /* ::smith_g1 (:341-365); in v[3], m[3]; out 1 */
#define DTOF_COMP_MICROFACET_SAMPLE        3   /* ::sample (:240-325); in wi[3], sample[2]; out m[3], pdf */
#define DTOF_COMP_FRESNEL                  4   /* fresnel (fresnel.h:21-63). params: eta; in cos_theta_i; out r, cos_theta_t, eta_it, eta_ti */
#define DTOF_COMP_FRESNEL_CONDUCTOR        5   /* fresnel_conductor (fresnel.h:93-117). params: eta, k; in cos_theta_i; out 1 */
#define DTOF_COMP_RFILTER                  6   /* ReconstructionFilter::eval. params: kind (0 box, 1 tent, 2 gaussian, 3 mitchell, 4 catmullrom, 5 lanczos: radius = lobes), radius, stddev, B, C; in x; out 1 */
#define DTOF_COMP_WARP_COSINE_HEMISPHERE   7   /* warp::square_to_cosine_hemisphere (warp.h:320-344); in sample[2]; out 3 */
#define DTOF_COMP_WARP_DISK_CONCENTRIC     8   /* warp::square_to_uniform_disk_concentric (warp.h:54-90); out 2 */
#define DTOF_COMP_WARP_UNIFORM_TRIANGLE    9   /* warp::square_to_uniform_triangle (warp.h:153-156); out 2 */
#define DTOF_COMP_WARP_UNIFORM_SPHERE     10   /* warp::square_to_uniform_sphere (warp.h:250-255); out 3 */
#define DTOF_COMP_COORDINATE_SYSTEM       11   /* coordinate_system (vector.h:116-136) = Frame3f(n); in n[3]; out s[3], t[3] */
#define DTOF_COMP_TEA_FLOAT32             12   /* sample_tea_float32 (random.h:33-67), 4 rounds; in v0, v1 (uint32 bit patterns); out 1 */
#define DTOF_COMP_MATH                    13   /* restated Dr.Jit math. params: 0 exp, 1 log, 2 tan, 3 erf, 4 erfinv, 5 sin, 6 cos, 7 acos; in x; out 1 */
int dtof_eval_component(int component, const float *params, int n_params, const float *in, int in_stride, float *out, int out_stride, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
