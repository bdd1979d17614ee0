/*
 * dtof_oracle.h -- flat scene/parameter records consumed by the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * The oracle is a scalar restatement of the reference's `dopplertofpath`
 * integrator + `correlated` sampler (juhyeonkim95/Mitsuba3DopplerToF @ 2024_08_07).
 * Every function in dtof_oracle.c cites the reference file:line it follows.
 *
 * PARITY STATUS: the reference ships no test, golden vector or fixture for this
 * path and cannot be built or imported here (Dr.Jit/Embree/pugixml submodules are
 * empty).  The oracle is therefore pinned only by the known-answer vectors that
 * do exist for its building blocks (TEA: src/core/tests/test_random.py:8-16,
 * PCG32: O'Neill's published demo vector, Kensler bijection property, closed-form
 * waveform values) -- for the path as a whole: "parity unpinned".
 */
#ifndef DTOF_ORACLE_H
#define DTOF_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_SHAPE_RECT = 0, ORC_SHAPE_MESH = 1, ORC_SHAPE_SPHERE = 2, ORC_SHAPE_DISK = 3, ORC_SHAPE_CYLINDER = 4 };
enum { ORC_OBJ_SHAPE = 0, ORC_OBJ_INSTANCE = 1 };
enum { ORC_EMITTER_POINT = 0, ORC_EMITTER_AREA = 1, ORC_EMITTER_SPOT = 2, ORC_EMITTER_CONSTANT = 3, ORC_EMITTER_ENVMAP = 4, ORC_EMITTER_DIRECTIONAL = 5 };
/* directional (src/emitters/directional.cpp): position = the direction of travel d (unit), intensity = irradiance, bsphere as for the environment */
enum { ORC_WAVE_SIN = 0, ORC_WAVE_RECT = 1, ORC_WAVE_TRI = 2, ORC_WAVE_TRAP = 3 };
enum { ORC_TIME_UNIFORM = 0, ORC_TIME_STRATIFIED = 1, ORC_TIME_ANTITHETIC = 2,
       ORC_TIME_ANTITHETIC_MIRROR = 3, ORC_TIME_PERIODIC = 4, ORC_TIME_REGULAR = 5 };   /* ETimeSampling, include/mitsuba/render/sampler.h:27-34 */
enum { ORC_BSDF_DIFFUSE = 0, ORC_BSDF_CONDUCTOR = 1, ORC_BSDF_DIELECTRIC = 2, ORC_BSDF_PLASTIC = 3, ORC_BSDF_ROUGHCONDUCTOR = 4, ORC_BSDF_ROUGHPLASTIC = 5, ORC_BSDF_THINDIELECTRIC = 6, ORC_BSDF_ROUGHDIELECTRIC = 7, ORC_BSDF_NULL = 8 };
enum { ORC_FILTER_BOX = 0, ORC_FILTER_TENT = 1, ORC_FILTER_GAUSSIAN = 2, ORC_FILTER_MITCHELL = 3, ORC_FILTER_CATMULLROM = 4, ORC_FILTER_LANCZOS = 5 };

/* All 4x4 matrices are row-major float32: m[4*r + c]. */

/* Texture on a BSDF's (diffuse) reflectance: src/textures/checkerboard.cpp, src/textures/bitmap.cpp (RGB variants).
 * to_uv = the 2x2 linear part of the `to_uv` transform: Transform4f::extract() (transform.h:340-360) copies the upper-left 2x2 block and
 * the bottom ROW, so a translation never reaches the 3x3 transform the plugins apply. */
enum { ORC_TEX_CHECKERBOARD = 0, ORC_TEX_BITMAP = 1 };
typedef struct {
    int32_t kind, filter /* 0 nearest, 1 bilinear */, wrap /* 0 repeat, 1 mirror, 2 clamp */, channels /* 1 or 3 */;
    int32_t width, height;
    float   to_uv[4];        /* m00, m01, m10, m11 */
    float   color0[3], color1[3];   /* checkerboard (constant colours) */
    const float *data;       /* bitmap: height * width * channels linear float32 texels, row 0 first */
    /* DiscreteDistribution2D over the texels (include/mitsuba/core/distr_2d.h:75-181; BitmapTexture::rebuild_internals, bitmap.cpp:674-734: the luminance of RGB
     * texels, the value of gray ones), built only for textures that are importance-sampled (the radiance of an area emitter): running sums kept in float32,
     * accumulated in double; NULL otherwise */
    const float *cond_cdf, *marg_cdf; float normalization, inv_normalization;
} orc_texture;

typedef struct orc_shape_s {
    int32_t kind;            /* ORC_SHAPE_* */
    int32_t twosided;        /* BSDF is twosided{diffuse} (1) or plain diffuse (0) */
    int32_t flip_normals;
    int32_t face_normals;    /* mesh: ignore vertex normals */
    float   reflectance[3];
    float   to_world[16];    /* rectangle: shape's own to_world (identity for instanced children) */
    float   to_object[16];   /* rectangle: float cast of the double-precision inverse */
    /* mesh data (already in the shape's world space, cube.cpp:150-160) */
    int32_t n_vertices, n_faces;
    const float    *positions;  /* n_vertices*3 */
    const float    *normals;    /* n_vertices*3 or NULL */
    const float    *texcoords;  /* n_vertices*2 or NULL */
    const uint32_t *faces;      /* n_faces*3 */
    /* area emitter attached to this (static, top-level) shape: src/emitters/area.cpp */
    int32_t emitter;            /* 0 / 1 */
    float   radiance[3];
    /* mesh emitters: Mesh::build_pmf (mesh.cpp:478-511) -> DiscreteDistribution over the faces (distr_1d.h:20-240),
     * filled by orc_mesh_area_table */
    const float *area_pmf, *area_cdf;   /* n_faces each */
    float   area_sum, area_norm;        /* float(sum), float(1 / sum) */
    int32_t area_lo, area_hi;           /* m_valid: first / last face with non-zero area */
    /* sphere (src/shapes/sphere.cpp:117-160), filled by orc_bake_sphere; to_world / to_object above hold the composed
     * to_world * translate(center) * scale(radius) and its inverse */
    float   center[3], radius, sphere_inv_area;
    /* BSDF: 0 diffuse (reflectance above) | 1 conductor (src/bsdfs/conductor.cpp) | 2 dielectric (src/bsdfs/dielectric.cpp);
     * `twosided` wraps kinds 0 and 1 (src/bsdfs/twosided.cpp) */
    int32_t bsdf;
    float   cond_eta[3], cond_k[3];            /* complex index of refraction per RGB channel */
    float   spec_refl[3], spec_trans[3];       /* specular_reflectance / specular_transmittance */
    float   diel_eta;                          /* int_ior / ext_ior */
    /* plastic (src/bsdfs/plastic.cpp): diffuse_reflectance = reflectance above, specular_reflectance = spec_refl, eta = diel_eta,
     * and the constants of SmoothPlastic::parameters_changed (:201-217), filled by orc_plastic_params */
    int32_t nonlinear;
    float   inv_eta_2, fdr_int, spec_sampling_weight;
    /* roughconductor (src/bsdfs/roughconductor.cpp) with the GGX distribution and visible-normal sampling
     * (include/mitsuba/render/microfacet.h): cond_eta / cond_k / spec_refl as for the conductor + the two roughness values */
    float   alpha_u, alpha_v;
    /* roughplastic (src/bsdfs/roughplastic.cpp), GGX + visible normals: the plastic fields with fdr_int = m_internal_reflectance,
     * alpha_u = alpha, and m_external_transmittance (64 values, orc_roughplastic_tables) */
    const float *rough_table;
    const orc_texture *tex_refl;   /* texture on `reflectance` / `diffuse_reflectance` (NULL: the constant colour above) */
    int32_t mf_type;         /* microfacet distribution of the rough BSDFs: 0 beckmann, 1 ggx (microfacet.h MicrofacetType) */
    int32_t sample_all;      /* rough BSDFs: sample_visible = false (sample all normals, Walter et al.'s roughness scaling; microfacet.h:240-290) */
    /* textures on the other slots (NULL: the constants above): specular_reflectance / specular_transmittance (Texture::eval) and the
     * roughness alpha / alpha_u / alpha_v of roughconductor / roughdielectric (Texture::eval_1) */
    const orc_texture *tex_spec, *tex_trans, *tex_alpha_u, *tex_alpha_v;
    /* the BSDF above sits inside a `mask` (src/bsdfs/mask.cpp): m_opacity as a constant or a texture (Texture::eval_1 per hit) */
    int32_t masked; float opacity; const orc_texture *tex_opacity;
    /* the plain BSDF sits inside a `normalmap` (src/bsdfs/normalmap.cpp; a twosided around it is applied first): its RGB texture (Texture::eval_3 per hit) */
    const orc_texture *tex_normal;
    /* ... or inside a `bumpmap` (src/bsdfs/bumpmap.cpp): tex_normal is its height texture (Texture::eval_1_grad per hit), bump_scale its `scale` */
    int32_t bumpmap; float bump_scale;
    /* texture on the `radiance` of the shape's area emitter (src/emitters/area.cpp: the emitter is then sampled through the texture; rectangles only) */
    const orc_texture *tex_radiance;
    /* `blendbsdf` (src/bsdfs/blendbsdf.cpp): everything above describes bsdf_0 (with its own twosided / normalmap / bumpmap); blend_other = a record whose BSDF fields
     * describe bsdf_1 (its geometry fields are unused); the weight is a constant or a texture (Texture::eval_1 per hit).  NULL: no blend. */
    const struct orc_shape_s *blend_other; float blend_weight; const orc_texture *tex_blend;
    /* `twosided` with TWO nested BSDFs (twosided.cpp:75-86): everything above describes the front side's, blend_other the back side's (both records two-sided) */
    int32_t two_bsdfs;
} orc_shape;

typedef struct {
    int32_t first_shape, n_shapes;   /* children = shapes[first_shape .. +n_shapes) */
} orc_group;

typedef struct {
    int32_t kind;            /* ORC_OBJ_* */
    int32_t index;           /* shape index (SHAPE) or group index (INSTANCE) */
    int32_t n_keys;          /* INSTANCE: 1 (static transform) or 2 (animated) */
    float   key_time[2];
    float   key[2][16];      /* keyframe matrices (float cast of the double compose) */
} orc_object;

typedef struct {
    int32_t kind;            /* ORC_EMITTER_* */
    float   position[3];     /* point */
    float   intensity[3];    /* point: intensity; area: radiance */
    int32_t shape;           /* area: index into shapes[] of the rectangle that carries it */
    /* spot (src/emitters/spot.cpp:75-100): inverse of to_world (float cast of the double inverse), falloff constants (orc_spot_params) */
    float   to_local[16], cutoff_angle, cos_cutoff, cos_beam, inv_transition;
    /* constant (src/emitters/constant.cpp): intensity = radiance; m_bsphere = the scene's bounding sphere, enlarged (set_scene, :73-83): centre[3], radius
     * (orc_scene_bsphere) */
    float   bsphere[4];
    /* envmap (src/emitters/envmap.cpp): the tables built by orc_envmap_create; to_local = inverse of to_world, bsphere as above */
    const struct orc_envmap *envmap;
    float   env_to_world[16];
} orc_emitter;

/* EnvironmentMapEmitter (src/emitters/envmap.cpp:130-224) with its Hierarchical2D<Float, 0> warp (include/mitsuba/core/distr_2d.h:376-482) */
#define ORC_ENV_MAX_LEVELS 32
typedef struct orc_envmap {
    int32_t w, h;             /* resolution of m_data: bitmap width + 1 (periodic column), bitmap height */
    float   scale;            /* m_scale */
    float  *data;             /* h * w * 3 */
    int32_t n_levels;         /* m_levels.size() */
    float  *level[ORC_ENV_MAX_LEVELS]; int32_t level_w[ORC_ENV_MAX_LEVELS], level_size[ORC_ENV_MAX_LEVELS];
    float   patch_size[2], inv_patch_size[2]; uint32_t max_patch[2];
} orc_envmap;
orc_envmap *orc_envmap_create(const float *rgb, int32_t width, int32_t height, float scale);
orc_envmap *orc_envmap_create2(const float *rgb, int32_t width, int32_t height, float scale, int32_t mis_compensation);
void     orc_envmap_free(orc_envmap *e);
orc_envmap *orc_hier2d_create(const float *values, int32_t width, int32_t height, int32_t normalize);   /* Hierarchical2D<Float, 0> over a plain grid (test_distr_2d.py) */
/* known-answer entry points: Hierarchical2D::sample / eval, the emitter's sample_direction / pdf_direction / eval */
void     orc_envmap_warp_sample(const orc_envmap *e, float sx, float sy, float *uv2_pdf);
float    orc_envmap_warp_eval(const orc_envmap *e, float x, float y);
void     orc_envmap_sample_direction(const orc_emitter *em, const float *ref_p, float sx, float sy, float *d_dist_pdf_w8);
float    orc_envmap_pdf_direction(const orc_emitter *em, const float *d);
void     orc_envmap_eval(const orc_emitter *em, const float *d, float *rgb);
float    orc_atan2f(float y, float x);

typedef struct {
    float   to_world[16];
    float   x_fov;           /* degrees, float cast of parse_fov() */
    float   near_clip, far_clip;
    float   shutter_open, shutter_close;
    int32_t film_w, film_h;
    int32_t crop_x, crop_y, crop_w, crop_h;
    int32_t filter;          /* ORC_FILTER_* */
    float   filter_radius;
    float   filter_stddev;   /* gaussian only (radius = 4 stddev, src/rfilters/gaussian.cpp:48-53) */
    float   filter_b, filter_c;   /* mitchell only: the B and C of the paper (src/rfilters/mitchell.cpp:38-45), radius 2 */
    int32_t kind;            /* ORC_SENSOR_*: src/sensors/{perspective,thinlens,orthographic}.cpp */
    float   aperture_radius; /* thinlens only (thinlens.cpp:142-147: 0 becomes dr::Epsilon<Float>) */
    float   focus_distance;  /* thinlens only (src/render/sensor.cpp:134: default far_clip) */
} orc_sensor;
enum { ORC_SENSOR_PERSPECTIVE = 0, ORC_SENSOR_THINLENS = 1, ORC_SENSOR_ORTHOGRAPHIC = 2 };

typedef struct {
    /* dopplertofpath.cpp:19-57 (all already rounded the way the ctor rounds them) */
    float   time;                 /* T */
    float   w_g_mhz, g_1, g_0, w_s_mhz;
    float   phase_offset;         /* m_sensor_modulation_phase_offset */
    float   hetero_frequency;     /* m_hetero_frequency */
    int32_t wave_type;
    int32_t low_frequency_component_only;
    /* integrator.cpp:54-100, 568-585 */
    int32_t time_sampling;
    float   antithetic_shift;
    int32_t stratify_each_interval;
    uint32_t path_correlation_depth;
    uint32_t max_depth;           /* -1 -> 0xffffffff */
    uint32_t rr_depth;
    int32_t hide_emitters;
    /* sampler (correlated.cpp:17-23, sampler.cpp:11-20) */
    uint32_t base_seed;
    int32_t time_correlate_number, path_correlate_number;
    /* 0 dopplertofpath | 1 path (src/integrators/path.cpp) | 2 velocity (src/integrators/velocity.cpp) -- SURVEY 8(f) #1 */
    int32_t integrator;
    int32_t sampler;         /* 0 correlated | 1 independent | 2 timestratified (src/samplers/{independent,timestratified}.cpp) */
    int32_t jitter;          /* timestratified.cpp:73 */
    uint32_t samples_per_pass;   /* SamplingIntegrator::m_samples_per_pass (integrator.cpp:54-56): 0xffffffff = one pass */
} orc_params;

typedef struct {
    const orc_shape   *shapes;   int32_t n_shapes;
    const orc_group   *groups;   int32_t n_groups;
    const orc_object  *objects;  int32_t n_objects;
    const orc_emitter *emitters; int32_t n_emitters;
    orc_sensor sensor;
} orc_scene;

/* Per-lane debug record: everything a lane-for-lane parity test wants. */
typedef struct {
    float    sample_pos[2];
    float    time;           /* after the wrap of dopplertofpath.cpp:93 */
    float    ray_o[3], ray_d[3];
    float    rgb[3];
    float    path_length;    /* at loop exit */
    uint32_t depth;          /* at loop exit */
    uint32_t valid;
} orc_lane;

/* ---- known-answer building blocks (exported for the KAT tests) ---- */
void     orc_tea32(uint32_t v0, uint32_t v1, int rounds, uint32_t *o0, uint32_t *o1);
float    orc_tea_float32(uint32_t v0, uint32_t v1, int rounds);
void     orc_pcg32_seed(uint64_t initstate, uint64_t initseq, uint64_t *state, uint64_t *inc);
uint32_t orc_pcg32_next_u32(uint64_t *state, uint64_t inc);
float    orc_pcg32_next_f32(uint64_t *state, uint64_t inc);
uint32_t orc_permute_kensler(uint32_t index, uint32_t n, uint32_t seed);
float    orc_waveform(float t, int wave_type);
float    orc_waveform_low_pass(float t, int wave_type);
float    orc_modulation_weight(const orc_params *p, float ray_time, float path_length);
void     orc_sincos(float x, float *s, float *c);

/* Sampler stream for one lane (KAT / lane parity): fills seeds and the first draws.
 * out_u32[0..5] = (rng.state, rng_time.state, rng_path.state) as lo,hi pairs after seeding,
 * out_u32[6] = permutation seed; out_f[0..1] = pixel jitter, out_f[2] = next_1d_time. */
void     orc_sampler_lane(const orc_params *p, uint32_t seed, uint32_t spp, uint32_t lane,
                          uint32_t *out_u32, float *out_f);

/* Camera ray for a film position (perspective.cpp:238-279). out[0..2]=o, [3..5]=d, [6]=maxt */
void     orc_camera_ray(const orc_sensor *s, float px, float py, float *out);
void     orc_camera_sample_ray(const orc_sensor *s, float ux, float uy, float a_x, float a_y, float *out);

/* Closest hit / occlusion against the flat scene (brute force).
 * hit[0]=t (inf if none), hit[1]=u, hit[2]=v; ids[0]=object, ids[1]=shape_in_group, ids[2]=prim */
int      orc_intersect(const orc_scene *sc, const float *o, const float *d, float time, float maxt,
                       float *hit, int32_t *ids);
int      orc_occluded(const orc_scene *sc, const float *o, const float *d, float time, float maxt);

/* Evaluate lanes [lane_begin, lane_begin+n) of the wavefront of W*H*spp lanes.  With several passes (samples_per_pass, or a wavefront
 * beyond 2^32 - 1 lanes) the index is pass * wavefront_size + lane, wavefront_size = W*H*spp_per_pass (orc_pass_layout). */
/* ConstantBackgroundEmitter::set_scene (constant.cpp:73-83): bounding sphere of Scene::bbox() (the union of the shapes' bboxes, instances over
 * their first and last keyframe, scene.cpp:42, instance.cpp:101-114), radius = max(RayEpsilon, r * (1 + RayEpsilon)); an empty scene: centre 0, radius 1 */
void     orc_scene_bsphere(const orc_scene *sc, float *out4);
int      orc_pass_layout(int32_t crop_w, int32_t crop_h, uint32_t spp, uint32_t samples_per_pass, uint32_t *spp_per_pass, uint32_t *n_passes);
void     orc_render_lanes(const orc_scene *sc, const orc_params *p, uint32_t seed, uint32_t spp,
                          uint64_t lane_begin, uint64_t n, orc_lane *out, int n_threads);

/* Full render of pixel rows [row_begin,row_end) (whole film: 0,H).
 * film_rgbw: crop_h*crop_w*4 floats (accumulated, must be zeroed by the caller);
 * out_rgb: crop_h*crop_w*3 developed image or NULL. Returns number of paths traced. */
uint64_t orc_render(const orc_scene *sc, const orc_params *p, uint32_t seed, uint32_t spp,
                    int32_t row_begin, int32_t row_end, float *film_rgbw, float *out_rgb,
                    int n_threads);
/* the same film with the splat terms summed in float64 (the order-independent value the float32 scatter-adds approximate); film: H*W*4 doubles */
uint64_t orc_render_exact(const orc_scene *sc, const orc_params *p, uint32_t seed, uint32_t spp,
                          int32_t row_begin, int32_t row_end, double *film, float *out_rgb, int n_threads);

void     orc_develop(const float *film_rgbw, float *out_rgb, int64_t n_pixels);

/* Cube mesh baking (cube.cpp:114-160): pos[72], nrm[72], uv[48], faces[36] */
void     orc_bake_cube(const float *to_world, const float *to_object, float *pos, float *nrm,
                       float *uv, uint32_t *faces);

/* OBJMesh / PLYMesh vertex baking (obj.cpp:218-246, ply.cpp:284-300): positions through to_world, vertex normals
 * through its inverse transpose then normalised.  nrm_in == NULL and !face_normals: the normals are computed as
 * Mesh::recompute_vertex_normals does (mesh.cpp:257-345, angle-weighted face normals; the reference accumulates them
 * with unordered float atomics, here: in face order, in double, rounded once).  nrm_out may be NULL iff face_normals. */
void     orc_bake_mesh(const float *to_world, const float *to_object, int32_t n_vertices, const float *pos_in,
                       const float *nrm_in, int32_t n_faces, const uint32_t *faces, int32_t face_normals,
                       float *pos_out, float *nrm_out);

/* fresnel (fresnel.h:21-63) -> out4 = r, cos_theta_t, eta_it, eta_ti ; fresnel_conductor (fresnel.h:93-117), one channel */
void     orc_fresnel_dielectric(float cos_theta_i, float eta, float *out4);
float    orc_fresnel_conductor(float cos_theta_i, float eta, float k);

/* SmoothPlastic::parameters_changed (plastic.cpp:201-217) + fresnel_diffuse_reflectance (fresnel.h:328-355), float32:
 * out3 = 1 / eta^2, fdr_int = fresnel_diffuse_reflectance(1 / eta), specular sampling weight s_mean / (d_mean + s_mean) */
/* RoughPlastic::parameters_changed (src/bsdfs/roughplastic.cpp:222-257) for a GGX distribution: table64 = m_external_transmittance
 * (eval_transmittance, include/mitsuba/render/microfacet.h:515-566, on mu = max(1e-6, linspace(0, 1, 64))), *internal_reflectance =
 * mean(eval_reflectance(1 / eta) * mu) * 2 (microfacet.h:463-512); Gauss-Legendre nodes from core/quad.h:27-86 */
void     orc_roughplastic_tables(int type, float alpha, float eta, float *table64, float *internal_reflectance);
void     orc_gauss_legendre(int n, float *nodes, float *weights);
/* SpotLight constructor (src/emitters/spot.cpp:91-99) in float32: degrees -> out4 = cutoff (rad), cos(cutoff), cos(beam), 1 / (cutoff - beam) */
void     orc_spot_params(float cutoff_deg, float beam_deg, float *out4);
float    orc_acos(float x);
void     orc_plastic_params(float eta, const float *diffuse3, const float *specular3, float *out3);

/* Sphere ctor + update (sphere.cpp:121-160), all in float32 as ScalarTransform4f is: composed = to_world * translate(center) *
 * scale(radius) (4x4 products, fmadd accumulation over k), its inverse from the factors' analytic inverses in the reverse
 * order; radius = |composed * (1,0,0)|, center = composed * (0,0,0); a mirroring transform (negative determinant) toggles
 * flip_normals; inv_area = rcp(4 pi r^2).  out8 = center[3], radius, inv_area, flip (as float 0/1), 2 spare. */
void     orc_bake_sphere(const float *to_world, const float *to_object, const float *center, float radius, int32_t flip_normals,
                         float *composed, float *composed_inv, float *out8);

/* Cylinder ctor + update (src/shapes/cylinder.cpp:100-147) in float32: composed = to_world * translate(p0) * to_frame(Frame3f((p1 - p0) / |p1 - p0|)) *
 * scale(radius, radius, |p1 - p0|), its inverse from the factors' inverses; out8 = m_radius (|composed * x|), m_length (|composed * z|), 1 / (2 pi r l),
 * flip (as float 0 / 1; a mirroring transform toggles it), 4 spare.  The unit cylinder x^2 + y^2 = 1, 0 <= z <= 1 lives in object space. */
void     orc_bake_cylinder(const float *to_world, const float *to_object, const float *p0, const float *p1, float radius, int32_t flip_normals,
                           float *composed, float *composed_inv, float *out8);

/* Mesh::build_pmf + DiscreteDistribution::compute_cdf (mesh.cpp:478-511, distr_1d.h:205-240): pmf[i] = .5 |e0 x e1| in
 * float32, running sum in double, cdf[i] = float(sum), sum / normalization rounded to float32 once. Returns 0 on success,
 * -1 for an empty mesh / no probability mass. */
int      orc_mesh_area_table(const float *positions, int32_t n_faces, const uint32_t *faces, float *pmf, float *cdf,
                             float *sum, float *norm, int32_t *lo, int32_t *hi);

/* ---- restated Dr.Jit math (Cephes expf / logf / tanf, Cephes series + A&S 7.1.26 erf, Giles erfinv) */
float    orc_expf(float x);
float    orc_logf(float x);
float    orc_tanf(float x);
float    orc_erff(float x);
float    orc_erfinvf(float x);

/* ---- known-answer entry points (see the end of dtof_oracle.c) */
void     orc_kat_microfacet(int type, float au, float av, int visible, int fn, const float *in, float *out);
float    orc_kat_filter(int kind, float radius, float stddev, float B, float C, float x);
void     orc_kat_warp(int fn, const float *in, float *out);
void     orc_kat_frame(const float *n, float *out6);
int      orc_kat_ray_intersect(const orc_scene *sc, const float *o, const float *d, float time, float maxt, float *out25, int32_t *ids);
void     orc_kat_bsdf(const orc_shape *sh, const float *wi, const float *wo, const float *s3, float *out13);
void     orc_texture_eval(const orc_texture *tex, float u, float v, float *out3);
float    orc_texture_eval_1(const orc_texture *tex, float u, float v);   /* Texture::eval_1: a 1-channel texel, the luminance of an RGB texel, the mean of a checkerboard colour */
void     orc_kat_sphere_sample_direction(const orc_shape *sh, const float *ref, float s_x, float s_y, float *out11);
float    orc_kat_shape_area(const orc_shape *sh);
/* Emitter::sample_direction of emitter `emitter_index` for the reference point `ref` and a 2-D sample: out = d[3], dist, pdf, delta, weight[3], p[3], usable */
void     orc_kat_emitter_sample(const orc_scene *sc, int emitter_index, const float *ref, float sx, float sy, float *out13);
void     orc_kat_splat(const orc_sensor *se, float *film, float x, float y, const float *rgb);
int      orc_kat_solve_quadratic(double a, double b, double c, double *out2);
void     orc_kat_texture_eval_1_grad(const orc_texture *t, float u, float v, float *out2);   /* d eval_1 / d(u, v): bitmap.cpp eval_1_grad */

#ifdef __cplusplus
}
#endif
#endif
