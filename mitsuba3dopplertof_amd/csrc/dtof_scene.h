// dtof_scene.h -- host-side scene description, plugin parameters, and the flat device
// "scene blob" the HIP kernels traverse.
#pragma once
#include <stdint.h>
#include <string>
#include <memory>
#include <vector>
#include <map>
#include <stdexcept>

namespace dtof {

// ---------------------------------------------------------------------------- enums
enum ShapeKind : uint32_t { SHAPE_RECT = 0, SHAPE_MESH = 1, SHAPE_SPHERE = 2, SHAPE_DISK = 3, SHAPE_CYLINDER = 4 };
enum ObjectKind : uint32_t { OBJ_SHAPE = 0, OBJ_INSTANCE = 1 };
enum WaveType : int32_t { WAVE_SIN = 0, WAVE_RECT = 1, WAVE_TRI = 2, WAVE_TRAP = 3 };
// ETimeSampling -- include/mitsuba/render/sampler.h:27-34
enum TimeSampling : int32_t { TIME_UNIFORM = 0, TIME_STRATIFIED = 1, TIME_ANTITHETIC = 2, TIME_ANTITHETIC_MIRROR = 3, TIME_PERIODIC = 4, TIME_REGULAR = 5 };   // ETimeSampling, include/mitsuba/render/sampler.h:27-34
enum FilterKind : int32_t { FILTER_BOX = 0, FILTER_TENT = 1, FILTER_GAUSSIAN = 2, FILTER_MITCHELL = 3, FILTER_CATMULLROM = 4, FILTER_LANCZOS = 5 };
enum ShapeFlags : uint32_t { SF_TWOSIDED = 1, SF_FLIP_NORMALS = 2, SF_FACE_NORMALS = 4, SF_EMITTER = 8, SF_BECKMANN = 16 /* rough BSDFs: Beckmann instead of GGX */, SF_SAMPLE_ALL = 64 /* rough BSDFs: sample_visible = false */, SF_MASK = 128 /* the BSDF sits inside a `mask` (src/bsdfs/mask.cpp): DShape::opacity / tex_opacity */,
                            SF_NORMALMAP = 256 /* the BSDF sits inside a `normalmap` (src/bsdfs/normalmap.cpp): DShape::tex_normal; a twosided around it is applied first */,
                            SF_BUMPMAP = 512 /* ... or inside a `bumpmap` (src/bsdfs/bumpmap.cpp): tex_normal is its height texture, bump_scale its `scale` */,
                            SF_BLEND = 1024 /* `blendbsdf` (src/bsdfs/blendbsdf.cpp): this record is bsdf_0, DShape::blend_other the record of bsdf_1 */,
                            SF_TWOSIDED2 = 2048 /* `twosided` with two nested BSDFs (twosided.cpp:75-86): this record is the front side's, DShape::blend_other the back side's */,
                            SF_TEXCOORDS = 32 /* mesh with vertex texcoords (si.uv interpolates them; otherwise si.uv = the barycentrics, mesh.cpp:720-737) */ };
enum BsdfKind : uint32_t { BSDF_DIFFUSE = 0, BSDF_CONDUCTOR = 1, BSDF_DIELECTRIC = 2, BSDF_PLASTIC = 3, BSDF_ROUGHCONDUCTOR = 4, BSDF_ROUGHPLASTIC = 5, BSDF_THINDIELECTRIC = 6, BSDF_ROUGHDIELECTRIC = 7, BSDF_NULL = 8 /* src/bsdfs/null.cpp: every sample passes straight through */ };
enum EmitterKind : uint32_t { EMITTER_POINT = 0, EMITTER_AREA = 1, EMITTER_SPOT = 2, EMITTER_CONSTANT = 3, EMITTER_ENVMAP = 4, EMITTER_DIRECTIONAL = 5 };

// ---------------------------------------------------------------------------- device blob records
// One contiguous byte blob (offsets from its base) so that small scenes can be staged
// whole into LDS and large ones can stage just the top of the TLAS.
struct BlobHeader {
    uint32_t n_nodes, n_objects, n_groups, n_shapes, n_tris, n_emitters;
    uint32_t off_nodes, off_objects, off_groups, off_shapes, off_tris, off_shading, off_emitters;
    uint32_t total_bytes, off_tables, tlas_depth, off_flat;      // off_tables: face distributions of mesh emitters (float / uint32 words)   // tlas_depth: stack entries a traversal can need (TLAS depth + deepest BLAS)
    uint32_t off_isect, n_tlas_nodes, off_nodes16;   // off_nodes16: DNode16[n_nodes] of scenes with a BLAS (0: none), see below;           // n_tlas_nodes: the first nodes of the array are the TLAS (the BLAS of the meshes follow)
                           // off_isect: DTriIsect[n_tris], what the triangle test reads   // off_flat: DFlatObject[n_objects] of a rectangle-only scene of at most kFlatObjects objects, else 0
};
static_assert(sizeof(BlobHeader) == 80, "BlobHeader");
// One top-level object of a small rectangle-only scene as trace_flat (dtof_traverse.h) reads it with ONE scalar load: a plain rectangle's
// world -> object matrix (a copy of its DShape::to_object), or the mark of an instance (which takes the general intersect_object).
constexpr uint32_t kFlatObjects = 8;
// The world -> object matrix is stored by COLUMNS (c0 .. c2 = the linear part, c3 = the translation; x, y, z entries each): trace_flat feeds the (x, y) pair of a
// column to ONE packed multiply-add as an SGPR-pair operand (v_pk_fma_f32 issues two multiply-adds in the 4 cycles a scalar-operand v_fma_f32 needs for one,
// profiles/r03_ubench_valu_rate.txt).
struct DFlatObject { float c0[3]; uint32_t instance; float c1[3]; uint32_t pad1; float c2[3]; uint32_t pad2; float c3[3]; uint32_t pad3; };   // 64 B; instance: 0 plain rectangle, 1 instance (general path), 2 instance of ONE rectangle (matrix = that rectangle's, in the group's space)
static_assert(sizeof(DFlatObject) == 64, "DFlatObject");

constexpr uint32_t kLeafFlag = 0x80000000u;
// TLAS leaves only: the object behind this leaf holds a mesh with a BLAS of its own (bit 30; the object index is the 30 bits below).  The ray kernels of large meshes can
// put such objects aside for a second, dense launch instead of entering them (dtof_kernels.hip: DEFER).
constexpr uint32_t kLeafBlas = 0x40000000u, kLeafObjMask = 0x3fffffffu;
constexpr uint32_t kNoChild = 0xffffffffu;
// BLAS (per triangle mesh, nodes appended to the same array): a leaf is kLeafFlag | (first triangle, relative to the mesh's
// first_tri) << kBlasLeafBits | (count - 1); meshes of at most kBlasMinTris triangles are looped over instead.
constexpr uint32_t kBlasLeafBits = 3, kBlasLeaf = 4, kBlasMinTris = 16;   // a leaf can hold up to 1 << kBlasLeafBits triangles; the builder stops splitting at kBlasLeaf (DTOF_BLAS_LEAF=2..8 overrides: development)
// TLAS node (64 B): the bounds of BOTH children live in the parent, so one fetch decides both
// descents.  child = kLeafFlag | object index for a leaf, inner-node index otherwise, kNoChild if absent.
struct BvhNode {
    float lmin[3]; uint32_t left;
    float lmax[3]; uint32_t right;
    float rmin[3]; uint32_t pad0;
    float rmax[3]; uint32_t pad1;
};
// The same tree in 32 bytes per node for the ray kernels of large meshes (k_trace / k_shadow<..., W8>, which saturate the texture-address / data path: a node step is then
// TWO 16-byte loads per lane instead of four): the child boxes as IEEE half floats, minima rounded DOWN and maxima rounded UP on the host (scene_build.cpp), so that a
// half box contains its float box.  The boxes only cull -- which primitive is hit, and where, is decided by the primitive tests on the full-precision records -- so the
// results do not change; the looser boxes cost a few more node steps.  Written only when every coordinate fits a half (|x| <= 65 000); the kernels fall back to DNode otherwise.
struct DNode16 { uint16_t lbox[6]; uint32_t left; uint16_t rbox[6]; uint32_t right; };   // lbox / rbox: min x y z, max x y z
static_assert(sizeof(DNode16) == 32, "DNode16");
typedef BvhNode DNode;   // (a quantised 4-wide node format was measured slower in round 3: tools/experiments/r03_bvh4.patch, profiles/r03_qbvh4_vs_bvh2.txt)
struct DObject {            // 128 B
    uint32_t kind, index, n_keys; float t0;
    float t1, pad[3];
    float key0[12], key1[12];
};
struct DGroup { uint32_t first_shape, n_shapes, pad[2]; };
struct DShape {             // 352 B
    uint32_t kind, flags, first_tri, n_tris;
    float refl[3]; uint32_t blas_root;                   // mesh: root node of its BLAS, kNoChild = loop over the triangles
    float to_world[12], to_object[12];
    // rectangle frame (Rectangle::update, rectangle.cpp:101-113).  Mesh emitters use the three spare words for the face
    // distribution (Mesh::build_pmf, mesh.cpp:478-511): byte offset of its table in the blob = cdf[n] | pmf[n] | slot[n]
    // (slot = position of face i in the BLAS-ordered triangle array), and m_valid = [emit_lo, emit_hi]
    // Spheres (src/shapes/sphere.cpp:138-160) keep m_center in n[] and m_radius in dp_du[0]; to_world / to_object are the composed
    // to_world * translate(center) * scale(radius) and its inverse.
    float n[3]; uint32_t emit_table; float dp_du[3]; uint32_t emit_lo; float dp_dv[3]; uint32_t emit_hi;
    float bmin[3], emit_sum, bmax[3]; uint32_t rough_table;             // padded bounds of the shape in ITS space (mesh: culls the triangle loop); emit_sum = float(sum of areas)
    float radiance[3], inv_area;                        // SF_EMITTER: AreaLight radiance, 1 / area (Rectangle::m_inv_surface_area, DiscreteDistribution::normalization)
    // BSDF: BSDF_DIFFUSE uses refl; BSDF_CONDUCTOR (src/bsdfs/conductor.cpp) cond_eta / cond_k / spec_refl; BSDF_DIELECTRIC
    // (src/bsdfs/dielectric.cpp) diel_eta = int_ior / ext_ior, spec_refl, spec_trans
    // BSDF_PLASTIC (src/bsdfs/plastic.cpp): refl = diffuse_reflectance, spec_refl, diel_eta and the constants of parameters_changed
    uint32_t bsdf; float diel_eta; uint32_t nonlinear; float inv_eta_2;   // nonlinear: bit 0 = the plastics' `nonlinear`; bits 1.. = (byte offset of the reflectance texture's DTexture in the blob) >> 4, 0 = none
    // BSDF_ROUGHCONDUCTOR (src/bsdfs/roughconductor.cpp, GGX + visible normals): cond_eta / cond_k / spec_refl + alpha_u, alpha_v
    // BSDF_ROUGHPLASTIC (src/bsdfs/roughplastic.cpp, GGX + visible normals): the plastic fields + alpha_u; fdr_int = m_internal_reflectance;
    // rough_table = byte offset in the blob of m_external_transmittance (64 floats)
    float cond_eta[3], fdr_int, cond_k[3], spec_sampling_weight, spec_refl[3], alpha_u, spec_trans[3], alpha_v;
    // textures on the other slots, (byte offset of the DTexture in the blob) >> 4, 0 = none: specular_reflectance, specular_transmittance (Texture::eval per hit) and the
    // roughness alpha_u / alpha_v of roughconductor / roughdielectric (Texture::eval_1 per hit; `alpha` fills both)
    uint32_t tex_spec, tex_trans, tex_alpha_u, tex_alpha_v;
    // SF_MASK: m_opacity of the enclosing `mask` BSDF (mask.cpp:95): the constant (or the texture's mean) and the texture record, as above (Texture::eval_1 per hit)
    float opacity; uint32_t tex_opacity;
    // SF_BLEND: index (into shapes[]) of the material-only record that describes bsdf_1, m_weight as a constant (or the texture's mean) and as a texture (Texture::eval_1 per hit)
    uint32_t blend_other; float blend_weight; uint32_t tex_blend;
    uint32_t tex_radiance;                               // SF_EMITTER: texture on the area emitter's `radiance` (record offset >> 4, 0 = the constant above); rectangles only (area.cpp:129-153)
    uint32_t tex_normal; float bump_scale;               // SF_NORMALMAP: m_normalmap of the enclosing `normalmap` BSDF (normalmap.cpp:97), Texture::eval_3 per hit;
                                                         // SF_BUMPMAP: m_nested_texture (Texture::eval_1_grad per hit) and m_scale of the enclosing `bumpmap` (bumpmap.cpp:93-112)
};
// Texture on a BSDF's diffuse reflectance (src/textures/checkerboard.cpp, src/textures/bitmap.cpp); the record and, for bitmaps, the
// linear float32 texels (row 0 first) live in the tables area of the blob.  to_uv: the 2x2 linear part of the plugin's `to_uv`
// (Transform4f::extract, transform.h:340-360, copies the upper-left block and the bottom ROW: a translation is lost there).
enum TextureKind : uint32_t { TEX_CHECKERBOARD = 0, TEX_BITMAP = 1 };
struct DTexture {           // 64 B
    uint32_t kind_flags;    // kind | filter << 8 (0 nearest, 1 bilinear) | wrap << 16 (0 repeat, 1 mirror, 2 clamp) | channels << 24
    uint32_t width, height, data_off;   // data_off: byte offset of the texels in the blob
    float to_uv[4], color0[3], color1[3];
    uint32_t distr_off;     // byte offset in the blob of the texels' DiscreteDistribution2D (distr_2d.h:75-181): normalization, 1 / normalization, marg_cdf[height], cond_cdf[height * width];
    uint32_t pad;           // 0 = none (built for the textures an area emitter's radiance is sampled through, bitmap.cpp:450-528)
};
struct DTri { float p0[3]; uint32_t face; float p1[4], p2[4]; };         // 48 B; face = index in the mesh's own order (tie rule)
// What Moeller-Trumbore reads of a triangle (tri_hit, dtof_traverse.h): the first vertex, the two edges e1 = p0 - p1, e2 = p2 - p0 and ng = cross(e2, e1), computed on the
// host with the very operations (dtof_math.h) the kernels used to repeat for every test -- 15 of the test's 45 instructions.  Same 48 bytes as the vertices; `face`
// (needed when two hits tie, and by shading) stays in DTri.
struct DTriIsect { float p0[3], ngx, e1[3], ngy, e2[3], ngz; };
struct DTriShade { float n0[3], n1[3], n2[3], uv0[2], uv1[2], uv2[2], pad; };   // 64 B
struct DEmitter {           // 96 B
    uint32_t kind; float pos[3]; float intensity[3]; uint32_t shape;   // area: intensity = radiance, shape = index into shapes[]
    // spot (src/emitters/spot.cpp:75-100): world -> local (3x4 affine part of to_world's inverse) and the constants of the falloff curve
    float to_local[12]; float cutoff_angle, cos_cutoff, cos_beam, inv_transition;
    // directional (src/emitters/directional.cpp): intensity = irradiance, to_local[0..2] = the direction of travel, pos / cutoff_angle = m_bsphere as for `constant`
    // constant (src/emitters/constant.cpp): intensity = radiance, pos = centre of m_bsphere, cutoff_angle = its (enlarged) radius (set_scene, :73-83)
};
// EnvironmentMapEmitter (src/emitters/envmap.cpp) in the tables area: m_data (h rows of w = bitmap width + 1 RGB texels) and the levels of its
// Hierarchical2D<Float, 0> warp (include/mitsuba/core/distr_2d.h:376-482; level 0 = the normalised luminance x sin(theta) grid, level k >= 1 in
// 2 x 2-block order, Level::index :766-770).  DEmitter::shape holds the byte offset of this record in the blob; pos / cutoff_angle the bounding
// sphere as for `constant`; to_local the world -> emitter rotation.
constexpr uint32_t kEnvMaxLevels = 24;
struct DEnvmap {
    uint32_t w, h, n_levels, data_off;                       // data_off and level_off: byte offsets in the blob
    float scale, patch_x, patch_y, inv_patch_x;
    float inv_patch_y; uint32_t max_px, max_py, pad;
    float to_world[12];
    uint32_t level_off[kEnvMaxLevels], level_w[kEnvMaxLevels];
};
static_assert(sizeof(DEnvmap) % 16 == 0, "DEnvmap");
static_assert(sizeof(DTexture) == 64, "DTexture");
static_assert(sizeof(BvhNode) == 64 && sizeof(DObject) == 128 && sizeof(DShape) == 352 && sizeof(DTri) == 48 && sizeof(DTriIsect) == 48 && sizeof(DTriShade) == 64 && sizeof(DEmitter) == 96, "blob records");

// ---------------------------------------------------------------------------- host description
struct Mat4d { double m[16]; };   // row-major

struct HostShape {
    uint32_t kind = SHAPE_RECT;
    bool twosided = false, flip_normals = false, face_normals = false;
    float refl[3] = { .5f, .5f, .5f };
    uint32_t bsdf = BSDF_DIFFUSE;   // + the parameters of the specular BSDFs
    float cond_eta[3] = { 0, 0, 0 }, cond_k[3] = { 1, 1, 1 }, spec_refl[3] = { 1, 1, 1 }, spec_trans[3] = { 1, 1, 1 }, diel_eta = 1.f;
    bool nonlinear = false; float inv_eta_2 = 1.f, fdr_int = 0.f, spec_sampling_weight = 0.f;   // plastic
    float alpha_u = .1f, alpha_v = .1f;   // roughconductor, roughplastic
    bool beckmann = false;                 // their `distribution` (microfacet.h MicrofacetType)
    bool sample_all = false;               // their `sample_visible` = false: all normals are sampled (microfacet.h:240-290), roughdielectric scales its roughness for sampling
    int tex_refl = -1;                     // texture on reflectance / diffuse_reflectance: index into HostScene::textures
    int tex_spec = -1, tex_trans = -1, tex_alpha_u = -1, tex_alpha_v = -1;   // textures on specular_reflectance / specular_transmittance / the roughness (alpha sets both)
    int tex_normal = -1;                                                      // the BSDF sits inside a `normalmap` (src/bsdfs/normalmap.cpp): its RGB texture
    bool two_bsdfs = false;                                                   // `twosided` with two nested BSDFs: *blend_other is the back side's
    std::shared_ptr<HostShape> blend_other; float blend_weight = .5f; int tex_blend = -1;   // `blendbsdf`: the fields above describe bsdf_0, *blend_other (BSDF fields only) bsdf_1
    int tex_radiance = -1;                                                    // texture on the area emitter's radiance
    bool bumpmap = false; float bump_scale = 1.f;                             // ... or inside a `bumpmap` (src/bsdfs/bumpmap.cpp): tex_normal is the height texture
    bool masked = false; float opacity = 1.f; int tex_opacity = -1;           // the BSDF sits inside a `mask` (src/bsdfs/mask.cpp): its opacity (float or texture, eval_1)
    std::vector<float> rough_table;        // roughplastic: m_external_transmittance (64 values)
    float to_world[16], to_object[16];     // float casts of the double transform and its double inverse
    // mesh: cube baked like src/shapes/cube.cpp:114-160; obj / ply through mesh_io.cpp
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> faces;
    float center[3] = { 0, 0, 0 }, radius = 1.f, sphere_inv_area = 0.f;   // sphere: m_center, m_radius, m_inv_surface_area after update()
    std::string id;
    bool emitter = false; float radiance[3] = { 0, 0, 0 };   // area emitter attached to the shape (src/emitters/area.cpp)
};
struct HostTexture {
    uint32_t kind = TEX_CHECKERBOARD, filter = 1, wrap = 0, channels = 3, width = 0, height = 0;
    float to_uv[4] = { 1, 0, 0, 1 }, color0[3] = { .4f, .4f, .4f }, color1[3] = { .2f, .2f, .2f };
    std::vector<float> data;       // bitmap: linear float32 texels
    float mean = 0.f;              // Texture::mean()
};
struct HostGroup { uint32_t first_shape = 0, n_shapes = 0; };
struct HostObject {
    uint32_t kind = OBJ_SHAPE, index = 0, n_keys = 0;
    float key_time[2] = { 0, 0 };
    float key[2][16];
};
struct HostEmitter { uint32_t kind = 0; float pos[3] = { 0, 0, 0 }; float intensity[3] = { 0, 0, 0 }; uint32_t shape = 0xffffffffu;
                     float to_local[12] = { 0 }, cutoff_angle = 0, cos_cutoff = 0, cos_beam = 0, inv_transition = 0;    // spot
                     std::vector<float> image; uint32_t image_w = 0, image_h = 0; float scale = 1.f, to_world[12] = { 0 }; bool mis_compensation = false; };   // envmap: linear RGB rows (top first), m_scale, emitter -> world
struct HostSensor {
    float to_world[16];
    float x_fov = 0, near_clip = 1e-2f, far_clip = 1e4f, shutter_open = 0, shutter_close = 0;
    int32_t film_w = 768, film_h = 576, crop_x = 0, crop_y = 0, crop_w = 768, crop_h = 576;
    int32_t filter = FILTER_TENT; float filter_radius = 1.f, filter_stddev = .5f, filter_b = 1.f / 3.f, filter_c = 1.f / 3.f;   // B, C: mitchell
    bool orthographic = false;   // src/sensors/orthographic.cpp
    bool thinlens = false; float aperture_radius = 0.f, focus_distance = 0.f;   // src/sensors/thinlens.cpp:138-156, src/render/sensor.cpp:134
    bool alpha = false;          // hdrfilm pixel_format = rgba: FilmFlags::Alpha (src/films/hdrfilm.cpp:172-177) -- develop() returns R, G, B, A
};

// A typed property bag (what the reference's Properties carries for a plugin).
struct PropValue { enum Type { Float, Int, Bool, String } type; double f = 0; int64_t i = 0; bool b = false; std::string s; };
struct PropBag {
    std::string plugin;
    std::map<std::string, PropValue> values;
    mutable std::map<std::string, bool> queried;
    bool has(const std::string &n) const { return values.count(n) != 0; }
    double get_float(const std::string &n, double def) const;
    int64_t get_int(const std::string &n, int64_t def) const;
    bool get_bool(const std::string &n, bool def) const;
    std::string get_string(const std::string &n, const std::string &def) const;
    std::vector<std::string> unqueried() const;
};

// Constructor-time parameters of DopplerToFPathIntegrator (+ bases) and CorrelatedSampler, rounded
// exactly as the reference's constructors round them.
enum SamplerKind : int32_t { SAMPLER_CORRELATED = 0, SAMPLER_INDEPENDENT = 1, SAMPLER_TIMESTRATIFIED = 2 };
enum IntegratorKind : int32_t { INTEGRATOR_DOPPLER = 0, INTEGRATOR_PATH = 1, INTEGRATOR_VELOCITY = 2 };
struct PluginParams {
    int32_t integrator = INTEGRATOR_DOPPLER;   // dopplertofpath | path | velocity
    // sampler plugin: correlated | independent (src/samplers/independent.cpp: the main stream only; Sampler::next_1d_time and
    // next_*_correlate fall back to next_1d, include/mitsuba/render/sampler.h:131-144) | timestratified (src/samplers/timestratified.cpp)
    int32_t sampler_kind = SAMPLER_CORRELATED; bool jitter = true;
    // src/integrators/dopplertofpath.cpp:19-57
    float time = 0.0015f, w_g_mhz = 30.f, g_1 = .5f, g_0 = .5f, w_s_mhz = 30.f, phase_offset = 0.f, hetero_frequency = 0.f;
    int32_t wave_type = WAVE_SIN; bool low_frequency_component_only = true;
    // src/render/integrator.cpp:22-28, 54-100, 568-585
    int32_t time_sampling = TIME_ANTITHETIC; float antithetic_shift = .5f; bool stratify_each_interval = true;
    uint32_t path_correlation_depth = 0, max_depth = 0xffffffffu, rr_depth = 5; bool hide_emitters = false;
    uint32_t samples_per_pass = 0xffffffffu;   // (uint32_t) -1: one pass
    // src/samplers/correlated.cpp:17-23, src/render/sampler.cpp:11-20
    uint32_t base_seed = 0, sample_count = 4; int32_t time_correlate_number = 2, path_correlate_number = 2;
};
void read_png(const std::string &path, std::vector<uint8_t> &pixels, uint32_t &width, uint32_t &height, uint32_t &channels);   // image_io.cpp
void read_jpeg(const std::string &path, std::vector<uint8_t> &pixels, uint32_t &width, uint32_t &height, uint32_t &channels);   // image_io.cpp: baseline JPEG
void read_radiance_image(const std::string &path, std::vector<float> &rgb, uint32_t &width, uint32_t &height, float (*srgb_to_linear_u8)(uint32_t));   // image_io.cpp: RGBE, PFM, PNG
PluginParams make_plugin_params(const PropBag &integrator, const PropBag &sampler);   // throws std::runtime_error

struct HostScene {
    std::vector<HostShape> shapes;
    std::vector<HostGroup> groups;
    std::vector<HostObject> objects;
    std::vector<HostEmitter> emitters;
    std::vector<HostTexture> textures;
    HostSensor sensor; bool has_sensor = true;   // a scene without a <sensor> loads, as in the reference; rendering it fails
    PropBag integrator, sampler;
};

// XML front end (scene_loader.cpp): the tag subset of SURVEY §8a row X1.  `base_dir` is what the reference's FileResolver
// holds for a scene file (its directory): relative `filename` properties of obj / ply shapes resolve against it.
HostScene load_scene_xml(const std::string &text, const std::map<std::string, std::string> &params, const std::string &base_dir = "");
std::string read_file(const std::string &path);

// Mesh files (mesh_io.cpp): raw object-space arrays of an .obj / .ply file, and the constructor-time baking
struct RawMesh {
    std::vector<float> positions, normals, texcoords; std::vector<uint32_t> faces;
    bool has_normals = false, has_texcoords = false;
};
RawMesh load_obj(const std::string &path, bool flip_tex_coords, bool face_normals);   // src/shapes/obj.cpp
RawMesh load_ply(const std::string &path, bool face_normals);                         // src/shapes/ply.cpp
RawMesh load_serialized(const std::string &path, int shape_index, bool face_normals); // src/shapes/serialized.cpp
void bake_mesh(HostShape &s, const RawMesh &raw);   // to_world / normals / Mesh::recompute_vertex_normals

// Blob + BVH (scene_build.cpp)
std::vector<uint8_t> build_scene_blob(const HostScene &scene);

}  // namespace dtof
