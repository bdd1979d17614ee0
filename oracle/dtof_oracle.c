/*
 * dtof_oracle.c -- scalar CPU restatement of the reference's Doppler-ToF path
 * tracer (`dopplertofpath` integrator + `correlated` sampler and everything they
 * call per lane).  TEST INFRASTRUCTURE ONLY -- see dtof_oracle.h.
 *
 * Reference = /root/reference (juhyeonkim95/Mitsuba3DopplerToF @ 2024_08_07), JIT
 * ("llvm_rgb") semantics: one lane per (pixel, sample), Float = float32,
 * Spectrum = Color3f.  Citations are relative to the reference root.
 *
 * PARITY: unpinned for the path as a whole (the reference has no test/fixture for
 * it and cannot be built here); pinned building blocks: TEA, PCG32, Kensler,
 * waveforms (tests/test_oracle_kat.py).
 *
 * Arithmetic conventions (compiled with -ffp-contract=off; every fused
 * multiply-add below is an explicit fmaf(), placed where the reference calls
 * dr::fmadd or where Dr.Jit's array primitives are fmadd chains):
 *   dot(a,b)      = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))       (Dr.Jit dot_ = fmadd chain)
 *   cross(a,b)    = fmsub(a.yzx, b.zxy, a.zxy*b.yzx)
 *   normalize(v)  = v * rsqrt(dot(v,v)),  rsqrt(x) = sqrt(1/x)  (Dr.Jit LLVM lowering)
 *   rcp(x)        = 1/x ;  array / lower-depth value = array * rcp(value)
 *   sincos/cos    = Cephes single-precision polynomials (Dr.Jit's own approximations
 *                   are Cephes-based; the source is absent, so this is a restatement
 *                   of the published Cephes algorithm, identical on CPU and GPU)
 *   Third-party pieces absent from the tree (Dr.Jit 0.4.0 PCG32, Embree 3 traversal
 *   and instance-matrix interpolation) are restated from their published
 *   algorithms; see SURVEY.md Appendix C.
 */
#define _GNU_SOURCE
#include "dtof_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

typedef struct { float x, y, z; } v3;

#define ORC_PI_F        3.14159265358979323846f
#define ORC_INV_PI_F    0.31830988618379067154f
#define ORC_RAY_EPS     (1500.f * 5.9604644775390625e-8f)   /* math.h:17-22: 1500 * 2^-24 */
#define ORC_SHADOW_EPS  (ORC_RAY_EPS * 10.f)
#define ORC_LARGEST     3.40282346638528859812e+38f         /* dr::Largest<float> */

/* ------------------------------------------------------------------ helpers */
static inline float f_rcp(float x)   { return 1.0f / x; }
static inline float f_rsqrt(float x) { return sqrtf(1.0f / x); }
static inline float f_sqr(float x)   { return x * x; }
static inline uint32_t f2u(float f)  { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u)  { float f; memcpy(&f, &u, 4); return f; }
/* dr::mulsign(a, b): a with its sign flipped when b's sign bit is set */
static inline float f_mulsign(float a, float b) { return u2f(f2u(a) ^ (f2u(b) & 0x80000000u)); }
static inline float f_mulsign_neg(float a, float b) { return u2f(f2u(a) ^ (~f2u(b) & 0x80000000u)); }
static inline float f_sign(float x) { return u2f(0x3f800000u | (f2u(x) & 0x80000000u)); } /* dr::sign: +-1 */
static inline float f_min(float a, float b) { return a < b ? a : b; }
static inline float f_max(float a, float b) { return a > b ? a : b; }

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_mul(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 v_neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 v_fma(v3 a, float s, v3 c) { return V(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z)); }
static inline float v_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 v_cross(v3 a, v3 b) {
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline v3 v_normalize(v3 a) { return v_mul(a, f_rsqrt(v_dot(a, a))); }
static inline float v_norm(v3 a) { return sqrtf(v_dot(a, a)); }

/* Transform::transform_affine(Point) -- include/mitsuba/core/transform.h:97-105 */
static inline v3 m_point(const float *m, v3 p) {
    return V(fmaf(m[2], p.z, fmaf(m[1], p.y, fmaf(m[0], p.x, m[3]))),
             fmaf(m[6], p.z, fmaf(m[5], p.y, fmaf(m[4], p.x, m[7]))),
             fmaf(m[10], p.z, fmaf(m[9], p.y, fmaf(m[8], p.x, m[11]))));
}
/* Transform::operator*(Vector) -- transform.h:125-134 */
static inline v3 m_vector(const float *m, v3 v) {
    return V(fmaf(m[2], v.z, fmaf(m[1], v.y, m[0] * v.x)),
             fmaf(m[6], v.z, fmaf(m[5], v.y, m[4] * v.x)),
             fmaf(m[10], v.z, fmaf(m[9], v.y, m[8] * v.x)));
}
/* Transform::operator*(Normal) with inverse_transpose = inv^T -- transform.h:140-149.
 * `inv` is the (row-major) inverse matrix, so inverse_transpose(r,c) = inv(c,r). */
static inline v3 m_normal(const float *inv, v3 n) {
    return V(fmaf(inv[8], n.z, fmaf(inv[4], n.y, inv[0] * n.x)),
             fmaf(inv[9], n.z, fmaf(inv[5], n.y, inv[1] * n.x)),
             fmaf(inv[10], n.z, fmaf(inv[6], n.y, inv[2] * n.x)));
}

/* Inverse of an affine 4x4 (last row taken as 0,0,0,1).  The reference inverts the
 * full 4x4 (dr::inverse_transpose inside Transform(Matrix), transform.h:54-56) and
 * then only ever uses the affine part (transform_affine); Dr.Jit's source is
 * absent, so the exact operation order is this file's own, shared with the GPU. */
static void m_affine_inverse(const float *m, float *inv) {
    float a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6],
          a20 = m[8], a21 = m[9], a22 = m[10];
    float c00 = fmaf(a11, a22, -(a12 * a21)), c01 = fmaf(a12, a20, -(a10 * a22)),
          c02 = fmaf(a10, a21, -(a11 * a20));
    float det = fmaf(a02, c02, fmaf(a01, c01, a00 * c00));
    float id = 1.0f / det;
    float i00 = c00 * id, i01 = fmaf(a02, a21, -(a01 * a22)) * id, i02 = fmaf(a01, a12, -(a02 * a11)) * id;
    float i10 = c01 * id, i11 = fmaf(a00, a22, -(a02 * a20)) * id, i12 = fmaf(a02, a10, -(a00 * a12)) * id;
    float i20 = c02 * id, i21 = fmaf(a01, a20, -(a00 * a21)) * id, i22 = fmaf(a00, a11, -(a01 * a10)) * id;
    float tx = m[3], ty = m[7], tz = m[11];
    inv[0] = i00; inv[1] = i01; inv[2] = i02;  inv[3]  = -fmaf(i02, tz, fmaf(i01, ty, i00 * tx));
    inv[4] = i10; inv[5] = i11; inv[6] = i12;  inv[7]  = -fmaf(i12, tz, fmaf(i11, ty, i10 * tx));
    inv[8] = i20; inv[9] = i21; inv[10] = i22; inv[11] = -fmaf(i22, tz, fmaf(i21, ty, i20 * tx));
    inv[12] = 0.f; inv[13] = 0.f; inv[14] = 0.f; inv[15] = 1.f;
}

/* ------------------------------------------------------------ sincos (Cephes) */
/* Restatement of the Cephes sinf/cosf kernel that Dr.Jit's dr::sincos is based on
 * (drjit/math.h, absent): range reduction by pi/4 octants with a 3-term Cody-Waite
 * split, degree-3 polynomials in z=y*y (Estrin form). */
void orc_sincos(float x, float *s_out, float *c_out) {
    float xa = fabsf(x);
    int32_t j = (int32_t) (xa * 1.2732395447351626862f);
    j = (j + 1) & ~1;
    float y = (float) j;
    uint32_t sign_sin = ((uint32_t) j << 29) ^ f2u(x);
    uint32_t sign_cos = (uint32_t) (~(j - 2)) << 29;
    y = xa - y * 0.78515625f;
    y = y - (float) j * 2.4187564849853515625e-4f;
    y = y - (float) j * 3.77489497744594108e-8f;
    float z = y * y;
    float s = fmaf(z * z, -1.9515295891e-4f, fmaf(z, 8.3321608736e-3f, -1.6666654611e-1f)) * z;
    float c = fmaf(z * z, 2.443315711809948e-5f, fmaf(z, -1.388731625493765e-3f, 4.166664568298827e-2f)) * z;
    s = fmaf(s, y, y);
    c = fmaf(c, z, fmaf(z, -0.5f, 1.0f));
    int poly = (j & 2) == 0;
    *s_out = u2f(f2u(poly ? s : c) ^ (sign_sin & 0x80000000u));
    *c_out = u2f(f2u(poly ? c : s) ^ (sign_cos & 0x80000000u));
}
static inline float orc_cos(float x) { float s, c; orc_sincos(x, &s, &c); return c; }
/* acos: the Cephes asinf kernel dr::acos builds on (Dr.Jit's source is not in the tree), Estrin form with fmadd -- SpotLight::falloff_curve */
float orc_acos(float x) {
    float xa = fabsf(x), x2 = x * x;
    int big = xa >= 0.5f;
    float x1 = 0.5f * (1.f - xa), x3 = big ? x1 : x2, x4 = big ? sqrtf(x1) : x;
    float a0 = fmaf(x3, 7.4953002686e-2f, 1.6666752422e-1f), a1 = fmaf(x3, 2.4181311049e-2f, 4.5470025998e-2f), y2 = x3 * x3;
    float z1 = fmaf(y2 * y2, 4.2163199048e-2f, fmaf(y2, a1, a0));
    z1 = fmaf(z1, x3 * x4, x4);
    float z2 = 2.f * z1, z3 = x < 0.f ? ORC_PI_F - z2 : z2, z4 = 0.5f * ORC_PI_F - z1;
    return big ? z3 : z4;
}

/* dr::atan2 (Dr.Jit 0.4 math.h; the submodule is empty in the reference tree): minimax fit of atan(sqrt(z)) / sqrt(z) in z = (min / max)^2,
 * evaluated in Estrin form with fmadd, then unfolded by octant. */
float orc_atan2f(float y, float x) {
    const float xa = fabsf(x), ya = fabsf(y), mn = ya < xa ? ya : xa, mx = xa > ya ? xa : ya;
    const float scale = mn / mx, z = scale * scale;
    const float z2 = z * z, z4 = z2 * z2;
    const float p01 = fmaf(z, -0.33326497518773606976f, 0.99999934166683966009f), p23 = fmaf(z, -0.13486708938456973185f, 0.19881342388439013552f);
    const float p45 = fmaf(z, -0.37006525670417265220e-1f, 0.83863120428809689910e-1f), p6 = 0.78613793713198150252e-2f;
    const float poly = fmaf(z4, fmaf(z2, p6, p45), fmaf(z2, p23, p01));
    float t = scale * poly;
    t = ya > xa ? 0.5f * ORC_PI_F - t : t;
    t = x < 0.f ? ORC_PI_F - t : t;
    float r = y < 0.f ? -t : t;
    return mx != 0.f ? r : 0.f;
}

/* ---------------------------------------------------------------------------------------------------------------- envmap
 * EnvironmentMapEmitter (src/emitters/envmap.cpp) in the rgb variant: the constructor (:130-224: periodic column, luminance x sin(theta),
 * Hierarchical2D), eval (:299-310), sample_direction (:363-406), pdf_direction (:408-425), eval_spectrum (:487-553); Hierarchical2D<Float, 0>
 * (include/mitsuba/core/distr_2d.h: constructor :376-482, sample :490-575, eval :668-699, Level::index :766-770) and the bilinear warps
 * (include/mitsuba/core/warp.h:355-429). */
static inline uint32_t env_level_index(uint32_t x, uint32_t y, uint32_t width) { return ((x & 1u) | (((x & ~1u) | (y & 1u)) << 1)) + ((y & ~1u) * width); }
static inline float f_lerp(float a, float b, float t) { return fmaf(b, t, fmaf(-a, t, a)); }   /* dr::lerp = fmadd(b, t, fnmadd(a, t, a)) */
static inline float f_clamp01(float x) { return f_min(f_max(x, 0.f), 1.f); }
static inline float interval_to_linear(float v0, float v1, float sample) {
    const float val = (v0 - sqrtf(f_max(f_lerp(f_sqr(v0), f_sqr(v1), sample), 0.f))) / (v0 - v1);
    return fabsf(v0 - v1) > 1e-4f * (v0 + v1) ? val : sample;
}
static uint32_t log2i_ceil_u32(uint32_t v) { uint32_t r = 31u - (uint32_t) __builtin_clz(v); if (v & (v - 1u)) r += 1u; return r; }

/* Hierarchical2D<Float, 0>(data, size, normalize) (include/mitsuba/core/distr_2d.h:376-482): level 0 = the (normalised) input grid of W x H
 * values, level k >= 1 = the patch averages, summed 2 x 2 per level, in the blocked order env_level_index() walks */
static void hier2d_build(orc_envmap *e, const float *lum, uint32_t W, uint32_t H, int normalize) {
    const uint32_t npx = W - 1u, npy = H - 1u, max_level = log2i_ceil_u32(npx > npy ? npx : npy);
    e->patch_size[0] = 1.f / (float) npx; e->patch_size[1] = 1.f / (float) npy;
    e->inv_patch_size[0] = (float) npx; e->inv_patch_size[1] = (float) npy;
    e->max_patch[0] = npx - 1u; e->max_patch[1] = npy - 1u;
    e->n_levels = (int32_t) max_level + 2;
    e->level_w[0] = (int32_t) W; e->level_size[0] = (int32_t) (W * H);
    e->level[0] = (float *) calloc((size_t) W * H, sizeof(float));
    {
        uint32_t lx = npx, ly = npy; int32_t k = 1;
        for (int32_t level = (int32_t) max_level; level >= 0; --level, ++k) {
            lx += lx & 1u; ly += ly & 1u;
            e->level_w[k] = (int32_t) lx; e->level_size[k] = (int32_t) (lx * ly);
            e->level[k] = (float *) calloc((size_t) lx * ly, sizeof(float));
            lx >>= 1; ly >>= 1;
        }
    }
    double sum = 0.0;
    for (uint32_t y = 0; y < npy; ++y)
        for (uint32_t x = 0; x < npx; ++x) {
            const float *in = lum + y * W + x;
            const float avg = .25f * (in[0] + in[1] + in[W] + in[W + 1u]);
            sum += (double) avg;
            e->level[1][env_level_index(x, y, (uint32_t) e->level_w[1])] = avg;
        }
    const float norm = normalize ? (float) ((double) (npx * npy) / sum) : 1.f;
    for (uint32_t i = 0; i < W * H; ++i) e->level[0][i] = lum[i] * norm;
    for (int32_t i = 0; i < e->level_size[1]; ++i) e->level[1][i] *= norm;
    {
        uint32_t lx = npx, ly = npy;
        for (uint32_t level = 2; level <= max_level + 1u; ++level) {
            lx = (lx + 1u) >> 1; ly = (ly + 1u) >> 1;
            for (uint32_t y = 0; y < ly; ++y)
                for (uint32_t x = 0; x < lx; ++x) {
                    const float *d0 = e->level[level - 1u] + env_level_index(x * 2u, y * 2u, (uint32_t) e->level_w[level - 1u]);
                    e->level[level][env_level_index(x, y, (uint32_t) e->level_w[level])] = d0[0] + d0[1] + d0[2] + d0[3];
                }
        }
    }
}
/* known-answer entry: the warp alone over a caller's grid (src/core/tests/test_distr_2d.py builds Hierarchical2D0 from plain arrays) */
orc_envmap *orc_hier2d_create(const float *values, int32_t width, int32_t height, int32_t normalize) {
    if (width < 2 || height < 2) return NULL;
    orc_envmap *e = (orc_envmap *) calloc(1, sizeof *e);
    e->w = width; e->h = height; e->scale = 1.f; e->data = NULL;
    hier2d_build(e, values, (uint32_t) width, (uint32_t) height, normalize);
    return e;
}
/* mis_compensation (envmap.cpp:157-185, 216): the sampling density is built from max(luminance - average luminance, 0) ("MIS Compensation", Karlik et al. 2019)
 * unless average and minimum are within 1 % of each other; eval() keeps the unmodified radiance */
orc_envmap *orc_envmap_create2(const float *rgb, int32_t width, int32_t height, float scale, int32_t mis_compensation) {
    if (width < 2 || height < 3) return NULL;
    orc_envmap *e = (orc_envmap *) calloc(1, sizeof *e);
    const uint32_t W = (uint32_t) width + 1u, H = (uint32_t) height;
    e->w = (int32_t) W; e->h = (int32_t) H; e->scale = scale;
    e->data = (float *) malloc(sizeof(float) * 3u * W * H);
    float *lum = (float *) malloc(sizeof(float) * W * H);
    float luminance_offset = 0.f;
    if (mis_compensation) {
        float min_lum = 0.f; double accum = 0.0;
        for (uint32_t i = 0; i < (uint32_t) width * H; ++i) {
            const float *in = rgb + 3u * i;
            const float l = in[0] * 0.212671f + in[1] * 0.715160f + in[2] * 0.072169f;
            min_lum = f_min(min_lum, l);
            accum += (double) l;
        }
        luminance_offset = (float) (accum / (double) ((uint32_t) width * H));
        if (luminance_offset - min_lum <= 0.01f * luminance_offset) luminance_offset = 0.f;   /* (nearly) constant maps: disabled */
    }
    const float theta_scale = 1.f / (float) (H - 1u) * ORC_PI_F;
    for (uint32_t y = 0; y < H; ++y) {
        const float sin_theta = sinf((float) y * theta_scale);
        for (uint32_t x = 0; x < (uint32_t) width; ++x) {
            const float *in = rgb + 3u * (y * (uint32_t) width + x);
            float l = in[0] * 0.212671f + in[1] * 0.715160f + in[2] * 0.072169f;   /* mitsuba::luminance (spectrum.h:431-434) */
            l = f_max(l - luminance_offset, 0.f);
            lum[y * W + x] = l * sin_theta;
            memcpy(e->data + 3u * (y * W + x), in, 12);
        }
        lum[y * W + (W - 1u)] = lum[y * W];                                      /* the last column mirrors the first */
        memcpy(e->data + 3u * (y * W + (W - 1u)), e->data + 3u * (y * W), 12);
    }
    hier2d_build(e, lum, W, H, 1);
    free(lum);
    return e;
}
orc_envmap *orc_envmap_create(const float *rgb, int32_t width, int32_t height, float scale) { return orc_envmap_create2(rgb, width, height, scale, 0); }
void orc_envmap_free(orc_envmap *e) {
    if (!e) return;
    for (int32_t i = 0; i < e->n_levels; ++i) free(e->level[i]);
    free(e->data); free(e);
}
/* Hierarchical2D::sample (distr_2d.h:490-575) */
static void env_warp_sample(const orc_envmap *e, float sx, float sy, float *ux, float *uy, float *pdf) {
    sx = f_clamp01(sx); sy = f_clamp01(sy);
    uint32_t ox = 0, oy = 0;
    for (int32_t l = e->n_levels - 2; l > 0; --l) {
        ox <<= 1; oy <<= 1;
        const float *v = e->level[l] + env_level_index(ox, oy, (uint32_t) e->level_w[l]);
        const float v00 = v[0], v10 = v[1], v01 = v[2], v11 = v[3];
        sx = f_clamp01(sx); sy = f_clamp01(sy);
        const float r0 = v00 + v10, r1 = v01 + v11;
        sy *= r0 + r1;
        int mask = sy > r0;
        if (mask) { oy += 1u; sy -= r0; }
        sy /= mask ? r1 : r0;
        const float c0 = mask ? v01 : v00, c1 = mask ? v11 : v10;
        sx *= c0 + c1;
        mask = sx > c0;
        if (mask) sx -= c0;
        sx /= mask ? c1 : c0;
        if (mask) ox += 1u;
    }
    const uint32_t W = (uint32_t) e->level_w[0], i = ox + oy * W;
    const float *L = e->level[0];
    const float v00 = L[i], v10 = L[i + 1u], v01 = L[i + W], v11 = L[i + W + 1u];
    /* warp::square_to_bilinear (warp.h:388-402) */
    const float r0 = v00 + v10, r1 = v01 + v11;
    sy = interval_to_linear(r0, r1, sy);
    const float c0 = f_lerp(v00, v01, sy), c1 = f_lerp(v10, v11, sy);
    sx = interval_to_linear(c0, c1, sx);
    *pdf = f_lerp(c0, c1, sx);
    *ux = ((float) (int32_t) ox + sx) * e->patch_size[0]; *uy = ((float) (int32_t) oy + sy) * e->patch_size[1];
}
/* Hierarchical2D::eval (distr_2d.h:668-699) */
static float env_warp_eval(const orc_envmap *e, float x, float y) {
    x = f_clamp01(x) * e->inv_patch_size[0]; y = f_clamp01(y) * e->inv_patch_size[1];
    uint32_t ox = (uint32_t) (int32_t) x, oy = (uint32_t) (int32_t) y;
    if (ox > e->max_patch[0]) ox = e->max_patch[0];
    if (oy > e->max_patch[1]) oy = e->max_patch[1];
    x -= (float) (int32_t) ox; y -= (float) (int32_t) oy;
    const uint32_t W = (uint32_t) e->level_w[0], i = ox + oy * W;
    const float *L = e->level[0];
    return f_lerp(f_lerp(L[i], L[i + 1u], x), f_lerp(L[i + W], L[i + W + 1u], x), y);   /* square_to_bilinear_pdf */
}
/* eval_spectrum (envmap.cpp:487-553), rgb */
static v3 env_eval_uv(const orc_envmap *e, float u, float v) {
    const uint32_t rx = (uint32_t) e->w, ry = (uint32_t) e->h;
    u -= .5f / (float) (rx - 1u);
    u -= floorf(u); v -= floorf(v);
    u *= (float) (rx - 1u); v *= (float) (ry - 1u);
    uint32_t px = (uint32_t) u, py = (uint32_t) v;
    if (px > rx - 2u) px = rx - 2u;
    if (py > ry - 2u) py = ry - 2u;
    const float w1x = u - (float) px, w1y = v - (float) py, w0x = 1.f - w1x, w0y = 1.f - w1y;
    const float *d = e->data + 3u * (py * rx + px);
    float out[3];
    for (int c = 0; c < 3; ++c) {
        const float v00 = d[c], v10 = d[3 + c], v01 = d[3u * rx + c], v11 = d[3u * rx + 3 + c];
        const float a = fmaf(w0x, v00, w1x * v10), b = fmaf(w0x, v01, w1x * v11);
        out[c] = fmaf(w0y, a, w1y * b) * e->scale;
    }
    return V(out[0], out[1], out[2]);
}
#define ORC_INV_PI_F      0.31830988618379067154f
#define ORC_INV_TWO_PI_F  0.15915494309189533577f
#define ORC_EPSILON_F     5.9604644775390625e-8f   /* dr::Epsilon<float> = 2^-24 */
static inline float orc_safe_acos(float x) { return orc_acos(f_min(f_max(x, -1.f), 1.f)); }
static inline void env_dir_to_uv(v3 d, float *u, float *v) { *u = orc_atan2f(d.x, -d.z) * ORC_INV_TWO_PI_F; *v = orc_safe_acos(d.y) * ORC_INV_PI_F; }
static inline float env_inv_sin_theta(v3 d) { return f_rsqrt(f_max(f_max(f_sqr(d.x) + f_sqr(d.z), f_sqr(ORC_EPSILON_F)), 0.f)); }
/* EnvironmentMapEmitter::eval (envmap.cpp:299-310): d = -si.wi = the direction of the ray that left the scene */
static v3 env_eval(const orc_emitter *em, v3 d) {
    const v3 l = m_vector(em->to_local, d);
    float u, v; env_dir_to_uv(l, &u, &v);
    return env_eval_uv(em->envmap, u, v);
}
/* pdf_direction (:408-425) */
static float env_pdf_direction(const orc_emitter *em, v3 dw) {
    const v3 d = m_vector(em->to_local, dw);
    float u, v; env_dir_to_uv(d, &u, &v);
    u -= .5f / (float) ((uint32_t) em->envmap->w - 1u);
    u -= floorf(u); v -= floorf(v);
    return env_warp_eval(em->envmap, u, v) * env_inv_sin_theta(d) * (1.f / (2.f * f_sqr(ORC_PI_F)));
}
/* sample_direction (:363-406); *active = pdf > 0 */
static void env_sample_direction(const orc_emitter *em, v3 ref_p, float sx, float sy, v3 *d_out, float *dist, float *pdf_out, v3 *weight, int *active) {
    float u, v, pdf; env_warp_sample(em->envmap, sx, sy, &u, &v, &pdf);
    u += .5f / (float) ((uint32_t) em->envmap->w - 1u);
    *active = pdf > 0.f;
    const float theta = v * ORC_PI_F, phi = u * (2.f * ORC_PI_F);
    float st, ct, sp, cp; orc_sincos(theta, &st, &ct); orc_sincos(phi, &sp, &cp);
    v3 d = V(cp * st, sp * st, ct);          /* dr::sphdir */
    d = V(d.y, d.z, -d.x);
    const float radius = f_max(em->bsphere[3], v_norm(v_sub(ref_p, V(em->bsphere[0], em->bsphere[1], em->bsphere[2]))));
    *dist = 2.f * radius;
    const float ist = env_inv_sin_theta(d);
    *d_out = m_vector(em->env_to_world, d);
    *pdf_out = *active ? pdf * ist * (1.f / (2.f * f_sqr(ORC_PI_F))) : 0.f;
    const v3 rad = env_eval_uv(em->envmap, u, v);
    const float ip = f_rcp(*pdf_out);         /* Spectrum / Float: multiplication by the reciprocal */
    *weight = *active ? V(rad.x * ip, rad.y * ip, rad.z * ip) : V(0, 0, 0);
}
void orc_envmap_warp_sample(const orc_envmap *e, float sx, float sy, float *out) { env_warp_sample(e, sx, sy, &out[0], &out[1], &out[2]); }
float orc_envmap_warp_eval(const orc_envmap *e, float x, float y) { return env_warp_eval(e, x, y); }
void orc_envmap_sample_direction(const orc_emitter *em, const float *p, float sx, float sy, float *out) {
    v3 d, w; float dist, pdf; int active;
    env_sample_direction(em, V(p[0], p[1], p[2]), sx, sy, &d, &dist, &pdf, &w, &active);
    out[0] = d.x; out[1] = d.y; out[2] = d.z; out[3] = dist; out[4] = pdf; out[5] = w.x; out[6] = w.y; out[7] = w.z;
}
float orc_envmap_pdf_direction(const orc_emitter *em, const float *d) { return env_pdf_direction(em, V(d[0], d[1], d[2])); }
void orc_envmap_eval(const orc_emitter *em, const float *d, float *rgb) { v3 r = env_eval(em, V(d[0], d[1], d[2])); rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z; }
/* ------------------------------------------------------------ exp / log / tan / erf / erfinv
 * Dr.Jit's dr::exp, dr::log, dr::tan, dr::erf and dr::erfinv (drjit/math.h) are not in the tree.  They are restated from the
 * published single-precision kernels Dr.Jit's math library derives from: Cephes expf / logf / tanf (S. Moshier), the Cephes
 * erff series inside |x| < 1 with Abramowitz & Stegun 7.1.26 outside, and M. Giles' single-precision erfinv polynomial
 * ("Approximating the erfinv function", GPU Computing Gems 2).  Needed by the Beckmann microfacet distribution
 * (include/mitsuba/render/microfacet.h:176-196,240-290,341-403).  The product's dtof_math.h states the same operations. */
float orc_expf(float x) {
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.278929903431851103f) return 0.f;
    float z = floorf(fmaf(1.44269504088896341f, x, 0.5f));
    x = fmaf(z, -0.693359375f, x);
    x = fmaf(z, 2.12194440e-4f, x);
    int32_t n = (int32_t) z;
    float x2 = x * x;
    float p = fmaf(1.9875691500e-4f, x, 1.3981999507e-3f);
    p = fmaf(p, x, 8.3334519073e-3f);
    p = fmaf(p, x, 4.1665795894e-2f);
    p = fmaf(p, x, 1.6666665459e-1f);
    p = fmaf(p, x, 5.0000001201e-1f);
    float r = fmaf(p, x2, x) + 1.f;
    /* ldexpf(r, n), n in [-149, 128]: two exact power-of-two factors keep the intermediate normal */
    int32_t n1 = n / 2, n2 = n - n1;
    return r * u2f((uint32_t) (n1 + 127) << 23) * u2f((uint32_t) (n2 + 127) << 23);
}
float orc_logf(float x) {
    if (x < 0.f) return NAN;
    if (x == 0.f) return -INFINITY;
    if (!(x < INFINITY)) return x;
    uint32_t u = f2u(x); int32_t e = 0;
    if (u < 0x00800000u) { x *= 8388608.f; u = f2u(x); e = -23; }   /* subnormal */
    e += (int32_t) (u >> 23) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);                   /* frexp: m in [0.5, 1) */
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.f; } else m = m - 1.f;
    float z = m * m;
    float y = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    y = fmaf(y, m, 1.1676998740e-1f);
    y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f);
    y = fmaf(y, m, -1.6668057665e-1f);
    y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f);
    y = fmaf(y, m, 3.3333331174e-1f);
    y = y * m * z;
    float fe = (float) e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    return fmaf(0.693359375f, fe, m + y);
}
float orc_tanf(float xx) {
    float x = fabsf(xx);
    int32_t j = (int32_t) (x * 1.2732395447351626862f);
    j = (j + 1) & ~1;
    float y = (float) j;
    float z = x - y * 0.78515625f;
    z = z - y * 2.4187564849853515625e-4f;
    z = z - y * 3.77489497744594108e-8f;
    float zz = z * z;
    float p = fmaf(9.38540185543e-3f, zz, 3.11992232697e-3f);
    p = fmaf(p, zz, 2.44301354525e-2f);
    p = fmaf(p, zz, 5.34112807005e-2f);
    p = fmaf(p, zz, 1.33387994085e-1f);
    p = fmaf(p, zz, 3.33331568548e-1f);
    float r = x > 1.0e-4f ? fmaf(p * zz, z, z) : z;
    if (j & 2) r = -1.f / r;
    return u2f(f2u(r) ^ (f2u(xx) & 0x80000000u));
}
float orc_erff(float x) {
    float xa = fabsf(x);
    if (xa < 1.f) {
        float z = x * x;
        float p = fmaf(7.853861353153693e-5f, z, -8.010193625184903e-4f);
        p = fmaf(p, z, 5.188327685732524e-3f);
        p = fmaf(p, z, -2.685381193529856e-2f);
        p = fmaf(p, z, 1.128358514861418e-1f);
        p = fmaf(p, z, -3.761262582423300e-1f);
        p = fmaf(p, z, 1.128379165726710e+0f);
        return x * p;
    }
    float t = 1.f / fmaf(0.3275911f, xa, 1.f);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    float r = fmaf(-(p * t), orc_expf(-(xa * xa)), 1.f);
    return u2f(f2u(r) | (f2u(x) & 0x80000000u));
}
float orc_erfinvf(float x) {
    float w = -orc_logf((1.f - x) * (1.f + x)), p;
    if (w < 5.f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = fmaf(p, w, 3.43273939e-07f);
        p = fmaf(p, w, -3.5233877e-06f);
        p = fmaf(p, w, -4.39150654e-06f);
        p = fmaf(p, w, 0.00021858087f);
        p = fmaf(p, w, -0.00125372503f);
        p = fmaf(p, w, -0.00417768164f);
        p = fmaf(p, w, 0.246640727f);
        p = fmaf(p, w, 1.50140941f);
    } else {
        w = sqrtf(w) - 3.f;
        p = -0.000200214257f;
        p = fmaf(p, w, 0.000100950558f);
        p = fmaf(p, w, 0.00134934322f);
        p = fmaf(p, w, -0.00367342844f);
        p = fmaf(p, w, 0.00573950773f);
        p = fmaf(p, w, -0.0076224613f);
        p = fmaf(p, w, 0.00943887047f);
        p = fmaf(p, w, 1.00167406f);
        p = fmaf(p, w, 2.83297682f);
    }
    return p * x;
}
void orc_spot_params(float cutoff_deg, float beam_deg, float *out4) {
    float cutoff = cutoff_deg * (ORC_PI_F / 180.f), beam = beam_deg * (ORC_PI_F / 180.f);
    out4[0] = cutoff; out4[1] = orc_cos(cutoff); out4[2] = orc_cos(beam); out4[3] = 1.0f / (cutoff - beam);
}

/* ------------------------------------------------------------------ RNG */
/* sample_tea_32 -- include/mitsuba/core/random.h:33-47 */
void orc_tea32(uint32_t v0, uint32_t v1, int rounds, uint32_t *o0, uint32_t *o1) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    *o0 = v0; *o1 = v1;
}
/* sample_tea_float32 -- random.h:63-67 */
float orc_tea_float32(uint32_t v0, uint32_t v1, int rounds) {
    uint32_t a, b; orc_tea32(v0, v1, rounds, &a, &b);
    return u2f((b >> 9) | 0x3f800000u) - 1.f;
}
/* dr::PCG32 (Dr.Jit 0.4.0 drjit/random.h, absent; O'Neill's PCG-XSH-RR 64/32) */
#define PCG32_MULT 0x5851f42d4c957f2dULL
uint32_t orc_pcg32_next_u32(uint64_t *state, uint64_t inc) {
    uint64_t old = *state;
    *state = old * PCG32_MULT + inc;
    uint32_t xorshift = (uint32_t) (((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t) (old >> 59);
    return (xorshift >> rot) | (xorshift << ((~rot + 1u) & 31));
}
/* PCG32::seed(size=1, initstate, initseq): inc = (initseq<<1)|1, two warm-up steps */
/* (Dr.Jit 0.4.0 drjit/random.h, not in the tree; call sites src/render/sampler.cpp:130, src/samplers/correlated.cpp:57-62) */
void orc_pcg32_seed(uint64_t initstate, uint64_t initseq, uint64_t *state, uint64_t *inc) {
    *state = 0; *inc = (initseq << 1) | 1u;
    orc_pcg32_next_u32(state, *inc);
    *state += initstate;
    orc_pcg32_next_u32(state, *inc);
}
/* PCG32::next_float32: bitcast((next_u32 >> 9) | 0x3f800000) - 1 (drjit/random.h; used through include/mitsuba/core/random.h:28) */
float orc_pcg32_next_f32(uint64_t *state, uint64_t inc) {
    return u2f((orc_pcg32_next_u32(state, inc) >> 9) | 0x3f800000u) - 1.f;
}
/* permute_kensler -- random.h:113-171 (cycle-walking form of the JIT loop :151-158) */
uint32_t orc_permute_kensler(uint32_t index, uint32_t n, uint32_t seed) {
    if (n <= 1) return 0;   /* n == 0 never leaves the cycle-walking loop (the reference hangs there too): callers reject it */
    uint32_t w = n - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        uint32_t tmp = index;
        tmp ^= seed;            tmp *= 0xe170893du;
        tmp ^= seed >> 16;      tmp ^= (tmp & w) >> 4;
        tmp ^= seed >> 8;       tmp *= 0x0929eb3fu;
        tmp ^= seed >> 23;      tmp ^= (tmp & w) >> 1;
        tmp *= 1 | seed >> 27;  tmp *= 0x6935fa69u;
        tmp ^= (tmp & w) >> 11; tmp *= 0x74dcb303u;
        tmp ^= (tmp & w) >> 2;  tmp *= 0x9e501cc3u;
        tmp ^= (tmp & w) >> 2;  tmp *= 0xc860a3dfu;
        tmp &= w;               tmp ^= tmp >> 5;
        index = tmp;
    } while (index >= n);
    return (index + seed) % n;
}

/* ------------------------------------------------------------------ sampler */
typedef struct {
    uint64_t s_main, i_main, s_time, i_time, s_path, i_path;
    uint32_t perm_seed, dim, sample_index;
} orc_sampler;

/* CorrelatedSampler::seed -- src/samplers/correlated.cpp:38-64, PCG32Sampler::seed
 * src/render/sampler.cpp:115-134, compute_per_sequence_seed :85-92 */
static void sampler_seed(orc_sampler *s, const orc_params *p, uint32_t seed, uint32_t spp, uint32_t lane) {
    uint32_t sv = p->base_seed + seed, v0, v1;
    orc_tea32(sv, lane, 4, &v0, &v1);
    orc_pcg32_seed(v0, v1, &s->s_main, &s->i_main);
    orc_tea32(sv + 1, lane / (uint32_t) p->time_correlate_number, 4, &v0, &v1);
    orc_pcg32_seed(v0, v1, &s->s_time, &s->i_time);
    orc_tea32(sv + 2, lane / (uint32_t) p->path_correlate_number, 4, &v0, &v1);
    orc_pcg32_seed(v0, v1, &s->s_path, &s->i_path);
    orc_tea32(p->base_seed, spp * (lane / spp) + seed, 4, &v0, &v1);
    s->perm_seed = v0;
    s->dim = 0;
    s->sample_index = spp > 1 ? lane % spp : 0;  /* sampler.cpp:94-103, m_sample_index = 0 */
}
/* next_1d_correlate -- correlated.cpp:156-161: both streams always advance */
static inline float sampler_next_1d_correlate(orc_sampler *s, int correlate) {
    float r1 = orc_pcg32_next_f32(&s->s_path, s->i_path);
    float r2 = orc_pcg32_next_f32(&s->s_main, s->i_main);
    return correlate ? r1 : r2;
}
/* Sampler::next_1d -- correlated.cpp:79-84: the independent (main) stream only; what the `path` / `velocity`
 * integrators and the non-Doppler branch of render_sample draw (integrator.cpp:416-431, path.cpp:197,213-214,273) */
static inline float sampler_next_1d(orc_sampler *s) { return orc_pcg32_next_f32(&s->s_main, s->i_main); }
static inline float sampler_draw(orc_sampler *s, int correlate, int plain) {
    return plain ? sampler_next_1d(s) : sampler_next_1d_correlate(s, correlate);
}
/* next_1d_time -- correlated.cpp:92-153 */
static float sampler_next_1d_time(orc_sampler *s, const orc_params *p, uint32_t spp) {
    int strategy = p->time_sampling;
    uint32_t tcn = (uint32_t) p->time_correlate_number;
    if (strategy == ORC_TIME_UNIFORM)
        return orc_pcg32_next_f32(&s->s_main, s->i_main);
    uint32_t si = s->sample_index;
    float r = (strategy == ORC_TIME_STRATIFIED) ? orc_pcg32_next_f32(&s->s_main, s->i_main)
                                                : orc_pcg32_next_f32(&s->s_time, s->i_time);
    if (p->stratify_each_interval) {
        int n_stratum = (int) (spp / tcn);
        float inv_n = 1.0f / (float) n_stratum;     /* array / scalar = array * rcp(scalar) */
        if (strategy == ORC_TIME_STRATIFIED) {
            uint32_t ps = s->perm_seed + s->dim++;
            uint32_t p1 = orc_permute_kensler(si / tcn, (uint32_t) n_stratum, ps);
            ps = s->perm_seed + s->dim++;
            uint32_t p2 = orc_permute_kensler(si / tcn, (uint32_t) n_stratum, ps);
            uint32_t pp = (si % tcn != 0) ? p1 : p2;
            r = ((float) pp + r) * inv_n;
        } else {
            uint32_t pp = si / tcn;
            r = ((float) pp + r) * inv_n;
        }
    }
    if (strategy == ORC_TIME_STRATIFIED) {
        uint32_t pp = si % tcn;
        return ((float) pp + r) * (1.0f / (float) tcn);
    } else if (strategy == ORC_TIME_ANTITHETIC) {
        uint32_t rem = si % tcn;
        if (tcn == 2) { float r2 = r + p->antithetic_shift; return rem != 1 ? r : r2; }
        return r + (float) rem / (float) tcn;
    } else if (strategy == ORC_TIME_ANTITHETIC_MIRROR) {
        float r2 = 1.0f - r + p->antithetic_shift;
        uint32_t rem = si % tcn;
        return rem != 1 ? r : r2;
    } else if (strategy == ORC_TIME_PERIODIC) {          /* correlated.cpp:147-150 */
        uint32_t rem = si % tcn;
        return r + (float) rem / (float) tcn;
    }
    return r;                                            /* TIME_SAMPLING_REGULAR: no branch taken (:152) */
}

/* ------------------------------------------------------------------ waveforms */
/* eval_modulation_function_value -- include/mitsuba/render/waveform_utils.h:24-33 */
float orc_waveform(float _t, int type) {
    float t = fmodf(_t, 2.f * ORC_PI_F);
    switch (type) {
        case ORC_WAVE_RECT: return fabsf(t - ORC_PI_F) > 0.5f * ORC_PI_F ? 1.f : -1.f;
        case ORC_WAVE_TRI:  return t < ORC_PI_F ? 1.f - 2.f * t * (1.0f / ORC_PI_F)
                                                : -3.f + 2.f * t * (1.0f / ORC_PI_F);
        default: return orc_cos(t);   /* sinusoidal; trapezoidal falls through (:27-32) */
    }
}
/* eval_modulation_function_value_low_pass -- waveform_utils.h:36-62 */
float orc_waveform_low_pass(float _t, int type) {
    float t = fmodf(_t, 2.f * ORC_PI_F);
    if (type == ORC_WAVE_SIN) return orc_cos(t);
    float a = t * (1.0f / ORC_PI_F), b = 2.f - a, c = a < b ? a : b;
    switch (type) {
        case ORC_WAVE_RECT: return 2.f - 4.f * c;
        case ORC_WAVE_TRI:  return (4.f * c * c * c - 6.f * c * c + 1.f) * 2.0f * (1.0f / 3.0f);
        case ORC_WAVE_TRAP: { float r = 2.f - 4.f * c; return f_min(f_max(2.0f * r, -2.0f), 2.0f); }
    }
    return orc_cos(t);
}
/* eval_modulation_weight -- src/integrators/dopplertofpath.cpp:60-77.  The scalar
 * prefactors are folded in double and rounded once to float (they multiply a JIT
 * Float, so Dr.Jit converts the double scalar to float32 first). */
float orc_modulation_weight(const orc_params *p, float ray_time, float path_length) {
    float w_g = (float) (2 * M_PI * (double) p->w_g_mhz * 1e6);
    float w_d = (float) (2 * M_PI / (double) p->time * (double) p->hetero_frequency);
    float phi = (float) ((2 * M_PI * (double) p->w_g_mhz) / 300) * path_length;
    if (p->low_frequency_component_only) {
        float t = w_d * ray_time + p->phase_offset + phi;
        return (float) (0.5 * (double) p->g_1) * orc_waveform_low_pass(t, p->wave_type);
    }
    float t1 = w_g * ray_time - phi;
    float t2 = (w_g + w_d) * ray_time + p->phase_offset;
    float g_t = p->g_1 * orc_waveform(t1, p->wave_type) + p->g_0;
    float s_t = orc_waveform(t2, p->wave_type);
    return s_t * g_t;
}

/* ------------------------------------------------------------------ camera */
/* 4x4 helpers for the camera set-up (column-major Dr.Jit product: fmadd chain over k) */
static void m4_mul(const float *a, const float *b, float *out) {
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = a[4 * i + 0] * b[0 + j];
            for (int k = 1; k < 4; ++k) s = fmaf(a[4 * i + k], b[4 * k + j], s);
            r[4 * i + j] = s;
        }
    memcpy(out, r, sizeof r);
}
static void m4_identity(float *m) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.f; }

/* sample_to_camera = inverse of perspective_projection(...) (include/mitsuba/render/sensor.h:226-262,
 * Transform::perspective transform.h:215-233, PerspectiveCamera::update_camera_transforms
 * src/sensors/perspective.cpp:172-198).  Transform keeps analytic inverses, so the
 * inverse is the reversed product of the factors' inverses, in float32. */
static void camera_sample_to_camera(const orc_sensor *s, float *inv_out) {
    float fw = (float) s->film_w, fh = (float) s->film_h;
    float rel_sx = (float) s->crop_w / fw, rel_sy = (float) s->crop_h / fh;
    float rel_ox = (float) s->crop_x / fw, rel_oy = (float) s->crop_y / fh;
    float aspect = fw / fh;
    float near_ = s->near_clip, far_ = s->far_clip;
    /* tan is evaluated in double and rounded (Dr.Jit's own float32 tan is absent) */
    float tanv = (float) tan((double) (s->x_fov * .5f) * (M_PI / 180.0));
    float S1i[16], T1i[16], S2i[16], T2i[16], Pi[16], tmp[16];
    m4_identity(S1i); S1i[0] = f_rcp(1.f / rel_sx); S1i[5] = f_rcp(1.f / rel_sy);
    m4_identity(T1i); T1i[3] = rel_ox; T1i[7] = rel_oy;          /* inverse of translate(-rel_offset) */
    m4_identity(S2i); S2i[0] = f_rcp(-0.5f); S2i[5] = f_rcp(-0.5f * aspect);
    m4_identity(T2i); T2i[3] = 1.f; T2i[7] = 1.f / aspect;        /* inverse of translate(-1,-1/aspect,0) */
    memset(Pi, 0, 64);                                            /* transform.h:226-230 inv_trafo */
    Pi[0] = tanv; Pi[5] = tanv; Pi[10] = 0.f; Pi[15] = f_rcp(near_);
    Pi[11] = 1.f; Pi[14] = (near_ - far_) / (far_ * near_);
    if (s->kind == ORC_SENSOR_ORTHOGRAPHIC) {
        /* orthographic_projection (sensor.h:266-299) ends in Transform::orthographic = scale(1, 1, 1 / (far - near)) * translate(0, 0, -near)
         * (transform.h:242-245); its inverse is translate(0, 0, near) * scale(rcp of the factors) */
        float OT[16], OS[16];
        m4_identity(OT); OT[11] = near_;
        m4_identity(OS); OS[0] = f_rcp(1.f); OS[5] = f_rcp(1.f); OS[10] = f_rcp(1.f / (far_ - near_));
        m4_mul(OT, OS, Pi);
    }
    /* (S1*T1*S2*T2*P)^-1 = P^-1*(T2^-1*(S2^-1*(T1^-1*S1^-1))) */
    m4_mul(T1i, S1i, tmp); m4_mul(S2i, tmp, tmp); m4_mul(T2i, tmp, tmp); m4_mul(Pi, tmp, inv_out);
}

/* PerspectiveCamera::sample_ray_differential -- src/sensors/perspective.cpp:238-279, and
 * ThinLensCamera::sample_ray_differential_impl -- src/sensors/thinlens.cpp:257-305 (a_x, a_y: the aperture sample)
 * (differentials are not needed by the BSDFs of this path and are not produced). */
typedef struct { v3 o, d; float maxt; } orc_ray;
static void concentric_disk(float sx, float sy, float *px, float *py);
static orc_ray camera_ray(const orc_sensor *s, const float *s2c, float ux, float uy, float a_x, float a_y) {
    /* Transform::operator*(Point): homogeneous, then head<3>(r) / r.w = head * rcp(w) */
    float r0 = fmaf(s2c[2], 0.f, fmaf(s2c[1], uy, fmaf(s2c[0], ux, s2c[3])));
    float r1 = fmaf(s2c[6], 0.f, fmaf(s2c[5], uy, fmaf(s2c[4], ux, s2c[7])));
    float r2 = fmaf(s2c[10], 0.f, fmaf(s2c[9], uy, fmaf(s2c[8], ux, s2c[11])));
    float r3 = fmaf(s2c[14], 0.f, fmaf(s2c[13], uy, fmaf(s2c[12], ux, s2c[15])));
    float iw = f_rcp(r3);
    v3 near_p = V(r0 * iw, r1 * iw, r2 * iw);
    orc_ray ray; v3 d;
    if (s->kind == ORC_SENSOR_ORTHOGRAPHIC) {   /* OrthographicCamera::sample_ray_differential (src/sensors/orthographic.cpp:169-196) */
        ray.o = m_point(s->to_world, near_p);
        ray.d = v_normalize(m_vector(s->to_world, V(0.f, 0.f, 1.f)));
        ray.maxt = s->far_clip - s->near_clip;
        return ray;
    }
    if (s->kind == ORC_SENSOR_THINLENS) {
        float tx, ty; concentric_disk(a_x, a_y, &tx, &ty);
        v3 aperture_p = V(s->aperture_radius * tx, s->aperture_radius * ty, 0.f);
        float f_dist = s->focus_distance / near_p.z;
        v3 focus_p = v_mul(near_p, f_dist);
        d = v_normalize(v_sub(focus_p, aperture_p));
        ray.o = m_point(s->to_world, aperture_p);
    } else {
        d = v_normalize(near_p);
        ray.o = V(s->to_world[3], s->to_world[7], s->to_world[11]);
    }
    ray.d = m_vector(s->to_world, d);
    float inv_z = f_rcp(d.z);
    float near_t = s->near_clip * inv_z, far_t = s->far_clip * inv_z;
    ray.o = v_add(ray.o, v_mul(ray.d, near_t));
    ray.maxt = far_t - near_t;
    return ray;
}
/* test entry: PerspectiveCamera::sample_ray_differential (src/sensors/perspective.cpp:238-279) at time 0 for a film position */
void orc_camera_ray(const orc_sensor *s, float px, float py, float *out) {
    float s2c[16]; camera_sample_to_camera(s, s2c);
    float sx = 1.f / (float) s->crop_w, sy = 1.f / (float) s->crop_h;
    float ux = fmaf(px, sx, -(float) s->crop_x * sx), uy = fmaf(py, sy, -(float) s->crop_y * sy);
    orc_ray r = camera_ray(s, s2c, ux, uy, .5f, .5f);
    out[0] = r.o.x; out[1] = r.o.y; out[2] = r.o.z; out[3] = r.d.x; out[4] = r.d.y; out[5] = r.d.z;
    out[6] = r.maxt;
}

/* test entry: Sensor::sample_ray(time, wavelength_sample, position_sample, aperture_sample) of either camera: o(3), d(3), maxt */
void orc_camera_sample_ray(const orc_sensor *s, float ux, float uy, float a_x, float a_y, float *out) {
    float s2c[16]; camera_sample_to_camera(s, s2c);
    orc_ray r = camera_ray(s, s2c, ux, uy, a_x, a_y);
    out[0] = r.o.x; out[1] = r.o.y; out[2] = r.o.z; out[3] = r.d.x; out[4] = r.d.y; out[5] = r.d.z;
    out[6] = r.maxt;
}

/* ------------------------------------------------------------------ geometry */
typedef struct { float t, u, v; int32_t obj, shape, prim; } orc_hit;

/* AnimatedTransform::eval -- include/mitsuba/core/transform.h:439-466: component-wise
 * lerp of the two keyframe matrices, t clamped to [0,1]. */
static void instance_to_world(const orc_object *o, float time, float *m) {
    if (o->n_keys <= 1) { memcpy(m, o->key[0], 64); return; }
    float t0 = o->key_time[0], t1 = o->key_time[1];
    float t = f_min(f_max((time - t0) / (t1 - t0), 0.f), 1.f);
    float omt = 1 - t;
    for (int i = 0; i < 16; ++i) m[i] = o->key[0][i] * omt + o->key[1][i] * t;
}

/* Rectangle::ray_intersect_preliminary_impl -- src/shapes/rectangle.cpp:201-224 */
static int rect_intersect(const orc_shape *sh, v3 o, v3 d, float maxt, float *t_out, float *u, float *v) {
    v3 lo = m_point(sh->to_object, o), ld = m_vector(sh->to_object, d);
    float t = -lo.z / ld.z;
    float lx = fmaf(ld.x, t, lo.x), ly = fmaf(ld.y, t, lo.y);
    if (t >= 0.f && t <= maxt && fabsf(lx) <= 1.f && fabsf(ly) <= 1.f) {
        *t_out = t; *u = lx; *v = ly; return 1;
    }
    return 0;
}
/* Disk::ray_intersect_preliminary_impl -- src/shapes/disk.cpp:216-232 */
static int disk_intersect(const orc_shape *sh, v3 o, v3 d, float maxt, float *t_out, float *u, float *v) {
    v3 lo = m_point(sh->to_object, o), ld = m_vector(sh->to_object, d);
    float t = -lo.z / ld.z;
    float lx = fmaf(ld.x, t, lo.x), ly = fmaf(ld.y, t, lo.y);
    if (t >= 0.f && t <= maxt && lx * lx + ly * ly <= 1.f) {
        *t_out = t; *u = lx; *v = ly; return 1;
    }
    return 0;
}
/* Triangle test: Embree 3's Moeller-Trumbore intersector (source absent; published
 * algorithm, kernels/geometry/triangle_intersector_moeller.h): tnear < t <= tfar,
 * u/v are the barycentrics of vertices 1 and 2. */
static int tri_intersect(v3 p0, v3 p1, v3 p2, v3 o, v3 d, float maxt, float *t_out, float *u, float *v) {
    v3 e1 = v_sub(p0, p1), e2 = v_sub(p2, p0), ng = v_cross(e2, e1);
    v3 c = v_sub(p0, o), r = v_cross(c, d);
    float den = v_dot(ng, d), aden = fabsf(den);
    uint32_t sgn = f2u(den) & 0x80000000u;
    float U = u2f(f2u(v_dot(r, e2)) ^ sgn), Vv = u2f(f2u(v_dot(r, e1)) ^ sgn);
    if (!(den != 0.f && U >= 0.f && Vv >= 0.f && U + Vv <= aden)) return 0;
    float T = u2f(f2u(v_dot(ng, c)) ^ sgn);
    if (!(0.f < T && T <= aden * maxt)) return 0;
    float rc = 1.0f / aden;
    *u = U * rc; *v = Vv * rc; *t_out = T * rc;
    return 1;
}
/* math::solve_quadratic (include/mitsuba/core/math.h:357-401) in double */
static int solve_quadratic_d(double a, double b, double c, double *x0, double *x1) {
    int linear = a == 0.0, valid_linear = linear && b != 0.0;
    *x0 = *x1 = -c / b;
    double discrim = fma(b, b, -(4.0 * a * c));
    int valid_quadratic = !linear && discrim >= 0.0;
    if (valid_quadratic) {
        double sq = sqrt(discrim);
        double temp = -0.5 * (b + copysign(sq, b));
        double x0p = temp / a, x1p = c / temp;
        *x0 = x0p < x1p ? x0p : x1p; *x1 = x0p < x1p ? x1p : x0p;
    }
    return valid_linear || valid_quadratic;
}
static inline double dot3d(const double *a, const double *b) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }
/* Sphere::ray_intersect_preliminary_impl (sphere.cpp:338-394): float64 on the llvm back end; the point on the ray closest
 * to the centre is evaluated with the FLOAT ray (Ray::operator() takes a Float, ray.h:61) */
static int sphere_intersect(const orc_shape *sh, v3 o, v3 d, float maxt_f, float *t_out) {
    const double radius = sh->radius, ctr[3] = { sh->center[0], sh->center[1], sh->center[2] }, maxt = maxt_f;
    const double dd[3] = { d.x, d.y, d.z }, l[3] = { (double) o.x - ctr[0], (double) o.y - ctr[1], (double) o.z - ctr[2] };
    const double nl[3] = { -l[0], -l[1], -l[2] };
    double plane_t = dot3d(nl, dd) / sqrt(dot3d(dd, dd));
    int no_hit = plane_t == 0.0 && (o.x != sh->center[0] && o.y != sh->center[1] && o.z != sh->center[2]);
    v3 pp = v_fma(d, (float) plane_t, o);
    const double oo[3] = { (double) pp.x - ctr[0], (double) pp.y - ctr[1], (double) pp.z - ctr[2] };
    no_hit = no_hit && sqrt(dot3d(oo, oo)) > radius;
    double A = dot3d(dd, dd), B = 2.0 * dot3d(oo, dd), C = dot3d(oo, oo) - radius * radius, near_t, far_t;
    int found = solve_quadratic_d(A, B, C, &near_t, &far_t);
    near_t += plane_t; far_t += plane_t;
    int out_bounds = !(near_t <= maxt && far_t >= 0.0), in_bounds = near_t < 0.0 && far_t > maxt;
    if (!(found && !no_hit && !out_bounds && !in_bounds)) return 0;
    *t_out = near_t < 0.0 ? (float) far_t : (float) near_t;
    return 1;
}
/* Sphere::ray_test_impl (sphere.cpp:396-431) */
static int sphere_test(const orc_shape *sh, v3 o, v3 d, float maxt_f) {
    const double radius = sh->radius, maxt = maxt_f, dd[3] = { d.x, d.y, d.z };
    const double oo[3] = { (double) o.x - (double) sh->center[0], (double) o.y - (double) sh->center[1], (double) o.z - (double) sh->center[2] };
    double A = dot3d(dd, dd), B = 2.0 * dot3d(oo, dd), C = dot3d(oo, oo) - radius * radius, near_t, far_t;
    int found = solve_quadratic_d(A, B, C, &near_t, &far_t);
    int out_bounds = !(near_t <= maxt && far_t >= 0.0), in_bounds = near_t < 0.0 && far_t > maxt;
    return found && !out_bounds && !in_bounds;
}
/* Cylinder::ray_intersect_preliminary_impl / ray_test_impl (src/shapes/cylinder.cpp:300-391): the unit cylinder in object space, float64
 * on the llvm back end (the ray is transformed in float32, then widened) */
static int cylinder_query(const orc_shape *sh, v3 o, v3 d, float maxt_f, float *t_out) {
    v3 lo = m_point(sh->to_object, o), ld = m_vector(sh->to_object, d);
    const double ox = lo.x, oy = lo.y, oz = lo.z, dx = ld.x, dy = ld.y, dz = ld.z, maxt = maxt_f;
    double A = dx * dx + dy * dy, B = 2.0 * (dx * ox + dy * oy), C = ox * ox + oy * oy - 1.0, near_t, far_t;
    int found = solve_quadratic_d(A, B, C, &near_t, &far_t);
    int out_bounds = !(near_t <= maxt && far_t >= 0.0), in_bounds = near_t < 0.0 && far_t > maxt;
    double z_near = oz + dz * near_t, z_far = oz + dz * far_t;
    int near_ok = z_near >= 0.0 && z_near <= 1.0 && near_t >= 0.0, far_ok = z_far >= 0.0 && z_far <= 1.0 && far_t <= maxt;
    if (!(found && !out_bounds && !in_bounds && (near_ok || far_ok))) return 0;
    *t_out = near_ok ? (float) near_t : (float) far_t;
    return 1;
}
static inline v3 mesh_pos(const orc_shape *sh, uint32_t i) { return V(sh->positions[3 * i], sh->positions[3 * i + 1], sh->positions[3 * i + 2]); }

/* closest hit in one shape.  Candidates are all primitives hit with t <= the ray's maxt; the
 * winner is the smallest t, exact ties going to the lowest (object, shape, prim) index (objects
 * are visited in index order and the comparison is strict).  A hit at exactly t == maxt counts
 * as a miss (hit = t != maxt, scene_embree.inl:313).  Traversal-order independent, so the GPU's
 * BVH walk reproduces it exactly. */
static void shape_closest(const orc_shape *sh, v3 o, v3 d, float maxt, int32_t obj, int32_t shape_idx, orc_hit *best) {
    float t, u, v;
    if (sh->kind == ORC_SHAPE_RECT) {
        if (rect_intersect(sh, o, d, maxt, &t, &u, &v) && t < best->t) {
            best->t = t; best->u = u; best->v = v; best->obj = obj; best->shape = shape_idx; best->prim = 0;
        }
    } else if (sh->kind == ORC_SHAPE_DISK) {
        if (disk_intersect(sh, o, d, maxt, &t, &u, &v) && t < best->t) {
            best->t = t; best->u = u; best->v = v; best->obj = obj; best->shape = shape_idx; best->prim = 0;
        }
    } else if (sh->kind == ORC_SHAPE_SPHERE) {
        if (sphere_intersect(sh, o, d, maxt, &t) && t < best->t) {
            best->t = t; best->u = 0.f; best->v = 0.f; best->obj = obj; best->shape = shape_idx; best->prim = 0;
        }
    } else if (sh->kind == ORC_SHAPE_CYLINDER) {
        if (cylinder_query(sh, o, d, maxt, &t) && t < best->t) {
            best->t = t; best->u = 0.f; best->v = 0.f; best->obj = obj; best->shape = shape_idx; best->prim = 0;
        }
    } else {
        for (int32_t f = 0; f < sh->n_faces; ++f) {
            const uint32_t *fi = sh->faces + 3 * f;
            if (tri_intersect(mesh_pos(sh, fi[0]), mesh_pos(sh, fi[1]), mesh_pos(sh, fi[2]), o, d, maxt, &t, &u, &v)
                && t < best->t) {
                best->t = t; best->u = u; best->v = v; best->obj = obj; best->shape = shape_idx; best->prim = f;
            }
        }
    }
}
static int shape_any(const orc_shape *sh, v3 o, v3 d, float maxt) {
    float t, u, v;
    if (sh->kind == ORC_SHAPE_RECT) return rect_intersect(sh, o, d, maxt, &t, &u, &v);
    if (sh->kind == ORC_SHAPE_DISK) return disk_intersect(sh, o, d, maxt, &t, &u, &v);
    if (sh->kind == ORC_SHAPE_SPHERE) return sphere_test(sh, o, d, maxt);
    if (sh->kind == ORC_SHAPE_CYLINDER) return cylinder_query(sh, o, d, maxt, &t);
    for (int32_t f = 0; f < sh->n_faces; ++f) {
        const uint32_t *fi = sh->faces + 3 * f;
        if (tri_intersect(mesh_pos(sh, fi[0]), mesh_pos(sh, fi[1]), mesh_pos(sh, fi[2]), o, d, maxt, &t, &u, &v)) return 1;
    }
    return 0;
}

/* Scene::ray_intersect_preliminary (src/render/scene_embree.inl:202-333): closest hit over
 * all top-level objects; instances intersect their group in object space with the ray
 * transformed by inverse(lerp(M0,M1,time)) (src/shapes/instance.cpp:295-311 + Embree). */
static orc_hit scene_closest(const orc_scene *sc, v3 o, v3 d, float time, float maxt) {
    orc_hit best; best.t = maxt; best.u = best.v = 0.f; best.obj = best.shape = best.prim = -1;
    for (int32_t i = 0; i < sc->n_objects; ++i) {
        const orc_object *ob = &sc->objects[i];
        if (ob->kind == ORC_OBJ_SHAPE) {
            shape_closest(&sc->shapes[ob->index], o, d, maxt, i, 0, &best);
        } else {
            float m[16], inv[16];
            instance_to_world(ob, time, m);
            m_affine_inverse(m, inv);
            v3 lo = m_point(inv, o), ld = m_vector(inv, d);
            const orc_group *g = &sc->groups[ob->index];
            for (int32_t k = 0; k < g->n_shapes; ++k)
                shape_closest(&sc->shapes[g->first_shape + k], lo, ld, maxt, i, k, &best);
        }
    }
    if (best.obj < 0) best.t = INFINITY;   /* hit = (t != maxt), scene_embree.inl:313-315 */
    return best;
}
static int scene_occluded(const orc_scene *sc, v3 o, v3 d, float time, float maxt) {
    for (int32_t i = 0; i < sc->n_objects; ++i) {
        const orc_object *ob = &sc->objects[i];
        if (ob->kind == ORC_OBJ_SHAPE) {
            if (shape_any(&sc->shapes[ob->index], o, d, maxt)) return 1;
        } else {
            float m[16], inv[16];
            instance_to_world(ob, time, m);
            m_affine_inverse(m, inv);
            v3 lo = m_point(inv, o), ld = m_vector(inv, d);
            const orc_group *g = &sc->groups[ob->index];
            for (int32_t k = 0; k < g->n_shapes; ++k)
                if (shape_any(&sc->shapes[g->first_shape + k], lo, ld, maxt)) return 1;
        }
    }
    return 0;
}
int orc_intersect(const orc_scene *sc, const float *o, const float *d, float time, float maxt, float *hit, int32_t *ids) {
    orc_hit h = scene_closest(sc, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), time, maxt);
    hit[0] = h.t; hit[1] = h.u; hit[2] = h.v; ids[0] = h.obj; ids[1] = h.shape; ids[2] = h.prim;
    return h.obj >= 0;
}
int orc_occluded(const orc_scene *sc, const float *o, const float *d, float time, float maxt) {
    return scene_occluded(sc, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), time, maxt);
}

/* coordinate_system -- include/mitsuba/core/vector.h:116-136 (Duff et al.) */
static void coordinate_system(v3 n, v3 *s, v3 *t) {
    float sign = f_sign(n.z), a = -f_rcp(sign + n.z), b = n.x * n.y * a;
    *s = V(f_mulsign(f_sqr(n.x) * a, n.z) + 1.f, f_mulsign(b, n.z), f_mulsign_neg(n.x, n.z));
    *t = V(b, fmaf(n.y, n.y * a, sign), -n.y);
}

typedef struct {
    v3 p, n, sh_n, sh_s, sh_t, dp_du, dp_dv, wi;
    float uv_u, uv_v;        /* si.uv (rectangles and meshes; what the textures are looked up with) */
    const orc_shape *shape;
} orc_si;

/* Rectangle::compute_surface_interaction -- src/shapes/rectangle.cpp:250-323 (non-diff
 * branch :289-294) with the frame of Rectangle::update :101-113 */
static void rect_si(const orc_shape *sh, v3 o, v3 d, float t, orc_si *si) {
    v3 dp_du = m_vector(sh->to_world, V(2.f, 0.f, 0.f));
    v3 dp_dv = m_vector(sh->to_world, V(0.f, 2.f, 0.f));
    v3 n = v_normalize(m_normal(sh->to_object, V(0.f, 0.f, 1.f)));
    v3 p = v_fma(d, t, o);
    v3 tr = V(sh->to_world[3], sh->to_world[7], sh->to_world[11]);
    float dist = v_dot(v_sub(tr, p), n);
    si->p = v_add(p, v_mul(n, dist));
    si->n = n; si->sh_n = n; si->dp_du = dp_du; si->dp_dv = dp_dv;
}
/* Disk::compute_surface_interaction -- src/shapes/disk.cpp:276-345 (primal branch :305-310, frame :316-336); (u, v) = local hit position */
static void disk_si(const orc_shape *sh, v3 o, v3 d, float t, float u, float v, orc_si *si) {
    v3 n = v_normalize(m_normal(sh->to_object, V(0.f, 0.f, 1.f)));
    v3 p = v_fma(d, t, o);
    v3 tr = V(sh->to_world[3], sh->to_world[7], sh->to_world[11]);
    float dist = v_dot(v_sub(tr, p), n);
    si->p = v_add(p, v_mul(n, dist));
    float r = sqrtf(fmaf(v, v, u * u)), inv_r = f_rcp(r);
    float cos_phi = r != 0.f ? u * inv_r : 1.f, sin_phi = r != 0.f ? v * inv_r : 0.f;
    si->n = n; si->sh_n = n;
    si->dp_du = m_vector(sh->to_world, V(cos_phi, sin_phi, 0.f));
    si->dp_dv = m_vector(sh->to_world, V(-sin_phi, cos_phi, 0.f));
}
/* Sphere::compute_surface_interaction -- src/shapes/sphere.cpp:435-560 (primal branch :509-513, dp_du :527-545) */
static void sphere_si(const orc_shape *sh, v3 o, v3 d, float t, orc_si *si) {
    v3 c = V(sh->center[0], sh->center[1], sh->center[2]);
    v3 n = v_normalize(v_sub(v_fma(d, t, o), c));
    si->p = v_fma(n, sh->radius, c);
    v3 local = m_point(sh->to_object, si->p);
    float rd = sqrtf(f_sqr(local.x) + f_sqr(local.y)), inv_rd = f_rcp(rd);
    v3 dpv = V(local.z * (local.x * inv_rd), local.z * (local.y * inv_rd), -rd);
    if (rd == 0.f) dpv = V(1.f, 0.f, 0.f);
    si->dp_du = v_mul(m_vector(sh->to_world, V(-local.y, local.x, 0.f)), 2.f * ORC_PI_F);
    si->dp_dv = v_mul(m_vector(sh->to_world, dpv), ORC_PI_F);
    if (sh->flip_normals) n = v_neg(n);
    si->sh_n = n; si->n = n;
}
/* Mesh::compute_surface_interaction -- src/render/mesh.cpp:632-864 (primal branch) */
/* Cylinder::compute_surface_interaction (cylinder.cpp:395-500, non-diff branch): the frame from the local hit point, the point shifted onto the surface */
static void cylinder_si(const orc_shape *sh, v3 o, v3 d, float t, orc_si *si) {
    v3 p = v_fma(d, t, o);
    v3 local = m_point(sh->to_object, p);
    si->dp_du = m_vector(sh->to_world, v_mul(V(-local.y, local.x, 0.f), 2.f * ORC_PI_F));
    si->dp_dv = m_vector(sh->to_world, V(0.f, 0.f, 1.f));
    v3 n = v_normalize(v_cross(si->dp_du, si->dp_dv));
    /* the shift uses the UNFLIPPED normal of the frame? no: `si.p += si.n * (1 - norm(head<2>(local)))` runs before si.n is assigned in this
     * branch of the reference -- si is zero-initialised there, so the shift adds nothing (cylinder.cpp:471-475 precede :487) */
    si->p = p;
    if (sh->flip_normals) n = v_neg(n);
    si->n = n; si->sh_n = n;
}
static void mesh_si(const orc_shape *sh, int32_t prim, float b1, float b2, orc_si *si) {
    const uint32_t *fi = sh->faces + 3 * prim;
    v3 p0 = mesh_pos(sh, fi[0]), p1 = mesh_pos(sh, fi[1]), p2 = mesh_pos(sh, fi[2]);
    float b0 = 1.f - b1 - b2;
    v3 dp0 = v_sub(p1, p0), dp1 = v_sub(p2, p0);
    si->p = v_fma(p0, b0, v_fma(p1, b1, v_mul(p2, b2)));
    si->n = v_normalize(v_cross(dp0, dp1));
    coordinate_system(si->n, &si->dp_du, &si->dp_dv);
    if (sh->texcoords) {
        const float *uv = sh->texcoords;
        float u0x = uv[2 * fi[0]], u0y = uv[2 * fi[0] + 1], u1x = uv[2 * fi[1]], u1y = uv[2 * fi[1] + 1],
              u2x = uv[2 * fi[2]], u2y = uv[2 * fi[2] + 1];
        float d0x = u1x - u0x, d0y = u1y - u0y, d1x = u2x - u0x, d1y = u2y - u0y;
        float det = fmaf(d0x, d1y, -(d0y * d1x)), inv_det = f_rcp(det);
        if (det != 0.f) {
            /* dp_du = fmsub(duv1.y, dp0, duv0.y*dp1) * inv_det ; dp_dv = fnmadd(duv1.x, dp0, duv0.x*dp1) * inv_det */
            si->dp_du = v_mul(V(fmaf(d1y, dp0.x, -(d0y * dp1.x)), fmaf(d1y, dp0.y, -(d0y * dp1.y)), fmaf(d1y, dp0.z, -(d0y * dp1.z))), inv_det);
            si->dp_dv = v_mul(V(fmaf(-d1x, dp0.x, d0x * dp1.x), fmaf(-d1x, dp0.y, d0x * dp1.y), fmaf(-d1x, dp0.z, d0x * dp1.z)), inv_det);
        }
    }
    if (sh->normals && !sh->face_normals) {
        const float *nn = sh->normals;
        v3 n0 = V(nn[3 * fi[0]], nn[3 * fi[0] + 1], nn[3 * fi[0] + 2]);
        v3 n1 = V(nn[3 * fi[1]], nn[3 * fi[1] + 1], nn[3 * fi[1] + 2]);
        v3 n2 = V(nn[3 * fi[2]], nn[3 * fi[2] + 1], nn[3 * fi[2] + 2]);
        v3 n = v_fma(n2, b2, v_fma(n1, b1, v_mul(n0, b0)));
        si->sh_n = v_mul(n, f_rsqrt(v_dot(n, n)));
    } else {
        si->sh_n = si->n;
    }
    if (sh->flip_normals) { si->n = v_neg(si->n); si->sh_n = v_neg(si->sh_n); }
}
/* PreliminaryIntersection::compute_surface_interaction (include/mitsuba/render/interaction.h:675-701)
 * -> Instance::compute_surface_interaction (src/shapes/instance.cpp:155-250) / shape CSI
 * -> finalize_surface_interaction (interaction.h:493-513) + initialize_sh_frame (:258-268) */
/* si.uv: rectangle (rectangle.cpp:312-313) fmadd(prim_uv, .5, .5) with prim_uv = the local hit position; mesh (mesh.cpp:720-737) the
 * interpolated vertex texcoords, or the barycentrics when the mesh has none */
static void surface_uv(const orc_shape *sh, const orc_hit *h, orc_si *si) {
    si->uv_u = si->uv_v = 0.f;
    if (sh->kind == ORC_SHAPE_RECT) { si->uv_u = fmaf(h->u, .5f, .5f); si->uv_v = fmaf(h->v, .5f, .5f); }
    else if (sh->kind == ORC_SHAPE_MESH) {
        float b1 = h->u, b2 = h->v, b0 = 1.f - b1 - b2;
        si->uv_u = b1; si->uv_v = b2;
        if (sh->texcoords) {
            const uint32_t *fi = sh->faces + 3 * h->prim; const float *uv = sh->texcoords;
            si->uv_u = fmaf(uv[2 * fi[2]], b2, fmaf(uv[2 * fi[1]], b1, uv[2 * fi[0]] * b0));
            si->uv_v = fmaf(uv[2 * fi[2] + 1], b2, fmaf(uv[2 * fi[1] + 1], b1, uv[2 * fi[0] + 1] * b0));
        }
    }
}
static void compute_si(const orc_scene *sc, const orc_hit *h, v3 o, v3 d, float time, orc_si *si) {
    const orc_object *ob = &sc->objects[h->obj];
    if (ob->kind == ORC_OBJ_SHAPE) {
        const orc_shape *sh = &sc->shapes[ob->index];
        si->shape = sh;
        if (sh->kind == ORC_SHAPE_RECT) rect_si(sh, o, d, h->t, si);
        else if (sh->kind == ORC_SHAPE_DISK) disk_si(sh, o, d, h->t, h->u, h->v, si);
        else if (sh->kind == ORC_SHAPE_SPHERE) sphere_si(sh, o, d, h->t, si);
        else if (sh->kind == ORC_SHAPE_CYLINDER) cylinder_si(sh, o, d, h->t, si);
        else mesh_si(sh, h->prim, h->u, h->v, si);
        surface_uv(sh, h, si);
    } else {
        float m[16], inv[16];
        instance_to_world(ob, time, m);
        m_affine_inverse(m, inv);
        const orc_shape *sh = &sc->shapes[sc->groups[ob->index].first_shape + h->shape];
        si->shape = sh;
        v3 lo = m_point(inv, o), ld = m_vector(inv, d);
        if (sh->kind == ORC_SHAPE_RECT) rect_si(sh, lo, ld, h->t, si);
        else if (sh->kind == ORC_SHAPE_DISK) disk_si(sh, lo, ld, h->t, h->u, h->v, si);
        else if (sh->kind == ORC_SHAPE_SPHERE) sphere_si(sh, lo, ld, h->t, si);
        else if (sh->kind == ORC_SHAPE_CYLINDER) cylinder_si(sh, lo, ld, h->t, si);
        else mesh_si(sh, h->prim, h->u, h->v, si);
        surface_uv(sh, h, si);
        si->p = m_point(m, si->p);
        si->n = v_normalize(m_normal(inv, si->n));
        si->sh_n = v_normalize(m_normal(inv, si->sh_n));
        si->dp_du = m_vector(m, si->dp_du);
        si->dp_dv = m_vector(m, si->dp_dv);
    }
    /* initialize_sh_frame */
    v3 s = v_normalize(v_fma(si->sh_n, -v_dot(si->sh_n, si->dp_du), si->dp_du));
    if (si->dp_du.x == 0.f && si->dp_du.y == 0.f && si->dp_du.z == 0.f) { v3 tt; coordinate_system(si->sh_n, &s, &tt); }
    si->sh_s = s;
    si->sh_t = v_cross(si->sh_n, s);
    v3 md = v_neg(d);
    si->wi = V(v_dot(md, si->sh_s), v_dot(md, si->sh_t), v_dot(md, si->sh_n));
}
static inline v3 si_to_local(const orc_si *si, v3 v) { return V(v_dot(v, si->sh_s), v_dot(v, si->sh_t), v_dot(v, si->sh_n)); }
/* Frame::to_world -- include/mitsuba/core/frame.h:44-46 */
static inline v3 si_to_world(const orc_si *si, v3 v) { return v_fma(si->sh_n, v.z, v_fma(si->sh_t, v.y, v_mul(si->sh_s, v.x))); }
/* Interaction::offset_p -- interaction.h:161-165 */
static inline v3 offset_p(const orc_si *si, v3 d) {
    float mag = (1.f + f_max(f_max(fabsf(si->p.x), fabsf(si->p.y)), fabsf(si->p.z))) * ORC_RAY_EPS;
    mag = f_mulsign(mag, v_dot(si->n, d));
    return v_fma(si->n, mag, si->p);
}

/* warp::square_to_uniform_disk_concentric / square_to_cosine_hemisphere -- warp.h:54-86,320-344 */
static void concentric_disk(float sx, float sy, float *px, float *py) {
    float x = fmaf(2.f, sx, -1.f), y = fmaf(2.f, sy, -1.f);
    int is_zero = (x == 0.f && y == 0.f), q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * ORC_PI_F * rp / r;
    if (q13) phi = 0.5f * ORC_PI_F - phi;
    if (is_zero) phi = 0.f;
    float s, c; orc_sincos(phi, &s, &c);
    *px = r * c; *py = r * s;
}
static v3 square_to_cosine_hemisphere(float sx, float sy) {
    float px, py; concentric_disk(sx, sy, &px, &py);
    float z = sqrtf(f_max(1.f - fmaf(py, py, px * px), 0.f));
    return V(px, py, z);
}

/* ------------------------------------------------------------------ integrator */
/* Rectangle::surface_area = |dp_du x dp_dv| (rectangle.cpp:127-129) and m_inv_surface_area = rcp(area) (:109) */
static float rect_inv_area(const orc_shape *sh) {
    v3 du = m_vector(sh->to_world, V(2.f, 0.f, 0.f)), dv = m_vector(sh->to_world, V(0.f, 2.f, 0.f));
    return f_rcp(v_norm(v_cross(du, dv)));
}
/* Disk::update + surface_area (src/shapes/disk.cpp:100-115,148-152) */
static float disk_inv_area(const orc_shape *sh) {
    v3 du = m_vector(sh->to_world, V(1.f, 0.f, 0.f)), dv = m_vector(sh->to_world, V(0.f, 1.f, 0.f));
    float m_du = v_norm(du), m_dv = v_norm(dv);
    v3 fs = v_mul(du, f_rcp(m_du)), ft = v_mul(dv, f_rcp(m_dv));
    float h = sqrtf(f_sqr(m_dv) - f_sqr(v_dot(v_mul(ft, m_dv), fs)));
    return f_rcp(ORC_PI_F * m_du * h);
}
/* Cylinder::surface_area (cylinder.cpp:243-245): 2 pi r l with r = |to_world x|, l = |to_world z| (update(), :131-132) */
static float cylinder_inv_area(const orc_shape *sh) {
    float r = v_norm(m_vector(sh->to_world, V(1.f, 0.f, 0.f))), l = v_norm(m_vector(sh->to_world, V(0.f, 0.f, 1.f)));
    return f_rcp((2.f * ORC_PI_F) * r * l);
}
static float shape_inv_area(const orc_shape *sh) {
    if (sh->kind == ORC_SHAPE_CYLINDER) return cylinder_inv_area(sh);
    return sh->kind == ORC_SHAPE_RECT ? rect_inv_area(sh) : sh->kind == ORC_SHAPE_DISK ? disk_inv_area(sh) : sh->kind == ORC_SHAPE_SPHERE ? sh->sphere_inv_area : sh->area_norm;
}
static inline float f_safe_sqrt(float x) { return sqrtf(f_max(x, 0.f)); }
#define ORC_INV_TWO_PI_F 0.15915494309189533577f
#define ORC_INV_FOUR_PI_F 0.07957747154594766788f   /* warp::square_to_uniform_sphere_pdf (warp.h:257-266) */
/* warp::square_to_uniform_cone_pdf (warp.h:475-485) */
static inline float uniform_cone_pdf(float cos_cutoff) { return ORC_INV_TWO_PI_F / (1.f - cos_cutoff); }
/* warp::square_to_uniform_sphere (warp.h:250-255) */
static v3 square_to_uniform_sphere(float s_x, float s_y) {
    float z = fmaf(-2.f, s_y, 1.f), r = f_safe_sqrt(fmaf(-z, z, 1.f)), sn, cs;
    orc_sincos(2.f * ORC_PI_F * s_x, &sn, &cs);
    return V(r * cs, r * sn, z);
}
/* warp::square_to_uniform_triangle (warp.h:153-156) */
static void square_to_uniform_triangle(float s_x, float s_y, float *bx, float *by) {
    float t = sqrtf(f_max(1.f - s_x, 0.f));
    *bx = 1.f - t; *by = t * s_y;
}
/* Sphere::sample_direction (sphere.cpp:222-296): cone sampling of the visible cap from outside, uniform sphere from inside.
 * Outputs the sampled point, its normal, the unit direction, distance and solid-angle density. */
static void sphere_sample_direction(const orc_shape *sh, v3 ref, float s_x, float s_y, v3 *p_out, v3 *n_out, v3 *d_out,
                                    float *dist_out, float *pdf_out) {
    const v3 center = V(sh->center[0], sh->center[1], sh->center[2]);
    const float radius = sh->radius;
    v3 dc_v = v_sub(center, ref);
    float dc_2 = v_dot(dc_v, dc_v);
    float radius_adj = radius * (sh->flip_normals ? (1.f + ORC_RAY_EPS) : (1.f - ORC_RAY_EPS));
    v3 dloc; float pdf;
    int outside = dc_2 > f_sqr(radius_adj);
    if (outside) {
        float inv_dc = f_rsqrt(dc_2), sin_theta_max = radius * inv_dc, sin_theta_max_2 = f_sqr(sin_theta_max),
              inv_sin_theta_max = f_rcp(sin_theta_max), cos_theta_max = f_safe_sqrt(1.f - sin_theta_max_2);
        float sin_theta_2 = sin_theta_max_2 > 0.00068523f ? 1.f - f_sqr(fmaf(cos_theta_max - 1.f, s_x, 1.f)) : sin_theta_max_2 * s_x;
        float cos_theta = f_safe_sqrt(1.f - sin_theta_2);
        float cos_alpha = sin_theta_2 * inv_sin_theta_max + cos_theta * f_safe_sqrt(fmaf(-sin_theta_2, f_sqr(inv_sin_theta_max), 1.f));
        float sin_alpha = f_safe_sqrt(fmaf(-cos_alpha, cos_alpha, 1.f));
        float sin_phi, cos_phi; orc_sincos(s_y * (2.f * ORC_PI_F), &sin_phi, &cos_phi);
        v3 fn = v_mul(dc_v, -inv_dc), fs, ft;
        coordinate_system(fn, &fs, &ft);
        v3 loc = V(cos_phi * sin_alpha, sin_phi * sin_alpha, cos_alpha);
        dloc = v_fma(fn, loc.z, v_fma(ft, loc.y, v_mul(fs, loc.x)));
        pdf = uniform_cone_pdf(cos_theta_max);
    } else {   /* warp::square_to_uniform_sphere (warp.h:250-255) */
        dloc = square_to_uniform_sphere(s_x, s_y);
        pdf = 0.f;
    }
    v3 p = v_fma(dloc, radius, center), dd = v_sub(p, ref);
    float dist2 = v_dot(dd, dd), dist = sqrtf(dist2);
    dd = v_mul(dd, f_rcp(dist));
    if (outside) { if (dist == 0.f) pdf = 0.f; }
    else pdf = sh->sphere_inv_area * dist2 / fabsf(v_dot(dd, dloc));
    *p_out = p; *n_out = sh->flip_normals ? v_neg(dloc) : dloc; *d_out = dd; *dist_out = dist; *pdf_out = pdf;
}
/* Sphere::pdf_direction (sphere.cpp:298-310) */
static float sphere_pdf_direction(const orc_shape *sh, v3 ref, v3 ds_d, v3 ds_n, float ds_dist) {
    const v3 center = V(sh->center[0], sh->center[1], sh->center[2]);
    float sin_alpha = sh->radius * f_rcp(v_norm(v_sub(center, ref))), cos_alpha = f_safe_sqrt(1.f - sin_alpha * sin_alpha);
    return sin_alpha < 0.99999994f ? uniform_cone_pdf(cos_alpha) : sh->sphere_inv_area * f_sqr(ds_dist) / fabsf(v_dot(ds_d, ds_n));
}
/* DiscreteDistribution::sample_reuse (distr_1d.h:113-160): first face in [lo, hi] whose cdf is not < value * sum
 * (dr::binary_search), then the sample re-stretched over that face's interval */
static uint32_t mesh_sample_face(const orc_shape *sh, float value, float *reuse) {
    float v = value * sh->area_sum;
    int32_t lo = sh->area_lo, hi = sh->area_hi;
    while (lo < hi) {
        int32_t mid = (int32_t) (((uint32_t) lo + (uint32_t) hi) >> 1);
        if (sh->area_cdf[mid] < v) lo = mid + 1 < hi ? mid + 1 : hi; else hi = mid;
    }
    float pmf = sh->area_pmf[lo] * sh->area_norm;
    float cdf = lo > 0 ? sh->area_cdf[lo - 1] * sh->area_norm : 0.f;
    *reuse = (value - cdf) / pmf;
    return (uint32_t) lo;
}
/* Mesh::sample_position (mesh.cpp:513-568) + warp::square_to_uniform_triangle (warp.h:153-156) */
static void mesh_sample_position(const orc_shape *sh, float s_x, float s_y, v3 *p_out, v3 *n_out) {
    float y;
    uint32_t f = mesh_sample_face(sh, s_y, &y);
    const uint32_t *fi = sh->faces + 3 * (size_t) f;
    const float *P = sh->positions;
    v3 p0 = V(P[3 * fi[0]], P[3 * fi[0] + 1], P[3 * fi[0] + 2]), p1 = V(P[3 * fi[1]], P[3 * fi[1] + 1], P[3 * fi[1] + 2]),
       p2 = V(P[3 * fi[2]], P[3 * fi[2] + 1], P[3 * fi[2] + 2]);
    v3 e0 = v_sub(p1, p0), e1 = v_sub(p2, p0);
    float bx, by; square_to_uniform_triangle(s_x, y, &bx, &by);
    *p_out = v_fma(e0, bx, v_fma(e1, by, p0));
    v3 n;
    if (sh->normals && !sh->face_normals) {
        const float *N = sh->normals;
        v3 n0 = V(N[3 * fi[0]], N[3 * fi[0] + 1], N[3 * fi[0] + 2]), n1 = V(N[3 * fi[1]], N[3 * fi[1] + 1], N[3 * fi[1] + 2]),
           n2 = V(N[3 * fi[2]], N[3 * fi[2] + 1], N[3 * fi[2] + 2]);
        n = v_fma(n0, 1.f - bx - by, v_fma(n1, bx, v_mul(n2, by)));
    } else n = v_cross(e0, e1);
    n = v_normalize(n);
    if (sh->flip_normals) n = v_neg(n);
    *n_out = n;
}
/* fresnel_conductor -- include/mitsuba/render/fresnel.h:93-117 (one colour channel) */
static float fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) {
    float cos_theta_i_2 = cos_theta_i * cos_theta_i, sin_theta_i_2 = 1.f - cos_theta_i_2, sin_theta_i_4 = sin_theta_i_2 * sin_theta_i_2;
    float temp_1 = eta_r * eta_r - eta_i * eta_i - sin_theta_i_2,
          a_2_pb_2 = f_safe_sqrt(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r),
          a = f_safe_sqrt(.5f * (a_2_pb_2 + temp_1));
    float term_1 = a_2_pb_2 + cos_theta_i_2, term_2 = 2.f * cos_theta_i * a;
    float r_s = (term_1 - term_2) / (term_1 + term_2);
    float term_3 = a_2_pb_2 * cos_theta_i_2 + sin_theta_i_4, term_4 = term_2 * sin_theta_i_2;
    float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return 0.5f * (r_s + r_p);
}
/* fresnel -- include/mitsuba/render/fresnel.h:21-63: (r, cos_theta_t, eta_it, eta_ti) */
static void fresnel_dielectric(float cos_theta_i, float eta, float *r_out, float *cos_theta_t, float *eta_it_out, float *eta_ti_out) {
    int outside = cos_theta_i >= 0.f;
    float rcp_eta = f_rcp(eta), eta_it = outside ? eta : rcp_eta, eta_ti = outside ? rcp_eta : eta;
    float cos_theta_t_sqr = fmaf(-fmaf(-cos_theta_i, cos_theta_i, 1.f), eta_ti * eta_ti, 1.f);
    float cos_theta_i_abs = fabsf(cos_theta_i), cos_theta_t_abs = f_safe_sqrt(cos_theta_t_sqr);
    int index_matched = eta == 1.f, special_case = index_matched || cos_theta_i_abs == 0.f;
    float r_sc = index_matched ? 0.f : 1.f;
    float a_s = fmaf(-eta_it, cos_theta_t_abs, cos_theta_i_abs) / fmaf(eta_it, cos_theta_t_abs, cos_theta_i_abs);
    float a_p = fmaf(-eta_it, cos_theta_i_abs, cos_theta_t_abs) / fmaf(eta_it, cos_theta_i_abs, cos_theta_t_abs);
    float r = 0.5f * (f_sqr(a_s) + f_sqr(a_p));
    if (special_case) r = r_sc;
    *r_out = r; *cos_theta_t = f_mulsign_neg(cos_theta_t_abs, cos_theta_i); *eta_it_out = eta_it; *eta_ti_out = eta_ti;
}
/* ---- MicrofacetDistribution (include/mitsuba/render/microfacet.h): Beckmann (type 0) and GGX (type 1); visible-normal
 * sampling is what the BSDF plugins use (sample_visible = true, their default), the plain sampling of all normals
 * (sample_visible = false) exists for the reference's own known answers (src/render/tests/test_microfacet.py) */
enum { ORC_MF_BECKMANN = 0, ORC_MF_GGX = 1 };
typedef struct { float au, av; int type, visible; } ggx_t;
static ggx_t mf_make(int type, float au, float av, int visible) {   /* configure() :425-428 */
    ggx_t g; g.au = f_max(au, 1e-4f); g.av = f_max(av, 1e-4f); g.type = type; g.visible = visible; return g;
}
static float ggx_eval(ggx_t g, v3 m) {   /* eval() :176-196 */
    float alpha_uv = g.au * g.av, cos_theta = m.z, cos_theta_2 = f_sqr(cos_theta), result;
    if (g.type == ORC_MF_BECKMANN)
        result = orc_expf(-(f_sqr(m.x / g.au) + f_sqr(m.y / g.av)) / cos_theta_2) / (ORC_PI_F * alpha_uv * f_sqr(cos_theta_2));
    else
        result = f_rcp(ORC_PI_F * alpha_uv * f_sqr(f_sqr(m.x / g.au) + f_sqr(m.y / g.av) + f_sqr(m.z)));
    return result * cos_theta > 1e-20f ? result : 0.f;
}
static float ggx_smith_g1(ggx_t g, v3 v, v3 m) {   /* smith_g1() :341-365 */
    float xy_alpha_2 = f_sqr(g.au * v.x) + f_sqr(g.av * v.y), tan_theta_alpha_2 = xy_alpha_2 / f_sqr(v.z), result;
    if (g.type == ORC_MF_BECKMANN) {
        float a = f_rsqrt(tan_theta_alpha_2), a_sqr = f_sqr(a);
        result = a >= 1.6f ? 1.f : (3.535f * a + 2.181f * a_sqr) / (1.f + 2.276f * a + 2.577f * a_sqr);
    } else
        result = 2.f / (1.f + sqrtf(1.f + tan_theta_alpha_2));
    if (xy_alpha_2 == 0.f) result = 1.f;
    if (v_dot(v, m) * v.z <= 0.f) result = 0.f;
    return result;
}
/* MicrofacetDistribution::pdf (microfacet.h:219-228): visible normals D * ((G1 * |wi.m|) / cos_theta_i), all normals D * cos_theta_m */
static float ggx_pdf(ggx_t g, v3 wi, v3 m) {
    return g.visible ? ggx_eval(g, m) * (ggx_smith_g1(g, wi, m) * fabsf(v_dot(wi, m)) / wi.z) : ggx_eval(g, m) * m.z;
}
/* sample_visible_11 (:368-420): slope of the visible normal for alpha = 1 */
static void mf_sample_visible_11(int type, float cos_theta_i, float s_x, float s_y, float *slope_x, float *slope_y) {
    if (type == ORC_MF_BECKMANN) {
        const float inv_sqrt_pi = 0.56418958354775628695f;
        float tan_theta_i = f_safe_sqrt(fmaf(-cos_theta_i, cos_theta_i, 1.f)) / cos_theta_i, cot_theta_i = f_rcp(tan_theta_i);
        float maxval = orc_erff(cot_theta_i);
        s_x = f_max(f_min(s_x, 1.f - 1e-6f), 1e-6f); s_y = f_max(f_min(s_y, 1.f - 1e-6f), 1e-6f);
        float x = maxval - (maxval + 1.f) * orc_erff(sqrtf(-orc_logf(s_x)));
        s_x *= 1.f + maxval + inv_sqrt_pi * tan_theta_i * orc_expf(-f_sqr(cot_theta_i));
        for (int i = 0; i < 3; ++i) {   /* three Newton iterations */
            float slope = orc_erfinvf(x);
            float value = 1.f + x + inv_sqrt_pi * tan_theta_i * orc_expf(-f_sqr(slope)) - s_x, derivative = 1.f - slope * tan_theta_i;
            x -= value / derivative;
        }
        *slope_x = orc_erfinvf(x); *slope_y = orc_erfinvf(fmaf(2.f, s_y, -1.f));
        return;
    }
    /* GGX: square_to_uniform_disk_concentric (warp.h:54-90), projection onto the chosen side of the hemisphere */
    float x = fmaf(2.f, s_x, -1.f), y = fmaf(2.f, s_y, -1.f);
    int is_zero = x == 0.f && y == 0.f, q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * ORC_PI_F * rp / r;
    if (q13) phi = 0.5f * ORC_PI_F - phi;
    if (is_zero) phi = 0.f;
    float sn, cs; orc_sincos(phi, &sn, &cs);
    float px = r * cs, py = r * sn;
    float s = 0.5f * (1.f + cos_theta_i);
    float a = f_safe_sqrt(1.f - f_sqr(px));
    py = fmaf(py, s, fmaf(-a, s, a));                      /* dr::lerp(a, py, s) = fmadd(py, s, fnmadd(a, s, a)) */
    float pz = f_safe_sqrt(1.f - fmaf(py, py, px * px));   /* squared_norm(p) = fmadd chain */
    float sin_theta_i = f_safe_sqrt(1.f - f_sqr(cos_theta_i));
    float norm = f_rcp(fmaf(sin_theta_i, py, cos_theta_i * pz));
    *slope_x = fmaf(cos_theta_i, py, -(sin_theta_i * pz)) * norm; *slope_y = px * norm;
}
/* sample() :240-325; returns m and the density of m */
static v3 ggx_sample(ggx_t g, v3 wi, float s_x, float s_y, float *pdf_out) {
    if (!g.visible) {   /* all normals :242-290 */
        float sin_phi, cos_phi, cos_theta, cos_theta_2, alpha_2, pdf;
        if (g.au == g.av) {
            orc_sincos((2.f * ORC_PI_F) * s_y, &sin_phi, &cos_phi);
            alpha_2 = g.au * g.au;
        } else {
            float ratio = g.av / g.au, tmp = ratio * orc_tanf((2.f * ORC_PI_F) * s_y);
            cos_phi = f_rsqrt(fmaf(tmp, tmp, 1.f));
            cos_phi = f_mulsign(cos_phi, fabsf(s_y - .5f) - .25f);
            sin_phi = cos_phi * tmp;
            alpha_2 = f_rcp(f_sqr(cos_phi / g.au) + f_sqr(sin_phi / g.av));
        }
        if (g.type == ORC_MF_BECKMANN) {
            cos_theta = f_rsqrt(fmaf(-alpha_2, orc_logf(1.f - s_x), 1.f));
            cos_theta_2 = f_sqr(cos_theta);
            float cos_theta_3 = f_max(cos_theta_2 * cos_theta, 1e-20f);
            pdf = (1.f - s_x) / (ORC_PI_F * g.au * g.av * cos_theta_3);
        } else {
            float tan_theta_m_2 = alpha_2 * s_x / (1.f - s_x);
            cos_theta = f_rsqrt(1.f + tan_theta_m_2);
            cos_theta_2 = f_sqr(cos_theta);
            float temp = 1.f + tan_theta_m_2 / alpha_2, cos_theta_3 = f_max(cos_theta_2 * cos_theta, 1e-20f);
            pdf = f_rcp(ORC_PI_F * g.au * g.av * cos_theta_3 * f_sqr(temp));
        }
        float sin_theta = sqrtf(1.f - cos_theta_2);
        *pdf_out = pdf;
        return V(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
    }
    v3 wi_p = v_normalize(V(g.au * wi.x, g.av * wi.y, wi.z));
    /* Frame3f::sincos_phi (frame.h:111-122) */
    float sin_theta_2 = fmaf(wi_p.x, wi_p.x, f_sqr(wi_p.y)), inv_sin_theta = f_rsqrt(sin_theta_2);
    float rx = wi_p.x * inv_sin_theta, ry = wi_p.y * inv_sin_theta;
    rx = f_min(f_max(rx, -1.f), 1.f); ry = f_min(f_max(ry, -1.f), 1.f);
    if (fabsf(sin_theta_2) <= 4.f * 5.9604644775390625e-8f) { rx = 1.f; ry = 0.f; }
    float sin_phi = ry, cos_phi = rx, cos_theta = wi_p.z;
    float slope_x, slope_y;
    mf_sample_visible_11(g.type, cos_theta, s_x, s_y, &slope_x, &slope_y);
    /* rotate & unstretch, normal, density */
    float sx = fmaf(cos_phi, slope_x, -(sin_phi * slope_y)) * g.au, sy = fmaf(sin_phi, slope_x, cos_phi * slope_y) * g.av;
    v3 m = v_normalize(V(-sx, -sy, 1.f));
    *pdf_out = ggx_eval(g, m) * ggx_smith_g1(g, wi, m) * fabsf(v_dot(wi, m)) / wi.z;
    return m;
}
/* RoughPlastic::lerp_gather (roughplastic.cpp:373-383) on the 64-entry table */
static float lerp_gather64(const float *data, float x) {
    x *= 63.f;
    uint32_t index = (uint32_t) x; if (index > 62u) index = 62u;
    float v0 = data[index], v1 = data[index + 1], t = x - (float) index;
    return fmaf(v1, t, fmaf(-v0, t, v0));   /* dr::lerp */
}
/* RoughPlastic::eval (:333-371) and pdf (:385-421), both cosines positive */
static void rough_plastic_eval_pdf(ggx_t g, const orc_shape *sh, const float *refl, const float *spec_refl, v3 wi, v3 wo, float t_i, float prob_specular, float prob_diffuse,
                                   v3 *value, float *pdf) {
    v3 H = v_normalize(v_add(wo, wi));
    float D = ggx_eval(g, H), F, t1, t2, t3;
    fresnel_dielectric(v_dot(wi, H), sh->diel_eta, &F, &t1, &t2, &t3);
    float G = ggx_smith_g1(g, wi, H) * ggx_smith_g1(g, wo, H);
    float spec = F * D * G / (4.f * wi.z);
    float t_o = lerp_gather64(sh->rough_table, wo.z);
    v3 diff = V(refl[0], refl[1], refl[2]);
    float ir = sh->fdr_int;
    diff = sh->nonlinear ? V(diff.x / (1.f - diff.x * ir), diff.y / (1.f - diff.y * ir), diff.z / (1.f - diff.z * ir))
                         : V(diff.x / (1.f - ir), diff.y / (1.f - ir), diff.z / (1.f - ir));
    float k = ORC_INV_PI_F * sh->inv_eta_2 * wo.z * t_i * t_o;
    *value = V(spec * spec_refl[0] + diff.x * k, spec * spec_refl[1] + diff.y * k, spec * spec_refl[2] + diff.z * k);
    float result = g.visible ? D * ggx_smith_g1(g, wi, H) / (4.f * wi.z) : ggx_pdf(g, wi, H) / (4.f * v_dot(wo, H));   /* roughplastic.cpp:467-470 */
    result *= prob_specular;
    *pdf = result + prob_diffuse * (ORC_INV_PI_F * wo.z);
}
/* RoughDielectric::eval_pdf (roughdielectric.cpp:503-611), TransportMode::Radiance */
static void rough_dielectric_eval_pdf(ggx_t g, const orc_shape *sh, const float *spec_refl, const float *spec_trans, v3 wi, v3 wo, v3 *value, float *pdf) {
    float cti = wi.z, cto = wo.z, m_eta = sh->diel_eta, m_inv_eta = f_rcp(m_eta);
    int reflect = cti * cto > 0.f;
    float eta = cti > 0.f ? m_eta : m_inv_eta, inv_eta = cti > 0.f ? m_inv_eta : m_eta;
    v3 m = v_normalize(v_add(wi, v_mul(wo, reflect ? 1.f : eta)));
    m = V(f_mulsign(m.x, m.z), f_mulsign(m.y, m.z), f_mulsign(m.z, m.z));
    float dwm = v_dot(wi, m), dom = v_dot(wo, m);
    int active = cti != 0.f && dwm * cti > 0.f && dom * cto > 0.f;
    float D = ggx_eval(g, m), F, t1, t2, t3;
    fresnel_dielectric(dwm, m_eta, &F, &t1, &t2, &t3);
    float G = ggx_smith_g1(g, wi, m) * ggx_smith_g1(g, wo, m);
    *value = V(0, 0, 0); *pdf = 0.f;
    if (!active) return;
    if (reflect) {
        float v = F * D * G / (4.f * fabsf(cti));
        *value = V(v * spec_refl[0], v * spec_refl[1], v * spec_refl[2]);
    } else {
        float scale = f_sqr(inv_eta);
        float v = fabsf((scale * (1.f - F) * D * G * eta * eta * dwm * dom) / (cti * f_sqr(dwm + eta * dom)));
        *value = V(v * spec_trans[0], v * spec_trans[1], v * spec_trans[2]);
    }
    ggx_t gs = g;   /* sample_distr: Walter et al.'s roughness scaling when all normals are sampled (roughdielectric.cpp:584-589) */
    if (!g.visible) { const float sc = 1.2f - .2f * sqrtf(fabsf(cti)); gs.au *= sc; gs.av *= sc; }
    float p = ggx_pdf(gs, V(f_mulsign(wi.x, cti), f_mulsign(wi.y, cti), f_mulsign(wi.z, cti)), m);
    p *= reflect ? F : 1.f - F;
    float dwh_dwo = reflect ? f_rcp(4.f * dom) : (eta * eta * dom) / f_sqr(dwm + eta * dom);
    *pdf = p * fabsf(dwh_dwo);
}
/* ---- textures on the diffuse reflectance
 * Checkerboard::eval (src/textures/checkerboard.cpp:70-89); BitmapTexture::eval -> interpolate_3 / interpolate_1 (src/textures/bitmap.cpp:
 * 633-670) -> dr::Texture<Float, 2>::eval (Dr.Jit 0.4.0 texture.h, absent from the tree; restated from its documented behaviour:
 * texel centres at (i + .5) / res, pos = fmadd(uv, res, -.5), the four neighbours wrapped per mode, bilinear weights combined as
 * fmadd(w0.y, fmadd(w0.x, v00, w1.x * v10), w1.y * fmadd(w0.x, v01, w1.x * v11)); nearest: floor(uv * res)). */
static int32_t tex_wrap(int32_t i, int32_t n, int mode) {
    if (mode == 2) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    int32_t div = i / n; if (i % n < 0) --div;                /* floor division */
    int32_t mod = i - div * n;
    if (mode == 1 && (div & 1)) mod = n - 1 - mod;            /* mirror: every other repetition is flipped */
    return mod;
}
void orc_texture_eval(const orc_texture *tex, float u, float v, float *out3) {
    /* m_transform.transform_affine(si.uv): result = col2; result = fmadd(col0, u, result); result = fmadd(col1, v, result), col2 = 0 */
    float tu = fmaf(tex->to_uv[1], v, fmaf(tex->to_uv[0], u, 0.f)), tv = fmaf(tex->to_uv[3], v, fmaf(tex->to_uv[2], u, 0.f));
    if (tex->kind == ORC_TEX_CHECKERBOARD) {
        int mx = tu - floorf(tu) > .5f, my = tv - floorf(tv) > .5f;
        const float *c = mx == my ? tex->color0 : tex->color1;
        out3[0] = c[0]; out3[1] = c[1]; out3[2] = c[2];
        return;
    }
    const int32_t W = tex->width, H = tex->height, C = tex->channels;
    float texel[3] = { 0.f, 0.f, 0.f };
    if (tex->filter == 0) {
        int32_t x = tex_wrap((int32_t) floorf(tu * (float) W), W, tex->wrap), y = tex_wrap((int32_t) floorf(tv * (float) H), H, tex->wrap);
        for (int c = 0; c < C; ++c) texel[c] = tex->data[((size_t) y * W + x) * C + c];
    } else {
        float px = fmaf(tu, (float) W, -.5f), py = fmaf(tv, (float) H, -.5f), fx = floorf(px), fy = floorf(py);
        float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
        int32_t x0 = tex_wrap((int32_t) fx, W, tex->wrap), x1 = tex_wrap((int32_t) fx + 1, W, tex->wrap);
        int32_t y0 = tex_wrap((int32_t) fy, H, tex->wrap), y1 = tex_wrap((int32_t) fy + 1, H, tex->wrap);
        for (int c = 0; c < C; ++c) {
            float v00 = tex->data[((size_t) y0 * W + x0) * C + c], v10 = tex->data[((size_t) y0 * W + x1) * C + c];
            float v01 = tex->data[((size_t) y1 * W + x0) * C + c], v11 = tex->data[((size_t) y1 * W + x1) * C + c];
            texel[c] = fmaf(w0y, fmaf(w0x, v00, w1x * v10), w1y * fmaf(w0x, v01, w1x * v11));
        }
    }
    if (C == 1) texel[1] = texel[2] = texel[0];
    out3[0] = texel[0]; out3[1] = texel[1]; out3[2] = texel[2];
}
/* Texture::eval_1 (bitmap.cpp:324-344: one channel as it is, three channels -> luminance (spectrum.h:431-434); checkerboard.cpp:91-110 with
 * constant colours: SRGBReflectanceSpectrum::eval_1 = mean of the colour, srgb.cpp:85-88) */
float orc_texture_eval_1(const orc_texture *tex, float u, float v) {
    float c[3];
    if (tex->kind == 0) {   /* checkerboard: the colour the lookup picks, reduced to its mean */
        float c0[3] = { tex->color0[0], tex->color0[1], tex->color0[2] };
        orc_texture_eval(tex, u, v, c);
        const int first = c[0] == c0[0] && c[1] == c0[1] && c[2] == c0[2];
        const float *k = first ? tex->color0 : tex->color1;
        return ((k[0] + k[1]) + k[2]) * (1.0f / 3.0f);
    }
    orc_texture_eval(tex, u, v, c);
    if (tex->channels == 1) return c[0];
    return c[0] * 0.212671f + c[1] * 0.715160f + c[2] * 0.072169f;
}
/* BitmapTexture::eval_1_grad (src/textures/bitmap.cpp:346-421): the gradient of the bilinear interpolant of the (luminance of the) four texels around the
 * lookup, through the transpose of the uv transform, times the resolution; the nearest filter (and a texture without eval_1_grad) has none */
static void orc_texture_eval_1_grad(const orc_texture *tex, float u, float v, float *gu, float *gv) {
    *gu = *gv = 0.f;
    if (tex->kind != ORC_TEX_BITMAP || tex->filter == 0) return;
    const float tu = fmaf(tex->to_uv[1], v, fmaf(tex->to_uv[0], u, 0.f)), tv = fmaf(tex->to_uv[3], v, fmaf(tex->to_uv[2], u, 0.f));
    const int32_t W = tex->width, H = tex->height, C = tex->channels;
    const float px = fmaf(tu, (float) W, -.5f), py = fmaf(tv, (float) H, -.5f), fx = floorf(px), fy = floorf(py);
    const float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    const int32_t x0 = tex_wrap((int32_t) fx, W, tex->wrap), x1 = tex_wrap((int32_t) fx + 1, W, tex->wrap);
    const int32_t y0 = tex_wrap((int32_t) fy, H, tex->wrap), y1 = tex_wrap((int32_t) fy + 1, H, tex->wrap);
    float f[4]; const int32_t xs[4] = { x0, x1, x0, x1 }, ys[4] = { y0, y0, y1, y1 };
    for (int i = 0; i < 4; ++i) {
        const float *t = tex->data + ((size_t) ys[i] * W + xs[i]) * C;
        f[i] = C == 1 ? t[0] : t[0] * 0.212671f + t[1] * 0.715160f + t[2] * 0.072169f;   /* luminance (spectrum.h:431-434) */
    }
    const float dfx = fmaf(w0y, f[1] - f[0], w1y * (f[3] - f[2])), dfy = fmaf(w0x, f[2] - f[0], w1x * (f[3] - f[1]));
    *gu = (float) W * (tex->to_uv[0] * dfx + tex->to_uv[2] * dfy);
    *gv = (float) H * (tex->to_uv[1] * dfx + tex->to_uv[3] * dfy);
}
/* DiscreteDistribution2D::sample (distr_2d.h:140-181): row from the marginal, column from the conditional CDF (dr::binary_search over [0, n - 1]: the first index whose
 * CDF value is not below the sample, the last index if there is none), the re-uniformised variate of both */
static uint32_t cdf_search(const float *cdf, uint32_t n, float x) {
    uint32_t lo = 0, hi = n - 1u;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cdf[mid] < x) lo = mid + 1u; else hi = mid; }
    return lo;
}
static void distr2d_sample(const orc_texture *t, float sx, float sy, uint32_t *col_out, uint32_t *row_out, float *pdf, float *rx, float *ry) {
    const uint32_t W = (uint32_t) t->width, H = (uint32_t) t->height;
    sx = f_min(f_max(sx, 1.17549435e-38f), 0.99999994f); sy = f_min(f_max(sy, 1.17549435e-38f), 0.99999994f);   /* clamp(sample, Smallest, OneMinusEpsilon) */
    sy *= t->inv_normalization;
    const uint32_t row = cdf_search(t->marg_cdf, H, sy), offset = row * W;
    sx *= t->cond_cdf[offset + W - 1u];
    const uint32_t col = cdf_search(t->cond_cdf + offset, W, sx);
    const float col_cdf_0 = col > 0 ? t->cond_cdf[offset + col - 1u] : 0.f, col_cdf_1 = t->cond_cdf[offset + col];
    const float row_cdf_0 = row > 0 ? t->marg_cdf[row - 1u] : 0.f, row_cdf_1 = t->marg_cdf[row];
    sx -= col_cdf_0; sy -= row_cdf_0;
    if (col_cdf_1 != col_cdf_0) sx /= col_cdf_1 - col_cdf_0;
    if (row_cdf_1 != row_cdf_0) sy /= row_cdf_1 - row_cdf_0;
    *col_out = col; *row_out = row; *pdf = (col_cdf_1 - col_cdf_0) * t->normalization; *rx = sx; *ry = sy;
}
static float distr2d_pdf(const orc_texture *t, int32_t x, int32_t y) {   /* DiscreteDistribution2D::pdf (:119-130) */
    const uint32_t index = (uint32_t) x + (uint32_t) y * (uint32_t) t->width;
    return (t->cond_cdf[index] - (x > 0 ? t->cond_cdf[index - 1u] : 0.f)) * t->normalization;
}
static inline float interval_to_tent(float s) {   /* warp.h:196-200 */
    s -= .5f;
    const float v = fmaf(fabsf(s), -2.f, 1.f), r = 1.f - (v > 0.f ? sqrtf(v) : 0.f);
    return copysignf(r, s);
}
/* Texture::sample_position (texture.cpp:56-59: the identity for every texture without its own) / BitmapTexture::sample_position (bitmap.cpp:450-487) */
static void texture_sample_position(const orc_texture *t, float sx, float sy, float *u, float *v, float *pdf) {
    if (t->kind != ORC_TEX_BITMAP) { *u = sx; *v = sy; *pdf = 1.f; return; }
    uint32_t col, row; float p, rx, ry;
    distr2d_sample(t, sx, sy, &col, &row, &p, &rx, &ry);
    const float iw = f_rcp((float) t->width), ih = f_rcp((float) t->height);
    float x, y;
    if (t->filter == 0) { x = ((float) col + rx) * iw; y = ((float) row + ry) * ih; }
    else {
        x = (((float) col + .5f) + interval_to_tent(rx)) * iw; y = (((float) row + .5f) + interval_to_tent(ry)) * ih;
        if (t->wrap == 0) { if (x < 0.f) x += 1.f; if (x > 1.f) x -= 1.f; if (y < 0.f) y += 1.f; if (y > 1.f) y -= 1.f; }
        else { if (x < 0.f) x = -x; if (x > 1.f) x = 2.f - x; if (y < 0.f) y = -y; if (y > 1.f) y = 2.f - y; }
    }
    *u = x; *v = y; *pdf = p * (float) (t->width * t->height);
}
/* Texture::pdf_position (texture.cpp:61-64) / BitmapTexture::pdf_position (bitmap.cpp:489-528) */
static float texture_pdf_position(const orc_texture *t, float u, float v) {
    if (t->kind != ORC_TEX_BITMAP) return 1.f;
    const int32_t W = t->width, H = t->height;
    if (t->filter == 0) {
        const int32_t x = tex_wrap((int32_t) floorf(u * (float) W), W, t->wrap), y = tex_wrap((int32_t) floorf(v * (float) H), H, t->wrap);
        return distr2d_pdf(t, x, y) * (float) (W * H);
    }
    const float px = fmaf(u, (float) W, -.5f), py = fmaf(v, (float) H, -.5f), fx = floorf(px), fy = floorf(py);
    const float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    const int32_t x0 = tex_wrap((int32_t) fx, W, t->wrap), x1 = tex_wrap((int32_t) fx + 1, W, t->wrap);
    const int32_t y0 = tex_wrap((int32_t) fy, H, t->wrap), y1 = tex_wrap((int32_t) fy + 1, H, t->wrap);
    const float v00 = distr2d_pdf(t, x0, y0), v10 = distr2d_pdf(t, x1, y0), v01 = distr2d_pdf(t, x0, y1), v11 = distr2d_pdf(t, x1, y1);
    const float v0 = fmaf(w0x, v00, w1x * v10), v1 = fmaf(w0x, v01, w1x * v11);
    return fmaf(w0y, v0, w1y * v1) * (float) (W * H);
}
/* known-answer entries (tests): DiscreteDistribution2D::sample -> col, row, pdf, re-uniformised sample; Texture::sample_position -> u, v, pdf; pdf_position */
void orc_kat_distr2d_sample(const orc_texture *t, float sx, float sy, float *out5) {
    uint32_t col, row; distr2d_sample(t, sx, sy, &col, &row, out5 + 2, out5 + 3, out5 + 4); out5[0] = (float) col; out5[1] = (float) row;
}
void orc_kat_texture_sample_position(const orc_texture *t, float sx, float sy, float *out3) { texture_sample_position(t, sx, sy, out3, out3 + 1, out3 + 2); }
float orc_kat_texture_pdf_position(const orc_texture *t, float u, float v) { return texture_pdf_position(t, u, v); }
void orc_kat_texture_eval_1_grad(const orc_texture *t, float u, float v, float *out2) { orc_texture_eval_1_grad(t, u, v, out2, out2 + 1); }
/* the material parameters of one hit: the shape's constants, or the lookups of the textures bound to their slots */
typedef struct { float spec_refl[3], spec_trans[3], alpha_u, alpha_v; } orc_mat;
static orc_mat material_at(const orc_shape *sh, float u, float v) {
    orc_mat m;
    for (int i = 0; i < 3; ++i) { m.spec_refl[i] = sh->spec_refl[i]; m.spec_trans[i] = sh->spec_trans[i]; }
    m.alpha_u = sh->alpha_u; m.alpha_v = sh->alpha_v;
    if (sh->tex_spec) orc_texture_eval(sh->tex_spec, u, v, m.spec_refl);
    if (sh->tex_trans) orc_texture_eval(sh->tex_trans, u, v, m.spec_trans);
    if (sh->tex_alpha_u) m.alpha_u = orc_texture_eval_1(sh->tex_alpha_u, u, v);
    if (sh->tex_alpha_v) m.alpha_v = orc_texture_eval_1(sh->tex_alpha_v, u, v);
    return m;
}
/* One BSDF interaction of the bounce loop: value and density for the emitter direction `wo` (only when `active_em`), and the
 * sampled continuation (BSDF::eval_pdf_sample, src/render/bsdf.cpp:20-29).  wi_in / wo / bs_wo are in the local shading frame. */
/* bs_null: the sampled lobe was BSDFFlags::Null (mask.cpp:148, thindielectric.cpp:179) -- such a vertex does not validate the ray (dopplertofpath.cpp:252-253) */
typedef struct { v3 val; float pdf; v3 weight; v3 wo; float bs_pdf, bs_eta; int bs_delta, bs_null; } orc_bsdf_out;
static void nested_bsdf_eval_pdf_sample(const orc_shape *sh, v3 wi_in, v3 wo, int active_em, float sample_1, float s2x, float s2y, float uv_u, float uv_v, orc_bsdf_out *out) {
    float refl[3] = { sh->reflectance[0], sh->reflectance[1], sh->reflectance[2] };   /* m_reflectance->eval(si) */
    if (sh->tex_refl) orc_texture_eval(sh->tex_refl, uv_u, uv_v, refl);
    const orc_mat m_ = material_at(sh, uv_u, uv_v);   /* m_specular_reflectance->eval(si), m_alpha_u->eval_1(si), ... */
    /* BSDF::eval_pdf_sample src/render/bsdf.cpp:20-29 over twosided{diffuse} / diffuse
     * (src/bsdfs/twosided.cpp:111-148,219-258; src/bsdfs/diffuse.cpp:101-125,160-180) */
    v3 bsdf_val = V(0, 0, 0), bsdf_weight = V(0, 0, 0), bs_wo = V(0, 0, 0);
    float bsdf_pdf = 0.f, bs_pdf = 0.f, bs_eta = 0.f; int bs_delta = 0, bs_null = 0;
    if (sh->bsdf == ORC_BSDF_CONDUCTOR) {
        /* SmoothConductor::sample (conductor.cpp:226-277) under TwoSidedBRDF::sample (twosided.cpp:111-148): eval / pdf of a
         * delta lobe are zero (conductor.cpp:279-290) */
        float cos_theta_i = sh->twosided ? fabsf(wi_in.z) : wi_in.z;
        if (cos_theta_i > 0.f) {
            bs_wo = V(-wi_in.x, -wi_in.y, wi_in.z);   /* reflect(wi); the two-sided flip of wi.z and of wo.z cancel */
            bs_eta = 1.f; bs_pdf = 1.f; bs_delta = 1;
            bsdf_weight = V(m_.spec_refl[0] * fresnel_conductor(cos_theta_i, sh->cond_eta[0], sh->cond_k[0]),
                            m_.spec_refl[1] * fresnel_conductor(cos_theta_i, sh->cond_eta[1], sh->cond_k[1]),
                            m_.spec_refl[2] * fresnel_conductor(cos_theta_i, sh->cond_eta[2], sh->cond_k[2]));
        }
    } else if (sh->bsdf == ORC_BSDF_DIELECTRIC) {
        /* SmoothDielectric::sample (dielectric.cpp:231-338), TransportMode::Radiance */
        float r_i, cos_theta_t, eta_it, eta_ti;
        fresnel_dielectric(wi_in.z, sh->diel_eta, &r_i, &cos_theta_t, &eta_it, &eta_ti);
        float t_i = 1.f - r_i;
        int selected_r = sample_1 <= r_i;
        bs_pdf = selected_r ? r_i : t_i; bs_delta = 1;
        bs_wo = selected_r ? V(-wi_in.x, -wi_in.y, wi_in.z) : V(-eta_ti * wi_in.x, -eta_ti * wi_in.y, cos_theta_t);
        bs_eta = selected_r ? 1.f : eta_it;
        if (selected_r) bsdf_weight = V(m_.spec_refl[0], m_.spec_refl[1], m_.spec_refl[2]);
        else { float f2 = f_sqr(eta_ti); bsdf_weight = V(m_.spec_trans[0] * f2, m_.spec_trans[1] * f2, m_.spec_trans[2] * f2); }
    } else if (sh->bsdf == ORC_BSDF_NULL) {
        /* Null::sample (src/bsdfs/null.cpp:42-66): wo = -wi, eta 1, pdf 1, weight 1, sampled_type = BSDFFlags::Null; eval / pdf are zero (:68-79) */
        bs_wo = V(-wi_in.x, -wi_in.y, -wi_in.z); bs_eta = 1.f; bs_pdf = 1.f; bs_delta = 1; bs_null = 1; bsdf_weight = V(1.f, 1.f, 1.f);
    } else if (sh->bsdf == ORC_BSDF_THINDIELECTRIC) {
        /* ThinDielectric::sample (thindielectric.cpp:173-226); eval / pdf are zero (:228-236) */
        float r, t1, t2, t3;
        fresnel_dielectric(fabsf(wi_in.z), sh->diel_eta, &r, &t1, &t2, &t3);
        r *= 2.f / (1.f + r);
        int selected_r = sample_1 <= r;
        bs_pdf = selected_r ? r : 1.f - r; bs_delta = 1; bs_eta = 1.f;
        bs_null = !selected_r;   /* bs.sampled_type = select(selected_r, DeltaReflection, Null) (:179) */
        bs_wo = selected_r ? V(-wi_in.x, -wi_in.y, wi_in.z) : V(-wi_in.x, -wi_in.y, -wi_in.z);
        bsdf_weight = selected_r ? V(m_.spec_refl[0], m_.spec_refl[1], m_.spec_refl[2]) : V(m_.spec_trans[0], m_.spec_trans[1], m_.spec_trans[2]);
    } else if (sh->bsdf == ORC_BSDF_ROUGHDIELECTRIC) {
        /* RoughDielectric::sample (roughdielectric.cpp:240-346); eval_pdf above for the emitter sample */
        ggx_t g = mf_make(sh->mf_type, m_.alpha_u, m_.alpha_v, !sh->sample_all);
        v3 wi = wi_in;
        if (active_em) rough_dielectric_eval_pdf(g, sh, m_.spec_refl, m_.spec_trans, wi, wo, &bsdf_val, &bsdf_pdf);
        if (wi.z != 0.f) {
            float mpdf;
            ggx_t gs = g;   /* sample_distr (:266-269) */
            if (!g.visible) { const float sc = 1.2f - .2f * sqrtf(fabsf(wi.z)); gs.au *= sc; gs.av *= sc; }
            v3 m = ggx_sample(gs, V(f_mulsign(wi.x, wi.z), f_mulsign(wi.y, wi.z), f_mulsign(wi.z, wi.z)), s2x, s2y, &mpdf);
            float dwm = v_dot(wi, m), F, cos_theta_t, eta_it, eta_ti;
            fresnel_dielectric(dwm, sh->diel_eta, &F, &cos_theta_t, &eta_it, &eta_ti);
            int selected_r = sample_1 <= F;
            bs_pdf = mpdf * (selected_r ? F : 1.f - F);
            bs_eta = selected_r ? 1.f : eta_it;
            float dwh_dwo; v3 w;
            if (selected_r) {
                bs_wo = V(fmaf(m.x, 2.f * dwm, -wi.x), fmaf(m.y, 2.f * dwm, -wi.y), fmaf(m.z, 2.f * dwm, -wi.z));
                w = V(m_.spec_refl[0], m_.spec_refl[1], m_.spec_refl[2]);
                dwh_dwo = f_rcp(4.f * v_dot(bs_wo, m));
            } else {
                float k = fmaf(dwm, eta_ti, cos_theta_t);
                bs_wo = V(fmaf(m.x, k, -(wi.x * eta_ti)), fmaf(m.y, k, -(wi.y * eta_ti)), fmaf(m.z, k, -(wi.z * eta_ti)));
                float f2 = f_sqr(eta_ti);
                w = V(f2 * m_.spec_trans[0], f2 * m_.spec_trans[1], f2 * m_.spec_trans[2]);
                float dom = v_dot(bs_wo, m);
                dwh_dwo = (f_sqr(bs_eta) * dom) / f_sqr(dwm + bs_eta * dom);
            }
            /* :345-349: smith_g1(wo, m) with visible normals, else G(wi, wo, m) dot(wi, m) / (cos_theta_i cos_theta(m)) */
            float g1 = g.visible ? ggx_smith_g1(g, bs_wo, m)
                                 : ggx_smith_g1(g, wi, m) * ggx_smith_g1(g, bs_wo, m) * dwm / (wi.z * m.z);
            bs_pdf *= fabsf(dwh_dwo);
            if (mpdf != 0.f) bsdf_weight = v_mul(w, g1);
        }
    } else if (sh->bsdf == ORC_BSDF_ROUGHCONDUCTOR) {
        /* RoughConductor::eval / pdf / sample (roughconductor.cpp:229-415), GGX + visible normals, under TwoSidedBRDF */
        v3 wi = wi_in, wo_l = wo;
        if (sh->twosided && wi.z < 0.f) { wi.z = -wi.z; wo_l.z = -wo_l.z; }   /* twosided.cpp:219-258 flips both */
        ggx_t g = mf_make(sh->mf_type, m_.alpha_u, m_.alpha_v, !sh->sample_all);
        if (wi.z > 0.f && wo_l.z > 0.f) {
            v3 H = v_normalize(v_add(wo_l, wi));
            float D = ggx_eval(g, H);
            if (D != 0.f) {   /* eval :317-375 */
                float G = ggx_smith_g1(g, wi, H) * ggx_smith_g1(g, wo_l, H);
                float result = D * G / (4.f * wi.z), c = v_dot(wi, H);
                bsdf_val = V(fresnel_conductor(c, sh->cond_eta[0], sh->cond_k[0]) * (result * m_.spec_refl[0]),
                             fresnel_conductor(c, sh->cond_eta[1], sh->cond_k[1]) * (result * m_.spec_refl[1]),
                             fresnel_conductor(c, sh->cond_eta[2], sh->cond_k[2]) * (result * m_.spec_refl[2]));
            }
            if (v_dot(wi, H) > 0.f && v_dot(wo_l, H) > 0.f)   /* pdf :377-415 */
                bsdf_pdf = g.visible ? ggx_eval(g, H) * ggx_smith_g1(g, wi, H) / (4.f * wi.z) : ggx_pdf(g, wi, H) / (4.f * v_dot(wo_l, H));   /* :405-409 */
        }
        if (wi.z > 0.f) {   /* sample :229-315 */
            float mpdf;
            v3 m = ggx_sample(g, wi, s2x, s2y, &mpdf);
            float dwm = v_dot(wi, m);
            v3 r = V(fmaf(m.x, 2.f * dwm, -wi.x), fmaf(m.y, 2.f * dwm, -wi.y), fmaf(m.z, 2.f * dwm, -wi.z));   /* reflect(wi, m) fresnel.h:282-284 */
            bs_wo = r; bs_eta = 1.f;
            int ok = mpdf != 0.f && r.z > 0.f;
            float weight = g.visible ? ggx_smith_g1(g, r, m) : ggx_smith_g1(g, wi, m) * ggx_smith_g1(g, r, m) * dwm / (wi.z * m.z);   /* :260-265 */
            bs_pdf = mpdf / (4.f * v_dot(r, m));
            if (ok) bsdf_weight = V(fresnel_conductor(dwm, sh->cond_eta[0], sh->cond_k[0]) * (weight * m_.spec_refl[0]),
                                    fresnel_conductor(dwm, sh->cond_eta[1], sh->cond_k[1]) * (weight * m_.spec_refl[1]),
                                    fresnel_conductor(dwm, sh->cond_eta[2], sh->cond_k[2]) * (weight * m_.spec_refl[2]));
            if (sh->twosided && wi_in.z < 0.f) bs_wo.z = -bs_wo.z;
        }
    } else if (sh->bsdf == ORC_BSDF_ROUGHPLASTIC) {
        /* RoughPlastic::sample (roughplastic.cpp:259-331) under TwoSidedBRDF; eval / pdf in rough_plastic_eval_pdf */
        v3 wi = wi_in, wo_l = wo;
        if (sh->twosided && wi.z < 0.f) { wi.z = -wi.z; wo_l.z = -wo_l.z; }
        ggx_t g = mf_make(sh->mf_type, m_.alpha_u, m_.alpha_u, !sh->sample_all);
        if (wi.z > 0.f) {
            float t_i = lerp_gather64(sh->rough_table, wi.z);
            float prob_specular = (1.f - t_i) * sh->spec_sampling_weight, prob_diffuse = t_i * (1.f - sh->spec_sampling_weight);
            prob_specular = prob_specular / (prob_specular + prob_diffuse);
            prob_diffuse = 1.f - prob_specular;
            if (wo_l.z > 0.f) rough_plastic_eval_pdf(g, sh, refl, m_.spec_refl, wi, wo_l, t_i, prob_specular, prob_diffuse, &bsdf_val, &bsdf_pdf);
            if (sample_1 < prob_specular) {
                float mpdf; v3 m = ggx_sample(g, wi, s2x, s2y, &mpdf);
                float dwm = v_dot(wi, m);
                bs_wo = V(fmaf(m.x, 2.f * dwm, -wi.x), fmaf(m.y, 2.f * dwm, -wi.y), fmaf(m.z, 2.f * dwm, -wi.z));   /* reflect(wi, m) */
            } else bs_wo = square_to_cosine_hemisphere(s2x, s2y);
            bs_eta = 1.f;
            v3 value = V(0, 0, 0);
            if (bs_wo.z > 0.f) rough_plastic_eval_pdf(g, sh, refl, m_.spec_refl, wi, bs_wo, t_i, prob_specular, prob_diffuse, &value, &bs_pdf);
            if (bs_pdf > 0.f) bsdf_weight = v_mul(value, f_rcp(bs_pdf));   /* Spectrum / Float: times the reciprocal */
            if (sh->twosided && wi_in.z < 0.f) bs_wo.z = -bs_wo.z;
        }
    } else if (sh->bsdf == ORC_BSDF_PLASTIC) {
        /* SmoothPlastic::eval / pdf / sample (plastic.cpp:219-360) under TwoSidedBRDF (twosided.cpp:111-148,219-258) */
        float wiz = wi_in.z, woz = wo.z;
        if (sh->twosided) { woz = f_mulsign(woz, wiz); wiz = fabsf(wiz); }
        float f_i, tmp1, tmp2, tmp3;
        fresnel_dielectric(wiz, sh->diel_eta, &f_i, &tmp1, &tmp2, &tmp3);
        const float w = sh->spec_sampling_weight;
        v3 diff = V(refl[0], refl[1], refl[2]);
        diff = sh->nonlinear ? V(diff.x / (1.f - diff.x * sh->fdr_int), diff.y / (1.f - diff.y * sh->fdr_int), diff.z / (1.f - diff.z * sh->fdr_int))
                             : V(diff.x / (1.f - sh->fdr_int), diff.y / (1.f - sh->fdr_int), diff.z / (1.f - sh->fdr_int));
        if (wiz > 0.f && woz > 0.f) {   /* eval (:309-332) and pdf (:334-360) of the diffuse lobe */
            float f_o; fresnel_dielectric(woz, sh->diel_eta, &f_o, &tmp1, &tmp2, &tmp3);
            float k = ORC_INV_PI_F * woz * sh->inv_eta_2 * (1.f - f_i) * (1.f - f_o);
            bsdf_val = V(diff.x * k, diff.y * k, diff.z * k);
            float prob_specular = f_i * w, prob_diffuse = (1.f - f_i) * (1.f - w);
            prob_diffuse = prob_diffuse / (prob_specular + prob_diffuse);
            bsdf_pdf = ORC_INV_PI_F * woz * prob_diffuse;
        }
        if (wiz > 0.f) {                /* sample (:219-307) */
            float prob_specular = f_i * w, prob_diffuse = (1.f - f_i) * (1.f - w);
            prob_specular = prob_specular / (prob_specular + prob_diffuse);
            prob_diffuse = 1.f - prob_specular;
            bs_eta = 1.f;
            if (sample_1 < prob_specular) {
                bs_wo = V(-wi_in.x, -wi_in.y, wiz);   /* reflect() of the (possibly flipped) wi */
                bs_pdf = prob_specular; bs_delta = 1;
                float value = f_i / bs_pdf;
                bsdf_weight = V(value * m_.spec_refl[0], value * m_.spec_refl[1], value * m_.spec_refl[2]);
            } else {
                bs_wo = square_to_cosine_hemisphere(s2x, s2y);
                bs_pdf = prob_diffuse * (ORC_INV_PI_F * bs_wo.z);
                float f_o; fresnel_dielectric(bs_wo.z, sh->diel_eta, &f_o, &tmp1, &tmp2, &tmp3);
                float k = sh->inv_eta_2 * (1.f - f_i) * (1.f - f_o) / prob_diffuse;
                bsdf_weight = V(diff.x * k, diff.y * k, diff.z * k);
            }
            if (sh->twosided) bs_wo.z = f_mulsign(bs_wo.z, wi_in.z);
        }
    } else {
        float wiz = wi_in.z, woz = wo.z;
        if (sh->twosided) { woz = f_mulsign(woz, wiz); wiz = fabsf(wiz); }
        if (wiz > 0.f && woz > 0.f) {
            bsdf_val = V(refl[0] * ORC_INV_PI_F * woz, refl[1] * ORC_INV_PI_F * woz,
                         refl[2] * ORC_INV_PI_F * woz);
            bsdf_pdf = ORC_INV_PI_F * woz;
        }
        if (wiz > 0.f) {
            bs_wo = square_to_cosine_hemisphere(s2x, s2y);
            bs_pdf = ORC_INV_PI_F * bs_wo.z;
            bs_eta = 1.f;
            if (bs_pdf > 0.f) bsdf_weight = V(refl[0], refl[1], refl[2]);
            if (sh->twosided) bs_wo.z = f_mulsign(bs_wo.z, wi_in.z);
        }
    }
    out->val = bsdf_val; out->pdf = bsdf_pdf; out->weight = bsdf_weight; out->wo = bs_wo;
    out->bs_pdf = bs_pdf; out->bs_eta = bs_eta; out->bs_delta = bs_delta; out->bs_null = bs_null;
}
/* NormalMap (src/bsdfs/normalmap.cpp:110-189) around the plain BSDF, itself inside the two-sided adapter if the shape has one: TwoSidedBRDF flips wi.z and
 * wo.z on the back side first (twosided.cpp:111-148,219-258), NormalMap::frame builds n = normalize(2 c - 1) from the texture, s = normalize(dp_du - n (n . dp_du))
 * with the interaction's dp_du as it is, t = n x s; wi and wo go into that frame, the nested BSDF is evaluated / sampled there, the sampled direction comes back
 * through the frame and the adapter.  cos_theta(wo) * cos_theta(perturbed wo) <= 0 (a light leak): no value, no density, no weight. */
typedef struct { v3 dp_du, dp_dv, n, sh_s, sh_t, sh_n; } orc_geo;   /* what the frames of normalmap / bumpmap read of the interaction */
static void framed_bsdf_eval_pdf_sample(const orc_shape *sh, const orc_geo *g, v3 wi_in, v3 wo, int active_em, float sample_1, float s2x, float s2y, float uv_u, float uv_v, orc_bsdf_out *out) {
    if (!sh->tex_normal) { nested_bsdf_eval_pdf_sample(sh, wi_in, wo, active_em, sample_1, s2x, s2y, uv_u, uv_v, out); return; }
    const v3 dp_du = g->dp_du;
    v3 n;
    if (sh->bumpmap) {   /* BumpMap::frame (src/bsdfs/bumpmap.cpp:199-222): the surface displaced along its shading normal by the height texture, to first order */
        float gu, gv; orc_texture_eval_1_grad(sh->tex_normal, uv_u, uv_v, &gu, &gv);
        gu *= sh->bump_scale; gv *= sh->bump_scale;
        const v3 du = v_fma(g->sh_n, gu - v_dot(g->sh_n, g->dp_du), g->dp_du), dv = v_fma(g->sh_n, gv - v_dot(g->sh_n, g->dp_dv), g->dp_dv);
        v3 nw = v_normalize(v_cross(du, dv));
        if (v_dot(g->n, nw) < 0.f) nw = v_neg(nw);
        n = V(v_dot(nw, g->sh_s), v_dot(nw, g->sh_t), v_dot(nw, g->sh_n));        /* si.to_local(n) */
    } else {
        float c[3]; orc_texture_eval(sh->tex_normal, uv_u, uv_v, c);             /* m_normalmap->eval_3(si) */
        n = v_normalize(V(fmaf(c[0], 2.f, -1.f), fmaf(c[1], 2.f, -1.f), fmaf(c[2], 2.f, -1.f)));
    }
    const float k = v_dot(n, dp_du);
    const v3 s = v_normalize(V(fmaf(-n.x, k, dp_du.x), fmaf(-n.y, k, dp_du.y), fmaf(-n.z, k, dp_du.z)));
    const v3 t = v_cross(n, s);
    const int back = sh->twosided && wi_in.z < 0.f;
    v3 wi_f = wi_in, wo_f = wo;
    if (back) { wi_f.z = -wi_f.z; wo_f.z = -wo_f.z; }
    const v3 wi_p = V(v_dot(wi_f, s), v_dot(wi_f, t), v_dot(wi_f, n)), wo_p = V(v_dot(wo_f, s), v_dot(wo_f, t), v_dot(wo_f, n));
    orc_shape plain = *sh; plain.twosided = 0;
    nested_bsdf_eval_pdf_sample(&plain, wi_p, wo_p, active_em, sample_1, s2x, s2y, uv_u, uv_v, out);
    if (!(wo_f.z * wo_p.z > 0.f)) { out->val = V(0, 0, 0); out->pdf = 0.f; }
    if (out->weight.x != 0.f || out->weight.y != 0.f || out->weight.z != 0.f) {
        const v3 pw = v_fma(n, out->wo.z, v_fma(t, out->wo.y, v_mul(s, out->wo.x)));   /* perturbed_si.to_world(bs.wo) */
        if (!(out->wo.z * pw.z > 0.f)) out->weight = V(0, 0, 0);
        out->wo = pw;
    }
    if (back) out->wo.z = -out->wo.z;
}
/* BlendBSDF (src/bsdfs/blendbsdf.cpp:114-213): eval and pdf are the weighted sums of both nested BSDFs; sample1 <= weight samples bsdf_1 with sample1 / weight,
 * otherwise bsdf_0 with (sample1 - weight) / (1 - weight), and the nested sample goes back AS IT IS (its own weight and density, not the mixture's) */
static void blended_bsdf_eval_pdf_sample(const orc_shape *sh, const orc_geo *g, v3 wi_in, v3 wo, int active_em, float sample_1, float s2x, float s2y, float uv_u, float uv_v, orc_bsdf_out *out) {
    if (!sh->blend_other) { framed_bsdf_eval_pdf_sample(sh, g, wi_in, wo, active_em, sample_1, s2x, s2y, uv_u, uv_v, out); return; }
    if (sh->two_bsdfs) {   /* TwoSidedBRDF with two nested BSDFs (twosided.cpp:111-148,219-258): the front side's for wi.z > 0, the back side's (after the flip) for wi.z < 0 */
        framed_bsdf_eval_pdf_sample(wi_in.z < 0.f ? (const orc_shape *) sh->blend_other : sh, g, wi_in, wo, active_em, sample_1, s2x, s2y, uv_u, uv_v, out);
        return;
    }
    float w = sh->tex_blend ? orc_texture_eval_1(sh->tex_blend, uv_u, uv_v) : sh->blend_weight;
    w = f_min(f_max(w, 0.f), 1.f);                                      /* eval_weight (:213-215) */
    const int pick_1 = sample_1 <= w;
    orc_bsdf_out o0, o1;
    framed_bsdf_eval_pdf_sample(sh, g, wi_in, wo, active_em, (sample_1 - w) / (1.f - w), s2x, s2y, uv_u, uv_v, &o0);
    framed_bsdf_eval_pdf_sample((const orc_shape *) sh->blend_other, g, wi_in, wo, active_em, sample_1 / w, s2x, s2y, uv_u, uv_v, &o1);
    *out = pick_1 ? o1 : o0;
    const float w0 = 1.f - w;
    out->val = V(o0.val.x * w0 + o1.val.x * w, o0.val.y * w0 + o1.val.y * w, o0.val.z * w0 + o1.val.z * w);
    out->pdf = o0.pdf * w0 + o1.pdf * w;
}
/* The shape's BSDF, seen through its `mask` if it has one.  MaskBSDF::eval_pdf (src/bsdfs/mask.cpp:184-207): value and density of the nested BSDF times the
 * opacity; MaskBSDF::sample (:125-163): sample1 < opacity samples the nested BSDF with sample1 / opacity (its sample and weight are passed on unchanged),
 * otherwise the null interaction: wo = -wi, eta 1, pdf 1 - opacity, weight 1 (BSDFFlags::Null is a delta type) */
static void bsdf_eval_pdf_sample(const orc_shape *sh, const orc_geo *g, v3 wi_in, v3 wo, int active_em, float sample_1, float s2x, float s2y, float uv_u, float uv_v, orc_bsdf_out *out) {
    if (!sh->masked) { blended_bsdf_eval_pdf_sample(sh, g, wi_in, wo, active_em, sample_1, s2x, s2y, uv_u, uv_v, out); return; }
    float opacity = sh->tex_opacity ? orc_texture_eval_1(sh->tex_opacity, uv_u, uv_v) : sh->opacity;
    opacity = f_min(f_max(opacity, 0.f), 1.f);                          /* eval_opacity (:219-221) */
    const int nested_pick = sample_1 < opacity;
    blended_bsdf_eval_pdf_sample(sh, g, wi_in, wo, active_em, sample_1 / opacity, s2x, s2y, uv_u, uv_v, out);
    out->val = v_mul(out->val, opacity); out->pdf *= opacity;
    if (!nested_pick) { out->wo = V(-wi_in.x, -wi_in.y, -wi_in.z); out->bs_eta = 1.f; out->bs_pdf = 1.f - opacity; out->bs_delta = 1; out->bs_null = 1; out->weight = V(1.f, 1.f, 1.f); }
}
static inline int bsdf_is_smooth(int32_t k) { return k == ORC_BSDF_DIFFUSE || k == ORC_BSDF_PLASTIC || k == ORC_BSDF_ROUGHCONDUCTOR || k == ORC_BSDF_ROUGHPLASTIC || k == ORC_BSDF_ROUGHDIELECTRIC; }
/* mis_weight -- dopplertofpath.cpp:296-301 */
static inline float mis_weight(float a, float b) { a *= a; b *= b; float w = a / (a + b); return isfinite(w) ? w : 0.f; }

/* spp = Sampler::sample_count (all passes), spw = samples per wavefront = spp_per_pass, n_passes = spp / spw (integrator.cpp:121-135,227-245) */
typedef struct { const orc_scene *sc; const orc_params *p; uint32_t seed, spp, spw, n_passes; float s2c[16]; } orc_ctx;

/* One lane: SamplingIntegrator::render (lane->pixel, src/render/integrator.cpp:273-290),
 * render_sample Doppler branch (:476-542), DopplerToFPathIntegrator::sample
 * (src/integrators/dopplertofpath.cpp:79-283). */
/* Rectangle::eval_parameterization (rectangle.cpp:173-192): the point of the rectangle at (u, v), found by a ray from one normal length above it straight down --
 * through the rectangle's own intersection routine and surface interaction, whose roundings the point and its uv then carry */
static int rect_eval_parameterization(const orc_shape *sh, float u, float v, orc_si *si) {
    const v3 p = m_point(sh->to_world, V(u * 2.f - 1.f, v * 2.f - 1.f, 0.f));
    const v3 n = v_normalize(m_normal(sh->to_object, V(0.f, 0.f, 1.f)));
    const v3 o = v_add(p, n), d = v_neg(n);
    float t, b1, b2;
    if (!rect_intersect(sh, o, d, ORC_LARGEST, &t, &b1, &b2)) return 0;
    rect_si(sh, o, d, t, si);
    si->uv_u = fmaf(b1, .5f, .5f); si->uv_v = fmaf(b2, .5f, .5f);   /* rectangle.cpp:312-313 */
    return 1;
}
/* Emitter::sample_direction of every supported emitter (point.cpp:118-147, constant.cpp:118-148, directional.cpp:148-176, envmap.cpp:363-406,
 * spot.cpp:152-187, area.cpp:116-159 -> Shape / Sphere::sample_direction), for the reference point `ref` and the 2-D sample (sx, e2):
 * sampled point, direction, distance, density, delta flag, importance weight, and whether the sample is usable.  Visibility is the caller's. */
static void emitter_sample_direction(const orc_scene *sc, const orc_emitter *em, v3 ref, float sx, float e2,
                                     v3 *dsp_out, v3 *dd_out, float *dist_out, float *pdf_out, int *delta_out, v3 *weight_out, int *active_out) {
    struct { v3 p; } si; si.p = ref;
    v3 dsp, dd, em_weight = V(0, 0, 0); int em_active = 1, ds_delta = 1; float ds_dist = 0.f, ds_pdf = 0.f;
    if (em->kind == ORC_EMITTER_POINT) {
        /* PointLight::sample_direction src/emitters/point.cpp:118-147 */
        dsp = V(em->position[0], em->position[1], em->position[2]);
        dd = v_sub(dsp, si.p);
        float dist2 = v_dot(dd, dd), inv_dist = f_rsqrt(dist2);
        ds_dist = sqrtf(dist2);
        dd = v_mul(dd, inv_dist);
        float id2 = f_sqr(inv_dist);
        em_weight = V(em->intensity[0] * id2, em->intensity[1] * id2, em->intensity[2] * id2);
        ds_pdf = 1.f; ds_delta = 1;
    } else if (em->kind == ORC_EMITTER_CONSTANT) {
        /* ConstantBackgroundEmitter::sample_direction (constant.cpp:118-148): a uniform direction; the sample point lies on a sphere
         * of twice the (enlarged) bounding radius around the reference point */
        dd = square_to_uniform_sphere(sx, e2);
        v3 c = V(em->bsphere[0], em->bsphere[1], em->bsphere[2]);
        float radius = f_max(em->bsphere[3], v_norm(v_sub(si.p, c)));
        ds_dist = 2.f * radius;
        dsp = v_fma(dd, ds_dist, si.p);
        ds_pdf = ORC_INV_FOUR_PI_F; ds_delta = 0;
        float ip = f_rcp(ds_pdf);
        em_weight = V(em->intensity[0] * ip, em->intensity[1] * ip, em->intensity[2] * ip);
    } else if (em->kind == ORC_EMITTER_DIRECTIONAL) {
        /* DirectionalEmitter::sample_direction (directional.cpp:148-176): a delta direction; the sample point lies outside the scene's bounding sphere */
        v3 dir = V(em->position[0], em->position[1], em->position[2]);
        v3 c = V(em->bsphere[0], em->bsphere[1], em->bsphere[2]);
        float radius = f_max(em->bsphere[3], v_norm(v_sub(si.p, c)));
        ds_dist = 2.f * radius;
        dsp = v_sub(si.p, v_mul(dir, ds_dist));
        dd = v_neg(dir);
        ds_pdf = 1.f; ds_delta = 1;
        em_weight = V(em->intensity[0], em->intensity[1], em->intensity[2]);
    } else if (em->kind == ORC_EMITTER_ENVMAP) {
        /* EnvironmentMapEmitter::sample_direction (envmap.cpp:363-406) */
        env_sample_direction(em, si.p, sx, e2, &dd, &ds_dist, &ds_pdf, &em_weight, &em_active);
        dsp = v_add(si.p, v_mul(dd, ds_dist));
        ds_delta = 0;
    } else if (em->kind == ORC_EMITTER_SPOT) {
        /* SpotLight::sample_direction (src/emitters/spot.cpp:152-187) with falloff_curve (:116-126) */
        dsp = V(em->position[0], em->position[1], em->position[2]);
        dd = v_sub(dsp, si.p);
        ds_dist = sqrtf(v_dot(dd, dd));
        float inv_dist = f_rcp(ds_dist);
        dd = v_mul(dd, inv_dist);
        v3 local = v_normalize(m_vector(em->to_local, v_neg(dd)));
        float cos_theta = local.z;
        float beam = cos_theta >= em->cos_beam ? 1.f : (em->cutoff_angle - orc_acos(cos_theta)) * em->inv_transition;
        float falloff = cos_theta > em->cos_cutoff ? beam : 0.f;
        float k = falloff * f_sqr(inv_dist);
        em_weight = falloff > 0.f ? V(em->intensity[0] * k, em->intensity[1] * k, em->intensity[2] * k) : V(0, 0, 0);
        ds_pdf = 1.f; ds_delta = 1;
    } else {
        /* AreaLight::sample_direction area.cpp:116-159 -> Shape::sample_direction shape.cpp:370-384 ->
         * Rectangle::sample_position rectangle.cpp:152-166 */
        const orc_shape *es = &sc->shapes[em->shape];
        v3 en;
        if (es->tex_radiance) {
            /* AreaLight::sample_direction with a spatially varying radiance (area.cpp:129-153): the TEXTURE is sampled (Texture::sample_position), the shape maps the uv
             * to a point (Rectangle::eval_parameterization), the density goes from uv space to solid angle with |dp_du x dp_dv| */
            float u, v, pdf; texture_sample_position(es->tex_radiance, sx, e2, &u, &v, &pdf);
            orc_si ps; const int valid = pdf != 0.f && rect_eval_parameterization(es, u, v, &ps);
            dsp = valid ? ps.p : si.p; en = valid ? ps.n : V(0.f, 0.f, 1.f);
            dd = v_sub(dsp, si.p);
            const float dist2 = v_dot(dd, dd);
            ds_dist = sqrtf(dist2);
            dd = v_mul(dd, f_rcp(ds_dist));
            const float dp = v_dot(dd, en);
            em_active = valid && dp < 0.f;
            ds_pdf = em_active ? pdf / v_norm(v_cross(ps.dp_du, ps.dp_dv)) * dist2 / -dp : 0.f;
            ds_delta = 0;
            float c[3] = { 0.f, 0.f, 0.f };
            if (em_active) orc_texture_eval(es->tex_radiance, ps.uv_u, ps.uv_v, c);      /* m_radiance->eval(si) / ds.pdf */
            em_weight = em_active ? V(c[0] / ds_pdf, c[1] / ds_pdf, c[2] / ds_pdf) : V(0, 0, 0);
            *dsp_out = dsp; *dd_out = dd; *dist_out = ds_dist; *pdf_out = ds_pdf; *delta_out = ds_delta; *weight_out = em_weight; *active_out = em_active;
            return;
        }
        if (es->kind == ORC_SHAPE_SPHERE) {   /* Sphere overrides Shape::sample_direction */
            sphere_sample_direction(es, si.p, sx, e2, &dsp, &en, &dd, &ds_dist, &ds_pdf);
        } else {
        if (es->kind == ORC_SHAPE_RECT) {
            dsp = m_point(es->to_world, V(sx * 2.f - 1.f, e2 * 2.f - 1.f, 0.f));
            en = v_normalize(m_normal(es->to_object, V(0.f, 0.f, 1.f)));
        } else if (es->kind == ORC_SHAPE_DISK) {   /* Disk::sample_position (disk.cpp:158-177) */
            v3 pd = square_to_cosine_hemisphere(sx, e2);   /* its x, y ARE square_to_uniform_disk_concentric */
            dsp = m_point(es->to_world, V(pd.x, pd.y, 0.f));
            en = v_normalize(m_normal(es->to_object, V(0.f, 0.f, 1.f)));
        } else mesh_sample_position(es, sx, e2, &dsp, &en);
        dd = v_sub(dsp, si.p);
        float dist2 = v_dot(dd, dd);
        ds_dist = sqrtf(dist2);
        dd = v_mul(dd, f_rcp(ds_dist));
        float dp = fabsf(v_dot(dd, en)), x = dist2 / dp;
        ds_pdf = shape_inv_area(es) * (isfinite(x) ? x : 0.f);
        }
        ds_delta = 0;
        em_active = v_dot(dd, en) < 0.f && ds_pdf != 0.f;
        float ip = f_rcp(ds_pdf);
        em_weight = em_active ? V(em->intensity[0] * ip, em->intensity[1] * ip, em->intensity[2] * ip) : V(0, 0, 0);
    }
    *dsp_out = dsp; *dd_out = dd; *dist_out = ds_dist; *pdf_out = ds_pdf; *delta_out = ds_delta; *weight_out = em_weight; *active_out = em_active;
}
/* known-answer entry (tests): out = d[3], dist, pdf, delta, weight[3], p[3], active */
void orc_kat_emitter_sample(const orc_scene *sc, int emitter_index, const float *ref, float sx, float sy, float *out13) {
    v3 dsp, dd, w; float dist, pdf; int delta, active;
    emitter_sample_direction(sc, &sc->emitters[emitter_index], V(ref[0], ref[1], ref[2]), sx, sy, &dsp, &dd, &dist, &pdf, &delta, &w, &active);
    out13[0] = dd.x; out13[1] = dd.y; out13[2] = dd.z; out13[3] = dist; out13[4] = pdf; out13[5] = (float) delta;
    out13[6] = w.x; out13[7] = w.y; out13[8] = w.z; out13[9] = dsp.x; out13[10] = dsp.y; out13[11] = dsp.z; out13[12] = (float) active;
}

static void eval_lane(const orc_ctx *cx, uint64_t lane64, orc_lane *out) {
    const orc_scene *sc = cx->sc; const orc_params *p = cx->p; const orc_sensor *se = &sc->sensor;
    /* lane64 = pass * wavefront_size + lane: the sampler is seeded once per wavefront lane (integrator.cpp:265) and its three RNG
     * streams run on through the passes (Sampler::advance only resets the dimension index and bumps the sample index,
     * sampler.cpp:52-55), so pass k of a lane is evaluated after its passes 0 .. k-1 */
    const uint32_t spp = cx->spp, spw = cx->spw;
    const uint64_t wavefront = (uint64_t) se->crop_w * (uint64_t) se->crop_h * spw;
    const uint32_t lane = (uint32_t) (lane64 % wavefront), pass_target = (uint32_t) (lane64 / wavefront);
    orc_sampler smp; sampler_seed(&smp, p, cx->seed, spw, lane);
    for (uint32_t pass = 0; pass <= pass_target; ++pass) {
    smp.dim = 0;
    smp.sample_index = pass * spw + (spw > 1 ? lane % spw : 0);   /* current_sample_index, sampler.cpp:94-103 */

    uint32_t pix = lane / spw, W = (uint32_t) se->crop_w;
    uint32_t py = pix / W, px = pix - W * py;
    float posx = (float) (px + (uint32_t) se->crop_x), posy = (float) (py + (uint32_t) se->crop_y);

    int correlate_pixel = p->path_correlation_depth > 0;
    const int plain = p->integrator != 0;   /* m_is_doppler_integrator == false: plain branch of render_sample */
    /* one stream only: the plain branch, and every sampler but `correlated` -- Sampler::next_1d_correlate / next_2d_correlate
     * default to next_1d / next_2d (include/mitsuba/render/sampler.h:141-144) */
    const int single = plain || p->sampler != 0;
    float jx = sampler_draw(&smp, correlate_pixel, single);
    float jy = sampler_draw(&smp, correlate_pixel, single);
    float spx = posx + jx, spy = posy + jy;
    float scx = 1.f / (float) se->crop_w, scy = 1.f / (float) se->crop_h;
    float ax = fmaf(spx, scx, -(float) se->crop_x * scx), ay = fmaf(spy, scy, -(float) se->crop_y * scy);

    /* needs_aperture_sample() (m_needs_sample_3, thinlens.cpp:155): a second 2-D draw of the same kind (integrator.cpp:421-423,490-492) */
    float apx = .5f, apy = .5f;
    if (se->kind == ORC_SENSOR_THINLENS) { apx = sampler_draw(&smp, correlate_pixel, single); apy = sampler_draw(&smp, correlate_pixel, single); }

    float time = se->shutter_open;
    float shutter_open_time = se->shutter_close - se->shutter_open;
    if (shutter_open_time > 0.f) {
        float u;
        if (plain || p->sampler == 1) u = sampler_next_1d(&smp);          /* Sampler::next_1d_time -> next_1d (sampler.h:131-132) */
        else if (p->sampler == 0) u = sampler_next_1d_time(&smp, p, spp);
        else {   /* TimeStratifiedSampler::next_1d_time (timestratified.cpp:117-129); m_inv_sample_count = rcp(float(spp)) (:78-82) */
            uint32_t q = orc_permute_kensler(smp.sample_index, spp, smp.perm_seed + smp.dim++);
            float j = p->jitter ? sampler_next_1d(&smp) : .5f;
            u = ((float) q + j) * (1.0f / (float) spp);
        }
        time += u * shutter_open_time;
    }

    orc_ray ray = camera_ray(se, cx->s2c, ax, ay, apx, apy);
    /* dopplertofpath.cpp:93 */
    if (!plain) time = time < p->time ? time : time - p->time;

    out->sample_pos[0] = spx; out->sample_pos[1] = spy; out->time = time;
    out->ray_o[0] = ray.o.x; out->ray_o[1] = ray.o.y; out->ray_o[2] = ray.o.z;
    out->ray_d[0] = ray.d.x; out->ray_d[1] = ray.d.y; out->ray_d[2] = ray.d.z;

    v3 thr = V(1.f, 1.f, 1.f), res = V(0.f, 0.f, 0.f);
    float path_length = 0.f, eta = 1.f;
    /* the environment emitter, if any (scene.cpp:53-57); valid_ray starts as !m_hide_emitters && environment != nullptr (dopplertofpath.cpp:101-102) */
    const orc_emitter *env = NULL;
    for (int32_t ei = 0; ei < sc->n_emitters; ++ei) if (sc->emitters[ei].kind == ORC_EMITTER_CONSTANT || sc->emitters[ei].kind == ORC_EMITTER_ENVMAP) env = &sc->emitters[ei];
    uint32_t depth = 0; int valid_ray = env && !p->hide_emitters, active = p->max_depth != 0;
    if (p->max_depth == 0) valid_ray = 0;   /* :87-88: return { 0.f, false } */
    v3 o = ray.o, d = ray.d; float maxt = ray.maxt;
    v3 prev_p = V(0, 0, 0); float prev_bsdf_pdf = 1.f; int prev_delta = 1;   /* dopplertofpath.cpp:106-108 */

    if (p->integrator == 2) {   /* VelocityIntegrator::sample, src/integrators/velocity.cpp:125-142 */
        orc_hit h1 = scene_closest(sc, o, d, 0.f, maxt), h2 = scene_closest(sc, o, d, p->time, maxt);
        int v1 = h1.obj >= 0, v2 = h2.obj >= 0;
        float vel = ((v2 ? h2.t : 0.f) - (v1 ? h1.t : 0.f)) * (1.0f / p->time);
        vel = (v1 && v2) ? vel : 0.f;
        out->rgb[0] = out->rgb[1] = out->rgb[2] = vel;
        out->path_length = 0.f; out->depth = 0; out->valid = (uint32_t) (v1 && v2);
        continue;
    }

    while (active) {
        int correlate = (depth + 1) < p->path_correlation_depth;
        orc_hit h = scene_closest(sc, o, d, time, maxt);
        int hit = h.obj >= 0;
        if (hit) path_length += h.t * eta;
        int active_next = (depth + 1 < p->max_depth) && hit;

        orc_si si; memset(&si, 0, sizeof si);
        v3 em_weight = V(0, 0, 0), wo = V(0, 0, 0); float ds_pdf = 0.f, ds_dist = 0.f; int ds_delta = 0;
        if (hit) compute_si(sc, &h, o, d, time, &si);
        const float pmf = sc->n_emitters ? 1.f / (float) sc->n_emitters : 0.f;   /* m_emitter_pmf, scene.cpp:96 */

        /* ---- direct emission (dopplertofpath.cpp:150-168 / path.cpp): the hit shape carries an area emitter */
        if (!hit && env) {
            /* si.emitter(scene) of a missed ray is the environment (interaction.h); DirectionSample(scene, si, prev_si): d = -si.wi = the ray
             * direction; Scene::pdf_emitter_direction -> ConstantBackgroundEmitter::pdf_direction = square_to_uniform_sphere_pdf (constant.cpp:150-155) */
            const int is_map = env->kind == ORC_EMITTER_ENVMAP;   /* EnvironmentMapEmitter::pdf_direction / eval (envmap.cpp:408-425,299-310) with ds.d = -si.wi = d */
            float em_pdf = prev_delta ? 0.f : (is_map ? env_pdf_direction(env, d) : ORC_INV_FOUR_PI_F) * pmf;
            float mis_bsdf = mis_weight(prev_bsdf_pdf, em_pdf);
            v3 le = prev_bsdf_pdf > 0.f ? (is_map ? env_eval(env, d) : V(env->intensity[0], env->intensity[1], env->intensity[2])) : V(0, 0, 0);
            v3 v = v_mul(le, mis_bsdf);
            if (!plain) v = v_mul(v, orc_modulation_weight(p, time, path_length));
            res = V(fmaf(thr.x, v.x, res.x), fmaf(thr.y, v.y, res.y), fmaf(thr.z, v.z, res.z));
        }
        if (hit && si.shape->emitter) {
            /* DirectionSample(scene, si, prev_si) -- include/mitsuba/render/records.h:173-180 */
            v3 rel = v_sub(si.p, prev_p);
            float dist = v_norm(rel);
            v3 dsd = v_mul(rel, f_rcp(dist));
            float em_pdf = 0.f;
            if (!prev_delta) {   /* Scene::pdf_emitter_direction scene.cpp:293-299 -> AreaLight::pdf_direction area.cpp:161-180 */
                float dp = v_dot(dsd, si.sh_n);   /* ds.n = si.sh_frame.n: PositionSample(si), records.h:63-65 */
                if (dp < 0.f && si.shape->tex_radiance) {   /* area.cpp:170-176: pdf_position of the texture at ds.uv = si.uv, through the parameterisation's |dp_du x dp_dv| */
                    orc_si ps;
                    if (rect_eval_parameterization(si.shape, si.uv_u, si.uv_v, &ps))
                        em_pdf = texture_pdf_position(si.shape->tex_radiance, si.uv_u, si.uv_v) * f_sqr(dist) / (v_norm(v_cross(ps.dp_du, ps.dp_dv)) * -dp) * pmf;
                } else
                if (dp < 0.f) {   /* Shape::pdf_direction shape.cpp:386-396; pdf_position = 1/area (rectangle.cpp:168-171, mesh.cpp:570-573) */
                    float adp = fabsf(dp);
                    float pdf = si.shape->kind == ORC_SHAPE_SPHERE ? sphere_pdf_direction(si.shape, prev_p, dsd, si.sh_n, dist)
                              : shape_inv_area(si.shape) * (adp != 0.f ? (dist * dist) / adp : 0.f);
                    em_pdf = pdf * pmf;
                }
            }
            float mis_bsdf = mis_weight(prev_bsdf_pdf, em_pdf);
            /* AreaLight::eval area.cpp:82-89, masked by prev_bsdf_pdf > 0 */
            int on = si.wi.z > 0.f && prev_bsdf_pdf > 0.f;
            v3 le = on ? V(si.shape->radiance[0], si.shape->radiance[1], si.shape->radiance[2]) : V(0, 0, 0);
            if (on && si.shape->tex_radiance) { float c[3]; orc_texture_eval(si.shape->tex_radiance, si.uv_u, si.uv_v, c); le = V(c[0], c[1], c[2]); }   /* m_radiance->eval(si) */
            v3 v = v_mul(le, mis_bsdf);
            if (!plain) v = v_mul(v, orc_modulation_weight(p, time, path_length));
            res = V(fmaf(thr.x, v.x, res.x), fmaf(thr.y, v.y, res.y), fmaf(thr.z, v.z, res.z));
        }

        /* has_flag(bsdf->flags(), BSDFFlags::Smooth), :178 -- diffuse and plastic have a smooth lobe */
        int active_em = active_next && hit && (bsdf_is_smooth(si.shape->bsdf) ||
                                               (si.shape->blend_other && bsdf_is_smooth(((const orc_shape *) si.shape->blend_other)->bsdf)));   /* a blend has the flags of both */

        /* emitter sampling: Scene::sample_emitter_direction src/render/scene.cpp:235-291 */
        float e1 = sampler_draw(&smp, correlate, single);
        float e2 = sampler_draw(&smp, correlate, single);
        if (active_em && sc->n_emitters > 0) {
            uint32_t ne = (uint32_t) sc->n_emitters, idx = 0; float em_w = 1.f, sx = e1;
            if (ne > 1) {   /* sample_emitter :171-189 */
                float scaled = e1 * (float) ne;
                idx = (uint32_t) scaled; if (idx > ne - 1) idx = ne - 1;
                em_w = (float) ne; sx = scaled - (float) idx;
            }
            const orc_emitter *em = &sc->emitters[idx];
            v3 dsp, dd; int em_active = 1;
            emitter_sample_direction(sc, em, si.p, sx, e2, &dsp, &dd, &ds_dist, &ds_pdf, &ds_delta, &em_weight, &em_active);
            ds_pdf *= pmf; em_weight = v_mul(em_weight, em_w);
            if (ds_pdf != 0.f && em_active) {
                /* Interaction::spawn_ray_to interaction.h:141-149 + ray_test */
                v3 so = offset_p(&si, v_sub(dsp, si.p));
                v3 sd = v_sub(dsp, so);
                float sdist = v_norm(sd);
                sd = v_mul(sd, f_rcp(sdist));
                if (scene_occluded(sc, so, sd, time, sdist * (1.f - ORC_SHADOW_EPS))) { em_weight = V(0, 0, 0); ds_pdf = 0.f; }
            }
            active_em = active_em && ds_pdf != 0.f;
            wo = si_to_local(&si, dd);
        } else {
            active_em = 0;
        }

        float sample_1 = sampler_draw(&smp, correlate, single);
        float s2x = sampler_draw(&smp, correlate, single);
        float s2y = sampler_draw(&smp, correlate, single);

        v3 bsdf_val = V(0, 0, 0), bsdf_weight = V(0, 0, 0), bs_wo = V(0, 0, 0);
        float bsdf_pdf = 0.f, bs_pdf = 0.f, bs_eta = 0.f; int bs_delta = 0, bs_null = 0;
        if (hit) {
            orc_bsdf_out bo;
            const orc_geo geo = { si.dp_du, si.dp_dv, si.n, si.sh_s, si.sh_t, si.sh_n };
            bsdf_eval_pdf_sample(si.shape, &geo, si.wi, wo, active_em, sample_1, s2x, s2y, si.uv_u, si.uv_v, &bo);
            bsdf_val = bo.val; bsdf_pdf = bo.pdf; bsdf_weight = bo.weight; bs_wo = bo.wo; bs_pdf = bo.bs_pdf; bs_eta = bo.bs_eta; bs_delta = bo.bs_delta; bs_null = bo.bs_null;
        }
        if (active_em) {   /* :214-226 */
            float mis_em = ds_delta ? 1.f : mis_weight(ds_pdf, bsdf_pdf);
            v3 v = V(bsdf_val.x * em_weight.x * mis_em, bsdf_val.y * em_weight.y * mis_em, bsdf_val.z * em_weight.z * mis_em);
            if (!plain) v = v_mul(v, orc_modulation_weight(p, time, path_length + ds_dist));   /* path.cpp has no length weight */
            res = V(fmaf(thr.x, v.x, res.x), fmaf(thr.y, v.y, res.y), fmaf(thr.z, v.z, res.z));
        }
        /* :232-262 */
        if (hit) {
            v3 nd = si_to_world(&si, bs_wo);
            o = offset_p(&si, nd); d = nd; maxt = ORC_LARGEST;
        }
        thr = V(thr.x * bsdf_weight.x, thr.y * bsdf_weight.y, thr.z * bsdf_weight.z);
        eta *= bs_eta;
        valid_ray |= hit && !bs_null;   /* :252-253: active && si.is_valid() && !has_flag(bsdf_sample.sampled_type, BSDFFlags::Null) */
        prev_p = si.p; prev_bsdf_pdf = bs_pdf; prev_delta = bs_delta;   /* :256-258 */
        if (hit) depth += 1;
        /* :264-276 */
        float thr_max = f_max(f_max(thr.x, thr.y), thr.z);
        float rr_prob = f_min(thr_max * f_sqr(eta), .95f);
        int rr_active = depth >= p->rr_depth;
        int rr_continue = sampler_draw(&smp, correlate, single) < rr_prob;
        if (rr_active) { float ir = f_rcp(rr_prob); thr = v_mul(thr, ir); }
        active = active_next && (!rr_active || rr_continue) && thr_max != 0.f;
    }
    out->rgb[0] = valid_ray ? res.x : 0.f; out->rgb[1] = valid_ray ? res.y : 0.f; out->rgb[2] = valid_ray ? res.z : 0.f;
    out->path_length = path_length; out->depth = depth; out->valid = (uint32_t) valid_ray;
    }   /* passes */
}

void orc_sampler_lane(const orc_params *p, uint32_t seed, uint32_t spp, uint32_t lane, uint32_t *ou, float *of) {
    orc_sampler s; sampler_seed(&s, p, seed, spp, lane);
    ou[0] = (uint32_t) s.s_main; ou[1] = (uint32_t) (s.s_main >> 32);
    ou[2] = (uint32_t) s.s_time; ou[3] = (uint32_t) (s.s_time >> 32);
    ou[4] = (uint32_t) s.s_path; ou[5] = (uint32_t) (s.s_path >> 32);
    ou[6] = s.perm_seed;
    int cp = p->path_correlation_depth > 0;
    of[0] = sampler_next_1d_correlate(&s, cp);
    of[1] = sampler_next_1d_correlate(&s, cp);
    of[2] = sampler_next_1d_time(&s, p, spp);
}

/* ------------------------------------------------------------------ threading */
typedef struct { const orc_ctx *cx; uint64_t begin, n; orc_lane *out; int tid, nt; } orc_job;
static void *lane_worker(void *arg) {
    orc_job *j = (orc_job *) arg;
    /* interleaved blocks of 256 lanes for load balance */
    uint64_t nblk = (j->n + 255) / 256;
    for (uint64_t b = (uint64_t) j->tid; b < nblk; b += (uint64_t) j->nt) {
        uint64_t s = b * 256, e = s + 256 < j->n ? s + 256 : j->n;
        for (uint64_t i = s; i < e; ++i) eval_lane(j->cx, j->begin + i, &j->out[i]);
    }
    return NULL;
}
static void run_lanes(const orc_ctx *cx, uint64_t begin, uint64_t n, orc_lane *out, int nt) {
    if (nt < 1) nt = 1;
    if (nt > 256) nt = 256;
    pthread_t th[256]; orc_job jobs[256];
    for (int t = 0; t < nt; ++t) {
        jobs[t].cx = cx; jobs[t].begin = begin; jobs[t].n = n; jobs[t].out = out; jobs[t].tid = t; jobs[t].nt = nt;
        if (nt == 1) lane_worker(&jobs[t]); else pthread_create(&th[t], NULL, lane_worker, &jobs[t]);
    }
    if (nt > 1) for (int t = 0; t < nt; ++t) pthread_join(th[t], NULL);
}
/* SamplingIntegrator::render (integrator.cpp:121-135,227-245): spp_per_pass = min(samples_per_pass, spp), which must divide spp;
 * a wavefront of more than 2^32 - 1 lanes is split further (integer division, as written there); Sampler::set_samples_per_wavefront
 * (sampler.cpp:75-83) then insists that sample_count is a multiple of it.  Returns 0, or -1 where the reference throws. */
int orc_pass_layout(int32_t crop_w, int32_t crop_h, uint32_t spp, uint32_t samples_per_pass, uint32_t *spw_out, uint32_t *n_passes_out) {
    if (spp == 0) return -1;
    uint32_t spp_per_pass = samples_per_pass == 0xffffffffu || samples_per_pass == 0 ? spp : (samples_per_pass < spp ? samples_per_pass : spp);
    if (spp % spp_per_pass != 0) return -1;
    uint64_t wavefront = (uint64_t) crop_w * (uint64_t) crop_h * spp_per_pass, limit = 0xffffffffull;
    if (wavefront > limit) {
        spp_per_pass /= (uint32_t) ((wavefront + limit - 1) / limit);
        if (spp_per_pass == 0 || spp % spp_per_pass != 0) return -1;
    }
    *spw_out = spp_per_pass; *n_passes_out = spp / spp_per_pass;
    return 0;
}
static void make_ctx(orc_ctx *cx, const orc_scene *sc, const orc_params *p, uint32_t seed, uint32_t spp) {
    cx->sc = sc; cx->p = p; cx->seed = seed; cx->spp = spp; cx->spw = spp; cx->n_passes = 1;
    orc_pass_layout(sc->sensor.crop_w, sc->sensor.crop_h, spp, p->samples_per_pass, &cx->spw, &cx->n_passes);
    camera_sample_to_camera(&sc->sensor, cx->s2c);
}
void orc_render_lanes(const orc_scene *sc, const orc_params *p, uint32_t seed, uint32_t spp,
                      uint64_t lane_begin, uint64_t n, orc_lane *out, int n_threads) {
    orc_ctx cx; make_ctx(&cx, sc, p, seed, spp);
    run_lanes(&cx, lane_begin, n, out, n_threads);
}

/* ImageBlock::put, coalesced path -- src/render/imageblock.cpp:414-531 (2.1/2.2), with the
 * box-filter special case :201-224; TentFilter::eval src/rfilters/tent.cpp:53-55.
 * aovs = [R,G,B,1] (integrator.cpp:528-541).  Accumulation is sequential in lane order
 * (the reference's atomic scatter order is unspecified). */
/* dr::detail::estrin_impl, 10 coefficients (drjit/math.h; Estrin pairing) */
static float estrin10(float x, const float *c) {
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    float a0 = fmaf(x, c[1], c[0]), a1 = fmaf(x, c[3], c[2]), a2 = fmaf(x, c[5], c[4]), a3 = fmaf(x, c[7], c[6]), a4 = fmaf(x, c[9], c[8]);
    float b0 = fmaf(x2, a1, a0), b1 = fmaf(x2, a3, a2);
    return fmaf(x8, a4, fmaf(x4, b1, b0));
}
/* GaussianFilter ctor + eval, non-CUDA branch -- src/rfilters/gaussian.cpp:48-96 */
static void gaussian_coeffs(float stddev, float radius, float *out) {
    static const double coeff[10] = { 9.992604880e-1, -4.977025247e-1, 1.222248550e-1, -1.932406282e-2, 2.136713061e-3,
                                      -1.679873860e-4, 9.202145248e-6, -3.329417433e-7, 7.128382794e-9, -6.821193280e-11 };
    double scale = 1;
    for (int i = 0; i < 10; ++i) { out[i] = (float) (coeff[i] * scale); scale /= (double) stddev * (double) stddev; }
    out[0] -= estrin10(radius * radius, out);
}
/* MitchellNetravaliFilter::eval (src/rfilters/mitchell.cpp:47-67; coefficients in ScalarFloat, Horner with fmadd) and
 * CatmullRomFilter::eval (src/rfilters/catmullrom.cpp:38-53; B = 0, C = 1/2 as float32 arrays, plain multiplies and adds) */
static float mitchell_eval(float x, float B, float C) {
    x = fabsf(x);
    float x2 = x * x, x3 = x2 * x;
    float a3 = (12.f - 9.f * B - 6.f * C), a2 = (-18.f + 12.f * B + 6.f * C), a0 = (6.f - 2.f * B),
          b3 = (-B - 6.f * C), b2 = (6.f * B + 30.f * C), b1 = (-12.f * B - 48.f * C), b0 = (8.f * B + 24.f * C);
    float r = (1.f / 6.f) * (x < 1.f ? fmaf(a3, x3, fmaf(a2, x2, a0)) : fmaf(b3, x3, fmaf(b2, x2, fmaf(b1, x, b0))));
    return x < 2.f ? r : 0.f;
}
/* CatmullRomFilter::eval (src/rfilters/catmullrom.cpp:39-52) */
static float catmullrom_eval(float x) {
    x = fabsf(x);
    float x2 = x * x, x3 = x2 * x, B = 0.f, C = .5f;
    float r = (1.f / 6.f) * (x < 1.f ? (12.f - 9.f * B - 6.f * C) * x3 + (-18.f + 12.f * B + 6.f * C) * x2 + (6.f - 2.f * B)
                                     : (-B - 6.f * C) * x3 + (6.f * B + 30.f * C) * x2 + (-12.f * B - 48.f * C) * x + (8.f * B + 24.f * C));
    return x < 2.f ? r : 0.f;
}
/* ReconstructionFilter::eval of the film's filter: src/rfilters/tent.cpp:53-55, gaussian.cpp:94-98, mitchell.cpp:60-79, catmullrom.cpp:39-52 */
/* LanczosSincFilter::eval (src/rfilters/lanczos.cpp:52-63): radius = lobes; dr::sin is the sine of orc_sincos */
static float lanczos_eval(float x, float radius) {
    x = fabsf(x);
    float x1 = ORC_PI_F * x, x2 = x1 / radius, s1, s2, c;
    orc_sincos(x1, &s1, &c); orc_sincos(x2, &s2, &c);
    float result = (s1 * s2) / (x1 * x2);
    return x < ORC_EPSILON_F ? 1.f : (x > radius ? 0.f : result);
}
static float filter_eval(const orc_sensor *se, float x, float inv_r, const float *gc) {
    switch (se->filter) {
        case ORC_FILTER_LANCZOS:    return lanczos_eval(x, se->filter_radius);
        case ORC_FILTER_GAUSSIAN:   return f_max(estrin10(x * x, gc), 0.f);
        case ORC_FILTER_MITCHELL:   return mitchell_eval(x, se->filter_b, se->filter_c);
        case ORC_FILTER_CATMULLROM: return catmullrom_eval(x);
        default:                    return f_max(0.f, 1.f - fabsf(x * inv_r));
    }
}
/* ImageBlock::put, non-coalesced accumulation of one sample into its filter footprint (src/render/imageblock.cpp:414-531; box filter: :119-133) */
static void splat(const orc_sensor *se, float *film, float spx, float spy, int pixel_x, int pixel_y, const float *rgb) {
    int W = se->crop_w, H = se->crop_h;
    float vals[4] = { rgb[0], rgb[1], rgb[2], 1.f };
    if (se->filter == ORC_FILTER_BOX) {
        /* "With box filter, ignore random offset": block->put(box_filter ? pos : sample_pos) (integrator.cpp:540-541);
         * pos is the lane's integer pixel, so floor(pos) - offset is the pixel itself */
        int x = pixel_x, y = pixel_y;
        if ((unsigned) x < (unsigned) W && (unsigned) y < (unsigned) H)
            for (int k = 0; k < 4; ++k) film[4 * ((size_t) y * W + x) + k] += vals[k];
        return;
    }
    float radius = se->filter_radius, inv_r = 1.f / radius, gc[10];
    const int gauss = se->filter == ORC_FILTER_GAUSSIAN;
    if (gauss) gaussian_coeffs(se->filter_stddev, radius, gc);
    int n = (int) ceilf(radius - .5f), count = 2 * n + 1;
    int pix = (int) floorf(spx) - n, piy = (int) floorf(spy) - n;
    float relx = (float) pix + .5f - spx, rely = (float) piy + .5f - spy;
    int lx = pix - se->crop_x, ly = piy - se->crop_y;
    for (int ys = 0; ys < count; ++ys) {
        float ry = rely + (float) ys;
        float wy = filter_eval(se, ry, inv_r, gc);
        for (int xs = 0; xs < count; ++xs) {
            float rx = relx + (float) xs;
            float wx = filter_eval(se, rx, inv_r, gc);
            float w = wx * wy;
            int x = lx + xs, y = ly + ys;
            if ((unsigned) x < (unsigned) W && (unsigned) y < (unsigned) H)
                for (int k = 0; k < 4; ++k) film[4 * ((size_t) y * W + x) + k] += vals[k] * w;
        }
    }
}
/* HDRFilm::develop -- src/films/hdrfilm.cpp:305-406: RGB / W unless W == 0 */
void orc_develop(const float *film, float *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        float w = film[4 * i + 3]; w = w == 0.f ? 1.f : w;
        out[3 * i] = film[4 * i] / w; out[3 * i + 1] = film[4 * i + 1] / w; out[3 * i + 2] = film[4 * i + 2] / w;
    }
}
/* Splatting is parallel too: the rows of a chunk are cut into one band per thread, every thread splats the lanes of its
 * band (in lane order) into a private film of band + 2 * border rows, and the bands are added to the film in band order.
 * With one thread this is the plain sequential accumulation in lane order; with more, only the order of the float additions
 * across band borders differs (the reference's atomic scatter order is unspecified anyway). */
typedef struct { const orc_sensor *se; const orc_lane *lanes; uint64_t lane0, n; uint32_t spp; int W, H, row0, rows, border; float *band; } splat_job;
static void *splat_worker(void *arg) {
    splat_job *j = (splat_job *) arg;
    orc_sensor se = *j->se;                 /* a sensor whose film is the band: crop origin moved, height = band + borders */
    const int band_h = j->rows + 2 * j->border;
    se.crop_y = j->se->crop_y + j->row0 - j->border; se.crop_h = band_h;
    for (uint64_t i = 0; i < j->n; ++i) {
        uint64_t pix = (j->lane0 + i) / j->spp;
        int px = (int) (pix % (uint64_t) j->W), py = (int) (pix / (uint64_t) j->W) - (j->row0 - j->border);
        splat(&se, j->band, j->lanes[i].sample_pos[0], j->lanes[i].sample_pos[1], px, py, j->lanes[i].rgb);
    }
    return NULL;
}
uint64_t orc_render(const orc_scene *sc, const orc_params *p, uint32_t seed, uint32_t spp,
                    int32_t row_begin, int32_t row_end, float *film, float *out_rgb, int nt) {
    orc_ctx cx; make_ctx(&cx, sc, p, seed, spp);
    int W = sc->sensor.crop_w, H = sc->sensor.crop_h;
    if (row_begin < 0) row_begin = 0;
    if (row_end > H) row_end = H;
    if (nt < 1) nt = 1;
    if (nt > 256) nt = 256;
    const uint32_t spw = cx.spw;
    const uint64_t wavefront = (uint64_t) W * (uint64_t) H * spw;
    uint64_t lanes_per_row = (uint64_t) W * spw, total = 0;
    int chunk_rows = (int) (8000000ull / lanes_per_row); if (chunk_rows < 1) chunk_rows = 1;
    const int border = sc->sensor.filter == ORC_FILTER_BOX ? 0 : (int) ceilf(sc->sensor.filter_radius - .5f);
    orc_lane *buf = (orc_lane *) malloc(sizeof(orc_lane) * lanes_per_row * (size_t) chunk_rows);
    for (uint32_t pass = 0; pass < cx.n_passes; ++pass)
    for (int r = row_begin; r < row_end; r += chunk_rows) {
        int re = r + chunk_rows < row_end ? r + chunk_rows : row_end;
        uint64_t n = lanes_per_row * (uint64_t) (re - r);
        run_lanes(&cx, (uint64_t) pass * wavefront + lanes_per_row * (uint64_t) r, n, buf, nt);
        int bands = nt < re - r ? nt : re - r;
        if (bands == 1) {                       /* one thread: straight into the film, lane order */
            for (uint64_t i = 0; i < n; ++i) {
                uint64_t pix = (lanes_per_row * (uint64_t) r + i) / spw;
                splat(&sc->sensor, film, buf[i].sample_pos[0], buf[i].sample_pos[1], (int) (pix % (uint64_t) W), (int) (pix / (uint64_t) W), buf[i].rgb);
            }
            total += n;
            continue;
        }
        splat_job jobs[256]; pthread_t th[256];
        for (int b = 0; b < bands; ++b) {
            int b0 = r + (int) ((int64_t) (re - r) * b / bands), b1 = r + (int) ((int64_t) (re - r) * (b + 1) / bands);
            splat_job *j = &jobs[b];
            j->se = &sc->sensor; j->spp = spw; j->W = W; j->H = H; j->row0 = b0; j->rows = b1 - b0; j->border = border;
            j->lane0 = lanes_per_row * (uint64_t) b0; j->n = lanes_per_row * (uint64_t) (b1 - b0);
            j->lanes = buf + (j->lane0 - lanes_per_row * (uint64_t) r);
            j->band = (float *) calloc((size_t) (j->rows + 2 * border) * W * 4 + 4, sizeof(float));
            pthread_create(&th[b], NULL, splat_worker, j);
        }
        for (int b = 0; b < bands; ++b) {
            splat_job *j = &jobs[b];
            pthread_join(th[b], NULL);
            for (int y = 0; y < j->rows + 2 * border; ++y) {
                int fy = j->row0 - border + y;
                if (fy < 0 || fy >= H) continue;
                float *dst = film + (size_t) fy * W * 4; const float *src = j->band + (size_t) y * W * 4;
                for (int x = 0; x < W * 4; ++x) dst[x] += src[x];
            }
            free(j->band);
        }
        total += n;
    }
    free(buf);
    if (out_rgb) orc_develop(film, out_rgb, (int64_t) W * H);
    return total;
}

/* The order-independent value of the same film: every splat term is the float32 product the reference adds (value * wx * wy,
 * imageblock.cpp:414-531), but the terms are summed in float64 and RGB / W is taken in float64 -- what any order of the reference's
 * unordered float32 scatter-adds (imageblock.cpp:119-133) scatters around.  Doppler images are sums of large terms of both signs, so a
 * float32 sum in ONE particular order is not a reference value for pixels that cancel to a small fraction of their terms: the parity
 * tests hold the GPU film AND the float32 film of orc_render against this one. */
uint64_t orc_render_exact(const orc_scene *sc, const orc_params *p, uint32_t seed, uint32_t spp,
                          int32_t row_begin, int32_t row_end, double *film, float *out_rgb, int nt) {
    orc_ctx cx; make_ctx(&cx, sc, p, seed, spp);
    const orc_sensor *se = &sc->sensor;
    int W = se->crop_w, H = se->crop_h;
    if (row_begin < 0) row_begin = 0;
    if (row_end > H) row_end = H;
    if (nt < 1) nt = 1;
    if (nt > 256) nt = 256;
    const uint32_t spw = cx.spw;
    const uint64_t wavefront = (uint64_t) W * (uint64_t) H * spw, lanes_per_row = (uint64_t) W * spw;
    uint64_t total = 0;
    int chunk_rows = (int) (8000000ull / lanes_per_row); if (chunk_rows < 1) chunk_rows = 1;
    orc_lane *buf = (orc_lane *) malloc(sizeof(orc_lane) * lanes_per_row * (size_t) chunk_rows);
    float radius = se->filter_radius, inv_r = 1.f / radius, gc[10];
    if (se->filter == ORC_FILTER_GAUSSIAN) gaussian_coeffs(se->filter_stddev, radius, gc);
    const int n = se->filter == ORC_FILTER_BOX ? 0 : (int) ceilf(radius - .5f), count = 2 * n + 1;
    for (uint32_t pass = 0; pass < cx.n_passes; ++pass)
    for (int r = row_begin; r < row_end; r += chunk_rows) {
        int re = r + chunk_rows < row_end ? r + chunk_rows : row_end;
        uint64_t nl = lanes_per_row * (uint64_t) (re - r);
        run_lanes(&cx, (uint64_t) pass * wavefront + lanes_per_row * (uint64_t) r, nl, buf, nt);
        for (uint64_t i = 0; i < nl; ++i) {
            const uint64_t pix = (lanes_per_row * (uint64_t) r + i) / spw;
            const float vals[4] = { buf[i].rgb[0], buf[i].rgb[1], buf[i].rgb[2], 1.f };
            if (se->filter == ORC_FILTER_BOX) {
                const int x = (int) (pix % (uint64_t) W), y = (int) (pix / (uint64_t) W);
                for (int k = 0; k < 4; ++k) film[4 * ((size_t) y * W + x) + k] += (double) vals[k];
                continue;
            }
            const float spx = buf[i].sample_pos[0], spy = buf[i].sample_pos[1];
            const int pxi = (int) floorf(spx) - n, pyi = (int) floorf(spy) - n;
            const float relx = (float) pxi + .5f - spx, rely = (float) pyi + .5f - spy;
            const int lx = pxi - se->crop_x, ly = pyi - se->crop_y;
            for (int ys = 0; ys < count; ++ys) {
                const float wy = filter_eval(se, rely + (float) ys, inv_r, gc);
                for (int xs = 0; xs < count; ++xs) {
                    const float wx = filter_eval(se, relx + (float) xs, inv_r, gc), w = wx * wy;
                    const int x = lx + xs, y = ly + ys;
                    if ((unsigned) x < (unsigned) W && (unsigned) y < (unsigned) H)
                        for (int k = 0; k < 4; ++k) film[4 * ((size_t) y * W + x) + k] += (double) (vals[k] * w);
                }
            }
        }
        total += nl;
    }
    free(buf);
    if (out_rgb) for (int64_t i = 0; i < (int64_t) W * H; ++i) {
        double w = film[4 * i + 3]; w = w == 0.0 ? 1.0 : w;
        out_rgb[3 * i] = (float) (film[4 * i] / w); out_rgb[3 * i + 1] = (float) (film[4 * i + 1] / w); out_rgb[3 * i + 2] = (float) (film[4 * i + 2] / w);
    }
    return total;
}

/* The alpha channel of an rgba film (hdrfilm.cpp:172-177 FilmFlags::Alpha; integrator.cpp:528-533: aovs[3] = select(valid, 1, 0), aovs[4] = 1; develop
 * divides every channel by the weight, hdrfilm.cpp:339-400): sum of valid * wx * wy over sum of wx * wy, the terms in float32 as the reference forms them,
 * the sums in float64 (orc_render_exact explains why).  out_alpha: crop_h * crop_w floats. */
uint64_t orc_render_alpha(const orc_scene *sc, const orc_params *p, uint32_t seed, uint32_t spp, float *out_alpha, int nt) {
    orc_ctx cx; make_ctx(&cx, sc, p, seed, spp);
    const orc_sensor *se = &sc->sensor;
    const int W = se->crop_w, H = se->crop_h;
    if (nt < 1) nt = 1;
    if (nt > 256) nt = 256;
    const uint32_t spw = cx.spw;
    const uint64_t wavefront = (uint64_t) W * (uint64_t) H * spw, lanes_per_row = (uint64_t) W * spw;
    uint64_t total = 0;
    int chunk_rows = (int) (8000000ull / lanes_per_row); if (chunk_rows < 1) chunk_rows = 1;
    orc_lane *buf = (orc_lane *) malloc(sizeof(orc_lane) * lanes_per_row * (size_t) chunk_rows);
    double *film = (double *) calloc((size_t) W * H * 2, sizeof(double));
    float radius = se->filter_radius, inv_r = 1.f / radius, gc[10];
    if (se->filter == ORC_FILTER_GAUSSIAN) gaussian_coeffs(se->filter_stddev, radius, gc);
    const int n = se->filter == ORC_FILTER_BOX ? 0 : (int) ceilf(radius - .5f), count = 2 * n + 1;
    for (uint32_t pass = 0; pass < cx.n_passes; ++pass)
    for (int r = 0; r < H; r += chunk_rows) {
        const int re = r + chunk_rows < H ? r + chunk_rows : H;
        const uint64_t nl = lanes_per_row * (uint64_t) (re - r);
        run_lanes(&cx, (uint64_t) pass * wavefront + lanes_per_row * (uint64_t) r, nl, buf, nt);
        for (uint64_t i = 0; i < nl; ++i) {
            const uint64_t pix = (lanes_per_row * (uint64_t) r + i) / spw;
            const float a = buf[i].valid ? 1.f : 0.f;
            if (se->filter == ORC_FILTER_BOX) {
                const int x = (int) (pix % (uint64_t) W), y = (int) (pix / (uint64_t) W);
                film[2 * ((size_t) y * W + x)] += (double) a; film[2 * ((size_t) y * W + x) + 1] += 1.0;
                continue;
            }
            const float spx = buf[i].sample_pos[0], spy = buf[i].sample_pos[1];
            const int pxi = (int) floorf(spx) - n, pyi = (int) floorf(spy) - n;
            const float relx = (float) pxi + .5f - spx, rely = (float) pyi + .5f - spy;
            const int lx = pxi - se->crop_x, ly = pyi - se->crop_y;
            for (int ys = 0; ys < count; ++ys) {
                const float wy = filter_eval(se, rely + (float) ys, inv_r, gc);
                for (int xs = 0; xs < count; ++xs) {
                    const float wx = filter_eval(se, relx + (float) xs, inv_r, gc), w = wx * wy;
                    const int x = lx + xs, y = ly + ys;
                    if ((unsigned) x < (unsigned) W && (unsigned) y < (unsigned) H) { film[2 * ((size_t) y * W + x)] += (double) (a * w); film[2 * ((size_t) y * W + x) + 1] += (double) w; }
                }
            }
        }
        total += nl;
    }
    for (int64_t i = 0; i < (int64_t) W * H; ++i) { double w = film[2 * i + 1]; w = w == 0.0 ? 1.0 : w; out_alpha[i] = (float) (film[2 * i] / w); }
    free(film); free(buf);
    return total;
}

/* Cube vertex baking -- src/shapes/cube.cpp:114-160 (scalar float32: positions through
 * to_world, normals through its inverse transpose then normalised with 1/sqrt). */
void orc_bake_cube(const float *to_world, const float *to_object, float *pos, float *nrm, float *uv, uint32_t *faces) {
    static const float vtx[24][3] = {
        { 1,-1,-1},{ 1,-1, 1},{-1,-1, 1},{-1,-1,-1},{ 1, 1,-1},{-1, 1,-1},{-1, 1, 1},{ 1, 1, 1},
        { 1,-1,-1},{ 1, 1,-1},{ 1, 1, 1},{ 1,-1, 1},{ 1,-1, 1},{ 1, 1, 1},{-1, 1, 1},{-1,-1, 1},
        {-1,-1, 1},{-1, 1, 1},{-1, 1,-1},{-1,-1,-1},{ 1, 1,-1},{ 1,-1,-1},{-1,-1,-1},{-1, 1,-1} };
    static const float nr[6][3] = { {0,-1,0},{0,1,0},{1,0,0},{0,0,1},{-1,0,0},{0,0,-1} };
    static const float tc[4][2] = { {0,1},{1,1},{1,0},{0,0} };
    static const uint32_t tri[12][3] = { {0,1,2},{3,0,2},{4,5,6},{7,4,6},{8,9,10},{11,8,10},
        {12,13,14},{15,12,14},{16,17,18},{19,16,18},{20,21,22},{23,20,22} };
    for (int i = 0; i < 24; ++i) {
        v3 p = m_point(to_world, V(vtx[i][0], vtx[i][1], vtx[i][2]));
        v3 n = m_normal(to_object, V(nr[i / 4][0], nr[i / 4][1], nr[i / 4][2]));
        n = v_mul(n, 1.0f / sqrtf(v_dot(n, n)));
        pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z;
        nrm[3 * i] = n.x; nrm[3 * i + 1] = n.y; nrm[3 * i + 2] = n.z;
        uv[2 * i] = tc[i % 4][0]; uv[2 * i + 1] = tc[i % 4][1];
    }
    memcpy(faces, tri, sizeof tri);
}

/* Mesh vertex baking -- obj.cpp:218-246 / ply.cpp:284-300 + mesh.cpp:257-345 (see the header). */
void orc_bake_mesh(const float *to_world, const float *to_object, int32_t n_vertices, const float *pos_in,
                   const float *nrm_in, int32_t n_faces, const uint32_t *faces, int32_t face_normals,
                   float *pos_out, float *nrm_out) {
    for (int32_t i = 0; i < n_vertices; ++i) {
        v3 p = m_point(to_world, V(pos_in[3 * i], pos_in[3 * i + 1], pos_in[3 * i + 2]));
        pos_out[3 * i] = p.x; pos_out[3 * i + 1] = p.y; pos_out[3 * i + 2] = p.z;
    }
    if (face_normals || !nrm_out) return;
    if (nrm_in) {
        for (int32_t i = 0; i < n_vertices; ++i) {
            v3 n = v_normalize(m_normal(to_object, V(nrm_in[3 * i], nrm_in[3 * i + 1], nrm_in[3 * i + 2])));
            nrm_out[3 * i] = n.x; nrm_out[3 * i + 1] = n.y; nrm_out[3 * i + 2] = n.z;
        }
        return;
    }
    double *acc = (double *) calloc((size_t) n_vertices * 3 + 1, sizeof(double));
    for (int32_t f = 0; f < n_faces; ++f) {
        const uint32_t *fi = faces + 3 * (size_t) f;
        double v[3][3];
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 3; ++c) v[k][c] = (double) pos_out[3 * (size_t) fi[k] + c];
        double s0[3], s1[3], n[3];
        for (int c = 0; c < 3; ++c) { s0[c] = v[1][c] - v[0][c]; s1[c] = v[2][c] - v[0][c]; }
        n[0] = s0[1] * s1[2] - s0[2] * s1[1]; n[1] = s0[2] * s1[0] - s0[0] * s1[2]; n[2] = s0[0] * s1[1] - s0[1] * s1[0];
        double l2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
        if (!(l2 > 0.0)) continue;
        double il = 1.0 / sqrt(l2);
        for (int c = 0; c < 3; ++c) n[c] *= il;
        for (int k = 0; k < 3; ++k) {
            double d0[3], d1[3], l0 = 0, l1 = 0, dt = 0;
            for (int c = 0; c < 3; ++c) { d0[c] = v[(k + 1) % 3][c] - v[k][c]; d1[c] = v[(k + 2) % 3][c] - v[k][c]; l0 += d0[c] * d0[c]; l1 += d1[c] * d1[c]; }
            l0 = 1.0 / sqrt(l0); l1 = 1.0 / sqrt(l1);
            for (int c = 0; c < 3; ++c) dt += (d0[c] * l0) * (d1[c] * l1);
            double ang = acos(dt > 1.0 ? 1.0 : (dt < -1.0 ? -1.0 : dt));
            for (int c = 0; c < 3; ++c) acc[3 * (size_t) fi[k] + c] += n[c] * ang;
        }
    }
    for (int32_t i = 0; i < n_vertices; ++i) {
        double *a = acc + 3 * (size_t) i, l = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        if (l != 0.0 && l == l) { nrm_out[3 * i] = (float) (a[0] / l); nrm_out[3 * i + 1] = (float) (a[1] / l); nrm_out[3 * i + 2] = (float) (a[2] / l); }
        else { nrm_out[3 * i] = 1.f; nrm_out[3 * i + 1] = 0.f; nrm_out[3 * i + 2] = 0.f; }
    }
    free(acc);
}

/* Mesh::build_pmf + DiscreteDistribution::compute_cdf -- see the header */
int orc_mesh_area_table(const float *P, int32_t n_faces, const uint32_t *faces, float *pmf, float *cdf,
                        float *sum_out, float *norm_out, int32_t *lo, int32_t *hi) {
    if (n_faces <= 0) return -1;
    double sum = 0.0; *lo = -1; *hi = -1;
    for (int32_t i = 0; i < n_faces; ++i) {
        const uint32_t *fi = faces + 3 * (size_t) i;
        v3 p0 = V(P[3 * fi[0]], P[3 * fi[0] + 1], P[3 * fi[0] + 2]), p1 = V(P[3 * fi[1]], P[3 * fi[1] + 1], P[3 * fi[1] + 2]),
           p2 = V(P[3 * fi[2]], P[3 * fi[2] + 1], P[3 * fi[2] + 2]);
        pmf[i] = .5f * v_norm(v_cross(v_sub(p1, p0), v_sub(p2, p0)));
        double value = (double) pmf[i];
        sum += value;
        cdf[i] = (float) sum;
        if (value > 0.0) { if (*lo < 0) *lo = i; *hi = i; }
    }
    if (*lo < 0) return -1;
    *sum_out = (float) sum; *norm_out = (float) (1.0 / sum);
    return 0;
}

/* Sphere ctor + update -- see the header */
static void m4_mul_f32(const float *a, const float *b, float *out) {   /* out_ij = fmadd chain over k (Dr.Jit Matrix operator*) */
    float r[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float sum = a[4 * i] * b[j];
        for (int k = 1; k < 4; ++k) sum = fmaf(a[4 * i + k], b[4 * k + j], sum);
        r[4 * i + j] = sum;
    }
    memcpy(out, r, sizeof r);
}
void orc_bake_sphere(const float *to_world, const float *to_object, const float *center, float radius, int32_t flip_normals,
                     float *composed, float *composed_inv, float *out8) {
    float T[16] = { 1, 0, 0, center[0], 0, 1, 0, center[1], 0, 0, 1, center[2], 0, 0, 0, 1 };
    float Ti[16] = { 1, 0, 0, -center[0], 0, 1, 0, -center[1], 0, 0, 1, -center[2], 0, 0, 0, 1 };
    float ir = 1.0f / radius;
    float S[16] = { radius, 0, 0, 0, 0, radius, 0, 0, 0, 0, radius, 0, 0, 0, 0, 1 }, Si[16] = { ir, 0, 0, 0, 0, ir, 0, 0, 0, 0, ir, 0, 0, 0, 0, 1 };
    float tmp[16];
    m4_mul_f32(to_world, T, tmp); m4_mul_f32(tmp, S, composed);
    m4_mul_f32(Ti, to_object, tmp); m4_mul_f32(Si, tmp, composed_inv);
    v3 c0 = V(composed[0], composed[4], composed[8]);
    float r = v_norm(c0);
    const float *m = composed;
    float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
    int flip = flip_normals != 0;
    if (det < 0.f) flip = !flip;
    out8[0] = composed[3]; out8[1] = composed[7]; out8[2] = composed[11]; out8[3] = r;
    out8[4] = 1.0f / ((4.f * ORC_PI_F) * f_sqr(r)); out8[5] = flip ? 1.f : 0.f; out8[6] = out8[7] = 0.f;
}

void orc_bake_cylinder(const float *to_world, const float *to_object, const float *p0, const float *p1, float radius, int32_t flip_normals,
                       float *composed, float *composed_inv, float *out8) {
    v3 d = V(p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]);
    float length = v_norm(d);
    v3 n = v_mul(d, f_rcp(length)), fs, ft;
    coordinate_system(n, &fs, &ft);
    float T[16] = { 1, 0, 0, p0[0], 0, 1, 0, p0[1], 0, 0, 1, p0[2], 0, 0, 0, 1 }, Ti[16] = { 1, 0, 0, -p0[0], 0, 1, 0, -p0[1], 0, 0, 1, -p0[2], 0, 0, 0, 1 };
    float F[16] = { fs.x, ft.x, n.x, 0, fs.y, ft.y, n.y, 0, fs.z, ft.z, n.z, 0, 0, 0, 0, 1 };       /* columns s, t, n (transform.h:286-296) */
    float Fi[16] = { fs.x, fs.y, fs.z, 0, ft.x, ft.y, ft.z, 0, n.x, n.y, n.z, 0, 0, 0, 0, 1 };
    float ir = 1.0f / radius, il = 1.0f / length;
    float S[16] = { radius, 0, 0, 0, 0, radius, 0, 0, 0, 0, length, 0, 0, 0, 0, 1 }, Si[16] = { ir, 0, 0, 0, 0, ir, 0, 0, 0, 0, il, 0, 0, 0, 0, 1 };
    float a[16], b[16];
    m4_mul_f32(to_world, T, a); m4_mul_f32(a, F, b); m4_mul_f32(b, S, composed);
    m4_mul_f32(Ti, to_object, a); m4_mul_f32(Fi, a, b); m4_mul_f32(Si, b, composed_inv);
    float r = v_norm(V(composed[0], composed[4], composed[8])), l = v_norm(V(composed[2], composed[6], composed[10]));
    const float *m = composed;
    float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
    int flip = flip_normals != 0;
    if (det < 0.f) flip = !flip;
    out8[0] = r; out8[1] = l; out8[2] = 1.0f / ((2.f * ORC_PI_F) * r * l); out8[3] = flip ? 1.f : 0.f; out8[4] = out8[5] = out8[6] = out8[7] = 0.f;
}
void orc_fresnel_dielectric(float cos_theta_i, float eta, float *out4) { fresnel_dielectric(cos_theta_i, eta, out4, out4 + 1, out4 + 2, out4 + 3); }
float orc_fresnel_conductor(float cos_theta_i, float eta, float k) { return fresnel_conductor(cos_theta_i, eta, k); }

/* fresnel_diffuse_reflectance (include/mitsuba/render/fresnel.h:328-355) and SmoothPlastic::parameters_changed (src/bsdfs/plastic.cpp:201-217) */
static float fresnel_diffuse_reflectance(float eta) {
    float inv_eta = 1.0f / eta;
    float approx_1 = fmaf(0.0636f, inv_eta, fmaf(eta, fmaf(eta, -1.4399f, 0.7099f), 0.6681f));
    /* dr::horner(x, c0, ..., c5) = c0 + x (c1 + x (c2 + ...)), evaluated with fmadd from the highest coefficient down */
    float h = -1.36881f;
    h = fmaf(h, inv_eta, 4.98554f); h = fmaf(h, inv_eta, -7.80989f); h = fmaf(h, inv_eta, 6.75335f);
    h = fmaf(h, inv_eta, -3.4793f); h = fmaf(h, inv_eta, 0.919317f);
    return eta < 1.f ? approx_1 : h;
}
void orc_plastic_params(float eta, const float *d, const float *sp, float *out3) {
    out3[0] = 1.f / (eta * eta);
    out3[1] = fresnel_diffuse_reflectance(1.f / eta);
    float d_mean = ((d[0] + d[1]) + d[2]) * (1.0f / 3.0f), s_mean = ((sp[0] + sp[1]) + sp[2]) * (1.0f / 3.0f);
    out3[2] = s_mean / (d_mean + s_mean);
}
/* quad::gauss_legendre (include/mitsuba/core/quad.h:27-86), math::legendre_pd (include/mitsuba/core/math.h:92-119) */
static void legendre_pd(int l, double x, double *lv, double *dv) {
    if (l == 0) { *lv = 1; *dv = 0; return; }
    if (l == 1) { *lv = x; *dv = 1; return; }
    double l_p_pred = 1, l_pred = x, d_p_pred = 0, d_pred = 1, k0 = 3, k1 = 2, k2 = 1, l_cur = 0, d_cur = 0;
    for (int ki = 2; ki <= l; ++ki) {
        l_cur = (k0 * x * l_pred - k2 * l_p_pred) / k1;
        d_cur = d_p_pred + k0 * l_pred;
        l_p_pred = l_pred; l_pred = l_cur; d_p_pred = d_pred; d_pred = d_cur;
        k2 = k1; k0 += 2; k1 += 1;
    }
    *lv = l_cur; *dv = d_cur;
}
void orc_gauss_legendre(int n, float *nodes, float *weights) {
    n--;
    if (n == 0) { nodes[0] = 0.f; weights[0] = 2.f; }
    else if (n == 1) { nodes[0] = (float) -sqrt(1.0 / 3.0); nodes[1] = -nodes[0]; weights[0] = weights[1] = 1.f; }
    int m = (n + 1) / 2;
    for (int i = 0; i < m; ++i) {
        double x = -cos((double) (2 * i + 1) / (double) (2 * n + 2) * 3.14159265358979323846), lv, dv;
        for (int it = 1; it <= 20; ++it) {   /* the reference throws after 20 iterations; it converges in a handful */
            legendre_pd(n + 1, x, &lv, &dv);
            double step = lv / dv;
            x -= step;
            if (fabs(step) <= 4 * fabs(x) * (2.220446049250313e-16 / 2)) break;
        }
        legendre_pd(n + 1, x, &lv, &dv);
        weights[i] = weights[n - i] = (float) (2 / ((1 - x * x) * (dv * dv)));
        nodes[i] = (float) x; nodes[n - i] = (float) -x;
    }
    if ((n % 2) == 0) {
        double lv, dv; legendre_pd(n + 1, 0.0, &lv, &dv);
        weights[n / 2] = (float) (2.0 / (dv * dv)); nodes[n / 2] = 0.f;
    }
}
/* eval_transmittance (transmit) / eval_reflectance for one incident direction: microfacet.h:463-566 */
static float rough_integral(ggx_t g, v3 wi, float eta, int transmit) {
    int res = eta > 1.f ? 32 : 128;
    float nodes[128], weights[128];
    orc_gauss_legendre(res, nodes, weights);
    float result = 0.f;
    for (int j = 0; j < res * res; ++j) {   /* dr::meshgrid: x runs fastest */
        float nx = fmaf(nodes[j % res], 0.5f, 0.5f), ny = fmaf(nodes[j / res], 0.5f, 0.5f), w = weights[j % res] * weights[j / res];
        float pdf, f, cos_theta_t, eta_it, eta_ti, smith;
        v3 m = ggx_sample(g, wi, nx, ny, &pdf);
        float dwm = v_dot(wi, m);
        fresnel_dielectric(dwm, eta, &f, &cos_theta_t, &eta_it, &eta_ti);
        if (transmit) {   /* refract(wi, m, cos_theta_t, eta_ti), fresnel.h:311-314 */
            float k = fmaf(dwm, eta_ti, cos_theta_t);
            v3 wo = V(fmaf(m.x, k, -(wi.x * eta_ti)), fmaf(m.y, k, -(wi.y * eta_ti)), fmaf(m.z, k, -(wi.z * eta_ti)));
            smith = ggx_smith_g1(g, wo, m) * (1.f - f);
            if (wo.z * wi.z >= 0.f) smith = 0.f;
        } else {
            v3 wo = V(fmaf(m.x, 2.f * dwm, -wi.x), fmaf(m.y, 2.f * dwm, -wi.y), fmaf(m.z, 2.f * dwm, -wi.z));
            smith = ggx_smith_g1(g, wo, m) * f;
            if (wo.z <= 0.f || wi.z <= 0.f) smith = 0.f;
        }
        result += smith * w * 0.25f;
    }
    return result;
}
void orc_roughplastic_tables(int type, float alpha, float eta, float *table64, float *internal_reflectance) {
    ggx_t g = mf_make(type, alpha, alpha, 1);
    float sum = 0.f;
    for (int i = 0; i < 64; ++i) {
        float mu = f_max(1e-6f, fmaf((float) i, 1.f / 63.f, 0.f));
        v3 wi = V(sqrtf(1.f - mu * mu), 0.f, mu);
        table64[i] = rough_integral(g, wi, eta, 1);
        sum += rough_integral(g, wi, 1.f / eta, 0) * wi.z;
    }
    *internal_reflectance = sum * (1.f / 64.f) * 2.f;
}

/* ------------------------------------------------------------------ known-answer entry points
 * Thin wrappers that expose the building blocks eval_lane is made of, so that tests/test_oracle_reference_kats.py can hold
 * them against the numbers of the reference's own unit tests (tests/golden/reference_kats.json.gz). */
void orc_kat_microfacet(int type, float au, float av, int visible, int fn, const float *in, float *out) {
    ggx_t g = mf_make(type, au, av, visible);
    switch (fn) {
        case 0: out[0] = ggx_eval(g, V(in[0], in[1], in[2])); break;
        case 1: out[0] = ggx_pdf(g, V(in[0], in[1], in[2]), V(in[3], in[4], in[5])); break;
        case 2: out[0] = ggx_smith_g1(g, V(in[0], in[1], in[2]), V(in[3], in[4], in[5])); break;
        default: { v3 m = ggx_sample(g, V(in[0], in[1], in[2]), in[3], in[4], out + 3); out[0] = m.x; out[1] = m.y; out[2] = m.z; }
    }
}
float orc_kat_filter(int kind, float radius, float stddev, float B, float C, float x) {
    orc_sensor se; memset(&se, 0, sizeof se);
    se.filter = kind; se.filter_radius = radius; se.filter_stddev = stddev; se.filter_b = B; se.filter_c = C;
    float gc[10];
    if (kind == ORC_FILTER_BOX) return (x >= -radius && x < radius) ? 1.f : 0.f;   /* BoxFilter::eval (src/rfilters/box.cpp) */
    if (kind == ORC_FILTER_GAUSSIAN) gaussian_coeffs(stddev, radius, gc);
    float v = filter_eval(&se, x, 1.f / radius, gc);
    return fabsf(x) < radius ? v : 0.f;
}
void orc_kat_warp(int fn, const float *in, float *out) {
    switch (fn) {
        case 0: { v3 d = square_to_cosine_hemisphere(in[0], in[1]); out[0] = d.x; out[1] = d.y; out[2] = d.z; } break;
        case 1: { v3 d = square_to_cosine_hemisphere(in[0], in[1]); out[0] = d.x; out[1] = d.y; } break;   /* x, y = the concentric disk */
        case 3: square_to_uniform_triangle(in[0], in[1], out, out + 1); break;
        default: { v3 d = square_to_uniform_sphere(in[0], in[1]); out[0] = d.x; out[1] = d.y; out[2] = d.z; }
    }
}
void orc_kat_frame(const float *n, float *out6) {
    v3 s, t; coordinate_system(V(n[0], n[1], n[2]), &s, &t);
    out6[0] = s.x; out6[1] = s.y; out6[2] = s.z; out6[3] = t.x; out6[4] = t.y; out6[5] = t.z;
}
/* Scene::ray_intersect: out = t, p[3], n[3], sh_n[3], sh_s[3], sh_t[3], dp_du[3], dp_dv[3], wi[3] (25 floats); returns hit */
int orc_kat_ray_intersect(const orc_scene *sc, const float *o, const float *d, float time, float maxt, float *out, int32_t *ids) {
    v3 ro = V(o[0], o[1], o[2]), rd = V(d[0], d[1], d[2]);
    orc_hit h = scene_closest(sc, ro, rd, time, maxt);
    ids[0] = h.obj; ids[1] = h.shape; ids[2] = h.prim;
    memset(out, 0, 25 * sizeof(float));
    if (h.obj < 0) { out[0] = INFINITY; return 0; }
    orc_si si; memset(&si, 0, sizeof si);
    compute_si(sc, &h, ro, rd, time, &si);
    const v3 *f[8] = { &si.p, &si.n, &si.sh_n, &si.sh_s, &si.sh_t, &si.dp_du, &si.dp_dv, &si.wi };
    out[0] = h.t;
    for (int i = 0; i < 8; ++i) { out[1 + 3 * i] = f[i]->x; out[2 + 3 * i] = f[i]->y; out[3 + 3 * i] = f[i]->z; }
    return 1;
}
/* BSDF::eval / pdf for `wo` and BSDF::sample with (sample1, sample2) of shape `sh`'s BSDF at local incident direction wi:
 * out = value[3], pdf, bs.wo[3], bs.pdf, bs.eta, bs.delta, weight[3] (13 floats) */
void orc_kat_bsdf(const orc_shape *sh, const float *wi, const float *wo, const float *s3, float *out) {
    orc_bsdf_out r;
    const orc_geo flat_geo = { V(1.f, 0.f, 0.f), V(0.f, 1.f, 0.f), V(0.f, 0.f, 1.f), V(1.f, 0.f, 0.f), V(0.f, 1.f, 0.f), V(0.f, 0.f, 1.f) };
    bsdf_eval_pdf_sample(sh, &flat_geo, V(wi[0], wi[1], wi[2]), V(wo[0], wo[1], wo[2]), 1, s3[0], s3[1], s3[2], 0.f, 0.f, &r);
    out[0] = r.val.x; out[1] = r.val.y; out[2] = r.val.z; out[3] = r.pdf;
    out[4] = r.wo.x; out[5] = r.wo.y; out[6] = r.wo.z; out[7] = r.bs_pdf; out[8] = r.bs_eta; out[9] = (float) r.bs_delta;
    out[10] = r.weight.x; out[11] = r.weight.y; out[12] = r.weight.z;
}
/* Sphere::sample_direction: out = p[3], n[3], d[3], dist, pdf */
void orc_kat_sphere_sample_direction(const orc_shape *sh, const float *ref, float s_x, float s_y, float *out) {
    v3 p, n, d; float dist, pdf;
    sphere_sample_direction(sh, V(ref[0], ref[1], ref[2]), s_x, s_y, &p, &n, &d, &dist, &pdf);
    out[0] = p.x; out[1] = p.y; out[2] = p.z; out[3] = n.x; out[4] = n.y; out[5] = n.z; out[6] = d.x; out[7] = d.y; out[8] = d.z;
    out[9] = dist; out[10] = pdf;
}
float orc_kat_shape_area(const orc_shape *sh) { return f_rcp(shape_inv_area(sh)); }
/* ImageBlock::put of one sample with values (rgb, 1) into a crop_w x crop_h x 4 film */
void orc_kat_splat(const orc_sensor *se, float *film, float x, float y, const float *rgb) {
    splat(se, film, x, y, (int) floorf(x), (int) floorf(y), rgb);
}
int orc_kat_solve_quadratic(double a, double b, double c, double *out2) { return solve_quadratic_d(a, b, c, out2, out2 + 1); }

/* ---- Scene::bbox() and the environment emitter's bounding sphere */
typedef struct { float lo[3], hi[3]; } orc_box;
static void box_init(orc_box *b) { for (int i = 0; i < 3; ++i) { b->lo[i] = INFINITY; b->hi[i] = -INFINITY; } }
static void box_add(orc_box *b, v3 p) {
    const float c[3] = { p.x, p.y, p.z };
    for (int i = 0; i < 3; ++i) { if (c[i] < b->lo[i]) b->lo[i] = c[i]; if (c[i] > b->hi[i]) b->hi[i] = c[i]; }
}
/* Rectangle::bbox (rectangle.cpp:115-125), Disk::bbox (disk.cpp:136-146), Sphere::bbox (sphere.cpp:177-182), Mesh::bbox (the vertices) */
static void shape_bbox(const orc_shape *sh, orc_box *b) {
    box_init(b);
    if (sh->kind == ORC_SHAPE_RECT || sh->kind == ORC_SHAPE_DISK) {
        static const float c[4][2] = { { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
        for (int k = 0; k < 4; ++k) box_add(b, m_point(sh->to_world, V(c[k][0], c[k][1], 0.f)));
    } else if (sh->kind == ORC_SHAPE_SPHERE) {
        box_add(b, V(sh->center[0] - sh->radius, sh->center[1] - sh->radius, sh->center[2] - sh->radius));
        box_add(b, V(sh->center[0] + sh->radius, sh->center[1] + sh->radius, sh->center[2] + sh->radius));
    } else if (sh->kind == ORC_SHAPE_CYLINDER) {   /* Cylinder::bbox (cylinder.cpp:166-179): the two end circles */
        v3 x1 = m_vector(sh->to_world, V(1.f, 0.f, 0.f)), x2 = m_vector(sh->to_world, V(0.f, 1.f, 0.f));
        v3 x = V(sqrtf(f_sqr(x1.x) + f_sqr(x2.x)), sqrtf(f_sqr(x1.y) + f_sqr(x2.y)), sqrtf(f_sqr(x1.z) + f_sqr(x2.z)));
        v3 p0 = m_point(sh->to_world, V(0.f, 0.f, 0.f)), p1 = m_point(sh->to_world, V(0.f, 0.f, 1.f));
        box_add(b, v_sub(p0, x)); box_add(b, v_sub(p1, x)); box_add(b, v_add(p0, x)); box_add(b, v_add(p1, x));
    } else for (int32_t i = 0; i < sh->n_vertices; ++i) box_add(b, mesh_pos(sh, (uint32_t) i));
}
void orc_scene_bsphere(const orc_scene *sc, float *out4) {
    orc_box all; box_init(&all);
    for (int32_t i = 0; i < sc->n_objects; ++i) {
        const orc_object *ob = &sc->objects[i];
        orc_box b;
        if (ob->kind == ORC_OBJ_SHAPE) shape_bbox(&sc->shapes[ob->index], &b);
        else {   /* Instance::bbox (instance.cpp:101-114): the group's box under the first and the last keyframe */
            orc_box g; box_init(&g);
            const orc_group *gr = &sc->groups[ob->index];
            for (int32_t k = 0; k < gr->n_shapes; ++k) {
                orc_box cb; shape_bbox(&sc->shapes[gr->first_shape + k], &cb);
                if (cb.lo[0] <= cb.hi[0]) { box_add(&g, V(cb.lo[0], cb.lo[1], cb.lo[2])); box_add(&g, V(cb.hi[0], cb.hi[1], cb.hi[2])); }
            }
            box_init(&b);
            if (g.lo[0] <= g.hi[0]) for (int c = 0; c < 8; ++c) {
                v3 corner = V(c & 1 ? g.hi[0] : g.lo[0], c & 2 ? g.hi[1] : g.lo[1], c & 4 ? g.hi[2] : g.lo[2]);
                box_add(&b, m_point(ob->key[0], corner));
                if (ob->n_keys > 1) box_add(&b, m_point(ob->key[1], corner));
            }
        }
        if (b.lo[0] <= b.hi[0]) { box_add(&all, V(b.lo[0], b.lo[1], b.lo[2])); box_add(&all, V(b.hi[0], b.hi[1], b.hi[2])); }
    }
    if (!(all.lo[0] <= all.hi[0])) { out4[0] = out4[1] = out4[2] = 0.f; out4[3] = 1.f; return; }
    /* BoundingBox::bounding_sphere (bbox.h:330-333): centre = (min + max) * .5, radius = |centre - max| */
    v3 c = V((all.lo[0] + all.hi[0]) * .5f, (all.lo[1] + all.hi[1]) * .5f, (all.lo[2] + all.hi[2]) * .5f);
    float r = v_norm(v_sub(c, V(all.hi[0], all.hi[1], all.hi[2])));
    out4[0] = c.x; out4[1] = c.y; out4[2] = c.z; out4[3] = f_max(ORC_RAY_EPS, r * (1.f + ORC_RAY_EPS));
}
