// dtof_math.h -- float32 building blocks shared by the host set-up code and the HIP
// kernels (everything is compiled by hipcc with -ffp-contract=off, so an fma happens
// exactly where fmaf() is written).
//
// The operation order of every helper mirrors the Dr.Jit primitive the reference
// uses at that point (reference paths relative to the Mitsuba3DopplerToF root):
//   dot       -> fmadd chain            (Frame::to_local, include/mitsuba/core/frame.h:34-36)
//   cross     -> fmsub(a.yzx*b.zxy ...) (interaction.h:267)
//   normalize -> v * rsqrt(dot(v,v))    (rsqrt = sqrt(1/x), the LLVM back end's lowering)
//   xf_point  -> Transform::transform_affine(Point)  include/mitsuba/core/transform.h:97-105
//   xf_vector -> Transform::operator*(Vector)        transform.h:125-134
//   xf_normal -> Transform::operator*(Normal)        transform.h:140-149
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define DTOF_HD __host__ __device__ __forceinline__

namespace dtof {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kRayEps = 1500.f * 5.9604644775390625e-8f;   // include/mitsuba/core/math.h:17-22
constexpr float kShadowEps = kRayEps * 10.f;
constexpr float kLargest = 3.40282346638528859812e+38f;       // dr::Largest<float>

struct V3 { float x, y, z; };

DTOF_HD uint32_t f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; __builtin_memcpy(&u, &f, 4); return u;
#endif
}
DTOF_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; __builtin_memcpy(&f, &u, 4); return f;
#endif
}
DTOF_HD float rcp(float x) { return 1.0f / x; }
DTOF_HD float rsqrt_(float x) { return sqrtf(1.0f / x); }
DTOF_HD float sqr(float x) { return x * x; }
DTOF_HD float mulsign(float a, float b) { return u2f(f2u(a) ^ (f2u(b) & 0x80000000u)); }
DTOF_HD float mulsign_neg(float a, float b) { return u2f(f2u(a) ^ (~f2u(b) & 0x80000000u)); }
DTOF_HD float signf(float x) { return u2f(0x3f800000u | (f2u(x) & 0x80000000u)); }
DTOF_HD float fmin_(float a, float b) { return a < b ? a : b; }
DTOF_HD float fmax_(float a, float b) { return a > b ? a : b; }

DTOF_HD V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DTOF_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DTOF_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DTOF_HD V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
DTOF_HD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
DTOF_HD V3 vfma(V3 a, float s, V3 c) { return mk(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z)); }
DTOF_HD float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
DTOF_HD V3 cross(V3 a, V3 b) {
    return mk(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
DTOF_HD V3 normalize(V3 a) { return a * rsqrt_(dot(a, a)); }
DTOF_HD float norm(V3 a) { return sqrtf(dot(a, a)); }

// 3x4 affine matrix, row-major: m[4*r + c], r < 3
struct M34 { float m[12]; };

DTOF_HD V3 xf_point(const float *m, V3 p) {
    return mk(fmaf(m[2], p.z, fmaf(m[1], p.y, fmaf(m[0], p.x, m[3]))),
              fmaf(m[6], p.z, fmaf(m[5], p.y, fmaf(m[4], p.x, m[7]))),
              fmaf(m[10], p.z, fmaf(m[9], p.y, fmaf(m[8], p.x, m[11]))));
}
DTOF_HD V3 xf_vector(const float *m, V3 v) {
    return mk(fmaf(m[2], v.z, fmaf(m[1], v.y, m[0] * v.x)),
              fmaf(m[6], v.z, fmaf(m[5], v.y, m[4] * v.x)),
              fmaf(m[10], v.z, fmaf(m[9], v.y, m[8] * v.x)));
}
// normal transform with the INVERSE matrix given (inverse_transpose(r,c) = inv(c,r))
DTOF_HD V3 xf_normal(const float *inv, V3 n) {
    return mk(fmaf(inv[8], n.z, fmaf(inv[4], n.y, inv[0] * n.x)),
              fmaf(inv[9], n.z, fmaf(inv[5], n.y, inv[1] * n.x)),
              fmaf(inv[10], n.z, fmaf(inv[6], n.y, inv[2] * n.x)));
}
// Inverse of an affine matrix (the reference builds Transform(Matrix) per hit on an
// instance, transform.h:54-56 via instance.cpp:161-162, and then only uses the affine part).
DTOF_HD void affine_inverse(const float *m, float *inv) {
    float a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
    float c00 = fmaf(a11, a22, -(a12 * a21)), c01 = fmaf(a12, a20, -(a10 * a22)), c02 = fmaf(a10, a21, -(a11 * a20));
    float det = fmaf(a02, c02, fmaf(a01, c01, a00 * c00));
    float id = 1.0f / det;
    float i00 = c00 * id, i01 = fmaf(a02, a21, -(a01 * a22)) * id, i02 = fmaf(a01, a12, -(a02 * a11)) * id;
    float i10 = c01 * id, i11 = fmaf(a00, a22, -(a02 * a20)) * id, i12 = fmaf(a02, a10, -(a00 * a12)) * id;
    float i20 = c02 * id, i21 = fmaf(a01, a20, -(a00 * a21)) * id, i22 = fmaf(a00, a11, -(a01 * a10)) * id;
    float tx = m[3], ty = m[7], tz = m[11];
    inv[0] = i00; inv[1] = i01; inv[2] = i02;  inv[3]  = -fmaf(i02, tz, fmaf(i01, ty, i00 * tx));
    inv[4] = i10; inv[5] = i11; inv[6] = i12;  inv[7]  = -fmaf(i12, tz, fmaf(i11, ty, i10 * tx));
    inv[8] = i20; inv[9] = i21; inv[10] = i22; inv[11] = -fmaf(i22, tz, fmaf(i21, ty, i20 * tx));
}

// Cephes single-precision sincos kernel (what dr::sincos is built on; Dr.Jit's source is
// not in the reference tree).  Only mul/sub/fma/int ops => bit-identical on host and device.
DTOF_HD void sincos_(float x, float &s_out, float &c_out) {
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
    bool poly = (j & 2) == 0;
    s_out = u2f(f2u(poly ? s : c) ^ (sign_sin & 0x80000000u));
    c_out = u2f(f2u(poly ? c : s) ^ (sign_cos & 0x80000000u));
}
DTOF_HD float cos_(float x) { float s, c; sincos_(x, s, c); return c; }

// acos, restated from the Cephes asinf kernel Dr.Jit's dr::acos builds on (its source is not in the reference tree): minimax polynomial
// in x^2 (|x| < 0.5) or in (1 - |x|) / 2 with a square root, evaluated in Estrin form with fmadd.  Used by SpotLight::falloff_curve.
DTOF_HD float acos_(float x) {
    const float xa = fabsf(x), x2 = x * x;
    const bool big = xa >= 0.5f;
    const float x1 = 0.5f * (1.f - xa), x3 = big ? x1 : x2, x4 = big ? sqrtf(x1) : x;
    const float a0 = fmaf(x3, 7.4953002686e-2f, 1.6666752422e-1f), a1 = fmaf(x3, 2.4181311049e-2f, 4.5470025998e-2f), y2 = x3 * x3;
    float z1 = fmaf(y2 * y2, 4.2163199048e-2f, fmaf(y2, a1, a0));
    z1 = fmaf(z1, x3 * x4, x4);
    const float z2 = 2.f * z1, z3 = x < 0.f ? kPi - z2 : z2, z4 = 0.5f * kPi - z1;
    return big ? z3 : z4;
}

// dr::atan2 (Dr.Jit's source is not in the reference tree): minimax fit of atan(sqrt(z)) / sqrt(z) in z = (min / max)^2, Estrin form with
// fmadd, unfolded by octant.  Used by the environment map's direction -> latitude-longitude lookup (envmap.cpp:303-306,414-416).
DTOF_HD float atan2_(float y, float x) {
    const float xa = fabsf(x), ya = fabsf(y), mn = ya < xa ? ya : xa, mx = xa > ya ? xa : ya;
    const float scale = mn / mx, z = scale * scale;
    const float z2 = z * z, z4 = z2 * z2;
    const float p01 = fmaf(z, -0.33326497518773606976f, 0.99999934166683966009f), p23 = fmaf(z, -0.13486708938456973185f, 0.19881342388439013552f);
    const float p45 = fmaf(z, -0.37006525670417265220e-1f, 0.83863120428809689910e-1f), p6 = 0.78613793713198150252e-2f;
    const float poly = fmaf(z4, fmaf(z2, p6, p45), fmaf(z2, p23, p01));
    float t = scale * poly;
    t = ya > xa ? 0.5f * kPi - t : t;
    t = x < 0.f ? kPi - t : t;
    const float r = y < 0.f ? -t : t;
    return mx != 0.f ? r : 0.f;
}
DTOF_HD float lerp_(float a, float b, float t) { return fmaf(b, t, fmaf(-a, t, a)); }   // dr::lerp = fmadd(b, t, fnmadd(a, t, a))

// dr::detail::estrin_impl for 10 coefficients (what GaussianFilter::eval evaluates, src/rfilters/gaussian.cpp:94-96)
DTOF_HD float estrin10(float x, const float *c) {
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    float a0 = fmaf(x, c[1], c[0]), a1 = fmaf(x, c[3], c[2]), a2 = fmaf(x, c[5], c[4]), a3 = fmaf(x, c[7], c[6]), a4 = fmaf(x, c[9], c[8]);
    float b0 = fmaf(x2, a1, a0), b1 = fmaf(x2, a3, a2);
    float c0 = fmaf(x4, b1, b0);
    return fmaf(x8, a4, c0);
}

// coordinate_system -- include/mitsuba/core/vector.h:116-136
DTOF_HD void coordinate_system(V3 n, V3 &s, V3 &t) {
    float sign = signf(n.z), a = -rcp(sign + n.z), b = n.x * n.y * a;
    s = mk(mulsign(sqr(n.x) * a, n.z) + 1.f, mulsign(b, n.z), mulsign_neg(n.x, n.z));
    t = mk(b, fmaf(n.y, n.y * a, sign), -n.y);
}

DTOF_HD float safe_sqrt(float x) { return sqrtf(fmax_(x, 0.f)); }
// fresnel -- include/mitsuba/render/fresnel.h:21-63
DTOF_HD void fresnel_dielectric(float cos_theta_i, float eta, float &r, float &cos_theta_t, float &eta_it, float &eta_ti) {
    const bool outside = cos_theta_i >= 0.f;
    const float rcp_eta = rcp(eta);
    eta_it = outside ? eta : rcp_eta; eta_ti = outside ? rcp_eta : eta;
    const float cos_theta_t_sqr = fmaf(-fmaf(-cos_theta_i, cos_theta_i, 1.f), eta_ti * eta_ti, 1.f);
    const float cos_theta_i_abs = fabsf(cos_theta_i), cos_theta_t_abs = safe_sqrt(cos_theta_t_sqr);
    const bool index_matched = eta == 1.f, special_case = index_matched || cos_theta_i_abs == 0.f;
    const float a_s = fmaf(-eta_it, cos_theta_t_abs, cos_theta_i_abs) / fmaf(eta_it, cos_theta_t_abs, cos_theta_i_abs);
    const float a_p = fmaf(-eta_it, cos_theta_i_abs, cos_theta_t_abs) / fmaf(eta_it, cos_theta_i_abs, cos_theta_t_abs);
    r = 0.5f * (sqr(a_s) + sqr(a_p));
    if (special_case) r = index_matched ? 0.f : 1.f;
    cos_theta_t = mulsign_neg(cos_theta_t_abs, cos_theta_i);
}
// exp / log / tan / erf / erfinv: Dr.Jit's dr::exp, dr::log, dr::tan, dr::erf, dr::erfinv (drjit/math.h) are not in the reference tree.
// Restated from the published single-precision kernels Dr.Jit's math library derives from -- Cephes expf / logf / tanf, the Cephes erff
// series inside |x| < 1 with Abramowitz & Stegun 7.1.26 outside, M. Giles' single-precision erfinv polynomial -- with explicit fmaf, so
// that host, device and the oracle produce the same bits.  Needed by the Beckmann distribution (microfacet.h:176-196,240-290,341-403).
DTOF_HD float exp_(float x) {
    if (x > 88.72283905206835f) return u2f(0x7f800000u);
    if (x < -103.278929903431851103f) return 0.f;
    const float z = floorf(fmaf(1.44269504088896341f, x, 0.5f));
    x = fmaf(z, -0.693359375f, x);
    x = fmaf(z, 2.12194440e-4f, x);
    const int32_t n = (int32_t) z;
    const float x2 = x * x;
    float p = fmaf(1.9875691500e-4f, x, 1.3981999507e-3f);
    p = fmaf(p, x, 8.3334519073e-3f);
    p = fmaf(p, x, 4.1665795894e-2f);
    p = fmaf(p, x, 1.6666665459e-1f);
    p = fmaf(p, x, 5.0000001201e-1f);
    const float r = fmaf(p, x2, x) + 1.f;
    const int32_t n1 = n / 2, n2 = n - n1;          // ldexp in two exact power-of-two factors
    return r * u2f((uint32_t) (n1 + 127) << 23) * u2f((uint32_t) (n2 + 127) << 23);
}
DTOF_HD float log_(float x) {
    if (x < 0.f) return u2f(0x7fc00000u);
    if (x == 0.f) return u2f(0xff800000u);
    if (!(x < u2f(0x7f800000u))) return x;
    uint32_t u = f2u(x); int32_t e = 0;
    if (u < 0x00800000u) { x *= 8388608.f; u = f2u(x); e = -23; }
    e += (int32_t) (u >> 23) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.f; } else m = m - 1.f;
    const float z = m * m;
    float y = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    y = fmaf(y, m, 1.1676998740e-1f);
    y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f);
    y = fmaf(y, m, -1.6668057665e-1f);
    y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f);
    y = fmaf(y, m, 3.3333331174e-1f);
    y = y * m * z;
    const float fe = (float) e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    return fmaf(0.693359375f, fe, m + y);
}
DTOF_HD float tan_(float xx) {
    const float x = fabsf(xx);
    int32_t j = (int32_t) (x * 1.2732395447351626862f);
    j = (j + 1) & ~1;
    const float y = (float) j;
    float z = x - y * 0.78515625f;
    z = z - y * 2.4187564849853515625e-4f;
    z = z - y * 3.77489497744594108e-8f;
    const float zz = z * z;
    float p = fmaf(9.38540185543e-3f, zz, 3.11992232697e-3f);
    p = fmaf(p, zz, 2.44301354525e-2f);
    p = fmaf(p, zz, 5.34112807005e-2f);
    p = fmaf(p, zz, 1.33387994085e-1f);
    p = fmaf(p, zz, 3.33331568548e-1f);
    float r = x > 1.0e-4f ? fmaf(p * zz, z, z) : z;
    if (j & 2) r = -1.f / r;
    return u2f(f2u(r) ^ (f2u(xx) & 0x80000000u));
}
DTOF_HD float erf_(float x) {
    const float xa = fabsf(x);
    if (xa < 1.f) {
        const float z = x * x;
        float p = fmaf(7.853861353153693e-5f, z, -8.010193625184903e-4f);
        p = fmaf(p, z, 5.188327685732524e-3f);
        p = fmaf(p, z, -2.685381193529856e-2f);
        p = fmaf(p, z, 1.128358514861418e-1f);
        p = fmaf(p, z, -3.761262582423300e-1f);
        p = fmaf(p, z, 1.128379165726710e+0f);
        return x * p;
    }
    const float t = 1.f / fmaf(0.3275911f, xa, 1.f);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = fmaf(-(p * t), exp_(-(xa * xa)), 1.f);
    return u2f(f2u(r) | (f2u(x) & 0x80000000u));
}
DTOF_HD float erfinv_(float x) {
    float w = -log_((1.f - x) * (1.f + x)), p;
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
// ---- MicrofacetDistribution (include/mitsuba/render/microfacet.h): Beckmann (type 0) and GGX (type 1).  The BSDF plugins sample
// visible normals (their default, sample_visible = true); sampling all normals exists for the reference's known answers
// (src/render/tests/test_microfacet.py) through dtof_eval_component.
enum { MF_BECKMANN = 0, MF_GGX = 1 };
struct Ggx { float au, av; int type; int visible; };
DTOF_HD Ggx mf_make(int type, float au, float av, int visible = 1) {   // configure() :425-428
    Ggx g; g.au = fmax_(au, 1e-4f); g.av = fmax_(av, 1e-4f); g.type = type; g.visible = visible; return g;
}
DTOF_HD float ggx_eval(Ggx g, V3 m) {   // eval() :176-196
    const float alpha_uv = g.au * g.av, cos_theta_2 = sqr(m.z);
    float result;
    if (g.type == MF_BECKMANN) result = exp_(-(sqr(m.x / g.au) + sqr(m.y / g.av)) / cos_theta_2) / (kPi * alpha_uv * sqr(cos_theta_2));
    else result = rcp(kPi * alpha_uv * sqr(sqr(m.x / g.au) + sqr(m.y / g.av) + sqr(m.z)));
    return result * m.z > 1e-20f ? result : 0.f;
}
DTOF_HD float ggx_smith_g1(Ggx g, V3 v, V3 m) {   // smith_g1() :341-365
    const float xy_alpha_2 = sqr(g.au * v.x) + sqr(g.av * v.y), tan_theta_alpha_2 = xy_alpha_2 / sqr(v.z);
    float result;
    if (g.type == MF_BECKMANN) {
        const float a = rsqrt_(tan_theta_alpha_2), a_sqr = sqr(a);
        result = a >= 1.6f ? 1.f : (3.535f * a + 2.181f * a_sqr) / (1.f + 2.276f * a + 2.577f * a_sqr);
    } else result = 2.f / (1.f + sqrtf(1.f + tan_theta_alpha_2));
    if (xy_alpha_2 == 0.f) result = 1.f;
    if (dot(v, m) * v.z <= 0.f) result = 0.f;
    return result;
}
// pdf() :219-228 (note the association of the visible-normal branch: D * ((G1 * |wi.m|) / cos_theta_i), unlike the density sample() returns)
DTOF_HD float ggx_pdf(Ggx g, V3 wi, V3 m) {
    return g.visible ? ggx_eval(g, m) * (ggx_smith_g1(g, wi, m) * fabsf(dot(wi, m)) / wi.z) : ggx_eval(g, m) * m.z;
}
// warp::square_to_uniform_disk_concentric (include/mitsuba/core/warp.h:54-90)
DTOF_HD void concentric_disk(float s_x, float s_y, float &px, float &py) {
    const float x = fmaf(2.f, s_x, -1.f), y = fmaf(2.f, s_y, -1.f);
    const bool is_zero = x == 0.f && y == 0.f, q13 = fabsf(x) < fabsf(y);
    const float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * kPi * rp / r;
    if (q13) phi = 0.5f * kPi - phi;
    if (is_zero) phi = 0.f;
    float sn, cs; sincos_(phi, sn, cs);
    px = r * cs; py = r * sn;
}
// sample_visible_11 (:368-420): slope of the visible normal for alpha = 1
DTOF_HD void mf_sample_visible_11(int type, float cos_theta_i, float s_x, float s_y, float &slope_x, float &slope_y) {
    if (type == MF_BECKMANN) {
        const float inv_sqrt_pi = 0.56418958354775628695f;
        const float tan_theta_i = safe_sqrt(fmaf(-cos_theta_i, cos_theta_i, 1.f)) / cos_theta_i, cot_theta_i = rcp(tan_theta_i);
        const float maxval = erf_(cot_theta_i);
        s_x = fmax_(fmin_(s_x, 1.f - 1e-6f), 1e-6f); s_y = fmax_(fmin_(s_y, 1.f - 1e-6f), 1e-6f);
        float x = maxval - (maxval + 1.f) * erf_(sqrtf(-log_(s_x)));
        s_x *= 1.f + maxval + inv_sqrt_pi * tan_theta_i * exp_(-sqr(cot_theta_i));
        for (int i = 0; i < 3; ++i) {   // three Newton iterations
            const float slope = erfinv_(x);
            const float value = 1.f + x + inv_sqrt_pi * tan_theta_i * exp_(-sqr(slope)) - s_x, derivative = 1.f - slope * tan_theta_i;
            x -= value / derivative;
        }
        slope_x = erfinv_(x); slope_y = erfinv_(fmaf(2.f, s_y, -1.f));
        return;
    }
    float px, py; concentric_disk(s_x, s_y, px, py);
    const float s = 0.5f * (1.f + cos_theta_i), a = safe_sqrt(1.f - sqr(px));
    py = fmaf(py, s, fmaf(-a, s, a));                              // dr::lerp(a, py, s)
    const float pz = safe_sqrt(1.f - fmaf(py, py, px * px));
    const float sin_theta_i = safe_sqrt(1.f - sqr(cos_theta_i));
    const float norm_ = rcp(fmaf(sin_theta_i, py, cos_theta_i * pz));
    slope_x = fmaf(cos_theta_i, py, -(sin_theta_i * pz)) * norm_; slope_y = px * norm_;
}
// sample() :240-325: microfacet normal and its density
DTOF_HD V3 ggx_sample(Ggx g, V3 wi, float s_x, float s_y, float &pdf) {
    if (!g.visible) {   // all normals :242-290
        float sin_phi, cos_phi, cos_theta, cos_theta_2, alpha_2;
        if (g.au == g.av) {
            sincos_((2.f * kPi) * s_y, sin_phi, cos_phi);
            alpha_2 = g.au * g.au;
        } else {
            const float ratio = g.av / g.au, tmp = ratio * tan_((2.f * kPi) * s_y);
            cos_phi = rsqrt_(fmaf(tmp, tmp, 1.f));
            cos_phi = mulsign(cos_phi, fabsf(s_y - .5f) - .25f);
            sin_phi = cos_phi * tmp;
            alpha_2 = rcp(sqr(cos_phi / g.au) + sqr(sin_phi / g.av));
        }
        if (g.type == MF_BECKMANN) {
            cos_theta = rsqrt_(fmaf(-alpha_2, log_(1.f - s_x), 1.f));
            cos_theta_2 = sqr(cos_theta);
            const float cos_theta_3 = fmax_(cos_theta_2 * cos_theta, 1e-20f);
            pdf = (1.f - s_x) / (kPi * g.au * g.av * cos_theta_3);
        } else {
            const float tan_theta_m_2 = alpha_2 * s_x / (1.f - s_x);
            cos_theta = rsqrt_(1.f + tan_theta_m_2);
            cos_theta_2 = sqr(cos_theta);
            const float temp = 1.f + tan_theta_m_2 / alpha_2, cos_theta_3 = fmax_(cos_theta_2 * cos_theta, 1e-20f);
            pdf = rcp(kPi * g.au * g.av * cos_theta_3 * sqr(temp));
        }
        const float sin_theta = sqrtf(1.f - cos_theta_2);
        return mk(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
    }
    const V3 wi_p = normalize(mk(g.au * wi.x, g.av * wi.y, wi.z));
    const float sin_theta_2 = fmaf(wi_p.x, wi_p.x, sqr(wi_p.y)), inv_sin_theta = rsqrt_(sin_theta_2);   // Frame3f::sincos_phi (frame.h:111-122)
    float rx = fmin_(fmax_(wi_p.x * inv_sin_theta, -1.f), 1.f), ry = fmin_(fmax_(wi_p.y * inv_sin_theta, -1.f), 1.f);
    if (fabsf(sin_theta_2) <= 4.f * 5.9604644775390625e-8f) { rx = 1.f; ry = 0.f; }
    const float sin_phi = ry, cos_phi = rx, cos_theta = wi_p.z;
    float slope_x, slope_y;
    mf_sample_visible_11(g.type, cos_theta, s_x, s_y, slope_x, slope_y);
    const float sx = fmaf(cos_phi, slope_x, -(sin_phi * slope_y)) * g.au, sy = fmaf(sin_phi, slope_x, cos_phi * slope_y) * g.av;
    const V3 m = normalize(mk(-sx, -sy, 1.f));
    pdf = ggx_eval(g, m) * ggx_smith_g1(g, wi, m) * fabsf(dot(wi, m)) / wi.z;
    return m;
}

// ---------------------------------------------------------------- integer helpers
// Division of a 32-bit unsigned by a launch-invariant divisor (Granlund-Montgomery / "round-up" form): the hardware has no integer
// divide and the generic expansion costs ~17 VALU instructions per quotient; this is one v_mul_hi_u32 + 4 simple ops and exact for every
// n < 2^32, d >= 1.  The lane -> pixel / pair / stratum mappings of integrator.cpp:273-285 and correlated.cpp:47-64,112-124 are
// all of this form (the reference leaves them to Dr.Jit's own division-by-opaque-constant code).
struct FastDiv { uint32_t mul, shifts; };   // shifts = sh1 | sh2 << 8
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f; f.mul = 1; f.shifts = 0;
    if (d <= 1) return f;                    // d == 1: q = mulhi(n, 1) = 0, t = n  (d == 0 never divides: callers guard)
    uint32_t L = 0; while ((1ull << L) < d) ++L;      // ceil(log2 d)
    f.mul = (uint32_t) ((((1ull << L) - d) << 32) / d + 1);
    f.shifts = 1u | ((L - 1) << 8);
    return f;
}
DTOF_HD uint32_t fdiv(uint32_t n, FastDiv d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t q = __umulhi(n, d.mul);
#else
    const uint32_t q = (uint32_t) (((uint64_t) n * d.mul) >> 32);
#endif
    return (((n - q) >> (d.shifts & 0xffu)) + q) >> (d.shifts >> 8);
}

// fmodf(x, y) for y > 0, exact like the C library's (the result of fmod is always representable): for |x| < 2^22 y the quotient
// estimate trunc(|x| * (1 / y)) is off by at most one, the remainder fma(-q, y, |x|) of an off-by-one quotient has the sign / size that
// tells which way, and the remainder of the right quotient is exact.  Everything else (huge, NaN, inf) takes the library path.
// ocml's generic fmodf is ~70 VALU instructions; eval_modulation_weight calls it once or twice per path vertex
// (waveform_utils.h:24-62 via dopplertofpath.cpp:60-77).
DTOF_HD float fmod_pos(float x, float y, float inv_y) {
    const float ax = fabsf(x);
    if (!(ax < 4194304.f * y)) return fmodf(x, y);
    float q = truncf(ax * inv_y), r = fmaf(-q, y, ax);
    if (r < 0.f) { q -= 1.f; r = fmaf(-q, y, ax); }
    else if (r >= y) { q += 1.f; r = fmaf(-q, y, ax); }
    return u2f(f2u(r) | (f2u(x) & 0x80000000u));
}

// ---------------------------------------------------------------- RNG (integer exact)
// sample_tea_32 -- include/mitsuba/core/random.h:33-47
DTOF_HD void tea32(uint32_t v0, uint32_t v1, uint32_t &o0, uint32_t &o1) {
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    o0 = v0; o1 = v1;
}
// dr::PCG32 (Dr.Jit 0.4.0, PCG-XSH-RR 64/32)
constexpr uint64_t kPcgMult = 0x5851f42d4c957f2dULL;
DTOF_HD uint32_t pcg_next_u32(uint64_t &state, uint64_t inc) {
    uint64_t old = state;
    state = old * kPcgMult + inc;
    uint32_t xs = (uint32_t) (((old >> 18) ^ old) >> 27), rot = (uint32_t) (old >> 59);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
DTOF_HD float pcg_next_f32(uint64_t &state, uint64_t inc) {
    return u2f((pcg_next_u32(state, inc) >> 9) | 0x3f800000u) - 1.f;
}
// the float draw that belongs to the state `old` the generator was in before its step
DTOF_HD float pcg_output_f32(uint64_t old) {
    uint32_t xs = (uint32_t) (((old >> 18) ^ old) >> 27), rot = (uint32_t) (old >> 59);
    return u2f((((xs >> rot) | (xs << ((~rot + 1u) & 31))) >> 9) | 0x3f800000u) - 1.f;
}
// six LCG steps at once: state * M^6 + inc * (M^5 + M^4 + M^3 + M^2 + M + 1)  (mod 2^64)
constexpr uint64_t pcg_pow(int n) { uint64_t r = 1; for (int i = 0; i < n; ++i) r *= kPcgMult; return r; }
constexpr uint64_t kPcgMult6 = pcg_pow(6), kPcgGeom6 = 1 + pcg_pow(1) + pcg_pow(2) + pcg_pow(3) + pcg_pow(4) + pcg_pow(5);
DTOF_HD uint64_t pcg_jump6(uint64_t state, uint64_t inc) { return state * kPcgMult6 + inc * kPcgGeom6; }
// PCG32::seed(1, initstate, initseq)
DTOF_HD void pcg_seed(uint32_t initstate, uint32_t initseq, uint64_t &state, uint64_t &inc) {
    state = 0; inc = ((uint64_t) initseq << 1) | 1u;
    pcg_next_u32(state, inc);
    state += (uint64_t) initstate;
    pcg_next_u32(state, inc);
}
// permute_kensler -- random.h:113-171
DTOF_HD uint32_t permute_kensler(uint32_t index, uint32_t n, uint32_t seed, FastDiv dn) {   // dn = make_fastdiv(n)
    if (n <= 1) return 0;   // n == 0 (sample_count < time_correlate_number) would never leave the cycle-walking loop below
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
    const uint32_t v = index + seed;
    return v - n * fdiv(v, dn);
}

}  // namespace dtof
