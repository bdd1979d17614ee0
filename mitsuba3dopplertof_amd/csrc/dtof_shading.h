#pragma once
// dtof_shading.h -- device side of a surface interaction: compute_surface (rectangle / mesh / sphere, through instances), ray
// spawning, emitter sampling helpers (mesh and sphere area lights) and the evaluation helpers of the microfacet BSDFs.
#include "dtof_traverse.h"

#ifndef DTOF_D
#define DTOF_D __device__ __forceinline__
#endif

namespace dtof {

struct Surface { V3 p, n, sh_n, sh_s, sh_t, wi; float u, v; const DShape *shape; V3 dp_du, dp_dv; };   // u, v = si.uv (rectangles and meshes); dp_du, dp_dv in world space (read by normalmap / bumpmap frames only)

// Shape::compute_surface_interaction for rectangle (rectangle.cpp:250-323) / mesh (mesh.cpp:632-864),
// Instance::compute_surface_interaction (instance.cpp:155-250), finalize (interaction.h:493-513)
// memo_m / memo_inv: the instance matrix and its inverse of object sv.memo_obj at this ray time, if the caller has them (instance memo)
template <bool MESH>
DTOF_D void compute_surface(const SceneView &sv, uint32_t oi, uint32_t shape_k, uint32_t prim, float t, float b1, float b2,
                            V3 o, V3 d, float time, Surface &si, bool use_memo, const float (&memo_m)[12], const float (&memo_inv)[12]) {
    const DObject &ob = sv.objects[oi];
    bool inst = ob.kind == OBJ_INSTANCE;
    float m[12], inv[12];
    V3 lo = o, ld = d;
    const DShape *sh;
    if (inst) {
        if (use_memo && oi == sv.memo_obj) {
            if (sv.memo_m) { instance_memo_load_matrix(sv, m); instance_memo_load(sv, inv); }   // both sit in the lane's LDS column since the lane was generated
            else {
#pragma unroll
                for (int i = 0; i < 12; ++i) { m[i] = memo_m[i]; inv[i] = memo_inv[i]; }
            }
        } else {
            instance_matrix(ob, time, m);
            affine_inverse(m, inv);
        }
        lo = xf_point(inv, o); ld = xf_vector(inv, d);
        sh = &sv.shapes[sv.groups[ob.index].first_shape + shape_k];
    } else sh = &sv.shapes[ob.index];
    si.shape = sh;
    si.u = si.v = 0.f;
    V3 dp_du, dp_dv;
    if (!MESH || sh->kind == SHAPE_RECT) {
        si.u = fmaf(b1, .5f, .5f); si.v = fmaf(b2, .5f, .5f);   // rectangle.cpp:312-313 (prim_uv = the local hit position)
        V3 n = mk(sh->n[0], sh->n[1], sh->n[2]);
        V3 p = vfma(ld, t, lo);
        V3 tr = mk(sh->to_world[3], sh->to_world[7], sh->to_world[11]);
        float dist = dot(tr - p, n);
        si.p = p + n * dist; si.n = n; si.sh_n = n;
        dp_du = mk(sh->dp_du[0], sh->dp_du[1], sh->dp_du[2]);
        dp_dv = mk(sh->dp_dv[0], sh->dp_dv[1], sh->dp_dv[2]);
    } else if (sh->kind == SHAPE_DISK) {   // Disk::compute_surface_interaction (disk.cpp:305-336); prim_uv = the local hit position
        V3 n = mk(sh->n[0], sh->n[1], sh->n[2]);
        V3 p = vfma(ld, t, lo);
        V3 tr = mk(sh->to_world[3], sh->to_world[7], sh->to_world[11]);
        float dist = dot(tr - p, n);
        si.p = p + n * dist; si.n = n; si.sh_n = n;
        const float r = sqrtf(fmaf(b2, b2, b1 * b1)), inv_r = rcp(r);
        const float cos_phi = r != 0.f ? b1 * inv_r : 1.f, sin_phi = r != 0.f ? b2 * inv_r : 0.f;
        dp_du = xf_vector(sh->to_world, mk(cos_phi, sin_phi, 0.f));
        dp_dv = xf_vector(sh->to_world, mk(-sin_phi, cos_phi, 0.f));
    } else if (sh->kind == SHAPE_CYLINDER) {   // Cylinder::compute_surface_interaction (cylinder.cpp:395-500, non-diff branch; its normal shift adds
        const V3 p = vfma(ld, t, lo);            // a zero vector: si.n is still unset where it runs, :471-475)
        const V3 local = xf_point(sh->to_object, p);
        dp_du = xf_vector(sh->to_world, mk(-local.y, local.x, 0.f) * (2.f * kPi));
        dp_dv = xf_vector(sh->to_world, mk(0.f, 0.f, 1.f));
        V3 n = normalize(cross(dp_du, dp_dv));
        if (sh->flags & SF_FLIP_NORMALS) n = -n;
        si.p = p; si.n = n; si.sh_n = n;
    } else if (sh->kind == SHAPE_SPHERE) {   // Sphere::compute_surface_interaction (sphere.cpp:509-513, 527-551)
        const V3 c = mk(sh->n[0], sh->n[1], sh->n[2]); const float radius = sh->dp_du[0];
        V3 n = normalize(vfma(ld, t, lo) - c);
        si.p = vfma(n, radius, c);
        const V3 local = xf_point(sh->to_object, si.p);
        const float rd = sqrtf(sqr(local.x) + sqr(local.y)), inv_rd = rcp(rd);
        V3 dpv = mk(local.z * (local.x * inv_rd), local.z * (local.y * inv_rd), -rd);
        if (rd == 0.f) dpv = mk(1.f, 0.f, 0.f);
        dp_du = xf_vector(sh->to_world, mk(-local.y, local.x, 0.f)) * (2.f * kPi);
        dp_dv = xf_vector(sh->to_world, dpv) * kPi;
        if (sh->flags & SF_FLIP_NORMALS) n = -n;
        si.n = n; si.sh_n = n;
    } else {
        const DTri &tr = sv.tris[sh->first_tri + prim];
        const DTriShade &ts = sv.shading[sh->first_tri + prim];
        V3 p0 = mk(tr.p0[0], tr.p0[1], tr.p0[2]), p1 = mk(tr.p1[0], tr.p1[1], tr.p1[2]), p2 = mk(tr.p2[0], tr.p2[1], tr.p2[2]);
        float b0 = 1.f - b1 - b2;
        V3 dp0 = p1 - p0, dp1 = p2 - p0;
        si.p = vfma(p0, b0, vfma(p1, b1, p2 * b2));
        si.u = b1; si.v = b2;                                    // mesh.cpp:720-737
        if (sh->flags & SF_TEXCOORDS) { si.u = fmaf(ts.uv2[0], b2, fmaf(ts.uv1[0], b1, ts.uv0[0] * b0)); si.v = fmaf(ts.uv2[1], b2, fmaf(ts.uv1[1], b1, ts.uv0[1] * b0)); }
        si.n = normalize(cross(dp0, dp1));
        coordinate_system(si.n, dp_du, dp_dv);
        float d0x = ts.uv1[0] - ts.uv0[0], d0y = ts.uv1[1] - ts.uv0[1], d1x = ts.uv2[0] - ts.uv0[0], d1y = ts.uv2[1] - ts.uv0[1];
        float det = fmaf(d0x, d1y, -(d0y * d1x)), inv_det = rcp(det);
        if (det != 0.f) {
            dp_du = mk(fmaf(d1y, dp0.x, -(d0y * dp1.x)), fmaf(d1y, dp0.y, -(d0y * dp1.y)), fmaf(d1y, dp0.z, -(d0y * dp1.z))) * inv_det;
            dp_dv = mk(fmaf(-d1x, dp0.x, d0x * dp1.x), fmaf(-d1x, dp0.y, d0x * dp1.y), fmaf(-d1x, dp0.z, d0x * dp1.z)) * inv_det;
        }
        if (!(sh->flags & SF_FACE_NORMALS)) {
            V3 n0 = mk(ts.n0[0], ts.n0[1], ts.n0[2]), n1 = mk(ts.n1[0], ts.n1[1], ts.n1[2]), n2 = mk(ts.n2[0], ts.n2[1], ts.n2[2]);
            V3 n = vfma(n2, b2, vfma(n1, b1, n0 * b0));
            si.sh_n = n * rsqrt_(dot(n, n));
        } else si.sh_n = si.n;
        if (sh->flags & SF_FLIP_NORMALS) { si.n = -si.n; si.sh_n = -si.sh_n; }
    }
    if (inst) {
        si.p = xf_point(m, si.p);
        si.n = normalize(xf_normal(inv, si.n));
        si.sh_n = normalize(xf_normal(inv, si.sh_n));
        dp_du = xf_vector(m, dp_du); dp_dv = xf_vector(m, dp_dv);   // instance.cpp:201-202
    }
    // initialize_sh_frame (interaction.h:258-268)
    V3 s = normalize(vfma(si.sh_n, -dot(si.sh_n, dp_du), dp_du));
    if (dp_du.x == 0.f && dp_du.y == 0.f && dp_du.z == 0.f) { V3 tt; coordinate_system(si.sh_n, s, tt); }
    si.sh_s = s; si.sh_t = cross(si.sh_n, s);
    si.dp_du = dp_du; si.dp_dv = dp_dv;
    V3 md = -d;
    si.wi = mk(dot(md, si.sh_s), dot(md, si.sh_t), dot(md, si.sh_n));
}
template <bool MESH>
DTOF_D void compute_surface(const SceneView &sv, uint32_t oi, uint32_t shape_k, uint32_t prim, float t, float b1, float b2,
                            V3 o, V3 d, float time, Surface &si) {
    const float none[12] = { 0 };
    compute_surface<MESH>(sv, oi, shape_k, prim, t, b1, b2, o, d, time, si, false, none, none);
}
// Interaction::offset_p (interaction.h:161-165)
DTOF_D V3 offset_p(const Surface &si, V3 d) {
    float mag = (1.f + fmax_(fmax_(fabsf(si.p.x), fabsf(si.p.y)), fabsf(si.p.z))) * kRayEps;
    mag = mulsign(mag, dot(si.n, d));
    return vfma(si.n, mag, si.p);
}
// warp::square_to_cosine_hemisphere (warp.h:54-86, 320-344)
DTOF_D V3 cosine_hemisphere(float sx, float sy) {
    float x = fmaf(2.f, sx, -1.f), y = fmaf(2.f, sy, -1.f);
    bool is_zero = x == 0.f && y == 0.f, q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * kPi * rp / r;
    if (q13) phi = 0.5f * kPi - phi;
    if (is_zero) phi = 0.f;
    float s, c; sincos_(phi, s, c);
    float px = r * c, py = r * s;
    return mk(px, py, sqrtf(fmax_(1.f - fmaf(py, py, px * px), 0.f)));
}
// warp::square_to_uniform_triangle (warp.h:153-156) and warp::square_to_uniform_sphere (warp.h:250-255)
DTOF_D void uniform_triangle(float s_x, float s_y, float &bx, float &by) {
    const float t = sqrtf(fmax_(1.f - s_x, 0.f));
    bx = 1.f - t; by = t * s_y;
}
DTOF_D V3 uniform_sphere(float s_x, float s_y) {
    const float z = fmaf(-2.f, s_y, 1.f), r = safe_sqrt(fmaf(-z, z, 1.f)); float sn, cs;
    sincos_(2.f * kPi * s_x, sn, cs);
    return mk(r * cs, r * sn, z);
}
// Mesh::sample_position (mesh.cpp:513-568): face by DiscreteDistribution::sample_reuse on sample.y (distr_1d.h:113-160,
// dr::binary_search over [m_valid.x, m_valid.y]), point by warp::square_to_uniform_triangle (warp.h:153-156), normal from
// the vertex normals if the mesh has them.
DTOF_D void mesh_sample_position(const SceneView &sv, const DShape &es, float s_x, float s_y, V3 &p, V3 &n) {
    const float *cdf = (const float *) (sv.base + es.emit_table), *pmf = cdf + es.n_tris;
    const uint32_t *slot = (const uint32_t *) (pmf + es.n_tris);
    const float v = s_y * es.emit_sum;
    uint32_t lo = es.emit_lo, hi = es.emit_hi;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] < v) lo = mid + 1 < hi ? mid + 1 : hi; else hi = mid;
    }
    const float pm = pmf[lo] * es.inv_area, cd = lo > 0 ? cdf[lo - 1] * es.inv_area : 0.f;
    const float y = (s_y - cd) / pm;
    const uint32_t k = es.first_tri + slot[lo];
    const DTri &tr = sv.tris[k];
    V3 p0 = mk(tr.p0[0], tr.p0[1], tr.p0[2]), p1 = mk(tr.p1[0], tr.p1[1], tr.p1[2]), p2 = mk(tr.p2[0], tr.p2[1], tr.p2[2]);
    V3 e0 = p1 - p0, e1 = p2 - p0;
    float bx, by; uniform_triangle(s_x, y, bx, by);
    p = vfma(e0, bx, vfma(e1, by, p0));
    if (!(es.flags & SF_FACE_NORMALS)) {
        const DTriShade &ts = sv.shading[k];
        V3 n0 = mk(ts.n0[0], ts.n0[1], ts.n0[2]), n1 = mk(ts.n1[0], ts.n1[1], ts.n1[2]), n2 = mk(ts.n2[0], ts.n2[1], ts.n2[2]);
        n = vfma(n0, 1.f - bx - by, vfma(n1, bx, n2 * by));
    } else n = cross(e0, e1);
    n = normalize(n);
    if (es.flags & SF_FLIP_NORMALS) n = -n;
}

constexpr float kInvTwoPi = 0.15915494309189533577f;
constexpr float kInvFourPi = 0.07957747154594766788f;   // warp::square_to_uniform_sphere_pdf (warp.h:257-266)
DTOF_D float uniform_cone_pdf(float cos_cutoff) { return kInvTwoPi / (1.f - cos_cutoff); }   // warp::square_to_uniform_cone_pdf (warp.h:475-485)
// Sphere::sample_direction (sphere.cpp:222-296): cone sampling of the visible cap from outside, uniform sphere from inside
DTOF_D void sphere_sample_direction(const DShape &sh, V3 ref, float s_x, float s_y, V3 &p, V3 &n, V3 &dd, float &dist, float &pdf) {
    const V3 center = mk(sh.n[0], sh.n[1], sh.n[2]); const float radius = sh.dp_du[0];
    const bool flip = sh.flags & SF_FLIP_NORMALS;
    const V3 dc_v = center - ref;
    const float dc_2 = dot(dc_v, dc_v), radius_adj = radius * (flip ? (1.f + kRayEps) : (1.f - kRayEps));
    const bool outside = dc_2 > sqr(radius_adj);
    V3 dloc;
    if (outside) {
        const float inv_dc = rsqrt_(dc_2), sin_theta_max = radius * inv_dc, sin_theta_max_2 = sqr(sin_theta_max),
                    inv_sin_theta_max = rcp(sin_theta_max), cos_theta_max = safe_sqrt(1.f - sin_theta_max_2);
        const float sin_theta_2 = sin_theta_max_2 > 0.00068523f ? 1.f - sqr(fmaf(cos_theta_max - 1.f, s_x, 1.f)) : sin_theta_max_2 * s_x;
        const float cos_theta = safe_sqrt(1.f - sin_theta_2);
        const float cos_alpha = sin_theta_2 * inv_sin_theta_max + cos_theta * safe_sqrt(fmaf(-sin_theta_2, sqr(inv_sin_theta_max), 1.f));
        const float sin_alpha = safe_sqrt(fmaf(-cos_alpha, cos_alpha, 1.f));
        float sin_phi, cos_phi; sincos_(s_y * (2.f * kPi), sin_phi, cos_phi);
        const V3 fn = dc_v * -inv_dc; V3 fs, ft;
        coordinate_system(fn, fs, ft);
        dloc = vfma(fn, cos_alpha, vfma(ft, sin_phi * sin_alpha, fs * (cos_phi * sin_alpha)));
        pdf = uniform_cone_pdf(cos_theta_max);
    } else {   // warp::square_to_uniform_sphere (warp.h:250-255)
        dloc = uniform_sphere(s_x, s_y);
        pdf = 0.f;
    }
    p = vfma(dloc, radius, center); dd = p - ref;
    const float dist2 = dot(dd, dd);
    dist = sqrtf(dist2);
    dd = dd * rcp(dist);
    if (outside) { if (dist == 0.f) pdf = 0.f; }
    else pdf = sh.inv_area * dist2 / fabsf(dot(dd, dloc));
    n = flip ? -dloc : dloc;
}
// Sphere::pdf_direction (sphere.cpp:298-310)
DTOF_D float sphere_pdf_direction(const DShape &sh, V3 ref, V3 ds_d, V3 ds_n, float ds_dist) {
    const V3 center = mk(sh.n[0], sh.n[1], sh.n[2]);
    const float sin_alpha = sh.dp_du[0] * rcp(norm(center - ref)), cos_alpha = safe_sqrt(1.f - sin_alpha * sin_alpha);
    return sin_alpha < 0.99999994f ? uniform_cone_pdf(cos_alpha) : sh.inv_area * sqr(ds_dist) / fabsf(dot(ds_d, ds_n));
}
// Textures on the diffuse reflectance: Checkerboard::eval (src/textures/checkerboard.cpp:70-89), BitmapTexture::eval -> interpolate_3 / _1
// (src/textures/bitmap.cpp:633-670) -> dr::Texture<Float, 2>::eval (Dr.Jit 0.4.0 texture.h, not in the tree; restated: texel centres at
// (i + .5) / res, the four neighbours wrapped per mode, weights combined as fmadd(w0.y, fmadd(w0.x, v00, w1.x * v10), w1.y * fmadd(...)))
DTOF_D int32_t tex_wrap(int32_t i, int32_t n, uint32_t mode) {
    if (mode == 2) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    int32_t div = i / n; if (i % n < 0) --div;
    int32_t mod = i - div * n;
    if (mode == 1 && (div & 1)) mod = n - 1 - mod;
    return mod;
}
DTOF_D V3 texture_eval(const SceneView &sv, uint32_t rec_off, float u, float v) {
    const DTexture &tex = *(const DTexture *) (sv.base + rec_off);
    const float tu = fmaf(tex.to_uv[1], v, fmaf(tex.to_uv[0], u, 0.f)), tv = fmaf(tex.to_uv[3], v, fmaf(tex.to_uv[2], u, 0.f));
    const uint32_t kind = tex.kind_flags & 0xffu, filter = (tex.kind_flags >> 8) & 0xffu, wrap = (tex.kind_flags >> 16) & 0xffu, C = tex.kind_flags >> 24;
    if (kind == TEX_CHECKERBOARD) {
        const bool mx = tu - floorf(tu) > .5f, my = tv - floorf(tv) > .5f;
        return mx == my ? mk(tex.color0[0], tex.color0[1], tex.color0[2]) : mk(tex.color1[0], tex.color1[1], tex.color1[2]);
    }
    const int32_t W = (int32_t) tex.width, H = (int32_t) tex.height;
    const float *data = (const float *) (sv.base + tex.data_off);
    float texel[3] = { 0.f, 0.f, 0.f };
    if (filter == 0) {
        const int32_t x = tex_wrap((int32_t) floorf(tu * (float) W), W, wrap), y = tex_wrap((int32_t) floorf(tv * (float) H), H, wrap);
        for (uint32_t c = 0; c < C; ++c) texel[c] = data[((size_t) y * W + x) * C + c];
    } else {
        const float px = fmaf(tu, (float) W, -.5f), py = fmaf(tv, (float) H, -.5f), fx = floorf(px), fy = floorf(py);
        const float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
        const int32_t x0 = tex_wrap((int32_t) fx, W, wrap), x1 = tex_wrap((int32_t) fx + 1, W, wrap);
        const int32_t y0 = tex_wrap((int32_t) fy, H, wrap), y1 = tex_wrap((int32_t) fy + 1, H, wrap);
        for (uint32_t c = 0; c < C; ++c) {
            const float v00 = data[((size_t) y0 * W + x0) * C + c], v10 = data[((size_t) y0 * W + x1) * C + c];
            const float v01 = data[((size_t) y1 * W + x0) * C + c], v11 = data[((size_t) y1 * W + x1) * C + c];
            texel[c] = fmaf(w0y, fmaf(w0x, v00, w1x * v10), w1y * fmaf(w0x, v01, w1x * v11));
        }
    }
    if (C == 1) texel[1] = texel[2] = texel[0];
    return mk(texel[0], texel[1], texel[2]);
}
// Texture::eval_1 (bitmap.cpp:324-344: one channel as it is, three channels -> luminance, spectrum.h:431-434; checkerboard.cpp:91-110 with constant
// colours: the mean of the colour the lookup picks, srgb.cpp:85-88)
DTOF_D float texture_eval_1(const SceneView &sv, uint32_t rec_off, float u, float v) {
    const DTexture &tex = *(const DTexture *) (sv.base + rec_off);
    const V3 c = texture_eval(sv, rec_off, u, v);
    if ((tex.kind_flags & 0xffu) == TEX_CHECKERBOARD) {
        const bool first = c.x == tex.color0[0] && c.y == tex.color0[1] && c.z == tex.color0[2];
        const float *k = first ? tex.color0 : tex.color1;
        return ((k[0] + k[1]) + k[2]) * (1.0f / 3.0f);
    }
    if ((tex.kind_flags >> 24) == 1u) return c.x;
    return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f;
}
// DiscreteDistribution2D over a bitmap's texels (include/mitsuba/core/distr_2d.h:75-181; tables built by scene_build.cpp): sample = row from the marginal, column
// from the conditional CDF (dr::binary_search over [0, n - 1]: the first index whose CDF value is not below the sample, the last index if there is none) and
// the re-uniformised variate of both; pdf = the texel's share
DTOF_D uint32_t cdf_search(const float *cdf, uint32_t n, float x) {
    uint32_t lo = 0, hi = n - 1u;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cdf[mid] < x) lo = mid + 1u; else hi = mid; }
    return lo;
}
DTOF_D float interval_to_tent(float s) {   // warp.h:196-200
    s -= .5f;
    const float v = fmaf(fabsf(s), -2.f, 1.f), r = 1.f - (v > 0.f ? sqrtf(v) : 0.f);
    return mulsign(r, s);   // copysign(r, s): r >= 0
}
// Texture::sample_position (texture.cpp:56-59: the identity for every texture without its own) / BitmapTexture::sample_position (bitmap.cpp:450-487)
DTOF_D void texture_sample_position(const SceneView &sv, uint32_t rec_off, float sx, float sy, float &u, float &v, float &pdf) {
    const DTexture &tex = *(const DTexture *) (sv.base + rec_off);
    if ((tex.kind_flags & 0xffu) != TEX_BITMAP || !tex.distr_off) { u = sx; v = sy; pdf = 1.f; return; }
    const uint32_t filter = (tex.kind_flags >> 8) & 0xffu, wrap = (tex.kind_flags >> 16) & 0xffu, W = tex.width, H = tex.height;
    const float *d = (const float *) (sv.base + tex.distr_off), *marg = d + 2, *cond = d + 2 + H;
    sx = fmin_(fmax_(sx, 1.17549435e-38f), 0.99999994f); sy = fmin_(fmax_(sy, 1.17549435e-38f), 0.99999994f);   // clamp(sample, Smallest, OneMinusEpsilon)
    sy *= d[1];
    const uint32_t row = cdf_search(marg, H, sy), offset = row * W;
    sx *= cond[offset + W - 1u];
    const uint32_t col = cdf_search(cond + offset, W, sx);
    const float col_cdf_0 = col > 0 ? cond[offset + col - 1u] : 0.f, col_cdf_1 = cond[offset + col];
    const float row_cdf_0 = row > 0 ? marg[row - 1u] : 0.f, row_cdf_1 = marg[row];
    sx -= col_cdf_0; sy -= row_cdf_0;
    if (col_cdf_1 != col_cdf_0) sx /= col_cdf_1 - col_cdf_0;
    if (row_cdf_1 != row_cdf_0) sy /= row_cdf_1 - row_cdf_0;
    const float p = (col_cdf_1 - col_cdf_0) * d[0];
    const float iw = rcp((float) W), ih = rcp((float) H);
    float x, y;
    if (filter == 0) { x = ((float) col + sx) * iw; y = ((float) row + sy) * ih; }
    else {
        x = (((float) col + .5f) + interval_to_tent(sx)) * iw; y = (((float) row + .5f) + interval_to_tent(sy)) * ih;
        if (wrap == 0) { if (x < 0.f) x += 1.f; if (x > 1.f) x -= 1.f; if (y < 0.f) y += 1.f; if (y > 1.f) y -= 1.f; }
        else { if (x < 0.f) x = -x; if (x > 1.f) x = 2.f - x; if (y < 0.f) y = -y; if (y > 1.f) y = 2.f - y; }
    }
    u = x; v = y; pdf = p * (float) (int32_t) (W * H);
}
// Texture::pdf_position (texture.cpp:61-64) / BitmapTexture::pdf_position (bitmap.cpp:489-528)
DTOF_D float texture_pdf_position(const SceneView &sv, uint32_t rec_off, float u, float v) {
    const DTexture &tex = *(const DTexture *) (sv.base + rec_off);
    if ((tex.kind_flags & 0xffu) != TEX_BITMAP || !tex.distr_off) return 1.f;
    const uint32_t filter = (tex.kind_flags >> 8) & 0xffu, wrap = (tex.kind_flags >> 16) & 0xffu;
    const int32_t W = (int32_t) tex.width, H = (int32_t) tex.height;
    const float *d = (const float *) (sv.base + tex.distr_off), *cond = d + 2 + H;
    auto texel_pdf = [&](int32_t x, int32_t y) { const uint32_t i = (uint32_t) x + (uint32_t) y * (uint32_t) W; return (cond[i] - (x > 0 ? cond[i - 1u] : 0.f)) * d[0]; };
    if (filter == 0) return texel_pdf(tex_wrap((int32_t) floorf(u * (float) W), W, wrap), tex_wrap((int32_t) floorf(v * (float) H), H, wrap)) * (float) (W * H);
    const float px = fmaf(u, (float) W, -.5f), py = fmaf(v, (float) H, -.5f), fx = floorf(px), fy = floorf(py);
    const float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    const int32_t x0 = tex_wrap((int32_t) fx, W, wrap), x1 = tex_wrap((int32_t) fx + 1, W, wrap);
    const int32_t y0 = tex_wrap((int32_t) fy, H, wrap), y1 = tex_wrap((int32_t) fy + 1, H, wrap);
    const float v0 = fmaf(w0x, texel_pdf(x0, y0), w1x * texel_pdf(x1, y0)), v1 = fmaf(w0x, texel_pdf(x0, y1), w1x * texel_pdf(x1, y1));
    return fmaf(w0y, v0, w1y * v1) * (float) (W * H);
}
// Rectangle::eval_parameterization (rectangle.cpp:173-192): the point of the rectangle at (u, v), found by a ray from one normal length above it straight down --
// through the rectangle's own intersection routine and surface interaction, whose roundings the point and its uv then carry.  Not for instanced rectangles.
DTOF_D bool rect_eval_parameterization(const DShape &sh, float u, float v, V3 &p, V3 &n, float &si_u, float &si_v, float &area_norm) {
    const V3 pw = xf_point(sh.to_world, mk(u * 2.f - 1.f, v * 2.f - 1.f, 0.f));
    n = mk(sh.n[0], sh.n[1], sh.n[2]);
    const V3 o = pw + n, d = -n;
    float t, b1, b2;
    if (!rect_hit(sh, o, d, kLargest, t, b1, b2)) return false;
    const V3 ph = vfma(d, t, o), tr = mk(sh.to_world[3], sh.to_world[7], sh.to_world[11]);
    p = ph + n * dot(tr - ph, n);                                                       // rectangle.cpp:289-294
    si_u = fmaf(b1, .5f, .5f); si_v = fmaf(b2, .5f, .5f);
    area_norm = norm(cross(mk(sh.dp_du[0], sh.dp_du[1], sh.dp_du[2]), mk(sh.dp_dv[0], sh.dp_dv[1], sh.dp_dv[2])));
    return true;
}
// The material parameters of one hit: the shape's constants, or the lookups of the textures bound to their slots (m_specular_reflectance->eval(si),
// m_alpha_u->eval_1(si), ...)
struct HitMaterial { float spec_refl[3], spec_trans[3], alpha_u, alpha_v; };
DTOF_D HitMaterial material_at(const SceneView &sv, const DShape *sh, float u, float v) {
    HitMaterial m;
#pragma unroll
    for (int i = 0; i < 3; ++i) { m.spec_refl[i] = sh->spec_refl[i]; m.spec_trans[i] = sh->spec_trans[i]; }
    m.alpha_u = sh->alpha_u; m.alpha_v = sh->alpha_v;
    if (sh->tex_spec) { const V3 c = texture_eval(sv, sh->tex_spec << 4, u, v); m.spec_refl[0] = c.x; m.spec_refl[1] = c.y; m.spec_refl[2] = c.z; }
    if (sh->tex_trans) { const V3 c = texture_eval(sv, sh->tex_trans << 4, u, v); m.spec_trans[0] = c.x; m.spec_trans[1] = c.y; m.spec_trans[2] = c.z; }
    if (sh->tex_alpha_u) m.alpha_u = texture_eval_1(sv, sh->tex_alpha_u << 4, u, v);
    if (sh->tex_alpha_v) m.alpha_v = texture_eval_1(sv, sh->tex_alpha_v << 4, u, v);
    return m;
}
// NormalMap::frame (src/bsdfs/normalmap.cpp:181-189): the frame the nested BSDF is evaluated in, from the RGB texture at the hit: n = normalize(2 c - 1),
// s = normalize(dp_du - n (n . dp_du)) with the interaction's dp_du AS IT IS (world space, as the reference writes it), t = n x s
struct LocalFrame { V3 s, t, n; };
DTOF_D LocalFrame normalmap_frame(const SceneView &sv, const DShape *sh, const Surface &si) {
    const V3 c = texture_eval(sv, sh->tex_normal << 4, si.u, si.v);                 // m_normalmap->eval_3(si)
    LocalFrame f;
    f.n = normalize(mk(fmaf(c.x, 2.f, -1.f), fmaf(c.y, 2.f, -1.f), fmaf(c.z, 2.f, -1.f)));
    const float k = dot(f.n, si.dp_du);
    f.s = normalize(mk(fmaf(-f.n.x, k, si.dp_du.x), fmaf(-f.n.y, k, si.dp_du.y), fmaf(-f.n.z, k, si.dp_du.z)));   // fnmadd(n, dot, dp_du)
    f.t = cross(f.n, f.s);
    return f;
}
// BitmapTexture::eval_1_grad (src/textures/bitmap.cpp:346-421): the gradient of the bilinear interpolant of the (luminance of the) four texels around the
// lookup, through the transpose of the uv transform, times the resolution; the nearest filter has none
DTOF_D void texture_eval_1_grad(const SceneView &sv, uint32_t rec_off, float u, float v, float &gu, float &gv) {
    const DTexture &tex = *(const DTexture *) (sv.base + rec_off);
    gu = gv = 0.f;
    const uint32_t filter = (tex.kind_flags >> 8) & 0xffu, wrap = (tex.kind_flags >> 16) & 0xffu, C = tex.kind_flags >> 24;
    if ((tex.kind_flags & 0xffu) != TEX_BITMAP || filter == 0) return;
    const float tu = fmaf(tex.to_uv[1], v, fmaf(tex.to_uv[0], u, 0.f)), tv = fmaf(tex.to_uv[3], v, fmaf(tex.to_uv[2], u, 0.f));
    const int32_t W = (int32_t) tex.width, H = (int32_t) tex.height;
    const float *data = (const float *) (sv.base + tex.data_off);
    const float px = fmaf(tu, (float) W, -.5f), py = fmaf(tv, (float) H, -.5f), fx = floorf(px), fy = floorf(py);
    const float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    const int32_t x0 = tex_wrap((int32_t) fx, W, wrap), x1 = tex_wrap((int32_t) fx + 1, W, wrap);
    const int32_t y0 = tex_wrap((int32_t) fy, H, wrap), y1 = tex_wrap((int32_t) fy + 1, H, wrap);
    auto fetch = [&](int32_t x, int32_t y) {
        const float *t = data + ((size_t) y * W + x) * C;
        return C == 1 ? t[0] : t[0] * 0.212671f + t[1] * 0.715160f + t[2] * 0.072169f;   // luminance (spectrum.h:431-434)
    };
    const float f00 = fetch(x0, y0), f10 = fetch(x1, y0), f01 = fetch(x0, y1), f11 = fetch(x1, y1);
    const float dfx = fmaf(w0y, f10 - f00, w1y * (f11 - f01)), dfy = fmaf(w0x, f01 - f00, w1x * (f11 - f10));
    gu = (float) W * (tex.to_uv[0] * dfx + tex.to_uv[2] * dfy);
    gv = (float) H * (tex.to_uv[1] * dfx + tex.to_uv[3] * dfy);
}
// BumpMap::frame (src/bsdfs/bumpmap.cpp:199-222): the surface displaced along its shading normal by the height texture, to first order
DTOF_D LocalFrame bumpmap_frame(const SceneView &sv, const DShape *sh, const Surface &si) {
    float gu, gv; texture_eval_1_grad(sv, sh->tex_normal << 4, si.u, si.v, gu, gv);
    gu *= sh->bump_scale; gv *= sh->bump_scale;
    const V3 dp_du = vfma(si.sh_n, gu - dot(si.sh_n, si.dp_du), si.dp_du), dp_dv = vfma(si.sh_n, gv - dot(si.sh_n, si.dp_dv), si.dp_dv);
    V3 n = normalize(cross(dp_du, dp_dv));
    if (dot(si.n, n) < 0.f) n = -n;
    LocalFrame f;
    f.n = mk(dot(n, si.sh_s), dot(n, si.sh_t), dot(n, si.sh_n));                  // si.to_local(n)
    const float k = dot(f.n, si.dp_du);
    f.s = normalize(mk(fmaf(-f.n.x, k, si.dp_du.x), fmaf(-f.n.y, k, si.dp_du.y), fmaf(-f.n.z, k, si.dp_du.z)));
    f.t = cross(f.n, f.s);
    return f;
}
DTOF_D V3 frame_to_local(const LocalFrame &f, V3 v) { return mk(dot(v, f.s), dot(v, f.t), dot(v, f.n)); }
DTOF_D V3 frame_to_world(const LocalFrame &f, V3 v) { return vfma(f.n, v.z, vfma(f.t, v.y, f.s * v.x)); }
// MaskBSDF::eval_opacity (mask.cpp:219-221)
DTOF_D float mask_opacity_at(const SceneView &sv, const DShape *sh, float u, float v) {
    const float o = sh->tex_opacity ? texture_eval_1(sv, sh->tex_opacity << 4, u, v) : sh->opacity;
    return fmin_(fmax_(o, 0.f), 1.f);
}
// has_flag(bsdf->flags(), BSDFFlags::Smooth): the BSDFs with a non-delta lobe
DTOF_D bool bsdf_is_smooth(uint32_t k) { return k == BSDF_DIFFUSE || k == BSDF_PLASTIC || k == BSDF_ROUGHCONDUCTOR || k == BSDF_ROUGHPLASTIC || k == BSDF_ROUGHDIELECTRIC; }
// RoughPlastic::lerp_gather (roughplastic.cpp:373-383) on the 64-entry transmittance table
DTOF_D float lerp_gather64(const float *data, float x) {
    x *= 63.f;
    uint32_t index = (uint32_t) x; if (index > 62u) index = 62u;
    const float v0 = data[index], v1 = data[index + 1], t = x - (float) index;
    return fmaf(v1, t, fmaf(-v0, t, v0));                        // dr::lerp(v0, v1, t)
}
// RoughPlastic::eval (:333-371) and pdf (:385-421) for wi.z > 0 and wo.z > 0
DTOF_D void rough_plastic_eval_pdf(Ggx g, const DShape *sh, const HitMaterial &hm, const float *table, V3 diff, V3 wi, V3 wo, float t_i, float prob_specular,
                                   float prob_diffuse, V3 &value, float &pdf) {
    const V3 H = normalize(wo + wi);
    const float D = ggx_eval(g, H);
    float F, t1, t2, t3; fresnel_dielectric(dot(wi, H), sh->diel_eta, F, t1, t2, t3);
    const float G = ggx_smith_g1(g, wi, H) * ggx_smith_g1(g, wo, H);
    const float spec = F * D * G / (4.f * wi.z);
    const float t_o = lerp_gather64(table, wo.z);
    const float k = kInvPi * sh->inv_eta_2 * wo.z * t_i * t_o;
    value = mk(spec * hm.spec_refl[0] + diff.x * k, spec * hm.spec_refl[1] + diff.y * k, spec * hm.spec_refl[2] + diff.z * k);
    float result = g.visible ? D * ggx_smith_g1(g, wi, H) / (4.f * wi.z) : ggx_pdf(g, wi, H) / (4.f * dot(wo, H));   // roughplastic.cpp:467-470
    result *= prob_specular;
    pdf = result + prob_diffuse * (kInvPi * wo.z);
}
// RoughDielectric::eval_pdf (roughdielectric.cpp:503-611), GGX + visible normals, TransportMode::Radiance
DTOF_D void rough_dielectric_eval_pdf(Ggx g, const DShape *sh, const HitMaterial &hm, V3 wi, V3 wo, V3 &value, float &pdf) {
    const float cti = wi.z, cto = wo.z, m_eta = sh->diel_eta, m_inv_eta = rcp(m_eta);
    const bool reflect = cti * cto > 0.f;
    const float eta = cti > 0.f ? m_eta : m_inv_eta, inv_eta = cti > 0.f ? m_inv_eta : m_eta;
    V3 m = normalize(wi + wo * (reflect ? 1.f : eta));
    m = mk(mulsign(m.x, m.z), mulsign(m.y, m.z), mulsign(m.z, m.z));
    const float dwm = dot(wi, m), dom = dot(wo, m);
    const bool active = cti != 0.f && dwm * cti > 0.f && dom * cto > 0.f;
    const float D = ggx_eval(g, m);
    float F, t1, t2, t3; fresnel_dielectric(dwm, m_eta, F, t1, t2, t3);
    const float G = ggx_smith_g1(g, wi, m) * ggx_smith_g1(g, wo, m);
    value = mk(0, 0, 0); pdf = 0.f;
    if (!active) return;
    if (reflect) {
        const float v = F * D * G / (4.f * fabsf(cti));
        value = mk(v * hm.spec_refl[0], v * hm.spec_refl[1], v * hm.spec_refl[2]);
    } else {
        const float scale = sqr(inv_eta);
        const float v = fabsf((scale * (1.f - F) * D * G * eta * eta * dwm * dom) / (cti * sqr(dwm + eta * dom)));
        value = mk(v * hm.spec_trans[0], v * hm.spec_trans[1], v * hm.spec_trans[2]);
    }
    Ggx gs = g;   // sample_distr: Walter et al.'s roughness scaling when all normals are sampled (roughdielectric.cpp:584-589)
    if (!g.visible) { const float sc = 1.2f - .2f * sqrtf(fabsf(cti)); gs.au *= sc; gs.av *= sc; }
    float p = ggx_pdf(gs, mk(mulsign(wi.x, cti), mulsign(wi.y, cti), mulsign(wi.z, cti)), m);
    p *= reflect ? F : 1.f - F;
    const float dwh_dwo = reflect ? rcp(4.f * dom) : (eta * eta * dom) / sqr(dwm + eta * dom);
    pdf = p * fabsf(dwh_dwo);
}
// fresnel_conductor -- include/mitsuba/render/fresnel.h:93-117 (one colour channel)
DTOF_D float fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) {
    const float cos_theta_i_2 = cos_theta_i * cos_theta_i, sin_theta_i_2 = 1.f - cos_theta_i_2, sin_theta_i_4 = sin_theta_i_2 * sin_theta_i_2;
    const float temp_1 = eta_r * eta_r - eta_i * eta_i - sin_theta_i_2,
                a_2_pb_2 = safe_sqrt(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r),
                a = safe_sqrt(.5f * (a_2_pb_2 + temp_1));
    const float term_1 = a_2_pb_2 + cos_theta_i_2, term_2 = 2.f * cos_theta_i * a;
    const float r_s = (term_1 - term_2) / (term_1 + term_2);
    const float term_3 = a_2_pb_2 * cos_theta_i_2 + sin_theta_i_4, term_4 = term_2 * sin_theta_i_2;
    const float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return 0.5f * (r_s + r_p);
}
DTOF_D float mis_weight(float a, float b) { a *= a; b *= b; float w = a / (a + b); return isfinite(w) ? w : 0.f; }


// ---------------------------------------------------------------------------- environment map (src/emitters/envmap.cpp, rgb)
// Hierarchical2D<Float, 0> (include/mitsuba/core/distr_2d.h): sample :490-575, eval :668-699; bilinear warps warp.h:355-429.
DTOF_D uint32_t env_level_index(uint32_t x, uint32_t y, uint32_t width) { return ((x & 1u) | (((x & ~1u) | (y & 1u)) << 1)) + ((y & ~1u) * width); }
DTOF_D float clamp01(float x) { return fmin_(fmax_(x, 0.f), 1.f); }
DTOF_D float interval_to_linear(float v0, float v1, float sample) {
    const float val = (v0 - safe_sqrt(lerp_(sqr(v0), sqr(v1), sample))) / (v0 - v1);
    return fabsf(v0 - v1) > 1e-4f * (v0 + v1) ? val : sample;
}
DTOF_D void env_warp_sample(const uint8_t *base, const DEnvmap &e, float sx, float sy, float &ux, float &uy, float &pdf) {
    sx = clamp01(sx); sy = clamp01(sy);
    uint32_t ox = 0, oy = 0;
    for (int l = (int) e.n_levels - 2; l > 0; --l) {
        ox <<= 1; oy <<= 1;
        const float4 v = *(const float4 *) ((const float *) (base + e.level_off[l]) + env_level_index(ox, oy, e.level_w[l]));   // one 2 x 2 block = 16 contiguous bytes
        const float v00 = v.x, v10 = v.y, v01 = v.z, v11 = v.w;
        sx = clamp01(sx); sy = clamp01(sy);
        const float r0 = v00 + v10, r1 = v01 + v11;
        sy *= r0 + r1;
        bool mask = sy > r0;
        if (mask) { oy += 1u; sy -= r0; }
        sy /= mask ? r1 : r0;
        const float c0 = mask ? v01 : v00, c1 = mask ? v11 : v10;
        sx *= c0 + c1;
        mask = sx > c0;
        if (mask) sx -= c0;
        sx /= mask ? c1 : c0;
        if (mask) ox += 1u;
    }
    const uint32_t W = e.level_w[0], i = ox + oy * W;
    const float *L = (const float *) (base + e.level_off[0]);
    const float v00 = L[i], v10 = L[i + 1u], v01 = L[i + W], v11 = L[i + W + 1u];
    const float r0 = v00 + v10, r1 = v01 + v11;   // warp::square_to_bilinear
    sy = interval_to_linear(r0, r1, sy);
    const float c0 = lerp_(v00, v01, sy), c1 = lerp_(v10, v11, sy);
    sx = interval_to_linear(c0, c1, sx);
    pdf = lerp_(c0, c1, sx);
    ux = ((float) (int32_t) ox + sx) * e.patch_x; uy = ((float) (int32_t) oy + sy) * e.patch_y;
}
DTOF_D float env_warp_eval(const uint8_t *base, const DEnvmap &e, float x, float y) {
    x = clamp01(x) * e.inv_patch_x; y = clamp01(y) * e.inv_patch_y;
    uint32_t ox = (uint32_t) (int32_t) x, oy = (uint32_t) (int32_t) y;
    if (ox > e.max_px) ox = e.max_px;
    if (oy > e.max_py) oy = e.max_py;
    x -= (float) (int32_t) ox; y -= (float) (int32_t) oy;
    const uint32_t W = e.level_w[0], i = ox + oy * W;
    const float *L = (const float *) (base + e.level_off[0]);
    return lerp_(lerp_(L[i], L[i + 1u], x), lerp_(L[i + W], L[i + W + 1u], x), y);   // square_to_bilinear_pdf
}
// eval_spectrum (envmap.cpp:487-553)
DTOF_D V3 env_eval_uv(const uint8_t *base, const DEnvmap &e, float u, float v) {
    const uint32_t rx = e.w, ry = e.h;
    u -= .5f / (float) (rx - 1u);
    u -= floorf(u); v -= floorf(v);
    u *= (float) (rx - 1u); v *= (float) (ry - 1u);
    uint32_t px = (uint32_t) u, py = (uint32_t) v;
    if (px > rx - 2u) px = rx - 2u;
    if (py > ry - 2u) py = ry - 2u;
    const float w1x = u - (float) px, w1y = v - (float) py, w0x = 1.f - w1x, w0y = 1.f - w1y;
    const float *d = (const float *) (base + e.data_off) + 3u * (py * rx + px);
    float out[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v00 = d[c], v10 = d[3 + c], v01 = d[3u * rx + c], v11 = d[3u * rx + 3 + c];
        const float a = fmaf(w0x, v00, w1x * v10), b = fmaf(w0x, v01, w1x * v11);
        out[c] = fmaf(w0y, a, w1y * b) * e.scale;
    }
    return mk(out[0], out[1], out[2]);
}
constexpr float kEpsilonF = 5.9604644775390625e-8f;   // dr::Epsilon<float> = 2^-24
DTOF_D void env_dir_to_uv(V3 d, float &u, float &v) { u = atan2_(d.x, -d.z) * kInvTwoPi; v = acos_(fmin_(fmax_(d.y, -1.f), 1.f)) * kInvPi; }
DTOF_D float env_inv_sin_theta(V3 d) { return rsqrt_(fmax_(fmax_(sqr(d.x) + sqr(d.z), sqr(kEpsilonF)), 0.f)); }
// EnvironmentMapEmitter::eval (:299-310): d = -si.wi, the direction of the ray that left the scene
DTOF_D V3 env_eval(const uint8_t *base, const DEmitter &em, V3 d) {
    const DEnvmap &e = *(const DEnvmap *) (base + em.shape);
    float u, v; env_dir_to_uv(xf_vector(em.to_local, d), u, v);
    return env_eval_uv(base, e, u, v);
}
// pdf_direction (:408-425)
DTOF_D float env_pdf_direction(const uint8_t *base, const DEmitter &em, V3 dw) {
    const DEnvmap &e = *(const DEnvmap *) (base + em.shape);
    const V3 d = xf_vector(em.to_local, dw);
    float u, v; env_dir_to_uv(d, u, v);
    u -= .5f / (float) (e.w - 1u);
    u -= floorf(u); v -= floorf(v);
    return env_warp_eval(base, e, u, v) * env_inv_sin_theta(d) * (1.f / (2.f * sqr(kPi)));
}
// sample_direction (:363-406)
DTOF_D void env_sample_direction(const uint8_t *base, const DEmitter &em, V3 ref_p, float sx, float sy, V3 &d_out, float &dist, float &pdf_out, V3 &weight, bool &active) {
    const DEnvmap &e = *(const DEnvmap *) (base + em.shape);
    float u, v, pdf; env_warp_sample(base, e, sx, sy, u, v, pdf);
    u += .5f / (float) (e.w - 1u);
    active = pdf > 0.f;
    const float theta = v * kPi, phi = u * (2.f * kPi);
    float st, ct, sp, cp; sincos_(theta, st, ct); sincos_(phi, sp, cp);
    V3 d = mk(cp * st, sp * st, ct);   // dr::sphdir
    d = mk(d.y, d.z, -d.x);
    const float radius = fmax_(em.cutoff_angle, norm(ref_p - mk(em.pos[0], em.pos[1], em.pos[2])));
    dist = 2.f * radius;
    const float ist = env_inv_sin_theta(d);
    d_out = xf_vector(e.to_world, d);
    pdf_out = active ? pdf * ist * (1.f / (2.f * sqr(kPi))) : 0.f;
    const V3 rad = env_eval_uv(base, e, u, v);
    const float ip = rcp(pdf_out);
    weight = active ? mk(rad.x * ip, rad.y * ip, rad.z * ip) : mk(0, 0, 0);
}

}  // namespace dtof
