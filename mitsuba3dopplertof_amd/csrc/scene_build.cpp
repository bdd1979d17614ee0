// scene_build.cpp -- flattens a HostScene into the device scene blob and builds the
// top-level BVH (TLAS) over scene objects.
//
// Instance bounds follow Instance::bbox (src/shapes/instance.cpp:101-114): the union of the
// shapegroup's 8 bbox corners transformed by the first and the last keyframe.  Because the
// per-ray transform is the component-wise lerp of the two keyframe matrices
// (include/mitsuba/core/transform.h:462-466), a point's position at any time is the lerp of
// its two end positions, so that union bounds the whole motion.
#include "dtof_scene.h"
#include "dtof_math.h"
#include "dtof_half.h"
#include <algorithm>
#include <cstring>
#include <cfloat>
#include <cstdlib>
#include <numeric>
#include <limits>

namespace dtof {

struct Box {
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    void add(V3 p) { lo[0] = std::min(lo[0], p.x); lo[1] = std::min(lo[1], p.y); lo[2] = std::min(lo[2], p.z);
                     hi[0] = std::max(hi[0], p.x); hi[1] = std::max(hi[1], p.y); hi[2] = std::max(hi[2], p.z); }
    void add(const Box &b) { for (int i = 0; i < 3; ++i) { lo[i] = std::min(lo[i], b.lo[i]); hi[i] = std::max(hi[i], b.hi[i]); } }
    float area() const { float d[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] }; return 2.f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]); }
    bool valid() const { return lo[0] <= hi[0]; }
    V3 corner(int i) const { return mk(i & 1 ? hi[0] : lo[0], i & 2 ? hi[1] : lo[1], i & 4 ? hi[2] : lo[2]); }
    // conservative padding so that slab-test rounding can never cull a primitive lying on a face
    void pad() {
        for (int i = 0; i < 3; ++i) {
            float e = std::max(std::max(std::fabs(lo[i]), std::fabs(hi[i])), hi[i] - lo[i]) * 1e-5f + 1e-6f;
            lo[i] -= e; hi[i] += e;
        }
    }
};

static Box shape_box(const HostShape &s) {
    Box b;
    if (s.kind == SHAPE_RECT || s.kind == SHAPE_DISK) {   // Rectangle::bbox (rectangle.cpp:115-125), Disk::bbox (disk.cpp:136-146): the same four corners
        const float c[4][2] = { { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
        for (auto &k : c) b.add(xf_point(s.to_world, mk(k[0], k[1], 0.f)));
    } else if (s.kind == SHAPE_CYLINDER) {   // Cylinder::bbox (src/shapes/cylinder.cpp:166-179): the two end circles
        const V3 x1 = xf_vector(s.to_world, mk(1.f, 0.f, 0.f)), x2 = xf_vector(s.to_world, mk(0.f, 1.f, 0.f));
        const V3 x = mk(sqrtf(sqr(x1.x) + sqr(x2.x)), sqrtf(sqr(x1.y) + sqr(x2.y)), sqrtf(sqr(x1.z) + sqr(x2.z)));
        const V3 p0 = xf_point(s.to_world, mk(0.f, 0.f, 0.f)), p1 = xf_point(s.to_world, mk(0.f, 0.f, 1.f));
        b.add(p0 - x); b.add(p1 - x); b.add(p0 + x); b.add(p1 + x);
    } else if (s.kind == SHAPE_SPHERE) {   // Sphere::bbox, src/shapes/sphere.cpp:177-182
        b.add(mk(s.center[0] - s.radius, s.center[1] - s.radius, s.center[2] - s.radius));
        b.add(mk(s.center[0] + s.radius, s.center[1] + s.radius, s.center[2] + s.radius));
    } else {
        for (size_t i = 0; i + 2 < s.positions.size(); i += 3) b.add(mk(s.positions[i], s.positions[i + 1], s.positions[i + 2]));
    }
    return b;
}

struct BuildItem { Box box; float c[3]; uint32_t obj; };

// One builder for both levels.  TLAS (max_leaf = 1): a leaf is kLeafFlag | object index.  BLAS (max_leaf = kBlasLeaf): the
// items (triangles) end up permuted so that every leaf is a contiguous range, a leaf is
// kLeafFlag | (first position in the mesh << kBlasLeafBits) | (count - 1).
struct BuildCtx { std::vector<BvhNode> &nodes; std::vector<BuildItem> &items; uint32_t max_leaf; uint32_t depth = 0, deepest = 0; };

static uint32_t build_node(BuildCtx &cx, size_t b, size_t e);

static uint32_t child_ref(BuildCtx &cx, size_t b, size_t e) {
    if (cx.max_leaf == 1) { if (e - b == 1) return kLeafFlag | cx.items[b].obj; }
    else if (e - b <= cx.max_leaf) return kLeafFlag | ((uint32_t) b << kBlasLeafBits) | (uint32_t) (e - b - 1);
    return build_node(cx, b, e);
}

static inline int bin_of(float c, float lo, float ext, int nb) {   // clamped on both sides, NaN -> 0
    const float v = (c - lo) / ext * (float) nb;
    return v >= 0.f ? (v < (float) (nb - 1) ? (int) v : nb - 1) : 0;
}
// binned SAH split over centroids (16 bins), falling back to a median split
static uint32_t build_node(BuildCtx &cx, size_t b, size_t e) {
    std::vector<BvhNode> &nodes = cx.nodes; std::vector<BuildItem> &items = cx.items;
    cx.deepest = std::max(cx.deepest, ++cx.depth);
    uint32_t idx = (uint32_t) nodes.size();
    nodes.emplace_back();
    Box cb;
    for (size_t i = b; i < e; ++i) cb.add(mk(items[i].c[0], items[i].c[1], items[i].c[2]));
    int best_axis = -1, best_bin = -1; float best_cost = FLT_MAX;
    constexpr int NB = 16;
    for (int ax = 0; ax < 3; ++ax) {
        float lo = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
        if (!(ext > 0.f)) continue;
        Box bb[NB]; int cnt[NB] = { 0 };
        for (size_t i = b; i < e; ++i) {
            int k = bin_of(items[i].c[ax], lo, ext, NB);
            bb[k].add(items[i].box); cnt[k]++;
        }
        Box acc; int n = 0; float la[NB]; int ln[NB];
        for (int k = 0; k < NB; ++k) { acc.add(bb[k]); n += cnt[k]; la[k] = n ? acc.area() : 0.f; ln[k] = n; }
        Box racc; int rn = 0;
        for (int k = NB - 1; k >= 1; --k) {
            racc.add(bb[k]); rn += cnt[k];
            if (ln[k - 1] == 0 || rn == 0) continue;
            float cost = la[k - 1] * ln[k - 1] + racc.area() * rn;
            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = k; }
        }
    }
    size_t mid;
    if (best_axis >= 0) {
        float lo = cb.lo[best_axis], ext = cb.hi[best_axis] - cb.lo[best_axis]; int ax = best_axis, bin = best_bin;
        auto it = std::stable_partition(items.begin() + b, items.begin() + e, [&](const BuildItem &it2) {
            return bin_of(it2.c[ax], lo, ext, NB) < bin; });
        mid = (size_t) (it - items.begin());
    } else {
        mid = (b + e) / 2;
    }
    if (mid == b || mid == e) mid = (b + e) / 2;
    Box lb, rb;
    for (size_t i = b; i < mid; ++i) lb.add(items[i].box);
    for (size_t i = mid; i < e; ++i) rb.add(items[i].box);
    uint32_t l = child_ref(cx, b, mid), r = child_ref(cx, mid, e);
    --cx.depth;
    BvhNode &n = nodes[idx];
    for (int i = 0; i < 3; ++i) { n.lmin[i] = lb.lo[i]; n.lmax[i] = lb.hi[i]; n.rmin[i] = rb.lo[i]; n.rmax[i] = rb.hi[i]; }
    n.left = l; n.right = r; n.pad0 = n.pad1 = 0;
    return idx;
}




static Box tri_box(const DTri &t) {
    Box b; b.add(mk(t.p0[0], t.p0[1], t.p0[2])); b.add(mk(t.p1[0], t.p1[1], t.p1[2])); b.add(mk(t.p2[0], t.p2[1], t.p2[2])); return b;
}

std::vector<uint8_t> build_scene_blob(const HostScene &sc) {
    // ---- shapes / triangles (+ one BLAS per larger mesh; its nodes follow the TLAS in the same array)
    std::vector<DShape> shapes(sc.shapes.size());
    std::vector<DTri> tris; std::vector<DTriShade> shading;
    std::vector<BvhNode> blas_nodes; uint32_t blas_depth = 0;
    std::vector<uint32_t> tables;   // mesh emitters: cdf[n] | pmf[n] (float bits) | slot[n]; roughplastic: 64 transmittances
    const char *blas_env = getenv("DTOF_BLAS"); const bool use_blas = !(blas_env && blas_env[0] == '0');   // DTOF_BLAS=0: loop over every triangle (debug)
    uint32_t blas_leaf = kBlasLeaf;                           // DTOF_BLAS_LEAF=2..8: triangles per BLAS leaf (development: the hits do not depend on it)
    if (const char *e = getenv("DTOF_BLAS_LEAF")) { const long v = strtol(e, nullptr, 10); if (v >= 2 && v <= (long) (1u << kBlasLeafBits)) blas_leaf = (uint32_t) v; }   // (max_leaf == 1 is the builder's TLAS form)
    std::vector<Box> shape_boxes(sc.shapes.size());
    struct TexUse { uint32_t shape, slot, rec; };           // slot: 0 reflectance (rides in `nonlinear`), 1 specular_reflectance, 2 specular_transmittance, 3 alpha_u, 4 alpha_v
    std::vector<TexUse> tex_recs;                          // rec: word offset of the DTexture in `tables`
    std::vector<uint32_t> tex_rec_of(sc.textures.size(), 0xffffffffu);   // texture index -> word offset of its record: every texture is stored once, however many shapes use it
    auto check_words = [&]() { if (tables.size() > 0x3fffffffu) throw std::runtime_error("scene tables exceed the 4 GiB the 32-bit blob offsets address"); };
    // the record of a texture and its texels go to the tables area, once per texture (the offsets are rebased below); returns the word offset of the record
    std::vector<bool> sampled_texture(sc.textures.size(), false);   // textures an area emitter's radiance is sampled through get a DiscreteDistribution2D
    for (const HostShape &hs : sc.shapes) if (hs.tex_radiance >= 0) sampled_texture[(size_t) hs.tex_radiance] = true;
    auto place_texture = [&](int index) -> uint32_t {
        const HostTexture &t = sc.textures[(size_t) index];
        uint32_t &rec = tex_rec_of[(size_t) index];
        if (rec == 0xffffffffu) {
            while (tables.size() % 4) tables.push_back(0);            // 16-byte aligned record (the offset is stored >> 4)
            check_words();
            rec = (uint32_t) tables.size();
            DTexture dt; memset(&dt, 0, sizeof dt);
            dt.kind_flags = t.kind | (t.filter << 8) | (t.wrap << 16) | (t.channels << 24);
            dt.width = t.width; dt.height = t.height; dt.data_off = (rec + (uint32_t) (sizeof(DTexture) / 4)) * 4u;
            memcpy(dt.to_uv, t.to_uv, 16); memcpy(dt.color0, t.color0, 12); memcpy(dt.color1, t.color1, 12);
            const uint32_t *w = (const uint32_t *) &dt;
            tables.insert(tables.end(), w, w + sizeof(DTexture) / 4);
            const size_t at = tables.size();
            tables.resize(at + t.data.size());
            if (!t.data.empty()) memcpy(&tables[at], t.data.data(), t.data.size() * 4);
            if (sampled_texture[(size_t) index] && t.kind == TEX_BITMAP) {
                // DiscreteDistribution2D(data, size) (distr_2d.h:92-117) over BitmapTexture::rebuild_internals' importance map (bitmap.cpp:689-724: the luminance of RGB texels,
                // the value of gray ones): running sums of each row and of the row totals, accumulated in double, stored as float32
                const uint32_t W = t.width, H = t.height, C = t.channels;
                const size_t at_d = tables.size();
                tables.resize(at_d + 2 + H + (size_t) W * H);
                double accum_marg = 0.0;
                for (uint32_t y = 0; y < H; ++y) {
                    double accum_cond = 0.0;
                    for (uint32_t x = 0; x < W; ++x) {
                        const float *px = &t.data[((size_t) y * W + x) * C];
                        const float imp = C == 1 ? px[0] : px[0] * 0.212671f + px[1] * 0.715160f + px[2] * 0.072169f;
                        accum_cond += (double) imp;
                        const float f = (float) accum_cond; memcpy(&tables[at_d + 2 + H + (size_t) y * W + x], &f, 4);
                    }
                    accum_marg += accum_cond;
                    const float f = (float) accum_marg; memcpy(&tables[at_d + 2 + y], &f, 4);
                }
                const float inv_norm = (float) accum_marg, norm = (float) (1.0 / accum_marg);
                memcpy(&tables[at_d], &norm, 4); memcpy(&tables[at_d + 1], &inv_norm, 4);
                tables[rec + 14] = (uint32_t) at_d * 4u;   // DTexture::distr_off, rebased with the record
            }
            check_words();
        }
        return rec;
    };
    // the material half of a shape record: flags, BSDF parameters, textures of its slots -- also run for the material-only records of blendbsdf partners
    auto fill_material = [&](const HostShape &h, size_t i) {
        DShape &d = shapes[i];
        d.flags = (h.twosided ? SF_TWOSIDED : 0) | (h.flip_normals ? SF_FLIP_NORMALS : 0) | (h.face_normals ? SF_FACE_NORMALS : 0) | (h.beckmann ? SF_BECKMANN : 0) | (h.sample_all ? SF_SAMPLE_ALL : 0) | (!h.texcoords.empty() ? SF_TEXCOORDS : 0) | (h.masked ? SF_MASK : 0) | (h.tex_normal >= 0 ? (h.bumpmap ? SF_BUMPMAP : SF_NORMALMAP) : 0);
        d.bump_scale = h.bump_scale;
        d.opacity = h.opacity;
        memcpy(d.refl, h.refl, 12); d.blas_root = kNoChild;
        d.bsdf = h.bsdf; d.diel_eta = h.diel_eta; d.nonlinear = h.nonlinear; d.inv_eta_2 = h.inv_eta_2; d.fdr_int = h.fdr_int; d.spec_sampling_weight = h.spec_sampling_weight; d.alpha_u = h.alpha_u; d.alpha_v = h.alpha_v;
        memcpy(d.cond_eta, h.cond_eta, 12); memcpy(d.cond_k, h.cond_k, 12); memcpy(d.spec_refl, h.spec_refl, 12); memcpy(d.spec_trans, h.spec_trans, 12);
        if (h.bsdf == BSDF_ROUGHPLASTIC) {   // m_external_transmittance; rebased to a blob offset below
            d.rough_table = (uint32_t) tables.size() * 4u;
            for (float v : h.rough_table) { uint32_t b; memcpy(&b, &v, 4); tables.push_back(b); }
        }
        const int tex_of_slot[7] = { h.tex_refl, h.tex_spec, h.tex_trans, h.tex_alpha_u, h.tex_alpha_v, h.tex_opacity, h.tex_normal };
        for (uint32_t slot = 0; slot < 7; ++slot) if (tex_of_slot[slot] >= 0) tex_recs.push_back({ (uint32_t) i, slot, place_texture(tex_of_slot[slot]) });
    };
    for (size_t i = 0; i < sc.shapes.size(); ++i) {
        memset(&shapes[i], 0, sizeof(DShape));
        shapes[i].kind = sc.shapes[i].kind;
        fill_material(sc.shapes[i], i);
        const HostShape &h = sc.shapes[i]; DShape &d = shapes[i];
        if (h.emitter) { d.flags |= SF_EMITTER; memcpy(d.radiance, h.radiance, 12); }
        memcpy(d.to_world, h.to_world, 48); memcpy(d.to_object, h.to_object, 48);
        if (h.kind == SHAPE_RECT) {   // Rectangle::update, rectangle.cpp:101-113
            V3 du = xf_vector(h.to_world, mk(2.f, 0.f, 0.f)), dv = xf_vector(h.to_world, mk(0.f, 2.f, 0.f));
            V3 n = normalize(xf_normal(h.to_object, mk(0.f, 0.f, 1.f)));
            d.n[0] = n.x; d.n[1] = n.y; d.n[2] = n.z;
            d.dp_du[0] = du.x; d.dp_du[1] = du.y; d.dp_du[2] = du.z;
            d.dp_dv[0] = dv.x; d.dp_dv[1] = dv.y; d.dp_dv[2] = dv.z;
            d.inv_area = rcp(norm(cross(du, dv)));   // Rectangle::surface_area / m_inv_surface_area (rectangle.cpp:109,127-129)
        } else if (h.kind == SHAPE_DISK) {   // Disk::update + surface_area (src/shapes/disk.cpp:100-115,148-152)
            const V3 du = xf_vector(h.to_world, mk(1.f, 0.f, 0.f)), dv = xf_vector(h.to_world, mk(0.f, 1.f, 0.f));
            const float m_du = norm(du), m_dv = norm(dv);
            const V3 n = normalize(xf_normal(h.to_object, mk(0.f, 0.f, 1.f))), fs = du * rcp(m_du), ft = dv * rcp(m_dv);
            d.n[0] = n.x; d.n[1] = n.y; d.n[2] = n.z;
            const float hh = sqrtf(sqr(m_dv) - sqr(dot(ft * m_dv, fs)));
            d.inv_area = rcp(kPi * m_du * hh);
        } else if (h.kind == SHAPE_SPHERE) {
            memcpy(d.n, h.center, 12); d.dp_du[0] = h.radius; d.inv_area = h.sphere_inv_area;
        } else if (h.kind == SHAPE_CYLINDER) {
            d.dp_du[0] = h.radius;      // everything else is in the composed to_world / to_object
        } else {
            d.first_tri = (uint32_t) tris.size(); d.n_tris = (uint32_t) (h.faces.size() / 3);
            for (uint32_t f = 0; f < d.n_tris; ++f) {
                DTri t; DTriShade s; memset(&t, 0, sizeof t); memset(&s, 0, sizeof s);
                const uint32_t *fi = &h.faces[3 * f];
                float *tp[3] = { t.p0, t.p1, t.p2 }; float *sn[3] = { s.n0, s.n1, s.n2 }; float *su[3] = { s.uv0, s.uv1, s.uv2 };
                t.face = f;
                for (int k = 0; k < 3; ++k) {
                    memcpy(tp[k], &h.positions[3 * fi[k]], 12);
                    if (!h.normals.empty()) memcpy(sn[k], &h.normals[3 * fi[k]], 12);
                    if (!h.texcoords.empty()) memcpy(su[k], &h.texcoords[2 * fi[k]], 8);
                }
                tris.push_back(t); shading.push_back(s);
            }
            std::vector<uint32_t> slot_of_face(d.n_tris);
            for (uint32_t f = 0; f < d.n_tris; ++f) slot_of_face[f] = f;
            if (use_blas && d.n_tris > kBlasMinTris) {
                std::vector<BuildItem> items(d.n_tris);
                for (uint32_t f = 0; f < d.n_tris; ++f) {
                    BuildItem &it = items[f]; it.box = tri_box(tris[d.first_tri + f]); it.obj = f;
                    for (int k = 0; k < 3; ++k) it.c[k] = 0.5f * (it.box.lo[k] + it.box.hi[k]);
                    it.box.pad();
                }
                if (d.n_tris >= (1u << (31 - kBlasLeafBits))) throw std::runtime_error("a mesh has more triangles than a BLAS leaf reference addresses");
                BuildCtx cx { blas_nodes, items, blas_leaf };
                d.blas_root = build_node(cx, 0, items.size());   // index within blas_nodes; rebased behind the TLAS below
                blas_depth = std::max(blas_depth, cx.deepest);
                std::vector<DTri> t2(d.n_tris); std::vector<DTriShade> s2(d.n_tris);
                for (uint32_t f = 0; f < d.n_tris; ++f) { t2[f] = tris[d.first_tri + items[f].obj]; s2[f] = shading[d.first_tri + items[f].obj]; }
                std::copy(t2.begin(), t2.end(), tris.begin() + d.first_tri); std::copy(s2.begin(), s2.end(), shading.begin() + d.first_tri);
                for (uint32_t f = 0; f < d.n_tris; ++f) slot_of_face[items[f].obj] = f;
            }
            if (h.emitter) {   // Mesh::build_pmf (mesh.cpp:478-511) + DiscreteDistribution::compute_cdf (distr_1d.h:205-240)
                if (d.n_tris == 0) throw std::runtime_error("Cannot create sampling table for an empty mesh");
                std::vector<float> pmf(d.n_tris), cdf(d.n_tris);
                double sum = 0.0; int64_t lo = -1, hi = -1;
                for (uint32_t f = 0; f < d.n_tris; ++f) {
                    const uint32_t *fi = &h.faces[3 * f]; const float *P = h.positions.data();
                    V3 p0 = mk(P[3 * fi[0]], P[3 * fi[0] + 1], P[3 * fi[0] + 2]), p1 = mk(P[3 * fi[1]], P[3 * fi[1] + 1], P[3 * fi[1] + 2]),
                       p2 = mk(P[3 * fi[2]], P[3 * fi[2] + 1], P[3 * fi[2] + 2]);
                    pmf[f] = .5f * norm(cross(p1 - p0, p2 - p0));
                    sum += (double) pmf[f]; cdf[f] = (float) sum;
                    if (pmf[f] > 0.f) { if (lo < 0) lo = f; hi = f; }
                }
                if (lo < 0) throw std::runtime_error("DiscreteDistribution: no probability mass found!");
                d.emit_table = (uint32_t) tables.size() * 4u;   // rebased to a blob offset below
                d.emit_lo = (uint32_t) lo; d.emit_hi = (uint32_t) hi; d.emit_sum = (float) sum; d.inv_area = (float) (1.0 / sum);
                for (float v : cdf) { uint32_t b; memcpy(&b, &v, 4); tables.push_back(b); }
                for (float v : pmf) { uint32_t b; memcpy(&b, &v, 4); tables.push_back(b); }
                tables.insert(tables.end(), slot_of_face.begin(), slot_of_face.end());
            }
        }
        for (float v : h.positions) if (!std::isfinite(v)) throw std::runtime_error("shape \"" + h.id + "\": non-finite vertex position (check its to_world transform)");
        for (int k = 0; k < 12; ++k) if (!std::isfinite(h.to_world[k]) || !std::isfinite(h.to_object[k]))
            throw std::runtime_error("shape \"" + h.id + "\": non-finite or singular to_world transform");
        if (h.kind == SHAPE_SPHERE && !(std::isfinite(h.radius) && std::isfinite(h.center[0]) && std::isfinite(h.center[1]) && std::isfinite(h.center[2])))
            throw std::runtime_error("shape \"" + h.id + "\": non-finite sphere centre or radius");
        shape_boxes[i] = shape_box(h);
        { Box pb = shape_boxes[i]; pb.pad(); for (int k = 0; k < 3; ++k) { d.bmin[k] = pb.lo[k]; d.bmax[k] = pb.hi[k]; } }
    }
    // ---- groups
    std::vector<DGroup> groups(sc.groups.size());
    std::vector<Box> group_boxes(sc.groups.size());
    for (size_t g = 0; g < sc.groups.size(); ++g) {
        groups[g].first_shape = sc.groups[g].first_shape; groups[g].n_shapes = sc.groups[g].n_shapes; groups[g].pad[0] = groups[g].pad[1] = 0;
        for (uint32_t k = 0; k < sc.groups[g].n_shapes; ++k) group_boxes[g].add(shape_boxes[sc.groups[g].first_shape + k]);
    }
    // ---- objects + TLAS items
    std::vector<DObject> objects(sc.objects.size());
    std::vector<BuildItem> items;
    bool has_instances = false;
    for (size_t i = 0; i < sc.objects.size(); ++i) {
        const HostObject &h = sc.objects[i]; DObject &d = objects[i];
        memset(&d, 0, sizeof d);
        d.kind = h.kind; d.index = h.index; d.n_keys = h.n_keys; d.t0 = h.key_time[0]; d.t1 = h.key_time[1];
        memcpy(d.key0, h.key[0], 48); memcpy(d.key1, h.key[1], 48);
        for (uint32_t kk = 0; kk < std::min(h.n_keys, 2u); ++kk) for (int c = 0; c < 12; ++c) if (!std::isfinite(h.key[kk][c]))
            throw std::runtime_error("object " + std::to_string(i) + ": non-finite instance / animation matrix");
        Box b;
        if (h.kind == OBJ_SHAPE) b = shape_boxes[h.index];
        else {
            has_instances = true;
            const Box &gb = group_boxes[h.index];
            if (gb.valid()) for (int c = 0; c < 8; ++c) {
                b.add(xf_point(h.key[0], gb.corner(c)));
                if (h.n_keys > 1) b.add(xf_point(h.key[1], gb.corner(c)));
            }
        }
        if (!b.valid()) continue;   // empty shapegroup: never hit
        b.pad();
        BuildItem it; it.box = b; it.obj = (uint32_t) i;
        for (int k = 0; k < 3; ++k) it.c[k] = 0.5f * (b.lo[k] + b.hi[k]);
        items.push_back(it);
    }
    // ConstantBackgroundEmitter::set_scene (constant.cpp:73-83): the bounding sphere of Scene::bbox() (the shapes' bboxes; instances over their
    // first and last keyframe), radius = max(RayEpsilon, r * (1 + RayEpsilon)); an empty scene: centre 0, radius 1
    float env_sphere[4] = { 0.f, 0.f, 0.f, 1.f };
    {
        Box all;
        for (size_t i = 0; i < sc.objects.size(); ++i) {
            const HostObject &ho = sc.objects[i]; Box b;
            if (ho.kind == OBJ_SHAPE) b = shape_boxes[ho.index];
            else {
                const Box &gb = group_boxes[ho.index];
                if (gb.valid()) for (int c = 0; c < 8; ++c) { b.add(xf_point(ho.key[0], gb.corner(c))); if (ho.n_keys > 1) b.add(xf_point(ho.key[1], gb.corner(c))); }
            }
            if (b.valid()) all.add(b);
        }
        if (all.valid()) {
            const V3 c = mk((all.lo[0] + all.hi[0]) * .5f, (all.lo[1] + all.hi[1]) * .5f, (all.lo[2] + all.hi[2]) * .5f);
            const float r = norm(c - mk(all.hi[0], all.hi[1], all.hi[2]));
            env_sphere[0] = c.x; env_sphere[1] = c.y; env_sphere[2] = c.z; env_sphere[3] = fmax_(kRayEps, r * (1.f + kRayEps));
        }
    }
    std::vector<BvhNode> nodes;
    if (items.size() == 1) {
        BvhNode n; memset(&n, 0, sizeof n);
        for (int i = 0; i < 3; ++i) { n.lmin[i] = items[0].box.lo[i]; n.lmax[i] = items[0].box.hi[i]; n.rmin[i] = FLT_MAX; n.rmax[i] = -FLT_MAX; }
        n.left = kLeafFlag | items[0].obj; n.right = kNoChild;
        nodes.push_back(n);
    } else if (items.size() > 1) {
        BuildCtx cx { nodes, items, 1 };
        build_node(cx, 0, items.size());
    }
    const uint32_t tlas_nodes = (uint32_t) nodes.size();
    {   // TLAS leaves of objects that hold a mesh behind a BLAS carry kLeafBlas
        if (objects.size() > kLeafObjMask) throw std::runtime_error("more top-level objects than a TLAS leaf reference addresses");
        auto object_has_blas = [&](uint32_t oi) {
            const DObject &ob = objects[oi];
            uint32_t first = ob.index, count = 1;
            if (ob.kind == OBJ_INSTANCE) { first = groups[ob.index].first_shape; count = groups[ob.index].n_shapes; }
            for (uint32_t k = 0; k < count; ++k) if (shapes[first + k].kind == SHAPE_MESH && shapes[first + k].blas_root != kNoChild) return true;
            return false;
        };
        for (BvhNode &n : nodes)
            for (uint32_t *c : { &n.left, &n.right })
                if (*c != kNoChild && (*c & kLeafFlag) && object_has_blas(*c & kLeafObjMask)) *c |= kLeafBlas;
    }
    for (BvhNode n : blas_nodes) {   // BLAS node indices (children and roots) move behind the TLAS
        if (n.left != kNoChild && !(n.left & kLeafFlag)) n.left += tlas_nodes;
        if (n.right != kNoChild && !(n.right & kLeafFlag)) n.right += tlas_nodes;
        nodes.push_back(n);
    }
    for (DShape &d : shapes) if (d.blas_root != kNoChild) d.blas_root += tlas_nodes;
    std::vector<BvhNode> &dev_nodes = nodes;
    uint32_t need_tlas = 0, need_blas = blas_depth ? blas_depth + 1 : 0;
    {   // deepest leaf below the root = stack entries a depth-first traversal can hold
        uint32_t deepest = 1;
        std::vector<std::pair<uint32_t, uint32_t>> todo; if (tlas_nodes) todo.emplace_back(0u, 1u);
        while (!todo.empty()) {
            auto [ni, d] = todo.back(); todo.pop_back();
            deepest = std::max(deepest, d);
            for (uint32_t c : { nodes[ni].left, nodes[ni].right })
                if (c != kNoChild && !(c & kLeafFlag)) todo.emplace_back(c, d + 1);
        }
        need_tlas = deepest + 1;
    }
    // ---- emitters
    std::vector<uint32_t> env_records;   // emitters whose `shape` is the table offset of a DEnvmap
    std::vector<DEmitter> emitters(sc.emitters.size());
    for (size_t i = 0; i < sc.emitters.size(); ++i) {
        emitters[i].kind = sc.emitters[i].kind; emitters[i].shape = sc.emitters[i].shape;
        memcpy(emitters[i].pos, sc.emitters[i].pos, 12); memcpy(emitters[i].intensity, sc.emitters[i].intensity, 12);
        memcpy(emitters[i].to_local, sc.emitters[i].to_local, 48);
        emitters[i].cutoff_angle = sc.emitters[i].cutoff_angle; emitters[i].cos_cutoff = sc.emitters[i].cos_cutoff;
        emitters[i].cos_beam = sc.emitters[i].cos_beam; emitters[i].inv_transition = sc.emitters[i].inv_transition;
        if (sc.emitters[i].kind == EMITTER_CONSTANT || sc.emitters[i].kind == EMITTER_ENVMAP || sc.emitters[i].kind == EMITTER_DIRECTIONAL) { memcpy(emitters[i].pos, env_sphere, 12); emitters[i].cutoff_angle = env_sphere[3]; }
        if (sc.emitters[i].kind == EMITTER_ENVMAP) {
            // EnvironmentMapEmitter's constructor (envmap.cpp:130-224): a periodic extra column, luminance x sin(theta) as the sampling density, and the
            // Hierarchical2D<Float, 0> built over it (distr_2d.h:376-482): level 0 = the normalised grid, level 1 = patch averages, then 2 x 2 sums
            const HostEmitter &he = sc.emitters[i];
            const uint32_t bw = he.image_w, W = bw + 1, H = he.image_h;
            std::vector<float> data((size_t) W * H * 3), lum((size_t) W * H);
            float luminance_offset = 0.f;   // mis_compensation (envmap.cpp:157-185): the mean luminance, unless the map is (nearly) constant
            if (he.mis_compensation) {
                float min_lum = 0.f; double accum = 0.0;
                for (size_t i = 0; i < (size_t) bw * H; ++i) {
                    const float *in = &he.image[i * 3];
                    const float l = in[0] * 0.212671f + in[1] * 0.715160f + in[2] * 0.072169f;
                    min_lum = fmin_(min_lum, l); accum += (double) l;
                }
                luminance_offset = (float) (accum / (double) ((size_t) bw * H));
                if (luminance_offset - min_lum <= 0.01f * luminance_offset) luminance_offset = 0.f;
            }
            const float theta_scale = 1.f / (float) (H - 1) * kPi;
            for (uint32_t y = 0; y < H; ++y) {
                const float sin_theta = sinf((float) y * theta_scale);   // ScalarFloat dr::sin
                for (uint32_t x = 0; x < bw; ++x) {
                    const float *in = &he.image[((size_t) y * bw + x) * 3];
                    const float l = fmax_(in[0] * 0.212671f + in[1] * 0.715160f + in[2] * 0.072169f - luminance_offset, 0.f);   // mitsuba::luminance (spectrum.h:431-434)
                    lum[(size_t) y * W + x] = l * sin_theta;
                    memcpy(&data[((size_t) y * W + x) * 3], in, 12);
                }
                lum[(size_t) y * W + bw] = lum[(size_t) y * W];
                memcpy(&data[((size_t) y * W + bw) * 3], &data[(size_t) y * W * 3], 12);
            }
            const uint32_t npx = W - 1, npy = H - 1;
            uint32_t max_level = 0; { const uint32_t v = std::max(npx, npy); while ((1u << max_level) < v) ++max_level; }   // math::log2i_ceil
            if (max_level + 2 > kEnvMaxLevels) throw std::runtime_error("envmap: the image is too large");
            auto index_of = [](uint32_t x, uint32_t y, uint32_t width) { return ((x & 1u) | (((x & ~1u) | (y & 1u)) << 1)) + ((y & ~1u) * width); };
            std::vector<std::vector<float>> levels(max_level + 2); std::vector<uint32_t> level_w(max_level + 2);
            levels[0].assign((size_t) W * H, 0.f); level_w[0] = W;
            { uint32_t lx = npx, ly = npy; for (uint32_t k = 1; k <= max_level + 1; ++k) { lx += lx & 1u; ly += ly & 1u; levels[k].assign((size_t) lx * ly, 0.f); level_w[k] = lx; lx >>= 1; ly >>= 1; } }
            double sum = 0.0;
            for (uint32_t y = 0; y < npy; ++y) for (uint32_t x = 0; x < npx; ++x) {
                const float *in = &lum[(size_t) y * W + x];
                const float avg = .25f * (in[0] + in[1] + in[W] + in[W + 1]);
                sum += (double) avg;
                levels[1][index_of(x, y, level_w[1])] = avg;
            }
            const float norm = (float) ((double) (npx * npy) / sum);
            for (size_t k = 0; k < lum.size(); ++k) levels[0][k] = lum[k] * norm;
            for (float &v : levels[1]) v *= norm;
            { uint32_t lx = npx, ly = npy;
              for (uint32_t level = 2; level <= max_level + 1; ++level) {
                  lx = (lx + 1) >> 1; ly = (ly + 1) >> 1;
                  for (uint32_t y = 0; y < ly; ++y) for (uint32_t x = 0; x < lx; ++x) {
                      const float *d0 = &levels[level - 1][index_of(x * 2, y * 2, level_w[level - 1])];
                      levels[level][index_of(x, y, level_w[level])] = d0[0] + d0[1] + d0[2] + d0[3];
                  }
              } }
            DEnvmap rec; memset(&rec, 0, sizeof rec);
            rec.w = W; rec.h = H; rec.n_levels = max_level + 2; rec.scale = he.scale;
            rec.patch_x = 1.f / (float) npx; rec.patch_y = 1.f / (float) npy; rec.inv_patch_x = (float) npx; rec.inv_patch_y = (float) npy;
            rec.max_px = npx - 1; rec.max_py = npy - 1;
            memcpy(rec.to_world, he.to_world, 48);
            while (tables.size() % 4) tables.push_back(0);
            const uint32_t rec_word = (uint32_t) tables.size();
            tables.resize(tables.size() + sizeof(DEnvmap) / 4);
            auto append = [&](const std::vector<float> &v) { while (tables.size() % 4) tables.push_back(0); const uint32_t at = (uint32_t) tables.size() * 4u; for (float f : v) { uint32_t b; memcpy(&b, &f, 4); tables.push_back(b); } return at; };
            rec.data_off = append(data);
            for (uint32_t k = 0; k < rec.n_levels; ++k) { rec.level_off[k] = append(levels[k]); rec.level_w[k] = level_w[k]; }
            memcpy(&tables[rec_word], &rec, sizeof rec);
            emitters[i].shape = rec_word * 4u;   // rebased to a blob offset (and the offsets inside the record with it) once off_tables is known
            env_records.push_back((uint32_t) i);
        }
    }
    // ---- pack: nodes first (so that "the first N bytes" = header + top of the TLAS in BFS-ish order)
    BlobHeader h; memset(&h, 0, sizeof h);
    h.n_nodes = (uint32_t) dev_nodes.size(); h.n_tlas_nodes = tlas_nodes; h.n_objects = (uint32_t) objects.size(); h.n_groups = (uint32_t) groups.size();
    {   // blendbsdf: one material-only record per blended shape, behind the real shapes (groups and objects index the real ones only)
        const size_t n_real = sc.shapes.size();
        for (size_t i = 0; i < n_real; ++i) if (sc.shapes[i].blend_other) {
            const HostShape &h = sc.shapes[i];
            shapes.emplace_back();
            const size_t k = shapes.size() - 1;
            memset(&shapes[k], 0, sizeof(DShape));
            fill_material(*h.blend_other, k);
            shapes[i].flags |= h.two_bsdfs ? SF_TWOSIDED2 : SF_BLEND; shapes[i].blend_other = (uint32_t) k; shapes[i].blend_weight = h.blend_weight;
            if (h.tex_blend >= 0) tex_recs.push_back({ (uint32_t) i, 7u, place_texture(h.tex_blend) });
        }
        for (size_t i = 0; i < n_real; ++i) if (sc.shapes[i].tex_radiance >= 0) tex_recs.push_back({ (uint32_t) i, 8u, place_texture(sc.shapes[i].tex_radiance) });   // textured area emitters
    }
    h.n_shapes = (uint32_t) shapes.size(); h.n_tris = (uint32_t) tris.size(); h.n_emitters = (uint32_t) emitters.size();
    (void) has_instances;
    h.tlas_depth = need_tlas + need_blas;
    // offsets are accumulated in 64 bits and the total is checked: every offset in the blob is a uint32_t
    uint64_t off = sizeof(BlobHeader);
    auto place = [&](uint64_t bytes) { const uint64_t at = off; off = (off + bytes + 15u) & ~(uint64_t) 15u;
                                       if (off > 0xffffffffull) throw std::runtime_error("scene blob exceeds the 4 GiB its 32-bit offsets address"); return (uint32_t) at; };
    h.off_nodes = place(dev_nodes.size() * sizeof(DNode));
    h.off_objects = place(objects.size() * sizeof(DObject));
    h.off_groups = place(groups.size() * sizeof(DGroup));
    h.off_shapes = place(shapes.size() * sizeof(DShape));
    h.off_emitters = place(emitters.size() * sizeof(DEmitter));
    h.off_tris = place(tris.size() * sizeof(DTri));
    h.off_shading = place(shading.size() * sizeof(DTriShade));
    std::vector<DTriIsect> isect(tris.size());   // in the final (BLAS) order of the triangles
    for (size_t i = 0; i < tris.size(); ++i) {
        const DTri &t = tris[i];
        const V3 p0 = mk(t.p0[0], t.p0[1], t.p0[2]), p1 = mk(t.p1[0], t.p1[1], t.p1[2]), p2 = mk(t.p2[0], t.p2[1], t.p2[2]);
        const V3 e1 = p0 - p1, e2 = p2 - p0, ng = cross(e2, e1);
        isect[i] = DTriIsect{ { p0.x, p0.y, p0.z }, ng.x, { e1.x, e1.y, e1.z }, ng.y, { e2.x, e2.y, e2.z }, ng.z };
    }
    h.off_isect = place(isect.size() * sizeof(DTriIsect));
    h.off_tables = place((uint64_t) tables.size() * 4);
    std::vector<DNode16> nodes16;   // scenes with a BLAS: the half-float copy of the node array (dtof_scene.h)
    if (!blas_nodes.empty() || (dev_nodes.size() > 1024 && dev_nodes.size() <= 2048)) {   // ... and TLAS-only scenes of 1 025 .. 2 048 nodes: the resident stage of the first-bounce kernel holds them as half-float planes (k_shade: RH16)
        bool fits = true;
        for (const BvhNode &n : dev_nodes) for (int i = 0; i < 3; ++i) {
            fits &= std::fabs(n.lmin[i]) <= 65000.f && std::fabs(n.lmax[i]) <= 65000.f;
            if (n.right != kNoChild) fits &= std::fabs(n.rmin[i]) <= 65000.f && std::fabs(n.rmax[i]) <= 65000.f;
        }
        if (fits) {
            nodes16.resize(dev_nodes.size());
            for (size_t i = 0; i < dev_nodes.size(); ++i) {
                const BvhNode &n = dev_nodes[i]; DNode16 &o = nodes16[i];
                const bool one = n.right == kNoChild;   // (a node with one child: its right box is never tested)
                for (int k = 0; k < 3; ++k) {
                    o.lbox[k] = half_toward(n.lmin[k], false); o.lbox[3 + k] = half_toward(n.lmax[k], true);
                    o.rbox[k] = one ? 0 : half_toward(n.rmin[k], false); o.rbox[3 + k] = one ? 0 : half_toward(n.rmax[k], true);
                }
                o.left = n.left; o.right = n.right;
            }
        }
    }
    { const uint32_t at = place(nodes16.size() * sizeof(DNode16)); h.off_nodes16 = nodes16.empty() ? 0u : at; }
    std::vector<DFlatObject> flat;   // small rectangle-only scenes: one 64-byte record per object for trace_flat
    {
        bool ok = !objects.empty() && objects.size() <= kFlatObjects && tris.empty();
        for (const DShape &d : shapes) ok &= d.kind == SHAPE_RECT;
        if (ok) for (const DObject &ob : objects) {
            DFlatObject f; memset(&f, 0, sizeof f);
            f.instance = ob.kind == OBJ_INSTANCE;
            const float *m = nullptr;   // row-major 3 x 4
            if (!f.instance) m = shapes[ob.index].to_object;
            else if (groups[ob.index].n_shapes == 1) {   // an instance of one rectangle: its object-space matrix rides along (mark 2), the ray is moved there with the memoised inverse
                f.instance = 2; m = shapes[groups[ob.index].first_shape].to_object;
            }
            if (m) for (int r = 0; r < 3; ++r) { f.c0[r] = m[4 * r]; f.c1[r] = m[4 * r + 1]; f.c2[r] = m[4 * r + 2]; f.c3[r] = m[4 * r + 3]; }
            flat.push_back(f);
        }
    }
    { const uint32_t at = place(flat.size() * sizeof(DFlatObject)); h.off_flat = flat.empty() ? 0u : at; }
    h.total_bytes = (uint32_t) off;
    for (DShape &d : shapes) if (d.kind == SHAPE_MESH && (d.flags & SF_EMITTER)) d.emit_table += h.off_tables;
    for (DShape &d : shapes) if (d.bsdf == BSDF_ROUGHPLASTIC) d.rough_table += h.off_tables;
    for (uint32_t ei : env_records) {
        DEnvmap *rec = (DEnvmap *) &tables[emitters[ei].shape / 4u];
        rec->data_off += h.off_tables;
        for (uint32_t k = 0; k < rec->n_levels; ++k) rec->level_off[k] += h.off_tables;
        emitters[ei].shape += h.off_tables;
    }
    std::vector<bool> rebased(tables.size() / 4 + 1, false);
    for (auto &tr : tex_recs) {   // record offsets (>> 4): the reflectance texture beside the `nonlinear` bit, the others in their own fields; texel offset inside the record
        const uint32_t rec_off = h.off_tables + tr.rec * 4u;
        DShape &d = shapes[tr.shape];
        if (tr.slot == 0) d.nonlinear |= (rec_off >> 4) << 1;
        else (tr.slot == 1 ? d.tex_spec : tr.slot == 2 ? d.tex_trans : tr.slot == 3 ? d.tex_alpha_u : tr.slot == 4 ? d.tex_alpha_v : tr.slot == 5 ? d.tex_opacity : tr.slot == 6 ? d.tex_normal : tr.slot == 7 ? d.tex_blend : d.tex_radiance) = rec_off >> 4;
        if (!rebased[tr.rec / 4]) { tables[tr.rec + 3] += h.off_tables; if (tables[tr.rec + 14]) tables[tr.rec + 14] += h.off_tables; rebased[tr.rec / 4] = true; }   // DTexture::data_off / distr_off, once per record
    }
    std::vector<uint8_t> blob(off, 0);
    memcpy(blob.data(), &h, sizeof h);
    if (!dev_nodes.empty()) memcpy(blob.data() + h.off_nodes, dev_nodes.data(), dev_nodes.size() * sizeof(DNode));
    if (!objects.empty()) memcpy(blob.data() + h.off_objects, objects.data(), objects.size() * sizeof(DObject));
    if (!groups.empty()) memcpy(blob.data() + h.off_groups, groups.data(), groups.size() * sizeof(DGroup));
    if (!shapes.empty()) memcpy(blob.data() + h.off_shapes, shapes.data(), shapes.size() * sizeof(DShape));
    if (!emitters.empty()) memcpy(blob.data() + h.off_emitters, emitters.data(), emitters.size() * sizeof(DEmitter));
    if (!tris.empty()) memcpy(blob.data() + h.off_tris, tris.data(), tris.size() * sizeof(DTri));
    if (!shading.empty()) memcpy(blob.data() + h.off_shading, shading.data(), shading.size() * sizeof(DTriShade));
    if (!isect.empty()) memcpy(blob.data() + h.off_isect, isect.data(), isect.size() * sizeof(DTriIsect));
    if (!tables.empty()) memcpy(blob.data() + h.off_tables, tables.data(), tables.size() * 4);
    if (!flat.empty()) memcpy(blob.data() + h.off_flat, flat.data(), flat.size() * sizeof(DFlatObject));
    if (!nodes16.empty()) memcpy(blob.data() + h.off_nodes16, nodes16.data(), nodes16.size() * sizeof(DNode16));
    return blob;
}

}  // namespace dtof
