// mesh_io.cpp -- the two triangle-mesh file formats of the reference's tutorial scenes, and the vertex baking the
// mesh plugins do in their constructors.
//
//   load_obj   OBJMesh  (src/shapes/obj.cpp:139-398): `v`, `vn`, `vt`, `f`; vertices de-duplicated by their (v, vt, vn)
//              triple in order of first use, polygons fan-triangulated, `flip_tex_coords` (default true)
//   load_ply   PLYMesh  (src/shapes/ply.cpp:160-441): ascii / binary_little_endian / binary_big_endian; typed vertex
//              properties x y z [nx ny nz] [u v | texture_u texture_v | s t]; a `vertex_index(/indices)` list that must
//              hold triangles; unknown elements are skipped; trailing content is an error
//   bake_mesh  positions through to_world, normals through its inverse transpose + normalise (obj.cpp:218-246,
//              ply.cpp:284-300); without normals (and without face_normals) Mesh::recompute_vertex_normals
//              (src/render/mesh.cpp:257-345): angle-weighted face normals -- the reference adds them with unordered
//              float atomics, here they are accumulated in face order in double and rounded once
#include "dtof_scene.h"
#include <zlib.h>
#include "dtof_math.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <unordered_map>

namespace dtof {

namespace {
[[noreturn]] void mesh_fail(const char *kind, const std::string &name, const std::string &msg, bool bang) {
    throw std::runtime_error(std::string("Error while loading ") + kind + " file \"" + name + "\": " + msg + (bang ? "!" : ""));
}
std::string base_name(const std::string &p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? p : p.substr(k + 1); }
bool slurp(const std::string &path, std::string &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss; ss << f.rdbuf(); out = ss.str(); return true;
}
inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }

struct Key3 { uint32_t a, b, c; bool operator==(const Key3 &o) const { return a == o.a && b == o.b && c == o.c; } };
struct Key3Hash { size_t operator()(const Key3 &k) const { return ((size_t) k.a * 0x9e3779b97f4a7c15ull) ^ ((size_t) k.b << 21) ^ ((size_t) k.c << 42); } };
}  // namespace

RawMesh load_obj(const std::string &path, bool flip_tex_coords, bool face_normals) {
    const std::string name = base_name(path);
    auto fail = [&](const std::string &m) { mesh_fail("OBJ", name, m, false); };
    std::string text;
    if (!slurp(path, text)) fail("file not found");
    std::vector<float> vs, ns, ts;
    std::vector<Key3> keys; std::unordered_map<Key3, uint32_t, Key3Hash> key_index;
    RawMesh out;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos); if (eol == std::string::npos) eol = text.size();
        std::string line = text.substr(pos, eol - pos); pos = eol + 1;
        const char *cur = line.c_str();
        while (*cur && is_blank(*cur)) ++cur;
        if (!cur[0] || !cur[1]) continue;
        auto read_floats = [&](const char *p, int n, std::vector<float> &dst) {
            for (int i = 0; i < n; ++i) {
                char *end; float v = strtof(p, &end);
                if (end == p) fail("could not parse line \"" + line + "\"");
                dst.push_back(v); p = end;
            }
        };
        if (cur[0] == 'v' && (cur[1] == ' ' || cur[1] == '\t')) read_floats(cur + 2, 3, vs);
        else if (cur[0] == 'v' && cur[1] == 'n' && (cur[2] == ' ' || cur[2] == '\t')) { if (!face_normals) read_floats(cur + 3, 3, ns); }
        else if (cur[0] == 'v' && cur[1] == 't' && (cur[2] == ' ' || cur[2] == '\t')) read_floats(cur + 3, 2, ts);
        else if (cur[0] == 'f' && (cur[1] == ' ' || cur[1] == '\t')) {
            cur += 2;
            uint32_t tri[3] = { 0, 0, 0 }; size_t count = 0;
            for (;;) {
                while (*cur && is_blank(*cur)) ++cur;
                if (!*cur) break;
                Key3 key { 0, 0, 0 }; uint32_t *slot[3] = { &key.a, &key.b, &key.c }; int ti = 0;
                for (;;) {
                    if (*cur >= '0' && *cur <= '9') { char *end; *slot[ti] = (uint32_t) strtoul(cur, &end, 10); cur = end; }
                    else if (ti == 0) fail("could not parse line \"" + line + "\"");
                    if (*cur == '/') { if (++ti > 2) fail("could not parse line \"" + line + "\""); ++cur; continue; }
                    break;
                }
                if (*cur && !is_blank(*cur)) fail("could not parse line \"" + line + "\"");
                if (key.a < 1 || key.a - 1 >= vs.size() / 3) fail("reference to invalid vertex " + std::to_string(key.a) + "!");
                auto it = key_index.find(key); uint32_t id;
                if (it == key_index.end()) { id = (uint32_t) keys.size(); key_index.emplace(key, id); keys.push_back(key); } else id = it->second;
                if (count < 3) tri[count] = id; else { tri[1] = tri[2]; tri[2] = id; }
                if (++count >= 3) { out.faces.push_back(tri[0]); out.faces.push_back(tri[1]); out.faces.push_back(tri[2]); }
            }
        }
    }
    if (flip_tex_coords) for (size_t i = 1; i < ts.size(); i += 2) ts[i] = 1.f - ts[i];
    const size_t nv = keys.size();
    out.positions.assign(nv * 3, 0.f);
    out.has_normals = !ns.empty() && !face_normals; out.has_texcoords = !ts.empty();
    if (out.has_normals) out.normals.assign(nv * 3, 0.f);
    if (out.has_texcoords) out.texcoords.assign(nv * 2, 0.f);
    for (size_t i = 0; i < nv; ++i) {
        const Key3 &k = keys[i];
        memcpy(&out.positions[3 * i], &vs[3 * (size_t) (k.a - 1)], 12);
        if (k.b) {
            if (k.b - 1 >= ts.size() / 2) fail("reference to invalid texture coordinate " + std::to_string(k.b) + "!");
            memcpy(&out.texcoords[2 * i], &ts[2 * (size_t) (k.b - 1)], 8);
        }
        if (!face_normals && k.c) {
            if (k.c - 1 >= ns.size() / 3) fail("reference to invalid normal " + std::to_string(k.c) + "!");
            memcpy(&out.normals[3 * i], &ns[3 * (size_t) (k.c - 1)], 12);
        }
    }
    return out;
}

namespace {
struct PlyProp { bool list = false; int count_type = 0, type = 0; std::string name; };
struct PlyElement { std::string name; size_t count = 0; std::vector<PlyProp> props; };
// type codes: 0 i8, 1 u8, 2 i16, 3 u16, 4 i32, 5 u32, 6 f32, 7 f64
int ply_type(const std::string &t) {
    static const std::pair<const char *, int> tab[] = { { "char", 0 }, { "int8", 0 }, { "uchar", 1 }, { "uint8", 1 }, { "short", 2 }, { "int16", 2 },
        { "ushort", 3 }, { "uint16", 3 }, { "int", 4 }, { "int32", 4 }, { "uint", 5 }, { "uint32", 5 }, { "float", 6 }, { "float32", 6 },
        { "double", 7 }, { "float64", 7 } };
    for (auto &e : tab) if (t == e.first) return e.second;
    return -1;
}
const int kPlySize[8] = { 1, 1, 2, 2, 4, 4, 4, 8 };
// one binary value -> double (exact for every PLY type) ; `swap` = file is big endian
double ply_read(const uint8_t *p, int type, bool swap) {
    uint8_t b[8]; int n = kPlySize[type];
    for (int i = 0; i < n; ++i) b[i] = swap ? p[n - 1 - i] : p[i];
    switch (type) {
        case 0: { int8_t v; memcpy(&v, b, 1); return v; }   case 1: { uint8_t v; memcpy(&v, b, 1); return v; }
        case 2: { int16_t v; memcpy(&v, b, 2); return v; }  case 3: { uint16_t v; memcpy(&v, b, 2); return v; }
        case 4: { int32_t v; memcpy(&v, b, 4); return v; }  case 5: { uint32_t v; memcpy(&v, b, 4); return v; }
        case 6: { float v; memcpy(&v, b, 4); return v; }    default: { double v; memcpy(&v, b, 8); return v; }
    }
}
}  // namespace

RawMesh load_ply(const std::string &path, bool face_normals) {
    const std::string name = base_name(path);
    auto fail = [&](const std::string &m) { mesh_fail("PLY", name, m, true); };
    std::string data;
    if (!slurp(path, data)) fail("file not found");
    size_t end = data.find("end_header");
    if (data.compare(0, 3, "ply") != 0 || end == std::string::npos) fail("invalid PLY header");
    size_t eol = data.find('\n', end); if (eol == std::string::npos) eol = data.size() - 1;
    std::istringstream hs(data.substr(0, end));
    std::string line, fmt; std::vector<PlyElement> elements;
    std::getline(hs, line);   // "ply"
    while (std::getline(hs, line)) {
        std::istringstream ls(line); std::vector<std::string> t; std::string w;
        while (ls >> w) t.push_back(w);
        if (t.empty() || t[0] == "comment" || t[0] == "obj_info") continue;
        if (t[0] == "format" && t.size() >= 2) fmt = t[1];
        else if (t[0] == "element" && t.size() >= 3) { PlyElement e; e.name = t[1]; e.count = (size_t) strtoull(t[2].c_str(), nullptr, 10); elements.push_back(e); }
        else if (t[0] == "property" && t.size() >= 3) {
            if (elements.empty()) fail("property before element");
            PlyProp p;
            if (t[1] == "list") { if (t.size() < 5) fail("invalid PLY header"); p.list = true; p.count_type = ply_type(t[2]); p.type = ply_type(t[3]); p.name = t[4]; }
            else { p.type = ply_type(t[1]); p.name = t[2]; }
            if (p.type < 0 || p.count_type < 0) fail("invalid PLY header: unknown type \"" + (p.list && p.count_type < 0 ? t[2] : t[p.list ? 3 : 1]) + "\"");
            elements.back().props.push_back(p);
        } else fail("invalid PLY header: unknown token \"" + t[0] + "\"");
    }
    const bool ascii = fmt == "ascii", big = fmt == "binary_big_endian";
    if (!ascii && !big && fmt != "binary_little_endian") fail("invalid PLY header: unknown format");
    const uint8_t *body = (const uint8_t *) data.data() + eol + 1; const size_t body_size = data.size() - (eol + 1);
    size_t off = 0;
    // ascii: whitespace-separated tokens
    const char *tp = (const char *) body, *tend = tp + body_size;
    auto next_token = [&](std::string &tok) -> bool {
        while (tp < tend && (is_blank(*tp) || *tp == '\n')) ++tp;
        if (tp >= tend) return false;
        const char *b = tp; while (tp < tend && !is_blank(*tp) && *tp != '\n') ++tp;
        tok.assign(b, tp - b); return true;
    };
    // integer view of a value read as floating point: out-of-range / NaN become an index no mesh has (checked by the callers)
    auto to_u64 = [](double v) -> uint64_t { return v >= 0.0 && v < 18446744073709549568.0 ? (uint64_t) v : ~0ull; };
    auto ascii_value = [&](int type, float &f, uint64_t &u) {
        std::string tok; if (!next_token(tok)) fail("unexpected end of file");
        if (type >= 6) { f = strtof(tok.c_str(), nullptr); u = to_u64((double) f); }
        else { long long v = strtoll(tok.c_str(), nullptr, 10); u = (uint64_t) v; f = (float) (double) v; }
    };
    auto binary_value = [&](int type, float &f, uint64_t &u) {
        if (off + (size_t) kPlySize[type] > body_size) fail("unexpected end of file");
        double v = ply_read(body + off, type, big); off += kPlySize[type];
        f = (float) v; u = to_u64(v);
    };
    RawMesh out; bool have_vertices = false, have_faces = false;
    for (auto &el : elements) {   // a (corrupted) count the file cannot hold would otherwise be allocated before the first read fails
        size_t min_bytes = 0;
        for (auto &p : el.props) min_bytes += ascii ? 2 : (size_t) kPlySize[p.list ? p.count_type : p.type];
        if (el.count > body_size / std::max<size_t>(1, min_bytes)) fail("invalid PLY header: element \"" + el.name + "\" has more entries than the file can hold");
    }
    for (auto &el : elements) {
        if (el.name == "vertex") {
            int ix[8] = { -1, -1, -1, -1, -1, -1, -1, -1 };   // x y z nx ny nz u v
            auto find = [&](const char *n) { for (size_t i = 0; i < el.props.size(); ++i) if (el.props[i].name == n) return (int) i; return -1; };
            for (auto &p : el.props) if (p.list) fail("incompatible contents -- is this a triangle mesh?");
            const char *xyz[3] = { "x", "y", "z" }, *nn[3] = { "nx", "ny", "nz" };
            for (int k = 0; k < 3; ++k) { ix[k] = find(xyz[k]); if (ix[k] < 0) fail(std::string("Unable to find field \"") + xyz[k] + "\""); ix[3 + k] = find(nn[k]); }
            ix[6] = find("u"); ix[7] = find("v");
            if (ix[6] < 0 || ix[7] < 0) { int a = find("texture_u"), b = find("texture_v"); if (a >= 0 && b >= 0) { ix[6] = a; ix[7] = b; } }
            if (ix[6] < 0 || ix[7] < 0) { int a = find("s"), b = find("t"); if (a >= 0 && b >= 0) { ix[6] = a; ix[7] = b; } }
            out.has_normals = !face_normals && ix[3] >= 0 && ix[4] >= 0 && ix[5] >= 0;
            out.has_texcoords = ix[6] >= 0 && ix[7] >= 0;
            out.positions.resize(el.count * 3);
            if (out.has_normals) out.normals.resize(el.count * 3);
            if (out.has_texcoords) out.texcoords.resize(el.count * 2);
            std::vector<float> row(el.props.size());
            for (size_t i = 0; i < el.count; ++i) {
                for (size_t j = 0; j < el.props.size(); ++j) { uint64_t u; if (ascii) ascii_value(el.props[j].type, row[j], u); else binary_value(el.props[j].type, row[j], u); }
                for (int k = 0; k < 3; ++k) out.positions[3 * i + k] = row[ix[k]];
                if (out.has_normals) for (int k = 0; k < 3; ++k) out.normals[3 * i + k] = row[ix[3 + k]];
                if (out.has_texcoords) { out.texcoords[2 * i] = row[ix[6]]; out.texcoords[2 * i + 1] = row[ix[7]]; }
            }
            have_vertices = true;
        } else {
            const bool is_face = el.name == "face";
            int li = -1;
            if (is_face) {
                for (size_t j = 0; j < el.props.size(); ++j)
                    if (el.props[j].list && (el.props[j].name == "vertex_index" || el.props[j].name == "vertex_indices")) { li = (int) j; break; }
                if (li < 0) fail("vertex_index/vertex_indices property not found");
                out.faces.resize(el.count * 3);
            }
            for (size_t i = 0; i < el.count; ++i)
                for (size_t j = 0; j < el.props.size(); ++j) {
                    const PlyProp &p = el.props[j]; float f; uint64_t u;
                    if (!p.list) { if (ascii) ascii_value(p.type, f, u); else binary_value(p.type, f, u); continue; }
                    if (ascii) ascii_value(p.count_type, f, u); else binary_value(p.count_type, f, u);
                    const uint64_t cnt = u;
                    if (is_face && (int) j == li && cnt != 3) fail("incompatible contents -- is this a triangle mesh?");
                    for (uint64_t k = 0; k < cnt; ++k) {
                        if (ascii) ascii_value(p.type, f, u); else binary_value(p.type, f, u);
                        if (is_face && (int) j == li) {
                            if (u > 0xffffffffull) fail("mesh face references a vertex out of range");
                            out.faces[3 * i + k] = (uint32_t) u;
                        }
                    }
                }
            if (is_face) have_faces = true;
        }
    }
    if (ascii) { std::string tok; if (next_token(tok)) fail("invalid file -- trailing content"); }
    else if (off != body_size) fail("invalid file -- trailing content");
    if (!have_vertices || !have_faces) fail("vertex or face element missing");
    return out;
}

void bake_mesh(HostShape &s, const RawMesh &raw) {
    const size_t nv = raw.positions.size() / 3, nf = raw.faces.size() / 3;
    for (uint32_t f : raw.faces) if (f >= nv) throw std::runtime_error("mesh face references a vertex out of range");
    s.positions.resize(nv * 3); s.faces = raw.faces;
    s.texcoords = raw.has_texcoords ? raw.texcoords : std::vector<float>();
    for (size_t i = 0; i < nv; ++i) {
        V3 p = xf_point(s.to_world, mk(raw.positions[3 * i], raw.positions[3 * i + 1], raw.positions[3 * i + 2]));
        s.positions[3 * i] = p.x; s.positions[3 * i + 1] = p.y; s.positions[3 * i + 2] = p.z;
    }
    s.normals.clear();
    if (s.face_normals) return;
    s.normals.resize(nv * 3);
    if (raw.has_normals) {
        for (size_t i = 0; i < nv; ++i) {
            V3 n = normalize(xf_normal(s.to_object, mk(raw.normals[3 * i], raw.normals[3 * i + 1], raw.normals[3 * i + 2])));
            s.normals[3 * i] = n.x; s.normals[3 * i + 1] = n.y; s.normals[3 * i + 2] = n.z;
        }
        return;
    }
    std::vector<double> acc(nv * 3, 0.0);
    for (size_t f = 0; f < nf; ++f) {
        const uint32_t *fi = &raw.faces[3 * f];
        double v[3][3];
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 3; ++c) v[k][c] = (double) s.positions[3 * (size_t) fi[k] + c];
        double s0[3], s1[3], n[3];
        for (int c = 0; c < 3; ++c) { s0[c] = v[1][c] - v[0][c]; s1[c] = v[2][c] - v[0][c]; }
        n[0] = s0[1] * s1[2] - s0[2] * s1[1]; n[1] = s0[2] * s1[0] - s0[0] * s1[2]; n[2] = s0[0] * s1[1] - s0[1] * s1[0];
        double l2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
        if (!(l2 > 0.0)) continue;
        double il = 1.0 / std::sqrt(l2);
        for (int c = 0; c < 3; ++c) n[c] *= il;
        for (int k = 0; k < 3; ++k) {
            double d0[3], d1[3], l0 = 0, l1 = 0, dt = 0;
            for (int c = 0; c < 3; ++c) { d0[c] = v[(k + 1) % 3][c] - v[k][c]; d1[c] = v[(k + 2) % 3][c] - v[k][c]; l0 += d0[c] * d0[c]; l1 += d1[c] * d1[c]; }
            l0 = 1.0 / std::sqrt(l0); l1 = 1.0 / std::sqrt(l1);
            for (int c = 0; c < 3; ++c) dt += (d0[c] * l0) * (d1[c] * l1);
            double ang = std::acos(dt > 1.0 ? 1.0 : (dt < -1.0 ? -1.0 : dt));
            for (int c = 0; c < 3; ++c) acc[3 * (size_t) fi[k] + c] += n[c] * ang;
        }
    }
    for (size_t i = 0; i < nv; ++i) {
        double *a = &acc[3 * i], l = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        if (l != 0.0 && l == l) { s.normals[3 * i] = (float) (a[0] / l); s.normals[3 * i + 1] = (float) (a[1] / l); s.normals[3 * i + 2] = (float) (a[2] / l); }
        else { s.normals[3 * i] = 1.f; s.normals[3 * i + 1] = 0.f; s.normals[3 * i + 2] = 0.f; }
    }
}

// ---------------------------------------------------------------------------- .serialized (src/shapes/serialized.cpp:234-390)
// Little-endian stream: uint16 0x041C, uint16 version (3 | 4), then ONE zlib stream per sub-mesh: uint32 flags, [v4: zero-terminated
// name], uint64 vertex count, uint64 face count, positions (3 floats or doubles per vertex), [normals], [texcoords], [colours],
// uint32 indices.  The file ends with the offsets of its sub-meshes (uint64 in v4, uint32 in v3) and their count (uint32).
RawMesh load_serialized(const std::string &path, int shape_index, bool face_normals) {
    const std::string name = base_name(path);
    auto fail = [&](const std::string &m) { throw std::runtime_error("Error while loading serialized file \"" + name + "\": " + m + "!"); };
    std::string data;
    if (!slurp(path, data)) fail("file not found");
    if (shape_index < 0) fail("shape index must be nonnegative!");
    auto rd16 = [&](size_t at) { if (at > data.size() || data.size() - at < 2) fail("premature end of file"); uint16_t v; memcpy(&v, &data[at], 2); return v; };
    auto rd32 = [&](size_t at) { if (at > data.size() || data.size() - at < 4) fail("premature end of file"); uint32_t v; memcpy(&v, &data[at], 4); return v; };
    auto rd64 = [&](size_t at) { if (at > data.size() || data.size() - at < 8) fail("premature end of file"); uint64_t v; memcpy(&v, &data[at], 8); return v; };
    const uint16_t format = rd16(0), version = rd16(2);
    if (format != 0x041C) fail("encountered an invalid file format!");
    if (version != 3 && version != 4) fail("encountered an incompatible file version!");
    size_t start = 4;
    if (shape_index != 0) {
        const uint32_t count = rd32(data.size() - 4);
        if ((uint32_t) shape_index > count) fail("Unable to unserialize mesh, shape index is out of range! (requested " + std::to_string(shape_index) + " out of 0.." + std::to_string((int) count - 1) + ")");
        const size_t back = version == 4 ? 8 * (size_t) (count - (uint32_t) shape_index) + 4 : 4 * ((size_t) (count - (uint32_t) shape_index) + 1);
        if (back > data.size()) fail("premature end of file");
        const size_t off = version == 4 ? (size_t) rd64(data.size() - back) : (size_t) rd32(data.size() - back);
        if (off > data.size()) fail("premature end of file");
        start = off + 4;   // the sub-mesh repeats the 4-byte header
    }
    if (start > data.size()) fail("premature end of file");
    // inflate the sub-mesh's zlib stream (its compressed length is not stored)
    std::string raw;
    {
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (inflateInit(&zs) != Z_OK) fail("inflateInit failed");
        zs.next_in = (Bytef *) &data[start]; zs.avail_in = (uInt) std::min<size_t>(data.size() - start, 0xffffffffu);
        char buf[1 << 16]; int rc;
        do {
            zs.next_out = (Bytef *) buf; zs.avail_out = sizeof buf;
            rc = inflate(&zs, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); fail("inflate(): stream error"); }
            raw.append(buf, sizeof buf - zs.avail_out);
        } while (rc != Z_STREAM_END);
        inflateEnd(&zs);
    }
    size_t pos = 0;
    auto need = [&](size_t n) { if (pos + n > raw.size()) fail("premature end of the compressed stream"); };
    need(4); uint32_t flags; memcpy(&flags, &raw[pos], 4); pos += 4;
    if (version == 4) { while (true) { need(1); if (raw[pos++] == 0) break; } }
    need(16); uint64_t nv, nf; memcpy(&nv, &raw[pos], 8); memcpy(&nf, &raw[pos + 8], 8); pos += 16;
    if (nv > (1ull << 31) || nf > (1ull << 31)) fail("implausible vertex / face count");
    const bool dp = flags & 0x2000, has_n = flags & 0x0001, has_uv = flags & 0x0002, has_col = flags & 0x0008;
    auto read_floats = [&](std::vector<float> *dst, size_t dim) {
        const size_t n = (size_t) nv * dim;
        need(n * (dp ? 8 : 4));
        if (dst) {
            dst->resize(n);
            if (dp) for (size_t i = 0; i < n; ++i) { double v; memcpy(&v, &raw[pos + 8 * i], 8); (*dst)[i] = (float) v; }
            else memcpy(dst->data(), &raw[pos], n * 4);
        }
        pos += n * (dp ? 8 : 4);
    };
    RawMesh m;
    read_floats(&m.positions, 3);
    if (has_n) read_floats(face_normals ? nullptr : &m.normals, 3);
    if (has_uv) read_floats(&m.texcoords, 2);
    if (has_col) read_floats(nullptr, 3);
    need((size_t) nf * 12);
    m.faces.resize((size_t) nf * 3); memcpy(m.faces.data(), &raw[pos], (size_t) nf * 12);
    for (uint32_t f : m.faces) if (f >= nv) fail("face references a vertex out of range");
    m.has_normals = has_n && !face_normals; m.has_texcoords = has_uv;
    return m;
}

}  // namespace dtof
