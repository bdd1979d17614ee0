// scene_loader.cpp -- scene.xml front end for the Doppler-ToF hot path.
//
// Implements the subset of Mitsuba 3's scene format that the reference's hot path
// consumes, with the reference's semantics (paths relative to the reference root):
//   <default>/$param substitution             src/core/xml.cpp:441-456, 630-648
//   <transform>/<animation> composition       src/core/xml.cpp:882-1007 (ops left-multiply, double precision)
//   <rgb>                                     src/core/xml.cpp:792-822
//   animated shape -> shapegroup + instance   src/core/xml.cpp:1165-1195
//   keyframes cast to float32                 src/core/transform.cpp:22-36, include/mitsuba/core/transform.h:384
//   plugin constructors                       src/integrators/dopplertofpath.cpp:19-57, src/render/integrator.cpp:54-100,568-585,
//                                             src/samplers/correlated.cpp:17-23, src/render/sampler.cpp:11-20,
//                                             src/render/sensor.cpp:14-20,127-203, src/sensors/perspective.cpp:139-152,
//                                             src/render/film.cpp:7-54, src/rfilters/tent.cpp:47-55, src/shapes/cube.cpp:114-160,
//                                             src/shapes/rectangle.cpp:91-113
// Unsupported plugins / tags raise std::runtime_error (surfaced through the C ABI as a
// status code + dtof_last_error()).
#include "dtof_scene.h"
#include "dtof_math.h"
#include <cmath>
#include <cstring>
#include <sys/stat.h>
#include <fstream>
#include <sstream>
#include <memory>
#include <algorithm>
#include <set>
#include <initializer_list>

namespace dtof {

[[noreturn]] static void fail(const std::string &msg) { throw std::runtime_error(msg); }
// xml.cpp:1204-1222: `unreferenced property "["a"]" in bsdf plugin of type "diffuse"` -- the reference prints the list of names (a std::vector
// of quoted strings) inside the quotes of its format string
[[noreturn]] static void fail_unreferenced(const std::vector<std::string> &names, const std::string &kind, const std::string &plugin) {
    std::string list = "[";
    for (size_t i = 0; i < names.size(); ++i) list += (i ? ", \"" : "\"") + names[i] + "\"";
    fail(std::string("unreferenced ") + (names.size() > 1 ? "properties" : "property") + " \"" + list + "]\" in " + kind + " plugin of type \"" + plugin + "\"");
}

// ---------------------------------------------------------------------------- tiny XML DOM
struct XNode {
    std::string tag;
    size_t offset = 0;   // position of the tag name in the source text (what pugixml's offset_debug() reports: errors quote it as "line L, col C")
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XNode>> children;
    const std::string *attr(const char *n) const {
        for (auto &a : attrs) if (a.first == n) return &a.second;
        return nullptr;
    }
    std::string get(const char *n, const std::string &def = "") const { auto *a = attr(n); return a ? *a : def; }
};

struct XParser {
    const std::string &s; size_t p = 0;
    explicit XParser(const std::string &text) : s(text) {}
    void skip_ws() { while (p < s.size() && isspace((unsigned char) s[p])) ++p; }
    bool starts(const char *lit) const { return s.compare(p, strlen(lit), lit) == 0; }
    void skip_misc() {
        for (;;) {
            skip_ws();
            if (starts("<!--")) { size_t e = s.find("-->", p); if (e == std::string::npos) fail("xml: unterminated comment"); p = e + 3; }
            else if (starts("<?")) { size_t e = s.find("?>", p); if (e == std::string::npos) fail("xml: unterminated declaration"); p = e + 2; }
            else if (starts("<!DOCTYPE")) { size_t e = s.find('>', p); if (e == std::string::npos) fail("xml: bad doctype"); p = e + 1; }
            else break;
        }
    }
    static std::string unescape(const std::string &v) {
        std::string o; o.reserve(v.size());
        for (size_t i = 0; i < v.size(); ++i) {
            if (v[i] == '&') {
                if (!v.compare(i, 4, "&lt;")) { o += '<'; i += 3; }
                else if (!v.compare(i, 4, "&gt;")) { o += '>'; i += 3; }
                else if (!v.compare(i, 5, "&amp;")) { o += '&'; i += 4; }
                else if (!v.compare(i, 6, "&quot;")) { o += '"'; i += 5; }
                else if (!v.compare(i, 6, "&apos;")) { o += '\''; i += 5; }
                else o += v[i];
            } else o += v[i];
        }
        return o;
    }
    std::string name() {
        size_t b = p;
        while (p < s.size() && (isalnum((unsigned char) s[p]) || s[p] == '_' || s[p] == '-' || s[p] == ':' || s[p] == '.')) ++p;
        if (p == b) fail("xml: expected a name at offset " + std::to_string(p));
        return s.substr(b, p - b);
    }
    std::unique_ptr<XNode> element() {
        if (p >= s.size() || s[p] != '<') fail("xml: expected '<' at offset " + std::to_string(p));
        ++p;
        auto n = std::make_unique<XNode>();
        n->offset = p;
        n->tag = name();
        for (;;) {
            skip_ws();
            if (p >= s.size()) fail("xml: unexpected end of input in <" + n->tag + ">");
            if (s[p] == '/') { if (p + 1 >= s.size() || s[p + 1] != '>') fail("xml: malformed tag"); p += 2; return n; }
            if (s[p] == '>') { ++p; break; }
            std::string an = name();
            skip_ws();
            if (p >= s.size() || s[p] != '=') fail("xml: expected '=' after attribute " + an);
            ++p; skip_ws();
            char q = p < s.size() ? s[p] : 0;
            if (q != '"' && q != '\'') fail("xml: expected quoted value for attribute " + an);
            size_t e = s.find(q, p + 1);
            if (e == std::string::npos) fail("xml: unterminated attribute value");
            n->attrs.emplace_back(an, unescape(s.substr(p + 1, e - p - 1)));
            p = e + 1;
        }
        for (;;) {
            skip_misc();
            if (p >= s.size()) fail("xml: missing </" + n->tag + ">");
            if (starts("</")) {
                p += 2; std::string cn = name(); skip_ws();
                if (cn != n->tag || p >= s.size() || s[p] != '>') fail("xml: mismatched closing tag </" + cn + "> for <" + n->tag + ">");
                ++p; return n;
            }
            if (s[p] != '<') fail("unexpected content");   // xml.cpp:463-464
            n->children.push_back(element());
        }
    }
    std::unique_ptr<XNode> document() { skip_misc(); auto r = element(); skip_misc(); return r; }
};

// ---------------------------------------------------------------------------- well-formedness of the scene description
// The checks parse_xml makes on every node before it looks at values (src/core/xml.cpp:470-560,258-310), with its messages:
// `Error while loading "<id>" (at line L, col C): <message>.` (XMLSource::throw_error, xml.cpp:213-217; the position is the tag name's).
// Pinned by the reference's own tests (src/core/tests/test_xml.py -> tests/golden/reference_xml_cases.json).
static thread_local const std::string *g_xml_text = nullptr;
static thread_local std::string g_xml_id = "<string>";
static std::string position_of(size_t offset) {
    if (!g_xml_text) return "byte offset " + std::to_string(offset);
    size_t line = 0, line_start = 0;
    for (size_t i = 0; i < offset && i < g_xml_text->size(); ++i) if ((*g_xml_text)[i] == '\n') { ++line; line_start = i + 1; }
    return "line " + std::to_string(line + 1) + ", col " + std::to_string(offset - line_start + 1);   // string_offset (xml.cpp:161-177), pugixml counts the '<'
}
[[noreturn]] static void fail_at(const XNode &n, const std::string &msg) {
    fail("Error while loading \"" + g_xml_id + "\" (at " + position_of(n.offset) + "): " + msg + ".");
}
enum TagKind { TAG_INVALID, TAG_OBJECT, TAG_PROPERTY, TAG_VECTOR, TAG_TRANSFORM, TAG_ANIMATION, TAG_TRANSFORM_OP, TAG_REF, TAG_DEFAULT, TAG_PATH, TAG_INCLUDE, TAG_ALIAS };
static TagKind tag_kind(const XNode &n) {
    static const char *objects[] = { "scene", "integrator", "sensor", "sampler", "film", "rfilter", "bsdf", "shape", "emitter", "texture", "medium", "phase", "volume" };
    for (auto *x : objects) if (n.tag == x) return TAG_OBJECT;
    if (n.tag == "spectrum" && n.attr("type")) return TAG_OBJECT;   // a tag with a `type` attribute that names a plugin class is an object (xml.cpp:480-483)
    for (auto *x : { "float", "integer", "boolean", "string", "rgb", "spectrum" }) if (n.tag == x) return TAG_PROPERTY;
    if (n.tag == "point" || n.tag == "vector") return TAG_VECTOR;
    if (n.tag == "transform") return TAG_TRANSFORM;
    if (n.tag == "animation") return TAG_ANIMATION;
    for (auto *x : { "translate", "rotate", "scale", "lookat", "matrix" }) if (n.tag == x) return TAG_TRANSFORM_OP;
    if (n.tag == "ref") return TAG_REF;
    if (n.tag == "default") return TAG_DEFAULT;
    if (n.tag == "path") return TAG_PATH;
    if (n.tag == "include") return TAG_INCLUDE;
    if (n.tag == "alias") return TAG_ALIAS;
    return TAG_INVALID;
}
// check_attributes (xml.cpp:273-287): every attribute must be one of `allowed`; with expect_all (or with no attribute at all) none may be missing
static void check_attributes(const XNode &n, std::vector<std::string> allowed, bool expect_all = true, bool may_be_empty = false) {
    bool found_one = may_be_empty;   // may_be_empty: attributes the reference adds itself when they are missing (`id`, `name` of objects and references)
    for (auto &a : n.attrs) {
        auto it = std::find(allowed.begin(), allowed.end(), a.first);
        if (it == allowed.end()) fail_at(n, "unexpected attribute \"" + a.first + "\" in element \"" + n.tag + "\"");
        allowed.erase(it); found_one = true;
    }
    if (!allowed.empty() && (!found_one || expect_all)) {
        std::sort(allowed.begin(), allowed.end());   // the reference keeps the names in a std::set and reports the first one left
        fail_at(n, "missing attribute \"" + allowed.front() + "\" in element \"" + n.tag + "\"");
    }
}
// upgrade_tree (xml.cpp:338-365) for scene descriptions older than 2.0.0: camelCase property names become underscore_case, <lookAt> becomes
// <lookat>, ids with a leading underscore are renamed, diffuse BSDFs' `diffuse_reflectance` becomes `reflectance`
static void upgrade_tree(XNode &n, const XNode *parent) {
    if (n.tag == "lookAt") n.tag = "lookat";
    for (auto &a : n.attrs) {
        if (a.first == "name" && n.tag != "default") {
            std::string &name = a.second;
            for (size_t i = 0; i + 1 < name.size(); ++i) {
                if (islower((unsigned char) name[i]) && isupper((unsigned char) name[i + 1])) {
                    name = name.substr(0, i + 1) + "_" + name.substr(i + 1);
                    i += 2;
                    while (i < name.size() && isupper((unsigned char) name[i])) { name[i] = (char) tolower((unsigned char) name[i]); ++i; }
                }
            }
            if (name == "diffuse_reflectance" && parent && parent->tag == "bsdf" && parent->get("type") == "diffuse") name = "reflectance";
        }
        if (a.first == "id" && !a.second.empty() && a.second[0] == '_') a.second = "ID" + a.second + "__UPGR";
    }
    for (auto &c : n.children) upgrade_tree(*c, &n);
}
struct CheckCtx { std::map<std::string, size_t> ids; };
static void check_tree(XNode &n, TagKind parent, int depth, CheckCtx &cc) {
    const TagKind kind = tag_kind(n);
    if (kind == TAG_INVALID) fail_at(n, "unexpected tag \"" + n.tag + "\"");
    const bool has_parent = parent != TAG_INVALID, parent_is_object = parent == TAG_OBJECT, parent_is_transform = parent == TAG_TRANSFORM;
    if (!has_parent && kind != TAG_OBJECT) fail_at(n, "root element \"" + n.tag + "\" must be an object");
    if (parent_is_transform != (kind == TAG_TRANSFORM_OP))
        fail_at(n, parent_is_transform ? "transform nodes can only contain transform operations" : "transform operations can only occur in a transform node");
    if (has_parent && !parent_is_object && !((parent_is_transform && kind == TAG_TRANSFORM_OP) || (parent == TAG_ANIMATION && kind == TAG_TRANSFORM)))
        fail_at(n, "node \"" + n.tag + "\" cannot occur as child of a property");
    if (depth == 0 && !n.attr("version")) fail_at(n, "missing version attribute in root element \"" + n.tag + "\"");
    if (auto *v = n.attr("version")) {
        unsigned major = 0, minor = 0, patch = 0; char tail = 0;
        if (sscanf(v->c_str(), "%u.%u.%u%c", &major, &minor, &patch, &tail) != 3) fail_at(n, "could not parse version number \"" + *v + "\"");
        if (major < 2) upgrade_tree(n, nullptr);
        for (size_t i = 0; i < n.attrs.size(); ++i) if (n.attrs[i].first == "version") { n.attrs.erase(n.attrs.begin() + (long) i); break; }
    }
    if (auto *name = n.attr("name")) {
        if (!name->empty() && (*name)[0] == '_')
            fail_at(n, "invalid parameter name \"" + *name + "\" in element \"" + n.tag + "\": leading underscores are reserved for internal identifiers");
    }
    if (auto *id = n.attr("id")) {
        if (!id->empty() && (*id)[0] == '_')
            fail_at(n, "invalid id \"" + *id + "\" in element \"" + n.tag + "\": leading underscores are reserved for internal identifiers");
    }
    switch (kind) {
        case TAG_OBJECT: {
            std::vector<std::string> allowed = { "id", "name" };
            if (n.tag != "scene") allowed.push_back("type");
            check_attributes(n, allowed, false, true);
            if (n.tag != "scene" && !n.attr("type")) fail_at(n, "missing attribute \"type\" in element \"" + n.tag + "\"");
            if (auto *id = n.attr("id")) {
                auto prev = cc.ids.find(*id);
                if (prev != cc.ids.end()) fail_at(n, "\"" + n.tag + "\" has duplicate id \"" + *id + "\" (previous was at " + position_of(prev->second) + ")");
            }
            std::vector<std::string> names;   // Properties::set_*: a name may be given once (properties.cpp:139-146)
            for (auto &c : n.children) {
                check_tree(*c, TAG_OBJECT, depth + 1, cc);
                const TagKind ck = tag_kind(*c);
                if (auto *cn = c->attr("name")) if (!cn->empty() && ck != TAG_DEFAULT) {
                    if (std::find(names.begin(), names.end(), *cn) != names.end()) fail_at(*c, "Property \"" + *cn + "\" was specified multiple times!");
                    names.push_back(*cn);
                }
            }
            if (auto *id = n.attr("id")) cc.ids[*id] = n.offset;
            return;
        }
        case TAG_REF: check_attributes(n, { "id", "name" }, false, true); if (!n.attr("id")) fail_at(n, "missing attribute \"id\" in element \"ref\""); break;
        case TAG_ALIAS: check_attributes(n, { "id", "as" }); break;
        case TAG_DEFAULT: {
            check_attributes(n, { "name", "value" });
            if (n.get("name").empty()) fail_at(n, "<default>: name must by nonempty");
            if (n.get("name").find(',') != std::string::npos) fail_at(n, "Invalid character in parameter name: ',' in " + n.get("name"));
            break;
        }
        case TAG_PATH: check_attributes(n, { "value" }); if (depth != 1) fail_at(n, "<path>: path can only be child of root"); break;
        case TAG_INCLUDE: check_attributes(n, { "filename" }); break;
        case TAG_PROPERTY:
            if (n.tag == "spectrum") check_attributes(n, { "name", "value", "filename" }, false, true);
            else check_attributes(n, { "name", "value" });
            break;
        case TAG_VECTOR:
        case TAG_TRANSFORM_OP:
            if (n.tag == "lookat") { check_attributes(n, { "origin", "target", "up" }, false, true); break; }
            if (n.tag == "matrix") { check_attributes(n, { "value" }); break; }
            if (n.attr("value")) {   // expand_value_to_xyz (xml.cpp:290-309)
                if (n.attr("x") || n.attr("y") || n.attr("z")) fail_at(n, "can't mix and match \"value\" and \"x\"/\"y\"/\"z\" attributes");
                size_t count = 0; bool in_tok = false;
                for (char ch : n.get("value")) { const bool sep = ch == ',' || isspace((unsigned char) ch); if (!sep && !in_tok) ++count; in_tok = !sep; }
                if (count != 1 && count != 3) fail_at(n, "\"value\" attribute must have exactly 1 or 3 elements");
            }
            if (kind == TAG_VECTOR) check_attributes(n, { "name", "x", "y", "z", "value" }, false, true);
            else if (n.tag == "rotate") check_attributes(n, { "angle", "x", "y", "z", "value" }, false, true);
            else check_attributes(n, { "x", "y", "z", "value" }, false, true);
            break;
        case TAG_TRANSFORM: check_attributes(n, parent == TAG_ANIMATION ? std::vector<std::string> { "time" } : std::vector<std::string> { "name" }, false, true); break;
        case TAG_ANIMATION: check_attributes(n, { "name" }); break;
        default: break;
    }
    for (auto &c : n.children) check_tree(*c, kind, depth + 1, cc);
}

// ---------------------------------------------------------------------------- double 4x4 helpers
static Mat4d m_identity() { Mat4d r; for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0 : 0.0; return r; }
static Mat4d m_mul(const Mat4d &a, const Mat4d &b) {
    Mat4d r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += a.m[4 * i + k] * b.m[4 * k + j];
        r.m[4 * i + j] = s;
    }
    return r;
}
// Gauss-Jordan with partial pivoting (the reference keeps analytic inverses for translate/scale/rotate
// and inverts <matrix> values numerically in double; either way the float32 cast below agrees)
static Mat4d m_inverse(const Mat4d &a) {
    double w[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { w[i][j] = a.m[4 * i + j]; w[i][4 + j] = i == j; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
        if (w[piv][c] == 0.0) fail("singular transformation matrix");
        if (piv != c) for (int j = 0; j < 8; ++j) std::swap(w[piv][j], w[c][j]);
        double d = 1.0 / w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] *= d;
        for (int r = 0; r < 4; ++r) if (r != c) { double f = w[r][c]; if (f != 0.0) for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j]; }
    }
    Mat4d r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[4 * i + j] = w[i][4 + j];
    return r;
}
static void to_f32(const Mat4d &a, float *out) { for (int i = 0; i < 16; ++i) out[i] = (float) a.m[i]; }

static std::vector<std::string> tokenize(const std::string &v) {   // string::tokenize(value, ", ")
    std::vector<std::string> t; std::string cur;
    for (char c : v) { if (c == ',' || isspace((unsigned char) c)) { if (!cur.empty()) { t.push_back(cur); cur.clear(); } } else cur += c; }
    if (!cur.empty()) t.push_back(cur);
    return t;
}
static double parse_double(const std::string &v) {
    size_t pos = 0; double d;
    try { d = std::stod(v, &pos); } catch (...) { fail("could not parse floating point value \"" + v + "\"."); }
    while (pos < v.size() && isspace((unsigned char) v[pos])) ++pos;
    if (pos != v.size()) fail("could not parse floating point value \"" + v + "\".");
    return d;
}
static int64_t parse_int(const std::string &v) {
    size_t pos = 0; long long d;
    try { d = std::stoll(v, &pos); } catch (...) { fail("could not parse integer value \"" + v + "\"."); }
    while (pos < v.size() && isspace((unsigned char) v[pos])) ++pos;
    if (pos != v.size()) fail("could not parse integer value \"" + v + "\".");
    return d;
}
static void parse_xyz(const XNode &n, double def, double out[3]) {   // detail::expand_value_to_xyz + parse_vector
    if (auto *v = n.attr("value")) {
        auto t = tokenize(*v);
        if (t.size() == 1) { t.push_back(t[0]); t.push_back(t[0]); }
        if (t.size() != 3) fail("\"value\" attribute must have exactly 1 or 3 elements");
        for (int i = 0; i < 3; ++i) out[i] = parse_double(t[i]);
        return;
    }
    const char *k[3] = { "x", "y", "z" };
    for (int i = 0; i < 3; ++i) { auto *a = n.attr(k[i]); out[i] = a ? parse_double(*a) : def; }
}
static void parse_named3(const XNode &n, const char *attr, double out[3]) {
    auto t = tokenize(n.get(attr));
    if (t.size() != 3) fail(std::string("could not parse 3D vector attribute \"") + attr + "\"");
    for (int i = 0; i < 3; ++i) out[i] = parse_double(t[i]);
}

// A transform is the pair (matrix, inverse), like the reference's Transform (matrix + inverse_transpose,
// include/mitsuba/core/transform.h:43-70): translate / scale / rotate / lookat carry analytic inverses and
// composition multiplies both; only <matrix> is inverted numerically.
struct Xf { Mat4d m, inv; };
static Mat4d m_transpose(const Mat4d &a) { Mat4d r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[4 * i + j] = a.m[4 * j + i]; return r; }
static void normalize3(double v[3]) { double il = 1.0 / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] *= il; v[1] *= il; v[2] *= il; }
static void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

static Xf parse_transform(const XNode &node) {
    Xf cur { m_identity(), m_identity() };
    for (auto &opp : node.children) {
        const XNode &op = *opp; Mat4d t = m_identity(), ti = m_identity();
        if (op.tag == "matrix") {
            auto tok = tokenize(op.get("value"));
            if (tok.size() == 16) { for (int i = 0; i < 16; ++i) t.m[i] = parse_double(tok[i]); }
            else if (tok.size() == 9) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t.m[4 * i + j] = parse_double(tok[3 * i + j]); }
            else fail("matrix: expected 16 or 9 values");
            ti = m_inverse(t);
        } else if (op.tag == "translate") {
            double v[3]; parse_xyz(op, 0.0, v); t.m[3] = v[0]; t.m[7] = v[1]; t.m[11] = v[2];
            ti.m[3] = -v[0]; ti.m[7] = -v[1]; ti.m[11] = -v[2];
        } else if (op.tag == "scale") {
            double v[3]; parse_xyz(op, 1.0, v); t.m[0] = v[0]; t.m[5] = v[1]; t.m[10] = v[2];
            ti.m[0] = 1.0 / v[0]; ti.m[5] = 1.0 / v[1]; ti.m[10] = 1.0 / v[2];
        } else if (op.tag == "rotate") {
            double a[3]; parse_xyz(op, 0.0, a);
            if (!op.attr("angle")) fail("rotate: missing \"angle\" attribute");
            double th = parse_double(op.get("angle")) * (M_PI / 180.0), s = std::sin(th), c = std::cos(th), cm = 1.0 - c;
            t.m[0] = a[0] * a[0] * cm + c;        t.m[1] = a[0] * a[1] * cm - a[2] * s; t.m[2] = a[0] * a[2] * cm + a[1] * s;
            t.m[4] = a[0] * a[1] * cm + a[2] * s; t.m[5] = a[1] * a[1] * cm + c;        t.m[6] = a[1] * a[2] * cm - a[0] * s;
            t.m[8] = a[0] * a[2] * cm - a[1] * s; t.m[9] = a[1] * a[2] * cm + a[0] * s; t.m[10] = a[2] * a[2] * cm + c;
            ti = m_transpose(t);
        } else if (op.tag == "lookat") {
            double o[3], tg[3], up[3] = { 0, 0, 0 };
            parse_named3(op, "origin", o); parse_named3(op, "target", tg);
            if (op.attr("up")) parse_named3(op, "up", up);
            double d[3] = { tg[0] - o[0], tg[1] - o[1], tg[2] - o[2] };
            normalize3(d);
            if (up[0] * up[0] + up[1] * up[1] + up[2] * up[2] == 0) {   // coordinate_system(dir).first
                double sg = std::copysign(1.0, d[2]), a = -1.0 / (sg + d[2]), b = d[0] * d[1] * a;
                up[0] = (d[0] * d[0] * a) * sg + 1.0; up[1] = b * sg; up[2] = -d[0] * sg;
            }
            double l[3], nu[3];
            cross3(up, d, l); normalize3(l); cross3(d, l, nu);
            const double *rows[3] = { l, nu, d };
            for (int r = 0; r < 3; ++r) { t.m[4 * r] = l[r]; t.m[4 * r + 1] = nu[r]; t.m[4 * r + 2] = d[r]; t.m[4 * r + 3] = o[r]; }
            for (int r = 0; r < 3; ++r) {
                ti.m[4 * r] = rows[r][0]; ti.m[4 * r + 1] = rows[r][1]; ti.m[4 * r + 2] = rows[r][2];
                ti.m[4 * r + 3] = -(rows[r][0] * o[0] + rows[r][1] * o[1] + rows[r][2] * o[2]);
            }
            for (double x : t.m) if (std::isnan(x)) fail("invalid lookat transformation");
        } else {
            fail("transform nodes can only contain transform operations");
        }
        cur.m = m_mul(t, cur.m);       // ctx.transform = T(op) * ctx.transform
        cur.inv = m_mul(cur.inv, ti);
    }
    return cur;
}

// ---------------------------------------------------------------------------- PropBag
double PropBag::get_float(const std::string &n, double def) const {
    auto it = values.find(n); if (it == values.end()) return def;
    queried[n] = true;
    if (it->second.type == PropValue::Float) return it->second.f;
    if (it->second.type == PropValue::Int) return (double) it->second.i;
    fail("The property \"" + n + "\" has the wrong type (expected <float>).");
}
int64_t PropBag::get_int(const std::string &n, int64_t def) const {
    auto it = values.find(n); if (it == values.end()) return def;
    queried[n] = true;
    if (it->second.type != PropValue::Int) fail("The property \"" + n + "\" has the wrong type (expected <integer>).");
    return it->second.i;
}
bool PropBag::get_bool(const std::string &n, bool def) const {
    auto it = values.find(n); if (it == values.end()) return def;
    queried[n] = true;
    if (it->second.type != PropValue::Bool) fail("The property \"" + n + "\" has the wrong type (expected <boolean>).");
    return it->second.b;
}
std::string PropBag::get_string(const std::string &n, const std::string &def) const {
    auto it = values.find(n); if (it == values.end()) return def;
    queried[n] = true;
    if (it->second.type != PropValue::String) fail("The property \"" + n + "\" has the wrong type (expected <string>).");
    return it->second.s;
}
std::vector<std::string> PropBag::unqueried() const {
    std::vector<std::string> r;
    for (auto &kv : values) if (!queried.count(kv.first)) r.push_back(kv.first);
    return r;
}

PluginParams make_plugin_params(const PropBag &ip, const PropBag &sp) {
    PluginParams p;
    if (ip.plugin == "dopplertofpath") p.integrator = INTEGRATOR_DOPPLER;
    else if (ip.plugin == "path") p.integrator = INTEGRATOR_PATH;           // src/integrators/path.cpp
    else if (ip.plugin == "velocity") p.integrator = INTEGRATOR_VELOCITY;   // src/integrators/velocity.cpp
    else fail("unsupported integrator plugin \"" + ip.plugin + "\" (this library implements \"dopplertofpath\", \"path\" and \"velocity\")");
    if (sp.plugin == "correlated") p.sampler_kind = SAMPLER_CORRELATED;
    else if (sp.plugin == "independent") p.sampler_kind = SAMPLER_INDEPENDENT;
    else if (sp.plugin == "timestratified") p.sampler_kind = SAMPLER_TIMESTRATIFIED;
    else fail("unsupported sampler plugin \"" + sp.plugin + "\" (this library implements \"correlated\", \"independent\" and \"timestratified\")");
    p.time = (float) ip.get_float("time", 0.0015f);
    p.w_g_mhz = (float) ip.get_float("w_g", 30.0f);
    p.g_1 = (float) ip.get_float("g_1", 0.5f);
    p.g_0 = (float) ip.get_float("g_0", 0.5f);
    p.w_s_mhz = (float) ip.get_float("w_s", 30.0f);
    p.phase_offset = (float) ip.get_float("sensor_phase_offset", 0.0f);
    if (ip.has("hetero_offset"))        // float * 2 (float) * M_PI (double) -> float
        p.phase_offset = (float) ((double) ((float) ip.get_float("hetero_offset", 0.0) * 2) * M_PI);
    if (ip.has("hetero_frequency")) {
        p.hetero_frequency = (float) ip.get_float("hetero_frequency", 1.0);
        p.w_s_mhz = (float) ((double) p.w_g_mhz + (double) (p.hetero_frequency / p.time) * 1e-6);
    } else {
        p.hetero_frequency = (float) ((double) (p.w_s_mhz - p.w_g_mhz) * 1e6 * (double) p.time);
    }
    std::string wf = ip.get_string("wave_function_type", "sinusoidal");
    if (wf == "sinusoidal") p.wave_type = WAVE_SIN; else if (wf == "rectangular") p.wave_type = WAVE_RECT;
    else if (wf == "triangular") p.wave_type = WAVE_TRI; else if (wf == "trapezoidal") p.wave_type = WAVE_TRAP;
    else fail("unknown wave_function_type \"" + wf + "\"");   // the reference leaves the enum uninitialised here
    p.low_frequency_component_only = ip.get_bool("low_frequency_component_only", true);
    (void) ip.get_bool("is_doppler_integrator", false);
    std::string ts = ip.get_string("time_sampling_method", "antithetic");
    if (ts == "uniform") p.time_sampling = TIME_UNIFORM; else if (ts == "stratified") p.time_sampling = TIME_STRATIFIED;
    else if (ts == "antithetic") p.time_sampling = TIME_ANTITHETIC; else if (ts == "antithetic_mirror") p.time_sampling = TIME_ANTITHETIC_MIRROR;
    // `periodic` / `regular`: ETimeSampling declares them (sampler.h:27-34) and CorrelatedSampler::next_1d_time implements them (correlated.cpp:147-152), but the
    // reference's integrator never parses the two strings (integrator.cpp:58-70 leaves the enum uninitialised); here the names select the values they name
    else if (ts == "periodic") p.time_sampling = TIME_PERIODIC; else if (ts == "regular") p.time_sampling = TIME_REGULAR;
    else fail("unknown time_sampling_method \"" + ts + "\"");
    p.antithetic_shift = (float) ip.get_float("antithetic_shift", p.time_sampling == TIME_ANTITHETIC ? 0.5 : 0.0);
    p.stratify_each_interval = ip.get_bool("use_stratified_sampling_for_each_interval", true);
    p.path_correlation_depth = (uint32_t) ip.get_int("path_correlation_depth", 0);
    p.samples_per_pass = (uint32_t) ip.get_int("samples_per_pass", -1);   // SamplingIntegrator::m_samples_per_pass (integrator.cpp:54-56); -1 = one pass
    (void) ip.get_int("block_size", 0); (void) ip.get_float("timeout", -1.0);
    int64_t md = ip.get_int("max_depth", -1);
    if (md < 0 && md != -1) fail("\"max_depth\" must be set to -1 (infinite) or a value >= 0");
    p.max_depth = (uint32_t) md;
    int64_t rr = ip.get_int("rr_depth", 5);
    if (rr <= 0) fail("\"rr_depth\" must be set to a value greater than zero!");
    p.rr_depth = (uint32_t) rr;
    p.hide_emitters = ip.get_bool("hide_emitters", false);
    p.sample_count = (uint32_t) sp.get_int("sample_count", 4);
    p.base_seed = (uint32_t) sp.get_int("seed", 0);
    const bool corr = p.sampler_kind == SAMPLER_CORRELATED;
    p.time_correlate_number = corr ? (int32_t) sp.get_int("time_correlate_number", 2) : 2;
    p.path_correlate_number = corr ? (int32_t) sp.get_int("path_correlate_number", p.time_correlate_number) : 2;
    if (p.sampler_kind == SAMPLER_TIMESTRATIFIED) p.jitter = sp.get_bool("jitter", true);   // timestratified.cpp:73-74
    if (p.time_correlate_number <= 0 || p.path_correlate_number <= 0) fail("correlate numbers must be positive");
    for (const PropBag *b : { &ip, &sp }) {
        auto u = b->unqueried();
        if (!u.empty()) fail_unreferenced(u, b == &ip ? "integrator" : "sampler", b->plugin);   // xml.cpp:1204-1215
    }
    return p;
}

// ---------------------------------------------------------------------------- object tree
struct Obj {
    std::string tag, plugin, id, name;       // name: the `name` attribute of the element (the property an object is assigned to)
    std::vector<std::string> ref_names;     // `name` attribute of <ref> children, by position in `children` ("" elsewhere)
    PropBag props;
    std::map<std::string, std::vector<double>> colors;            // <rgb>/<spectrum>
    std::map<std::string, std::vector<double>> vectors;           // <point>/<vector>
    std::map<std::string, Xf> transforms;
    std::map<std::string, std::vector<std::pair<float, Xf>>> animations;
    std::vector<std::pair<std::string, std::shared_ptr<Obj>>> children;   // document order; refs resolved later
    std::vector<std::pair<size_t, std::string>> refs;                     // (position in children, id)
};
static bool is_object_tag(const std::string &t) {
    static const char *tags[] = { "scene", "integrator", "sensor", "sampler", "film", "rfilter", "bsdf", "shape", "emitter", "texture" };
    for (auto *x : tags) if (t == x) return true;
    return false;
}

struct LoadCtx {
    std::map<std::string, std::shared_ptr<Obj>> registry;
    std::vector<std::pair<std::string, std::string>> defaults;   // (name, value)
    std::set<std::string> used;                                  // parameters some attribute referred to ("Unused parameter", xml.cpp:1067-1070)
};

// FileResolver (src/core/fresolver.cpp): the directories a relative file name is looked up in, first match wins; holds the scene file's
// directory (src/mitsuba/mitsuba.cpp, python load_file) and whatever <path value=".."/> prepends (xml.cpp:651-668)
static thread_local std::vector<std::string> g_search_paths;
static bool file_exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
static std::string resolve_path(const std::string &fn) {
    if (fn.empty() || fn[0] == '/') return fn;
    for (auto &d : g_search_paths) if (file_exists(d + "/" + fn)) return d + "/" + fn;
    return g_search_paths.empty() ? fn : g_search_paths.back() + "/" + fn;   // not found: the name under the scene's directory, for the error message
}
static std::string slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("could not open \"" + path + "\"");
    std::stringstream ss; ss << f.rdbuf(); return ss.str();
}
constexpr int kMaxIncludeDepth = 15;   // MI_XML_INCLUDE_MAX_RECURSION (xml.cpp:40)

// $parameter substitution in document order (xml.cpp:441-456,630-648), with the two tags that change what later nodes see:
// <path> (resolver directories, xml.cpp:651-668) and <include> (xml.cpp:670-725: the children of an included <scene>, or the included object itself,
// take the place of the tag; parameters defined so far carry over, <default>s of the included file stay defined afterwards)
static void expand_children(XNode &n, LoadCtx &ctx, int depth, int include_depth, const std::string &src_dir);
static void substitute(XNode &n, LoadCtx &ctx, int depth = 0, int include_depth = 0, const std::string &src_dir = "") {
    if (!ctx.defaults.empty()) {
        auto sorted = ctx.defaults;
        std::stable_sort(sorted.begin(), sorted.end(), [](auto &a, auto &b) { return a.first.size() > b.first.size(); });
        for (auto &a : n.attrs) {
            if (a.second.find('$') == std::string::npos) continue;
            for (auto &d : sorted) {
                std::string key = "$" + d.first; size_t pos = 0;
                while ((pos = a.second.find(key, pos)) != std::string::npos) { a.second.replace(pos, key.size(), d.second); pos += d.second.size(); ctx.used.insert(d.first); }
            }
            if (a.second.find('$') != std::string::npos) fail("undefined parameter(s) in string: \"" + a.second + "\"!");
        }
    } else {
        for (auto &a : n.attrs) if (a.second.find('$') != std::string::npos) fail("undefined parameter(s) in string: \"" + a.second + "\"!");
    }
    if (n.tag == "default") {
        std::string name = n.get("name"), value = n.get("value");
        if (name.empty()) fail("<default>: name must by nonempty");
        bool found = false; for (auto &d : ctx.defaults) if (d.first == name) found = true;
        if (!found) { ctx.defaults.emplace_back(name, value); ctx.used.insert(name); }   // a <default> of the file itself counts as used (xml.cpp:646)
    }
    if (n.tag == "path") {
        if (depth != 1) fail("<path>: path can only be child of root");
        std::string p = n.get("value");
        if (!p.empty() && p[0] != '/') {
            const std::string local = src_dir.empty() ? p : src_dir + "/" + p;
            p = file_exists(local) ? local : resolve_path(p);
        }
        if (!file_exists(p)) fail("<path>: folder \"" + p + "\" not found");
        g_search_paths.insert(g_search_paths.begin(), p);
    }
    expand_children(n, ctx, depth, include_depth, src_dir);
}
static void expand_children(XNode &n, LoadCtx &ctx, int depth, int include_depth, const std::string &src_dir) {
    for (size_t i = 0; i < n.children.size(); ++i) {
        XNode &c = *n.children[i];
        substitute(c, ctx, depth + 1, include_depth, src_dir);
        if (c.tag != "include") continue;
        for (auto &a : c.attrs) if (a.first != "filename") fail("unexpected attribute \"" + a.first + "\" in element \"include\"");
        if (!c.attr("filename")) fail("missing attribute \"filename\" in element \"include\"");
        const std::string file = resolve_path(c.get("filename"));
        if (!file_exists(file)) fail("included file \"" + file + "\" not found");
        if (include_depth + 1 > kMaxIncludeDepth) fail("Exceeded <include> recursion limit of " + std::to_string(kMaxIncludeDepth));
        const std::string text = slurp(file);
        XParser xp(text);
        std::unique_ptr<XNode> root;
        try { root = xp.document(); } catch (const std::exception &e) { fail("error while loading \"" + file + "\": " + e.what()); }
        const size_t slash = file.find_last_of('/');
        const std::string dir = slash == std::string::npos ? std::string(".") : file.substr(0, slash);
        XNode holder;   // the nodes that take the tag's place: the children of an included <scene> (parsed at depth 1), or the included object (depth 0)
        int holder_depth = 0;
        if (root->tag == "scene") holder.children = std::move(root->children);
        else { holder.children.push_back(std::move(root)); holder_depth = -1; }
        expand_children(holder, ctx, holder_depth, include_depth + 1, dir);
        const size_t count = holder.children.size();
        n.children.erase(n.children.begin() + (long) i);
        n.children.insert(n.children.begin() + (long) i, std::make_move_iterator(holder.children.begin()), std::make_move_iterator(holder.children.end()));
        i += count; --i;   // (size_t wrap-around at count == 0, i == 0 is undone by the loop's ++i)
    }
}

static std::shared_ptr<Obj> parse_object(const XNode &n, LoadCtx &ctx) {
    auto o = std::make_shared<Obj>();
    o->tag = n.tag; o->plugin = n.get("type"); o->id = n.get("id"); o->name = n.get("name"); o->props.plugin = o->plugin;
    for (auto &cp : n.children) {
        const XNode &c = *cp; std::string name = c.get("name");
        if (!name.empty() && name[0] == '_') fail("invalid parameter name \"" + name + "\": leading underscores are reserved");
        PropValue v;
        if (c.tag == "default" || c.tag == "path") continue;
        else if (c.tag == "alias") {   // xml.cpp:608-628: a second id for an object declared earlier
            for (auto &a : c.attrs) if (a.first != "id" && a.first != "as") fail("unexpected attribute \"" + a.first + "\" in element \"alias\"");
            const std::string src = c.get("id"), dst = c.get("as");
            if (ctx.registry.count(dst)) fail("\"alias\" has duplicate id \"" + dst + "\"");
            auto it = ctx.registry.find(src);
            if (it == ctx.registry.end()) fail("referenced id \"" + src + "\" not found");
            ctx.registry[dst] = it->second;
        }
        else if (is_object_tag(c.tag)) { o->children.emplace_back(c.tag, parse_object(c, ctx)); o->ref_names.emplace_back(); }
        else if (c.tag == "ref") {
            if (!c.attr("id")) fail("<ref>: missing \"id\" attribute");
            o->refs.emplace_back(o->children.size(), c.get("id")); o->children.emplace_back("ref", nullptr); o->ref_names.push_back(name);
        }
        else if (c.tag == "float") { v.type = PropValue::Float; v.f = parse_double(c.get("value")); o->props.values[name] = v; }
        else if (c.tag == "integer") { v.type = PropValue::Int; v.i = parse_int(c.get("value")); o->props.values[name] = v; }
        else if (c.tag == "boolean") {
            std::string b = c.get("value"); std::transform(b.begin(), b.end(), b.begin(), ::tolower);
            if (b != "true" && b != "false") fail("could not parse boolean value \"" + b + "\" -- must be \"true\" or \"false\".");
            v.type = PropValue::Bool; v.b = b == "true"; o->props.values[name] = v;
        }
        else if (c.tag == "string") { v.type = PropValue::String; v.s = c.get("value"); o->props.values[name] = v; }
        else if (c.tag == "point" || c.tag == "vector") { double x[3]; parse_xyz(c, 0.0, x); o->vectors[name] = { x[0], x[1], x[2] }; }
        else if (c.tag == "rgb") {
            auto t = tokenize(c.get("value"));
            if (t.size() == 1) { t.push_back(t[0]); t.push_back(t[0]); }
            if (t.size() != 3) fail("'rgb' tag requires one or three values (got \"" + c.get("value") + "\")");
            o->colors[name] = { parse_double(t[0]), parse_double(t[1]), parse_double(t[2]) };
        }
        else if (c.tag == "spectrum") {
            auto t = tokenize(c.get("value"));
            if (t.size() != 1) fail("only constant <spectrum> values are supported");
            double d = parse_double(t[0]); o->colors[name] = { d, d, d };
        }
        else if (c.tag == "transform") { o->transforms[name] = parse_transform(c); }
        else if (c.tag == "animation") {
            std::vector<std::pair<float, Xf>> keys;
            for (auto &tr : c.children) {
                if (tr->tag != "transform" || !tr->attr("time")) fail("<animation> may only contain <transform time=...> nodes");
                float time = (float) parse_double(tr->get("time"));
                if (!keys.empty() && time <= keys.back().first)
                    fail("AnimatedTransform::append(): time values must be strictly monotonically increasing!");
                keys.emplace_back(time, parse_transform(*tr));
            }
            o->animations[name] = keys;
        }
        else fail("unexpected tag \"" + c.tag + "\"");
    }
    if (!o->id.empty()) {
        if (ctx.registry.count(o->id)) fail("\"" + o->tag + "\" has duplicate id \"" + o->id + "\"");
        ctx.registry[o->id] = o;
    }
    return o;
}
static void resolve_refs(Obj &o, LoadCtx &ctx) {
    for (auto &r : o.refs) {
        auto it = ctx.registry.find(r.second);
        if (it == ctx.registry.end()) fail("reference to unknown object \"" + r.second + "\"!");
        o.children[r.first] = { it->second->tag, it->second };
    }
    o.refs.clear();
    for (auto &c : o.children) if (c.second) resolve_refs(*c.second, ctx);
}

// ---------------------------------------------------------------------------- assembly
// include/mitsuba/render/ior.h:16-44 (physical constants) + lookup_ior :71-77
static float lookup_ior(const Obj &b, const char *name, const char *def) {
    auto it = b.props.values.find(name);
    if (it != b.props.values.end() && (it->second.type == PropValue::Float || it->second.type == PropValue::Int)) return (float) b.props.get_float(name, 0.0);
    static const std::pair<const char *, float> table[] = { { "vacuum", 1.0f }, { "helium", 1.000036f }, { "hydrogen", 1.000132f }, { "air", 1.000277f },
        { "carbon dioxide", 1.00045f }, { "water", 1.3330f }, { "acetone", 1.36f }, { "ethanol", 1.361f }, { "carbon tetrachloride", 1.461f },
        { "glycerol", 1.4729f }, { "benzene", 1.501f }, { "silicone oil", 1.52045f }, { "bromine", 1.661f }, { "water ice", 1.31f },
        { "fused quartz", 1.458f }, { "pyrex", 1.470f }, { "acrylic glass", 1.49f }, { "polypropylene", 1.49f }, { "bk7", 1.5046f },
        { "sodium chloride", 1.544f }, { "amber", 1.55f }, { "pet", 1.5750f }, { "diamond", 2.419f } };
    std::string key = b.props.get_string(name, def);
    std::transform(key.begin(), key.end(), key.begin(), ::tolower);
    for (auto &e : table) if (key == e.first) return e.second;
    fail("Unable to find an IOR value for \"" + key + "\"!");
}
// <rgb> / <spectrum> children become texture OBJECTS of the plugin's Properties (xml.cpp:792-875): one the plugin does not ask for is reported as
// an unreferenced object when the plugin has been instantiated (xml.cpp:1204-1213)
static void check_colors(const Obj &o, std::initializer_list<const char *> known) {
    for (auto &c : o.colors) {
        bool ok = false;
        for (const char *k : known) ok |= c.first == k;
        if (!ok) fail("unreferenced object \"" + c.first + "\" (within " + o.tag + " of type \"" + o.plugin + "\")");
    }
}
static void color_of(const Obj &b, const char *name, float def, float out[3]) {
    auto c = b.colors.find(name);
    if (c != b.colors.end()) { for (int i = 0; i < 3; ++i) out[i] = (float) c->second[i]; }
    else { float r = (float) b.props.get_float(name, def); out[0] = out[1] = out[2] = r; }
}
// quad::gauss_legendre (include/mitsuba/core/quad.h:27-86) with math::legendre_pd (math.h:92-119): nodes and weights on [-1, 1],
// Newton iteration in double, stored as float
static void legendre_pd(int l, double x, double &lv, double &dv) {
    if (l == 0) { lv = 1; dv = 0; return; }
    if (l == 1) { lv = x; dv = 1; return; }
    double l_p_pred = 1, l_pred = x, d_p_pred = 0, d_pred = 1, k0 = 3, k1 = 2, k2 = 1; lv = 0; dv = 0;
    for (int ki = 2; ki <= l; ++ki) {
        lv = (k0 * x * l_pred - k2 * l_p_pred) / k1;
        dv = d_p_pred + k0 * l_pred;
        l_p_pred = l_pred; l_pred = lv; d_p_pred = d_pred; d_pred = dv;
        k2 = k1; k0 += 2; k1 += 1;
    }
}
static void gauss_legendre(int n, std::vector<float> &nodes, std::vector<float> &weights) {
    nodes.assign(n, 0.f); weights.assign(n, 0.f);
    n--;
    if (n == 0) { nodes[0] = 0.f; weights[0] = 2.f; }
    else if (n == 1) { nodes[0] = (float) -std::sqrt(1.0 / 3.0); nodes[1] = -nodes[0]; weights[0] = weights[1] = 1.f; }
    const int m = (n + 1) / 2;
    for (int i = 0; i < m; ++i) {
        double x = -std::cos((double) (2 * i + 1) / (double) (2 * n + 2) * 3.14159265358979323846), lv, dv;
        for (int it = 1; ; ++it) {
            if (it > 20) fail("gauss_legendre(" + std::to_string(n) + "): did not converge after 20 iterations!");
            legendre_pd(n + 1, x, lv, dv);
            const double step = lv / dv;
            x -= step;
            if (std::fabs(step) <= 4 * std::fabs(x) * (2.220446049250313e-16 / 2)) break;     // dr::Epsilon<double> = 2^-53
        }
        legendre_pd(n + 1, x, lv, dv);
        weights[i] = weights[n - i] = (float) (2 / ((1 - x * x) * (dv * dv)));
        nodes[i] = (float) x; nodes[n - i] = (float) -x;
    }
    if ((n % 2) == 0) {
        double lv, dv; legendre_pd(n + 1, 0.0, lv, dv);
        weights[n / 2] = (float) (2.0 / (dv * dv)); nodes[n / 2] = 0.f;
    }
}
// eval_transmittance / eval_reflectance (include/mitsuba/render/microfacet.h:464-566) of a microfacet distribution with visible-normal
// sampling for ONE incident direction: tensor Gauss-Legendre rule over the sample square (32 x 32 nodes for eta > 1, else 128 x 128;
// dr::meshgrid order: x runs fastest), accumulated in float in node order.
static float rough_integral(Ggx g, V3 wi, float eta, bool transmit) {
    const int res = eta > 1.f ? 32 : 128;
    static thread_local std::vector<float> nodes, weights; static thread_local int have = 0;
    if (have != res) { gauss_legendre(res, nodes, weights); have = res; }
    float result = 0.f;
    for (int j = 0; j < res * res; ++j) {
        const float nx = fmaf(nodes[j % res], 0.5f, 0.5f), ny = fmaf(nodes[j / res], 0.5f, 0.5f), w = weights[j % res] * weights[j / res];
        float pdf, f, cos_theta_t, eta_it, eta_ti;
        const V3 m = ggx_sample(g, wi, nx, ny, pdf);
        const float dwm = dot(wi, m);
        fresnel_dielectric(dwm, eta, f, cos_theta_t, eta_it, eta_ti);
        float smith;
        if (transmit) {
            const float k = fmaf(dwm, eta_ti, cos_theta_t);                                     // refract(wi, m, cos_theta_t, eta_ti), fresnel.h:311-314
            const V3 wo = mk(fmaf(m.x, k, -(wi.x * eta_ti)), fmaf(m.y, k, -(wi.y * eta_ti)), fmaf(m.z, k, -(wi.z * eta_ti)));
            smith = ggx_smith_g1(g, wo, m) * (1.f - f);
            if (wo.z * wi.z >= 0.f) smith = 0.f;
        } else {
            const V3 wo = mk(fmaf(m.x, 2.f * dwm, -wi.x), fmaf(m.y, 2.f * dwm, -wi.y), fmaf(m.z, 2.f * dwm, -wi.z));   // reflect(wi, m)
            smith = ggx_smith_g1(g, wo, m) * f;
            if (wo.z <= 0.f || wi.z <= 0.f) smith = 0.f;
        }
        result += smith * w * 0.25f;
    }
    return result;
}
// RoughPlastic::parameters_changed (src/bsdfs/roughplastic.cpp:222-257): m_external_transmittance on MI_ROUGH_TRANSMITTANCE_RES = 64
// cosines mu = max(1e-6, linspace(0, 1, 64)) and m_internal_reflectance = mean(eval_reflectance(1 / eta) * mu) * 2
void rough_plastic_tables(int type, float alpha, float eta, float *table, float *internal_reflectance) {
    const Ggx g = mf_make(type, alpha, alpha);
    float sum = 0.f;
    for (int i = 0; i < 64; ++i) {
        const float mu = fmax_(1e-6f, fmaf((float) i, 1.f / 63.f, 0.f));
        const V3 wi = mk(sqrtf(1.f - mu * mu), 0.f, mu);
        table[i] = rough_integral(g, wi, eta, true);
        sum += rough_integral(g, wi, 1.f / eta, false) * wi.z;
    }
    *internal_reflectance = sum * (1.f / 64.f) * 2.f;
}

// ---- textures on the diffuse reflectances (src/textures/checkerboard.cpp:55-62, src/textures/bitmap.cpp:113-262, RGB variants)
static thread_local std::vector<HostTexture> *g_textures = nullptr;   // the scene being assembled
static thread_local std::set<const void *> *g_attached_emitters = nullptr;   // area emitters already attached to a shape (an emitter declared at scene level can be referenced by ONE shape, endpoint.cpp:36-40)
static thread_local std::map<const void *, int> *g_texture_index = nullptr;   // texture object -> its index in *g_textures: a texture referenced by many BSDFs / shapes is decoded and stored once
static thread_local std::string g_base_dir;
static float srgb_to_linear_u8(uint32_t v) {   // StructConverter::linearize + dr::srgb_to_linear (src/core/struct.cpp:1600-1625)
    const double x = (double) v / 255.0;
    return (float) (x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4));
}
static HostTexture texture_of(const Obj &t) {
    HostTexture tex;
    auto uv = t.transforms.find("to_uv");
    if (uv != t.transforms.end()) {   // Transform4f::extract(): the upper-left 2x2 block (transform.h:340-360; the translation column is not copied)
        const Mat4d &m = uv->second.m;
        tex.to_uv[0] = (float) m.m[0]; tex.to_uv[1] = (float) m.m[1]; tex.to_uv[2] = (float) m.m[4]; tex.to_uv[3] = (float) m.m[5];
    }
    if (t.plugin == "checkerboard") {
        for (auto &c : t.children) if (c.first == "texture" || c.first == "ref") fail("checkerboard: nested textures are not supported (constant colours only)");
        tex.kind = TEX_CHECKERBOARD;
        color_of(t, "color0", .4f, tex.color0); color_of(t, "color1", .2f, tex.color1);
        const float third = 1.0f / 3.0f;
        const float m0 = ((tex.color0[0] + tex.color0[1]) + tex.color0[2]) * third, m1 = ((tex.color1[0] + tex.color1[1]) + tex.color1[2]) * third;
        tex.mean = .5f * (m0 + m1);
    } else if (t.plugin == "bitmap") {
        const std::string fn = t.props.get_string("filename", "");
        if (fn.empty()) fail("Property \"filename\" has not been specified!");
        const std::string path = resolve_path(fn);
        const std::string ft = t.props.get_string("filter_type", "bilinear"), wm = t.props.get_string("wrap_mode", "repeat");
        if (ft != "nearest" && ft != "bilinear") fail("Invalid filter type \"" + ft + "\", must be one of: \"nearest\", or \"bilinear\"!");
        if (wm != "repeat" && wm != "mirror" && wm != "clamp") fail("Invalid wrap mode \"" + wm + "\", must be one of: \"repeat\", \"mirror\", or \"clamp\"!");
        const bool raw = t.props.get_bool("raw", false);
        (void) t.props.get_bool("accel", true);
        std::vector<uint8_t> px; uint32_t w, h, ch;
        {   // by signature, as Bitmap::detect_file_format does (bitmap.cpp:700-734): JPEG (FF D8) or PNG
            FILE *probe = fopen(path.c_str(), "rb"); unsigned char sig[2] = { 0, 0 };
            if (probe) { if (fread(sig, 1, 2, probe) != 2) sig[0] = 0; fclose(probe); }
            if (sig[0] == 0xff && sig[1] == 0xd8) read_jpeg(path, px, w, h, ch); else read_png(path, px, w, h, ch);
        }
        if (w < 2 || h < 2) fail("bitmap: the image must be at least 2x2 pixels in size");
        tex.kind = TEX_BITMAP; tex.filter = ft == "bilinear"; memset(tex.color0, 0, 12); memset(tex.color1, 0, 12); tex.wrap = wm == "repeat" ? 0 : wm == "mirror" ? 1 : 2;
        tex.width = w; tex.height = h; tex.channels = ch;
        float lut[256];
        for (uint32_t i = 0; i < 256; ++i) lut[i] = raw ? (float) i * (1.0f / 255.0f) : srgb_to_linear_u8(i);
        tex.data.resize(px.size());
        for (size_t i = 0; i < px.size(); ++i) tex.data[i] = lut[px[i]];
        double sum = 0.0;   // m_mean: luminance (3 channels) or the value, accumulated in double (bitmap.cpp:221-262)
        const size_t n = (size_t) w * h;
        if (ch == 3) for (size_t i = 0; i < n; ++i) sum += (double) (tex.data[3 * i] * 0.212671f + tex.data[3 * i + 1] * 0.715160f + tex.data[3 * i + 2] * 0.072169f);
        else for (size_t i = 0; i < n; ++i) sum += (double) tex.data[i];
        tex.mean = (float) (sum / (double) n);
    } else fail("unsupported texture plugin \"" + t.plugin + "\" (supported: bitmap, checkerboard)");
    auto u = t.props.unqueried();
    if (!u.empty()) fail_unreferenced(u, "texture", t.plugin);
    return tex;
}
// a BSDF's reflectance-like property: a colour (-> out, returns -1) or a texture child of that name (-> out = its mean, returns its index)
// index of a texture object in the scene's texture table (decoded and stored on first use)
static int texture_index_of(const Obj *c) {
    if (!g_textures || !g_texture_index) fail("internal error: no texture table");
    auto known = g_texture_index->find((const void *) c);
    if (known == g_texture_index->end()) {
        g_textures->push_back(texture_of(*c));
        known = g_texture_index->emplace((const void *) c, (int) g_textures->size() - 1).first;
    }
    return known->second;
}
static int reflectance_of(const Obj &b, const char *name, float def, float out[3]) {
    for (size_t i = 0; i < b.children.size(); ++i) {
        const Obj *c = b.children[i].second.get();
        if (!c) continue;
        const std::string &cname = i < b.ref_names.size() && !b.ref_names[i].empty() ? b.ref_names[i] : c->name;
        if (cname != name) continue;
        if (c->tag != "texture") fail(std::string("property \"") + name + "\" must be a colour or a texture");
        const int index = texture_index_of(c);
        out[0] = out[1] = out[2] = (*g_textures)[(size_t) index].mean;
        return index;
    }
    color_of(b, name, def, out);
    return -1;
}

// the roughness of roughconductor / roughdielectric (roughconductor.cpp:189-199, roughdielectric.cpp:213-223): `alpha`, or `alpha_u` and `alpha_v`, each a
// float or a texture (Texture::eval_1 per hit; the constant then holds the texture's mean)
static void roughness_of(const Obj &b, HostShape &s) {
    auto slot = [&](const char *name, float &value) -> int {
        float c[3]; const int t = reflectance_of(b, name, 0.1f, c);
        if (t >= 0) { value = c[0]; return t; }
        value = (float) b.props.get_float(name, 0.1);
        return -1;
    };
    auto given = [&](const char *name) {
        if (b.props.has(name)) return true;
        for (size_t i = 0; i < b.children.size(); ++i) {
            const Obj *c = b.children[i].second.get();
            if (!c) continue;
            if ((i < b.ref_names.size() && !b.ref_names[i].empty() ? b.ref_names[i] : c->name) == name) return true;
        }
        return false;
    };
    if (given("alpha_u") || given("alpha_v")) {
        if (!given("alpha_u") || !given("alpha_v")) fail("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.");
        if (given("alpha")) fail("Microfacet model: please specifyeither 'alpha' or 'alpha_u'/'alpha_v'.");
        s.tex_alpha_u = slot("alpha_u", s.alpha_u); s.tex_alpha_v = slot("alpha_v", s.alpha_v);
    } else { s.tex_alpha_u = slot("alpha", s.alpha_u); s.tex_alpha_v = s.tex_alpha_u; s.alpha_v = s.alpha_u; }
}

// diffuse (src/bsdfs/diffuse.cpp), conductor (conductor.cpp:171-188), dielectric (dielectric.cpp:176-203), twosided{...} (twosided.cpp:40-70)
static void bsdf_of(const Obj &b, HostShape &s) {
    if (b.plugin == "mask") {   // src/bsdfs/mask.cpp:93-117: one nested BSDF seen through an opacity (float or texture, default 0.5)
        const Obj *inner = nullptr;
        for (auto &c : b.children) if (c.first == "bsdf") { if (inner) fail("Cannot specify more than one child BSDF"); inner = c.second.get(); }
        if (!inner) fail("Child BSDF not specified");
        if (inner->plugin == "mask") fail("mask: a mask nested in a mask is not supported");
        bsdf_of(*inner, s);
        const bool inner_twosided = s.twosided;
        if (b.colors.count("opacity")) fail("mask: an rgb \"opacity\" is not supported (give a float or a texture)");
        float c[3]; const int t = reflectance_of(b, "opacity", 0.5f, c);
        s.masked = true; s.tex_opacity = t;
        s.opacity = t >= 0 ? c[0] : (float) b.props.get_float("opacity", 0.5);
        s.twosided = inner_twosided;
        auto u = b.props.unqueried();
        if (!u.empty()) fail_unreferenced(u, "bsdf", b.plugin);
        for (size_t i = 0; i < b.children.size(); ++i) {
            const Obj *c2 = b.children[i].second.get();
            if (!c2 || c2->tag != "texture") continue;
            const std::string &cname = i < b.ref_names.size() && !b.ref_names[i].empty() ? b.ref_names[i] : c2->name;
            if (cname != "opacity") fail("unreferenced object \"" + cname + "\" in plugin of type \"mask\"");
        }
        return;
    }
    if (b.plugin == "blendbsdf") {   // src/bsdfs/blendbsdf.cpp:80-104: two nested BSDFs and a weight (float or texture, no default)
        std::vector<const Obj *> inner;
        for (auto &c : b.children) if (c.first == "bsdf") { if (inner.size() == 2) fail("BlendBSDF: Cannot specify more than two child BSDFs"); inner.push_back(c.second.get()); }
        if (b.colors.count("weight")) fail("blendbsdf: an rgb \"weight\" is not supported (give a float or a texture)");
        float c[3]; const int t = reflectance_of(b, "weight", 0.5f, c);
        if (t < 0 && !b.props.has("weight")) fail("Property \"weight\" has not been specified!");
        if (inner.size() != 2) fail("BlendBSDF: Two child BSDFs must be specified!");
        for (const Obj *in : inner) if (in->plugin == "mask" || in->plugin == "blendbsdf") fail("blendbsdf: a \"" + in->plugin + "\" nested in a blendbsdf is not supported in this build");
        bsdf_of(*inner[0], s);
        auto other = std::make_shared<HostShape>();
        bsdf_of(*inner[1], *other);
        if (s.masked || other->masked || s.blend_other || other->blend_other) fail("blendbsdf: a mask or blendbsdf nested in a blendbsdf is not supported in this build");
        s.blend_other = other; s.tex_blend = t;
        s.blend_weight = t >= 0 ? c[0] : (float) b.props.get_float("weight", 0.5);
        auto u = b.props.unqueried();
        if (!u.empty()) fail_unreferenced(u, "bsdf", b.plugin);
        for (size_t i = 0; i < b.children.size(); ++i) {
            const Obj *c2 = b.children[i].second.get();
            if (!c2 || c2->tag != "texture") continue;
            const std::string &cname = i < b.ref_names.size() && !b.ref_names[i].empty() ? b.ref_names[i] : c2->name;
            if (cname != "weight") fail("unreferenced object \"" + cname + "\" in plugin of type \"blendbsdf\"");
        }
        return;
    }
    if (b.plugin == "bumpmap") {   // src/bsdfs/bumpmap.cpp:84-112: one nested BSDF in the frame the gradient of ONE height texture (any property name) gives
        const Obj *inner = nullptr, *tex = nullptr;
        for (auto &c : b.children) {
            if (c.first == "bsdf") { if (inner) fail("Only a single BSDF child object can be specified."); inner = c.second.get(); }
            else if (c.first == "texture") { if (tex) fail("Only a single Texture child object can be specified."); tex = c.second.get(); }
        }
        if (!inner) fail("Exactly one BSDF child object must be specified.");
        if (!tex) fail("Exactly one Texture child object must be specified.");
        if (inner->plugin == "twosided" || inner->plugin == "mask" || inner->plugin == "normalmap" || inner->plugin == "bumpmap" || inner->plugin == "blendbsdf")
            fail("bumpmap: a \"" + inner->plugin + "\" nested in a bumpmap is not supported in this build (nest the bumpmap inside it instead)");
        if (tex->plugin != "bitmap") fail("bumpmap: the height texture must be a bitmap (\"" + tex->plugin + "\" has no eval_1_grad)");
        bsdf_of(*inner, s);
        if (s.tex_normal >= 0) fail("bumpmap: internal error");
        s.tex_normal = texture_index_of(tex);
        s.bumpmap = true; s.bump_scale = (float) b.props.get_float("scale", 1.0);
        auto u = b.props.unqueried();
        if (!u.empty()) fail_unreferenced(u, "bsdf", b.plugin);
        return;
    }
    if (b.plugin == "normalmap") {   // src/bsdfs/normalmap.cpp:84-108: one nested BSDF evaluated in the frame an RGB texture gives
        const Obj *inner = nullptr;
        for (auto &c : b.children) if (c.first == "bsdf") { if (inner) fail("Only a single BSDF child object can be specified."); inner = c.second.get(); }
        if (!inner) fail("Exactly one BSDF child object must be specified.");
        if (inner->plugin == "twosided" || inner->plugin == "mask" || inner->plugin == "normalmap" || inner->plugin == "bumpmap" || inner->plugin == "blendbsdf")
            fail("normalmap: a \"" + inner->plugin + "\" nested in a normalmap is not supported in this build (nest the normalmap inside it instead)");
        bsdf_of(*inner, s);
        float c[3]; s.tex_normal = reflectance_of(b, "normalmap", 0.f, c);
        if (s.tex_normal < 0) fail("Property \"normalmap\" has not been specified!");
        if ((*g_textures)[(size_t) s.tex_normal].channels != 3 && (*g_textures)[(size_t) s.tex_normal].kind == TEX_BITMAP) fail("normalmap: the texture must have three channels");
        auto u = b.props.unqueried();
        if (!u.empty()) fail_unreferenced(u, "bsdf", b.plugin);
        for (size_t i = 0; i < b.children.size(); ++i) {
            const Obj *c2 = b.children[i].second.get();
            if (!c2 || c2->tag != "texture") continue;
            const std::string &cname = i < b.ref_names.size() && !b.ref_names[i].empty() ? b.ref_names[i] : c2->name;
            if (cname != "normalmap") fail("unreferenced object \"" + cname + "\" in plugin of type \"normalmap\"");
        }
        return;
    }
    if (b.plugin == "twosided") {
        const Obj *inner = nullptr, *back = nullptr; int n = 0;
        for (auto &c : b.children) if (c.first == "bsdf") { (n == 0 ? inner : back) = c.second.get(); ++n; }
        if (n > 2) fail("At most two nested BSDFs can be specified!");
        if (n == 0) fail("A nested one-sided material is required!");
        bsdf_of(*inner, s);
        if (n == 2) {   // twosided.cpp:75-86: the second BSDF is the back side's
            auto other = std::make_shared<HostShape>();
            bsdf_of(*back, *other);
            auto transmits2 = [](const HostShape &h) { return h.bsdf == BSDF_DIELECTRIC || h.bsdf == BSDF_THINDIELECTRIC || h.bsdf == BSDF_ROUGHDIELECTRIC || h.bsdf == BSDF_NULL; };
            if (s.blend_other || other->blend_other || s.masked || other->masked) fail("twosided: a blendbsdf or mask as one of two nested BSDFs is not supported in this build");
            if (transmits2(s) || transmits2(*other)) fail("Only materials without a transmission component can be nested!");
            s.twosided = other->twosided = true;
            s.blend_other = other; s.two_bsdfs = true;
            return;
        }
        auto transmits = [](const HostShape &h) { return h.bsdf == BSDF_DIELECTRIC || h.bsdf == BSDF_THINDIELECTRIC || h.bsdf == BSDF_ROUGHDIELECTRIC || h.bsdf == BSDF_NULL; };   // BSDFFlags::Transmission includes Null
        if (transmits(s) || s.masked || (s.blend_other && transmits(*s.blend_other))) fail("Only materials without a transmission component can be nested!");
        s.twosided = true;
        if (s.blend_other) s.blend_other->twosided = true;   // twosided{ blendbsdf{ a, b } } flips wi / wo before either nested BSDF sees them: the same as blendbsdf{ twosided{a}, twosided{b} }
        return;
    }
    s.twosided = false;
    if (b.plugin == "diffuse") { s.bsdf = BSDF_DIFFUSE; s.tex_refl = reflectance_of(b, "reflectance", 0.5f, s.refl); }
    else if (b.plugin == "conductor") {
        std::string material = b.props.get_string("material", "none");
        if (material != "none") fail(b.props.has("eta") || b.colors.count("eta") ? "Should specify either (eta, k) or material, not both."
            : "conductor: named materials need the spectral IOR data files, which this build does not ship; give \"eta\" and \"k\"");
        s.bsdf = BSDF_CONDUCTOR;
        color_of(b, "eta", 0.f, s.cond_eta); color_of(b, "k", 1.f, s.cond_k); s.tex_spec = reflectance_of(b, "specular_reflectance", 1.f, s.spec_refl);
    } else if (b.plugin == "dielectric") {
        const float int_ior = lookup_ior(b, "int_ior", "bk7"), ext_ior = lookup_ior(b, "ext_ior", "air");
        if (int_ior < 0 || ext_ior < 0) fail("The interior and exterior indices of refraction must be positive!");
        s.bsdf = BSDF_DIELECTRIC; s.diel_eta = int_ior / ext_ior;
        s.tex_spec = reflectance_of(b, "specular_reflectance", 1.f, s.spec_refl); s.tex_trans = reflectance_of(b, "specular_transmittance", 1.f, s.spec_trans);
    } else if (b.plugin == "null") {   // src/bsdfs/null.cpp:36-40: no parameters; one component, BSDFFlags::Null | FrontSide | BackSide
        s.bsdf = BSDF_NULL;
    } else if (b.plugin == "thindielectric") {   // src/bsdfs/thindielectric.cpp:137-158
        const float int_ior = lookup_ior(b, "int_ior", "bk7"), ext_ior = lookup_ior(b, "ext_ior", "air");
        if (int_ior < 0 || ext_ior < 0) fail("The interior and exterior indices of refraction must be positive!");
        s.bsdf = BSDF_THINDIELECTRIC; s.diel_eta = int_ior / ext_ior;
        s.tex_spec = reflectance_of(b, "specular_reflectance", 1.f, s.spec_refl); s.tex_trans = reflectance_of(b, "specular_transmittance", 1.f, s.spec_trans);
    } else if (b.plugin == "roughdielectric") {   // src/bsdfs/roughdielectric.cpp:163-238
        const float int_ior = lookup_ior(b, "int_ior", "bk7"), ext_ior = lookup_ior(b, "ext_ior", "air");
        if (int_ior < 0 || ext_ior < 0 || int_ior == ext_ior) fail("The interior and exterior indices of refraction must be positive and differ!");
        s.bsdf = BSDF_ROUGHDIELECTRIC; s.diel_eta = int_ior / ext_ior;
        s.tex_spec = reflectance_of(b, "specular_reflectance", 1.f, s.spec_refl); s.tex_trans = reflectance_of(b, "specular_transmittance", 1.f, s.spec_trans);
        std::string distr = b.props.get_string("distribution", "beckmann");
        std::transform(distr.begin(), distr.end(), distr.begin(), ::tolower);
        if (distr != "beckmann" && distr != "ggx") fail("Specified an invalid distribution \"" + distr + "\", must be \"beckmann\" or \"ggx\"!");
        s.beckmann = distr != "ggx";   // MicrofacetType (microfacet.h:30-36)
        s.sample_all = !b.props.get_bool("sample_visible", true);
        roughness_of(b, s);
    } else if (b.plugin == "roughconductor") {   // src/bsdfs/roughconductor.cpp:177-227
        std::string material = b.props.get_string("material", "none");
        if (material != "none") fail(b.props.has("eta") || b.colors.count("eta") ? "Should specify either (eta, k) or material, not both."
            : "roughconductor: named materials need the spectral IOR data files, which this build does not ship; give \"eta\" and \"k\"");
        std::string distr = b.props.get_string("distribution", "beckmann");
        std::transform(distr.begin(), distr.end(), distr.begin(), ::tolower);
        if (distr != "beckmann" && distr != "ggx") fail("Specified an invalid distribution \"" + distr + "\", must be \"beckmann\" or \"ggx\"!");
        s.beckmann = distr != "ggx";   // MicrofacetType (microfacet.h:30-36)
        s.sample_all = !b.props.get_bool("sample_visible", true);
        roughness_of(b, s);
        s.bsdf = BSDF_ROUGHCONDUCTOR;
        color_of(b, "eta", 0.f, s.cond_eta); color_of(b, "k", 1.f, s.cond_k); s.tex_spec = reflectance_of(b, "specular_reflectance", 1.f, s.spec_refl);
    } else if (b.plugin == "plastic") {   // src/bsdfs/plastic.cpp:167-217
        const float int_ior = lookup_ior(b, "int_ior", "polypropylene"), ext_ior = lookup_ior(b, "ext_ior", "air");
        if (int_ior < 0 || ext_ior < 0) fail("The interior and exterior indices of refraction must be positive!");
        s.bsdf = BSDF_PLASTIC; s.diel_eta = int_ior / ext_ior;
        s.tex_refl = reflectance_of(b, "diffuse_reflectance", 0.5f, s.refl); s.tex_spec = reflectance_of(b, "specular_reflectance", 1.f, s.spec_refl);
        s.nonlinear = b.props.get_bool("nonlinear", false);
        const float eta = s.diel_eta;
        s.inv_eta_2 = 1.f / (eta * eta);
        {   // fresnel_diffuse_reflectance(1 / eta), include/mitsuba/render/fresnel.h:328-355
            const float e = 1.f / eta, inv_e = 1.0f / e;
            const float approx_1 = fmaf(0.0636f, inv_e, fmaf(e, fmaf(e, -1.4399f, 0.7099f), 0.6681f));
            float h = -1.36881f;
            h = fmaf(h, inv_e, 4.98554f); h = fmaf(h, inv_e, -7.80989f); h = fmaf(h, inv_e, 6.75335f); h = fmaf(h, inv_e, -3.4793f); h = fmaf(h, inv_e, 0.919317f);
            s.fdr_int = e < 1.f ? approx_1 : h;
        }
        // d_mean = m_diffuse_reflectance->mean(): the mean of a colour's three channels, or the texture's own mean
        const float d_mean = s.tex_refl >= 0 ? s.refl[0] : ((s.refl[0] + s.refl[1]) + s.refl[2]) * (1.0f / 3.0f), s_mean = s.tex_spec >= 0 ? s.spec_refl[0] : ((s.spec_refl[0] + s.spec_refl[1]) + s.spec_refl[2]) * (1.0f / 3.0f);
        s.spec_sampling_weight = s_mean / (d_mean + s_mean);
    } else if (b.plugin == "roughplastic") {   // src/bsdfs/roughplastic.cpp:170-257
        const float int_ior = lookup_ior(b, "int_ior", "polypropylene"), ext_ior = lookup_ior(b, "ext_ior", "air");
        if (int_ior < 0 || ext_ior < 0 || int_ior == ext_ior) fail("The interior and exterior indices of refraction must be positive and differ!");
        s.bsdf = BSDF_ROUGHPLASTIC; s.diel_eta = int_ior / ext_ior;
        s.tex_refl = reflectance_of(b, "diffuse_reflectance", 0.5f, s.refl); s.tex_spec = reflectance_of(b, "specular_reflectance", 1.f, s.spec_refl);
        const bool has_spec = b.props.has("specular_reflectance") || b.colors.count("specular_reflectance") || s.tex_spec >= 0;
        s.nonlinear = b.props.get_bool("nonlinear", false);
        std::string distr = b.props.get_string("distribution", "beckmann");
        std::transform(distr.begin(), distr.end(), distr.begin(), ::tolower);
        if (distr != "beckmann" && distr != "ggx") fail("Specified an invalid distribution \"" + distr + "\", must be \"beckmann\" or \"ggx\"!");
        s.beckmann = distr != "ggx";   // MicrofacetType (microfacet.h:30-36)
        s.sample_all = !b.props.get_bool("sample_visible", true);
        if (b.props.has("alpha_u") || b.props.has("alpha_v")) {
            if (!b.props.has("alpha_u") || !b.props.has("alpha_v")) fail("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.");
            if (b.props.has("alpha")) fail("Microfacet model: please specifyeither 'alpha' or 'alpha_u'/'alpha_v'.");
            s.alpha_u = (float) b.props.get_float("alpha_u", 0.1); s.alpha_v = (float) b.props.get_float("alpha_v", 0.1);
            if (s.alpha_u != s.alpha_v) fail("The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!");
        } else s.alpha_u = s.alpha_v = (float) b.props.get_float("alpha", 0.1);
        s.inv_eta_2 = 1.f / (s.diel_eta * s.diel_eta);
        const float d_mean = s.tex_refl >= 0 ? s.refl[0] : ((s.refl[0] + s.refl[1]) + s.refl[2]) * (1.0f / 3.0f),
                    s_mean = has_spec ? (s.tex_spec >= 0 ? s.spec_refl[0] : ((s.spec_refl[0] + s.spec_refl[1]) + s.spec_refl[2]) * (1.0f / 3.0f)) : 1.f;
        s.spec_sampling_weight = s_mean / (d_mean + s_mean);
        s.rough_table.resize(64);
        rough_plastic_tables(s.beckmann ? MF_BECKMANN : MF_GGX, s.alpha_u, s.diel_eta, s.rough_table.data(), &s.fdr_int);       // fdr_int carries m_internal_reflectance
    } else fail("unsupported BSDF plugin \"" + b.plugin + "\" (supported: diffuse, plastic, roughplastic, conductor, roughconductor, dielectric, thindielectric, roughdielectric, null, twosided, mask, blendbsdf, normalmap, bumpmap)");
    auto u = b.props.unqueried();
    if (!u.empty()) fail_unreferenced(u, "bsdf", b.plugin);
    check_colors(b, { "reflectance", "diffuse_reflectance", "specular_reflectance", "specular_transmittance", "eta", "k" });
    // texture children: the slot that takes one was read above; a texture bound to any other property (or to a misspelt name) must not be dropped
    // silently -- the reference either uses it or raises "unreferenced object" (xml.cpp:1204-1215)
    auto takes_texture = [&](const std::string &name) {
        if (b.plugin == "diffuse") return name == "reflectance";
        if (name == "specular_reflectance") return true;                                                  // every other BSDF of this library has one
        if (name == "diffuse_reflectance") return b.plugin == "plastic" || b.plugin == "roughplastic";
        if (name == "specular_transmittance") return b.plugin == "dielectric" || b.plugin == "thindielectric" || b.plugin == "roughdielectric";
        if (name == "alpha" || name == "alpha_u" || name == "alpha_v") return b.plugin == "roughconductor" || b.plugin == "roughdielectric";
        return false;
    };
    for (size_t i = 0; i < b.children.size(); ++i) {
        const Obj *c = b.children[i].second.get();
        if (!c || c->tag != "texture") continue;
        const std::string &cname = i < b.ref_names.size() && !b.ref_names[i].empty() ? b.ref_names[i] : c->name;
        if (takes_texture(cname)) continue;
        static const char *known[] = { "reflectance", "diffuse_reflectance", "specular_reflectance", "specular_transmittance", "alpha", "alpha_u", "alpha_v", "eta", "k" };
        bool is_known = false;
        for (const char *k : known) is_known |= cname == k;
        fail(is_known ? "property \"" + cname + "\" of plugin \"" + b.plugin + "\" does not accept a texture in this build (constant values only)"
                      : "unreferenced object \"" + cname + "\" in plugin of type \"" + b.plugin + "\"");
    }
}

static void bake_cube(HostShape &s) {   // src/shapes/cube.cpp:114-160
    static const float vtx[24][3] = {
        { 1,-1,-1},{ 1,-1, 1},{-1,-1, 1},{-1,-1,-1},{ 1, 1,-1},{-1, 1,-1},{-1, 1, 1},{ 1, 1, 1},
        { 1,-1,-1},{ 1, 1,-1},{ 1, 1, 1},{ 1,-1, 1},{ 1,-1, 1},{ 1, 1, 1},{-1, 1, 1},{-1,-1, 1},
        {-1,-1, 1},{-1, 1, 1},{-1, 1,-1},{-1,-1,-1},{ 1, 1,-1},{ 1,-1,-1},{-1,-1,-1},{-1, 1,-1} };
    static const float nr[6][3] = { {0,-1,0},{0,1,0},{1,0,0},{0,0,1},{-1,0,0},{0,0,-1} };
    static const float tc[4][2] = { {0,1},{1,1},{1,0},{0,0} };
    static const uint32_t tri[36] = { 0,1,2, 3,0,2, 4,5,6, 7,4,6, 8,9,10, 11,8,10, 12,13,14, 15,12,14, 16,17,18, 19,16,18, 20,21,22, 23,20,22 };
    s.positions.resize(72); s.normals.resize(72); s.texcoords.resize(48); s.faces.assign(tri, tri + 36);
    for (int i = 0; i < 24; ++i) {
        V3 p = xf_point(s.to_world, mk(vtx[i][0], vtx[i][1], vtx[i][2]));
        V3 n = xf_normal(s.to_object, mk(nr[i / 4][0], nr[i / 4][1], nr[i / 4][2]));
        n = n * (1.0f / sqrtf(dot(n, n)));     // scalar-mode dr::normalize
        s.positions[3 * i] = p.x; s.positions[3 * i + 1] = p.y; s.positions[3 * i + 2] = p.z;
        s.normals[3 * i] = n.x; s.normals[3 * i + 1] = n.y; s.normals[3 * i + 2] = n.z;
        s.texcoords[2 * i] = tc[i % 4][0]; s.texcoords[2 * i + 1] = tc[i % 4][1];
    }
}

// Sphere ctor + update (src/shapes/sphere.cpp:121-160), in float32 like ScalarTransform4f: composed = to_world * translate(center)
// * scale(radius) (4x4 products accumulate with fmadd over k, Dr.Jit's Matrix operator*), the inverse from the factors' analytic
// inverses in reverse order; m_radius = |composed * (1,0,0)|, m_center = composed * (0,0,0); a mirroring transform toggles flip_normals.
static void m4_mul_f32(const float *a, const float *b, float *out) {
    float r[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float sum = a[4 * i] * b[j];
        for (int k = 1; k < 4; ++k) sum = fmaf(a[4 * i + k], b[4 * k + j], sum);
        r[4 * i + j] = sum;
    }
    memcpy(out, r, sizeof r);
}
static void bake_sphere(HostShape &s, const Obj &o) {
    double c[3] = { 0, 0, 0 };
    auto pv = o.vectors.find("center");
    if (pv != o.vectors.end()) for (int i = 0; i < 3; ++i) c[i] = pv->second[i];
    const float center[3] = { (float) c[0], (float) c[1], (float) c[2] }, radius = (float) o.props.get_float("radius", 1.0);
    float T[16] = { 1, 0, 0, center[0], 0, 1, 0, center[1], 0, 0, 1, center[2], 0, 0, 0, 1 };
    float Ti[16] = { 1, 0, 0, -center[0], 0, 1, 0, -center[1], 0, 0, 1, -center[2], 0, 0, 0, 1 };
    const float ir = 1.0f / radius;
    float S[16] = { radius, 0, 0, 0, 0, radius, 0, 0, 0, 0, radius, 0, 0, 0, 0, 1 }, Si[16] = { ir, 0, 0, 0, 0, ir, 0, 0, 0, 0, ir, 0, 0, 0, 0, 1 };
    float tmp[16], comp[16], comp_inv[16];
    m4_mul_f32(s.to_world, T, tmp); m4_mul_f32(tmp, S, comp);
    m4_mul_f32(Ti, s.to_object, tmp); m4_mul_f32(Si, tmp, comp_inv);
    memcpy(s.to_world, comp, sizeof comp); memcpy(s.to_object, comp_inv, sizeof comp_inv);
    s.radius = norm(mk(comp[0], comp[4], comp[8]));
    s.center[0] = comp[3]; s.center[1] = comp[7]; s.center[2] = comp[11];
    const float *m = comp;
    float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
    if (det < 0.f) s.flip_normals = !s.flip_normals;
    s.sphere_inv_area = 1.0f / ((4.f * kPi) * sqr(s.radius));
}

// Cylinder ctor + update (src/shapes/cylinder.cpp:100-147), in float32 like ScalarTransform4f: composed = to_world * translate(p0) *
// to_frame(Frame3f((p1 - p0) / |p1 - p0|)) * scale(radius, radius, |p1 - p0|); the unit cylinder x^2 + y^2 = 1, 0 <= z <= 1 lives in object space
static void bake_cylinder(HostShape &s, const Obj &o) {
    double a[3] = { 0, 0, 0 }, b[3] = { 0, 0, 1 };
    auto pa = o.vectors.find("p0"), pb = o.vectors.find("p1");
    if (pa != o.vectors.end()) for (int i = 0; i < 3; ++i) a[i] = pa->second[i];
    if (pb != o.vectors.end()) for (int i = 0; i < 3; ++i) b[i] = pb->second[i];
    const float p0[3] = { (float) a[0], (float) a[1], (float) a[2] }, p1[3] = { (float) b[0], (float) b[1], (float) b[2] }, radius = (float) o.props.get_float("radius", 1.0);
    const V3 d = mk(p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]);
    const float length = norm(d);
    const V3 n = d * rcp(length); V3 fs, ft;
    coordinate_system(n, fs, ft);
    float T[16] = { 1, 0, 0, p0[0], 0, 1, 0, p0[1], 0, 0, 1, p0[2], 0, 0, 0, 1 }, Ti[16] = { 1, 0, 0, -p0[0], 0, 1, 0, -p0[1], 0, 0, 1, -p0[2], 0, 0, 0, 1 };
    float F[16] = { fs.x, ft.x, n.x, 0, fs.y, ft.y, n.y, 0, fs.z, ft.z, n.z, 0, 0, 0, 0, 1 };       // columns s, t, n (transform.h:286-296)
    float Fi[16] = { fs.x, fs.y, fs.z, 0, ft.x, ft.y, ft.z, 0, n.x, n.y, n.z, 0, 0, 0, 0, 1 };
    const float ir = 1.0f / radius, il = 1.0f / length;
    float S[16] = { radius, 0, 0, 0, 0, radius, 0, 0, 0, 0, length, 0, 0, 0, 0, 1 }, Si[16] = { ir, 0, 0, 0, 0, ir, 0, 0, 0, 0, il, 0, 0, 0, 0, 1 };
    float t1[16], t2[16], comp[16], comp_inv[16];
    m4_mul_f32(s.to_world, T, t1); m4_mul_f32(t1, F, t2); m4_mul_f32(t2, S, comp);
    m4_mul_f32(Ti, s.to_object, t1); m4_mul_f32(Fi, t1, t2); m4_mul_f32(Si, t2, comp_inv);
    memcpy(s.to_world, comp, sizeof comp); memcpy(s.to_object, comp_inv, sizeof comp_inv);
    s.radius = norm(mk(comp[0], comp[4], comp[8]));
    const float *m = comp;
    float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
    if (det < 0.f) s.flip_normals = !s.flip_normals;
}

static HostShape make_shape(const Obj &o, bool strip_to_world, const std::string &base_dir) {
    HostShape s; s.id = o.id;
    const bool mesh_file = o.plugin == "obj" || o.plugin == "ply" || o.plugin == "serialized";
    if (o.plugin == "rectangle") s.kind = SHAPE_RECT; else if (o.plugin == "cube" || mesh_file) s.kind = SHAPE_MESH;
    else if (o.plugin == "sphere") s.kind = SHAPE_SPHERE;
    else if (o.plugin == "disk") s.kind = SHAPE_DISK;
    else if (o.plugin == "cylinder") s.kind = SHAPE_CYLINDER;
    else fail("unsupported shape plugin \"" + o.plugin + "\" (supported: rectangle, disk, cube, obj, ply, serialized, sphere, cylinder, shapegroup, instance)");
    Xf tw { m_identity(), m_identity() };
    if (!strip_to_world) { auto t = o.transforms.find("to_world"); if (t != o.transforms.end()) tw = t->second; }
    s.flip_normals = o.props.get_bool("flip_normals", false);
    s.face_normals = o.props.get_bool("face_normals", false);
    if ((s.kind == SHAPE_RECT || s.kind == SHAPE_DISK) && s.flip_normals) {   // rectangle.cpp:91-99, disk.cpp:91-95
        Mat4d f = m_identity(); f.m[10] = -1.0; Mat4d fi = m_identity(); fi.m[10] = 1.0 / -1.0;
        tw.m = m_mul(tw.m, f); tw.inv = m_mul(fi, tw.inv); s.flip_normals = false;
    }
    to_f32(tw.m, s.to_world); to_f32(tw.inv, s.to_object);
    if (s.kind == SHAPE_SPHERE) bake_sphere(s, o);
    if (s.kind == SHAPE_CYLINDER) bake_cylinder(s, o);
    const Obj *bsdf = nullptr;
    for (auto &c : o.children) {
        if (c.first == "bsdf") { if (bsdf) fail("Only a single BSDF child object can be specified per shape."); bsdf = c.second.get(); }
        else if (c.first == "emitter") {   // src/emitters/area.cpp:64-76; supported on static rectangles
            const Obj &e = *c.second;
            if (s.emitter) fail("Only a single Emitter child object can be specified per shape.");
            if (e.plugin != "area") fail("unsupported emitter plugin \"" + e.plugin + "\" inside a shape (supported: area)");
            if (g_attached_emitters && !g_attached_emitters->insert((const void *) &e).second) fail("An endpoint can be only be attached to a single shape.");   // endpoint.cpp:36-40
            if (strip_to_world) fail("Instancing of emitters is not supported");   // shapegroup.cpp:27-28: an animated (or grouped) shape becomes an instance (xml.cpp:1165-1195), which cannot carry an emitter in the reference either
            if (e.transforms.count("to_world")) fail("Found a 'to_world' transformation -- this is not allowed. The area light inherits this transformation from its parent shape.");
            bool textured = false;
            for (size_t i = 0; i < e.children.size(); ++i) {
                const Obj *c2 = e.children[i].second.get();
                if (!c2 || c2->tag != "texture") continue;
                const std::string &cname = i < e.ref_names.size() && !e.ref_names[i].empty() ? e.ref_names[i] : c2->name;
                if (cname != "radiance") fail("unreferenced object \"" + cname + "\" in plugin of type \"area\"");
                textured = true;
            }
            if (textured) {   // area.cpp:73: a texture makes the emitter spatially varying: it is then sampled through the texture (area.cpp:129-153)
                if (s.kind != SHAPE_RECT) fail("area emitter: a textured radiance is supported on rectangles only");
                s.tex_radiance = reflectance_of(e, "radiance", 1.f, s.radiance);
            } else {
                auto rc = e.colors.find("radiance");
                if (rc != e.colors.end()) for (int i = 0; i < 3; ++i) s.radiance[i] = (float) rc->second[i];
                else { float v = (float) e.props.get_float("radiance", 1.0); s.radiance[0] = s.radiance[1] = s.radiance[2] = v; }
            }
            s.emitter = true;
        }
        else fail("unsupported child <" + c.first + "> in shape");
    }
    if (bsdf) bsdf_of(*bsdf, s);   // else default diffuse: 0.5, or 0 for an emitter (src/render/shape.cpp:66-72)
    // (an emitter on a shape whose BSDF has a null lobe -- thindielectric, null, mask -- is fine: the kernels carry the integrators' valid_ray flag since round 4)
    if (s.emitter && s.kind == SHAPE_CYLINDER) fail("cylinder: area emitters on cylinders are not supported");
    if (s.emitter && !bsdf) s.refl[0] = s.refl[1] = s.refl[2] = 0.f;   // only the DEFAULT BSDF of an emitter is black: a given one keeps its reflectance
    RawMesh raw;
    if (mesh_file) {   // src/shapes/obj.cpp:139-143, ply.cpp:160-166: `filename` through the file resolver
        if (!o.props.has("filename")) fail("Property \"filename\" has not been specified!");
        std::string fn = o.props.get_string("filename", "");
        std::string path = resolve_path(fn);
        raw = o.plugin == "obj" ? load_obj(path, o.props.get_bool("flip_tex_coords", true), s.face_normals)
            : o.plugin == "ply" ? load_ply(path, s.face_normals) : load_serialized(path, (int) o.props.get_int("shape_index", 0), s.face_normals);
    }
    auto u = o.props.unqueried();
    if (!u.empty()) fail_unreferenced(u, "shape", o.plugin);
    check_colors(o, {});
    if (mesh_file) bake_mesh(s, raw);
    else if (s.kind == SHAPE_MESH) bake_cube(s);
    return s;
}

static HostObject make_instance(const Obj &o, uint32_t group) {
    HostObject ob; ob.kind = OBJ_INSTANCE; ob.index = group;
    memset(ob.key, 0, sizeof ob.key);
    auto a = o.animations.find("to_world");
    if (a != o.animations.end()) {
        // AnimatedTransform::eval only interpolates keyframes 0 and 1 (include/mitsuba/core/transform.h:458-466)
        ob.n_keys = (uint32_t) std::min<size_t>(a->second.size(), 2);
        if (ob.n_keys == 0) { ob.n_keys = 1; to_f32(m_identity(), ob.key[0]); }
        for (uint32_t i = 0; i < ob.n_keys && i < a->second.size(); ++i) { ob.key_time[i] = a->second[i].first; to_f32(a->second[i].second.m, ob.key[i]); }
    } else {
        auto t = o.transforms.find("to_world");
        ob.n_keys = 1; to_f32(t != o.transforms.end() ? t->second.m : m_identity(), ob.key[0]);
    }
    return ob;
}

static double parse_fov(const Obj &s, double aspect) {   // src/render/sensor.cpp:149-203
    bool has_fov = s.props.has("fov"), has_fl = s.props.has("focal_length");
    if (has_fov && has_fl) fail("Please specify either a focal length ('focal_length') or a field of view ('fov')!");
    double fov; std::string axis;
    if (has_fov) {
        fov = s.props.get_float("fov", 0);
        axis = s.props.get_string("fov_axis", "x");
        std::transform(axis.begin(), axis.end(), axis.begin(), ::tolower);
        if (axis == "smaller") axis = aspect > 1 ? "y" : "x"; else if (axis == "larger") axis = aspect > 1 ? "x" : "y";
    } else {
        std::string f = s.props.get_string("focal_length", "50mm");
        if (f.size() > 2 && f.substr(f.size() - 2) == "mm") f = f.substr(0, f.size() - 2);
        double value = parse_double(f);
        fov = 2.0 * (std::atan(std::sqrt(double(36 * 36 + 24 * 24)) / (2.0 * value)) * (180.0 / M_PI));
        axis = "diagonal";
    }
    double r;
    if (axis == "x") r = fov;
    else if (axis == "y") r = (2.0 * std::atan(std::tan(0.5 * (fov * (M_PI / 180.0))) * aspect)) * (180.0 / M_PI);
    else if (axis == "diagonal") {
        double diagonal = 2.0 * std::tan(0.5 * (fov * (M_PI / 180.0)));
        double width = diagonal / std::sqrt(1.0 + 1.0 / (aspect * aspect));
        r = (2.0 * std::atan(width * 0.5)) * (180.0 / M_PI);
    } else fail("The 'fov_axis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!");
    if (r <= 0.0 || r >= 180.0) fail("The horizontal field of view must be in the range [0, 180]!");
    return r;
}

static void make_sensor(const Obj &o, HostScene &sc) {
    if (o.plugin != "perspective" && o.plugin != "thinlens" && o.plugin != "orthographic")
        fail("unsupported sensor plugin \"" + o.plugin + "\" (supported: perspective, thinlens, orthographic)");
    HostSensor &se = sc.sensor;
    se.orthographic = o.plugin == "orthographic";
    const Obj *film = nullptr, *sampler = nullptr;
    for (auto &c : o.children) {
        if (c.first == "film") { if (film) fail("Only one film can be specified per sensor."); film = c.second.get(); }
        else if (c.first == "sampler") { if (sampler) fail("Only one sampler can be specified per sensor."); sampler = c.second.get(); }
        else fail("unsupported child <" + c.first + "> in sensor");
    }
    bool have_filter = false;
    if (film) {
        if (film->plugin != "hdrfilm") fail("unsupported film plugin \"" + film->plugin + "\" (supported: hdrfilm)");
        se.film_w = (int32_t) film->props.get_int("width", 768); se.film_h = (int32_t) film->props.get_int("height", 576);
        se.crop_w = (int32_t) film->props.get_int("crop_width", se.film_w); se.crop_h = (int32_t) film->props.get_int("crop_height", se.film_h);
        se.crop_x = (int32_t) film->props.get_int("crop_offset_x", 0); se.crop_y = (int32_t) film->props.get_int("crop_offset_y", 0);
        std::string pf = film->props.get_string("pixel_format", "rgb");
        for (auto &ch : pf) ch = (char) std::tolower((unsigned char) ch);   // string::to_lower (hdrfilm.cpp:143-144)
        // hdrfilm.cpp:160-192: rgba sets FilmFlags::Alpha -- the block gains an alpha channel fed by the integrator's valid_ray (integrator.cpp:528-533)
        if (pf == "rgba") se.alpha = true;
        else if (pf == "luminance" || pf == "luminance_alpha" || pf == "xyz" || pf == "xyza" || pf == "transient")
            fail("unsupported pixel_format \"" + pf + "\" (supported: rgb, rgba)");
        else if (pf != "rgb") fail("The \"pixel_format\" parameter must either be equal to \"luminance\", \"luminance_alpha\", \"rgb\", \"rgba\",  \"xyz\", \"xyza\". Found " + pf + ".");
        (void) film->props.get_string("file_format", "openexr"); (void) film->props.get_string("component_format", "float16");
        if (film->props.get_bool("sample_border", false)) fail("sample_border=true is not supported");
        (void) film->props.get_bool("compensate", false);
        for (auto &c : film->children) {
            if (c.first != "rfilter") fail("unsupported child <" + c.first + "> in film");
            const Obj &rf = *c.second;
            if (rf.plugin == "tent") { se.filter = FILTER_TENT; se.filter_radius = (float) rf.props.get_float("radius", 1.0); }
            else if (rf.plugin == "box") { se.filter = FILTER_BOX; se.filter_radius = .5f; }
            else if (rf.plugin == "gaussian") {   // src/rfilters/gaussian.cpp:48-53: cut off after 4 standard deviations
                se.filter = FILTER_GAUSSIAN; se.filter_stddev = (float) rf.props.get_float("stddev", .5f); se.filter_radius = 4 * se.filter_stddev;
            }
            else if (rf.plugin == "mitchell") {   // src/rfilters/mitchell.cpp:38-45
                se.filter = FILTER_MITCHELL; se.filter_radius = 2.f;
                se.filter_b = (float) rf.props.get_float("B", 1.f / 3.f); se.filter_c = (float) rf.props.get_float("C", 1.f / 3.f);
            }
            else if (rf.plugin == "catmullrom") { se.filter = FILTER_CATMULLROM; se.filter_radius = 2.f; }   // src/rfilters/catmullrom.cpp:33-36
            else if (rf.plugin == "lanczos") { se.filter = FILTER_LANCZOS; se.filter_radius = (float) rf.props.get_int("lobes", 3); }   // src/rfilters/lanczos.cpp:47-50
            else fail("unsupported rfilter plugin \"" + rf.plugin + "\" (supported: tent, box, gaussian, mitchell, catmullrom, lanczos)");
            have_filter = true;
        }
        auto u = film->props.unqueried();
        if (!u.empty()) fail_unreferenced(u, "film", film->plugin);
        if (se.film_w <= 0 || se.film_h <= 0 || se.crop_w <= 0 || se.crop_h <= 0 || se.crop_x < 0 || se.crop_y < 0 ||
            se.crop_x + se.crop_w > se.film_w || se.crop_y + se.crop_h > se.film_h) fail("invalid film size / crop window");
    }
    if (!have_filter) { se.filter = FILTER_GAUSSIAN; se.filter_stddev = .5f; se.filter_radius = 2.f; }   // film.cpp:49-53: default gaussian
    auto t = o.transforms.find("to_world");
    to_f32(t != o.transforms.end() ? t->second.m : m_identity(), se.to_world);
    // perspective.cpp:143-144 / thinlens.cpp:149-150: Transform::has_scale (transform.h:325-337)
    for (int i = 0; i < 3 && !se.orthographic; ++i) for (int j = i; j < 3; ++j) {   // the orthographic camera takes its extent from the scale of to_world
        float sum = 0.f;
        for (int k = 0; k < 3; ++k) sum += se.to_world[4 * i + k] * se.to_world[4 * j + k];
        if (std::fabs(sum - (i == j ? 1.f : 0.f)) > 1e-3f) fail("Scale factors in the camera-to-world transformation are not allowed!");
    }
    se.shutter_open = (float) o.props.get_float("shutter_open", 0.0);
    se.shutter_close = (float) o.props.get_float("shutter_close", 0.0);
    if (se.shutter_close - se.shutter_open < 0) fail("Shutter opening time must be less than or equal to the shutter closing time!");
    se.near_clip = (float) o.props.get_float("near_clip", 1e-2f);
    se.far_clip = (float) o.props.get_float("far_clip", 1e4f);
    if (se.near_clip <= 0.f) fail("The 'near_clip' parameter must be greater than zero!");
    if (se.near_clip >= se.far_clip) fail("The 'near_clip' parameter must be smaller than 'far_clip'.");
    se.x_fov = se.orthographic ? 0.f : (float) parse_fov(o, se.film_w / (double) se.film_h);
    se.focus_distance = (float) o.props.get_float("focus_distance", se.far_clip);   // ProjectiveCamera (sensor.cpp:134): read by both cameras
    if (o.plugin == "thinlens") {   // thinlens.cpp:138-156
        if (!o.props.has("aperture_radius")) fail("Property \"aperture_radius\" has not been specified!");
        se.thinlens = true;
        se.aperture_radius = (float) o.props.get_float("aperture_radius", 0.0);
        if (se.aperture_radius == 0.f) se.aperture_radius = 5.9604644775390625e-8f;   // dr::Epsilon<float>
    }
    (void) o.props.get_float("principal_point_offset_x", 0.0); (void) o.props.get_float("principal_point_offset_y", 0.0);
    if (sampler) sc.sampler = sampler->props; else { sc.sampler = PropBag(); sc.sampler.plugin = "independent"; }
    auto u = o.props.unqueried();
    if (!u.empty()) fail_unreferenced(u, "sensor", o.plugin);
    check_colors(o, {});
}

HostScene load_scene_xml(const std::string &text, const std::map<std::string, std::string> &params, const std::string &base_dir) {
    XParser xp(text);
    auto root = xp.document();
    g_xml_text = &text; g_xml_id = "<string>";
    struct TextScope { ~TextScope() { g_xml_text = nullptr; } } text_scope;
    { CheckCtx cc; check_tree(*root, TAG_INVALID, 0, cc); }
    LoadCtx ctx;
    for (auto &kv : params) ctx.defaults.emplace_back(kv.first, kv.second);
    g_search_paths.clear();
    if (!base_dir.empty()) g_search_paths.push_back(base_dir);
    substitute(*root, ctx, 0, 0, base_dir);
    for (auto &kv : params) if (!ctx.used.count(kv.first)) fail("Unused parameter \"" + kv.first + "\"!");
    const bool scene_root = root->tag == "scene";
    std::shared_ptr<Obj> top;
    if (scene_root) top = parse_object(*root, ctx);
    else {   // any object may be the root of a description (xml.cpp:489-490); it is instantiated like a scene's child, but only scenes can be rendered
        top = std::make_shared<Obj>(); top->tag = "scene"; top->plugin = "scene"; top->props.plugin = "scene";
        XNode holder; holder.tag = "scene";
        top->children.emplace_back(root->tag, parse_object(*root, ctx)); top->ref_names.emplace_back();
    }
    resolve_refs(*top, ctx);

    HostScene sc; bool have_sensor = false, have_integrator = false;
    std::map<const void *, int> texture_index;
    std::set<const void *> attached_emitters;
    g_textures = &sc.textures; g_texture_index = &texture_index; g_base_dir = base_dir; g_attached_emitters = &attached_emitters;
    struct TexScope { ~TexScope() { g_textures = nullptr; g_texture_index = nullptr; g_attached_emitters = nullptr; } } tex_scope;
    std::map<const Obj *, uint32_t> group_of;
    for (auto &c : top->children) {
        const Obj &o = *c.second;
        if (o.tag == "integrator") {
            if (have_integrator) fail("Only one integrator can be specified per scene.");
            sc.integrator = o.props; have_integrator = true;
        } else if (o.tag == "sensor") {
            if (have_sensor) fail("only one sensor is supported");
            make_sensor(o, sc); have_sensor = true;
        } else if (o.tag == "emitter" && o.plugin == "area") {
            // an area emitter declared at scene level waits for the shape that references it (scene.cpp:44-47 skips surface emitters among the scene's
            // children: they are counted through their shape)
        } else if (o.tag == "emitter") {
            if (o.plugin != "point" && o.plugin != "spot" && o.plugin != "constant" && o.plugin != "envmap" && o.plugin != "directional") fail("unsupported emitter plugin \"" + o.plugin + "\" (supported: point, spot, directional, constant, envmap; area inside a shape)");
            HostEmitter e; e.kind = 0;
            if (o.plugin == "constant") {   // src/emitters/constant.cpp:58-67: the scene's environment (scene.cpp:53-57); its bounding sphere follows in build_scene_blob
                for (auto &pe : sc.emitters) if (pe.kind == EMITTER_CONSTANT || pe.kind == EMITTER_ENVMAP) fail("Only one environment emitter can be specified per scene.");
                e.kind = EMITTER_CONSTANT;
                auto rc = o.colors.find("radiance");
                if (rc != o.colors.end()) for (int i = 0; i < 3; ++i) e.intensity[i] = (float) rc->second[i];
                else { float v = (float) o.props.get_float("radiance", 1.0); e.intensity[0] = e.intensity[1] = e.intensity[2] = v; }
            } else if (o.plugin == "directional") {   // src/emitters/directional.cpp:65-91: direction of travel = to_world * (0, 0, 1), or the normalised `direction`
                e.kind = EMITTER_DIRECTIONAL;
                auto dv = o.vectors.find("direction"); auto tws = o.transforms.find("to_world");
                if (dv != o.vectors.end()) {
                    if (tws != o.transforms.end()) fail("Only one of the parameters 'direction' and 'to_world' can be specified at the same time!'");
                    float v[3] = { (float) dv->second[0], (float) dv->second[1], (float) dv->second[2] };
                    for (int pass = 0; pass < 2; ++pass) {   // dr::normalize of the property, then look_at normalises target - origin once more (both in float32)
                        const float inv = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                        for (int k = 0; k < 3; ++k) v[k] *= inv;
                    }
                    for (int k = 0; k < 3; ++k) e.to_local[k] = v[k];
                } else {
                    Xf xf; if (tws != o.transforms.end()) xf = tws->second; else { xf.m = m_identity(); xf.inv = m_identity(); }
                    float m[16]; to_f32(xf.m, m);
                    e.to_local[0] = m[2]; e.to_local[1] = m[6]; e.to_local[2] = m[10];
                }
                auto ics = o.colors.find("irradiance");
                if (ics != o.colors.end()) for (int i = 0; i < 3; ++i) e.intensity[i] = (float) ics->second[i];
                else { float v = (float) o.props.get_float("irradiance", 1.0); e.intensity[0] = e.intensity[1] = e.intensity[2] = v; }
            } else if (o.plugin == "envmap") {   // src/emitters/envmap.cpp:116-224; tables and bounding sphere follow in build_scene_blob
                for (auto &pe : sc.emitters) if (pe.kind == EMITTER_CONSTANT || pe.kind == EMITTER_ENVMAP) fail("Only one environment emitter can be specified per scene.");
                e.kind = EMITTER_ENVMAP;
                const std::string fn = o.props.get_string("filename", "");
                if (fn.empty()) fail("Property \"filename\" has not been specified!");
                e.mis_compensation = o.props.get_bool("mis_compensation", false);   // envmap.cpp:157-185: the sampling tables are built from max(luminance - mean, 0)
                const std::string path = resolve_path(fn);
                read_radiance_image(path, e.image, e.image_w, e.image_h, srgb_to_linear_u8);
                if (e.image_w < 2 || e.image_h < 3) fail("\"" + fn.substr(fn.find_last_of('/') == std::string::npos ? 0 : fn.find_last_of('/') + 1) + "\": the environment map resolution must be at least 2x3 pixels");
                e.scale = (float) o.props.get_float("scale", 1.0);
                auto tws = o.transforms.find("to_world");
                Xf xf; if (tws != o.transforms.end()) xf = tws->second; else { xf.m = m_identity(); xf.inv = m_identity(); }
                float m[16], inv[16]; to_f32(xf.m, m); to_f32(xf.inv, inv);
                for (int k = 0; k < 12; ++k) { e.to_world[k] = m[k]; e.to_local[k] = inv[k]; }
            } else if (o.plugin == "spot") {   // src/emitters/spot.cpp:75-100; position = translation of to_world, axis = its +z
                e.kind = EMITTER_SPOT;
                auto tws = o.transforms.find("to_world");
                Xf xf; if (tws != o.transforms.end()) xf = tws->second; else { xf.m = m_identity(); xf.inv = m_identity(); }
                float m[16], inv[16]; to_f32(xf.m, m); to_f32(xf.inv, inv);
                e.pos[0] = m[3]; e.pos[1] = m[7]; e.pos[2] = m[11];
                for (int k = 0; k < 12; ++k) e.to_local[k] = inv[k];
                float cutoff = (float) o.props.get_float("cutoff_angle", 20.0);
                float beam = (float) o.props.get_float("beam_width", (double) (cutoff * 3.0f / 4.0f));
                if (!std::isfinite(cutoff) || !std::isfinite(beam) || std::fabs(cutoff) > 360.f || std::fabs(beam) > 360.f) fail("spot: cutoff_angle and beam_width must be finite angles in degrees");
                cutoff = cutoff * (kPi / 180.f); beam = beam * (kPi / 180.f);                     // dr::deg_to_rad
                e.cutoff_angle = cutoff; e.inv_transition = 1.0f / (cutoff - beam);
                e.cos_cutoff = cos_(cutoff); e.cos_beam = cos_(beam);
                if (!(cutoff >= beam)) fail("spot: cutoff_angle must not be smaller than beam_width");
                if (o.props.has("texture") || o.colors.count("texture")) fail("spot: textured spot lights are not supported");
                auto ics = o.colors.find("intensity");
                if (ics != o.colors.end()) for (int i = 0; i < 3; ++i) e.intensity[i] = (float) ics->second[i];
                else { float v = (float) o.props.get_float("intensity", 1.0); e.intensity[0] = e.intensity[1] = e.intensity[2] = v; }
            } else {
            auto pv = o.vectors.find("position"); auto tw = o.transforms.find("to_world");
            if (pv != o.vectors.end()) {
                if (tw != o.transforms.end()) fail("Only one of the parameters 'position' and 'to_world' can be specified at the same time!'");
                for (int i = 0; i < 3; ++i) e.pos[i] = (float) pv->second[i];
            } else {
                float m[16]; to_f32(tw != o.transforms.end() ? tw->second.m : m_identity(), m);
                e.pos[0] = m[3]; e.pos[1] = m[7]; e.pos[2] = m[11];
            }
            auto ic = o.colors.find("intensity");
            if (ic != o.colors.end()) for (int i = 0; i < 3; ++i) e.intensity[i] = (float) ic->second[i];
            else { float v = (float) o.props.get_float("intensity", 1.0); e.intensity[0] = e.intensity[1] = e.intensity[2] = v; }
            }
            {
                auto u = o.props.unqueried();
                if (!u.empty()) fail_unreferenced(u, "emitter", o.plugin);
                check_colors(o, { "intensity", "radiance", "irradiance" });
            }
            sc.emitters.push_back(e);
        } else if (o.tag == "shape") {
            if (o.plugin == "shapegroup") {
                HostGroup g; g.first_shape = (uint32_t) sc.shapes.size();
                for (auto &ch : o.children) {
                    if (ch.first != "shape") fail("Tried to add an unsupported object to a shapegroup");
                    if (ch.second->plugin == "instance") fail("Nested instancing is not permitted");
                    if (ch.second->plugin == "shapegroup") fail("Nested ShapeGroup is not permitted");
                    for (auto &c2 : ch.second->children) if (c2.first == "sensor") fail("Instancing of sensors is not supported");   // shapegroup.cpp:29-30
                    sc.shapes.push_back(make_shape(*ch.second, false, base_dir));
                    if (sc.shapes.back().emitter) fail("Instancing of emitters is not supported");
                }
                g.n_shapes = (uint32_t) sc.shapes.size() - g.first_shape;
                group_of[&o] = (uint32_t) sc.groups.size(); sc.groups.push_back(g);
            } else if (o.plugin == "instance") {
                const Obj *grp = nullptr;
                for (auto &ch : o.children) if (ch.first == "shape" && ch.second->plugin == "shapegroup") {
                    if (grp) fail("Only a single shapegroup can be specified per instance.");
                    grp = ch.second.get();
                }
                if (!grp) fail("A reference to a 'shapegroup' must be specified!");
                auto g = group_of.find(grp);
                if (g == group_of.end()) fail("an instance must reference a shapegroup declared before it at scene level");
                sc.objects.push_back(make_instance(o, g->second));
            } else if (o.animations.count("to_world")) {
                // xml.cpp:1165-1195: shape with an animated to_world => shapegroup{shape} + instance{animated to_world}
                HostGroup g; g.first_shape = (uint32_t) sc.shapes.size(); g.n_shapes = 1;
                sc.shapes.push_back(make_shape(o, true, base_dir));
                sc.groups.push_back(g);
                sc.objects.push_back(make_instance(o, (uint32_t) sc.groups.size() - 1));
            } else {
                HostObject ob; ob.kind = OBJ_SHAPE; ob.index = (uint32_t) sc.shapes.size(); ob.n_keys = 0; memset(ob.key, 0, sizeof ob.key);
                sc.shapes.push_back(make_shape(o, false, base_dir));
                sc.objects.push_back(ob);
                if (sc.shapes.back().emitter) {   // scene.cpp:33-35: the shape's emitter joins the list at the shape's position
                    HostEmitter e; e.kind = EMITTER_AREA; e.shape = ob.index; memcpy(e.intensity, sc.shapes.back().radiance, 12);
                    sc.emitters.push_back(e);
                }
            }
        } else if (o.tag == "bsdf") {
            HostShape probe; bsdf_of(o, probe);   // top-level declarations are referenced by id; every object is instantiated, so a malformed one fails even if nothing refers to it
        } else if (o.tag == "texture") {
            // instantiated when a BSDF refers to it
        } else if (o.tag == "sampler" || o.tag == "film" || o.tag == "rfilter") {
            fail("unreferenced object \"" + o.plugin + "\" (within scene of type \"scene\")");   // Scene takes sensors, emitters, shapes and integrators (scene.cpp:102-121)
        } else fail("unsupported top-level element <" + o.tag + ">");
    }
    {   // the scene is a plugin too: properties given at scene level that Scene does not query are unreferenced (xml.cpp:1204-1218)
        auto u = top->props.unqueried();
        if (!u.empty()) fail_unreferenced(u, "scene", "scene");
        check_colors(*top, {});
        if (!top->vectors.empty()) fail_unreferenced({ top->vectors.begin()->first }, "scene", "scene");
        if (!top->transforms.empty()) fail_unreferenced({ top->transforms.begin()->first }, "scene", "scene");
    }
    if (!scene_root) fail("root element \"" + root->tag + "\": only <scene> descriptions can be rendered by this library");
    sc.has_sensor = have_sensor;   // a scene without a sensor loads (as in the reference); rendering it is the error
    if (!have_integrator) { sc.integrator = PropBag(); sc.integrator.plugin = "path"; }
    return sc;
}

std::string read_file(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("could not open \"" + path + "\"");
    std::ostringstream ss; ss << f.rdbuf();
    return ss.str();
}

}  // namespace dtof
