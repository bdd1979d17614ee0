"""Oracle restatement of the reference's scene-XML semantics for the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/dtof_oracle.h).  Pure Python / numpy, written
independently of the product's C++ loader so the two can be compared.

Follows (reference paths relative to /root/reference):
  * src/core/xml.cpp:441-456,630-648   <default> + $param substitution (longest name first)
  * src/core/xml.cpp:882-1007          <transform>/<animation>, ops left-multiply, composed in double
  * src/core/xml.cpp:792-822           <rgb> with 1 or 3 tokens
  * src/core/xml.cpp:1165-1195         animated shape -> shapegroup + instance rewrite
  * src/core/transform.cpp:22-36       AnimatedTransform::append (keyframes cast to float32)
  * src/render/sensor.cpp:14-20,127-203, src/sensors/perspective.cpp:139-152   sensor parameters
  * src/render/film.cpp:7-54, src/rfilters/tent.cpp:47-55                        film / filter
  * src/integrators/dopplertofpath.cpp:19-57, src/render/integrator.cpp:54-100,568-585,
    src/samplers/correlated.cpp:17-23, src/render/sampler.cpp:11-20             plugin parameters
"""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- transforms (double)
# A transform is a pair (matrix, inverse) of row-major 4x4 lists of Python floats (IEEE double), like the
# reference's Transform (matrix + inverse_transpose, include/mitsuba/core/transform.h:43-70): translate / scale /
# rotate / lookat carry ANALYTIC inverses and composition multiplies both, only <matrix> is inverted numerically.
# Products are plain multiply-add loops in k order so that the C++ loader can mirror them bit for bit.
def _ident():
    return [1.0 if i % 5 == 0 else 0.0 for i in range(16)]


def _mul(a, b):
    r = [0.0] * 16
    for i in range(4):
        for j in range(4):
            acc = 0.0
            for k in range(4):
                acc += a[4 * i + k] * b[4 * k + j]
            r[4 * i + j] = acc
    return r


def _transpose(a):
    return [a[4 * (i % 4) + i // 4] for i in range(16)]


def _inverse(a):
    """Gauss-Jordan with partial pivoting (same algorithm as the product's m_inverse)"""
    w = [[a[4 * i + j] for j in range(4)] + [1.0 if i == j else 0.0 for j in range(4)] for i in range(4)]
    for c in range(4):
        piv = c
        for r in range(c + 1, 4):
            if abs(w[r][c]) > abs(w[piv][c]):
                piv = r
        if w[piv][c] == 0.0:
            raise ValueError("singular transformation matrix")
        if piv != c:
            w[piv], w[c] = w[c], w[piv]
        d = 1.0 / w[c][c]
        w[c] = [x * d for x in w[c]]
        for r in range(4):
            if r != c:
                f = w[r][c]
                if f != 0.0:
                    w[r] = [x - f * y for x, y in zip(w[r], w[c])]
    return [w[i][4 + j] for i in range(4) for j in range(4)]


def _translate(v):
    m, inv = _ident(), _ident()
    m[3], m[7], m[11] = v
    inv[3], inv[7], inv[11] = -v[0], -v[1], -v[2]
    return m, inv


def _scale(v):
    m, inv = _ident(), _ident()
    m[0], m[5], m[10] = v
    inv[0], inv[5], inv[10] = 1.0 / v[0], 1.0 / v[1], 1.0 / v[2]
    return m, inv


def _rotate(axis, angle_deg):
    # Transform4f::rotate (transform.h:180-184) -> dr::rotate<Matrix4>(axis, rad): Rodrigues, axis used as given;
    # inverse = transpose
    x, y, z = axis
    th = angle_deg * (math.pi / 180.0)
    s, c = math.sin(th), math.cos(th)
    cm = 1.0 - c
    m = _ident()
    m[0] = x * x * cm + c;      m[1] = x * y * cm - z * s;  m[2] = x * z * cm + y * s
    m[4] = x * y * cm + z * s;  m[5] = y * y * cm + c;      m[6] = y * z * cm - x * s
    m[8] = x * z * cm - y * s;  m[9] = y * z * cm + x * s;  m[10] = z * z * cm + c
    return m, _transpose(m)


def _normalize(v):
    il = 1.0 / math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    return [v[0] * il, v[1] * il, v[2] * il]


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def _coordinate_system(n):
    # include/mitsuba/core/vector.h:116-136 (double); returns the first vector only
    sign = math.copysign(1.0, n[2])
    a = -1.0 / (sign + n[2])
    b = n[0] * n[1] * a
    return [(n[0] * n[0] * a) * sign + 1.0, b * sign, -n[0] * sign]


def _look_at(o, t, u):
    # transform.h:255-283: columns left, new_up, dir, origin; inverse = [R^T | -R^T o]
    d = _normalize([t[0] - o[0], t[1] - o[1], t[2] - o[2]])
    left = _normalize(_cross(u, d))
    nu = _cross(d, left)
    m, inv = _ident(), _ident()
    for r in range(3):
        m[4 * r], m[4 * r + 1], m[4 * r + 2], m[4 * r + 3] = left[r], nu[r], d[r], o[r]
    rows = (left, nu, d)
    for r in range(3):
        inv[4 * r], inv[4 * r + 1], inv[4 * r + 2] = rows[r]
        inv[4 * r + 3] = -(rows[r][0] * o[0] + rows[r][1] * o[1] + rows[r][2] * o[2])
    return m, inv


def _tokens(s):
    return [t for t in s.replace(",", " ").split() if t]


def _vec(node, default=0.0):
    # detail::expand_value_to_xyz + parse_vector (xml.cpp)
    def num(x):
        try:
            return float(x)
        except ValueError:
            raise ValueError('could not parse floating point value "%s".' % x)
    if node.get("value") is not None:
        t = _tokens(node.get("value"))
        if len(t) == 1:
            t = t * 3
        return [num(x) for x in t]
    return [num(node.get(k, default)) for k in ("x", "y", "z")]


def _parse_transform(node):
    m, inv = _ident(), _ident()
    for op in node:
        tag = op.tag
        if tag == "matrix":
            t = [float(x) for x in _tokens(op.get("value"))]
            if len(t) == 16:
                mm = t
            elif len(t) == 9:
                mm = _ident()
                for i in range(3):
                    for j in range(3):
                        mm[4 * i + j] = t[3 * i + j]
            else:
                raise ValueError("matrix: expected 16 or 9 values")
            mi = _inverse(mm)
        elif tag == "translate":
            mm, mi = _translate(_vec(op))
        elif tag == "scale":
            mm, mi = _scale(_vec(op, 1.0))
        elif tag == "rotate":
            mm, mi = _rotate(_vec(op), float(op.get("angle")))
        elif tag == "lookat":
            o = [float(x) for x in _tokens(op.get("origin"))]
            t = [float(x) for x in _tokens(op.get("target"))]
            u = [float(x) for x in _tokens(op.get("up", "0,0,0"))]
            if u[0] * u[0] + u[1] * u[1] + u[2] * u[2] == 0:
                u = _coordinate_system(_normalize([t[0] - o[0], t[1] - o[1], t[2] - o[2]]))
            mm, mi = _look_at(o, t, u)
            if any(math.isnan(x) for x in mm):
                raise ValueError("invalid lookat transformation")
        else:
            raise ValueError("transform nodes can only contain transform operations (got <%s>)" % tag)
        m, inv = _mul(mm, m), _mul(inv, mi)   # ctx.transform = T(op) * ctx.transform (matrix and inverse)
    return m, inv


# ----------------------------------------------------------------------------- property bags
class Props(dict):
    """name -> (type, value); type in {float,int,bool,string,rgb,transform,animation,object,ref}"""

    def __init__(self, plugin, ident=None):
        super().__init__()
        self.plugin = plugin
        self.id = ident
        self.children = []   # nested objects in document order: (tag, Props)
        self.queried = set()

    def get_f(self, name, default):
        if name in self:
            self.queried.add(name)
            t, v = self[name]
            if t not in ("float", "int"):
                raise ValueError('property "%s" has the wrong type' % name)
            return float(v)
        return default

    def get_i(self, name, default):
        if name in self:
            self.queried.add(name)
            t, v = self[name]
            if t != "int":
                raise ValueError('property "%s" has the wrong type (expected <integer>)' % name)
            return int(v)
        return default

    def get_b(self, name, default):
        if name in self:
            self.queried.add(name)
            t, v = self[name]
            if t != "bool":
                raise ValueError('property "%s" has the wrong type (expected <boolean>)' % name)
            return bool(v)
        return default

    def get_s(self, name, default):
        if name in self:
            self.queried.add(name)
            t, v = self[name]
            if t != "string":
                raise ValueError('The property "%s" has the wrong type (expected <string>).' % name)
            return v
        return default

    def check_unreferenced(self, kind, known_colours=(), scalars_only=True):
        """xml.cpp:1204-1222 after a plugin has been instantiated: <rgb> / <spectrum> children are texture OBJECTS of the plugin's Properties, one
        it does not ask for is an unreferenced object; any other property it did not query is an unreferenced property (the reference prints
        the list of names inside the quotes of its format string)"""
        for name, (t, _v) in self.items():
            if t == "rgb" and name not in known_colours:
                raise ValueError('unreferenced object "%s" (within %s of type "%s")' % (name, kind, self.plugin))
        left = [n for n, (t, _v) in self.items() if n not in self.queried and (t in ("float", "int", "bool", "string") or not scalars_only) and t != "rgb"]
        if left:
            raise ValueError('unreferenced %s "[%s]" in %s plugin of type "%s"' % ("properties" if len(left) > 1 else "property",
                             ", ".join('"%s"' % n for n in left), kind, self.plugin))


_OBJECT_TAGS = {"scene", "integrator", "sensor", "sampler", "film", "rfilter", "bsdf", "shape", "emitter", "texture"}


# ---------------------------------------------------------------------------------------------------- well-formedness (xml.cpp:258-310,470-560)
# The checks parse_xml makes on every node before it looks at values, with the reference's messages
# `Error while loading "<id>" (at line L, col C): <message>.` (XMLSource::throw_error, xml.cpp:213-217; the position is that of the tag name).
# Pinned by the reference's own tests: tests/golden/reference_xml_cases.json (src/core/tests/test_xml.py).
_POS = {}      # id(element) -> (line, col) of its tag name, 1-based


def _parse_text(text):
    """the document as an ElementTree, built over expat so that every element's position (of its tag name) is known"""
    import xml.parsers.expat as expat
    parser = expat.ParserCreate()
    parser.ordered_attributes = False
    stack, roots = [], []

    def start(tag, attrs):
        el = ET.Element(tag, attrs) if not stack else ET.SubElement(stack[-1], tag, attrs)
        _POS[id(el)] = (parser.CurrentLineNumber, parser.CurrentColumnNumber + 2)
        if not stack:
            roots.append(el)
        stack.append(el)

    def end(_tag):
        stack.pop()

    def chars(data):
        if data.strip():
            raise ValueError('Error while loading "<string>" (at line %d, col %d): unexpected content.' % (parser.CurrentLineNumber, parser.CurrentColumnNumber + 1))
    parser.StartElementHandler, parser.EndElementHandler, parser.CharacterDataHandler = start, end, chars
    try:
        parser.Parse(text, True)
    except expat.ExpatError as e:
        raise ValueError('Error while loading "<string>" (at line %d, col %d): %s.' % (e.lineno, e.offset + 1, expat.ErrorString(e.code)))
    if not roots:
        raise ValueError('Error while loading "<string>": no root element')
    return roots[0]


def _fail_at(node, msg):
    line, col = _POS.get(id(node), (0, 0))
    raise ValueError('Error while loading "<string>" (at line %d, col %d): %s.' % (line, col, msg))


_TRANSFORM_OPS = ("translate", "rotate", "scale", "lookat", "matrix")
_PROPERTY_TAGS = ("float", "integer", "boolean", "string", "rgb", "spectrum")


def _kind(node):
    t = node.tag
    if t in _OBJECT_TAGS or t in ("medium", "phase", "volume") or (t == "spectrum" and node.get("type") is not None):
        return "object"
    if t in _PROPERTY_TAGS:
        return "property"
    if t in ("point", "vector"):
        return "vector"
    if t in ("transform", "animation", "ref", "default", "path", "include", "alias"):
        return t
    if t in _TRANSFORM_OPS:
        return "op"
    return None


def _check_attributes(node, allowed, expect_all=True, may_be_empty=False):
    allowed = list(allowed)
    found_one = may_be_empty        # `id` / `name` of objects and references are added by the reference itself when missing
    for a in node.attrib:
        if a not in allowed:
            _fail_at(node, 'unexpected attribute "%s" in element "%s"' % (a, node.tag))
        allowed.remove(a); found_one = True
    if allowed and (not found_one or expect_all):
        _fail_at(node, 'missing attribute "%s" in element "%s"' % (sorted(allowed)[0], node.tag))


def _upgrade_tree(node, parent=None):
    """upgrade_tree (xml.cpp:338-365), scene descriptions older than 2.0.0: camelCase names -> underscore_case, lookAt -> lookat, reserved ids renamed"""
    if node.tag == "lookAt":
        node.tag = "lookat"
    name = node.get("name")
    if name is not None and node.tag != "default":
        out, i = name, 0
        while i + 1 < len(out):
            if out[i].islower() and out[i + 1].isupper():
                out = out[:i + 1] + "_" + out[i + 1:]
                i += 2
                while i < len(out) and out[i].isupper():
                    out = out[:i] + out[i].lower() + out[i + 1:]
                    i += 1
            i += 1
        if out == "diffuse_reflectance" and parent is not None and parent.tag == "bsdf" and parent.get("type") == "diffuse":
            out = "reflectance"
        node.set("name", out)
    ident = node.get("id")
    if ident and ident.startswith("_"):
        node.set("id", "ID" + ident + "__UPGR")
    for ch in node:
        _upgrade_tree(ch, node)


def _check_tree(node, parent_kind, depth, ids):
    kind = _kind(node)
    if kind is None:
        _fail_at(node, 'unexpected tag "%s"' % node.tag)
    if parent_kind is None and kind != "object":
        _fail_at(node, 'root element "%s" must be an object' % node.tag)
    if (parent_kind == "transform") != (kind == "op"):
        _fail_at(node, "transform nodes can only contain transform operations" if parent_kind == "transform"
                 else "transform operations can only occur in a transform node")
    if parent_kind is not None and parent_kind != "object" and not ((parent_kind == "transform" and kind == "op") or (parent_kind == "animation" and kind == "transform")):
        _fail_at(node, 'node "%s" cannot occur as child of a property' % node.tag)
    if depth == 0 and node.get("version") is None:
        _fail_at(node, 'missing version attribute in root element "%s"' % node.tag)
    version = node.get("version")
    if version is not None:
        parts = version.split(".")
        if len(parts) != 3 or not all(x.isdigit() for x in parts):
            _fail_at(node, 'could not parse version number "%s"' % version)
        if int(parts[0]) < 2:
            _upgrade_tree(node)
        del node.attrib["version"]
    name, ident = node.get("name"), node.get("id")
    if name is not None and name.startswith("_"):
        _fail_at(node, 'invalid parameter name "%s" in element "%s": leading underscores are reserved for internal identifiers' % (name, node.tag))
    if ident is not None and ident.startswith("_"):
        _fail_at(node, 'invalid id "%s" in element "%s": leading underscores are reserved for internal identifiers' % (ident, node.tag))
    if kind == "object":
        _check_attributes(node, ["id", "name"] + ([] if node.tag == "scene" else ["type"]), False, True)
        if node.tag != "scene" and node.get("type") is None:
            _fail_at(node, 'missing attribute "type" in element "%s"' % node.tag)
        if ident is not None and ident in ids:
            _fail_at(node, '"%s" has duplicate id "%s" (previous was at line %d, col %d)' % ((node.tag, ident) + ids[ident]))
        names = set()
        for ch in node:
            _check_tree(ch, "object", depth + 1, ids)
            cn = ch.get("name")
            if cn and ch.tag != "default":
                if cn in names:
                    _fail_at(ch, 'Property "%s" was specified multiple times!' % cn)
                names.add(cn)
        if ident is not None:
            ids[ident] = _POS.get(id(node), (0, 0))
        return
    if kind == "ref":
        _check_attributes(node, ["id", "name"], False, True)
        if node.get("id") is None:
            _fail_at(node, 'missing attribute "id" in element "ref"')
    elif kind == "alias":
        _check_attributes(node, ["id", "as"])
    elif kind == "default":
        _check_attributes(node, ["name", "value"])
        if not node.get("name"):
            _fail_at(node, "<default>: name must by nonempty")
        if "," in node.get("name"):
            _fail_at(node, "Invalid character in parameter name: ',' in %s" % node.get("name"))
    elif kind == "path":
        _check_attributes(node, ["value"])
        if depth != 1:
            _fail_at(node, "<path>: path can only be child of root")
    elif kind == "include":
        _check_attributes(node, ["filename"])
    elif kind == "property":
        if node.tag == "spectrum":
            _check_attributes(node, ["name", "value", "filename"], False, True)
        else:
            _check_attributes(node, ["name", "value"])
    elif kind in ("vector", "op"):
        if node.tag == "lookat":
            _check_attributes(node, ["origin", "target", "up"], False, True)
        elif node.tag == "matrix":
            _check_attributes(node, ["value"])
        else:
            if node.get("value") is not None:      # expand_value_to_xyz (xml.cpp:290-309)
                if any(node.get(k) is not None for k in "xyz"):
                    _fail_at(node, 'can\'t mix and match "value" and "x"/"y"/"z" attributes')
                if len(_tokens(node.get("value"))) not in (1, 3):
                    _fail_at(node, '"value" attribute must have exactly 1 or 3 elements')
            _check_attributes(node, (["name"] if kind == "vector" else ["angle"] if node.tag == "rotate" else []) + ["x", "y", "z", "value"], False, True)
    elif kind == "transform":
        _check_attributes(node, ["time"] if parent_kind == "animation" else ["name"], False, True)
    elif kind == "animation":
        _check_attributes(node, ["name"])
    for ch in node:
        _check_tree(ch, kind, depth + 1, ids)


def _number(text, kind):
    """string::stof / detail::stoll as the XML front end uses them (xml.cpp:736-757): the whole value must parse, surrounding blanks are allowed"""
    try:
        t = text.strip()
        if kind == "integer":
            if not t or not (t.lstrip("+-").isdigit()):
                raise ValueError
            return int(t)
        if not t or any(c.isspace() or c == "," for c in t) or t.lower().rstrip("f") != t.lower() and True and t[-1] in "fF":
            raise ValueError
        return float(t)
    except ValueError:
        raise ValueError('could not parse %s value "%s".' % ("integer" if kind == "integer" else "floating point", text))


_SEARCH_PATHS = []   # FileResolver (src/core/fresolver.cpp): the scene file's directory and whatever <path> prepends (xml.cpp:651-668)
MAX_INCLUDE_DEPTH = 15   # MI_XML_INCLUDE_MAX_RECURSION


def resolve_path(fn):
    if not fn or os.path.isabs(fn):
        return fn
    for d in _SEARCH_PATHS:
        if os.path.exists(os.path.join(d, fn)):
            return os.path.join(d, fn)
    return os.path.join(_SEARCH_PATHS[-1], fn) if _SEARCH_PATHS else fn


def _substitute(root, params, base_dir=""):
    # xml.cpp:441-456: replace $name in every attribute, longest names first; undefined => error.  In the same document-order walk:
    # <path> (xml.cpp:651-668) and <include> (xml.cpp:670-725: the children of an included <scene>, or the included object itself, replace the tag)
    defaults = {}
    used = set()
    def expand(node, depth, inc_depth, src_dir):
        i = 0
        while i < len(node):
            ch = node[i]
            walk(ch, depth + 1, inc_depth, src_dir)
            if ch.tag != "include":
                i += 1
                continue
            extra = [k for k in ch.attrib if k != "filename"]
            if extra:
                raise ValueError('unexpected attribute "%s" in element "include"' % extra[0])
            if ch.get("filename") is None:
                raise ValueError('missing attribute "filename" in element "include"')
            path = resolve_path(ch.get("filename"))
            if not os.path.exists(path):
                raise ValueError('included file "%s" not found' % path)
            if inc_depth + 1 > MAX_INCLUDE_DEPTH:
                raise ValueError("Exceeded <include> recursion limit of %d" % MAX_INCLUDE_DEPTH)
            try:
                inc_root = ET.parse(path).getroot()
            except ET.ParseError as e:
                raise ValueError('error while loading "%s": %s' % (path, e))
            holder = ET.Element("holder")
            if inc_root.tag == "scene":
                holder.extend(list(inc_root)); hd = 0
            else:
                holder.append(inc_root); hd = -1
            expand(holder, hd, inc_depth + 1, os.path.dirname(os.path.abspath(path)))
            node.remove(ch)
            for k, new in enumerate(list(holder)):
                node.insert(i + k, new)
            i += len(holder)
    def walk(node, depth=0, inc_depth=0, src_dir=""):
        names = sorted(defaults, key=len, reverse=True)
        for k, v in list(node.attrib.items()):
            if "$" in v:
                for n in names:
                    if "$" + n in v:
                        used.add(n)
                    v = v.replace("$" + n, defaults[n])
                if "$" in v:
                    raise ValueError('undefined parameter(s) in string: "%s"!' % v)
                node.set(k, v)
        if node.tag == "default":
            n = node.get("name")
            if n not in defaults:
                defaults[n] = node.get("value")
                used.add(n)
        if node.tag == "path":
            if depth != 1:
                raise ValueError("<path>: path can only be child of root")
            p = node.get("value")
            if not os.path.isabs(p):
                local = os.path.join(src_dir, p) if src_dir else p
                p = local if os.path.exists(local) else resolve_path(p)
            if not os.path.exists(p):
                raise ValueError('<path>: folder "%s" not found' % p)
            _SEARCH_PATHS.insert(0, p)
        expand(node, depth, inc_depth, src_dir)
    defaults.update({k: str(v) for k, v in params.items()})
    del _SEARCH_PATHS[:]
    if base_dir:
        _SEARCH_PATHS.append(base_dir)
    walk(root, 0, 0, base_dir)
    for k in params:            # xml.cpp:1067-1070: a parameter handed to the loader that no attribute referred to
        if k not in used:
            raise ValueError('Unused parameter "%s"!' % k)


def _parse_object(node, registry):
    p = Props(node.get("type"), node.get("id"))
    for ch in node:
        tag, name = ch.tag, ch.get("name")
        if tag in ("default", "path"):
            continue
        if tag == "alias":   # xml.cpp:608-628: a second id for an object declared earlier
            src, dst = ch.get("id"), ch.get("as")
            if dst in registry:
                raise ValueError('"alias" has duplicate id "%s"' % dst)
            if src not in registry:
                raise ValueError('referenced id "%s" not found' % src)
            registry[dst] = registry[src]
            continue
        if tag in _OBJECT_TAGS:
            child = _parse_object(ch, registry)
            p.children.append((tag, child, name))
        elif tag == "ref":
            p.children.append(("ref", ch.get("id"), name))
        elif tag == "float":
            p[name] = ("float", _number(ch.get("value"), "float"))
        elif tag == "integer":
            p[name] = ("int", _number(ch.get("value"), "integer"))
        elif tag == "boolean":
            v = ch.get("value").lower()
            if v not in ("true", "false"):
                raise ValueError('could not parse boolean value "%s" -- must be "true" or "false".' % v)
            p[name] = ("bool", v == "true")
        elif tag == "string":
            p[name] = ("string", ch.get("value"))
        elif tag in ("point", "vector"):
            p[name] = ("vector", _vec(ch))
        elif tag == "rgb":
            t = _tokens(ch.get("value"))
            if len(t) == 1:
                t = t * 3
            if len(t) != 3:
                raise ValueError("'rgb' tag requires one or three values")
            p[name] = ("rgb", [float(x) for x in t])
        elif tag == "spectrum":
            t = _tokens(ch.get("value"))
            if len(t) != 1:
                raise ValueError("only constant <spectrum> values are supported")
            p[name] = ("rgb", [float(t[0])] * 3)
        elif tag == "transform":
            p[name] = ("transform", _parse_transform(ch))
        elif tag == "animation":
            keys = [(F32(float(tr.get("time"))), _parse_transform(tr)) for tr in ch]
            for a, b in zip(keys, keys[1:]):
                if not b[0] > a[0]:
                    raise ValueError("AnimatedTransform::append(): time values must be strictly monotonically increasing!")
            p[name] = ("animation", keys)
        else:
            raise ValueError('unexpected tag "%s"' % tag)
    if p.id is not None:
        registry[p.id] = (node.tag, p)
    return p


# ----------------------------------------------------------------------------- flat description
def _m32(m):
    return np.asarray(m, dtype=np.float64).reshape(4, 4).astype(F32)


# include/mitsuba/render/ior.h:16-44 (physical constants)
IOR_TABLE = {"vacuum": 1.0, "helium": 1.000036, "hydrogen": 1.000132, "air": 1.000277, "carbon dioxide": 1.00045, "water": 1.3330,
             "acetone": 1.36, "ethanol": 1.361, "carbon tetrachloride": 1.461, "glycerol": 1.4729, "benzene": 1.501,
             "silicone oil": 1.52045, "bromine": 1.661, "water ice": 1.31, "fused quartz": 1.458, "pyrex": 1.470, "acrylic glass": 1.49,
             "polypropylene": 1.49, "bk7": 1.5046, "sodium chloride": 1.544, "amber": 1.55, "pet": 1.5750, "diamond": 2.419}


def _lookup_ior(props, name, default):
    """lookup_ior (ior.h:71-77): a <float> is taken as is, a <string> is a material name"""
    if name in props and props[name][0] in ("float", "int"):
        return F32(props.get_f(name, 0.0))
    key = props.get_s(name, default).lower()
    if key not in IOR_TABLE:
        raise ValueError('Unable to find an IOR value for "%s"!' % key)
    return F32(IOR_TABLE[key])


def _color(props, name, default):
    if name in props:
        t, v = props[name]
        props.queried.add(name)
        return np.asarray([v] * 3 if t in ("float", "int") else v, dtype=np.float64).astype(F32)
    return np.asarray([default] * 3, dtype=np.float64).astype(F32)


_SRGB_LUT = None


def _srgb_lut():
    """UInt8 sRGB -> linear float32 (StructConverter::linearize + dr::srgb_to_linear, src/core/struct.cpp:1600-1625)"""
    global _SRGB_LUT
    if _SRGB_LUT is None:
        x = np.arange(256, dtype=np.float64) / 255.0
        _SRGB_LUT = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4).astype(F32)
    return _SRGB_LUT


def read_radiance_image(path):
    """Bitmap(path).convert(RGB, Float32, srgb_gamma=false) for the formats the envmap fixtures use: PFM (bitmap.cpp:2164-2217), RGBE
    (:1988-2096), and 8-bit PNG / JPEG through PIL (sRGB -> linear).  Returns float32 (height, width, 3), row 0 = top."""
    import struct as _st
    data = open(path, "rb").read()
    if data[:2] in (b"PF", b"Pf"):
        tok, pos = [], 2
        while len(tok) < 3:
            while data[pos:pos + 1].isspace():
                pos += 1
            q = pos
            while not data[q:q + 1].isspace():
                q += 1
            tok.append(data[pos:q]); pos = q
        pos += 1
        w, h, so = int(tok[0]), int(tok[1]), float(tok[2])
        ch = 3 if data[1:2] == b"F" else 1
        a = np.frombuffer(data, "<f4" if so <= 0 else ">f4", w * h * ch, pos).astype(F32).reshape(h, w, ch)
        if abs(so) != 1:
            a = a * F32(abs(so))
        a = a[::-1]
        return np.ascontiguousarray(np.repeat(a, 3, axis=2) if ch == 1 else a)
    if data[:2] == b"#?":
        lines, pos, w, h, ok = [], 0, 0, 0, False
        while True:
            e = data.index(b"\n", pos); line = data[pos:e].decode("latin-1"); pos = e + 1
            if line.startswith("FORMAT=32-bit_rle_rgbe"):
                ok = True
            t = line.split()
            if len(t) == 4 and t[0] == "-Y" and t[2] == "+X":
                h, w = int(t[1]), int(t[3]); break
        if not ok:
            raise ValueError("read_rgbe(): unrecognized format!")
        px = np.zeros((h * w, 4), np.uint8)
        def flat(count, at):
            return np.frombuffer(data, np.uint8, 4 * count, at).reshape(count, 4)
        if w < 8 or w > 0x7fff:
            px[:] = flat(h * w, pos)
        else:
            y = 0
            while y < h:
                r = data[pos:pos + 4]
                if r[0] != 2 or r[1] != 2 or r[2] & 0x80:      # not run-length encoded from here on
                    px[y * w:] = flat(h * w - y * w, pos); break
                pos += 4
                if ((r[2] << 8) | r[3]) != w:
                    raise ValueError("read_rgbe(): wrong scanline width!")
                row = bytearray()
                for c in range(4):
                    end = (c + 1) * w
                    while len(row) < end:
                        n, v = data[pos], data[pos + 1]; pos += 2
                        if n > 128:
                            n -= 128
                            if n == 0 or n > end - len(row):
                                raise ValueError("read_rgbe(): bad scanline data!")
                            row += bytes([v]) * n
                        else:
                            if n == 0 or n > end - len(row):
                                raise ValueError("read_rgbe(): bad scanline data!")
                            row += bytes([v]) + data[pos:pos + n - 1]; pos += n - 1
                px[y * w:(y + 1) * w] = np.frombuffer(bytes(row), np.uint8).reshape(4, w).T
                y += 1
        f = np.ldexp(F32(1.0), px[:, 3].astype(np.int32) - 136).astype(F32)
        out = px[:, :3].astype(F32) * f[:, None]
        out[px[:, 3] == 0] = 0
        return out.reshape(h, w, 3)
    if data[:4] == b"\x76\x2f\x31\x01":      # OpenEXR, scan lines, compression NONE / ZIPS / ZIP, channels R G B (or Y), HALF or FLOAT
        import zlib
        pos, attrs = 8, {}
        while data[pos] != 0:
            e = data.index(b"\0", pos); name = data[pos:e].decode(); pos = e + 1
            e = data.index(b"\0", pos); pos = e + 1
            size = _st.unpack_from("<i", data, pos)[0]; pos += 4
            attrs[name] = data[pos:pos + size]; pos += size
        pos += 1
        comp = attrs["compression"][0]
        if comp == 4:      # PIZ: the stand-alone reader that decoded the authors' scene.exr (tools/exr_piz.py, test infrastructure like this file)
            import sys as _sys
            _sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
            import exr_piz
            ch, _ = exr_piz.read_exr(path)
            if all(c in ch for c in "RGB"):
                return np.ascontiguousarray(np.stack([ch["R"], ch["G"], ch["B"]], -1).astype(F32))
            if "Y" in ch:
                return np.ascontiguousarray(np.repeat(ch["Y"][..., None], 3, axis=2).astype(F32))
            raise ValueError("read_exr(): no R, G, B or Y channels")
        if comp not in (0, 2, 3):
            raise ValueError("read_exr(): only uncompressed, ZIP- and PIZ-compressed scan-line files are supported")
        x0, y0, x1, y1 = _st.unpack("<4i", attrs["dataWindow"])
        w, h = x1 - x0 + 1, y1 - y0 + 1
        chans, cd, p = [], attrs["channels"], 0
        while cd[p] != 0:
            e = cd.index(b"\0", p); chans.append((cd[p:e].decode(), _st.unpack_from("<i", cd, e + 1)[0])); p = e + 17
        lines = 16 if comp == 3 else 1
        planes = {n: np.zeros((h, w), F32) for n, _ in chans}
        for off in _st.unpack_from("<%dQ" % ((h + lines - 1) // lines), data, pos):
            y, size = _st.unpack_from("<2i", data, off)
            ny = min(lines, y1 - y + 1)
            raw_len = sum(2 if t == 1 else 4 for _, t in chans) * w * ny
            buf = data[off + 8:off + 8 + size]
            if comp and size < raw_len:
                d = np.frombuffer(zlib.decompress(buf), np.uint8).astype(np.int64)
                t = ((np.cumsum(d) - 128 * np.arange(len(d))) & 255).astype(np.uint8)     # undo the predictor: t[i] = t[i-1] + d[i] - 128
                half = (len(t) + 1) // 2
                b = np.empty(len(t), np.uint8); b[0::2] = t[:half]; b[1::2] = t[half:]
                buf = b.tobytes()
            q = 0
            for r in range(ny):
                for n, t in chans:
                    if t == 1:
                        planes[n][y - y0 + r] = np.frombuffer(buf, "<f2", w, q).astype(F32); q += 2 * w
                    else:
                        planes[n][y - y0 + r] = np.frombuffer(buf, "<f4" if t == 2 else "<u4", w, q).astype(F32); q += 4 * w
        if all(c in planes for c in "RGB"):
            return np.ascontiguousarray(np.stack([planes["R"], planes["G"], planes["B"]], -1))
        if "Y" in planes:
            return np.ascontiguousarray(np.repeat(planes["Y"][..., None], 3, axis=2))
        raise ValueError("read_exr(): no R, G, B or Y channels")
    from PIL import Image
    im = Image.open(path)
    im = im.convert("RGB")
    return np.ascontiguousarray(_srgb_lut()[np.asarray(im)])


def _texture_of(tp, base_dir):
    """src/textures/checkerboard.cpp:55-62, src/textures/bitmap.cpp:113-262 (RGB variants)"""
    m = tp["to_uv"][1][0] if "to_uv" in tp else _ident()
    tp.queried.add("to_uv")
    # Transform4f::extract() (transform.h:340-360): the upper-left 2x2 block (and the bottom row) -- the translation column is not copied
    tex = dict(to_uv=np.array([m[0], m[1], m[4], m[5]], np.float64).astype(F32), filter=1, wrap=0, channels=3, width=0, height=0,
               color0=np.zeros(3, F32), color1=np.zeros(3, F32), data=None)
    if tp.plugin == "checkerboard":
        for c in tp.children:
            if c[0] == "texture" or c[0] == "ref":
                raise ValueError("checkerboard: nested textures are not supported (constant colours only)")
        tex.update(kind=0, color0=_color(tp, "color0", 0.4), color1=_color(tp, "color1", 0.2))
        third = F32(1.0 / 3.0)
        m0 = ((tex["color0"][0] + tex["color0"][1]) + tex["color0"][2]) * third
        m1 = ((tex["color1"][0] + tex["color1"][1]) + tex["color1"][2]) * third
        tex["mean"] = F32(0.5) * (m0 + m1)
    elif tp.plugin == "bitmap":
        fn = tp.get_s("filename", "")
        if not fn:
            raise ValueError('Property "filename" has not been specified!')
        path = resolve_path(fn)
        ft, wm = tp.get_s("filter_type", "bilinear"), tp.get_s("wrap_mode", "repeat")
        if ft not in ("nearest", "bilinear"):
            raise ValueError('Invalid filter type "%s", must be one of: "nearest", or "bilinear"!' % ft)
        if wm not in ("repeat", "mirror", "clamp"):
            raise ValueError('Invalid wrap mode "%s", must be one of: "repeat", "mirror", or "clamp"!' % wm)
        raw = tp.get_b("raw", False)
        tp.get_b("accel", True)
        from PIL import Image      # the oracle's own decoder (the product reads PNG chunks itself, over zlib)
        im = Image.open(path)
        if im.mode in ("P", "RGBA", "CMYK"):
            im = im.convert("RGB")
        elif im.mode in ("LA", "1"):
            im = im.convert("L")
        if im.mode == "RGB":
            a = np.asarray(im, np.uint8)
        elif im.mode == "L":
            a = np.asarray(im, np.uint8)[..., None]
        else:
            raise ValueError('bitmap: unsupported pixel layout "%s" in "%s" (8-bit gray / RGB[A] / palette PNG)' % (im.mode, fn))
        data = (a.astype(F32) * F32(1.0 / 255.0)) if raw else _srgb_lut()[a]
        if data.shape[0] < 2 or data.shape[1] < 2:
            raise ValueError("bitmap: the image must be at least 2x2 pixels in size")
        if data.shape[2] == 3:     # m_mean: luminance, accumulated in double (bitmap.cpp:221-262)
            lum = data[..., 0] * F32(0.212671) + data[..., 1] * F32(0.715160) + data[..., 2] * F32(0.072169)
            mean = F32(float(np.sum(lum.astype(np.float64))) / lum.size)
        else:
            mean = F32(float(np.sum(data.astype(np.float64))) / data[..., 0].size)
        tex.update(kind=1, filter=int(ft == "bilinear"), wrap=("repeat", "mirror", "clamp").index(wm), channels=int(data.shape[2]),
                   width=int(data.shape[1]), height=int(data.shape[0]), data=np.ascontiguousarray(data, F32), mean=mean)
    else:
        raise ValueError('unsupported texture plugin "%s" (supported: bitmap, checkerboard)' % tp.plugin)
    unq = tp.unqueried() if hasattr(tp, "unqueried") else []
    return tex


def _reflectance(props, name, default, registry, base_dir):
    """(constant colour, texture record or None) of a BSDF's reflectance-like property"""
    for tag, child, cname in props.children:
        if cname != name:
            continue
        if tag == "ref":
            tag, child = registry[child]
        if tag != "texture":
            raise ValueError('property "%s" must be a colour or a texture' % name)
        tex = _texture_of(child, base_dir)
        return np.asarray([tex["mean"]] * 3, F32), tex
    return _color(props, name, default), None


def _bsdf_of(props, registry, base_dir=""):
    """BSDF record of a diffuse / conductor / dielectric BSDF, optionally inside twosided{...}"""
    if props.plugin == "mask":   # src/bsdfs/mask.cpp:93-117: one nested BSDF seen through an opacity (float or texture, default 0.5)
        inner = [c for c in props.children if c[0] == "bsdf" or (c[0] == "ref" and registry[c[1]][0] == "bsdf")]
        if len(inner) > 1:
            raise ValueError("Cannot specify more than one child BSDF")
        if not inner:
            raise ValueError("Child BSDF not specified")
        ip = inner[0][1] if inner[0][0] == "bsdf" else registry[inner[0][1]][1]
        if ip.plugin == "mask":
            raise ValueError("mask: a mask nested in a mask is not supported")
        rec = _bsdf_of(ip, registry, base_dir)
        if "opacity" in props and props["opacity"][0] == "rgb":
            raise ValueError('mask: an rgb "opacity" is not supported (give a float or a texture)')
        tex = _slot_texture(props, "opacity", registry, base_dir)
        rec["masked"] = 1
        rec["tex_opacity"] = tex
        rec["opacity"] = F32(tex["mean"]) if tex is not None else F32(props.get_f("opacity", 0.5))
        props.check_unreferenced("bsdf", ())
        return rec
    if props.plugin == "blendbsdf":   # src/bsdfs/blendbsdf.cpp:80-104: two nested BSDFs and a weight (float or texture, no default)
        inner = [c for c in props.children if c[0] == "bsdf" or (c[0] == "ref" and registry[c[1]][0] == "bsdf")]
        if len(inner) > 2:
            raise ValueError("BlendBSDF: Cannot specify more than two child BSDFs")
        tex = _slot_texture(props, "weight", registry, base_dir)
        if tex is None and "weight" not in props:
            raise ValueError('Property "weight" has not been specified!')
        if tex is None and props["weight"][0] == "rgb":
            raise ValueError('blendbsdf: an rgb "weight" is not supported (give a float or a texture)')
        if len(inner) != 2:
            raise ValueError("BlendBSDF: Two child BSDFs must be specified!")
        recs = []
        for c in inner:
            ip = c[1] if c[0] == "bsdf" else registry[c[1]][1]
            r = _bsdf_of(ip, registry, base_dir)
            if r.get("masked") or r.get("blend_other") is not None:
                raise ValueError('blendbsdf: a "%s" nested in a blendbsdf is not supported in this build' % ip.plugin)
            recs.append(r)
        rec = recs[0]
        rec["blend_other"] = recs[1]
        rec["tex_blend"] = tex
        rec["blend_weight"] = F32(tex["mean"]) if tex is not None else F32(props.get_f("weight", 0.5))
        props.check_unreferenced("bsdf", ())
        return rec
    if props.plugin == "bumpmap":   # src/bsdfs/bumpmap.cpp:84-112: one nested BSDF in the frame the gradient of ONE height texture (any property name) gives
        inner = [c for c in props.children if c[0] == "bsdf" or (c[0] == "ref" and registry[c[1]][0] == "bsdf")]
        texs = [c for c in props.children if c[0] == "texture" or (c[0] == "ref" and registry[c[1]][0] == "texture")]
        if len(inner) > 1:
            raise ValueError("Only a single BSDF child object can be specified.")
        if len(texs) > 1:
            raise ValueError("Only a single Texture child object can be specified.")
        if not inner:
            raise ValueError("Exactly one BSDF child object must be specified.")
        if not texs:
            raise ValueError("Exactly one Texture child object must be specified.")
        ip = inner[0][1] if inner[0][0] == "bsdf" else registry[inner[0][1]][1]
        if ip.plugin in ("twosided", "mask", "normalmap", "bumpmap", "blendbsdf"):
            raise ValueError('bumpmap: a "%s" nested in a bumpmap is not supported in this build (nest the bumpmap inside it instead)' % ip.plugin)
        tp = texs[0][1] if texs[0][0] == "texture" else registry[texs[0][1]][1]
        if tp.plugin != "bitmap":
            raise ValueError('bumpmap: the height texture must be a bitmap ("%s" has no eval_1_grad)' % tp.plugin)
        rec = _bsdf_of(ip, registry, base_dir)
        rec["tex_normal"] = _texture_of(tp, base_dir)
        rec["bumpmap"] = 1
        rec["bump_scale"] = F32(props.get_f("scale", 1.0))
        props.check_unreferenced("bsdf", ())
        return rec
    if props.plugin == "normalmap":   # src/bsdfs/normalmap.cpp:84-108: one nested BSDF evaluated in the frame an RGB texture gives
        inner = [c for c in props.children if c[0] == "bsdf" or (c[0] == "ref" and registry[c[1]][0] == "bsdf")]
        if len(inner) > 1:
            raise ValueError("Only a single BSDF child object can be specified.")
        if not inner:
            raise ValueError("Exactly one BSDF child object must be specified.")
        ip = inner[0][1] if inner[0][0] == "bsdf" else registry[inner[0][1]][1]
        if ip.plugin in ("twosided", "mask", "normalmap", "bumpmap", "blendbsdf"):
            raise ValueError('normalmap: a "%s" nested in a normalmap is not supported in this build (nest the normalmap inside it instead)' % ip.plugin)
        rec = _bsdf_of(ip, registry, base_dir)
        tex = _slot_texture(props, "normalmap", registry, base_dir)
        if tex is None:
            raise ValueError('Property "normalmap" has not been specified!')
        if tex["kind"] == 1 and tex["channels"] != 3:
            raise ValueError("normalmap: the texture must have three channels")
        rec["tex_normal"] = tex
        props.check_unreferenced("bsdf", ())
        return rec
    if props.plugin == "twosided":
        inner = [c for c in props.children if c[0] == "bsdf" or (c[0] == "ref" and registry[c[1]][0] == "bsdf")]
        if len(inner) > 2:
            raise ValueError("At most two nested BSDFs can be specified!")
        if not inner:
            raise ValueError("A nested one-sided material is required!")
        ip = inner[0][1] if inner[0][0] == "bsdf" else registry[inner[0][1]][1]
        rec = _bsdf_of(ip, registry, base_dir)
        if len(inner) == 2:   # twosided.cpp:75-86: the second BSDF is the back side's
            ip2 = inner[1][1] if inner[1][0] == "bsdf" else registry[inner[1][1]][1]
            back = _bsdf_of(ip2, registry, base_dir)
            if rec.get("blend_other") is not None or back.get("blend_other") is not None or rec.get("masked") or back.get("masked"):
                raise ValueError("twosided: a blendbsdf or mask as one of two nested BSDFs is not supported in this build")
            if rec["bsdf"] in (2, 6, 7, 8) or back["bsdf"] in (2, 6, 7, 8):
                raise ValueError("Only materials without a transmission component can be nested!")
            rec["twosided"] = back["twosided"] = 1
            rec["blend_other"], rec["two_bsdfs"] = back, 1
            return rec
        other = rec.get("blend_other")
        if rec["bsdf"] in (2, 6, 7, 8) or rec.get("masked") or (other is not None and other["bsdf"] in (2, 6, 7, 8)):   # twosided.cpp:47-52
            raise ValueError("Only materials without a transmission component can be nested!")
        rec["twosided"] = 1
        if other is not None:   # twosided{ blendbsdf{ a, b } } flips wi / wo before either nested BSDF sees them: the same as blendbsdf{ twosided{a}, twosided{b} }
            other["twosided"] = 1
        return rec
    rec = dict(twosided=0, bsdf=0, reflectance=np.array([0.5] * 3, F32), cond_eta=np.zeros(3, F32), cond_k=np.ones(3, F32),
               spec_refl=np.ones(3, F32), spec_trans=np.ones(3, F32), diel_eta=F32(1.0), nonlinear=0, alpha_u=F32(0.1), alpha_v=F32(0.1))
    if props.plugin == "diffuse":
        rec["reflectance"], rec["tex_refl"] = _reflectance(props, "reflectance", 0.5, registry, base_dir)
    elif props.plugin == "conductor":   # src/bsdfs/conductor.cpp:171-188
        material = props.get_s("material", "none")
        if material != "none":
            raise ValueError("Should specify either (eta, k) or material, not both." if "eta" in props else
                             'conductor: named materials need the spectral IOR data files, which this build does not ship; give "eta" and "k"')
        rec.update(bsdf=1, cond_eta=_color(props, "eta", 0.0), cond_k=_color(props, "k", 1.0), spec_refl=_color(props, "specular_reflectance", 1.0))
    elif props.plugin == "dielectric":  # src/bsdfs/dielectric.cpp:176-203
        int_ior, ext_ior = _lookup_ior(props, "int_ior", "bk7"), _lookup_ior(props, "ext_ior", "air")
        if int_ior < 0 or ext_ior < 0:
            raise ValueError("The interior and exterior indices of refraction must be positive!")
        rec.update(bsdf=2, diel_eta=F32(int_ior / ext_ior), spec_refl=_color(props, "specular_reflectance", 1.0),
                   spec_trans=_color(props, "specular_transmittance", 1.0))
    elif props.plugin == "null":   # src/bsdfs/null.cpp:36-40: no parameters
        rec.update(bsdf=8)
    elif props.plugin == "thindielectric":   # src/bsdfs/thindielectric.cpp:137-158
        int_ior, ext_ior = _lookup_ior(props, "int_ior", "bk7"), _lookup_ior(props, "ext_ior", "air")
        if int_ior < 0 or ext_ior < 0:
            raise ValueError("The interior and exterior indices of refraction must be positive!")
        rec.update(bsdf=6, diel_eta=F32(int_ior / ext_ior), spec_refl=_color(props, "specular_reflectance", 1.0),
                   spec_trans=_color(props, "specular_transmittance", 1.0))
    elif props.plugin == "roughdielectric":   # src/bsdfs/roughdielectric.cpp:163-238
        int_ior, ext_ior = _lookup_ior(props, "int_ior", "bk7"), _lookup_ior(props, "ext_ior", "air")
        if int_ior < 0 or ext_ior < 0 or int_ior == ext_ior:
            raise ValueError("The interior and exterior indices of refraction must be positive and differ!")
        rec.update(bsdf=7, diel_eta=F32(int_ior / ext_ior), spec_refl=_color(props, "specular_reflectance", 1.0),
                   spec_trans=_color(props, "specular_transmittance", 1.0))
        distr = props.get_s("distribution", "beckmann").lower()
        if distr not in ("beckmann", "ggx"):
            raise ValueError('Specified an invalid distribution "%s", must be "beckmann" or "ggx"!' % distr)
        rec.update(mf_type=int(distr == "ggx"))   # MicrofacetType: 0 beckmann, 1 ggx
        rec.update(sample_all=int(not props.get_b("sample_visible", True)))
        if "alpha_u" in props or "alpha_v" in props:
            if not ("alpha_u" in props and "alpha_v" in props):
                raise ValueError("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.")
            if "alpha" in props:
                raise ValueError("Microfacet model: please specifyeither 'alpha' or 'alpha_u'/'alpha_v'.")
            rec.update(alpha_u=F32(props.get_f("alpha_u", 0.1)), alpha_v=F32(props.get_f("alpha_v", 0.1)))
        else:
            a = F32(props.get_f("alpha", 0.1))
            rec.update(alpha_u=a, alpha_v=a)
    elif props.plugin == "roughconductor":   # src/bsdfs/roughconductor.cpp:177-227
        material = props.get_s("material", "none")
        if material != "none":
            raise ValueError("Should specify either (eta, k) or material, not both." if "eta" in props else
                             'roughconductor: named materials need the spectral IOR data files, which this build does not ship; give "eta" and "k"')
        distr = props.get_s("distribution", "beckmann").lower()
        if distr not in ("beckmann", "ggx"):
            raise ValueError('Specified an invalid distribution "%s", must be "beckmann" or "ggx"!' % distr)
        rec.update(mf_type=int(distr == "ggx"))   # MicrofacetType: 0 beckmann, 1 ggx
        rec.update(sample_all=int(not props.get_b("sample_visible", True)))
        if "alpha_u" in props or "alpha_v" in props:
            if not ("alpha_u" in props and "alpha_v" in props):
                raise ValueError("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.")
            if "alpha" in props:
                raise ValueError("Microfacet model: please specifyeither 'alpha' or 'alpha_u'/'alpha_v'.")
            au, av = F32(props.get_f("alpha_u", 0.1)), F32(props.get_f("alpha_v", 0.1))
        else:
            au = av = F32(props.get_f("alpha", 0.1))
        rec.update(bsdf=4, cond_eta=_color(props, "eta", 0.0), cond_k=_color(props, "k", 1.0), spec_refl=_color(props, "specular_reflectance", 1.0),
                   alpha_u=au, alpha_v=av)
    elif props.plugin == "plastic":     # src/bsdfs/plastic.cpp:167-199
        int_ior, ext_ior = _lookup_ior(props, "int_ior", "polypropylene"), _lookup_ior(props, "ext_ior", "air")
        if int_ior < 0 or ext_ior < 0:
            raise ValueError("The interior and exterior indices of refraction must be positive!")
        refl, rec["tex_refl"] = _reflectance(props, "diffuse_reflectance", 0.5, registry, base_dir)
        rec.update(bsdf=3, diel_eta=F32(int_ior / ext_ior), reflectance=refl,
                   spec_refl=_color(props, "specular_reflectance", 1.0), nonlinear=int(props.get_b("nonlinear", False)))
    elif props.plugin == "roughplastic":   # src/bsdfs/roughplastic.cpp:170-220
        int_ior, ext_ior = _lookup_ior(props, "int_ior", "polypropylene"), _lookup_ior(props, "ext_ior", "air")
        if int_ior < 0 or ext_ior < 0 or int_ior == ext_ior:
            raise ValueError("The interior and exterior indices of refraction must be positive and differ!")
        refl, rec["tex_refl"] = _reflectance(props, "diffuse_reflectance", 0.5, registry, base_dir)
        rec.update(bsdf=5, diel_eta=F32(int_ior / ext_ior), reflectance=refl,
                   has_spec_refl=int("specular_reflectance" in props), spec_refl=_color(props, "specular_reflectance", 1.0),
                   nonlinear=int(props.get_b("nonlinear", False)))
        distr = props.get_s("distribution", "beckmann").lower()
        if distr not in ("beckmann", "ggx"):
            raise ValueError('Specified an invalid distribution "%s", must be "beckmann" or "ggx"!' % distr)
        rec.update(mf_type=int(distr == "ggx"))   # MicrofacetType: 0 beckmann, 1 ggx
        rec.update(sample_all=int(not props.get_b("sample_visible", True)))
        if "alpha_u" in props or "alpha_v" in props:
            if not ("alpha_u" in props and "alpha_v" in props):
                raise ValueError("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.")
            if "alpha" in props:
                raise ValueError("Microfacet model: please specifyeither 'alpha' or 'alpha_u'/'alpha_v'.")
            au, av = F32(props.get_f("alpha_u", 0.1)), F32(props.get_f("alpha_v", 0.1))
            if au != av:
                raise ValueError("The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!")
        else:
            au = av = F32(props.get_f("alpha", 0.1))
        rec.update(alpha_u=au, alpha_v=av)
    else:
        raise ValueError('unsupported BSDF plugin "%s"' % props.plugin)
    props.check_unreferenced("bsdf", ("reflectance", "diffuse_reflectance", "specular_reflectance", "specular_transmittance", "eta", "k"))
    # textures on the specular colours (Texture::eval per hit; the constants become the texture's mean, which is what the plastics' sampling weight uses,
    # plastic.cpp:201-217, roughplastic.cpp:243-257) and on the roughness of roughconductor / roughdielectric (Texture::eval_1 per hit)
    slots = {"specular_reflectance": ("tex_spec", "spec_refl", (1, 2, 3, 4, 5, 6, 7)), "specular_transmittance": ("tex_trans", "spec_trans", (2, 6, 7))}
    for name, (key, const, kinds) in slots.items():
        tex = _slot_texture(props, name, registry, base_dir)
        if tex is not None:
            if rec["bsdf"] not in kinds:
                raise ValueError('property "%s" of plugin "%s" does not accept a texture' % (name, props.plugin))
            rec[key] = tex; rec[const] = np.asarray([tex["mean"]] * 3, F32); rec[const + "_mean"] = F32(tex["mean"])
            if name == "specular_reflectance":
                rec["has_spec_refl"] = 1
    if rec["bsdf"] in (4, 7):
        both = _slot_texture(props, "alpha", registry, base_dir)
        tu, tv = _slot_texture(props, "alpha_u", registry, base_dir), _slot_texture(props, "alpha_v", registry, base_dir)
        if both is not None and (tu is not None or tv is not None or "alpha_u" in props or "alpha_v" in props):
            raise ValueError("Microfacet model: please specifyeither 'alpha' or 'alpha_u'/'alpha_v'.")
        if (tu is None) != (tv is None) and not ("alpha_u" in props or "alpha_v" in props):
            raise ValueError("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.")
        if both is not None:
            tu = tv = both
        if tu is not None:
            rec["tex_alpha_u"] = tu; rec["alpha_u"] = F32(tu["mean"])
        if tv is not None:
            rec["tex_alpha_v"] = tv; rec["alpha_v"] = F32(tv["mean"])
    return rec


def _slot_texture(props, name, registry, base_dir):
    for tag, child, cname in props.children:
        if cname != name:
            continue
        if tag == "ref":
            tag, child = registry[child]
        if tag != "texture":
            raise ValueError('property "%s" must be a colour or a texture' % name)
        return _texture_of(child, base_dir)
    return None


_ATTACHED = object()   # registry key of the set of emitters already attached to a shape


class FlatScene:
    def __init__(self):
        self.shapes, self.groups, self.objects, self.emitters = [], [], [], []
        self.sensor = None
        self.integrator = None   # Props
        self.sampler = None      # Props


def _no_colours(props, kind):
    for n, (t, _v) in props.items():
        if t == "rgb":
            raise ValueError('unreferenced object "%s" (within %s of type "%s")' % (n, kind, props.plugin))


def _shape_record(sp, registry, strip_to_world, base_dir=""):
    _no_colours(sp, "shape")
    kind = {"rectangle": 0, "cube": 1, "obj": 1, "ply": 1, "serialized": 1, "sphere": 2, "disk": 3, "cylinder": 4}.get(sp.plugin)
    if kind is None:
        raise ValueError('unsupported shape plugin "%s"' % sp.plugin)
    mesh_raw = None
    if sp.plugin in ("obj", "ply", "serialized"):   # src/shapes/obj.cpp:139-143, ply.cpp:160-166, serialized.cpp:242-244: filename through the file resolver
        from . import mesh_io
        fn = sp.get_s("filename", None)
        if fn is None:
            raise ValueError('Property "filename" has not been specified!')
        path = resolve_path(fn)
        fnorm = sp.get_b("face_normals", False)
        mesh_raw = (mesh_io.read_obj(path, sp.get_b("flip_tex_coords", True), fnorm) if sp.plugin == "obj"
                    else mesh_io.read_ply(path, fnorm) if sp.plugin == "ply"
                    else mesh_io.read_serialized(path, sp.get_i("shape_index", 0), fnorm))
    tw, tinv = _ident(), _ident()
    if not strip_to_world and "to_world" in sp and sp["to_world"][0] == "transform":
        tw, tinv = sp["to_world"][1]
    flip = sp.get_b("flip_normals", False)
    if kind in (0, 3) and flip:   # rectangle.cpp:91-99, disk.cpp:91-95: to_world * scale(1, 1, -1)
        fm, fi = _scale([1.0, 1.0, -1.0])
        tw, tinv = _mul(tw, fm), _mul(fi, tinv)
        flip = False
    bsdfs = [c for c in sp.children if c[0] == "bsdf" or (c[0] == "ref" and registry[c[1]][0] == "bsdf")]
    ems = [c[1] if c[0] == "emitter" else registry[c[1]][1] for c in sp.children if c[0] == "emitter" or (c[0] == "ref" and registry[c[1]][0] == "emitter")]
    emitter, radiance, tex_radiance = 0, np.zeros(3, F32), None
    if ems:   # src/emitters/area.cpp:64-76 on a static shape (rectangle or triangle mesh)
        if len(ems) > 1:
            raise ValueError("Only a single Emitter child object can be specified per shape.")
        if ems[0].plugin != "area":
            raise ValueError('unsupported emitter plugin "%s" inside a shape (supported: area)' % ems[0].plugin)
        attached = registry.setdefault(_ATTACHED, set())   # an emitter declared at scene level can be referenced by ONE shape (endpoint.cpp:36-40)
        if id(ems[0]) in attached:
            raise ValueError("An endpoint can be only be attached to a single shape.")
        attached.add(id(ems[0]))
        if strip_to_world:
            raise ValueError("Instancing of emitters is not supported")   # shapegroup.cpp:27-28 (an animated shape becomes an instance, xml.cpp:1165-1195)
        if "to_world" in ems[0]:
            raise ValueError("Found a 'to_world' transformation -- this is not allowed.")
        tex_radiance = _slot_texture(ems[0], "radiance", registry, base_dir)   # area.cpp:73: a texture makes the emitter spatially varying
        if tex_radiance is not None:
            if kind != 0:
                raise ValueError("area emitter: a textured radiance is supported on rectangles only")
            radiance = np.asarray([tex_radiance["mean"]] * 3, F32)
        else:
            rad = ems[0]["radiance"] if "radiance" in ems[0] else ("float", 1.0)
            radiance = np.asarray([rad[1]] * 3 if rad[0] in ("float", "int") else rad[1], dtype=np.float64).astype(F32)
        emitter = 1
    if bsdfs:
        bp = bsdfs[0][1] if bsdfs[0][0] == "bsdf" else registry[bsdfs[0][1]][1]
        brec = _bsdf_of(bp, registry, base_dir)
    else:   # shape.cpp:66-72: default diffuse, reflectance 0 when the shape is an emitter
        brec = dict(twosided=0, bsdf=0, reflectance=np.array([0.0 if emitter else 0.5] * 3, dtype=F32), cond_eta=np.zeros(3, F32),
                    cond_k=np.ones(3, F32), spec_refl=np.ones(3, F32), spec_trans=np.ones(3, F32), diel_eta=F32(1.0), nonlinear=0)
    twosided, refl = brec["twosided"], brec["reflectance"]
    sphere = None
    if kind == 2:   # src/shapes/sphere.cpp:121-131: center (point, default 0) and radius (default 1) on top of to_world
        c = sp["center"][1] if "center" in sp else [0.0, 0.0, 0.0]
        sp.queried.add("center")
        sphere = dict(center=np.asarray(c, dtype=np.float64).astype(F32), radius=F32(sp.get_f("radius", 1.0)))
    cylinder = None
    if kind == 4:   # src/shapes/cylinder.cpp:100-118: p0 (default 0), p1 (default (0, 0, 1)), radius (default 1) on top of to_world
        p0 = sp["p0"][1] if "p0" in sp else [0.0, 0.0, 0.0]
        p1 = sp["p1"][1] if "p1" in sp else [0.0, 0.0, 1.0]
        sp.queried.add("p0"); sp.queried.add("p1")
        cylinder = dict(p0=np.asarray(p0, dtype=np.float64).astype(F32), p1=np.asarray(p1, dtype=np.float64).astype(F32), radius=F32(sp.get_f("radius", 1.0)))
        if emitter:
            raise ValueError("cylinder: area emitters on cylinders are not supported")
    return dict(kind=kind, twosided=twosided, flip_normals=int(flip), face_normals=int(sp.get_b("face_normals", False)), cylinder=cylinder,
                reflectance=refl, to_world=_m32(tw), to_object=_m32(tinv), emitter=emitter, radiance=radiance, mesh_raw=mesh_raw,
                sphere=sphere, bsdf=brec["bsdf"], cond_eta=brec["cond_eta"], cond_k=brec["cond_k"], spec_refl=brec["spec_refl"],
                spec_trans=brec["spec_trans"], diel_eta=brec["diel_eta"], nonlinear=brec.get("nonlinear", 0),
                alpha_u=brec.get("alpha_u", F32(0.1)), alpha_v=brec.get("alpha_v", F32(0.1)), has_spec_refl=brec.get("has_spec_refl", 0),
                mf_type=brec.get("mf_type", 1), sample_all=brec.get("sample_all", 0), tex_refl=brec.get("tex_refl"),
                tex_spec=brec.get("tex_spec"), tex_trans=brec.get("tex_trans"), tex_alpha_u=brec.get("tex_alpha_u"), tex_alpha_v=brec.get("tex_alpha_v"),
                masked=brec.get("masked", 0), opacity=brec.get("opacity", F32(1.0)), tex_opacity=brec.get("tex_opacity"), tex_normal=brec.get("tex_normal"), bumpmap=brec.get("bumpmap", 0), bump_scale=brec.get("bump_scale", F32(1.0)), tex_radiance=tex_radiance,
                blend_other=brec.get("blend_other"), blend_weight=brec.get("blend_weight", F32(0.5)), tex_blend=brec.get("tex_blend"), two_bsdfs=brec.get("two_bsdfs", 0),
                spec_refl_mean=brec.get("spec_refl_mean"))


def load(source, params=None, is_string=False):
    """Parse a scene XML (file path or string) into a FlatScene."""
    _POS.clear()
    root = _parse_text(source if is_string else open(source).read())
    base_dir = "" if is_string else os.path.dirname(os.path.abspath(source))
    _check_tree(root, None, 0, {})
    _substitute(root, params or {}, base_dir)
    registry = {}
    scene_root = root.tag == "scene"
    if scene_root:
        top = _parse_object(root, registry)
    else:      # any object may be the root (xml.cpp:489-490); it is instantiated like a scene's child, but only scenes can be rendered
        top = Props("scene")
        top.children.append((root.tag, _parse_object(root, registry), None))
    fs = FlatScene()
    group_of = {}   # id(props of shapegroup) -> group index
    for tag, child, _name in top.children:
        if tag == "ref":
            if child not in registry:
                raise ValueError('reference to unknown object "%s"!' % child)
            tag, child = registry[child]
        if tag == "integrator":
            fs.integrator = child
        elif tag == "sensor":
            fs.sensor = _sensor_record(child)
            fs.sampler = next((c[1] for c in child.children if c[0] == "sampler"), None)
        elif tag == "emitter" and child.plugin == "area":
            pass   # declared at scene level, attached by the shape that references it (scene.cpp:44-47 skips surface emitters among the scene's children)
        elif tag == "emitter":
            if child.plugin == "directional":   # src/emitters/directional.cpp:65-91
                if "direction" in child:
                    if "to_world" in child:
                        raise ValueError("Only one of the parameters 'direction' and 'to_world' can be specified at the same time!'")
                    v = np.asarray(child["direction"][1], dtype=np.float64).astype(F32)
                    for _ in range(2):   # dr::normalize of the property, then look_at normalises target - origin once more (both in float32)
                        v = (v * (F32(1.0) / np.sqrt(F32(v[0] * v[0]) + F32(v[1] * v[1]) + F32(v[2] * v[2]), dtype=F32))).astype(F32)
                    d = v
                else:
                    tw = child["to_world"][1][0] if "to_world" in child else _ident()
                    d = _m32(tw)[:3, 2].copy()     # to_world * (0, 0, 1)
                child.queried.add("to_world"); child.queried.add("direction")
                irr = child["irradiance"] if "irradiance" in child else ("float", 1.0)
                child.queried.add("irradiance")
                iv = [irr[1]] * 3 if irr[0] in ("float", "int") else irr[1]
                fs.emitters.append(dict(kind=5, position=np.asarray(d, F32), intensity=np.asarray(iv, dtype=np.float64).astype(F32)))
                continue
            if child.plugin == "spot":   # src/emitters/spot.cpp:75-100
                tw, tinv = child["to_world"][1] if "to_world" in child else (_ident(), _ident())
                cutoff = F32(child.get_f("cutoff_angle", 20.0))
                beam = F32(child.get_f("beam_width", float(cutoff * F32(3.0) / F32(4.0))))
                if not (np.isfinite(cutoff) and np.isfinite(beam) and abs(cutoff) <= 360 and abs(beam) <= 360):
                    raise ValueError("spot: cutoff_angle and beam_width must be finite angles in degrees")
                if "texture" in child:
                    raise ValueError("spot: textured spot lights are not supported")
                if not cutoff >= beam:
                    raise ValueError("spot: cutoff_angle must not be smaller than beam_width")
                inten = child["intensity"] if "intensity" in child else ("float", 1.0)
                child.queried.add("intensity"); child.queried.add("to_world")
                iv = [inten[1]] * 3 if inten[0] in ("float", "int") else inten[1]
                fs.emitters.append(dict(kind=2, position=_m32(tw)[:3, 3].copy(), intensity=np.asarray(iv, dtype=np.float64).astype(F32),
                                        to_local=_m32(tinv), cutoff_deg=cutoff, beam_deg=beam))
                continue
            if child.plugin == "constant":   # src/emitters/constant.cpp:58-67; the environment of the scene (scene.cpp:53-57)
                if any(e["kind"] in (3, 4) for e in fs.emitters):
                    raise ValueError("Only one environment emitter can be specified per scene.")
                rad = child["radiance"] if "radiance" in child else ("float", 1.0)
                child.queried.add("radiance")
                rv = [rad[1]] * 3 if rad[0] in ("float", "int") else rad[1]
                fs.emitters.append(dict(kind=3, position=np.zeros(3, F32), intensity=np.asarray(rv, dtype=np.float64).astype(F32)))
                continue
            if child.plugin == "envmap":     # src/emitters/envmap.cpp:116-224
                if any(e["kind"] in (3, 4) for e in fs.emitters):
                    raise ValueError("Only one environment emitter can be specified per scene.")
                fn = child.get_s("filename", None)
                if fn is None:
                    raise ValueError('Property "filename" has not been specified!')
                img = read_radiance_image(resolve_path(fn))
                if img.shape[1] < 2 or img.shape[0] < 3:
                    raise ValueError('"%s": the environment map resolution must be at least 2x3 pixels' % os.path.basename(fn))
                tw, tinv = child["to_world"][1] if "to_world" in child and child["to_world"][0] == "transform" else (_ident(), _ident())
                child.queried.add("to_world")
                fs.emitters.append(dict(kind=4, position=np.zeros(3, F32), intensity=np.zeros(3, F32), image=img, scale=F32(child.get_f("scale", 1.0)),
                                        to_world=_m32(tw), to_local=_m32(tinv), mis_compensation=child.get_b("mis_compensation", False)))
                continue
            if child.plugin != "point":
                raise ValueError('unsupported emitter plugin "%s"' % child.plugin)
            if "position" in child:
                pos = np.asarray(child["position"][1], dtype=np.float64).astype(F32)
            else:
                tw = child["to_world"][1][0] if "to_world" in child else _ident()
                pos = _m32(tw)[:3, 3]
            inten = child["intensity"] if "intensity" in child else ("float", 1.0)
            iv = [inten[1]] * 3 if inten[0] in ("float", "int") else inten[1]
            fs.emitters.append(dict(kind=0, position=pos, intensity=np.asarray(iv, dtype=np.float64).astype(F32)))
        if tag == "emitter":
            for _n, (_t, _v) in child.items():
                if _t == "rgb" and _n not in ("intensity", "radiance", "irradiance"):
                    raise ValueError('unreferenced object "%s" (within emitter of type "%s")' % (_n, child.plugin))
        if tag == "shape":
            if child.plugin == "shapegroup":
                first = len(fs.shapes)
                for t2, c2, _n in child.children:
                    if t2 == "ref":
                        t2, c2 = registry[c2]
                    if t2 != "shape":
                        raise ValueError("Tried to add an unsupported object to a shapegroup")
                    # shapegroup.cpp:17-36: what a group refuses
                    if c2.plugin == "instance":
                        raise ValueError("Nested instancing is not permitted")
                    if c2.plugin == "shapegroup":
                        raise ValueError("Nested ShapeGroup is not permitted")
                    if any(c3[0] == "sensor" for c3 in c2.children):
                        raise ValueError("Instancing of sensors is not supported")
                    fs.shapes.append(_shape_record(c2, registry, False, base_dir))
                    if fs.shapes[-1]["emitter"]:
                        raise ValueError("Instancing of emitters is not supported")
                group_of[id(child)] = len(fs.groups)
                fs.groups.append(dict(first_shape=first, n_shapes=len(fs.shapes) - first))
            elif child.plugin == "instance":
                grp = None
                for t2, c2, _n in child.children:
                    if t2 == "ref":
                        t2, c2 = registry[c2]
                    if t2 == "shape" and c2.plugin == "shapegroup":
                        grp = group_of[id(c2)]
                if grp is None:
                    raise ValueError("A reference to a 'shapegroup' must be specified!")
                fs.objects.append(_instance_record(child.get("to_world"), grp))
            elif "to_world" in child and child["to_world"][0] == "animation":
                # xml.cpp:1165-1195: shape{animated to_world} -> shapegroup{shape} + instance
                first = len(fs.shapes)
                fs.shapes.append(_shape_record(child, registry, True, base_dir))
                fs.groups.append(dict(first_shape=first, n_shapes=1))
                fs.objects.append(_instance_record(child["to_world"], len(fs.groups) - 1))
            else:
                fs.shapes.append(_shape_record(child, registry, False, base_dir))
                fs.objects.append(dict(kind=0, index=len(fs.shapes) - 1, n_keys=0,
                                       key_time=np.zeros(2, F32), key=np.zeros((2, 4, 4), F32)))
                if fs.shapes[-1]["emitter"]:   # scene.cpp:33-35: a shape's emitter joins the list at the shape's position
                    fs.emitters.append(dict(kind=1, position=np.zeros(3, F32), intensity=fs.shapes[-1]["radiance"],
                                            shape=len(fs.shapes) - 1))
        elif tag == "bsdf":
            _bsdf_of(child, registry, base_dir)      # every object is instantiated: a malformed declaration fails even if nothing refers to it
        elif tag in ("sampler", "film", "rfilter"):
            raise ValueError('unreferenced object "%s" (within scene of type "scene")' % child.plugin)
    top.check_unreferenced("scene", (), scalars_only=False)
    if not scene_root:
        raise ValueError('root element "%s": only <scene> descriptions can be rendered' % root.tag)
    return fs         # fs.sensor is None for a scene without a sensor: it loads, as in the reference; rendering it is the error


def _instance_record(tw, group):
    key = np.zeros((2, 4, 4), F32)
    kt = np.zeros(2, F32)
    if tw is None:
        key[0] = np.eye(4)
        n = 1
    elif tw[0] == "transform":
        key[0] = _m32(tw[1][0])
        n = 1
    else:
        keys = tw[1]
        n = min(len(keys), 2)   # AnimatedTransform::eval only looks at keyframes 0 and 1 (transform.h:458-466)
        for i in range(n):
            kt[i] = keys[i][0]
            key[i] = _m32(keys[i][1][0])
    return dict(kind=1, index=group, n_keys=n, key_time=kt, key=key)


def _parse_fov(sp, aspect):
    # src/render/sensor.cpp:149-203
    if "fov" in sp and "focal_length" in sp:
        raise ValueError("Please specify either a focal length ('focal_length') or a field of view ('fov')!")
    if "fov" in sp:
        fov = sp.get_f("fov", None)
        axis = sp.get_s("fov_axis", "x").lower()
        if axis == "smaller":
            axis = "y" if aspect > 1 else "x"
        elif axis == "larger":
            axis = "x" if aspect > 1 else "y"
    else:
        f = sp.get_s("focal_length", "50mm")
        if f.endswith("mm"):
            f = f[:-2]
        fov = 2.0 * math.degrees(math.atan(math.sqrt(36.0 * 36 + 24 * 24) / (2.0 * float(f))))
        axis = "diagonal"
    if axis == "x":
        r = fov
    elif axis == "y":
        r = math.degrees(2.0 * math.atan(math.tan(0.5 * math.radians(fov)) * aspect))
    elif axis == "diagonal":
        diag = 2.0 * math.tan(0.5 * math.radians(fov))
        width = diag / math.sqrt(1.0 + 1.0 / (aspect * aspect))
        r = math.degrees(2.0 * math.atan(width * 0.5))
    else:
        raise ValueError("The 'fov_axis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!")
    if r <= 0.0 or r >= 180.0:
        raise ValueError("The horizontal field of view must be in the range [0, 180]!")
    return r


def _sensor_record(sp):
    if sp.plugin not in ("perspective", "thinlens", "orthographic"):
        raise ValueError('unsupported sensor plugin "%s"' % sp.plugin)
    _no_colours(sp, "sensor")
    film = next((c[1] for c in sp.children if c[0] == "film"), None)
    w, h, cx, cy = 768, 576, 0, 0
    alpha = False
    filt, radius, stddev = None, 0.0, 0.5
    fb = fc = 1.0 / 3.0
    if film is not None:
        w, h = film.get_i("width", 768), film.get_i("height", 576)
        cw, ch = film.get_i("crop_width", w), film.get_i("crop_height", h)
        cx, cy = film.get_i("crop_offset_x", 0), film.get_i("crop_offset_y", 0)
        pf = film.get_s("pixel_format", "rgb").lower()   # hdrfilm.cpp:143-192: rgba sets FilmFlags::Alpha
        if pf in ("luminance", "luminance_alpha", "xyz", "xyza", "transient"):
            raise ValueError('unsupported pixel_format "%s" (supported: rgb, rgba)' % pf)
        if pf not in ("rgb", "rgba"):
            raise ValueError('The "pixel_format" parameter must either be equal to "luminance", "luminance_alpha", "rgb", "rgba",  "xyz", "xyza". Found %s.' % pf)
        alpha = pf == "rgba"
        rf = next((c[1] for c in film.children if c[0] == "rfilter"), None)
        if rf is not None:
            if rf.plugin == "tent":
                filt, radius = 1, rf.get_f("radius", 1.0)
            elif rf.plugin == "box":
                filt, radius = 0, 0.5
            elif rf.plugin == "gaussian":   # gaussian.cpp:48-53
                stddev = rf.get_f("stddev", 0.5)
                filt, radius = 2, float(F32(4) * F32(stddev))
            elif rf.plugin == "mitchell":   # mitchell.cpp:38-45
                filt, radius = 3, 2.0
                fb, fc = float(F32(rf.get_f("B", float(F32(1.0) / F32(3.0))))), float(F32(rf.get_f("C", float(F32(1.0) / F32(3.0)))))
            elif rf.plugin == "catmullrom":  # catmullrom.cpp:33-36
                filt, radius = 4, 2.0
            elif rf.plugin == "lanczos":     # lanczos.cpp:47-50: radius = lobes
                filt, radius = 5, float(rf.get_i("lobes", 3))
            else:
                raise ValueError('unsupported rfilter plugin "%s"' % rf.plugin)
    else:
        cw, ch = w, h
    if filt is None:   # film.cpp:49-53: gaussian by default
        filt, radius, stddev = 2, 2.0, 0.5
    tw = sp["to_world"][1][0] if "to_world" in sp else _ident()
    so = sp.get_f("shutter_open", 0.0)
    sc = sp.get_f("shutter_close", 0.0)
    near, far = sp.get_f("near_clip", 1e-2), sp.get_f("far_clip", 1e4)
    # perspective.cpp:143-144 / thinlens.cpp:149-150 (Transform::has_scale, transform.h:325-337)
    m3 = np.asarray(_m32(tw), np.float32).reshape(4, 4)[:3, :3]
    if sp.plugin != "orthographic" and np.any(np.abs(m3 @ m3.T - np.eye(3, dtype=np.float32)) > 1e-3):
        raise ValueError("Scale factors in the camera-to-world transformation are not allowed!")
    lens = dict(kind=2 if sp.plugin == "orthographic" else 0, aperture_radius=F32(0), focus_distance=F32(sp.get_f("focus_distance", float(F32(far)))))
    if sp.plugin == "thinlens":   # thinlens.cpp:138-156; focus_distance: sensor.cpp:134
        if "aperture_radius" not in sp:
            raise ValueError('Property "aperture_radius" has not been specified!')
        ar = F32(sp.get_f("aperture_radius", 0.0))
        if ar == 0:
            ar = F32(2.0 ** -24)   # dr::Epsilon<float>
        lens = dict(kind=1, aperture_radius=ar, focus_distance=F32(sp.get_f("focus_distance", float(F32(far)))))
    return dict(**lens, to_world=_m32(tw), x_fov=F32(0 if sp.plugin == "orthographic" else _parse_fov(sp, w / float(h))), near_clip=F32(near), far_clip=F32(far),
                shutter_open=F32(so), shutter_close=F32(sc), film_w=w, film_h=h, crop_x=cx, crop_y=cy,
                crop_w=cw, crop_h=ch, filter=filt, filter_radius=F32(radius), filter_stddev=F32(stddev), filter_b=F32(fb), filter_c=F32(fc), alpha=alpha)


# ----------------------------------------------------------------------------- plugin parameters
WAVE = {"sinusoidal": 0, "rectangular": 1, "triangular": 2, "trapezoidal": 3}
# periodic / regular: declared (sampler.h:27-34) and implemented (correlated.cpp:147-152) but never parsed by the reference's integrator (integrator.cpp:58-70)
TIME = {"uniform": 0, "stratified": 1, "antithetic": 2, "antithetic_mirror": 3, "periodic": 4, "regular": 5}


def integrator_params(ip, sp):
    """Restates the constructors (dopplertofpath.cpp:19-57, integrator.cpp:22-28,54-100,568-585,
    correlated.cpp:17-23, sampler.cpp:11-20) with their float32 roundings. ip/sp: Props or dict."""
    def as_props(x, plugin):
        if isinstance(x, Props):
            return x
        p = Props(plugin)
        for k, v in (x or {}).items():
            if k == "type":
                p.plugin = v
            elif isinstance(v, bool):
                p[k] = ("bool", v)
            elif isinstance(v, int):
                p[k] = ("int", v)
            elif isinstance(v, float):
                p[k] = ("float", v)
            else:
                p[k] = ("string", v)
        return p
    ip, sp = as_props(ip, "dopplertofpath"), as_props(sp, "correlated")
    kinds = {"dopplertofpath": 0, "path": 1, "velocity": 2}   # path / velocity: SURVEY 8(f) #1
    if ip.plugin not in kinds:
        raise ValueError('unsupported integrator plugin "%s"' % ip.plugin)
    samplers = {"correlated": 0, "independent": 1, "timestratified": 2}   # src/samplers/{correlated,independent,timestratified}.cpp
    if sp.plugin not in samplers:
        raise ValueError('unsupported sampler plugin "%s"' % sp.plugin)
    T = F32(ip.get_f("time", 0.0015))
    w_g = F32(ip.get_f("w_g", 30.0))
    g_1, g_0 = F32(ip.get_f("g_1", 0.5)), F32(ip.get_f("g_0", 0.5))
    w_s = F32(ip.get_f("w_s", 30.0))
    phase = F32(ip.get_f("sensor_phase_offset", 0.0))
    if "hetero_offset" in ip:
        phase = F32(np.float64(F32(ip.get_f("hetero_offset", 0.0)) * F32(2)) * math.pi)
    if "hetero_frequency" in ip:
        hf = F32(ip.get_f("hetero_frequency", 1.0))
        w_s = F32(np.float64(w_g) + np.float64(hf / T) * 1e-6)
    else:
        hf = F32(np.float64(w_s - w_g) * 1e6 * np.float64(T))
    wave = ip.get_s("wave_function_type", "sinusoidal")
    if wave not in WAVE:
        raise ValueError('unknown wave_function_type "%s"' % wave)   # documented deviation
    tsm = ip.get_s("time_sampling_method", "antithetic")
    if tsm not in TIME:
        raise ValueError('unknown time_sampling_method "%s"' % tsm)
    shift = F32(ip.get_f("antithetic_shift", 0.5 if tsm == "antithetic" else 0.0))
    max_depth = ip.get_i("max_depth", -1)
    if max_depth < 0 and max_depth != -1:
        raise ValueError('"max_depth" must be set to -1 (infinite) or a value >= 0')
    rr_depth = ip.get_i("rr_depth", 5)
    if rr_depth <= 0:
        raise ValueError('"rr_depth" must be set to a value greater than zero!')
    tcn = sp.get_i("time_correlate_number", 2) if sp.plugin == "correlated" else 2
    if ip.plugin == "dopplertofpath" and sp.plugin == "correlated" and tsm == "antithetic_mirror" and tcn != 2:
        raise ValueError("antithetic_mirror time sampling needs time_correlate_number == 2")   # Assert(m_time_correlate_number == 2), correlated.cpp:142
    return dict(
        time=T, w_g_mhz=w_g, g_1=g_1, g_0=g_0, w_s_mhz=w_s, phase_offset=phase, hetero_frequency=hf,
        wave_type=WAVE[wave], low_frequency_component_only=int(ip.get_b("low_frequency_component_only", True)),
        time_sampling=TIME[tsm], antithetic_shift=shift,
        stratify_each_interval=int(ip.get_b("use_stratified_sampling_for_each_interval", True)),
        path_correlation_depth=ip.get_i("path_correlation_depth", 0) & 0xffffffff,
        max_depth=max_depth & 0xffffffff, rr_depth=rr_depth, hide_emitters=int(ip.get_b("hide_emitters", False)),
        base_seed=sp.get_i("seed", 0) & 0xffffffff, time_correlate_number=tcn,
        path_correlate_number=sp.get_i("path_correlate_number", tcn) if sp.plugin == "correlated" else 2,
        sample_count=sp.get_i("sample_count", 4), integrator=kinds[ip.plugin], sampler=samplers[sp.plugin],
        jitter=int(sp.get_b("jitter", True)) if sp.plugin == "timestratified" else 1,
        # SamplingIntegrator::m_samples_per_pass (integrator.cpp:54-56): -1 -> (uint32_t) -1 = one pass
        samples_per_pass=int(ip.get_i("samples_per_pass", -1)) & 0xffffffff)
