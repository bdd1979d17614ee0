"""Evaluates the known-answer records harvested from the reference's unit tests (tests/golden/reference_kats.json.gz, produced by
tests/golden/extract_reference_kats.py) against a backend: the CPU oracle (tests/test_oracle_reference_kats.py) or the product's
C ABI on the GPU (tests/test_gpu_parity.py::test_reference_kats_*).

A record relates a *symbolic* expression -- the Mitsuba calls the reference's test made, with plain numbers as arguments -- to the
numbers the reference's test expects.  This module is a small facade of exactly those Mitsuba calls (`mi.fresnel`,
`mi.MicrofacetDistribution(...).smith_g1`, `mi.load_dict({'type': 'sphere', ...}).ray_test(ray)`, ...) over a backend object;
whatever the hot path does not contain (other plugins, spectral variants, AD) raises Skip and is counted, not hidden.
TEST INFRASTRUCTURE ONLY."""
import gzip
import json
import math
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json.gz")


class Skip(Exception):
    """the record needs a Mitsuba facility outside the dopplertofpath hot path (or one this backend does not expose)"""


def load_records():
    with gzip.open(GOLDEN, "rt") as fh:
        return json.load(fh)["records"]


# ---------------------------------------------------------------------------------------------------- scene description -> XML
class Transform:
    def __init__(self, m):
        self.m = np.asarray(m, np.float64).reshape(4, 4)

    @property
    def matrix(self):
        return self.m


_SHAPES = ("rectangle", "sphere", "disk", "cube", "cylinder", "obj", "ply")
_BSDFS = ("diffuse", "twosided", "conductor", "dielectric", "plastic", "thindielectric", "roughconductor", "roughdielectric", "roughplastic")
_EMITTERS = ("point", "area", "spot")
_FILTERS = ("box", "tent", "gaussian", "mitchell", "catmullrom", "lanczos")
_POINT_NAMES = ("center", "position", "origin", "target", "p0", "p1")


def _tag_of(plugin):
    if plugin in _SHAPES:
        return "shape"
    if plugin in _BSDFS:
        return "bsdf"
    if plugin in _EMITTERS:
        return "emitter"
    if plugin in _FILTERS:
        return "rfilter"
    if plugin == "hdrfilm":
        return "film"
    if plugin in ("perspective", "thinlens", "orthographic"):
        return "sensor"
    if plugin in ("independent", "correlated"):
        return "sampler"
    raise Skip("plugin '%s' is outside the hot path" % plugin)


def _fmt(x):
    return repr(float(x))


def to_xml(name, v, indent="  "):
    """one property of a mi.load_dict dictionary as scene XML (the subset the hot path knows)"""
    nm = ' name="%s"' % name if name else ""
    if isinstance(v, Transform):
        return '%s<transform%s><matrix value="%s"/></transform>\n' % (indent, nm, " ".join(_fmt(x) for x in v.m.reshape(-1)))
    if isinstance(v, dict):
        plugin = v.get("type")
        if plugin == "rgb":
            val = v["value"]
            val = [val] if np.isscalar(val) else list(np.asarray(val).reshape(-1))
            return '%s<rgb%s value="%s"/>\n' % (indent, nm, ", ".join(_fmt(x) for x in val))
        if plugin in ("d65", "regular", "uniform", "srgb", "bitmap", "checkerboard", "blackbody", "irregular", "srgb_d65"):
            raise Skip("spectrum / texture plugin '%s'" % plugin)
        tag = _tag_of(plugin)
        out = '%s<%s type="%s"%s>\n' % (indent, tag, plugin, nm if tag in ("bsdf",) and False else "")
        for k, x in v.items():
            if k == "type":
                continue
            child_name = None if isinstance(x, dict) and x.get("type") not in ("rgb",) else k
            out += to_xml(child_name, x, indent + "  ")
        return out + "%s</%s>\n" % (indent, tag)
    if isinstance(v, bool):
        return '%s<boolean%s value="%s"/>\n' % (indent, nm, "true" if v else "false")
    if isinstance(v, (int, np.integer)):
        return '%s<integer%s value="%d"/>\n' % (indent, nm, int(v))
    if isinstance(v, (float, np.floating)):
        return '%s<float%s value="%s"/>\n' % (indent, nm, _fmt(v))
    if isinstance(v, str):
        return '%s<string%s value="%s"/>\n' % (indent, nm, v)
    if isinstance(v, (list, tuple, np.ndarray)):
        a = np.asarray(v, np.float64).reshape(-1)
        if a.size == 3:
            return '%s<%s%s x="%s" y="%s" z="%s"/>\n' % (indent, "point" if name in _POINT_NAMES else "vector", nm, _fmt(a[0]), _fmt(a[1]), _fmt(a[2]))
    raise Skip("property %s of type %s" % (name, type(v).__name__))


_DEFAULT_SENSOR = """  <sensor type="perspective">
    <float name="fov" value="40"/>
    <transform name="to_world"><lookat origin="0, 0, 50" target="0, 0, 0" up="0, 1, 0"/></transform>
    <sampler type="correlated"><integer name="sample_count" value="4"/></sampler>
    <film type="hdrfilm"><integer name="width" value="4"/><integer name="height" value="4"/><rfilter type="box"/></film>
  </sensor>
"""


def scene_xml(shapes=(), sensor=None, extra=""):
    body = "".join(to_xml(None, s) for s in shapes)
    return ('<scene version="3.0.0">\n  <integrator type="dopplertofpath"/>\n' + (sensor or _DEFAULT_SENSOR) + body + extra + "</scene>\n")


# ---------------------------------------------------------------------------------------------------- Mitsuba-side value types
class Rec:
    """a plain record built by a Mitsuba constructor in the reference's test (Ray3f, SurfaceInteraction3f, Interaction3f, BSDFContext)"""

    def __init__(self, kind, args=(), kwargs=None, attrs=None):
        self.kind, self.args, self.kwargs = kind, list(args), dict(kwargs or {})
        for k, v in (attrs or {}).items():
            setattr(self, k, v)


def _v3(x):
    return np.asarray(x, np.float32).reshape(3)


def ray_of(r):
    """(o, d, time) of a mi.Ray3f record"""
    if not isinstance(r, Rec) or r.kind != "Ray3f":
        raise Skip("not a ray")
    if len(r.args) == 1 and isinstance(r.args[0], Rec):       # copy constructor
        o, d, t = ray_of(r.args[0])
    else:
        o = r.kwargs.get("o", r.args[0] if len(r.args) > 0 else None)
        d = r.kwargs.get("d", r.args[1] if len(r.args) > 1 else None)
        t = r.kwargs.get("time", r.args[2] if len(r.args) > 2 else 0.0)
    o = getattr(r, "o", o)
    d = getattr(r, "d", d)
    t = getattr(r, "time", t)
    if o is None or d is None:
        raise Skip("ray without origin / direction")
    return _v3(o), _v3(d), float(t)


class Frame:
    def __init__(self, s, t, n):
        self.s, self.t, self.n = s, t, n


class SurfaceHit:
    """what Scene::ray_intersect returns, as far as the hot path computes it"""

    def __init__(self, valid, vals, time=0.0):
        self._valid = bool(valid)
        f = [np.asarray(vals[1 + 3 * i: 4 + 3 * i], np.float32) for i in range(8)]
        self._f = dict(t=np.float32(vals[0]), p=f[0], n=f[1], sh_frame=Frame(f[3], f[4], f[2]), dp_du=f[5], dp_dv=f[6], wi=f[7],
                       time=np.float32(time))

    def is_valid(self):
        return self._valid

    def __getattr__(self, k):
        if k.startswith("_"):
            raise AttributeError(k)
        if k == "t" and not self._valid:
            return np.float32(np.inf)
        if not self._valid:
            raise Skip("field of a missed ray")
        if k == "uv":
            raise Skip("si.uv is not computed on the hot path (no textures)")
        if k in self._f:
            if k in ("dp_du", "dp_dv") and np.isnan(self._f[k]).any():
                raise Skip("dp_du / dp_dv are internal to the GPU's surface interaction")
            return self._f[k]
        raise Skip("SurfaceInteraction3f.%s is not part of the hot path" % k)


class ShapeFacade:
    """mi.load_dict({'type': <shape>, ...}) or a scene holding shapes"""

    def __init__(self, be, shapes):
        self.be, self.shapes = be, shapes
        self._scene = None

    _xml = None

    def scene(self):
        if self._scene is None:
            self._scene = self.be.load_scene(self._xml or scene_xml(self.shapes))
        return self._scene

    def ray_intersect(self, ray, *a, **k):
        o, d, t = ray_of(ray)
        valid, vals = self.be.ray_intersect(self.scene(), o, d, t)
        return SurfaceHit(valid, vals, t)

    def ray_test(self, ray, *a, **k):
        o, d, t = ray_of(ray)
        return self.be.ray_test(self.scene(), o, d, t)

    def surface_area(self):
        if self.shapes[0]["type"] in ("cube", "obj", "ply"):
            # the face-area table of a mesh exists only for emitters (Mesh::build_pmf, mesh.cpp:478-511): give it one
            lit = dict(self.shapes[0])
            lit["emitter"] = {"type": "area", "radiance": {"type": "rgb", "value": 1.0}}
            return self.be.shape_area(self.be.load_scene(scene_xml([lit])), 0)
        return self.be.shape_area(self.scene(), 0)

    def primitive_count(self):
        kind = self.shapes[0]["type"]
        if kind == "cube":
            return 12
        if kind in ("rectangle", "sphere", "disk", "cylinder"):
            return 1
        raise Skip("primitive_count of " + kind)

    def sample_direction(self, it, sample, *a, **k):
        if self.shapes[0]["type"] != "sphere":
            raise Skip("Shape::sample_direction is exposed for spheres only")
        out = self.be.sphere_sample_direction(self.scene(), 0, _v3(it.p), float(sample[0]), float(sample[1]))
        return Rec("DirectionSample", attrs=dict(p=out[0:3], n=out[3:6], d=out[6:9], dist=out[9], pdf=out[10]))

    def bbox(self):
        raise Skip("bounding boxes are a BVH-builder detail, not part of the per-lane path")


class BsdfFacade:
    """mi.load_dict({'type': <bsdf>}) / mi.load_string(<bsdf xml>): the BSDF sits on a unit rectangle"""

    def __init__(self, be, bsdf_xml):
        self.be = be
        shape = '  <shape type="rectangle">\n' + bsdf_xml + "  </shape>\n"
        self.scene = be.load_scene(scene_xml((), extra=shape))

    @staticmethod
    def _wi(si):
        if not hasattr(si, "wi"):
            raise Skip("si.wi not set")
        return _v3(si.wi)

    @staticmethod
    def _mode(ctx):
        if isinstance(ctx, Rec) and ctx.args and "Importance" in str(ctx.args[0]):
            raise Skip("TransportMode::Importance: the integrator only ever uses Radiance")
        if isinstance(ctx, Rec) and (getattr(ctx, "component", None) not in (None, 0xffffffff, -1) or getattr(ctx, "type_mask", None) is not None):
            raise Skip("component / lobe selection through BSDFContext is not used by the integrator")

    def _eval_pdf(self, ctx, si, wo):
        self._mode(ctx)
        out = self.be.bsdf(self.scene, 0, self._wi(si), _v3(wo), (0.5, 0.5, 0.5))
        wo = _v3(wo)
        # the path evaluates f * |cos| (BSDF::eval includes the cosine foreshortening term in Mitsuba 3 as well)
        return out[0:3], out[3]

    def eval(self, ctx, si, wo=None, *a, **k):
        return self._eval_pdf(ctx, si, wo)[0]

    def pdf(self, ctx, si, wo=None, *a, **k):
        return self._eval_pdf(ctx, si, wo)[1]

    def eval_pdf(self, ctx, si, wo=None, *a, **k):
        return self._eval_pdf(ctx, si, wo)

    def sample(self, ctx, si, sample1, sample2, *a, **k):
        self._mode(ctx)
        s2 = np.asarray(sample2, np.float32).reshape(-1)
        out = self.be.bsdf(self.scene, 0, self._wi(si), np.array([0, 0, 1], np.float32), (float(sample1), float(s2[0]), float(s2[1])))
        bs = Rec("BSDFSample3f", attrs=dict(wo=out[4:7], pdf=out[7], eta=out[8], delta=bool(out[9])))
        return bs, out[10:13]

    def component_count(self):
        raise Skip("lobe bookkeeping (component_count / flags) is not part of the per-lane path")


class FilterFacade:
    def __init__(self, be, d):
        self.be, self.d = be, d
        kind = d["type"]
        if kind not in _FILTERS:
            raise Skip("rfilter '%s'" % kind)
        self.kind = _FILTERS.index(kind)
        self.stddev = float(d.get("stddev", 0.5))
        self.B, self.C = float(d.get("B", 1 / 3)), float(d.get("C", 1 / 3))
        self.radius = float({"box": 0.5, "tent": d.get("radius", 1.0), "gaussian": 4 * self.stddev, "mitchell": 2.0, "catmullrom": 2.0, "lanczos": d.get("lobes", 3)}[kind])

    def eval(self, x, *a):
        return self.be.filter_eval(self.kind, self.radius, self.stddev, self.B, self.C, float(x))

    def eval_discretized(self, x, *a):
        """ReconstructionFilter::init_discretization + eval_discretized (src/core/rfilter.cpp:9-24, rfilter.h:66-75; the scalar
        variants' 32-entry table of eval): the table is rebuilt from the backend's eval"""
        res = 31   # MI_FILTER_RESOLUTION
        idx = min(int(abs(np.float32(x) * np.float32(res / self.radius))), res)
        return 0.0 if idx == res else self.eval(np.float32(self.radius * idx) / np.float32(res))

    def border_size(self):
        raise Skip("border bookkeeping")


class Microfacet:
    def __init__(self, be, *args):
        self.be = be
        t = str(args[0])
        self.type = 0 if "Beckmann" in t else 1
        rest = list(args[1:])
        self.visible = True
        if rest and isinstance(rest[-1], bool):
            self.visible = rest.pop()
        self.au = float(rest[0])
        self.av = float(rest[1]) if len(rest) > 1 else self.au

    def alpha_u(self):
        return max(self.au, 1e-4)

    def alpha_v(self):
        return max(self.av, 1e-4)

    def sample_visible(self):
        return self.visible

    def is_isotropic(self):
        return self.au == self.av

    def is_anisotropic(self):
        return self.au != self.av

    def type_(self):
        return self.type

    @staticmethod
    def _dirs(v):
        a = np.asarray(v, np.float32) if not isinstance(v, (list, tuple)) else None
        if a is None:
            comps = [np.atleast_1d(np.asarray(c, np.float32)) for c in v]
            n = max(c.size for c in comps)
            a = np.stack([np.broadcast_to(c, (n,)) for c in comps], axis=1)
        elif a.ndim == 1:
            a = a.reshape(1, 3)
        return np.ascontiguousarray(a, np.float32)

    def _call(self, fn, a, b=None):
        a = self._dirs(a)
        if b is not None:
            b = self._dirs(b)
            n = max(len(a), len(b))
            a, b = np.broadcast_to(a, (n, a.shape[1])), np.broadcast_to(b, (n, b.shape[1]))
            inp = np.concatenate([a, b], axis=1)
        else:
            inp = a
        return self.be.microfacet(self.type, self.au, self.av, self.visible, fn, np.ascontiguousarray(inp, np.float32))

    def eval(self, m):
        return self._call(0, m)[:, 0]

    def pdf(self, wi, m):
        return self._call(1, wi, m)[:, 0]

    def smith_g1(self, v, m):
        return self._call(2, v, m)[:, 0]

    def sample(self, wi, u):
        u = [np.atleast_1d(np.asarray(c, np.float32)) for c in u]
        uu = np.stack([np.broadcast_to(u[0], (max(u[0].size, u[1].size),)), np.broadcast_to(u[1], (max(u[0].size, u[1].size),))], axis=1)
        out = self._call(3, wi, uu)
        return out[:, 0:3], out[:, 3]


class ImageBlockFacade:
    def __init__(self, be, kwargs, calls):
        if kwargs.get("normalize") or "size" not in kwargs or kwargs.get("border", False):
            raise Skip("HDRFilm's block: normalize = false, no border (hdrfilm.cpp:266-271)")
        self.be, self.kw, self.calls = be, kwargs, calls
        if int(kwargs.get("channel_count", 0)) != 1:
            raise Skip("channel count")

    def tensor(self):
        w, h = [int(x) for x in self.kw["size"]]
        f = self.kw.get("rfilter")
        if not isinstance(f, FilterFacade):
            raise Skip("block without a reconstruction filter")
        film = np.zeros((h, w, 4), np.float32)
        for c in self.calls:
            if c["method"] != "put":
                raise Skip("ImageBlock." + c["method"])
            pos = c["kwargs"].get("pos", c["args"][0] if c["args"] else None)
            vals = c["kwargs"].get("values", c["args"][1] if len(c["args"]) > 1 else None)
            v = float(np.asarray(vals[0]).reshape(-1)[0])
            self.be.splat(film, f, float(pos[0]), float(pos[1]), (v, v, v))
        return Rec("TensorXf", attrs=dict(array=film[:, :, 0].reshape(-1)))


class Sensor:
    def __init__(self, be, d):
        self.be, self.d = be, d
        self.scene = be.load_scene(scene_xml((), sensor=to_xml(None, d)))
        self.film_size = (int(d["film"]["width"]), int(d["film"]["height"]))

    def near_clip(self):
        return float(self.d.get("near_clip", 1e-2))

    def far_clip(self):
        return float(self.d.get("far_clip", 1e4))

    def shutter_open(self):
        return self.be.sensor_info(self.scene)["shutter_open"]

    def shutter_open_time(self):
        i = self.be.sensor_info(self.scene)
        return i["shutter_close"] - i["shutter_open"]

    def needs_aperture_sample(self):
        return self.d["type"] == "thinlens"

    def focus_distance(self):
        if self.d["type"] != "thinlens":
            raise Skip("focus_distance is not used by the perspective (pinhole) camera")
        return self.be.sensor_info(self.scene)["focus_distance"]

    def bbox(self):
        raise Skip("Sensor::bbox")

    def world_transform(self):
        return Transform(self.be.sensor_info(self.scene)["to_world"])

    def _ray(self, time, pos, ap=(.5, .5)):
        pos, ap = np.asarray(pos, np.float64), np.asarray(ap, np.float64)
        if ap.ndim == 0:
            ap = np.array([float(ap), float(ap)])
        if pos.ndim == 2 or ap.ndim == 2:      # vectorised variants: [[x0, x1, ..], [y0, y1, ..]]
            n = pos.shape[1] if pos.ndim == 2 else ap.shape[1]
            pos = pos if pos.ndim == 2 else np.repeat(pos[:, None], n, axis=1)
            ap = ap if ap.ndim == 2 else np.repeat(ap[:, None], n, axis=1)
            rays = [self.be.camera_ray(self.scene, float(px) * self.film_size[0], float(py) * self.film_size[1], float(ax), float(ay))
                    for px, py, ax, ay in zip(pos[0], pos[1], ap[0], ap[1])]
            o = np.stack([r[0] for r in rays], axis=1)
            d = np.stack([r[1] for r in rays], axis=1)
        else:
            o, d = self.be.camera_ray(self.scene, float(pos[0]) * self.film_size[0], float(pos[1]) * self.film_size[1], float(ap[0]), float(ap[1]))
        return Rec("Ray3f", attrs=dict(o=o, d=d, time=np.float32(time))), Skipper("spectral weight of sample_ray")

    def sample_ray(self, time, wav, pos, ap, *a):
        return self._ray(time, pos, ap)

    def sample_ray_differential(self, time, wav, pos, ap, *a):
        r, w = self._ray(time, pos, ap)
        return r, w


class Film:
    """mi.load_dict({'type': 'hdrfilm', ...}): Film::size / crop_size / crop_offset (src/render/film.cpp:7-54)"""

    def __init__(self, be, d):
        sensor = '  <sensor type="perspective">\n    <float name="fov" value="40"/>\n' + to_xml(None, d, "    ") + "  </sensor>\n"
        self.info = be.film_info(be.load_scene(scene_xml((), sensor=sensor)))

    def size(self):
        return np.array(self.info["size"])

    def crop_size(self):
        return np.array(self.info["crop_size"])

    def crop_offset(self):
        return np.array(self.info["crop_offset"])

    def rfilter(self):
        raise Skip("Film::rfilter object")

    def sample_border(self):
        raise Skip("sample_border")


class Skipper:
    """a value the hot path does not have: any use of it skips the record"""

    def __init__(self, why):
        self.why = why

    def __getattr__(self, k):
        raise Skip(self.why)

    def __getitem__(self, k):
        raise Skip(self.why)


# ---------------------------------------------------------------------------------------------------- expression evaluation
def _numeric(v):
    if isinstance(v, (list, tuple)):
        try:
            return np.asarray(v, np.float64)
        except (ValueError, TypeError):
            raise Skip("ragged / non-numeric list")
    return v


_OPS = {
    "Add": lambda a, b: a + b, "Sub": lambda a, b: a - b, "Mult": lambda a, b: a * b, "Div": lambda a, b: a / b, "Pow": lambda a, b: a ** b,
    "Eq": lambda a, b: a == b, "NotEq": lambda a, b: a != b, "Lt": lambda a, b: a < b, "LtE": lambda a, b: a <= b, "Gt": lambda a, b: a > b,
    "GtE": lambda a, b: a >= b, "And": lambda a, b: np.logical_and(a, b), "Or": lambda a, b: np.logical_or(a, b),
    "USub": lambda a: -a, "Not": lambda a: np.logical_not(a), "Mod": lambda a, b: a % b,
}
def _dot(a, b):
    """dr.dot of Dr.Jit vectors: coordinates first, so a (3, n) array of n vectors dotted with one (3,) vector gives n values"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.ndim == 2 and b.ndim == 1:
        return (a * b[:, None]).sum(axis=0)
    if a.ndim == 1 and b.ndim == 2:
        return (a[:, None] * b).sum(axis=0)
    if a.ndim == 2 and b.ndim == 2:
        return (a * b).sum(axis=0)
    return np.dot(a, b)


_DRFN = {
    "dr.abs": np.abs, "dr.sqrt": np.sqrt, "dr.cos": np.cos, "dr.sin": np.sin, "dr.acos": np.arccos, "dr.asin": np.arcsin, "dr.tan": np.tan,
    "dr.norm": lambda v: np.linalg.norm(np.asarray(v, np.float64)), "dr.dot": lambda a, b: _dot(a, b),
    "dr.normalize": lambda v: np.asarray(v, np.float64) / np.linalg.norm(np.asarray(v, np.float64)), "dr.rcp": lambda x: 1.0 / x,
    "dr.select": lambda c, a, b: np.where(c, a, b), "dr.sqr": lambda x: x * x, "dr.maximum": np.maximum, "dr.minimum": np.minimum,
    "dr.max": np.max, "dr.min": np.min, "dr.sum": np.sum, "dr.exp": np.exp, "dr.log": np.log, "fn.vector": lambda *a: np.asarray(a[0] if len(a) == 1 else a, np.float64),
    "dr.isnan": np.isnan, "dr.all": np.all, "dr.any": np.any, "dr.cross": lambda a, b: np.cross(np.asarray(a, np.float64), np.asarray(b, np.float64)),
    "dr.rsqrt": lambda x: 1.0 / np.sqrt(x), "dr.atan2": np.arctan2,
    "dr.allclose": lambda a, b, rtol=1e-5, atol=1e-8: allclose(a, b, rtol, atol),
}


class Evaluator:
    def __init__(self, backend):
        self.be = backend
        self.cache = {}

    # -- values
    def ev(self, v):
        if isinstance(v, dict):
            if "sym" in v:
                return self.sym(v)
            if "f32" in v:
                return np.asarray(v["f32"], np.float32)
            if "transform" in v:
                return Transform(v["transform"])
            if "nd" in v:
                return np.asarray(v["nd"], np.float64)
            if set(v.keys()) == {"type"} and v["type"] in ("Float", "UInt32"):
                return v["type"]
            return {k: self.ev(x) for k, x in v.items()}
        if isinstance(v, list):
            return [self.ev(x) for x in v]
        return v

    def sym(self, n):
        k = n["sym"]
        if k == "call":
            key = json.dumps(n, sort_keys=True)
            if key in self.cache:
                return self.cache[key]
            r = self.call(n)
            if len(key) < 20000:
                self.cache[key] = r
            return r
        if k == "attr":
            base = self.ev(n["of"])
            return self.attr(base, n["name"])
        if k == "item":
            base = self.ev(n["of"])
            idx = n["index"]
            if isinstance(idx, list):
                idx = slice(*idx)
            if isinstance(base, (Skipper,)):
                raise Skip(base.why)
            return base[idx]
        if k == "name":
            return ("name", n["name"])
        raise Skip("symbol kind " + k)

    def attr(self, base, name):
        if isinstance(base, np.ndarray) and name in "xyzw":
            return base["xyzw".index(name)]
        if isinstance(base, (list, tuple)) and name in "xyzw":
            return base["xyzw".index(name)]
        if isinstance(base, tuple) and len(base) == 2 and base[0] == "name":
            return ("name", base[1] + "." + name)
        try:
            return getattr(base, name)
        except AttributeError:
            raise Skip("%s.%s is not part of the hot path's facade" % (type(base).__name__, name))

    def call(self, n):
        fn = n["fn"]
        args = [self.ev(a) for a in n.get("args", [])]
        kwargs = {k: self.ev(x) for k, x in n.get("kwargs", {}).items()}
        attrs = {k: self.ev(x) for k, x in n.get("set", {}).items() if k != "__calls__"}
        calls = [{"method": c["method"], "args": [self.ev(a) for a in c["args"]], "kwargs": {k: self.ev(x) for k, x in c["kwargs"].items()}}
                 for c in n.get("set", {}).get("__calls__", [])]
        if fn.get("sym") == "name":
            r = self.api(fn["name"], args, kwargs, attrs, calls)
        else:
            f = self.ev(fn)
            if isinstance(f, tuple) and len(f) == 2 and f[0] == "name":
                r = self.api(f[1], args, kwargs, attrs, calls)
            elif callable(f):
                r = f(*args, **kwargs)
            else:
                raise Skip("call of a non-callable")
        if attrs and isinstance(r, Rec):
            for k, v in attrs.items():
                setattr(r, k, v)
        return r

    # -- the Mitsuba calls the reference's tests make
    def api(self, name, args, kwargs, attrs, calls):
        be = self.be
        if name in ("op.BitOr", "op.BitAnd"):
            return ("name", "flags")          # RayFlags / BSDFFlags combinations: the hot path always computes everything it has
        if name.startswith("op."):
            a = [_numeric(x) for x in args]
            for x in a:
                if isinstance(x, Skipper):
                    raise Skip(x.why)
            with np.errstate(all="ignore"):
                return _OPS[name[3:]](*a)
        if name in _DRFN:
            with np.errstate(all="ignore"):
                return _DRFN[name](*[_numeric(x) for x in args], **kwargs)
        if name == "dr.zeros" or name == "dr.ones":
            t = args[0]
            if isinstance(t, tuple) and t[0] == "name":
                return Rec(t[1].split(".")[-1], attrs=attrs)
            n_ = int(args[1]) if len(args) > 1 else 1
            return (np.zeros if name == "dr.zeros" else np.ones)(n_, np.float32)
        if name in ("mi.Ray3f", "mi.RayDifferential3f", "mi.SurfaceInteraction3f", "mi.Interaction3f", "mi.BSDFContext"):
            kind = "Ray3f" if "Ray" in name else name.split(".")[-1]
            return Rec(kind, args, kwargs, attrs)
        if name == "mi.fresnel":
            c, eta = np.atleast_1d(np.asarray(args[0], np.float32)), float(args[1])
            out = np.stack([be.fresnel(float(x), eta) for x in c], axis=1)
            return tuple(out[i] if c.size > 1 else out[i, 0] for i in range(4))
        if name == "mi.fresnel_conductor":
            raise Skip("complex-IOR overload")
        if name == "mi.MicrofacetDistribution":
            return Microfacet(be, *args, **kwargs)
        if name.startswith("mi.MicrofacetType") or name.startswith("mi.TransportMode") or name.startswith("mi.BSDFFlags"):
            return name
        if name == "mi.load_dict":
            return self.load_dict(args[0])
        if name == "mi.load_string":
            xml = args[0]
            if "<bsdf" in xml and "<shape" not in xml and "<scene" not in xml:
                body = xml.replace(' version="3.0.0"', "")
                return BsdfFacade(be, body)
            raise Skip("load_string of something else than a BSDF")
        if name == "mi.ImageBlock":
            return ImageBlockFacade(be, kwargs, calls)
        if name == "mi.sample_tea_float32":
            return be.tea_float32(int(args[0]), int(args[1]), int(args[2]) if len(args) > 2 else 4)
        if name == "mi.Frame3f":
            if len(args) == 1:
                n_ = _v3(args[0])
                s, t = be.coordinate_system(n_)
                return Frame(s, t, n_)
            raise Skip("Frame3f from three vectors is plain storage")
        if name == "mi.quad.gauss_legendre":
            return be.gauss_legendre(int(args[0]))
        if name == "mi.math.solve_quadratic":
            return be.solve_quadratic(*[float(x) for x in args])
        if name.startswith("mi.warp."):
            w = name[len("mi.warp."):]
            table = {"square_to_cosine_hemisphere": (0, 3), "square_to_uniform_disk_concentric": (1, 2), "square_to_uniform_triangle": (3, 2),
                     "square_to_uniform_sphere": (4, 3)}
            if w not in table:
                raise Skip("warp '%s' is not used on the hot path" % w)
            s = np.asarray(args[0], np.float32).reshape(-1)
            return be.warp(table[w][0], float(s[0]), float(s[1]))[:table[w][1]]
        raise Skip("Mitsuba API '%s' is outside the hot path" % name)

    def load_dict(self, d):
        t = d.get("type")
        if t == "scene":
            shapes, xml = [], ""
            for key, v in d.items():
                if not isinstance(v, dict):
                    continue
                vt = v.get("type")
                if vt in _SHAPES:
                    shapes.append(v)
                    xml += to_xml(None, v)
                elif vt == "shapegroup":      # src/render/shapegroup.cpp: the children, addressed by the dictionary key
                    xml += '  <shape type="shapegroup" id="%s">\n' % key
                    for ck, cv in v.items():
                        if ck != "type":
                            xml += to_xml(None, cv, "    ")
                    xml += "  </shape>\n"
                elif vt == "instance":        # src/shapes/instance.cpp
                    ref = [cv for cv in v.values() if isinstance(cv, dict) and cv.get("type") == "ref"]
                    if len(ref) != 1:
                        raise Skip("instance without exactly one group reference")
                    xml += '  <shape type="instance">\n    <ref id="%s"/>\n' % ref[0]["id"]
                    if "to_world" in v:
                        xml += to_xml("to_world", v["to_world"], "    ")
                    xml += "  </shape>\n"
                else:
                    raise Skip("scene child of type '%s'" % vt)
            if not xml:
                raise Skip("empty scene")
            f = ShapeFacade(self.be, shapes)
            f._xml = scene_xml((), extra=xml)
            return f
        if t == "hdrfilm":
            return Film(self.be, d)
        if t in _SHAPES:
            return ShapeFacade(self.be, [d])
        if t in _BSDFS:
            return BsdfFacade(self.be, to_xml(None, d, "    "))
        if t in _FILTERS or t == "lanczos":
            return FilterFacade(self.be, d)
        if t in ("perspective", "thinlens", "orthographic"):
            return Sensor(self.be, d)
        raise Skip("load_dict of plugin '%s'" % t)


# ---------------------------------------------------------------------------------------------------- checking one record
def allclose(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    try:
        a, b = np.broadcast_arrays(a, b)
    except ValueError:
        if a.size == b.size:
            a, b = a.reshape(-1), b.reshape(-1)
        elif a.T.shape == b.shape:
            a = a.T
        else:
            raise
    if a.shape != b.shape and a.T.shape == b.shape:
        a = a.T
    return bool(np.all(np.abs(a - b) <= atol + rtol * np.abs(b)))   # dr.allclose: |a - b| <= |b| * rtol + atol


def _fix_layout(lhs, rhs):
    """Dr.Jit arrays of vectors are component-major ([3][n]); the facade returns [n][3]"""
    a, b = np.asarray(lhs, np.float64), np.asarray(rhs, np.float64)
    if a.ndim == 2 and b.ndim == 2 and a.shape != b.shape and a.T.shape == b.shape:
        return a.T, b
    return a, b


def check(rec, ev):
    """True / False = the backend agrees / disagrees with the reference's expectation; raises Skip if not applicable"""
    lhs, rhs = ev.ev(rec["lhs"]), ev.ev(rec["rhs"])
    for x in (lhs, rhs):
        if isinstance(x, Skipper):
            raise Skip(x.why)
    kind = rec["kind"]
    with np.errstate(all="ignore"):
        if kind == "allclose":
            a, b = _fix_layout(_numeric(lhs), _numeric(rhs))
            return allclose(a, b, float(rec.get("rtol", 1e-5)), float(rec.get("atol", 1e-8)))
        if kind == "truth":
            return bool(np.all(lhs)) == bool(rhs)
        base = kind.split(".")[-1]
        if base in _OPS:
            r = _OPS[base](_numeric(lhs), _numeric(rhs))
            return bool(np.any(r)) if kind.startswith("any.") else bool(np.all(r))
    raise Skip("assertion kind " + kind)


def run_all(backend, records=None, only_files=None):
    """-> dict file -> {'pass': n, 'fail': [(line, test, detail)], 'skip': Counter(reason)}"""
    import collections
    ev = Evaluator(backend)
    out = {}
    for rec in records or load_records():
        f = rec["file"]
        if only_files and f not in only_files:
            continue
        st = out.setdefault(f, {"pass": 0, "fail": [], "skip": collections.Counter(), "passed_tests": collections.Counter()})
        try:
            # an assertion under `if <something Mitsuba computes>:` applies only when the backend makes that condition true too
            if not all(bool(np.all(ev.ev(c))) for c in rec.get("conditions", [])):
                st["skip"]["condition of the enclosing `if` is false"] += 1
                continue
            ok = check(rec, ev)
        except Skip as e:
            st["skip"][str(e)[:90]] += 1
            continue
        except Exception as e:      # noqa: BLE001 - a facade bug must show up as a failure of that record, not end the run
            st["fail"].append((rec["line"], rec["test"], "%s: %s" % (type(e).__name__, str(e)[:100])))
            continue
        if ok:
            st["pass"] += 1
            st["passed_tests"][rec["test"]] += 1
        elif rec["kind"] == "Eq" and isinstance(rec["lhs"], dict) and isinstance(rec["rhs"], dict):
            # a relation between two hit / miss decisions (instanced shape vs the same shape placed directly, test_instance.py): rays that
            # graze the primitive's edge may fall either way in float32 -- counted separately, bounded by the test
            st.setdefault("edge", []).append((rec["line"], rec["test"]))
        else:
            st["fail"].append((rec["line"], rec["test"], json.dumps(rec.get("params"))[:120]))
    return out
