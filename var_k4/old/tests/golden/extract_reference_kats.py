#!/usr/bin/env python3
"""Harvests the known-answer numbers the reference's own unit tests hold (build container only: reads /root/reference).

The reference (juhyeonkim95/Mitsuba3DopplerToF) cannot be built or imported here (Dr.Jit / Embree submodules are empty), so
its tests cannot be *run*; but they can be *read*.  Most of them have the shape

    assert dr.allclose(<something only Mitsuba can compute>(<plain numbers>), <plain numbers>, atol=...)

This script partially evaluates the test functions with Python's `ast` module: everything that is plain arithmetic on literals
(`dr.linspace`, `dr.cos`, lists, loops, `pytest.mark.parametrize` bindings, `mi.ScalarTransform4f.translate(...) @ ...`) is
computed with numpy; everything that needs Mitsuba (`mi.fresnel(...)`, `mdf.smith_g1(v, wi)`, `scene.ray_intersect(ray)`)
stays a *symbolic* call description whose arguments are numbers.  Every assertion that relates such a symbolic call to numbers
becomes one record:  {file, line, test, lhs, rhs, kind, atol, rtol}.

Only VALUES are stored (tests/golden/reference_kats.json.gz): call names, argument numbers, expected numbers, tolerances, and the
file:line they come from -- never the text of the reference's files.  tests/test_oracle_reference_kats.py maps the call names
onto oracle/ entry points (CPU) and onto the product's dtof_eval_* entry points (GPU) and checks them with the tolerance the
reference's test states.

Usage:  python tests/golden/extract_reference_kats.py [/root/reference]  ->  tests/golden/reference_kats.json.gz
"""
import ast
import io
import itertools
import json
import math
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

# reference test files that cover components of the dopplertofpath hot path (SURVEY 8a rows in brackets)
FILES = [
    "src/render/tests/test_microfacet.py",      # GGX / Beckmann distributions (8f-3 rough BSDFs)
    "src/render/tests/test_fresnel.py",         # fresnel, fresnel_conductor (8f-3)
    "src/rfilters/tests/test_rfilter.py",       # reconstruction filters (I1)
    "src/core/tests/test_warp.py",              # warps (M1, E1)
    "src/core/tests/test_random.py",            # TEA, Kensler (S7)
    "src/core/tests/test_transform.py",         # Transform4f (X1, G2)
    "src/core/tests/test_frame.py",             # Frame3f (G4)
    "src/core/tests/test_math.py",              # solve_quadratic etc. (sphere)
    "src/core/tests/test_quad.py",              # Gauss-Legendre (roughplastic tables)
    "src/shapes/tests/test_rectangle.py",       # G3
    "src/shapes/tests/test_sphere.py",
    "src/shapes/tests/test_disk.py",
    "src/shapes/tests/test_cube.py",
    "src/shapes/tests/test_cylinder.py",
    "src/shapes/tests/test_instance.py",        # G2
    "src/shapes/tests/test_mesh.py",
    "src/emitters/tests/test_point.py",         # E1
    "src/emitters/tests/test_area.py",
    "src/emitters/tests/test_spot.py",
    "src/sensors/tests/test_perspective.py",    # C1
    "src/sensors/tests/test_thinlens.py",       # D2 (the aperture draw), 8(f) thinlens sensor
    "src/sensors/tests/test_orthographic.py",   # 8(f) orthographic sensor
    "src/render/tests/test_imageblock.py",      # I1
    "src/bsdfs/tests/test_diffuse.py",          # M1
    "src/bsdfs/tests/test_twosided.py",
    "src/bsdfs/tests/test_conductor.py",
    "src/bsdfs/tests/test_dielectric.py",
    "src/bsdfs/tests/test_plastic.py",
    "src/bsdfs/tests/test_thindielectric.py",
    "src/bsdfs/tests/test_rough_conductor.py",
    "src/bsdfs/tests/test_rough_dielectric.py",
    "src/bsdfs/tests/test_rough_plastic.py",
    "src/samplers/tests/test_independent.py",   # S2
    "src/films/tests/test_hdrfilm.py",          # I1
]

# files whose assertions relate two Mitsuba computations to each other (instanced == plain shape): kept as relations
RELATIONS = ("src/shapes/tests/test_instance.py", "src/shapes/tests/test_cylinder.py")

MAX_LOOP = 700          # iterations of one `for`
MAX_RECORDS_PER_TEST = 3000
MAX_PARAM_COMBOS = 64


class Unknown(Exception):
    """raised when an expression cannot be evaluated numerically or symbolically"""


class Sym:
    """A value only Mitsuba can compute, described by how the test obtained it.
    kind: 'name' (mi.warp.square_to_uniform_disk), 'call' (base(*args, **kwargs)), 'attr' (base.name), 'item' (base[index]).
    Objects built by a constructor call may get attributes assigned later (`si.wi = [...]`); `attrs` holds those numbers."""

    def __init__(self, kind, base=None, name=None, args=None, kwargs=None):
        self.kind, self.base, self.name, self.args, self.kwargs = kind, base, name, args or [], kwargs or {}
        self.attrs = {}

    def snapshot(self):
        s = Sym(self.kind, self.base.snapshot() if isinstance(self.base, Sym) else self.base, self.name,
                [freeze(a) for a in self.args], {k: freeze(v) for k, v in self.kwargs.items()})
        s.attrs = {k: freeze(v) for k, v in self.attrs.items()}
        return s


def freeze(v):
    if isinstance(v, Sym):
        return v.snapshot()
    if isinstance(v, np.ndarray):
        return v.copy()
    if isinstance(v, list):
        return [freeze(x) for x in v]
    if isinstance(v, tuple):
        return tuple(freeze(x) for x in v)
    if isinstance(v, dict):
        return {k: freeze(x) for k, x in v.items()}
    return v


def has_sym(v):
    if isinstance(v, Sym):
        return True
    if isinstance(v, (list, tuple)):
        return any(has_sym(x) for x in v)
    if isinstance(v, dict):
        return any(has_sym(x) for x in v.values())
    return False


class Xf:
    """ScalarTransform4f / Transform4f: a 4x4 matrix in double precision (Properties::Float is double, xml.cpp:91)"""

    def __init__(self, m):
        self.m = np.asarray(m, np.float64).reshape(4, 4)

    def __matmul__(self, o):
        if isinstance(o, Xf):
            return Xf(self.m @ o.m)
        raise Unknown("transform applied to a value")


def _vec(v, n=3):
    a = np.asarray(v, np.float64).reshape(-1)
    if a.size == 1:
        a = np.repeat(a, n)
    return a


def xf_translate(v):
    m = np.eye(4)
    m[:3, 3] = _vec(v)
    return Xf(m)


def xf_scale(v):
    return Xf(np.diag(np.append(_vec(v), 1.0)))


def xf_rotate(axis, angle):
    a = _vec(axis)
    a = a / np.linalg.norm(a)
    s, c = math.sin(math.radians(angle)), math.cos(math.radians(angle))
    x, y, z = a
    m = np.eye(4)
    m[:3, :3] = [[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                 [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                 [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]]
    return Xf(m)


def xf_look_at(origin, target, up):
    o, t, u = _vec(origin), _vec(target), _vec(up)
    d = (t - o) / np.linalg.norm(t - o)
    left = np.cross(u, d)
    left /= np.linalg.norm(left)
    nu = np.cross(d, left)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, nu, d, o
    return Xf(m)


F32 = np.float32


def f32(x):
    return np.asarray(x, dtype=F32)


def _linspace(_t, a, b, n, endpoint=True):
    # drjit linspace: step = (max - min) / (n - endpoint) in float32, value_i = fmadd(i, step, min)
    lo, hi = F32(a), F32(b)
    step = F32((hi - lo) / F32(int(n) - (1 if endpoint else 0))) if int(n) > 1 else F32(0)
    return (np.arange(int(n), dtype=np.float64) * np.float64(step) + np.float64(lo)).astype(F32)


def _meshgrid(x, y):     # drjit.meshgrid default indexing 'xy': x varies fastest
    x, y = f32(x), f32(y)
    return (np.tile(x, y.size), np.repeat(y, x.size))


def _num1(fn):
    def g(x):
        if isinstance(x, (list, tuple)):
            x = f32(x)
        if isinstance(x, np.ndarray):
            return fn(x.astype(F32)).astype(F32)
        return float(fn(np.float64(x)))
    return g


DR = {
    "pi": math.pi, "inv_pi": 1.0 / math.pi, "two_pi": 2 * math.pi, "inv_two_pi": 0.5 / math.pi, "inv_four_pi": 0.25 / math.pi,
    "inf": math.inf, "nan": math.nan, "epsilon": lambda *_: 2.0 ** -24,
    "sqrt": _num1(np.sqrt), "cos": _num1(np.cos), "sin": _num1(np.sin), "tan": _num1(np.tan), "acos": _num1(np.arccos),
    "asin": _num1(np.arcsin), "atan": _num1(np.arctan), "exp": _num1(np.exp), "log": _num1(np.log), "abs": _num1(np.abs),
    "rcp": _num1(lambda x: 1.0 / x), "rsqrt": _num1(lambda x: 1.0 / np.sqrt(x)), "sqr": _num1(lambda x: x * x),
    "deg2rad": _num1(np.deg2rad), "rad2deg": _num1(np.rad2deg),
    "linspace": _linspace,
    "arange": lambda _t, *a: np.arange(*[int(x) for x in a]),
    "full": lambda _t, v, n=1: np.full(int(n), v, F32),
    "zeros": lambda _t, n=1: np.zeros(int(n), F32) if _t is FLOAT else (_ for _ in ()).throw(Unknown("zeros of a struct")),
    "ones": lambda _t, n=1: np.ones(int(n), F32),
    "meshgrid": _meshgrid,
    "norm": lambda v: float(np.linalg.norm(np.asarray(v, np.float64))),
    "normalize": lambda v: np.asarray(v, np.float64) / np.linalg.norm(np.asarray(v, np.float64)),
    "dot": lambda a, b: float(np.dot(np.asarray(a, np.float64), np.asarray(b, np.float64))),
    "cross": lambda a, b: np.cross(np.asarray(a, np.float64), np.asarray(b, np.float64)),
    "maximum": np.maximum, "minimum": np.minimum, "max": np.max, "min": np.min, "sum": np.sum,
    "atan2": lambda a, b: float(np.arctan2(a, b)), "fma": lambda a, b, c: a * b + c, "fmadd": lambda a, b, c: a * b + c,
    "select": lambda c, a, b: np.where(c, a, b), "clamp": lambda x, a, b: np.clip(x, a, b), "clip": lambda x, a, b: np.clip(x, a, b),
    "eval": lambda *a: None,
}
def _named(name, fn):
    def g(*a, **k):
        return fn(*a, **k)
    g.__name__ = name
    return g


DR = {k: (_named("dr." + k, v) if callable(v) else v) for k, v in DR.items()}
FLOAT = object()      # the type token mi.Float / drjit.scalar.ArrayXf
UINT = object()


def mi_vector(*a):
    """mi.Vector3f / Point3f / ... : plain numbers"""
    if len(a) == 1:
        a = a[0]
    if has_sym(a):
        raise Unknown("vector of symbols")
    return np.asarray(a, np.float64)


# mi.* names that are plain data (everything else under mi.* is symbolic)
MI_NUMERIC = {
    "Float": FLOAT, "ScalarFloat": FLOAT, "UInt32": UINT, "UInt": UINT, "Float32": FLOAT,
    "Vector3f": mi_vector, "Point3f": mi_vector, "Normal3f": mi_vector, "Vector2f": mi_vector, "Point2f": mi_vector,
    "ScalarVector3f": mi_vector, "ScalarPoint3f": mi_vector, "ScalarNormal3f": mi_vector, "ScalarVector2f": mi_vector,
    "ScalarPoint2f": mi_vector, "Color3f": mi_vector, "ScalarColor3f": mi_vector, "Vector2u": mi_vector, "ScalarVector2u": mi_vector,
    "ScalarVector2i": mi_vector, "ScalarPoint2i": mi_vector, "ScalarPoint2u": mi_vector, "Point2u": mi_vector, "Vector2i": mi_vector,
}
XF_STATIC = {"translate": xf_translate, "scale": xf_scale, "rotate": xf_rotate, "look_at": xf_look_at}


class XfType:
    def __call__(self, *a):
        if len(a) == 0:
            return Xf(np.eye(4))
        return Xf(np.asarray(a[0], np.float64))


XFTYPE = XfType()


class Namespace:
    def __init__(self, name):
        self.name = name


class Interp:
    def __init__(self, relpath):
        self.relpath = relpath
        self.records = []
        self.test = None
        self.binding = None
        self.n_test_records = 0

    # ------------------------------------------------------------------ expressions
    def ev(self, node, env):
        m = getattr(self, "ev_" + type(node).__name__, None)
        if m is None:
            raise Unknown(type(node).__name__)
        try:
            return m(node, env)
        except Unknown:
            raise
        except RecursionError:
            raise
        except Exception as e:      # noqa: BLE001 - whatever plain Python cannot do with these values is "unknown"
            raise Unknown("%s: %s" % (type(e).__name__, e))

    def ev_Constant(self, n, env):
        return n.value

    def ev_Name(self, n, env):
        if n.id in env:
            v = env[n.id]
            if isinstance(v, Unknown):
                raise v
            return v
        builtins = {"abs": abs, "range": range, "len": len, "float": float, "int": int, "min": min, "max": max, "sum": sum,
                    "list": list, "tuple": tuple, "zip": zip, "enumerate": enumerate, "True": True, "False": False, "None": None,
                    "round": round, "sorted": sorted, "bool": bool, "str": str, "all": all, "any": any, "pow": pow}
        if n.id in builtins:
            return builtins[n.id]
        raise Unknown("name " + n.id)

    def ev_List(self, n, env):
        return [self.ev(e, env) for e in n.elts]

    def ev_Tuple(self, n, env):
        return tuple(self.ev(e, env) for e in n.elts)

    def ev_Dict(self, n, env):
        return {self.ev(k, env): self.ev(v, env) for k, v in zip(n.keys, n.values)}

    def ev_JoinedStr(self, n, env):
        raise Unknown("f-string")

    def ev_UnaryOp(self, n, env):
        v = self.ev(n.operand, env)
        if isinstance(v, Sym):
            return Sym("call", Sym("name", name="op." + type(n.op).__name__), args=[v])
        if isinstance(v, (list, tuple)):
            v = np.asarray(v, np.float64)
        if isinstance(n.op, ast.USub):
            return -v
        if isinstance(n.op, ast.UAdd):
            return +v
        if isinstance(n.op, ast.Not):
            return not v
        if isinstance(n.op, ast.Invert):
            return ~v
        raise Unknown("unary")

    def ev_BinOp(self, n, env):
        a, b = self.ev(n.left, env), self.ev(n.right, env)
        op = type(n.op).__name__
        if isinstance(a, Sym) or isinstance(b, Sym):
            return Sym("call", Sym("name", name="op." + op), args=[a, b])
        if isinstance(n.op, ast.MatMult):
            if isinstance(a, Xf):
                return a @ b
            raise Unknown("matmul")
        if isinstance(a, Xf) or isinstance(b, Xf):
            raise Unknown("transform arithmetic")
        if isinstance(n.op, ast.Mult) and isinstance(a, (list, tuple)) and isinstance(b, int):
            return a * b
        if isinstance(n.op, ast.Add) and isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
            return list(a) + list(b)
        if isinstance(n.op, ast.Add) and isinstance(a, str) and isinstance(b, str):
            return a + b
        if isinstance(n.op, ast.Mod) and isinstance(a, str):
            raise Unknown("string format")
        if isinstance(a, (list, tuple)):
            a = np.asarray(a, np.float64)
        if isinstance(b, (list, tuple)):
            b = np.asarray(b, np.float64)
        fn = {"Add": lambda: a + b, "Sub": lambda: a - b, "Mult": lambda: a * b, "Div": lambda: a / b, "Pow": lambda: a ** b,
              "FloorDiv": lambda: a // b, "Mod": lambda: a % b, "BitOr": lambda: a | b, "BitAnd": lambda: a & b,
              "LShift": lambda: a << b, "RShift": lambda: a >> b, "BitXor": lambda: a ^ b}.get(op)
        if fn is None:
            raise Unknown(op)
        with np.errstate(all="ignore"):
            return fn()

    def ev_BoolOp(self, n, env):
        vals = [self.ev(v, env) for v in n.values]
        if any(has_sym(v) for v in vals):
            return Sym("call", Sym("name", name="op." + type(n.op).__name__), args=vals)
        if isinstance(n.op, ast.And):
            r = True
            for v in vals:
                r = r and v
            return r
        r = False
        for v in vals:
            r = r or v
        return r

    def ev_Compare(self, n, env):
        left = self.ev(n.left, env)
        res = None
        for op, rn in zip(n.ops, n.comparators):
            right = self.ev(rn, env)
            name = type(op).__name__
            if isinstance(left, Sym) or isinstance(right, Sym) or has_sym(left) or has_sym(right):
                cur = Sym("call", Sym("name", name="op." + name), args=[left, right])
            else:
                a = np.asarray(left, np.float64) if isinstance(left, (list, tuple)) else left
                b = np.asarray(right, np.float64) if isinstance(right, (list, tuple)) else right
                fn = {"Eq": lambda: a == b, "NotEq": lambda: a != b, "Lt": lambda: a < b, "LtE": lambda: a <= b,
                      "Gt": lambda: a > b, "GtE": lambda: a >= b, "Is": lambda: a is b, "IsNot": lambda: a is not b,
                      "In": lambda: a in b, "NotIn": lambda: a not in b}[name]
                cur = fn()
            res = cur if res is None else (res and cur if not isinstance(res, Sym) and not isinstance(cur, Sym) else
                                           Sym("call", Sym("name", name="op.And"), args=[res, cur]))
            left = right
        return res

    def ev_IfExp(self, n, env):
        c = self.ev(n.test, env)
        if has_sym(c):
            raise Unknown("symbolic condition")
        return self.ev(n.body if c else n.orelse, env)

    def ev_Subscript(self, n, env):
        base = self.ev(n.value, env)
        sl = n.slice
        if isinstance(sl, ast.Slice):
            idx = slice(*(None if p is None else self.ev(p, env) for p in (sl.lower, sl.upper, sl.step)))
        else:
            idx = self.ev(sl, env)
        if isinstance(base, Sym):
            if has_sym(idx):
                raise Unknown("symbolic index")
            return Sym("item", base, name=idx if not isinstance(idx, slice) else [idx.start, idx.stop, idx.step])
        if isinstance(idx, np.generic):
            idx = idx.item()
        if isinstance(idx, tuple) and isinstance(base, np.ndarray):
            return base[idx]
        return base[idx]

    def ev_Attribute(self, n, env):
        # dotted names rooted at a module alias
        base = self.ev(n.value, env)
        a = n.attr
        if isinstance(base, Namespace):
            full = base.name + "." + a
            if base.name in ("dr", "dr.scalar", "drjit"):
                if a == "scalar":
                    return Namespace("dr.scalar")
                if base.name == "dr.scalar":
                    if a in ("ArrayXf", "Array3f", "Array2f", "Array4f"):
                        return FLOAT if a == "ArrayXf" else mi_vector
                    if a in ("ArrayXu", "ArrayXi"):
                        return UINT
                    raise Unknown(full)
                if a in DR:
                    return DR[a]
                if a in ("allclose", "all", "any", "isnan", "isfinite", "isinf", "none", "count"):
                    return Sym("name", name="dr." + a)
                raise Unknown(full)
            if base.name in ("np", "numpy"):
                if a in ("array", "tile", "linspace", "zeros", "ones", "sqrt", "cos", "sin", "pi", "float32", "float64", "arange",
                         "abs", "stack", "concatenate", "meshgrid", "full", "eye", "dot", "cross", "radians", "deg2rad", "tan",
                         "exp", "log", "inf", "allclose", "all", "any", "uint32", "int32", "column_stack", "vstack", "hstack",
                         "repeat", "mean", "sum", "max", "min", "isnan", "maximum", "minimum", "square", "arccos", "arcsin", "arctan2"):
                    return getattr(np, a)
                if a == "linalg":
                    return Namespace("np.linalg")
                raise Unknown(full)
            if base.name == "np.linalg":
                return getattr(np.linalg, a)
            if base.name == "math":
                return getattr(math, a)
            if base.name == "pytest":
                if a == "approx":
                    return lambda v, **kw: v
                raise Unknown(full)
            if base.name.startswith("mi"):
                if base.name in ("mi", "mi.scalar_rgb") and a in MI_NUMERIC:
                    return MI_NUMERIC[a]
                if base.name in ("mi", "mi.scalar_rgb") and a in ("Transform4f", "ScalarTransform4f"):
                    return XFTYPE
                return Namespace(full) if a[0].islower() and a in ("warp", "chi2", "math", "quad", "spline", "scalar_rgb", "test", "util", "mueller", "xml", "python") \
                    else Sym("name", name=full)
            raise Unknown(full)
        if base is XFTYPE:
            if a in XF_STATIC:
                return XF_STATIC[a]
            raise Unknown("Transform4f." + a)
        if isinstance(base, Xf):
            if a == "matrix":
                return base.m
            if a == "inverse":
                return lambda: Xf(np.linalg.inv(base.m))
            raise Unknown("transform." + a)
        if isinstance(base, Sym):
            if base.kind == "name":
                return Sym("name", name=base.name + "." + a)
            if a in base.attrs:
                return base.attrs[a]
            return Sym("attr", base, name=a)
        if isinstance(base, np.ndarray):
            comp = {"x": 0, "y": 1, "z": 2, "w": 3}
            if a in comp and base.ndim >= 1 and base.shape[0] > comp[a]:
                return base[comp[a]]
            if a in ("shape", "size", "T", "ndim"):
                return getattr(base, a)
            if a in ("astype", "reshape", "tolist", "copy", "flatten", "ravel"):
                return getattr(base, a)
            raise Unknown("ndarray." + a)
        if isinstance(base, dict) and a in ("keys", "values", "items", "get", "copy"):
            return getattr(base, a)
        if isinstance(base, (list, tuple)):
            comp = {"x": 0, "y": 1, "z": 2, "w": 3}
            if a in comp:
                return base[comp[a]]
            if a in ("append", "index", "count"):
                return getattr(base, a)
        if isinstance(base, str) and a in ("format", "join", "split", "strip", "replace"):
            return getattr(base, a)
        raise Unknown("attribute " + a)

    def ev_Call(self, n, env):
        fn = self.ev(n.func, env)
        args = []
        for a in n.args:
            if isinstance(a, ast.Starred):
                args.extend(self.ev(a.value, env))
            else:
                args.append(self.ev(a, env))
        kwargs = {}
        for k in n.keywords:
            if k.arg is None:
                kwargs.update(self.ev(k.value, env))
            else:
                kwargs[k.arg] = self.ev(k.value, env)
        if isinstance(fn, Sym):
            return Sym("call", fn, args=[freeze(a) for a in args], kwargs={k: freeze(v) for k, v in kwargs.items()})
        if isinstance(fn, FuncDef):
            return fn.call(self, args, kwargs)
        if fn is FLOAT:
            return f32(args[0]) if args else F32(0)
        if fn is UINT:
            return np.asarray(args[0], np.uint32) if args else np.uint32(0)
        if callable(fn):
            if has_sym(args) or has_sym(kwargs):
                # numeric helper applied to symbolic data (e.g. dr.abs(si.t - 1)): keep it symbolic
                nm = getattr(fn, "__name__", "fn")
                if fn is mi_vector:
                    nm = "vector"
                if not nm.startswith("dr."):
                    nm = "fn." + nm
                return Sym("call", Sym("name", name=nm), args=[freeze(a) for a in args], kwargs=kwargs)
            try:
                with np.errstate(all="ignore"):
                    return fn(*args, **kwargs)
            except Unknown:
                raise
            except Exception as e:      # noqa: BLE001 - anything a numeric helper cannot do is simply "unknown"
                raise Unknown("call failed: %s" % e)
        raise Unknown("call of non-callable")

    def ev_ListComp(self, n, env):
        if len(n.generators) != 1:
            raise Unknown("nested comprehension")
        g = n.generators[0]
        it = self.ev(g.iter, env)
        if isinstance(it, Sym):
            raise Unknown("symbolic iterable")
        out = []
        for v in it:
            e2 = dict(env)
            self.bind(g.target, v, e2)
            if all(self.ev(c, e2) for c in g.ifs):
                out.append(self.ev(n.elt, e2))
        return out

    def ev_Lambda(self, n, env):
        return FuncDef(n.args, [ast.Return(value=n.body)], env, self)

    # ------------------------------------------------------------------ statements
    def bind(self, target, value, env):
        if isinstance(target, ast.Name):
            env[target.id] = value
        elif isinstance(target, (ast.Tuple, ast.List)):
            if isinstance(value, Sym):
                for i, t in enumerate(target.elts):
                    self.bind(t, Sym("item", value, name=i), env)
            else:
                try:
                    vals = list(value)
                except TypeError:
                    raise Unknown("unpack of a non-sequence")
                if len(vals) != len(target.elts):
                    raise Unknown("unpack")
                for t, v in zip(target.elts, vals):
                    self.bind(t, v, env)
        elif isinstance(target, ast.Attribute):
            base = self.ev(target.value, env)
            if isinstance(base, Sym):
                base.attrs[target.attr] = freeze(value)
            elif isinstance(base, np.ndarray) and target.attr in "xyzw":
                base["xyzw".index(target.attr)] = value
            else:
                raise Unknown("attribute store")
        elif isinstance(target, ast.Subscript):
            base = self.ev(target.value, env)
            idx = self.ev(target.slice, env)
            if isinstance(base, (list, np.ndarray, dict)) and not has_sym(idx):
                base[idx] = value
            else:
                raise Unknown("subscript store")
        else:
            raise Unknown("bind target")

    def poison(self, target, env, why):
        for nd in ast.walk(target):
            if isinstance(nd, ast.Name):
                env[nd.id] = Unknown(why)

    def run_block(self, body, env):
        for st in body:
            r = self.run_stmt(st, env)
            if r is not None:
                return r
        return None

    def run_stmt(self, st, env):
        try:
            if isinstance(st, ast.Assign):
                try:
                    v = self.ev(st.value, env)
                except Unknown as e:
                    for t in st.targets:
                        self.poison(t, env, str(e))
                    return None
                for t in st.targets:
                    try:
                        self.bind(t, v, env)
                    except Unknown as e:
                        self.poison(t, env, str(e))
            elif isinstance(st, ast.AugAssign):
                try:
                    cur = self.ev(st.target, env)
                    v = self.ev(ast.BinOp(left=st.target, op=st.op, right=st.value), env)
                    del cur
                    self.bind(st.target, v, env)
                except Unknown as e:
                    self.poison(st.target, env, str(e))
            elif isinstance(st, ast.Expr):
                try:
                    v = self.ev(st.value, env)
                    if isinstance(v, Sym) and isinstance(st.value, ast.Call):
                        for a in list(st.value.args) + [k.value for k in st.value.keywords]:
                            if isinstance(a, ast.Name) and isinstance(env.get(a.id), (np.ndarray, list)):
                                env[a.id] = Unknown("possibly written by a Mitsuba call")
                    # a method called for its side effect on a Mitsuba object (`ib.put(pos=..., values=...)`): remember it on the object
                    if isinstance(v, Sym) and v.kind == "call" and isinstance(v.base, Sym) and v.base.kind == "attr" \
                            and isinstance(v.base.base, Sym) and v.base.base.kind == "call":
                        v.base.base.attrs.setdefault("__calls__", []).append(
                            {"method": v.base.name, "args": [freeze(a) for a in v.args], "kwargs": {k: freeze(x) for k, x in v.kwargs.items()}})
                except Unknown:
                    pass
            elif isinstance(st, ast.Assert):
                self.do_assert(st, env)
            elif isinstance(st, ast.For):
                try:
                    it = self.ev(st.iter, env)
                    if isinstance(it, Sym):
                        raise Unknown("symbolic iterable")
                    it = list(it)
                except (Unknown, TypeError):
                    self.poison(st.target, env, "loop")
                    return None
                for v in it[:MAX_LOOP]:
                    if self.n_test_records >= (700 if self.relpath in RELATIONS else MAX_RECORDS_PER_TEST):
                        break
                    try:
                        self.bind(st.target, v, env)
                    except Unknown:
                        break
                    r = self.run_block(st.body, env)
                    if r is not None:
                        return r
            elif isinstance(st, ast.If):
                try:
                    c = self.ev(st.test, env)
                except Unknown:
                    c = Sym("name", name="?")
                if has_sym(c):
                    # a branch taken only if Mitsuba says so (e.g. `if si_found:`): its assertions are conditional
                    self.cond_depth = getattr(self, "cond_depth", 0) + 1
                    self.cond_stack = getattr(self, "cond_stack", []) + [freeze(c)]
                    e2 = dict(env)
                    self.run_block(st.body, e2)
                    self.cond_depth -= 1
                    self.cond_stack = self.cond_stack[:-1]
                    for k in e2:
                        if k not in env or e2[k] is not env.get(k):
                            env[k] = Unknown("assigned under a symbolic condition")
                else:
                    return self.run_block(st.body if c else st.orelse, env)
            elif isinstance(st, ast.FunctionDef):
                env[st.name] = FuncDef(st.args, st.body, env, self)
            elif isinstance(st, ast.Return):
                return ("return", self.ev(st.value, env) if st.value is not None else None)
            elif isinstance(st, ast.With):
                return self.run_block(st.body, env)
            elif isinstance(st, (ast.Import, ast.ImportFrom)):
                self.do_import(st, env)
            # everything else (try, while, del, ...) is skipped
        except Unknown:
            pass
        return None

    def do_import(self, st, env):
        if isinstance(st, ast.Import):
            for a in st.names:
                nm = a.asname or a.name
                root = {"drjit": "dr", "mitsuba": "mi", "numpy": "np", "math": "math", "pytest": "pytest"}.get(a.name)
                if root:
                    env[nm] = Namespace(root)
        else:
            mod = st.module or ""
            for a in st.names:
                nm = a.asname or a.name
                if mod == "drjit.scalar" and a.name == "ArrayXf":
                    env[nm] = FLOAT
                elif mod == "drjit.scalar" and a.name in ("ArrayXu", "ArrayXi"):
                    env[nm] = UINT
                elif mod == "math":
                    env[nm] = getattr(math, a.name)
                elif mod == "mitsuba" and a.name in MI_NUMERIC:
                    env[nm] = MI_NUMERIC[a.name]
                elif mod == "mitsuba" and a.name in ("Transform4f", "ScalarTransform4f"):
                    env[nm] = XFTYPE
                elif mod.startswith("mitsuba"):
                    env[nm] = Sym("name", name="mi." + mod[len("mitsuba"):].lstrip(".") + ("." if len(mod) > 7 else "") + a.name)

    # ------------------------------------------------------------------ assertions -> records
    def do_assert(self, st, env):
        t = st.test
        rec = None
        try:
            if isinstance(t, ast.Call):
                fn = self.ev(t.func, env)
                if isinstance(fn, Sym) and fn.kind == "name" and fn.name == "dr.allclose" and len(t.args) >= 2:
                    a, b = self.ev(t.args[0], env), self.ev(t.args[1], env)
                    kw = {k.arg: self.ev(k.value, env) for k in t.keywords}
                    rec = dict(kind="allclose", lhs=a, rhs=b, rtol=kw.get("rtol", 1e-5), atol=kw.get("atol", 1e-8))
                elif fn is np.allclose and len(t.args) >= 2:
                    a, b = self.ev(t.args[0], env), self.ev(t.args[1], env)
                    kw = {k.arg: self.ev(k.value, env) for k in t.keywords}
                    rec = dict(kind="allclose", lhs=a, rhs=b, rtol=kw.get("rtol", 1e-5), atol=kw.get("atol", 1e-8))
                elif isinstance(fn, Sym) and fn.kind == "name" and fn.name in ("dr.all", "dr.any") and len(t.args) == 1:
                    inner = self.ev(t.args[0], env)
                    if isinstance(inner, Sym) and inner.kind == "call" and inner.base.kind == "name" and inner.base.name.startswith("op.") \
                            and len(inner.args) == 2:
                        rec = dict(kind=fn.name[3:] + "." + inner.base.name[3:], lhs=inner.args[0], rhs=inner.args[1])
                    else:
                        rec = dict(kind=fn.name[3:] + ".truth", lhs=inner, rhs=True)
                else:
                    v = self.ev(t, env)
                    rec = dict(kind="truth", lhs=v, rhs=True)
            elif isinstance(t, ast.Compare) and len(t.ops) == 1:
                a, b = self.ev(t.left, env), self.ev(t.comparators[0], env)
                rec = dict(kind=type(t.ops[0]).__name__, lhs=a, rhs=b)
            elif isinstance(t, ast.UnaryOp) and isinstance(t.op, ast.Not):
                v = self.ev(t.operand, env)
                rec = dict(kind="truth", lhs=v, rhs=False)
            else:
                v = self.ev(t, env)
                rec = dict(kind="truth", lhs=v, rhs=True)
        except Unknown:
            return
        both = has_sym(rec["lhs"]) and has_sym(rec["rhs"]) if rec is not None else False
        if rec is None or (has_sym(rec["lhs"]) == has_sym(rec["rhs"]) and not (both and self.relpath in RELATIONS)):
            return      # a known answer relates something Mitsuba computes (one side) to plain numbers (the other side)
        if rec["kind"] in ("Is", "IsNot"):
            return      # `x is not None`
        if "chi2" in json.dumps(to_json(rec["lhs"]))[:4000]:
            return      # statistical tests need the reference's own sampler loop
        if getattr(self, "cond_depth", 0) > 1:
            return      # nested Mitsuba-dependent conditions (finite-difference checks inside `if hit: if hit2:`)
        rec.update(file=self.relpath, line=st.lineno, test=self.test, params=self.binding)
        if getattr(self, "cond_depth", 0) > 0:
            rec["conditions"] = list(self.cond_stack)     # the assertion only applies when Mitsuba makes all of these true
        try:
            rec = to_json(rec)
        except Unknown:
            return
        self.records.append(rec)
        self.n_test_records += 1


class FuncDef:
    def __init__(self, args, body, env, interp):
        self.args, self.body, self.env, self.interp = args, body, env, interp

    def call(self, interp, args, kwargs):
        e = dict(self.env)
        names = [a.arg for a in self.args.args]
        defaults = self.args.defaults
        for nm, d in zip(names[len(names) - len(defaults):], defaults):
            e[nm] = interp.ev(d, self.env)
        for nm, v in zip(names, args):
            e[nm] = v
        for k, v in kwargs.items():
            e[k] = v
        for nm in names:
            if nm not in e:
                raise Unknown("missing argument " + nm)
        r = interp.run_block(self.body, e)
        return r[1] if r is not None else None


def to_json(v):
    if isinstance(v, Sym):
        d = {"sym": v.kind}
        if v.kind == "name":
            d["name"] = v.name
        elif v.kind == "call":
            d["fn"] = to_json(v.base)
            d["args"] = [to_json(a) for a in v.args]
            if v.kwargs:
                d["kwargs"] = {k: to_json(x) for k, x in v.kwargs.items()}
        elif v.kind == "attr":
            d["of"], d["name"] = to_json(v.base), v.name
        elif v.kind == "item":
            d["of"], d["index"] = to_json(v.base), to_json(v.name)
        if v.attrs:
            d["set"] = {k: to_json(x) for k, x in v.attrs.items()}
        return d
    if isinstance(v, Xf):
        return {"transform": v.m.tolist()}
    if isinstance(v, np.ndarray):
        if v.ndim >= 2:
            return {"nd": v.astype(np.float64).tolist()}      # a numpy matrix (row-major), as opposed to a Dr.Jit array of components
        if v.dtype == np.float32:
            return {"f32": v.tolist()}
        if v.dtype.kind in "iu":
            return v.tolist()
        if v.dtype.kind == "b":
            return v.tolist()
        if v.ndim >= 2:
            return {"nd": v.astype(np.float64).tolist()}      # a numpy matrix (row-major), as opposed to a Dr.Jit array of components
        return v.astype(np.float64).tolist()
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, (list, tuple)):
        return [to_json(x) for x in v]
    if isinstance(v, dict):
        return {str(k): to_json(x) for k, x in v.items()}
    if isinstance(v, (int, float, bool, str)) or v is None:
        return v
    if v is FLOAT:
        return {"type": "Float"}
    if v is UINT:
        return {"type": "UInt32"}
    if isinstance(v, Namespace):
        return {"sym": "name", "name": v.name}
    raise Unknown("unserialisable %r" % type(v))


def parametrizations(fn, interp, env):
    """the bindings of the test's pytest.mark.parametrize decorators (cartesian product, in decorator order)"""
    axes = []
    for d in fn.decorator_list:
        if isinstance(d, ast.Call) and isinstance(d.func, ast.Attribute) and d.func.attr == "parametrize" and len(d.args) >= 2:
            try:
                names = interp.ev(d.args[0], env)
                vals = interp.ev(d.args[1], env)
                if isinstance(vals, Sym):
                    continue
                vals = list(vals)
            except Unknown:
                continue
            names = [s.strip() for s in names.split(",")] if isinstance(names, str) else list(names)
            axes.append((names, vals))
    if not axes:
        return [{}]
    combos = []
    for pick in itertools.product(*[a[1] for a in axes]):
        b = {}
        for (names, _), v in zip(axes, pick):
            if len(names) == 1:
                b[names[0]] = v
            else:
                for nm, x in zip(names, v):
                    b[nm] = x
        combos.append(b)
        if len(combos) >= MAX_PARAM_COMBOS:
            break
    return combos


def harvest(relpath):
    path = os.path.join(REF, relpath)
    if not os.path.exists(path):
        return []
    tree = ast.parse(open(path).read())
    it = Interp(relpath)
    env = {}
    # module level: imports, constants, helper functions
    for st in tree.body:
        if isinstance(st, ast.FunctionDef) and st.name.startswith("test"):
            continue
        it.run_stmt(st, env)
    for st in tree.body:
        if not (isinstance(st, ast.FunctionDef) and st.name.startswith("test")):
            continue
        for binding in parametrizations(st, it, env):
            e = dict(env)
            for a in st.args.args:       # fixtures (variant_*, tmpdir, ...) are unknown
                e[a.arg] = Unknown("fixture")
            clean = {}
            for k, v in binding.items():
                e[k] = v
                try:
                    clean[k] = to_json(v)
                except Unknown:
                    clean[k] = None
            it.test, it.binding, it.n_test_records = st.name, clean or None, 0
            it.cond_depth, it.cond_stack = 0, []
            it.run_block(st.body, e)
    return it.records


def main():
    out = []
    for f in FILES:
        recs = harvest(f)
        print("%-48s %5d records" % (f, len(recs)))
        out.extend(recs)
    dst = os.path.join(HERE, "reference_kats.json.gz")
    import gzip
    with gzip.GzipFile(dst, "wb", mtime=0) as gz, io.TextIOWrapper(gz, encoding="utf-8") as fh:
        json.dump({"source": "juhyeonkim95/Mitsuba3DopplerToF @ 2024_08_07, values harvested by tests/golden/extract_reference_kats.py",
                   "records": out}, fh, separators=(",", ":"))
    print("%d records -> %s (%.1f KiB)" % (len(out), dst, os.path.getsize(dst) / 1024))


if __name__ == "__main__":
    main()
