"""Restated cases of reference tests that cannot be harvested mechanically (they build their inputs with mi.load_dict / mi.Mesh):

  * src/shapes/tests/test_shapegroup.py   test01_create (bounds of a group), test02_error (the four nesting / instancing errors)
  * src/render/tests/test_scene.py        test01_emitter_checks (emitter counts, "can be only be attached to a single shape")
  * src/render/tests/test_mesh.py         test02_ply_triangle, test03_ply_computed_normals, test04_normal_weighting_scheme,
                                          test10_ray_intersect_preliminary (rectangle mesh: t, p, uv, dp_du, dp_dv)
  * src/core/tests/test_distr_2d.py       test01_sample_inverse_discrete for Hierarchical2D0 (the warp of the envmap emitter)

Each case states the reference test's INPUT as the XML (or file) equivalent of its load_dict call and the EXPECTED numbers / messages the test
asserts; the files the reference's tests read from its empty `resources/data` submodule are replaced by fixtures written here with the
properties the assertions rely on.  Product loader through the C ABI (no GPU needed for loading) and the oracle's loader side by side."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

SCENE = '<scene version="3.0.0">%s</scene>'
SENSOR = ('<sensor type="perspective"><film type="hdrfilm"><integer name="width" value="4"/><integer name="height" value="4"/></film>'
          '<sampler type="independent"><integer name="sample_count" value="1"/></sampler></sensor>')


def both_loaders(mi, orc):
    return [("product", lambda xml: mi.load_string(xml)), ("oracle", lambda xml: orc.Scene(xml, {}, is_string=True))]


# ------------------------------------------------------------------------------------------------ test_shapegroup.py
GROUP = ('<shape type="shapegroup" id="g">'
         '<shape type="sphere"><float name="radius" value="1"/><transform name="to_world"><translate x="-2"/></transform></shape>'
         '<shape type="sphere"><float name="radius" value="1"/><transform name="to_world"><translate x="2"/></transform></shape>'
         '</shape>')


def test_shapegroup_create_two_spheres(mi, orc):
    """test_shapegroup.py:6-28: a group of two unit spheres at x = -2 and x = +2: primitive_count 2, effective_primitive_count 0 (a group alone puts
    nothing into the scene), bounds [-3, -1, -1] .. [3, 1, 1]"""
    sc = mi.load_string(SCENE % GROUP)
    i = sc.info()
    assert i["n_groups"] == 1 and i["n_shapes"] == 2 and i["n_objects"] == 0          # two primitives in the group, none instantiated
    o = orc.Scene(SCENE % GROUP, {}, is_string=True)
    assert len(o.flat.groups) == 1 and o.flat.groups[0]["n_shapes"] == 2 and len(o.flat.objects) == 0
    # the group's bounds = the union of its shapes' bounds (kind 8: centre, radius of each sphere)
    sph = np.asarray(sc.export(8), dtype=np.float64).reshape(-1, 6)
    lo = (sph[:, :3] - sph[:, 3:4]).min(axis=0); hi = (sph[:, :3] + sph[:, 3:4]).max(axis=0)
    np.testing.assert_allclose(lo, [-3, -1, -1], atol=1e-6); np.testing.assert_allclose(hi, [3, 1, 1], atol=1e-6)
    np.testing.assert_allclose((lo + hi) / 2, [0, 0, 0], atol=1e-6)
    # instantiated once, the scene holds one object
    sc2 = mi.load_string(SCENE % (GROUP + '<shape type="instance"><ref id="g"/></shape>'))
    assert sc2.info()["n_objects"] == 1


@pytest.mark.parametrize("inner,message", [
    ('<shape type="instance"><shape type="shapegroup"><shape type="sphere"/></shape></shape>', "Nested instancing is not permitted"),
    ('<shape type="shapegroup"><shape type="sphere"/></shape>', "Nested ShapeGroup is not permitted"),
    ('<shape type="sphere"><emitter type="area"/></shape>', "Instancing of emitters is not supported"),
    ('<shape type="sphere"><sensor type="perspective"/></shape>', "Instancing of sensors is not supported"),
])
def test_shapegroup_errors(mi, orc, inner, message):
    """test_shapegroup.py:31-73: what a shapegroup refuses, with the reference's messages"""
    xml = SCENE % ('<shape type="shapegroup">%s</shape>' % inner)
    for name, load in both_loaders(mi, orc):
        with pytest.raises(Exception, match=".*%s.*" % message):
            load(xml)


# ------------------------------------------------------------------------------------------------ test_scene.py
@pytest.fixture()
def rect_obj(tmp_path):
    """stands in for resources/data/tests/obj/rectangle_uv.obj (absent submodule): a two-triangle rectangle with texture coordinates"""
    p = tmp_path / "rectangle_uv.obj"
    p.write_text("v -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3\nf 1/1 3/3 4/4\n")
    return str(p)


def test_scene_emitter_checks(mi, orc, rect_obj):
    """test_scene.py:9-49: how many emitters a scene ends up with, and that one area emitter cannot serve two shapes"""
    shape = '<shape type="obj"><string name="filename" value="%s"/>%%s</shape>' % rect_obj
    cases = [
        ('<emitter type="constant"/>', 1, None),
        (shape % '<emitter type="area"/>', 1, None),
        ('<emitter type="area" id="my_emitter"/>' + shape % '<ref id="my_emitter"/>', 1, None),
        ('<emitter type="area" id="my_emitter"/>' + shape % '<ref id="my_emitter"/>' + shape % '<ref id="my_emitter"/>', 2, "can be only be attached to a single shape"),
        ('<emitter type="constant"/><emitter type="point"/><emitter type="area" id="my_emitter"/>'
         + shape % '<emitter type="area" id="my_inner_emitter"/>' + shape % '<ref id="my_emitter"/>', 4, None),
    ]
    for xml, count, error in cases:
        for name, load in both_loaders(mi, orc):
            if error is None:
                sc = load(SCENE % xml)
                n = sc.info()["n_emitters"] if name == "product" else len(sc.flat.emitters)
                assert n == count, (name, xml, n)
            else:
                with pytest.raises(Exception, match=".*%s.*" % error):
                    load(SCENE % xml)


# ------------------------------------------------------------------------------------------------ test_mesh.py
TRIANGLE_PLY = "ply\nformat ascii 1.0\ncomment this file contains a triangle\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n" \
               "element face 1\nproperty list uchar int vertex_index\nend_header\n0 0 0\n0 0 1\n0 1 0\n3 0 1 2\n"      # = src/render/tests/data/triangle.ply


def mesh_arrays(sc):
    pos = np.asarray(sc.export(4), dtype=np.float32).reshape(-1, 3)
    nrm = np.asarray(sc.export(5), dtype=np.float32).reshape(-1, 3)
    faces = np.asarray(sc.export(7), dtype=np.float32).view(np.uint32).reshape(-1, 3)
    return pos, nrm, faces


def test_ply_triangle_and_computed_normals(mi, orc, tmp_path):
    """test_mesh.py:34-70: triangle.ply -- 3 vertices, 1 face; with face_normals no vertex normals; without, normals are computed: [-1, 0, 0] at every vertex"""
    (tmp_path / "triangle.ply").write_text(TRIANGLE_PLY)
    for face_normals in (True, False):
        xml = SCENE % ('<shape type="ply"><string name="filename" value="%s"/><boolean name="face_normals" value="%s"/></shape>'
                       % (tmp_path / "triangle.ply", "true" if face_normals else "false"))
        sc = mi.load_string(xml)
        pos, nrm, faces = mesh_arrays(sc)
        assert pos.size == 9 and faces.size == 3
        np.testing.assert_allclose(pos, [[0, 0, 0], [0, 0, 1], [0, 1, 0]])
        assert faces.ravel().tolist() == [0, 1, 2]
        o = orc.Scene(xml, {}, is_string=True)
        m = o.flat.shapes[0]          # the baked arrays (orc_bake_mesh) sit beside the raw ones
        np.testing.assert_allclose(np.asarray(m["positions"]).reshape(-1, 3), pos)
        if face_normals:
            assert not np.any(nrm) and (m["normals"] is None or not np.any(m["normals"]))     # has_vertex_normals() == False: the normals stay unset
        else:
            np.testing.assert_allclose(nrm, [[-1, 0, 0]] * 3, atol=1e-6)
            np.testing.assert_allclose(np.asarray(m["normals"]).reshape(-1, 3), nrm, atol=0)


def test_normal_weighting_scheme(mi, orc, tmp_path):
    """test_mesh.py:73-98: vertex normals are weighted by the angle a face subtends at the vertex -- the shared vertex of a face in the z = 0 plane
    (angle pi / 2 there, normal -z) and a face in the y = 0 plane (angle acos(3 / 5), normal +y)"""
    a, b = 1.0, 0.5
    verts = [(0, 0, 0), (-a, 1, 0), (a, 1, 0), (-b, 0, 1), (b, 0, 1)]
    ply = "ply\nformat ascii 1.0\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\nelement face 2\nproperty list uchar int vertex_index\nend_header\n"
    ply += "".join("%g %g %g\n" % v for v in verts) + "3 0 1 2\n3 0 3 4\n"
    (tmp_path / "w.ply").write_text(ply)
    xml = SCENE % ('<shape type="ply"><string name="filename" value="%s"/></shape>' % (tmp_path / "w.ply"))
    n0, n1 = np.array([0.0, 0.0, -1.0]), np.array([0.0, 1.0, 0.0])
    n2 = n0 * (math.pi / 2.0) + n1 * math.acos(3.0 / 5.0); n2 /= np.linalg.norm(n2)
    expected = np.vstack([n2, n0, n0, n1, n1])
    pos, nrm, faces = mesh_arrays(mi.load_string(xml))
    np.testing.assert_allclose(nrm, expected, atol=5e-4)
    o = orc.Scene(xml, {}, is_string=True)
    np.testing.assert_allclose(np.asarray(o.flat.shapes[0]["normals"]).reshape(-1, 3), expected, atol=5e-4)


# texture coordinates as an OBJ exporter writes them: the loader flips v (flip_tex_coords, obj.cpp), after which uv = (p + 1) / 2
RECT_OBJ = "v -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nvt 0 1\nvt 1 1\nvt 1 0\nvt 0 0\nvn 0 0 1\nf 2/2/1 4/4/1 1/1/1\nf 2/2/1 3/3/1 4/4/1\n"


@pytest.mark.gpu
def test_rectangle_mesh_ray_intersect(mi, tmp_path):
    """test_mesh.py:258-292 (resources/data/common/meshes/rectangle.obj is absent; this rectangle has its triangulation -- the diagonal from (1, -1) to
    (-1, 1), the triangle below it first): a ray down the z axis at (-0.3, -0.3) meets face 0 at t = 10 with prim_uv (0.35, 0.3), si.p on the plane,
    si.uv = (0.35, 0.35), dp_du = (2, 0, 0), dp_dv = (0, 2, 0); at (0.3, 0.3) face 1 with prim_uv (0.3, 0.35), uv (0.65, 0.65)"""
    (tmp_path / "rectangle.obj").write_text(RECT_OBJ)
    sc = mi.load_string(SCENE % (SENSOR + '<shape type="obj" id="rect"><string name="filename" value="%s"/></shape>' % (tmp_path / "rectangle.obj")))
    for (x, y), prim, prim_uv, uv in (((-0.3, -0.3), 0, (0.35, 0.3), (0.35, 0.35)), ((0.3, 0.3), 1, (0.3, 0.35), (0.65, 0.65))):
        si = sc.ray_intersect([[x, y, -10.0]], [[0.0, 0.0, 1.0]])
        assert bool(si["valid"][0])
        np.testing.assert_allclose(si["t"][0], 10, rtol=1e-6)
        assert int(si["prim_index"][0]) == prim
        np.testing.assert_allclose(si["prim_uv"][0], prim_uv, atol=1e-6)
        np.testing.assert_allclose(si["p"][0], [x, y, 0.0], atol=1e-6)
        np.testing.assert_allclose(si["uv"][0], uv, atol=1e-6)
        # dp_du = (2, 0, 0), dp_dv = (0, 2, 0): the shading frame the interaction is finalised with has s along dp_du, n = +z
        np.testing.assert_allclose(si["sh_s"][0], [1.0, 0.0, 0.0], atol=1e-6)
        np.testing.assert_allclose(si["sh_t"][0], [0.0, 1.0, 0.0], atol=1e-6)
        np.testing.assert_allclose(si["n"][0], [0.0, 0.0, 1.0], atol=1e-6)


# ------------------------------------------------------------------------------------------------ test_distr_2d.py
class Hier2D:
    """Hierarchical2D<Float, 0> of the oracle over a plain grid (oracle/dtof_oracle.c: orc_hier2d_create; the envmap emitter builds the same warp)"""

    def __init__(self, orc, values, normalize):
        self.L = orc.lib()
        self.L.orc_hier2d_create.restype = C.c_void_p
        self.L.orc_hier2d_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        self.L.orc_envmap_warp_sample.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        self.L.orc_envmap_warp_eval.argtypes = [C.c_void_p, C.c_float, C.c_float]
        self.L.orc_envmap_warp_eval.restype = C.c_float
        self.L.orc_envmap_free.argtypes = [C.c_void_p]
        v = np.ascontiguousarray(values, dtype=np.float32)
        self.h = self.L.orc_hier2d_create(v.ctypes.data, v.shape[1], v.shape[0], int(normalize))
        assert self.h

    def sample(self, s):
        out = np.zeros(3, np.float32)
        self.L.orc_envmap_warp_sample(self.h, C.c_float(s[0]), C.c_float(s[1]), out.ctypes.data)
        return out[:2].astype(np.float64), float(out[2])

    def eval(self, p):
        return float(self.L.orc_envmap_warp_eval(self.h, C.c_float(p[0]), C.c_float(p[1])))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_envmap_free(self.h)


def bilinear_to_square(v00, v10, v01, v11, p):
    """the inverse of sampling a bilinear density on the unit square (what the reference test computes with mi.warp.bilinear_to_square): the sample that
    maps to position p, and the density there (normalised to a unit integral) -- marginal CDF in y, then the conditional CDF in x"""
    x, y = p
    r0, r1 = v00 + v10, v01 + v11
    sy = y * (2 * r0 + y * (r1 - r0)) / (r0 + r1)                      # integral of the row weight r0 + y (r1 - r0), normalised
    c0, c1 = v00 + y * (v01 - v00), v10 + y * (v11 - v10)
    sx = x * (2 * c0 + x * (c1 - c0)) / (c0 + c1)
    return (sx, sy), ((1 - x) * c0 + x * c1) / (0.25 * (v00 + v10 + v01 + v11))


@pytest.mark.parametrize("normalize", [True, False])
def test_hierarchical2d_spot_checks(orc, normalize):
    """test_distr_2d.py:7-49 for Hierarchical2D0: a 3 x 2 grid (two patches, odd number of columns), corners, the transition between the patches, a position
    inside each patch, sample -> eval consistency (atol 1e-6 as in the reference test)"""
    ref = np.array([[1, 2, 5], [9, 7, 2]], dtype=np.float32)
    intg = np.array([19, 16]) / 35
    d = Hier2D(orc, ref, normalize)
    s = 35 / 8.0 if not normalize else 1

    def close(got, want):
        (p, pdf), (wp, wpdf) = got, want
        np.testing.assert_allclose(p, wp, atol=1e-6); assert abs(pdf - wpdf) < 1e-6 * max(1.0, abs(wpdf)) + 1e-6

    close(d.sample([0, 0]), ([0, 0], s * 8.0 / 35.0))
    close(d.sample([1, 1]), ([1, 1], s * 16.0 / 35.0))
    close(d.sample([intg[0], 0]), ([0.5, 0], s * 16.0 / 35.0))
    assert abs(d.eval([0, 0]) - s * 8.0 / 35.0) < 1e-6 and abs(d.eval([1, 1]) - s * 16.0 / 35.0) < 1e-6 and abs(d.eval([0.5, 0]) - s * 16.0 / 35.0) < 1e-6
    # a position inside each patch: the reference multiplies bilinear_to_square's density by 8 / 35 x s, which is the bilinear interpolant of the grid
    # values x 8 / 35 x s (8 / 35 = 2 patches / the sum of the patch averages 19 / 4 + 16 / 4)
    sample, _ = bilinear_to_square(1, 2, 9, 7, [0.4, 0.3])
    sample = (sample[0] * intg[0], sample[1])
    want = ((1 - 0.4) * (1 + 0.3 * 8) + 0.4 * (2 + 0.3 * 5)) * 8.0 / 35.0 * s
    close(d.sample(sample), ([0.2, 0.3], want)); assert abs(d.eval([0.2, 0.3]) - want) < 1e-6
    sample, _ = bilinear_to_square(2, 5, 7, 2, [0.4, 0.3])
    sample = (sample[0] * intg[1] + intg[0], sample[1])
    want = ((1 - 0.4) * (2 + 0.3 * 5) + 0.4 * (5 - 0.3 * 3)) * 8.0 / 35.0 * s
    close(d.sample(sample), ([0.7, 0.3], want)); assert abs(d.eval([0.7, 0.3]) - want) < 1e-6


@pytest.mark.parametrize("normalize", [True, False])
def test_hierarchical2d_sample_agrees_with_eval(orc, normalize):
    """test_distr_2d.py:87-116 (forward half; the envmap emitter never inverts its warp): on random grids the density sample() reports is eval() at the
    position it returns (atol 1e-4), positions stay in the unit square, a constant grid is the identity warp"""
    rng = np.random.default_rng(0)
    for i in range(10):
        shape = rng.integers(2, 8, 2)
        values = rng.random(shape) * 10 if i < 9 else np.ones(shape)
        d = Hier2D(orc, values.astype(np.float32), normalize)
        for _ in range(10):
            u = rng.random(2)
            p, pdf = d.sample(u)
            assert np.all(p >= 0) and np.all(p <= 1)
            assert abs(pdf - d.eval(p)) < 1e-4 * max(1.0, pdf)
            if i == 9:
                np.testing.assert_allclose(p, u, atol=1e-5)
                assert abs(pdf - 1.0) < 1e-5
