"""`cylinder` shape (src/shapes/cylinder.cpp; SURVEY 8(f)-3 leftovers): loader parity with the oracle's independent loader (composed transform,
radius, flip), analytic intersections, lanes bit-exact on the GPU."""
import os

import numpy as np
import pytest

from conftest import SCENES

NCPU = min(16, os.cpu_count() or 1)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_cylinder_loader_matches_the_oracle(mi, orc):
    path = os.path.join(SCENES, "cornell_cylinders.xml")
    sc, osc = mi.load_file(path), orc.Scene(path, {})
    xf = sc.export(1).reshape(-1, 32)
    cyl = [(i, s) for i, s in enumerate(osc.flat.shapes) if s["kind"] == 4]
    assert len(cyl) == 3
    for i, s in cyl:
        assert np.array_equal(bits(xf[i, :16]), bits(s["to_world"].reshape(-1))) and np.array_equal(bits(xf[i, 16:]), bits(s["to_object"].reshape(-1))), i
    pillar = cyl[0][1]["cylinder_baked"]
    assert abs(pillar[0] - 0.28) < 1e-6 and abs(pillar[1] - 1.3) < 1e-6 and pillar[3] == 0
    squat = cyl[2][1]["cylinder_baked"]
    assert abs(squat[0] - 0.3) < 1e-6 and abs(squat[1] - 0.25) < 1e-6 and squat[3] == 1     # flip_normals = true


def test_cylinder_intersections_are_analytic(orc):
    """a unit-radius cylinder along z from (0, 0, 0) to (0, 0, 2): hits from outside, from inside, through the open ends, grazing and missing rays"""
    import ctypes as C
    xml = ('<scene version="3.0.0"><integrator type="path"/><sensor type="perspective"><float name="fov" value="40"/>'
           '<film type="hdrfilm"><integer name="width" value="4"/><integer name="height" value="4"/></film></sensor>'
           '<shape type="cylinder"><point name="p1" x="0" y="0" z="2"/></shape></scene>')
    sc = orc.Scene(xml, is_string=True)
    L = orc.lib()

    def hit(o, d, maxt=1e30):
        out, ids = np.zeros(25, np.float32), np.zeros(3, np.int32)
        o, d = np.asarray(o, np.float32), np.asarray(d, np.float32)
        ok = L.orc_kat_ray_intersect(C.byref(sc.c), o.ctypes.data, d.ctypes.data, C.c_float(0), C.c_float(maxt), out.ctypes.data, ids.ctypes.data)
        return bool(ok), out
    ok, r = hit([3, 0, 1], [-1, 0, 0])
    assert ok and abs(r[0] - 2) < 1e-6 and np.allclose(r[4:7], [1, 0, 0], atol=1e-6)          # outside: the near wall, normal towards the ray
    ok, r = hit([0, 0, 1], [0, 1, 0])
    assert ok and abs(r[0] - 1) < 1e-6 and np.allclose(r[4:7], [0, 1, 0], atol=1e-6)          # inside: the far wall, the (outward) normal
    ok, r = hit([0.5, 0, -1], [0, 0, 1])
    assert not ok                                                                                # along the axis through the open ends
    ok, r = hit([3, 0, 2.5], [-1, 0, 0])
    assert not ok                                                                                # above the upper end
    ok, r = hit([3, 0, 1.99], [-1, 0, -0.2])
    assert ok and abs(r[0] - 2) < 1e-5                                                           # enters below the rim (t in units of the unnormalised d)
    ok, r = hit([3, 0, 2.5], [-1, 0, -0.15])
    assert ok and abs(r[0] - 4) < 1e-5                                                                     # misses the near wall above the rim, hits the far wall from inside
    ok, r = hit([3, 1.0001, 1], [-1, 0, 0])
    assert not ok                                                                                # grazing outside
    ok, r = hit([3, 0, 1], [-1, 0, 0], maxt=1.5)
    assert not ok                                                                                # maxt before the wall
    ok, r = hit([0, 0, 1], [1, 0, 0], maxt=0.5)
    assert not ok                                                                                # inside, both roots beyond maxt


def test_cylinder_errors(mi):
    xml = ('<scene version="3.0.0"><integrator type="path"/><sensor type="perspective"><float name="fov" value="40"/>'
           '<film type="hdrfilm"><integer name="width" value="4"/><integer name="height" value="4"/></film></sensor>'
           '<shape type="cylinder">%s</shape></scene>')
    mi.load_string(xml % "")
    with pytest.raises(mi.DtofError, match="area emitters on cylinders"):
        mi.load_string(xml % '<emitter type="area"><rgb name="radiance" value="1"/></emitter>')
    with pytest.raises(mi.DtofError, match="unreferenced property"):
        mi.load_string(xml % '<float name="height" value="2"/>')
