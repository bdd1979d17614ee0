"""The analytic `disk` shape (src/shapes/disk.cpp): plane test with a circular bound, polar shading frame, disk area lights.
CPU: oracle sanity; GPU: per-lane parity."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENES

sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_scenes as ms  # noqa: E402

NCPU = os.cpu_count() or 1


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def lit_room(light):
    s = ms.HEADER.format(spp=16, res=32, tsm="antithetic", shift="0.5") + ms.SENSOR.format(fov="19.5", cam=ms.CAM)
    for b in ms.BSDFS:
        s += ms.bsdf(*b)
    for name, m, b in ms.WALLS:
        s += ms.rect(name, m, b)
    return s + light + "</scene>\n"


LIGHT = ('\t<shape type="%s" id="L"><transform name="to_world"><scale x="%s" y="%s" z="1" /><rotate x="1" angle="90" />'
         '<translate x="0" y="1.98" z="0" /></transform><emitter type="area"><rgb name="radiance" value="17, 12, 4" /></emitter></shape>\n')


def test_disk_semantics(mi, orc, tmp_path):
    """(1) analytic hits through the oracle's intersect entry: inside / outside the unit circle of a scaled, translated disk, t along the
    ray, occlusion; (2) a disk light and a rectangle light of the same area and radiance light the room equally in expectation
    (Disk::surface_area of an ellipse, sampling, pdf); (3) loader: flip_normals and the error for unknown shapes name `disk`."""
    import ctypes as C
    p = str(tmp_path / "one.xml")
    open(p, "w").write(lit_room('\t<shape type="disk" id="D"><transform name="to_world"><scale x="0.5" y="0.25" z="1" /><translate x="0.2" y="1.0" z="-0.4" />'
                                '</transform></shape>\n'))
    osc = orc.Scene(p, {})
    L = orc.lib()
    disk_obj = len(osc.flat.objects) - 1

    def hit(o, d, maxt=1e9):
        out, ids = (C.c_float * 3)(), (C.c_int32 * 3)()
        ok = L.orc_intersect(C.byref(osc.c), (C.c_float * 3)(*o), (C.c_float * 3)(*d), 0.0, maxt, out, ids)
        return (ok, out[0], ids[0])
    ok, t, obj = hit((0.2, 1.0, 2.0), (0, 0, -1))
    assert ok and obj == disk_obj and abs(t - 2.4) < 1e-6                      # the centre
    assert hit((0.2 + 0.49, 1.0, 2.0), (0, 0, -1))[2] == disk_obj               # just inside the long semi-axis
    assert hit((0.2 + 0.51, 1.0, 2.0), (0, 0, -1))[2] != disk_obj               # just outside: the back wall behind it
    assert hit((0.2, 1.0 + 0.24, 2.0), (0, 0, -1))[2] == disk_obj and hit((0.2, 1.0 + 0.26, 2.0), (0, 0, -1))[2] != disk_obj
    assert hit((0.2 + 0.4, 1.0 + 0.2, 2.0), (0, 0, -1))[2] != disk_obj          # inside the bounding rectangle, outside the ellipse
    assert L.orc_occluded(C.byref(osc.c), (C.c_float * 3)(0.2, 1.0, 2.0), (C.c_float * 3)(0, 0, -1), 0.0, 2.5) == 1
    assert L.orc_occluded(C.byref(osc.c), (C.c_float * 3)(0.2, 1.0, 2.0), (C.c_float * 3)(0, 0, -1), 0.0, 2.3) == 0

    def render(xml, name):
        q = str(tmp_path / name)
        open(q, "w").write(xml)
        s = orc.Scene(q, dict(resx=16, resy=16))
        pd = s.params(integrator=dict(type="path", max_depth=4))
        return np.mean([s.render(pd, seed=k, spp=256, threads=NCPU)[0] for k in range(2)], axis=0)
    a, b = 0.3, 0.2                                                             # ellipse semi-axes; the rectangle with the same area and aspect
    k = float(np.sqrt(np.pi) / 2)
    disk = render(lit_room(LIGHT % ("disk", a, b)), "disk.xml")
    rect = render(lit_room(LIGHT % ("rectangle", a * k, b * k)), "rect.xml")
    assert abs(disk.mean() - rect.mean()) < 0.03 * rect.mean(), (disk.mean(), rect.mean())
    sc = mi.load_file(os.path.join(SCENES, "cornell_disk.xml"))
    assert sc.info()["n_shapes"] == 8 and sc.info()["n_emitters"] == 1
    with pytest.raises(mi.DtofError, match="supported: rectangle, disk"):
        mi.load_string(lit_room('\t<shape type="sdfgrid" />\n'))


DISK_CASES = [("disk_doppler", dict(resx=40, resy=40), 8, None), ("disk_path_depth6", dict(resx=32, resy=32), 8, dict(type="path", max_depth=6)),
              ("disk_rr", dict(resx=24, resy=24), 8, dict(type="path", max_depth=-1, rr_depth=2))]


@pytest.mark.gpu
@pytest.mark.parametrize("name,params,spp,integ", DISK_CASES, ids=[c[0] for c in DISK_CASES])
def test_disk_scenes_are_bit_exact_per_lane(mi, orc, name, params, spp, integ):
    path = os.path.join(SCENES, "cornell_disk.xml")
    sc, osc = mi.load_file(path, **params), orc.Scene(path, params)
    if integ:
        sc.set_integrator(integ)
    pd = osc.params(integrator=integ) if integ else osc.params()
    w, h = sc.size
    n = w * h * spp
    g = sc.sample_lanes(31, spp, 0, n)
    o = osc.render_lanes(pd, 31, spp, 0, n, threads=NCPU)
    for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
        assert np.array_equal(bits(g[k]), bits(o[k])), (name, k, int((bits(g[k]) != bits(o[k])).sum()))
    assert (g["rgb"] != 0).mean() > 0.3
    img = sc.render(seed=31, spp=spp)
    ref, _ = osc.render(pd, seed=31, spp=spp, threads=NCPU)
    assert float(np.abs(img - ref).max() / np.abs(ref).max()) <= 5e-5
