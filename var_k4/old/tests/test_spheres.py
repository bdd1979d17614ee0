"""`sphere` shapes: constructor/transform baking, the float64 ray-sphere intersection of the llvm back end,
surface interaction, and spherical area lights (cone sampling of the visible cap).

CPU: loader parity (C++ through the C-ABI export vs the oracle's Python + C baking), analytic known answers for the oracle's
intersection, and the sphere light against a finely tessellated mesh light (different sampling code, same integrand).
GPU: every lane bit-exact against the oracle."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENES

sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_mesh  # noqa: E402
import make_scenes as ms  # noqa: E402

NCPU = os.cpu_count() or 1


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def scene_with(shapes_xml, light=True):
    s = ms.HEADER.format(spp=8, res=32, tsm="antithetic", shift="0.5") + ms.SENSOR.format(fov="19.5", cam=ms.CAM)
    for b in ms.BSDFS:
        s += ms.bsdf(*b)
    for name, m, b in ms.WALLS:
        s += ms.rect(name, m, b)
    return s + shapes_xml + (ms.LIGHT if light else "") + "</scene>\n"


TRANSFORMED = ('\t<shape type="sphere" id="S">\n\t\t<point name="center" x="0.1" y="0.2" z="-0.1" />\n\t\t<float name="radius" value="0.5" />\n'
               '\t\t<transform name="to_world">\n\t\t\t<scale value="0.8" />\n\t\t\t<rotate y="1" angle="35" />\n\t\t\t<rotate x="1" angle="-20" />\n'
               '\t\t\t<translate x="-0.2" y="0.5" z="0.1" />\n\t\t</transform>\n%s\t\t<ref id="TallBoxBSDF" />\n\t</shape>\n')


def test_sphere_loader_matches_the_oracle(mi, orc, tmp_path):
    for i, extra in enumerate(("", '\t\t<boolean name="flip_normals" value="true" />\n')):
        path = str(tmp_path / ("s%d.xml" % i))
        open(path, "w").write(scene_with(TRANSFORMED % extra + ms.sphere("Unit", "ShortBoxBSDF", ("0", "0", "0"), "1", extra='\t\t<transform name="to_world">\n\t\t\t<scale x="-0.2" y="0.2" z="0.2" />\n\t\t</transform>\n')))
        sc, osc = mi.load_file(path), orc.Scene(path, {})
        sph = [s for s in osc.flat.shapes if s["kind"] == 2]
        assert len(sph) == 2
        got = sc.export(8).reshape(-1, 6)
        want = np.stack([s["sphere_baked"][:6] for s in sph])
        assert np.array_equal(bits(got), bits(want))
        # radius = 0.5 * 0.8, centre = to_world * (0.1, 0.2, -0.1); the mirrored unit sphere flips its normals (sphere.cpp:151-154)
        assert abs(got[0, 3] - 0.4) < 1e-6 and abs(got[1, 3] - 0.2) < 1e-6
        assert got[0, 5] == (1.0 if extra else 0.0) and got[1, 5] == 1.0
        assert abs(got[0, 4] - 1.0 / (4 * np.pi * 0.16)) < 1e-5
        tf = sc.export(1).reshape(-1, 32)
        otf = np.stack([np.concatenate([np.asarray(s["to_world"]).reshape(-1), np.asarray(s["to_object"]).reshape(-1)]) for s in osc.flat.shapes])
        assert np.array_equal(bits(tf), bits(otf))
        m = tf[-2, :16].reshape(4, 4).astype(np.float64) @ tf[-2, 16:].reshape(4, 4).astype(np.float64)
        assert np.allclose(m, np.eye(4), atol=1e-5)


def test_oracle_sphere_intersection_known_answers(orc, tmp_path):
    """Sphere::ray_intersect_preliminary_impl / ray_test_impl (sphere.cpp:338-431) on a centred sphere of radius 0.5"""
    import ctypes as C
    path = str(tmp_path / "k.xml")
    open(path, "w").write(ms.HEADER.format(spp=4, res=8, tsm="antithetic", shift="0.5") + ms.SENSOR.format(fov="19.5", cam=ms.CAM) +
                          ms.bsdf("B", "0.5, 0.5, 0.5") + ms.sphere("S", "B", ("0", "0", "0"), "0.5") + ms.LIGHT + "</scene>\n")
    osc = orc.Scene(path, {})
    L = orc.lib()
    def hit(o, d, maxt=1e30):
        o, d = (C.c_float * 3)(*o), (C.c_float * 3)(*d)
        out, ids = (C.c_float * 3)(), (C.c_int32 * 3)()
        found = L.orc_intersect(C.byref(osc.c), o, d, 0.0, maxt, out, ids)
        occ = L.orc_occluded(C.byref(osc.c), o, d, 0.0, maxt)
        return found, out[0], occ
    assert hit((0, 0, 3), (0, 0, -1))[:2] == (1, 2.5) and hit((0, 0, 3), (0, 0, -1))[2] == 1          # front face
    assert hit((0, 0, 0), (0, 0, 1))[:2] == (1, 0.5)                                                    # from the centre: far root
    assert hit((0, 0, 3), (0, 0, -1), maxt=2.0)[0] == 0 and hit((0, 0, 3), (0, 0, -1), maxt=2.0)[2] == 0   # beyond maxt
    assert hit((0, 0, 0.2), (0, 0, 1), maxt=0.1)[0] == 0                                                # entirely inside (in_bounds)
    assert hit((0.6, 0, 3), (0, 0, -1))[0] == 0 and hit((0, 0, 3), (0, 0, 1))[0] == 0                   # miss, behind
    f, t, _ = hit((0.3, 0, 3), (0, 0, -2))                                                              # unnormalised direction
    assert f == 1 and abs(t - (3 - 0.4) / 2) < 1e-6
    f, t, _ = hit((0.4999, 0, 3), (0, 0, -1))                                                           # almost grazing
    assert f == 1 and abs(t - 3.0) < 0.02
    assert hit((0.5, 0, 3), (0, 0, -1))[0] == 0    # exactly tangent: B = C = 0 -> c / temp = 0 / -0 = NaN -> rejected (math.h:387-390)


def test_sphere_light_equals_a_tessellated_mesh_light_in_expectation(orc, tmp_path):
    """Sphere::sample_direction / pdf_direction (sphere.cpp:222-310) vs Mesh::sample_position on a 96 x 48 UV sphere"""
    d = str(tmp_path)
    n_u, n_v, pos, faces = 96, 48, [], []
    for j in range(n_v + 1):
        for i in range(n_u + 1):
            th, ph = np.pi * j / n_v, 2 * np.pi * i / n_u
            pos.append((np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)))
    W = n_u + 1
    for j in range(n_v):
        for i in range(n_u):
            a, b, c, e = j * W + i, j * W + i + 1, (j + 1) * W + i, (j + 1) * W + i + 1
            if j != 0:
                faces.append((a, b, c))
            if j != n_v - 1:
                faces.append((b, e, c))
    make_mesh.write_ply(os.path.join(d, "ball.ply"), pos, pos, [(0, 0)] * len(pos), faces)
    sphere_xml = open(os.path.join(SCENES, "cornell_sphere_light.xml")).read()
    mesh_xml = sphere_xml.replace('<shape type="sphere" id="Light">\n\t\t<point name="center" x="0" y="1.7" z="0" />\n\t\t<float name="radius" value="0.12" />',
                                  '<shape type="ply" id="Light">\n\t\t<string name="filename" value="ball.ply" />\n\t\t<transform name="to_world">\n'
                                  '\t\t\t<scale value="0.12" />\n\t\t\t<translate x="0" y="1.7" z="0" />\n\t\t</transform>')
    assert mesh_xml != sphere_xml
    open(os.path.join(d, "mesh_light.xml"), "w").write(mesh_xml)
    P, integ = dict(resx=16, resy=16), dict(type="path", max_depth=3)
    a, b = orc.Scene(os.path.join(SCENES, "cornell_sphere_light.xml"), P), orc.Scene(os.path.join(d, "mesh_light.xml"), P)
    ia = np.mean([a.render(a.params(integrator=integ), seed=s, spp=256, threads=NCPU)[0] for s in range(3)], axis=0)
    ib = np.mean([b.render(b.params(integrator=integ), seed=s, spp=256, threads=NCPU)[0] for s in range(3)], axis=0)
    assert abs(ia.mean() - ib.mean()) < 0.02 * ia.mean(), (ia.mean(), ib.mean())     # the mesh has 0.2 % less area
    assert np.abs(ia - ib).mean() < 0.08 * ia.mean()


GPU_CASES = [("cornell_spheres", os.path.join(SCENES, "cornell_spheres.xml"), dict(resx=48, resy=48), 8),
             ("cornell_sphere_light", os.path.join(SCENES, "cornell_sphere_light.xml"), dict(resx=32, resy=32), 8),
             ("sphere_light_depth6", os.path.join(SCENES, "cornell_sphere_light.xml"), dict(resx=24, resy=24, max_depth=6, time_sampling_method="stratified"), 8),
             ("transformed", None, dict(resx=32, resy=32), 8), ("transformed_flipped", None, dict(resx=32, resy=32), 4),
             ("camera_inside_a_sphere", None, dict(resx=24, resy=24), 8)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,path,params,spp", GPU_CASES, ids=[c[0] for c in GPU_CASES])
def test_sphere_scenes_are_bit_exact_per_lane(mi, orc, tmp_path, name, path, params, spp):
    if path is None:
        path = str(tmp_path / (name + ".xml"))
        if name.startswith("transformed"):
            xml = scene_with(TRANSFORMED % ('\t\t<boolean name="flip_normals" value="true" />\n' if "flipped" in name else ""))
        else:   # a big inverted sphere around the whole room, lit from inside by a small sphere light next to the camera
            xml = scene_with(ms.sphere("Shell", "TallBoxBSDF", ("0", "1", "0"), "12", extra='\t\t<boolean name="flip_normals" value="true" />\n') +
                             ms.sphere("Lamp", None, ("0.5", "1.2", "6.0"), "0.2", emitter="30, 30, 30"), light=False)
        open(path, "w").write(xml)
    sc, osc = mi.load_file(path, **params), orc.Scene(path, params)
    w, h = sc.size
    n = w * h * spp
    for integ in (None, dict(type="path", max_depth=4)):
        if integ:
            sc.set_integrator(integ)
        pd = osc.params(integrator=integ) if integ else osc.params()
        g = sc.sample_lanes(6, spp, 0, n)
        o = osc.render_lanes(pd, 6, spp, 0, n, threads=NCPU)
        for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
            assert np.array_equal(bits(g[k]), bits(o[k])), (name, integ, k, int((bits(g[k]) != bits(o[k])).sum()))
        assert (g["rgb"] != 0).mean() > 0.2
    img = sc.render(seed=6, spp=spp)
    ref, _ = osc.render(pd, seed=6, spp=spp, threads=NCPU)
    assert float(np.abs(img - ref).max() / np.abs(ref).max()) <= 1e-5
