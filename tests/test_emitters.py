"""Delta emitters beyond the point light: `spot` (src/emitters/spot.cpp).  CPU: oracle sanity + loader parity; GPU: per-lane parity."""
import ctypes as C
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


def room(emitters, res=32):
    """the empty Cornell room (rectangles only) lit by the given emitter XML"""
    s = ms.HEADER.format(spp=16, res=res, tsm="antithetic", shift="0.5") + ms.SENSOR.format(fov="19.5", cam=ms.CAM)
    for b in ms.BSDFS:
        s += ms.bsdf(*b)
    for name, m, b in ms.WALLS:
        s += ms.rect(name, m, b)
    return s + emitters + "</scene>\n"


DOWN = ('\t<emitter type="spot"><transform name="to_world"><lookat origin="0, 1.9, 0" target="0, 0, 0" up="0, 0, 1" /></transform>'
        '<rgb name="intensity" value="%s" /><float name="cutoff_angle" value="%s" />%s</emitter>\n')
POINT = '\t<emitter type="point"><point name="position" x="0" y="1.9" z="0" /><rgb name="intensity" value="%s" /></emitter>\n'


def test_spot_light_semantics(mi, orc, tmp_path):
    """SpotLight (spot.cpp:75-187): (1) the acos restatement against numpy; (2) loader constants bit-identical to the oracle's and equal
    to the closed forms; (3) direct light only: floor points inside the beam are lit exactly like under a point light of the same intensity,
    points outside the cutoff cone are black, and the falloff ring lies in between; (4) errors."""
    xs = np.linspace(-1, 1, 4001, dtype=np.float32)
    mine = np.array([orc.lib().orc_acos(C.c_float(float(x))) for x in xs], np.float32)
    assert np.abs(mine - np.arccos(xs.astype(np.float64))).max() < 4e-7
    path = os.path.join(SCENES, "cornell_spot.xml")
    sc, osc = mi.load_file(path), orc.Scene(path, {})
    rec = sc.export(11).reshape(-1, 22)
    em = [e for e in osc.flat.emitters if e["kind"] == 2]
    assert len(em) == 1 and rec.shape[0] == 1 and len(osc.flat.emitters) == 2
    assert np.array_equal(bits(rec[0, 18:22]), bits(em[0]["spot_params"])) and np.array_equal(bits(rec[0, 6:18]), bits(em[0]["to_local"].reshape(-1)[:12]))
    assert np.array_equal(bits(rec[0, :3]), bits(em[0]["position"])) and np.array_equal(bits(rec[0, 3:6]), bits(em[0]["intensity"]))
    assert abs(rec[0, 18] - np.radians(35)) < 1e-6 and abs(rec[0, 19] - np.cos(np.radians(35))) < 1e-6 and abs(rec[0, 21] - 1 / np.radians(15)) < 1e-4

    def render(xml, name):
        p = str(tmp_path / name)
        open(p, "w").write(xml)
        s = orc.Scene(p, dict(resx=32, resy=32))
        return s.render(s.params(integrator=dict(type="path", max_depth=2)), seed=0, spp=64, threads=NCPU)[0]   # direct light only
    spot = render(room(DOWN % ("50", "30", '<float name="beam_width" value="15" />')), "spot.xml")
    point = render(room(POINT % "50"), "point.xml")
    # the floor (y = 0) is 1.9 below the light: the beam (15 deg) covers r < 0.51 around its centre, the cone (30 deg) r < 1.10, so the
    # walls stay dark except next to the floor; the camera sees the floor foreshortened in the bottom rows of the image
    lit = spot > 0
    assert lit.any() and not lit.all() and lit[:24].sum() == 0
    centre = (slice(30, 31), slice(15, 17))
    assert np.abs(spot[centre] - point[centre]).max() <= 0.08 * point[centre].max()          # inside the beam: the same illumination (the pixel footprints reach into the falloff ring)
    assert spot[2:6, 14:18].max() == 0.0                                                      # the ceiling above the light is behind it
    ring = (spot > 0) & (spot < 0.98 * point) & (point > 0)
    assert ring.sum() > 20                                                                    # a smooth falloff zone exists
    assert (spot <= point * (1 + 1e-5) + 1e-9).all()                                          # falloff never exceeds 1
    with pytest.raises(mi.DtofError, match="cutoff_angle must not be smaller"):
        mi.load_string(room(DOWN % ("50", "10", '<float name="beam_width" value="15" />')))
    with pytest.raises(mi.DtofError, match="unreferenced property"):
        mi.load_string(room(DOWN % ("50", "30", '<float name="beamwidth" value="15" />')))
    with pytest.raises(mi.DtofError, match="unsupported emitter plugin"):
        mi.load_string(room('\t<emitter type="directional" />\n'))


SPOT_CASES = [("spot_room_fused", None, dict(resx=32, resy=32), 8, dict(type="path", max_depth=5)),
              ("spot_boxes_doppler", "cornell_spot.xml", dict(resx=40, resy=40), 8, None),
              ("spot_boxes_depth6_rr", "cornell_spot.xml", dict(resx=24, resy=24), 8, dict(type="path", max_depth=-1, rr_depth=3))]


@pytest.mark.gpu
@pytest.mark.parametrize("name,scene,params,spp,integ", SPOT_CASES, ids=[c[0] for c in SPOT_CASES])
def test_spot_scenes_are_bit_exact_per_lane(mi, orc, tmp_path, name, scene, params, spp, integ):
    if scene is None:      # rectangles only: the fused pipeline
        path = str(tmp_path / "spot_room.xml")
        open(path, "w").write(room(DOWN % ("50, 40, 30", "40", "") + POINT % "5"))
    else:
        path = os.path.join(SCENES, scene)
    sc, osc = mi.load_file(path, **params), orc.Scene(path, params)
    if integ:
        sc.set_integrator(integ)
    pd = osc.params(integrator=integ) if integ else osc.params()
    w, h = sc.size
    n = w * h * spp
    g = sc.sample_lanes(29, spp, 0, n)
    o = osc.render_lanes(pd, 29, spp, 0, n, threads=NCPU)
    for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
        assert np.array_equal(bits(g[k]), bits(o[k])), (name, k, int((bits(g[k]) != bits(o[k])).sum()))
    assert (g["rgb"] != 0).mean() > 0.3
    img = sc.render(seed=29, spp=spp)
    ref, _ = osc.render(pd, seed=29, spp=spp, threads=NCPU)
    assert float(np.abs(img - ref).max() / np.abs(ref).max()) <= 5e-5


def test_constant_environment(mi, orc):
    """`constant` emitter (src/emitters/constant.cpp): loader parity (radiance, the scene's enlarged bounding sphere), a furnace check with
    the oracle -- inside a closed sphere of radiance L every primary ray returns L times the reflected series -- and hide_emitters."""
    path = os.path.join(SCENES, "cornell_env.xml")
    osc = orc.Scene(path, dict(resx=16, resy=16))
    env = [e for e in osc.flat.emitters if e["kind"] == 3]
    assert len(env) == 1 and np.allclose(env[0]["intensity"], [0.8, 0.9, 1.2])
    c, r = env[0]["bsphere"][:3], env[0]["bsphere"][3]
    assert np.allclose(c, [0, 1, 0], atol=1e-6) and 1.73 < r < 1.7323        # the room spans [-1, 1] x [0, 2] x [-1, 1]: radius sqrt(3) * (1 + eps)
    # sky pixels: the developed image shows the radiance itself (plain path integrator); hidden emitters: black
    pd = osc.params(integrator=dict(type="path", max_depth=4))
    img, _ = osc.render(pd, seed=0, spp=8, threads=NCPU)
    assert np.allclose(img[0, 8], [0.8, 0.9, 1.2], rtol=1e-5)
    hidden, _ = osc.render(osc.params(integrator=dict(type="path", max_depth=4, hide_emitters=True)), seed=0, spp=8, threads=NCPU)
    assert np.array_equal(hidden[0, 8], [0, 0, 0]) and hidden[12, 8].sum() > 0
    # white furnace: a diffuse sphere of albedo a around the camera, radiance L from the environment only:
    # pixel = L * (a + a^2 + ...) truncated at max_depth - 1 bounces of the light path
    xml = ('<scene version="3.0.0"><integrator type="path"><integer name="max_depth" value="%d"/></integrator>'
           '<sensor type="perspective"><float name="fov" value="40"/><sampler type="independent"><integer name="sample_count" value="64"/></sampler>'
           '<film type="hdrfilm"><integer name="width" value="4"/><integer name="height" value="4"/><rfilter type="box"/></film></sensor>'
           '<shape type="sphere"><float name="radius" value="2"/><boolean name="flip_normals" value="true"/>'
           '<bsdf type="diffuse"><rgb name="reflectance" value="0.5"/></bsdf></shape>'
           '<emitter type="constant"><rgb name="radiance" value="2"/></emitter></scene>')
    # from inside a closed sphere the environment is never seen: every path is absorbed -> black (the emitter is sampled but always occluded)
    closed = orc.Scene(xml % 6, is_string=True)
    img, _ = closed.render(closed.params(), seed=1, spp=64, threads=NCPU)
    assert np.abs(img).max() == 0.0
