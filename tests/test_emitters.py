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
        mi.load_string(room('\t<emitter type="projector" />\n'))


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


def _envmap_emitter(orc, img, scale=1.0):
    import ctypes as C
    L = orc.lib()
    L.orc_envmap_create.restype = C.c_void_p; L.orc_envmap_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float]
    L.orc_envmap_sample_direction.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    L.orc_envmap_pdf_direction.restype = C.c_float; L.orc_envmap_pdf_direction.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_envmap_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    img = np.ascontiguousarray(img, np.float32)
    em = orc.OrcEmitter(); em.kind = 4
    em.envmap = L.orc_envmap_create(img.ctypes.data, img.shape[1], img.shape[0], C.c_float(scale))
    ident = (C.c_float * 16)(*np.eye(4, dtype=np.float32).ravel())
    em.to_local = ident; em.env_to_world = ident; em.bsphere = (C.c_float * 4)(0, 0, 0, 1)
    return L, em


def test_envmap_known_answers_of_the_reference(orc):
    """src/emitters/tests/test_envmap.py restated for the oracle: test02_sampling_weights (a 10 x 100 map with ONE pixel on: the weights of
    sample_direction stay within (0.018, 0.02) and equal eval / pdf_direction within 1e-3 -- numbers the reference's own test holds) and the
    content of test01_chi2 (sampled directions follow pdf_direction: sparse, constant high-res and constant 2 x 3 maps)."""
    import ctypes as C
    rng = np.random.default_rng(3)
    img = np.zeros((100, 10, 3), np.float32); img[40, 5] = 1
    L, em = _envmap_emitter(orc, img)
    w, w2, w3 = [], [], []
    for sx, sy in rng.random((3000, 2)):
        out = (C.c_float * 8)()
        L.orc_envmap_sample_direction(C.byref(em), (C.c_float * 3)(0, 0, 0), float(sx), float(sy), out)
        d = (C.c_float * 3)(out[0], out[1], out[2]); rgb = (C.c_float * 3)()
        L.orc_envmap_eval(C.byref(em), d, rgb)
        w.append(out[5]); w2.append(rgb[0] / L.orc_envmap_pdf_direction(C.byref(em), d)); w3.append(rgb[0] / out[4])
        assert abs(np.linalg.norm(out[0:3]) - 1) < 1e-5 and out[3] == 2.0
    w, w2, w3 = np.array(w), np.array(w2), np.array(w3)
    assert np.allclose(w, w2, rtol=1e-3) and np.allclose(w, w3, rtol=1e-3)        # test_envmap.py:68-69
    assert w.min() > 0.018 and w.max() < 0.02                                     # test_envmap.py:70
    # chi^2-style check on a coarse spherical histogram (test01_chi2: iterations 0 - 2)
    for im in (img, np.ones((100, 100, 3), np.float32), np.ones((3, 2, 3), np.float32)):
        L, em = _envmap_emitter(orc, im)
        n, nb = 40000, (8, 16)
        hist = np.zeros(nb); s = rng.random((n, 2))
        for sx, sy in s:
            out = (C.c_float * 8)()
            L.orc_envmap_sample_direction(C.byref(em), (C.c_float * 3)(0, 0, 0), float(sx), float(sy), out)
            ct, ph = np.clip(out[2], -1, 1), np.arctan2(out[1], out[0]) % (2 * np.pi)
            hist[min(int((ct + 1) / 2 * nb[0]), nb[0] - 1), min(int(ph / (2 * np.pi) * nb[1]), nb[1] - 1)] += 1
        # expected counts: integrate pdf_direction over each (cos theta, phi) cell with a 40 x 40 midpoint rule
        exp = np.zeros(nb); m = 40
        for i in range(nb[0]):
            for j in range(nb[1]):
                acc = 0.0
                for a in range(m):
                    for b in range(m):
                        ct = -1 + 2 * (i + (a + .5) / m) / nb[0]; ph = 2 * np.pi * (j + (b + .5) / m) / nb[1]; st = np.sqrt(1 - ct * ct)
                        acc += L.orc_envmap_pdf_direction(C.byref(em), (C.c_float * 3)(st * np.cos(ph), st * np.sin(ph), ct))
                exp[i, j] = acc / (m * m) * (4 * np.pi / (nb[0] * nb[1])) * n
        assert abs(exp.sum() / n - 1) < 0.02, exp.sum() / n                       # the density integrates to one
        big = exp > 50
        z = (hist[big] - exp[big]) / np.sqrt(exp[big])
        assert np.abs(z).max() < 6 and abs(hist[~big].sum() - exp[~big].sum()) < 6 * np.sqrt(exp[~big].sum() + 1) + 0.01 * n, (np.abs(z).max(),)


def _piz_stub(tmp_path):
    """an EXR header that declares RLE compression (the reader must refuse it before touching any chunk)"""
    import make_scenes
    q = str(tmp_path / "piz.exr")
    make_scenes.write_exr(q, make_scenes.env_pixels(8, 4), compression=0)
    d = open(q, "rb").read()
    i = d.index(b"compression\0compression\0") + 24 + 4
    open(q, "wb").write(d[:i] + b"\x01" + d[i + 1:])
    return q


def test_envmap_scene(mi, orc, tmp_path):
    """`envmap` emitter end to end on the CPU side: the three file formats decode to the same map (RGBE, PFM exactly; PNG through sRGB), loader
    parity with the product (blob tables = the oracle's tables), radiance lookup of sky pixels, hide_emitters, error messages."""
    from oracle import scene_xml as sx
    a, b = sx.read_radiance_image(os.path.join(SCENES, "env_sky.hdr")), sx.read_radiance_image(os.path.join(SCENES, "env_sky.pfm"))
    assert a.shape == b.shape == (16, 32, 3) and np.abs(a - b).max() <= b.max() / 128 and np.array_equal(b[0, 0], np.float32([0.5, 0.7, 1.15]))
    path = os.path.join(SCENES, "cornell_envmap.xml")
    osc = orc.Scene(path, dict(resx=16, resy=16))
    env = [e for e in osc.flat.emitters if e["kind"] == 4]
    assert len(env) == 1 and env[0]["image"].shape == (16, 32, 3) and env[0]["scale"] == np.float32(0.6)
    # a pixel that looks out of the open back of the room shows the (rotated, scaled) map: positive, finite, no larger than scale * max(map)
    pd = osc.params(integrator=dict(type="path", max_depth=4))
    img, _ = osc.render(pd, seed=0, spp=8, threads=NCPU)
    assert np.isfinite(img).all() and img[0, 8].min() > 0 and img.max() <= 0.6 * a.max() * 1.01
    hidden, _ = osc.render(osc.params(integrator=dict(type="path", max_depth=4, hide_emitters=True)), seed=0, spp=8, threads=NCPU)
    assert np.array_equal(hidden[0, 8], [0, 0, 0]) and hidden[12, 8].sum() > 0
    from oracle.orc import envmap_export
    sc = mi.load_file(path, resx=16, resy=16)
    ours, theirs = sc.export(16), envmap_export(osc.c.emitters[[e["kind"] for e in osc.flat.emitters].index(4)])
    assert ours.size == theirs.size and np.array_equal(ours.view(np.uint32), theirs.view(np.uint32))   # radiance, bounding sphere, rotation, every level: bit for bit
    # the product builds the same tables: the three formats load, resolution limits and unsupported options raise as the reference does
    text = open(path).read()
    for fn in ("env_sky.pfm", "env_sky.png"):
        q = os.path.join(SCENES, "_tmp_" + fn + ".xml")
        open(q, "w").write(text.replace("env_sky.hdr", fn))
        try:
            assert mi.load_file(q).info()["n_emitters"] == 2 and len(orc.Scene(q, {}).flat.emitters) == 2
        finally:
            os.remove(q)
    absolute = text.replace("env_sky.hdr", os.path.join(SCENES, "env_sky.hdr"))

    def load(name, xml):
        (tmp_path / name).write_text(xml)
        return mi.load_file(str(tmp_path / name))
    # OpenEXR radiance maps: uncompressed, ZIPS and ZIP chunks, HALF and FLOAT, decreasing line order, an extra alpha channel -- the same tables again;
    # the uncompressed container also decodes with the independent reader of tools/exr_piz.py (the one that decoded the authors' scene.exr)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exr_piz
    sky = ms.env_pixels(37, 19)
    for k, kw in enumerate((dict(compression=0), dict(compression=2, half=False), dict(compression=3), dict(compression=3, decreasing_y=True, alpha=True, half=False))):
        q = str(tmp_path / ("sky%d.exr" % k))
        ms.write_exr(q, sky, **kw)
        if kw["compression"] == 0:
            ch, _ = exr_piz.read_exr(q)
            assert np.array_equal(ch["R"], np.asarray(sky, np.float32)[..., 0].astype(np.float16).astype(np.float32))
        (tmp_path / ("exr%d.xml" % k)).write_text(text.replace("env_sky.hdr", q))
        sce, osce = mi.load_file(str(tmp_path / ("exr%d.xml" % k))), orc.Scene(str(tmp_path / ("exr%d.xml" % k)), {})
        theirs = envmap_export(osce.c.emitters[[e["kind"] for e in osce.flat.emitters].index(4)])
        assert np.array_equal(sce.export(16).view(np.uint32), theirs.view(np.uint32)), kw
        ref = np.asarray(sky, np.float32) if not kw.get("half", True) else np.asarray(sky, np.float32).astype(np.float16).astype(np.float32)
        assert np.array_equal(theirs[32:32 + 38 * 19 * 3].reshape(19, 38, 3)[:, :37], ref), kw        # m_data: the decoded pixels (+ the periodic column)
    # what the package's own writer produces (mitsuba3dopplertof_amd.io.write_exr: ZIP by default) is read back by the library's radiance-map reader
    from mitsuba3dopplertof_amd import io as dio
    for k, kw in enumerate((dict(), dict(half=False, compression="zips"), dict(half=False, compression="none"))):
        q = str(tmp_path / ("own%d.exr" % k))
        dio.write_exr(q, np.asarray(sky, np.float32), **kw)
        (tmp_path / ("own%d.xml" % k)).write_text(text.replace("env_sky.hdr", q))
        sce, osce = mi.load_file(str(tmp_path / ("own%d.xml" % k))), orc.Scene(str(tmp_path / ("own%d.xml" % k)), {})
        theirs = envmap_export(osce.c.emitters[[e["kind"] for e in osce.flat.emitters].index(4)])
        assert np.array_equal(sce.export(16).view(np.uint32), theirs.view(np.uint32)), kw
        ref = np.asarray(sky, np.float32) if not kw.get("half", True) else np.asarray(sky, np.float32).astype(np.float16).astype(np.float32)
        assert np.array_equal(theirs[32:32 + 38 * 19 * 3].reshape(19, 38, 3)[:, :37], ref), kw
    with pytest.raises(mi.DtofError, match="not RLE"):
        (tmp_path / "rle.xml").write_text(text.replace("env_sky.hdr", _piz_stub(tmp_path)))
        mi.load_file(str(tmp_path / "rle.xml"))
    # PIZ (wavelet + Huffman): the one real file at hand is the authors' configs_example/scene.exr (256 x 256 HALF, written by Mitsuba 3.2); where
    # the reference tree is present (the build container) the product's decoder must reproduce the pixels tools/exr_piz.py decoded into
    # tests/golden/reference_configs_example_scene_exr.npy, and build the same tables as the oracle
    piz = "/root/reference/configs_example/scene.exr"
    if os.path.exists(piz):
        (tmp_path / "piz.xml").write_text(text.replace("env_sky.hdr", piz))
        scp, oscp = mi.load_file(str(tmp_path / "piz.xml")), orc.Scene(str(tmp_path / "piz.xml"), {})
        ours = scp.export(16)
        assert np.array_equal(ours.view(np.uint32), envmap_export(oscp.c.emitters[[e["kind"] for e in oscp.flat.emitters].index(4)]).view(np.uint32))
        golden = np.load(os.path.join(ROOT, "tests", "golden", "reference_configs_example_scene_exr.npy")).astype(np.float32)
        assert np.array_equal(ours[32:32 + 257 * 256 * 3].reshape(256, 257, 3)[:, :256], golden)
    # a JPEG radiance map (4:2:0): product decoder vs PIL, then the same tables bit for bit
    from PIL import Image
    Image.open(os.path.join(SCENES, "env_sky.png")).convert("RGB").resize((40, 22)).save(str(tmp_path / "sky.jpg"), "JPEG", quality=88, subsampling=2)
    (tmp_path / "jpg.xml").write_text(text.replace("env_sky.hdr", str(tmp_path / "sky.jpg")))
    scj, oscj = mi.load_file(str(tmp_path / "jpg.xml")), orc.Scene(str(tmp_path / "jpg.xml"), {})
    assert np.array_equal(scj.export(16).view(np.uint32), envmap_export(oscj.c.emitters[[e["kind"] for e in oscj.flat.emitters].index(4)]).view(np.uint32))
    with pytest.raises(mi.DtofError, match="Only one environment emitter"):
        load("two.xml", absolute.replace("</scene>", '<emitter type="constant"/></scene>'))
    # mis_compensation (envmap.cpp:157-185): the sampling tables are built from max(luminance - mean luminance, 0); both builds make the same tables, texels
    # below the mean are never sampled, the radiance data is untouched
    (tmp_path / "mis.xml").write_text(absolute.replace('<float name="scale" value="0.6" />', '<float name="scale" value="0.6" /><boolean name="mis_compensation" value="true" />'))
    scm, oscm = mi.load_file(str(tmp_path / "mis.xml")), orc.Scene(str(tmp_path / "mis.xml"), {})
    comp = scm.export(16)
    assert np.array_equal(comp.view(np.uint32), envmap_export(oscm.c.emitters[[e["kind"] for e in oscm.flat.emitters].index(4)]).view(np.uint32))
    plain = load("plain.xml", absolute).export(16)
    w_, h_ = int(plain[0]), int(plain[1]); n_data = w_ * h_ * 3
    assert np.array_equal(comp[:32 + n_data], plain[:32 + n_data]) and not np.array_equal(comp[32 + n_data:], plain[32 + n_data:])
    lvl0_c, lvl0_p = comp[32 + n_data + 2:32 + n_data + 2 + w_ * h_], plain[32 + n_data + 2:32 + n_data + 2 + w_ * h_]
    assert (lvl0_c == 0).sum() > (lvl0_p == 0).sum() + w_ and lvl0_c.max() > lvl0_p.max()
    sys.path.insert(0, SCENES)
    import make_scenes
    make_scenes.write_pfm(str(tmp_path / "tiny.pfm"), [[(1.0, 1.0, 1.0)] * 2] * 2)
    with pytest.raises(mi.DtofError, match="must be at least 2x3 pixels"):
        load("tiny.xml", text.replace("env_sky.hdr", str(tmp_path / "tiny.pfm")))
    with pytest.raises(mi.DtofError, match="could not open"):
        load("missing.xml", text.replace("env_sky.hdr", str(tmp_path / "nope.hdr")))


def test_directional_emitter(mi, orc, tmp_path):
    """`directional` (src/emitters/directional.cpp): loader parity (direction by `direction` -- normalised twice in float32, as the constructor's
    normalize + look_at do -- and by `to_world`; irradiance), the analytic irradiance of an unoccluded patch (E cos(theta) rho / pi), the refusal
    of both parameters at once."""
    path = os.path.join(SCENES, "cornell_sun.xml")
    sc, osc = mi.load_file(path), orc.Scene(path, {})
    ours = sc.export(18).reshape(-1, 10)
    suns = [e for e in osc.flat.emitters if e["kind"] == 5]
    assert len(suns) == 2 and [int(k) for k in ours[:, 0]] == [e["kind"] for e in osc.flat.emitters]
    for row, e in zip(ours[ours[:, 0] == 5], suns):
        assert np.array_equal(row[7:10].view(np.uint32), np.float32(e["position"]).view(np.uint32))      # the direction, bit for bit
        assert np.array_equal(row[4:7].view(np.uint32), np.float32(e["intensity"]).view(np.uint32))
        assert abs(np.linalg.norm(row[7:10]) - 1) < 1e-6
    assert np.allclose(ours[0, 7:10], np.float32([-0.3, -1, -0.4]) / np.linalg.norm([-0.3, -1, -0.4]), atol=1e-7)
    # one white diffuse floor under a vertical sun of irradiance E, seen from above: radiance = E * rho / pi at depth 2 (direct light only)
    xml = ('<scene version="3.0.0"><integrator type="path"><integer name="max_depth" value="2"/></integrator>'
           '<sensor type="perspective"><float name="fov" value="20"/><transform name="to_world"><lookat origin="0, 4, 0.001" target="0, 0, 0" up="0, 1, 0"/></transform>'
           '<sampler type="independent"><integer name="sample_count" value="16"/></sampler>'
           '<film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/><rfilter type="box"/></film></sensor>'
           '<shape type="rectangle"><transform name="to_world"><rotate x="1" angle="-90"/><scale value="10"/></transform>'
           '<bsdf type="diffuse"><rgb name="reflectance" value="0.5"/></bsdf></shape>'
           '<emitter type="directional"><vector name="direction" x="0" y="-1" z="0"/><rgb name="irradiance" value="2.0, 4.0, 6.0"/></emitter></scene>')
    flat = orc.Scene(xml, is_string=True)
    img, _ = flat.render(flat.params(), seed=1, spp=16, threads=NCPU)
    assert np.allclose(img[4, 4], np.float32([2.0, 4.0, 6.0]) * 0.5 / np.pi, rtol=1e-5)
    with pytest.raises(mi.DtofError, match="Only one of the parameters 'direction' and 'to_world'"):
        mi.load_string(xml.replace('<vector name="direction" x="0" y="-1" z="0"/>', '<vector name="direction" x="0" y="-1" z="0"/><transform name="to_world"><rotate x="1" angle="10"/></transform>'))


# ------------------------------------------------------------------------------------------------ the reference's own emitter tests, geometry halves
# src/emitters/tests/test_{point,spot,directional,constant}.py run in spectral variants only and compare the sampled weight with a spectrum object, so the
# harvester (tests/golden/extract_reference_kats.py) can take nothing from them; what they assert about GEOMETRY -- ds.d, ds.pdf, ds.delta, the distance
# falloff and the spot light's falloff curve, for the inputs the tests hold -- is independent of the colour representation and is restated here with
# those inputs (cited), against the oracle's Emitter::sample_direction (orc_kat_emitter_sample; the GPU kernels are lane-for-lane bit-exact with it).
def _emitter_scene(orc, emitter_xml):
    xml = ('<scene version="3.0.0"><integrator type="path"/><sensor type="perspective"><float name="fov" value="40"/><film type="hdrfilm">'
           '<integer name="width" value="4"/><integer name="height" value="4"/></film></sensor>'
           '<shape type="rectangle"><transform name="to_world"><scale value="0.5"/></transform></shape>' + emitter_xml + '</scene>')
    return orc.Scene(xml, is_string=True)


def _sample(orc, sc, ref, sx, sy, index=0):
    out = np.zeros(13, np.float32)
    ref = np.ascontiguousarray(ref, np.float32)
    orc.lib().orc_kat_emitter_sample(C.byref(sc.c), index, ref.ctypes.data_as(C.c_void_p), C.c_float(sx), C.c_float(sy), out.ctypes.data_as(C.c_void_p))
    return dict(d=out[0:3].astype(np.float64), dist=float(out[3]), pdf=float(out[4]), delta=bool(out[5]), weight=out[6:9].astype(np.float64), p=out[9:12].astype(np.float64))


def _lookat_xml(origin, target, up):
    return '<transform name="to_world"><lookat origin="%s" target="%s" up="%s"/></transform>' % tuple(", ".join(repr(float(x)) for x in v) for v in (origin, target, up))


def test_reference_point_light_sample_direction(orc):
    """test_point.py:60-89 (test02_point_sample_direction): emitter at [10, -1, 2], it.p = [0, -2, 4.5], sample [0.1, 0.5]"""
    pos, p = np.array([10.0, -1.0, 2.0]), np.array([0.0, -2.0, 4.5])
    sc = _emitter_scene(orc, '<emitter type="point"><point name="position" x="10" y="-1" z="2"/><rgb name="intensity" value="3, 5, 7"/></emitter>')
    ds = _sample(orc, sc, p, 0.1, 0.5)
    d = pos - p; dist = np.linalg.norm(d); d /= dist
    assert ds["pdf"] == 1.0 and ds["delta"] and np.allclose(ds["d"], d, rtol=1e-5, atol=1e-8) and np.isclose(ds["dist"], dist, rtol=1e-6)
    assert np.allclose(ds["weight"], np.array([3.0, 5.0, 7.0]) / dist ** 2, rtol=1e-5)          # res == spectrum / dist**2


@pytest.mark.parametrize("it_pos", [[2.0, 0.5, 0.0], [1.0, 0.5, -5.0]])
@pytest.mark.parametrize("cutoff_angle", [20, 80])
@pytest.mark.parametrize("lookat", [([0, 1, 0], [0, 0, 0], [1, 0, 0]), ([0, 0, 1], [0, 0, 0], [0, -1, 0])])
def test_reference_spot_light_sample_direction(orc, it_pos, cutoff_angle, lookat):
    """test_spot.py:43-93 (test_sample_direction) over its parametrisation: it_pos, cutoff_angle in {20, 80}, the two look_at transforms; beam width
    = 3/4 of the cutoff (the plugin's default), falloff (cutoff - angle) / (cutoff - beam) between the two, 0 beyond the cutoff, 1 / dist^2"""
    origin, target, up = (np.array(v, np.float64) for v in lookat)
    sc = _emitter_scene(orc, '<emitter type="spot">%s<float name="cutoff_angle" value="%d"/><rgb name="intensity" value="2, 4, 8"/></emitter>' % (_lookat_xml(origin, target, up), cutoff_angle))
    cutoff = np.radians(cutoff_angle); beam = cutoff * 0.75
    p = np.array(it_pos, np.float64)
    d = origin - p; dist = np.linalg.norm(d); d /= dist                          # lookat.translation() is the light's position
    axis = (target - origin) / np.linalg.norm(target - origin)                   # (trafo.inverse() @ (-d))[2] = cos of the angle to the light's axis
    angle = np.arccos(np.clip(np.dot(-d, axis), -1, 1))
    if abs(angle - beam) < 1e-3:
        angle = beam
    if abs(angle - cutoff) < 1e-3:
        angle = cutoff
    spec = np.array([2.0, 4.0, 8.0])
    if angle > beam:
        spec = spec * ((cutoff - angle) / (cutoff - beam))
    if angle > cutoff:
        spec = spec * 0
    ds = _sample(orc, sc, p, 0.0, 0.0)
    assert ds["pdf"] == 1.0 and ds["delta"] and np.allclose(ds["d"], d, rtol=1e-5, atol=1e-7)
    assert np.allclose(ds["weight"], spec / dist ** 2, rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("direction", [[0, 0, -1], [1, 1, 1], [0, 0, 1]])
def test_reference_directional_emitter_sample_direction(orc, direction):
    """test_directional.py:88-114 (test_sample_direction): it.p = [-0.5, 0.3, -0.1], samples [0.85, 0.13]; ds.d = -direction / |direction|, pdf 1, no distance
    attenuation.  (The directions are those of the file's `direction` fixture.)"""
    sc = _emitter_scene(orc, '<emitter type="directional"><vector name="direction" x="%g" y="%g" z="%g"/><rgb name="irradiance" value="1.5, 2.5, 3.5"/></emitter>' % tuple(direction))
    ds = _sample(orc, sc, [-0.5, 0.3, -0.1], 0.85, 0.13)
    dn = np.array(direction, np.float64); dn /= np.linalg.norm(dn)
    assert np.allclose(ds["d"], -dn, rtol=1e-5, atol=1e-7) and np.isclose(ds["pdf"], 1.0) and ds["delta"]
    assert np.allclose(ds["weight"], [1.5, 2.5, 3.5], rtol=1e-6)


def test_reference_constant_emitter_sample_direction(orc):
    """test_constant.py:68-89 (test03_sample_direction): three points inside the unit sphere, samples [[0.4, 0.5, 0.3], [0.1, 0.4, 0.9]]: pdf = 1 / (4 pi),
    ds.d = square_to_uniform_sphere(sample), weight = radiance * 4 pi"""
    sc = _emitter_scene(orc, '<emitter type="constant"><rgb name="radiance" value="0.5, 1, 2"/></emitter>')
    pts = [[-0.5, 0.3, -0.1], [0.8, -0.3, -0.2], [-0.2, 0.6, -0.6]]
    sx, sy = [0.4, 0.5, 0.3], [0.1, 0.4, 0.9]
    for p, a, b in zip(pts, sx, sy):
        ds = _sample(orc, sc, p, a, b)
        z = 1.0 - 2.0 * b; r = np.sqrt(max(0.0, 1.0 - z * z))                    # warp::square_to_uniform_sphere (warp.h:278-288)
        expect = np.array([r * np.cos(2 * np.pi * a), r * np.sin(2 * np.pi * a), z])
        assert np.isclose(ds["pdf"], 1 / (4 * np.pi), rtol=1e-6) and not ds["delta"] and np.allclose(ds["d"], expect, atol=2e-6)
        assert np.allclose(ds["weight"], np.array([0.5, 1.0, 2.0]) * 4 * np.pi, rtol=1e-5)
