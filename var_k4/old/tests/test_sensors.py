"""`thinlens` and `orthographic` sensors (src/sensors/{thinlens,orthographic}.cpp) and the aperture draw of render_sample
(src/render/integrator.cpp:421-423,490-492).

CPU: the oracle's camera against closed forms (rays through one film point meet on the focal plane; the origin lies on the lens, pushed to the
near plane), the draw order of a lane (pixel jitter, aperture sample, time sample) and the loader's error behaviour.  The reference's own
test_thinlens.py assertions run in tests/test_oracle_reference_kats.py.  GPU: lanes and images against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import SCENES

NCPU = os.cpu_count() or 1
IMG_TOL = 1e-3   # BASELINE north_star: <= 1e-3 relative L-infinity on the image


def rel_linf(a, ref):
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))

XML = ('<scene version="3.0.0"><integrator type="dopplertofpath"><integer name="max_depth" value="3"/><integer name="path_correlation_depth" value="%d"/></integrator>'
       '<sensor type="%s"><float name="fov" value="30"/>%s<float name="near_clip" value="0.5"/><float name="far_clip" value="50"/>'
       '<transform name="to_world"><lookat origin="0.5, 1, 6" target="0, 0.5, 0" up="0, 1, 0"/>%s</transform>'
       '<float name="shutter_open" value="0"/><float name="shutter_close" value="0.0015"/>'
       '<sampler type="correlated"><integer name="sample_count" value="8"/></sampler>'
       '<film type="hdrfilm"><integer name="width" value="24"/><integer name="height" value="16"/><rfilter type="tent"/></film></sensor>'
       '<shape type="rectangle"><transform name="to_world"><scale value="3"/></transform><bsdf type="diffuse"><rgb name="reflectance" value="0.6, 0.5, 0.4"/></bsdf></shape>'
       '<emitter type="point"><point name="position" x="0" y="2" z="4"/><rgb name="intensity" value="30"/></emitter></scene>')
LENS = '<float name="aperture_radius" value="0.3"/><float name="focus_distance" value="5.5"/>'


def scene_text(plugin="thinlens", lens=LENS, extra_xf="", pcd=2):
    return XML % (pcd, plugin, lens, extra_xf)


def oracle_ray(orc, sc, ux, uy, ax, ay):
    out = np.zeros(7, np.float32)
    orc.lib().orc_camera_sample_ray(C.byref(sc.c.sensor), ux, uy, ax, ay, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[0:3].astype(np.float64), out[3:6].astype(np.float64), float(out[6])


def test_thinlens_rays_meet_on_the_focal_plane(orc):
    """thinlens.cpp:257-305: every aperture sample of one film position passes through the same point of the plane z = focus_distance (camera space);
    the origin is the lens point advanced to the near plane; the centre of the aperture reproduces the pinhole direction; maxt spans near .. far."""
    sc = orc.Scene(scene_text(), is_string=True)
    pin = orc.Scene(scene_text("perspective", ""), is_string=True)
    se = sc.flat.sensor
    assert se["kind"] == 1 and float(se["aperture_radius"]) == np.float32(0.3) and float(se["focus_distance"]) == 5.5
    to_world = np.asarray(se["to_world"], np.float64).reshape(4, 4)
    inv = np.linalg.inv(to_world)
    rng = np.random.default_rng(5)
    for ux, uy in rng.random((6, 2)):
        o0, d0, _ = oracle_ray(orc, sc, ux, uy, .5, .5)
        po, pd, _ = oracle_ray(orc, pin, ux, uy, .5, .5)
        assert np.allclose(d0, pd, atol=1e-6)
        dl0 = inv[:3, :3] @ d0
        focus = (inv @ np.append(o0, 1))[:3] + dl0 * ((5.5 - (inv @ np.append(o0, 1))[2]) / dl0[2])
        for ax, ay in rng.random((8, 2)):
            o, d, maxt = oracle_ray(orc, sc, ux, uy, ax, ay)
            ol, dl = (inv @ np.append(o, 1))[:3], inv[:3, :3] @ d
            assert abs(np.linalg.norm(d) - 1) < 1e-6
            assert abs(ol[2] - 0.5) < 1e-5                                     # on the near plane
            lens = ol - dl * (0.5 / dl[2])                                     # back to z = 0: inside the aperture
            assert abs(lens[2]) < 1e-6 and np.hypot(lens[0], lens[1]) <= 0.3 + 1e-6
            hit = ol + dl * ((5.5 - ol[2]) / dl[2])
            assert np.allclose(hit, focus, atol=2e-5)
            assert abs(maxt - (50 - 0.5) / dl[2]) < 1e-3 * maxt


def test_aperture_sample_is_drawn_between_jitter_and_time(orc):
    """integrator.cpp:486-496: the lens draw is a correlated 2-D draw made after the pixel jitter and before the time sample, so against the pinhole
    camera of the same scene the jitter is unchanged while later draws move; with path_correlation_depth > 0 the two lanes of a correlated pair share
    their lens point (next_2d_correlate, correlated.cpp:156-167), without it they do not."""
    for pcd in (2, 0):
        lens = orc.Scene(scene_text(pcd=pcd), is_string=True)
        pin = orc.Scene(scene_text("perspective", "", pcd=pcd), is_string=True)
        a = lens.render_lanes(lens.params(), 7, 8, 0, 256, threads=1)
        b = pin.render_lanes(pin.params(), 7, 8, 0, 256, threads=1)
        assert np.array_equal(a["sample_pos"], b["sample_pos"])
        assert not np.array_equal(a["ray_o"], b["ray_o"])
        pair = a["ray_o"].reshape(-1, 2, 3)
        shared = np.all(pair[:, 0] == pair[:, 1], axis=1)
        if pcd:    # jitter and lens point shared: both rays of a pair start at the same point of the near plane unless their directions differ
            same_dir = np.all(a["ray_d"].reshape(-1, 2, 3)[:, 0] == a["ray_d"].reshape(-1, 2, 3)[:, 1], axis=1)
            assert same_dir.all() and shared.all()
        else:
            assert not shared.any()


@pytest.mark.parametrize("who", ["oracle", pytest.param("product", marks=pytest.mark.gpu)])
def test_thinlens_loader_errors(who, orc, request):
    """thinlens.cpp:138-156: aperture_radius is required, 0 becomes dr::Epsilon; focus_distance defaults to far_clip (sensor.cpp:134); scale factors in
    to_world are refused by both cameras (perspective.cpp:143-144)."""
    if who == "oracle":
        from oracle import scene_xml
        load, err = (lambda t: scene_xml.load(t, {}, is_string=True).sensor), ValueError
        field = lambda s, k: float(s[k])
    else:
        mi = request.getfixturevalue("mi")
        load, err = (lambda t: mi.load_string(t).export(2)), mi.DtofError
        field = lambda s, k: float(s[{"kind": 21, "aperture_radius": 22, "focus_distance": 23}[k]])
    s = load(scene_text(lens='<float name="aperture_radius" value="0"/>'))
    assert field(s, "kind") == 1 and field(s, "aperture_radius") == 2.0 ** -24 and field(s, "focus_distance") == 50.0
    s = load(scene_text("perspective", ""))
    assert field(s, "kind") == 0
    with pytest.raises(err, match="aperture_radius"):
        load(scene_text(lens='<float name="focus_distance" value="2"/>'))
    with pytest.raises(err, match="Scale factors in the camera-to-world transformation are not allowed"):
        load(scene_text(extra_xf='<scale value="1.5"/>'))
    with pytest.raises(err, match="Scale factors in the camera-to-world transformation are not allowed"):
        load(scene_text("perspective", "", extra_xf='<scale x="1" y="1.01" z="1"/>'))
    # ProjectiveCamera reads focus_distance for every projective sensor (sensor.cpp:134): the pinhole camera accepts and ignores it ...
    assert field(load(scene_text("perspective", '<float name="focus_distance" value="2"/>')), "kind") == 0
    if who == "product":   # ... while aperture_radius is nobody's there (xml.cpp:1204-1215; the oracle's reader does not track queried properties)
        with pytest.raises(err, match='unreferenced property .*"aperture_radius"'):
            load(scene_text("perspective", LENS))


ORTHO_XF = '<scale x="2.5" y="2" z="1"/>'   # the extent of an orthographic view is the scale of to_world: a 5 x 4 window here


def ortho_text(pcd=2):
    return scene_text("orthographic", "", ORTHO_XF, pcd).replace('<float name="fov" value="30"/>', "")


def test_orthographic_rays_are_parallel_and_start_on_the_near_plane(orc):
    """orthographic.cpp:169-196 with orthographic_projection (sensor.h:266-299): one direction (the normalised image of +z), origins on the plane
    z = near_clip of camera space, spread linearly over [-1, 1] x [-1 / aspect, 1 / aspect] (x flipped: sample (0, 0) is the top-left pixel, camera +x
    points left) times the scale of to_world; maxt = far - near; no aperture draw."""
    sc = orc.Scene(ortho_text(), is_string=True)
    se = sc.flat.sensor
    assert se["kind"] == 2
    to_world = np.asarray(se["to_world"], np.float64).reshape(4, 4)
    inv = np.linalg.inv(to_world)
    zdir = to_world[:3, 2] / np.linalg.norm(to_world[:3, 2])
    aspect = 24 / 16
    for ux, uy in [(0, 0), (1, 1), (.5, .5), (.25, .8)]:
        o, d, maxt = oracle_ray(orc, sc, ux, uy, .1, .9)
        assert np.allclose(d, zdir, atol=1e-6) and abs(maxt - 49.5) < 1e-4
        ol = (inv @ np.append(o, 1))[:3]
        assert np.allclose(ol, [1 - 2 * ux, (1 - 2 * uy) / aspect, 0.5], atol=1e-5)
    lanes = sc.render_lanes(sc.params(), 7, 8, 0, 64, threads=1)
    pin = orc.Scene(scene_text("perspective", ""), is_string=True)
    ref = pin.render_lanes(pin.params(), 7, 8, 0, 64, threads=1)
    assert np.array_equal(lanes["sample_pos"], ref["sample_pos"]) and np.array_equal(lanes["time"], ref["time"])   # same draws as the pinhole camera
    assert np.all(lanes["ray_d"] == lanes["ray_d"][0])


@pytest.mark.gpu
def test_orthographic_lanes_and_image_match_the_oracle(mi, orc):
    """bit-identical lanes and an image within 1e-3 for the orthographic camera (Doppler and plain path integrators); the loader takes the scale
    in to_world that the perspective cameras refuse"""
    for integ in ("dopplertofpath", "path"):
        text = ortho_text().replace('type="dopplertofpath"', 'type="%s"' % integ)
        if integ == "path":
            text = text.replace('<integer name="path_correlation_depth" value="2"/>', "")
        sc, osc = mi.load_string(text), orc.Scene(text, is_string=True)
        assert float(sc.export(2)[21]) == 2.0
        n = 24 * 16 * 8
        ours = sc.sample_lanes(3, 8, 0, n)
        ref = osc.render_lanes(osc.params(), 3, 8, 0, n, threads=NCPU)
        for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
            assert np.array_equal(np.ascontiguousarray(ours[k]).view(np.uint32), np.ascontiguousarray(ref[k], np.float32).view(np.uint32)), (integ, k)
        assert (ref["rgb"] != 0).any()
        img = sc.render(seed=3, spp=8)
        exp, _ = osc.render(osc.params(), seed=3, spp=8, threads=NCPU)
        assert rel_linf(img, exp) <= IMG_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("pcd", [2, 0])
def test_thinlens_lanes_and_image_match_the_oracle(mi, orc, pcd):
    """bit-identical lanes (film position, time, ray origin and direction, radiance) and an image within 1e-3 for the thin-lens camera, with and
    without correlated pixel / lens draws, for the Doppler integrator and for the plain path integrator (its own branch of render_sample)."""
    for integ in ("dopplertofpath", "path"):
        text = scene_text(pcd=pcd).replace('type="dopplertofpath"', 'type="%s"' % integ)
        if integ == "path":
            text = text.replace('<integer name="path_correlation_depth" value="%d"/>' % pcd, "")
        sc, osc = mi.load_string(text), orc.Scene(text, is_string=True)
        n = 24 * 16 * 8
        ours = sc.sample_lanes(3, 8, 0, n)
        ref = osc.render_lanes(osc.params(), 3, 8, 0, n, threads=NCPU)
        for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
            assert np.array_equal(np.ascontiguousarray(ours[k]).view(np.uint32), np.ascontiguousarray(ref[k], np.float32).view(np.uint32)), (integ, k)
        img = sc.render(seed=3, spp=8)
        exp, _ = osc.render(osc.params(), seed=3, spp=8, threads=NCPU)
        assert rel_linf(img, exp) <= IMG_TOL


@pytest.mark.gpu
def test_thinlens_blurs_out_of_focus_edges_only(mi):
    """depth of field as an image property (plain `path` integrator, direct light of a point source: noise-free up to the pixel jitter, which the
    lens draw leaves unchanged): a vanishing aperture reproduces the pinhole image; focused on the rectangle a 0.3 aperture keeps its silhouette
    close to the pinhole one; focused far in front of it the silhouette smears over several pixels."""
    def render(plugin, lens):
        text = scene_text(plugin, lens).replace('type="dopplertofpath"', 'type="path"').replace('<integer name="path_correlation_depth" value="2"/>', "") \
                                       .replace('name="max_depth" value="3"', 'name="max_depth" value="2"').replace('<scale value="3"/>', '<scale value="0.8"/>')
        return mi.load_string(text).render(seed=1, spp=64)
    pin = render("perspective", "")
    assert pin.max() > 0 and (pin[:, :, 0] == 0).any()                        # the rectangle does not fill the frame: there are edges to blur
    tiny = render("thinlens", '<float name="aperture_radius" value="0"/><float name="focus_distance" value="6.04"/>')
    assert rel_linf(tiny, pin) <= 1e-4
    sharp = render("thinlens", '<float name="aperture_radius" value="0.3"/><float name="focus_distance" value="6.04"/>')
    soft = render("thinlens", '<float name="aperture_radius" value="0.3"/><float name="focus_distance" value="2.0"/>')
    assert np.isfinite(soft).all()
    assert rel_linf(soft, pin) > 3 * rel_linf(sharp, pin)
    assert abs(float(soft.sum()) / float(pin.sum()) - 1) < 0.1               # blur redistributes the energy, it does not create any
