"""output writers, the tutorial-harness formulas and the command line (CPU parts; GPU parts marked)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENES


def test_exr_and_pfm_round_trip(tmp_path):
    from mitsuba3dopplertof_amd import io
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exr_piz
    rng = np.random.default_rng(1)
    img = rng.normal(0, 1e-3, (17, 23, 3)).astype(np.float32)
    for half in (True, False):
        p = str(tmp_path / ("a_%d.exr" % half))
        io.write_exr(p, img, half=half)
        ch, attrs = exr_piz.read_exr(p)
        back = np.stack([ch["R"], ch["G"], ch["B"]], -1)
        exp = img.astype(np.float16).astype(np.float32) if half else img
        assert np.array_equal(back, exp)
    p = str(tmp_path / "a.pfm")
    io.write_pfm(p, img)
    raw = open(p, "rb").read()
    head = b"PF\n23 17\n-1.0\n"
    assert raw.startswith(head)
    assert np.array_equal(np.frombuffer(raw[len(head):], "<f4").reshape(17, 23, 3)[::-1], img)
    io.write_image(str(tmp_path / "a.npy"), img)
    assert np.array_equal(np.load(str(tmp_path / "a.npy")), img)
    with pytest.raises(ValueError):
        io.write_image(str(tmp_path / "a.png"), img)


def test_reference_exr_fixture_matches_the_decoder_output():
    """The committed fixture is what tools/exr_piz.py decodes (checked when the reference tree is present)."""
    src = "/root/reference/configs_example/scene.exr"
    if not os.path.exists(src):
        pytest.skip("reference tree not mounted (GPU box)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exr_piz
    ch, attrs = exr_piz.read_exr(src)
    img = np.stack([ch["R"], ch["G"], ch["B"]], -1)
    fix = np.load(os.path.join(ROOT, "tests", "golden", "reference_configs_example_scene_exr.npy")).astype(np.float32)
    assert np.array_equal(img, fix)


def test_velocity_from_homodyne_heterodyne_closed_form():
    """A target receding at v shifts the illumination frequency by dw = -2 v w_g / c; with heterodyne frequency 1/T the
    sinusoidal correlation integrates to homodyne ~ cos(phi) sinc-like terms whose ratio inverts to v (image_utils.py:140-168)."""
    from mitsuba3dopplertof_amd import harness
    T, w_g = 0.0015, 30.0
    v_true = np.array([[-10.0, -3.0, 0.5, 8.0]])
    dw = -2.0 * v_true * (w_g * 1e6) / 3e8                       # Hz
    ratio = dw * T / (dw * T - 1.0)                              # heterodyne / homodyne for a perfect measurement
    homo = np.full_like(ratio, 2e-4)
    v = harness.calc_velocity_from_homo_hetero(homo, ratio * homo, exposure_time=T, w_g=w_g)
    assert np.allclose(v, v_true, rtol=1e-9)
    v2 = harness.calc_velocity_from_homo_heteros([homo, 2 * homo], [ratio * homo, 2 * ratio * homo], exposure_time=T, w_g=w_g)
    assert np.allclose(v2, v_true, rtol=1e-3)
    d = harness.doppler_integrator_dict(time_sampling_method="antithetic_mirror")
    assert d["antithetic_shift"] == 0.0 and harness.doppler_integrator_dict()["antithetic_shift"] == 0.5
    assert set(d) == {"type", "is_doppler_integrator", "max_depth", "w_g", "time", "hetero_frequency", "hetero_offset", "antithetic_shift",
                      "time_sampling_method", "path_correlation_depth", "low_frequency_component_only", "wave_function_type",
                      "use_stratified_sampling_for_each_interval"}


def test_cli_reports_loader_errors_like_the_reference_cli(tmp_path):
    bad = tmp_path / "bad.xml"
    bad.write_text('<scene version="3.0.0"><integrator type="volpath"/></scene>')
    r = subprocess.run([sys.executable, "-m", "mitsuba3dopplertof_amd", str(bad)], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 1 and "Error:" in r.stderr


@pytest.mark.gpu
def test_cli_and_harness_on_the_gpu(mi, tmp_path):
    from mitsuba3dopplertof_amd import harness
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exr_piz
    out = str(tmp_path / "frame.exr")
    r = subprocess.run([sys.executable, "-m", "mitsuba3dopplertof_amd", os.path.join(SCENES, "cornell_boxes.xml"), "-D", "resx=32", "-D", "resy=24",
                        "--spp", "16", "--seed", "3", "-o", out, "-v"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ch, _ = exr_piz.read_exr(out)
    img = np.stack([ch["R"], ch["G"], ch["B"]], -1)
    ref = mi.load_file(os.path.join(SCENES, "cornell_boxes.xml"), resx=32, resy=24).render(seed=3, spp=16)
    assert np.abs(img - ref).max() <= 1e-3 * np.abs(ref).max()           # half-float storage
    # the tutorial pipeline on the moving-wall scene: ground-truth velocity vs the homodyne/heterodyne estimate
    sc = mi.load_file(os.path.join(SCENES, "cornell_wall.xml"), resx=32, resy=32)
    gt = harness.run_scene_velocity(sc, total_spp=16)
    assert np.all(np.abs(gt[12:20, 12:20, 0] + 10.0) < 0.15)
    kw = dict(total_spp=4096, time_sampling_method="antithetic", path_correlation_depth=16, max_depth=2)
    homo = mi.to_tof_image(harness.run_scene_doppler_tof(sc, hetero_frequency=0.0, **kw))
    hetero = mi.to_tof_image(harness.run_scene_doppler_tof(sc, hetero_frequency=1.0, **kw))
    v = harness.calc_velocity_from_homo_hetero(homo, hetero)
    centre = v[12:20, 12:20]
    assert abs(np.median(centre) + 10.0) < 2.5, np.median(centre)


@pytest.mark.gpu
def test_native_cli_writes_the_same_image(mi, tmp_path):
    """mitsuba3dopplertof_amd/dtof-render (C++ over the C ABI) == the Python binding, byte for byte up to film atomics."""
    exe = os.path.join(ROOT, "mitsuba3dopplertof_amd", "dtof-render")
    out = str(tmp_path / "o.npy")
    r = subprocess.run([exe, os.path.join(SCENES, "cornell_wall.xml"), "-D", "resx=40", "-D", "resy=24", "--spp", "8", "--seed", "5", "-o", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    img = np.load(out)
    ref = mi.load_file(os.path.join(SCENES, "cornell_wall.xml"), resx=40, resy=24).render(seed=5, spp=8)
    assert img.shape == (24, 40, 3) and np.abs(img - ref).max() <= 1e-5 * np.abs(ref).max()
    bad = subprocess.run([exe, os.path.join(SCENES, "cornell_wall.xml"), "-D", "wave_function_type=sawtooth"], capture_output=True, text=True)
    assert bad.returncode != 0 and "unknown wave_function_type" in bad.stderr
