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
    sizes = {}
    for half in (True, False):
        for compression in ("zip", "zips", "none"):
            p = str(tmp_path / ("a_%d_%s.exr" % (half, compression)))
            io.write_exr(p, img, half=half, compression=compression)
            ch, attrs = exr_piz.read_exr(p)
            back = np.stack([ch["R"], ch["G"], ch["B"]], -1)
            exp = img.astype(np.float16).astype(np.float32) if half else img
            assert np.array_equal(back, exp), (half, compression)
            sizes[half, compression] = os.path.getsize(p)
    smooth = np.tile(np.linspace(0, 1, 64, dtype=np.float32)[None, :, None], (40, 1, 3))      # compressible content: ZIP must shrink it
    io.write_exr(str(tmp_path / "s_zip.exr"), smooth); io.write_exr(str(tmp_path / "s_none.exr"), smooth, compression="none")
    assert os.path.getsize(str(tmp_path / "s_zip.exr")) < 0.5 * os.path.getsize(str(tmp_path / "s_none.exr"))
    ch, _ = exr_piz.read_exr(str(tmp_path / "s_zip.exr"))
    assert np.array_equal(ch["G"], smooth[..., 1].astype(np.float16).astype(np.float32))
    with pytest.raises(ValueError, match="unsupported OpenEXR compression"):
        io.write_exr(str(tmp_path / "x.exr"), img, compression="dwaa")
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
    # --gpus G: one host thread per GPU, interleaved stripes, one RCCL reduce of the films to GPU 0.  This box has one GPU: more is an error;
    # DTOF_CLI_FORCE_RCCL takes the collective path with a one-rank communicator (ncclCommInitAll, ncclReduce, device-side develop), and
    # DTOF_CLI_SHARE_GPU (development switch) puts three shards on GPU 0 so that the thread / stripe logic runs (host sum: RCCL cannot
    # place two ranks on one device).
    out1 = str(tmp_path / "o1.npy")
    r = subprocess.run([exe, os.path.join(SCENES, "cornell_wall.xml"), "-D", "resx=40", "-D", "resy=24", "--spp", "8", "--seed", "5", "-o", out1, "--gpus", "1"],
                       capture_output=True, text=True, env=dict(os.environ, DTOF_CLI_FORCE_RCCL="1"))
    assert r.returncode == 0, r.stderr
    img1 = np.load(out1)
    assert img1.shape == (24, 40, 3) and np.abs(img1 - ref).max() <= 5e-5 * np.abs(ref).max()
    many = subprocess.run([exe, os.path.join(SCENES, "cornell_wall.xml"), "--gpus", "64"], capture_output=True, text=True)
    assert many.returncode != 0 and "GPU(s) are visible" in many.stderr
    out3 = str(tmp_path / "o3.npy")
    r = subprocess.run([exe, os.path.join(SCENES, "cornell_wall.xml"), "-D", "resx=40", "-D", "resy=24", "--spp", "8", "--seed", "5", "-o", out3,
                        "--gpus", "3", "--stripes", "5"], capture_output=True, text=True, env=dict(os.environ, DTOF_CLI_SHARE_GPU="1"))
    assert r.returncode == 0, r.stderr
    img3 = np.load(out3)
    assert img3.shape == (24, 40, 3) and np.abs(img3 - ref).max() <= 5e-5 * np.abs(ref).max()


@pytest.mark.gpu
def test_native_cli_over_two_gpus_reproduces_the_single_gpu_image(mi, tmp_path):
    """dtof-render --gpus 2 on a node with at least two GPUs: two host threads, two RCCL ranks (ncclCommInitAll), interleaved stripes, ONE ncclReduce of the
    films over xGMI, develop on GPU 0.  The driver's GPU box has one GPU: skipped there (the one-rank communicator and the shared-GPU switch cover the code
    path in test_native_cli_writes_the_same_image); first hardware run of the 2-rank collective happens wherever this test finds two devices."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    exe = os.path.join(ROOT, "mitsuba3dopplertof_amd", "dtof-render")
    scene = os.path.join(SCENES, "domino_small.xml")
    ref = mi.load_file(scene, resx=64, resy=48).render(seed=3, spp=16)
    for stripes in (4, 7):
        out = str(tmp_path / ("two_%d.npy" % stripes))
        r = subprocess.run([exe, scene, "-D", "resx=64", "-D", "resy=48", "--spp", "16", "--seed", "3", "-o", out, "--gpus", "2", "--stripes", str(stripes)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        img = np.load(out)
        assert img.shape == ref.shape and np.abs(img - ref).max() <= 5e-5 * np.abs(ref).max()


@pytest.mark.gpu
def test_native_cli_reports_a_failing_rank_instead_of_hanging(tmp_path):
    """a rank that cannot load the scene must not leave the others waiting in the collective: every rank finishes what can fail, all meet at a host
    barrier, and the reduce is entered by all or by none (dtof_cli.cpp).  Run with the shared-GPU development switch so that three ranks exist on one GPU."""
    exe = os.path.join(ROOT, "mitsuba3dopplertof_amd", "dtof-render")
    # one sample per pixel under per-interval stratification with time_correlate_number = 2: the render call of every rank fails (the scene itself loads)
    r = subprocess.run([exe, os.path.join(SCENES, "cornell_wall.xml"), "-D", "resx=16", "-D", "resy=16", "--spp", "1", "--gpus", "3",
                        "-o", str(tmp_path / "x.npy")], capture_output=True, text=True, timeout=120, env=dict(os.environ, DTOF_CLI_SHARE_GPU="1"))
    assert r.returncode != 0 and "Error: GPU" in r.stderr and "sample count" in r.stderr, r.stderr


def _read_png(path):
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xffffffff
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 2)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def test_png_previews_and_experiment_grids(tmp_path):
    """image_utils.py:90-135 (colour-mapped previews) and main_experiment.py:74-139 (the four experiment grids)"""
    from mitsuba3dopplertof_amd import experiments as E, io
    ramp = np.tile(np.linspace(-5, 5, 64, dtype=np.float32), (8, 1))
    io.save_speed_image(ramp, str(tmp_path / "v.png"))
    px = _read_png(str(tmp_path / "v.png")).astype(int)
    assert px.shape == (8, 64, 3)
    assert px[0, 0, 0] > px[0, 0, 2] + 50 and px[0, -1, 2] > px[0, -1, 0] + 50 and px[0, 32].min() > 220     # red .. white .. blue
    io.save_tof_image(ramp, str(tmp_path / "t.png"))
    t = _read_png(str(tmp_path / "t.png")).astype(int)
    assert t[0, 2, 2] > t[0, 2, 1] and t[0, -2, 0] > 200 and t[0, -2, 1] > 200                                # purple .. yellow
    io.save_hdr_image(np.abs(ramp)[..., None] * np.ones(3), str(tmp_path / "h.png"))
    assert _read_png(str(tmp_path / "h.png")).shape == (8, 64, 3)
    assert [len(E.experiment_settings(i)) for i in range(4)] == [1, 16, 12, 22]
    names = [s[1] for s in E.experiment_settings(2)]
    assert names[0] == "stratified_path_corr_depth_0_no_further_stratification" and all("uniform" not in n for n in names)
    s3 = E.experiment_settings(3)
    assert s3[5][1] == "antithetic_shift_0.5" and s3[5][3]["antithetic_shift"] == 0.5 and s3[-1][1] == "antithetic_mirror_shift_1.0"
    assert E.SCENE_CONFIGS["veach-ajar"] == {"max_depth": 8, "reference_spp": 131072, "spp": 1024}


@pytest.mark.gpu
def test_experiment_driver_layout_and_resume(mi, tmp_path):
    from mitsuba3dopplertof_amd import experiments as E
    base = str(tmp_path)
    scene = mi.load_file(os.path.join(SCENES, "cornell_boxes.xml"), resx=16, resy=16)
    log = []
    files = E.run_experiment(scene, "cornell-box", 3, base, grid=2, spp=8, log=log.append)
    # grid=2: frequencies {0,1} x offsets {0,1} x (2 methods x 2 shifts)
    assert len(files) == 16 and not log
    d = os.path.join(base, "results", "antithetic_shift_comparison", "cornell-box", "sinusoidal", "freq_1.000_offset_0.000")
    assert sorted(os.listdir(d)) == sorted(n + e for n in ("antithetic_shift_0.0", "antithetic_shift_1.0", "antithetic_mirror_shift_0.0",
                                                             "antithetic_mirror_shift_1.0") for e in (".npy", ".png"))
    a = np.load(os.path.join(d, "antithetic_shift_0.0.npy"))
    ref = mi.harness.run_scene_doppler_tof(scene, total_spp=8, hetero_frequency=1.0, hetero_offset=0.0, time_sampling_method="antithetic",
                                           path_correlation_depth=16, antithetic_shift=0.0, max_depth=4) if hasattr(mi, "harness") else None
    if ref is not None:
        assert np.abs(a - ref).max() <= 1e-5 * np.abs(ref).max()
    assert a.shape == (16, 16, 3) and np.isfinite(a).all()
    assert E.run_experiment(scene, "cornell-box", 3, base, grid=2, spp=8, log=log.append) == [] and len(log) == 16   # resumes
    # the command line (program entry of main_experiment.py)
    out = subprocess.run([sys.executable, "-m", "mitsuba3dopplertof_amd.experiments", "--scene_name", "cornell-box", "--expnumber", "0",
                          "--basedir", base, "--scene", os.path.join(SCENES, "cornell_boxes.xml"), "--grid", "2", "--reference_spp", "4",
                          "-D", "resx=8", "-D", "resy=8"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0 and "wrote 4 files" in out.stdout, out.stderr
    assert os.path.exists(os.path.join(base, "results", "gt_images", "cornell-box", "sinusoidal", "freq_0.000_offset_1.000", "reference.png"))


@pytest.mark.gpu
def test_animation_driver_reconstructs_the_wall_velocity(mi, tmp_path):
    """main_animation.py:58-157 on one synthetic frame: the Cornell room whose back wall moves at 10 m/s towards the camera.
    Ground-truth radial velocity on the wall = -10 m/s (velocity.cpp:125-142) and the heterodyne/homodyne reconstruction
    (image_utils.py:140-199) lands on it."""
    import shutil
    from mitsuba3dopplertof_amd import experiments as E
    base = str(tmp_path)
    d = os.path.join(base, "scenes_animation", "wall")
    os.makedirs(d)
    shutil.copy(os.path.join(SCENES, "cornell_wall.xml"), os.path.join(d, "animation_0.xml"))
    shutil.copy(os.path.join(SCENES, "cornell_wall.xml"), os.path.join(d, "no_animation_0.xml"))
    cfg = dict(max_depth=2, total_spp=512, animation_length=2, intervals=1, w_g=30, homodyne_spp=512)
    files = E.run_animation("wall", base, config=cfg, defines=dict(resx=24, resy=24), log=lambda m: None)
    assert len(files) == 2 + 2 + 6
    out = os.path.join(base, "results_animation", "wall")
    vel = np.load(os.path.join(out, "velocity_gt", "frame_0.npy"))
    centre = vel[8:16, 8:16, 0]
    assert np.allclose(centre, -10.0, atol=0.05)                       # the back wall fills the centre of the frame
    for name in ("velocity_gt/frame_0.png", "radiance/frame_0.png", "sinusoidal/freq_0.000_offset_0.250/frame_0.png",
                 "sinusoidal/antithetic_path_corr_depth_16/velocity/frame_0.png", "sinusoidal/uniform_path_corr_depth_0/velocity_0.000/frame_0.png"):
        assert os.path.getsize(os.path.join(out, name)) > 100, name
    T = 0.0015
    homo = [mi.to_tof_image(np.load(os.path.join(out, "sinusoidal", "freq_0.000_offset_%.3f" % o, "frame_0.npy")), T) for o in (0.0, 0.25)]
    het = [mi.to_tof_image(np.load(os.path.join(out, "sinusoidal", "antithetic_path_corr_depth_16", "freq_1.000_offset_%.3f" % o, "frame_0.npy")), T) for o in (0.0, 0.25)]
    v = E.calc_velocity_from_homo_heteros(homo, het, exposure_time=T, w_g=30)[8:16, 8:16]
    assert abs(np.median(v) + 10.0) < 2.5, np.median(v)
    assert E.run_animation("wall", base, config=cfg, defines=dict(resx=24, resy=24), log=lambda m: None) == []   # everything cached


def test_variant_selection_mirrors_the_tutorials_preamble(mi):
    mi.set_variant("cuda_rgb")                       # program_runner.py:2
    mi.set_variant("scalar_spectral", "llvm_rgb")    # first usable wins
    assert mi.variant() == "hip_rgb" and mi.variants() == ["hip_rgb"]
    for bad in ("cuda_spectral", "cuda_ad_rgb", "llvm_mono_polarized"):
        with pytest.raises(ImportError, match="unsupported variant"):
            mi.set_variant(bad)


@pytest.mark.gpu
def test_the_tutorial_call_sequence_runs_as_is(mi):
    """What program_runner.py:11-31,124-146 does with `mitsuba`, done with this package under the same alias: set_variant,
    load_file, load_dict of the integrator dictionary, integrator.render(scene, seed=i, spp=n) per pass, in-place accumulation."""
    mi.set_variant("cuda_rgb")
    scene = mi.load_file(os.path.join(SCENES, "cornell_boxes.xml"), resx=24, resy=24)
    integrator = mi.load_dict({"type": "dopplertofpath", "is_doppler_integrator": True, "max_depth": 4, "w_g": 30, "time": 0.0015,
                               "hetero_frequency": 1.0, "hetero_offset": 0.0, "antithetic_shift": 0.5, "time_sampling_method": "antithetic",
                               "path_correlation_depth": 16, "low_frequency_component_only": True, "wave_function_type": "sinusoidal",
                               "use_stratified_sampling_for_each_interval": True})
    total = None
    for i in range(3):
        img = integrator.render(scene, seed=i, spp=16)
        if i == 0:
            total = img
        else:
            total += img
    mean = total / 3
    assert mean.shape == (24, 24, 3) and np.isfinite(mean).all()
    assert np.allclose(mean, mi.render_multi_pass(scene, integrator, 48, 16), rtol=0, atol=1e-6 * np.abs(mean).max())


def test_error_tables_of_the_experiment_grids(tmp_path):
    """analysis.export_error / error_curves / plot_experiment against main_plot.py:20-104 restated with plain numpy on a synthetic 3 x 3 grid:
    the six error columns (PSNR = 10 log10(range^2 / MSE) of the exposure-scaled images), the per-frequency mean and ddof-1 deviation over the
    offsets, the experiment names and folders of the three figures, the resume switch."""
    from mitsuba3dopplertof_amd import analysis
    rng = np.random.default_rng(11)
    base = str(tmp_path)
    family, out_family, names = analysis.experiment_expnames(1)
    assert family == "time_spatial_sampling_comparison" and out_family == "time_spatial_sampling_comparison_full_plot"
    assert names[:3] == ["uniform_path_corr_depth_0", "uniform_path_corr_depth_16", "stratified_path_corr_depth_0"] and len(names) == 8
    assert analysis.experiment_expnames(2)[2][:3] == ["uniform_path_corr_depth_16", "stratified_path_corr_depth_16", "stratified_path_corr_depth_16_no_further_stratification"]
    assert analysis.experiment_expnames(3, "antithetic_mirror")[2][3] == "antithetic_mirror_shift_0.3" and len(analysis.experiment_expnames(3)[2]) == 11
    grid, T = 3, 0.0015
    images = {}
    for f in np.linspace(0, 1, grid):
        for o in np.linspace(0, 1, grid):
            cell = "freq_%.3f_offset_%.3f" % (f, o)
            ref = rng.normal(size=(6, 5, 3)).astype(np.float32)
            d = os.path.join(base, "results", "gt_images", "cornell-box", "sinusoidal", cell); os.makedirs(d)
            np.save(os.path.join(d, "reference.npy"), ref)
            d = os.path.join(base, "results", family, "cornell-box", "sinusoidal", cell); os.makedirs(d)
            for k, n in enumerate(names):
                img = (ref + rng.normal(scale=0.05 * (k + 1), size=ref.shape)).astype(np.float32)
                np.save(os.path.join(d, n + ".npy"), img)
                images[(round(float(f), 3), round(float(o), 3), n)] = (img, ref)
    tables = analysis.plot_experiment(1, base, ["cornell-box"], ["sinusoidal"], grid=grid, log=lambda *_: None)
    rows = tables["cornell-box/sinusoidal"]
    assert len(rows) == grid * grid * len(names)
    out_dir = os.path.join(base, "results", out_family, "cornell-box", "sinusoidal")
    back = analysis.read_result(os.path.join(out_dir, "result.csv"))
    assert back == rows and list(back[0]) == list(analysis.COLUMNS)
    for r in rows[::7]:
        img, ref = images[(round(r["freq"], 3), round(r["offset"], 3), r["expname"])]
        a, b = img * np.float32(T), ref * np.float32(T)
        mae, rmse = np.mean(np.abs(a - b)), np.sqrt(np.mean((a - b) ** 2))
        assert np.isclose(r["MAE"], mae, rtol=1e-6) and np.isclose(r["RMSE"], rmse, rtol=1e-6)
        assert np.isclose(r["RelativeMAE"], mae / np.mean(np.abs(b)), rtol=1e-6) and np.isclose(r["RelativeRMSE"], rmse / np.mean(np.abs(b)), rtol=1e-6)
        assert np.isclose(r["SNR"], -10 * np.log10(rmse / np.mean(np.abs(b))), rtol=1e-6)
        mse64 = np.mean((b.astype(np.float64) - a.astype(np.float64)) ** 2)
        assert np.isclose(r["PSNR"], 10 * np.log10(float(b.max() - b.min()) ** 2 / mse64), rtol=1e-9)
    curves = analysis.error_curves(rows, names, "freq", "RMSE")
    x, y, sd = curves[names[2]]
    assert x.tolist() == [0.0, 0.5, 1.0]
    v = np.array([r["RMSE"] for r in rows if r["expname"] == names[2] and r["freq"] == 0.5])
    assert len(v) == grid and np.isclose(y[1], v.mean()) and np.isclose(sd[1], v.std(ddof=1))
    assert curves[names[7]][1].mean() > curves[names[0]][1].mean()          # the noisier experiment has the larger error
    xs, ys, ss = analysis.error_curves(rows, names, "offset", "PSNR", other_value=0.5)[names[0]]
    assert np.all(ss == 0) and np.isclose(ys[0], [r["PSNR"] for r in rows if r["expname"] == names[0] and r["freq"] == 0.5 and r["offset"] == 0.0][0])
    try:
        import matplotlib  # noqa: F401
        assert os.path.getsize(os.path.join(base, "results", out_family, "plot_total.png")) > 10000
    except ImportError:
        pass
    # main_show_image.py: relative RMSE of the luminance images at offset 0 over the frequencies
    strip = analysis.show_image(names[:2], os.path.join(base, "results", family), "cornell-box/sinusoidal", os.path.join(base, "results", "images_over_hetero_frequency"),
                                os.path.join(base, "results", "gt_images"), grid=grid, log=lambda *_: None)
    img, ref = images[(0.5, 0.0, names[1])]
    lum = lambda v: (0.2126 * v[..., 0] + 0.7152 * v[..., 1] + 0.0722 * v[..., 2]) * T
    assert np.isclose(strip[names[1]][1], np.sqrt(np.mean((lum(img) - lum(ref)) ** 2)) / np.sqrt(np.mean(lum(ref) ** 2)), rtol=1e-5) and len(strip[names[0]]) == grid
    assert analysis.main(["--expnumber", "1", "--basedir", base, "--scene_names", "cornell-box", "--grid", str(grid), "--show_images", "--no_plots"]) == 0
    # exit_if_file_exists: the table is read back, not recomputed
    os.remove(os.path.join(base, "results", "gt_images", "cornell-box", "sinusoidal", "freq_0.000_offset_0.000", "reference.npy"))
    again = analysis.export_error(os.path.join(base, "results", family), "cornell-box/sinusoidal", names, os.path.join(base, "results", out_family),
                                  os.path.join(base, "results", "gt_images"), grid - 1, grid - 1, exit_if_file_exists=True)
    assert again == rows
    assert analysis.main(["--expnumber", "1", "--basedir", base, "--scene_names", "cornell-box", "--grid", str(grid), "--no_plots"]) == 0


@pytest.mark.gpu
def test_fused_splat_and_splat_kernel_agree_to_the_rounding_of_their_sums(mi, monkeypatch):
    """the fused splat (the wave of a 64-spp pixel reduces its 36 footprint values with a butterfly and issues the atomics) sums the samples of a pixel in another
    order than k_splat_x8, so the two films differ in the last bits.  The bound: every film value is a sum of <= 9 x 64 terms t_i; reordering a float32 sum moves it by at
    most ~n eps sum|t_i|.  sum|t_i| is not available per pixel, but the W channel is a sum of POSITIVE terms of the same weights: there the two paths must agree to
    n eps relative, and the colour channels to n eps x (W-weighted bound on the radiance).  The stats say which path ran (n_fused_splat_launches)."""
    import torch
    sc = mi.load_file(os.path.join(SCENES, "cornell_wall.xml"), resx=96, resy=64)
    W, H = sc.size
    films = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("DTOF_FUSE_SPLAT", fuse)
        film = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        st = sc.render_rows(film.data_ptr(), 7, 64, 0, H)
        assert (st["n_fused_splat_launches"] > 0) == (fuse == "1")
        films[fuse] = film.cpu().numpy().astype(np.float64)
    a, b = films["1"], films["0"]
    eps, n = 2.0 ** -24, 9 * 64
    assert np.abs(a[..., 3] - b[..., 3]).max() <= n * eps * b[..., 3].max()
    lanes = sc.sample_lanes(7, 64, 0, W * H * 64)["rgb"]
    bound = n * eps * b[..., 3].max() * np.abs(lanes).max()       # |t_i| <= weight x max |radiance|
    assert np.abs(a[..., :3] - b[..., :3]).max() <= bound
    assert not np.array_equal(a, b)                               # they ARE different sums: if this ever fails the test no longer tests anything
    # 256 spp (four waves per pixel) and K = 4 films never fuse
    monkeypatch.setenv("DTOF_FUSE_SPLAT", "1")
    assert sc.render_rows(torch.zeros((H, W, 4), dtype=torch.float32, device="cuda").data_ptr(), 7, 256, 0, H)["n_fused_splat_launches"] == 0
    assert sc.render_rows(torch.zeros((4, H, W, 4), dtype=torch.float32, device="cuda").data_ptr(), 7, 64, 0, H, offsets=[0, .25, .5, .75])["n_fused_splat_launches"] == 0


@pytest.mark.gpu
def test_async_frames_equal_synchronous_frames(mi):
    """dtof_render_rows_async / dtof_clear_async / dtof_develop_async / dtof_async_collect: frames enqueued back to back on the scene's stream give the films of the
    synchronous calls (same lanes; the film's float atomics add them in another order), one event-timed record per frame, launch counters summed"""
    import torch
    sc = mi.load_file(os.path.join(SCENES, "cornell_wall.xml"), resx=128, resy=96)
    W, H = sc.size
    film = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    ref = []
    for seed in (0, 1, 2):
        film.zero_(); torch.cuda.synchronize()
        st = sc.render_rows(film.data_ptr(), seed=seed, spp=16, row_begin=0, row_end=H)
        ref.append((film.cpu().numpy().copy(), st))
    films = [torch.zeros_like(film) for _ in range(3)]
    rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for seed in (0, 1, 2):
        sc.clear_async(films[seed].data_ptr(), films[seed].numel() * 4)
        sc.render_rows_async(films[seed].data_ptr(), seed, 16, 0, H)
    sc.develop_async(films[2].data_ptr(), rgb.data_ptr(), H * W)
    st, ms = sc.collect()
    assert len(ms) == 3 and np.all(ms > 0) and abs(st["ms_total"] - ms.sum()) < 1e-6
    assert st["n_paths"] == 3 * W * H * 16 and st["n_launches_first"] == sum(r[1]["n_launches_first"] for r in ref) and st["ms_first"] > 0
    for seed in (0, 1, 2):
        a, b = films[seed].cpu().numpy(), ref[seed][0]
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()
    dev = rgb.cpu().numpy()
    assert np.isfinite(dev).all() and np.abs(dev).max() > 0
    st2, ms2 = sc.collect()                    # nothing pending: an empty collect
    assert len(ms2) == 0 and st2["n_paths"] == 0
    img = sc.render(seed=0, spp=16)            # the synchronous path afterwards
    assert np.isfinite(img).all()
