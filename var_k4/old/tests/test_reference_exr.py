"""K6: the only artefact of this path the reference ships is its own render configs_example/scene.exr
(256x256, 1024 spp, antithetic shift 0.5, heterodyne, path_correlation_depth 4; README.md:86-89 `mitsuba scene.xml -m cuda_rgb`).
tools/exr_piz.py (a from-scratch PIZ/half OpenEXR reader) decoded it into
tests/golden/reference_configs_example_scene_exr.npy (float16, lossless: the file stores halfs).

The reference used another seed / back end, so the comparison is statistical: is the reference's image one of OUR renders
of the same scene (scenes/cornell_boxes.xml reproduces configs_example/scene.xml)?
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCENES

REF = os.path.join(GOLDEN, "reference_configs_example_scene_exr.npy")
REGIONS = {"image": (slice(0, 256), slice(0, 256)), "tall_box": (slice(110, 230), slice(60, 130)),
           "short_box": (slice(180, 245), slice(128, 200)), "back_wall": (slice(20, 100), slice(40, 220)),
           "floor": (slice(246, 256), slice(20, 240))}


def test_oracle_render_is_unbiased_against_the_reference_exr(orc):
    """CPU: one 256-spp oracle render; per region, mean(ours - ref) must vanish within its standard error (estimated
    from the spatial scatter of the per-pixel differences, which are independent across pixels)."""
    ref = np.load(REF).astype(np.float64)
    sc = orc.Scene(os.path.join(SCENES, "cornell_boxes.xml"), dict(resx=256, resy=256))
    img, _ = sc.render(sc.params(), seed=11, spp=256, threads=os.cpu_count())
    d = img.astype(np.float64) - ref
    for name, (ys, xs) in REGIONS.items():
        for c in range(3):
            r = d[ys, xs, c].ravel()
            z = r.mean() / (r.std(ddof=1) / np.sqrt(r.size))
            assert abs(z) < 4.5, (name, c, z)
    # and the images agree structurally: block-averaged correlation
    blk = lambda a: a.reshape(32, 8, 32, 8, 3).mean((1, 3))
    assert np.corrcoef(blk(img).ravel(), blk(ref).ravel())[0, 1] > 0.99


@pytest.mark.gpu
def test_reference_exr_is_statistically_one_of_our_renders(mi):
    """GPU: 32 seeds x 1024 spp.  Summary statistics of the reference image must fall inside our seed-to-seed
    distribution and the per-pixel z-scores must be standard normal (cf. the z-test of src/render/tests/test_renders.py)."""
    ref = np.load(REF).astype(np.float64)
    sc = mi.load_file(os.path.join(SCENES, "cornell_boxes.xml"), resx=256, resy=256)
    n = 32
    imgs = np.stack([sc.render(seed=100 + s, spp=1024).astype(np.float64) for s in range(n)])
    mean, sd = imgs.mean(0), imgs.std(0, ddof=1)
    for name, (ys, xs) in REGIONS.items():
        for c in range(3):
            ours = imgs[:, ys, xs, c].mean((1, 2))
            z = (ref[ys, xs, c].mean() - ours.mean()) / ours.std(ddof=1)
            assert abs(z) < 4.5, (name, c, z)
    z = (ref - mean) / np.sqrt(sd ** 2 * (1 + 1.0 / n) + (np.abs(ref) * 2.0 ** -11) ** 2 + 1e-30)
    assert abs(z.mean()) < 0.05 and 0.85 < z.std() < 1.2
    # the per-pixel sample distribution is heavy tailed and sd is itself estimated from n seeds: one of OUR seeds against
    # the others gives 0.997 / 0.9998 here
    assert (np.abs(z) < 4).mean() > 0.995 and (np.abs(z) < 6).mean() > 0.9995
    # VARIANCE per region: the antithetic time pairs and the correlated paths change the variance of a Doppler image, not its mean.  The
    # reference's residual against our mean must carry OUR per-pixel variance in every region and channel (z-scores of unit spread) ...
    for name, (ys, xs) in REGIONS.items():
        for c in range(3):
            zs = z[ys, xs, c].std()
            assert 0.8 < zs < 1.25, (name, c, zs)
    # ... and the check has power: with uncorrelated uniform time sampling (same mean, same scene) the per-pixel variance is several times
    # larger, and the reference's residual is NOT compatible with it
    sc.set_integrator(dict(type="dopplertofpath", max_depth=4, w_g=30.0, hetero_frequency=1.0, hetero_offset=0.0, time_sampling_method="uniform",
                           path_correlation_depth=0, wave_function_type="sinusoidal"))
    uni = np.stack([sc.render(seed=300 + s, spp=1024).astype(np.float64) for s in range(8)])
    var_uni, var_anti = uni.var(0, ddof=1), sd ** 2
    for name in ("back_wall", "tall_box", "short_box"):
        ys, xs = REGIONS[name]
        ratio = var_uni[ys, xs].mean() / var_anti[ys, xs].mean()
        resid = ((ref - mean)[ys, xs] ** 2).mean()
        assert ratio > 2.0, (name, ratio)
        assert 0.75 < resid / var_anti[ys, xs].mean() < 1.35 and resid / var_uni[ys, xs].mean() < 0.6, (name, resid / var_anti[ys, xs].mean(), resid / var_uni[ys, xs].mean())
