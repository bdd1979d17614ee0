"""Pins the CPU oracle (oracle/) against every known-answer vector that exists for the building blocks of the
hot path and against the committed golden vectors.  CPU only.

The reference ships no test or fixture for dopplertofpath/correlated themselves, so the path as a
whole stays "parity unpinned"; what CAN be pinned is pinned here:
  K1  sample_tea_float32 -- the 8 exact values of src/core/tests/test_random.py:8-16
  K2  PCG32 -- O'Neill's published pcg32 demo vector (seed 42, stream 54) and the Dr.Jit default-seed stream
  K3  permute_kensler bijection for n = 2^(p+1)+p, p<3, 75 seeds (test_random.py:69-75)
  K4  closed-form waveform values (include/mitsuba/render/waveform_utils.h:24-62)
  K5  analytic properties of the modulation weight and of antithetic sampling
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCENES


def test_k1_tea_known_answers(orc):
    L = orc.lib()
    exp = {(1, 1): 0.5424730777740479, (1, 2): 0.5079904794692993, (1, 3): 0.4171961545944214,
           (1, 4): 0.008385419845581055, (1, 5): 0.8085528612136841, (2, 1): 0.6939879655838013,
           (3, 1): 0.6978365182876587, (4, 1): 0.4897364377975464}
    for (a, b), v in exp.items():
        assert L.orc_tea_float32(a, b, 4) == np.float32(v)


def test_k2_pcg32_known_answers(orc):
    L = orc.lib()
    st, inc = C.c_uint64(), C.c_uint64()
    L.orc_pcg32_seed(42, 54, C.byref(st), C.byref(inc))
    got = [L.orc_pcg32_next_u32(C.byref(st), inc) for _ in range(6)]
    assert got == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]
    # seed() must leave state/inc exactly as the published pcg32_srandom_r does
    L.orc_pcg32_seed(42, 54, C.byref(st), C.byref(inc))
    assert inc.value == (54 << 1) | 1
    # next_float32 = (u >> 9 | 0x3f800000) - 1 in [0, 1)
    f = [L.orc_pcg32_next_f32(C.byref(st), inc) for _ in range(1000)]
    assert min(f) >= 0.0 and max(f) < 1.0
    u = 0xa15c02b7
    assert f[0] == np.uint32((u >> 9) | 0x3f800000).view(np.float32) - np.float32(1)


def test_k3_kensler_is_a_permutation(orc):
    L = orc.lib()
    for p in range(3):
        n = 2 ** (p + 1) + p
        for seed in range(75):
            perm = sorted(L.orc_permute_kensler(i, n, seed) for i in range(n))
            assert perm == list(range(n))
    for n in (1, 2, 7, 32, 100, 513):
        for seed in (0, 1, 0xdeadbeef):
            assert sorted(L.orc_permute_kensler(i, n, seed) for i in range(n)) == list(range(n))


def test_k4_waveforms_closed_form(orc):
    L = orc.lib()
    pts = [0.0, math.pi / 2, math.pi, 3 * math.pi / 2]
    lp = {0: [1, 0, -1, 0], 1: [2, 0, -2, 0], 2: [2 / 3, 0, -2 / 3, 0], 3: [2, 0, -2, 0]}
    for wave, exp in lp.items():
        for t, e in zip(pts, exp):
            assert abs(L.orc_waveform_low_pass(t, wave) - e) < 2e-6, (wave, t)
    # non-low-pass: cos / square / triangle; trapezoid has no case and evaluates cos (waveform_utils.h:27-32)
    assert abs(L.orc_waveform(0.3, 0) - math.cos(0.3)) < 1e-6
    assert L.orc_waveform(0.1, 1) == 1.0 and L.orc_waveform(math.pi, 1) == -1.0 and L.orc_waveform(6.0, 1) == 1.0
    assert abs(L.orc_waveform(0.0, 2) - 1.0) < 1e-6 and abs(L.orc_waveform(math.pi, 2) + 1.0) < 1e-6
    assert abs(L.orc_waveform(0.3, 3) - math.cos(0.3)) < 1e-6
    # periodicity through fmod(t, 2pi)
    for wave in range(4):
        assert abs(L.orc_waveform_low_pass(1.0, wave) - L.orc_waveform_low_pass(1.0 + 2 * math.pi, wave)) < 1e-5
    # trapezoid is the clamped, doubled rectangle low-pass
    for t in np.linspace(0, 6.2, 50):
        r = L.orc_waveform_low_pass(float(t), 1)
        assert abs(L.orc_waveform_low_pass(float(t), 3) - min(max(2 * r, -2), 2)) < 1e-6


def test_sincos_matches_libm(orc):
    L = orc.lib()
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for x in np.linspace(-20, 20, 8001).astype(np.float32):
        L.orc_sincos(float(x), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - math.sin(float(x))), abs(c.value - math.cos(float(x))))
    assert worst < 2.5e-7


def test_k5_modulation_weight_properties(orc):
    L = orc.lib()
    sc = orc.Scene(os.path.join(SCENES, "cornell_boxes.xml"), dict(resx=8, resy=8))
    p0 = orc.make_params(sc.params(integrator=dict(type="dopplertofpath", hetero_frequency=1.0, hetero_offset=0.0)))
    p5 = orc.make_params(sc.params(integrator=dict(type="dopplertofpath", hetero_frequency=1.0, hetero_offset=0.5)))
    ph = orc.make_params(sc.params(integrator=dict(type="dopplertofpath", hetero_frequency=0.0)))
    for t, ln in [(0.0, 1.0), (0.0004, 7.3), (0.0012, 13.9)]:
        w0 = L.orc_modulation_weight(C.byref(p0), t, ln)
        w5 = L.orc_modulation_weight(C.byref(p5), t, ln)
        assert abs(w0 + w5) < 2e-6                       # hetero_offset 0 vs 0.5: sign flip (sinusoidal)
        # W = 0.25 cos(w_d t + phi), phi = 2 pi f/c L with f = 30 MHz, c = 300 m/us
        wd = 2 * math.pi / 0.0015
        assert abs(w0 - 0.25 * math.cos(wd * t + 2 * math.pi * 30 / 300 * ln)) < 5e-6
        assert abs(L.orc_modulation_weight(C.byref(ph), t, ln) - 0.25 * math.cos(2 * math.pi * 30 / 300 * ln)) < 5e-6


def test_constructor_rounding(orc):
    sc = orc.Scene(os.path.join(SCENES, "cornell_boxes.xml"), dict(resx=8, resy=8))
    pd = sc.params()
    assert pd["time"] == np.float32(0.0015) and pd["hetero_frequency"] == np.float32(1.0)
    assert pd["w_s_mhz"] == np.float32(30.0 + float(np.float32(1.0) / np.float32(0.0015)) * 1e-6)
    pd = sc.params(integrator=dict(type="dopplertofpath", w_s=30.001))
    assert pd["hetero_frequency"] == np.float32(float(np.float32(30.001) - np.float32(30.0)) * 1e6 * float(np.float32(0.0015)))
    # defaults: time_sampling_method "antithetic" with shift 0.5, other methods shift 0 (integrator.cpp:58,72-76)
    assert pd["time_sampling"] == 2 and pd["antithetic_shift"] == 0.5 and pd["max_depth"] == 0xffffffff and pd["rr_depth"] == 5
    assert sc.params(integrator=dict(type="dopplertofpath", time_sampling_method="stratified"))["antithetic_shift"] == 0.0
    with pytest.raises(ValueError):
        sc.params(integrator=dict(type="dopplertofpath", wave_function_type="sawtooth"))
    with pytest.raises(ValueError):
        sc.params(integrator=dict(type="volpath"))


def test_antithetic_pairs_cancel_on_a_static_scene(orc):
    """With shift 0.5 and hetero_frequency 1 the two lanes of a pair see cos(x) and cos(x + pi); with fully
    correlated paths a static scene integrates to ~0 (float32 rounding only), while uniform sampling does not."""
    xml = open(os.path.join(SCENES, "cornell_boxes.xml")).read().replace('z="0.015"', 'z="0.0"').replace('z="-0.015"', 'z="0.0"')
    sc = orc.Scene(xml, dict(resx=16, resy=16), is_string=True)
    img, _ = sc.render(sc.params(), seed=0, spp=16, threads=os.cpu_count())
    uni, _ = sc.render(sc.params(integrator=dict(type="dopplertofpath", max_depth=4, hetero_frequency=1.0,
                                                 time_sampling_method="uniform")), seed=0, spp=16, threads=os.cpu_count())
    assert np.abs(img).max() < 1e-6 < 1e-3 < np.abs(uni).max()


def test_lane_streams_are_tile_invariant(orc):
    """Every lane is a pure function of its global index: evaluating a sub-range reproduces the full wavefront."""
    sc = orc.Scene(os.path.join(SCENES, "cornell_wall.xml"), dict(resx=16, resy=16))
    pd = sc.params()
    full = sc.render_lanes(pd, 1, 8, 0, 16 * 16 * 8)
    part = sc.render_lanes(pd, 1, 8, 777, 300)
    assert np.array_equal(full[777:1077].tobytes(), part.tobytes())
    film_a, _ = sc.render(pd, seed=1, spp=8, raw=True)
    film_b, _ = sc.render(pd, seed=1, spp=8, rows=(0, 7), raw=True)
    film_c, _ = sc.render(pd, seed=1, spp=8, rows=(7, 16), raw=True)
    assert np.abs(film_a - (film_b + film_c)).max() <= 1e-5 * np.abs(film_a[..., :3]).max()


def test_film_weights_sum_to_spp_in_the_interior(orc):
    sc = orc.Scene(os.path.join(SCENES, "cornell_wall.xml"), dict(resx=16, resy=16))
    film, n = sc.render(sc.params(), seed=0, spp=8, raw=True)
    assert n == 16 * 16 * 8
    # the tent weights of a sample sum to 1 unless part of its footprint falls off the film
    assert 0.9 * n < film[..., 3].sum() <= n * (1 + 1e-5)
    assert np.all(film[2:-2, 2:-2, 3] > 0)


def test_oracle_reproduces_golden_vectors(orc, configs):
    """Regression pin: the committed vectors (tests/golden/make_golden.py) are what this oracle produces."""
    for name, xml, params, spp in configs:
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        sc = orc.Scene(os.path.join(SCENES, xml), params)
        pd = sc.params()
        n = g["lane_rgb"].shape[0]
        lanes = sc.render_lanes(pd, 3, spp, 0, n)
        for key, field in (("lane_rgb", "rgb"), ("lane_pos", "sample_pos"), ("lane_time", "time"), ("lane_ray_d", "ray_d")):
            assert np.array_equal(g[key].view(np.uint32), np.ascontiguousarray(lanes[field]).view(np.uint32)), (name, key)
        if name in ("c1_boxes_antithetic", "boxes_trap_depth6_spp6"):
            img, _ = sc.render(pd, seed=3, spp=spp, threads=1)         # one thread splats in lane order: bit-exact
            assert np.array_equal(img, g["image"]), name
            par, _ = sc.render(pd, seed=3, spp=spp, threads=os.cpu_count())   # row bands per thread: float addition order only
            assert np.abs(par - g["image"]).max() <= 1e-5 * np.abs(g["image"]).max(), name


def test_velocity_and_path_integrators_physical_sanity(orc):
    """SURVEY 8(f) #1.  The back wall of cornell_wall translates 0.015 towards the camera in 1.5 ms: the velocity integrator
    must report -10 m/s / cos(angle to the optical axis) there; `path` is the radiance image (positive, seed-stable mean)."""
    sc = orc.Scene(os.path.join(SCENES, "cornell_wall.xml"), dict(resx=32, resy=32))
    vel, _ = sc.render(sc.params(integrator=dict(type="velocity")), seed=0, spp=4, threads=os.cpu_count())
    centre = vel[12:20, 12:20, 0]
    assert np.all(np.abs(centre + 10.0) < 0.15)
    rad0, _ = sc.render(sc.params(integrator=dict(type="path", max_depth=4)), seed=0, spp=16, threads=os.cpu_count())
    rad1, _ = sc.render(sc.params(integrator=dict(type="path", max_depth=4)), seed=1, spp=16, threads=os.cpu_count())
    assert rad0.min() >= 0 and abs(rad0.mean() - rad1.mean()) < 0.03 * rad0.mean()
    # the Doppler image with w_g -> 0 and homodyne detection is 0.25 x the radiance image (W = 0.25 cos(0)) -- same streams
    # when the path integrator's plain draws are replaced by fully uncorrelated Doppler draws? No: different streams; compare means.
    dop, _ = sc.render(sc.params(integrator=dict(type="dopplertofpath", max_depth=4, w_g=0.0, hetero_frequency=0.0,
                                                 time_sampling_method="uniform")), seed=0, spp=16, threads=os.cpu_count())
    assert abs(4.0 * dop.mean() - rad0.mean()) < 0.05 * rad0.mean()


def test_area_light_mis_reproduces_the_analytic_direct_illumination(orc):
    """Emitter-hit + NEE combined by the power heuristic (dopplertofpath.cpp:150-168,214-226; here through `path`, max_depth 2)
    against a brute-force quadrature of the rendering equation: L_o = rho/pi * L_e * sum over the light of V cos cos' / r^2 dA,
    with V from the oracle's own occlusion query.  Wrong pdfs or MIS weights would break the energy balance."""
    L = orc.lib()
    xml = open(os.path.join(SCENES, "cornell_area.xml")).read().replace('<rfilter type="tent" />', '<rfilter type="box" />')
    sc = orc.Scene(xml, dict(resx=64, resy=64), is_string=True)
    img, _ = sc.render(sc.params(integrator=dict(type="path", max_depth=2)), seed=0, spp=1024, threads=os.cpu_count(), rows=(60, 61))
    radiance = np.array([17.0, 12.0, 4.0])
    checked = 0
    for px in (6, 10, 14):   # lit floor left of the boxes (penumbra pixels would need a much finer quadrature)
        expect, ok, sub = np.zeros(3), True, 6
        for sx in range(sub):                      # a pixel of row 60 covers ~0.2 of floor depth: integrate over its footprint
            for sy in range(sub):
                out = (C.c_float * 7)()
                L.orc_camera_ray(C.byref(sc.c.sensor), px + (sx + .5) / sub, 60 + (sy + .5) / sub, out)
                o, d = np.array(out[0:3], np.float64), np.array(out[3:6], np.float64)
                hit, ids = (C.c_float * 3)(), (C.c_int32 * 3)()
                L.orc_intersect(C.byref(sc.c), (C.c_float * 3)(*o), (C.c_float * 3)(*d), 0.0, out[6], hit, ids)
                if ids[0] != 0:      # object 0 is the floor (reflectance 0.725, 0.71, 0.68, normal +y)
                    ok = False
                    continue
                p = o + d * hit[0]
                acc, n = 0.0, 20
                for ix in range(n):
                    for iz in range(n):
                        q = np.array([-0.25 + 0.5 * (ix + .5) / n, 1.98, -0.2 + 0.4 * (iz + .5) / n])
                        w = q - p; r = np.linalg.norm(w); w /= r
                        so = p + np.array([0, 1e-3, 0])
                        if not L.orc_occluded(C.byref(sc.c), (C.c_float * 3)(*so), (C.c_float * 3)(*w), 0.0, float(r * 0.999)):
                            acc += w[1] * w[1] / (r * r)          # cos at the floor = cos at the light = w.y
                expect += np.array([0.725, 0.71, 0.68]) / np.pi * radiance * acc * (0.5 * 0.4 / (n * n)) / (sub * sub)
        if not ok:
            continue
        got = img[60, px]
        assert np.all(np.abs(got / expect - 1) < 0.05), (px, got, expect)
        checked += 1
    assert checked >= 3


def test_independent_and_timestratified_samplers(orc):
    """SURVEY 8(f) #4.  `independent` under the Doppler integrator = the main PCG32 stream only, uniform time
    (sampler.h:131-144); `timestratified` = one time sample per stratum of every pixel, strata visited in a Kensler
    permutation, jitter optional (timestratified.cpp:117-129); all other draws come from the same main stream."""
    osc = orc.Scene(os.path.join(SCENES, "cornell_boxes.xml"), dict(resx=8, resy=8))
    spp, n, T = 16, 8 * 8 * 16, np.float32(0.0015)
    ind = osc.render_lanes(osc.params(sampler=dict(type="independent")), 3, spp, 0, n)
    ts = osc.render_lanes(osc.params(sampler=dict(type="timestratified")), 3, spp, 0, n)
    nj = osc.render_lanes(osc.params(sampler=dict(type="timestratified", jitter=False)), 3, spp, 0, n)
    cor = osc.render_lanes(osc.params(integrator=dict(type="dopplertofpath", time_sampling_method="uniform", path_correlation_depth=0, max_depth=4)), 3, spp, 0, n)
    # pixel jitter = the first two draws of the main stream in all of them
    assert np.array_equal(ind["sample_pos"], ts["sample_pos"]) and np.array_equal(ind["sample_pos"], nj["sample_pos"])
    # correlated draws BOTH streams per call and returns main when uncorrelated: x is the same first draw, y is not
    assert np.array_equal(cor["sample_pos"][:, 0], ind["sample_pos"][:, 0])
    for lanes, exact in ((ts, False), (nj, True)):
        strata = np.floor(lanes["time"].reshape(-1, spp).astype(np.float64) / float(T) * spp + (1e-4 if exact else 0.0)).astype(int)
        assert np.array_equal(np.sort(strata, axis=1), np.tile(np.arange(spp), (strata.shape[0], 1)))
        if exact:
            assert np.allclose(lanes["time"].reshape(-1, spp) / T * spp - strata, 0.5, atol=1e-3)
    assert not np.array_equal(np.argsort(nj["time"].reshape(-1, spp), axis=1)[0], np.arange(spp))   # permuted, not in order
    u = ind["time"] / T
    assert 0.4 < u.mean() < 0.6 and u.min() >= 0 and u.max() < 1
    for lanes in (ind, ts, nj):
        assert np.isfinite(lanes["rgb"]).all() and (lanes["rgb"] != 0).any()


def test_periodic_and_regular_time_sampling(orc):
    """the two ETimeSampling values the reference declares (sampler.h:27-34) and implements in CorrelatedSampler::next_1d_time
    (correlated.cpp:147-152) but never parses.  Both draw from the shared time stream m_rng_time (one number per group of `tcn` lanes, :103-107) and go through
    the per-interval stratification of the non-stratified strategies (r = (s / tcn + r) / (spp / tcn), :121-124); `periodic` then adds (s % tcn) / tcn -- the tcn
    samples of a group sit exactly 1 / tcn of the exposure apart -- and `regular` returns r as it is: the samples of a group share ONE time."""
    import ctypes as C
    L = orc.lib()
    spp, n = 16, 16 * 64
    base = dict(time=0.0015, w_g_mhz=30.0, g_1=.5, g_0=.5, w_s_mhz=30.0, phase_offset=0.0, hetero_frequency=1.0, wave_type=0, low_frequency_component_only=1,
                antithetic_shift=0.0, path_correlation_depth=1, max_depth=4, rr_depth=5, hide_emitters=0, base_seed=3, path_correlate_number=2)

    def times(strategy, tcn, strat):
        p = orc.make_params(dict(base, time_sampling=strategy, stratify_each_interval=int(strat), time_correlate_number=tcn))
        ou, of = (C.c_uint32 * 7)(), (C.c_float * 3)()
        out = np.zeros(n, np.float32)
        for lane in range(n):
            L.orc_sampler_lane(C.byref(p), 9, spp, lane, ou, of); out[lane] = of[2]
        return out

    for tcn in (2, 4):
        n_stratum = spp // tcn
        for strat in (True, False):
            reg = times(5, tcn, strat).reshape(-1, spp // tcn, tcn)      # [pixel, group, member]
            per = times(4, tcn, strat).reshape(-1, spp // tcn, tcn)
            anti = times(2, tcn, strat).reshape(-1, spp // tcn, tcn)
            # regular: every member of a group has the group's number, bit for bit
            assert np.array_equal(reg, np.repeat(reg[:, :, :1], tcn, axis=2))
            assert reg.min() >= 0 and reg.max() < 1
            if strat:   # group g of a pixel lies in stratum g of its n_stratum strata
                assert np.array_equal(np.floor(reg[:, :, 0].astype(np.float64) * n_stratum).astype(int), np.tile(np.arange(n_stratum), (reg.shape[0], 1)))
            else:
                assert 0.4 < reg.mean() < 0.6
            # periodic: member m = the group's number + float32(m) / float32(tcn), the reference's operation order
            want = reg[:, :, :1] + (np.arange(tcn, dtype=np.float32) / np.float32(tcn))[None, None, :]
            assert np.array_equal(per, want.astype(np.float32))
            # ... which is what `antithetic` does for tcn != 2 (correlated.cpp:135-138); for tcn == 2 antithetic adds the shift (0 here) to the odd member instead
            if tcn != 2:
                assert np.array_equal(per, anti)
            else:
                assert np.array_equal(anti, reg)


@pytest.mark.parametrize("rfilter,exact", [('<rfilter type="tent" />', True), ('<rfilter type="mitchell" />', True),
                                           ('<rfilter type="catmullrom" />', True), ('<rfilter type="mitchell"><float name="B" value="0.2" /><float name="C" value="0.7" /></rfilter>', True),
                                           ("", False), ('<rfilter type="lanczos" />', False)])
def test_reconstruction_filters_are_partitions_of_unity(orc, rfilter, exact):
    """tent, Mitchell-Netravali (any B, C) and Catmull-Rom satisfy sum_k f(x + k) = 1, so the weight channel of a sample whose
    footprint lies inside the film sums to exactly one sample (src/rfilters/{tent,mitchell,catmullrom}.cpp); the default
    gaussian (stddev 0.5, cut at 4 sigma) only approximately."""
    text = open(os.path.join(SCENES, "cornell_wall.xml")).read().replace('<rfilter type="tent" />', rfilter)
    sc = orc.Scene(text, dict(resx=16, resy=16), is_string=True)
    film, n = sc.render(sc.params(), seed=0, spp=8, raw=True)
    # every sample of the 12 x 12 interior pixels spreads its unit weight inside the film: total = samples of ... not separable per
    # pixel, so compare the grand total over the film with the count of samples whose footprint (radius <= 2) cannot leave it
    total, inner = film[..., 3].sum(), film[3:-3, 3:-3, 3].sum()
    assert total <= n * (1 + (1e-5 if exact else 1.0))      # the unnormalised gaussian integrates to ~1.57 per sample
    if exact:
        # weights of samples in pixel rows/cols 3..12 land within rows/cols 1..14 and sum to 1 each: the mass that reaches the
        # 10 x 10 centre from outside equals the mass that leaves it, up to the random sample positions -> within 2 %
        assert abs(inner / (10 * 10 * 8) - 1.0) < 0.02


def test_multi_pass_layout_and_sample_index(orc):
    """orc_pass_layout restates integrator.cpp:121-135,227-245; one pass holding every sample equals the plain render; with stratified
    time sampling the strata of a pixel are still hit exactly once each over all passes (the sample index runs through the passes)."""
    import ctypes as C
    L = orc.lib()
    def layout(w, h, spp, per_pass):
        a, b = C.c_uint32(0), C.c_uint32(0)
        rc = L.orc_pass_layout(w, h, spp, per_pass & 0xffffffff, C.byref(a), C.byref(b))
        return rc, a.value, b.value
    assert layout(64, 64, 16, -1) == (0, 16, 1) and layout(64, 64, 16, 4) == (0, 4, 4) and layout(64, 64, 16, 64) == (0, 16, 1)
    assert layout(64, 64, 10, 4)[0] == -1                                        # spp % spp_per_pass != 0
    # 2^33 lanes: ceil(2^33 / (2^32 - 1)) = 3 passes, 512 / 3 = 170 does not divide 512 -> the reference throws (sampler.cpp:75-83)
    assert layout(4096, 4096, 512, -1)[0] == -1
    assert layout(4096, 4096, 768, -1) == (0, 192, 4) and layout(4096, 4096, 255, -1) == (0, 255, 1)   # ceil(3.0000000007) = 4
    sc = orc.Scene(os.path.join(SCENES, "cornell_wall.xml"), dict(resx=6, resy=5))
    integ = dict(type="dopplertofpath", max_depth=3, path_correlation_depth=3, time_sampling_method="stratified")
    one, _ = sc.render(sc.params(integrator=integ), seed=2, spp=16)
    same, _ = sc.render(sc.params(integrator=dict(integ, samples_per_pass=16)), seed=2, spp=16)
    assert np.array_equal(one, same)
    pd = sc.params(integrator=dict(integ, samples_per_pass=4))
    lanes = sc.render_lanes(pd, 2, 16, 0, 6 * 5 * 16)                           # 4 passes x 120 lanes
    t = lanes["time"].reshape(4, 30, 4).transpose(1, 0, 2).reshape(30, 16) / 0.0015
    assert np.array_equal(np.sort(np.floor(t * 16).astype(int), axis=1), np.tile(np.arange(16), (30, 1)))
