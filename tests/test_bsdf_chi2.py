"""The reference's chi^2 tests of its BSDFs, restated on the oracle (CPU): src/bsdfs/tests/test_diffuse.py (test02_chi2), test_roughconductor.py
(test0x_chi2_* : isotropic / anisotropic, Beckmann / GGX), test_roughplastic.py, test_roughdielectric.py (reflection + transmission: the whole sphere),
test_twosided.py and test_blendbsdf.py (their chi2 cases).  The reference's harness (python/mitsuba/python/chi2.py: ChiSquareTest over a SphericalDomain
with a BSDFAdapter) histograms the directions BSDF::sample returns over (cos theta, phi) cells and compares the counts with BSDF::pdf integrated over each
cell; cells with few expected samples are pooled and the p-value of the chi^2 statistic must exceed the significance level.  Here the same is done with
numpy / scipy on `orc_kat_bsdf` (eval / pdf / sample of one shape's BSDF chain: mask -> blend -> frame -> nested BSDF).  The GPU kernels are held to the
oracle bit for bit (test_gpu_parity.py), so this pins both."""
import ctypes as C

import numpy as np
import pytest
from scipy import stats

SCENE = '<scene version="3.0.0"><shape type="rectangle">%s</shape></scene>'
DIFFUSE = '<bsdf type="diffuse"><rgb name="reflectance" value="0.2, 0.5, 0.8"/></bsdf>'
CASES = {
    "diffuse": DIFFUSE,
    "roughconductor_beckmann": '<bsdf type="roughconductor"><float name="alpha" value="0.3"/><string name="distribution" value="beckmann"/></bsdf>',
    "roughconductor_ggx": '<bsdf type="roughconductor"><float name="alpha" value="0.3"/><string name="distribution" value="ggx"/></bsdf>',
    "roughconductor_aniso_ggx": '<bsdf type="roughconductor"><float name="alpha_u" value="0.2"/><float name="alpha_v" value="0.5"/><string name="distribution" value="ggx"/></bsdf>',
    "roughconductor_all_normals": '<bsdf type="roughconductor"><float name="alpha" value="0.4"/><boolean name="sample_visible" value="false"/></bsdf>',
    "roughplastic": '<bsdf type="roughplastic"><float name="alpha" value="0.25"/><rgb name="diffuse_reflectance" value="0.5"/></bsdf>',
    "roughdielectric_beckmann": '<bsdf type="roughdielectric"><float name="alpha" value="0.3"/><string name="distribution" value="beckmann"/></bsdf>',
    "roughdielectric_ggx": '<bsdf type="roughdielectric"><float name="alpha" value="0.4"/><string name="distribution" value="ggx"/></bsdf>',
    "twosided_from_behind": '<bsdf type="twosided">%s</bsdf>' % DIFFUSE,
    "blendbsdf": '<bsdf type="blendbsdf"><float name="weight" value="0.4"/>%s<bsdf type="roughconductor"><float name="alpha" value="0.3"/></bsdf></bsdf>' % DIFFUSE,
}
RES_C, RES_P, SUB = 20, 40, 16         # cells over cos theta in [-1, 1] and phi in [0, 2 pi); pdf quadrature points per cell and axis
N = 60000


def _bsdf(L, shape, wi, wo, s3):
    out = np.zeros(13, np.float32)
    L.orc_kat_bsdf(C.byref(shape), wi.ctypes.data, wo.ctypes.data, s3.ctypes.data, out.ctypes.data)
    return out


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("theta_i", [20.0, 65.0])
def test_sampled_directions_follow_the_pdf(orc, name, theta_i):
    L = orc.lib()
    osc = orc.Scene(SCENE % CASES[name], {}, is_string=True)
    shape = osc.c.shapes[0]
    t = np.radians(theta_i)
    wi = np.float32([np.sin(t) * np.cos(0.7), np.sin(t) * np.sin(0.7), np.cos(t)])
    if name == "twosided_from_behind":
        wi[2] = -wi[2]
    rng = np.random.default_rng(int(theta_i) + len(name))
    # histogram of BSDF::sample (samples of zero density -- failed draws -- land nowhere, as in the reference's adapter, whose weights are then zero)
    hist = np.zeros((RES_C, RES_P))
    dummy = np.float32([0, 0, 1])
    for s3 in rng.random((N, 3)).astype(np.float32):
        out = _bsdf(L, shape, wi, dummy, s3)
        if out[7] <= 0 or out[9] != 0 or not out[10:13].any():      # chi2.py's BSDFAdapter: a sample whose weight is zero counts for nothing (a direction below the horizon)
            continue
        wo = out[4:7].astype(np.float64)
        ci = min(int((wo[2] * 0.5 + 0.5) * RES_C), RES_C - 1)
        pi_ = min(int((np.arctan2(wo[1], wo[0]) % (2 * np.pi)) / (2 * np.pi) * RES_P), RES_P - 1)
        hist[ci, pi_] += 1
    # BSDF::pdf integrated over every cell (midpoint rule, SUB x SUB points; d omega = d cos theta d phi)
    expected = np.zeros((RES_C, RES_P))
    zero3 = np.float32([0.5, 0.5, 0.5])
    for ci in range(RES_C):
        for pi_ in range(RES_P):
            acc = 0.0
            for a in range(SUB):
                c = -1 + 2 * (ci + (a + 0.5) / SUB) / RES_C
                s_ = np.sqrt(max(0.0, 1 - c * c))
                for b in range(SUB):
                    p = 2 * np.pi * (pi_ + (b + 0.5) / SUB) / RES_P
                    acc += _bsdf(L, shape, wi, np.float32([s_ * np.cos(p), s_ * np.sin(p), c]), zero3)[3]
            expected[ci, pi_] = acc / (SUB * SUB) * (2.0 / RES_C) * (2 * np.pi / RES_P) * N
    assert 0.3 * N < expected.sum() < 1.02 * N, (name, expected.sum() / N)        # a density (failed draws may take a share)
    assert abs(hist.sum() - expected.sum()) < 0.02 * N, (name, hist.sum(), expected.sum())
    # chi^2 with pooling of the cells that expect fewer than five samples (chi2.py: `chi2` of mitsuba.math, pooling threshold 5)
    order = np.argsort(expected, axis=None)
    e, o = expected.flatten()[order], hist.flatten()[order]
    pooled_e, pooled_o, stat, dof = 0.0, 0.0, 0.0, 0
    for ev, ov in zip(e, o):
        if ev < 5:                     # (cells the quadrature sees as empty included: a sample on the edge of the support)
            pooled_e += ev; pooled_o += ov
            continue
        stat += (ov - ev) ** 2 / ev; dof += 1
    if pooled_e > 0:
        stat += (pooled_o - pooled_e) ** 2 / pooled_e; dof += 1
    p_value = stats.chi2.sf(stat, dof - 1)
    # the reference's significance level is 0.01 with a Sidak correction over the tests of a run; 20 cases here
    assert p_value > 1 - (1 - 0.01) ** (1 / 20.0), (name, theta_i, stat, dof, p_value)


# ------------------------------------------------------------------------------------------------ src/core/tests/test_warp.py: the chi^2 tests of the warps on the path
def _chi2(hist, expected, n_tests):
    order = np.argsort(expected, axis=None)
    e, o = expected.flatten()[order], hist.flatten()[order]
    pooled_e = pooled_o = stat = 0.0
    dof = 0
    for ev, ov in zip(e, o):
        if ev < 5:
            pooled_e += ev; pooled_o += ov
            continue
        stat += (ov - ev) ** 2 / ev; dof += 1
    if pooled_e > 0:
        stat += (pooled_o - pooled_e) ** 2 / pooled_e; dof += 1
    return stats.chi2.sf(stat, dof - 1) > 1 - (1 - 0.01) ** (1.0 / n_tests), stat, dof


WARPS = {"cosine_hemisphere": (0, "warp_cosine_hemisphere"), "uniform_disk_concentric": (1, "warp_disk_concentric"), "uniform_sphere": (2, "warp_uniform_sphere"),
         "uniform_triangle": (3, "warp_uniform_triangle")}


def _warp_chi2(warp, warped, n):
    """histogram of the warped points against the closed-form densities of warp.h: cos theta / pi, 1 / pi inside the unit disk, 1 / (4 pi), 2 inside the triangle u + v <= 1"""
    fn = WARPS[warp][0]
    res, sub = 24, 8
    w = np.asarray(warped, np.float64)
    if fn in (0, 2):                                       # spherical domain: (cos theta, phi)
        a = np.minimum(((w[:, 2] * 0.5 + 0.5) * res).astype(int), res - 1)
        b = np.minimum(((np.arctan2(w[:, 1], w[:, 0]) % (2 * np.pi)) / (2 * np.pi) * 2 * res).astype(int), 2 * res - 1)
        shape = (res, 2 * res)
    elif fn == 1:                                          # planar domain [-1, 1]^2
        a, b, shape = np.minimum(((w[:, 1] * 0.5 + 0.5) * res).astype(int), res - 1), np.minimum(((w[:, 0] * 0.5 + 0.5) * res).astype(int), res - 1), (res, res)
    else:                                                  # planar domain [0, 1]^2
        a, b, shape = np.minimum((w[:, 1] * res).astype(int), res - 1), np.minimum((w[:, 0] * res).astype(int), res - 1), (res, res)
    hist = np.zeros(shape)
    np.add.at(hist, (a, b), 1)
    expected = np.zeros(shape)
    t, t32 = (np.arange(sub) + 0.5) / sub, (np.arange(32) + 0.5) / 32
    for a_ in range(shape[0]):
        for b_ in range(shape[1]):
            if fn in (0, 2):
                c = -1 + 2 * (a_ + t) / res
                pdf = np.where(c > 0, c / np.pi, 0.0).mean() if fn == 0 else 1 / (4 * np.pi)
                expected[a_, b_] = pdf * (2.0 / res) * (2 * np.pi / (2 * res)) * n
            elif fn == 1:
                y, x = np.meshgrid(-1 + 2 * (a_ + t32) / res, -1 + 2 * (b_ + t32) / res, indexing="ij")      # the rim cuts cells: a finer rule there
                expected[a_, b_] = ((x * x + y * y <= 1) / np.pi).mean() * (2.0 / res) ** 2 * n
            else:                                          # cells below the diagonal are inside, the diagonal's cells half inside (exact)
                inside = 1.0 if a_ + b_ < res - 1 else (0.5 if a_ + b_ == res - 1 else 0.0)
                expected[a_, b_] = inside * 2.0 * (1.0 / res) ** 2 * n
    assert abs(expected.sum() - n) < 0.01 * n and hist.sum() == n
    ok, stat, dof = _chi2(hist, expected, 4)
    assert ok, (warp, stat, dof)


@pytest.mark.parametrize("warp", sorted(WARPS))
def test_warps_follow_their_densities(orc, warp):
    """test_warp.py (test_square_to_cosine_hemisphere / _uniform_disk_concentric / _uniform_sphere / _uniform_triangle, the ChiSquareTest cases): the warped unit
    square of the ORACLE's warps against the closed-form densities of warp.h"""
    L = orc.lib()
    n = 100000
    rng = np.random.default_rng(WARPS[warp][0] + 3)
    out = np.zeros((n, 3), np.float32)
    for i, u in enumerate(rng.random((n, 2)).astype(np.float32)):
        L.orc_kat_warp(WARPS[warp][0], u.ctypes.data, out[i].ctypes.data)
    _warp_chi2(warp, out, n)


@pytest.mark.gpu
@pytest.mark.parametrize("warp", sorted(WARPS))
def test_device_warps_follow_their_densities(mi, warp):
    """the same chi^2 cases on the device functions the kernels are built from (dtof_eval_component), a million points each"""
    n = 1000000
    u = np.random.default_rng(WARPS[warp][0] + 11).random((n, 2)).astype(np.float32)
    out = mi.eval_component(WARPS[warp][1], u)
    if out.shape[1] == 2:
        out = np.concatenate([out, np.zeros((n, 1), np.float32)], axis=1)
    _warp_chi2(warp, out, n)


# ------------------------------------------------------------------------------------------------ src/render/tests/test_microfacet.py:288-309: the chi^2 tests of MicrofacetDistribution
@pytest.mark.gpu
@pytest.mark.parametrize("visible", [1, 0])
@pytest.mark.parametrize("mf_type,alpha_u,alpha_v", [(0, 0.1, 0.1), (0, 0.6, 0.2), (1, 0.1, 0.1), (1, 0.25, 0.5)], ids=["beckmann", "beckmann_aniso", "ggx", "ggx_aniso"])
@pytest.mark.parametrize("theta_i", [30.0, 80.0])
def test_device_microfacet_normals_follow_their_density(mi, mf_type, alpha_u, alpha_v, visible, theta_i):
    """test04_chi2_* / test05 of test_microfacet.py (MicrofacetAdapter: sample() of normals against pdf(), Beckmann / GGX, isotropic / anisotropic, all normals and
    visible normals, steep and grazing incidence) on the device functions of the kernels: a million normals histogrammed over (theta, phi) of the upper hemisphere,
    pdf integrated per cell (8 x 8 midpoints), pooling below 5"""
    n, res_c, res_p, sub = 1000000, 32, 64, 8
    params = [mf_type, alpha_u, alpha_v, visible]
    t = np.radians(theta_i)
    wi = np.float32([np.sin(t) * np.cos(0.4), np.sin(t) * np.sin(0.4), np.cos(t)])
    u = np.random.default_rng(int(theta_i) + mf_type * 7 + visible).random((n, 2)).astype(np.float32)
    out = mi.eval_component("microfacet_sample", np.concatenate([np.tile(wi, (n, 1)), u], axis=1), params)
    m = out[:, :3].astype(np.float64)
    ok = out[:, 3] > 0
    a = np.minimum((np.arccos(np.clip(m[ok, 2], -1, 1)) / (np.pi / 2) * res_c).astype(int), res_c - 1)      # cells uniform in THETA: a lobe of alpha = 0.1 spans several
    b = np.minimum(((np.arctan2(m[ok, 1], m[ok, 0]) % (2 * np.pi)) / (2 * np.pi) * res_p).astype(int), res_p - 1)
    hist = np.zeros((res_c, res_p))
    np.add.at(hist, (a, b), 1)
    # pdf over every cell: all quadrature points in one call
    tt = (np.arange(sub) + 0.5) / sub
    th = ((np.arange(res_c)[:, None] + tt[None, :]) / res_c * (np.pi / 2)).reshape(-1)         # theta of the normal
    p = (2 * np.pi * (np.arange(res_p)[:, None] + tt[None, :]) / res_p).reshape(-1)
    tth, pp = np.meshgrid(th, p, indexing="ij")
    pts = np.stack([np.sin(tth) * np.cos(pp), np.sin(tth) * np.sin(pp), np.cos(tth)], axis=-1).reshape(-1, 3).astype(np.float32)
    pdf = mi.eval_component("microfacet_pdf", np.concatenate([np.tile(wi, (len(pts), 1)), pts], axis=1), params)[:, 0].astype(np.float64)
    expected = (pdf * np.sin(tth).reshape(-1)).reshape(res_c, sub, res_p, sub).mean(axis=(1, 3)) * (np.pi / 2 / res_c) * (2 * np.pi / res_p) * n      # d omega = sin theta d theta d phi
    assert abs(expected.sum() / n - 1) < 0.02 and abs(hist.sum() - expected.sum()) < 0.02 * n, (expected.sum() / n, hist.sum() / n)
    ok_, stat, dof = _chi2(hist, expected, 16)
    assert ok_, (mf_type, alpha_u, alpha_v, visible, theta_i, stat, dof)
