"""Smooth conductor / dielectric BSDFs (src/bsdfs/{conductor,dielectric,twosided}.cpp, fresnel.h):
closed-form Fresnel known answers, loader behaviour, physical sanity of the oracle, GPU-vs-oracle bit-exact lanes."""
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


def test_fresnel_known_answers(orc):
    L = orc.lib()
    def diel(c, eta):
        out = (C.c_float * 4)()
        L.orc_fresnel_dielectric(c, eta, out)
        return list(out)
    r, ct, eit, eti = diel(1.0, 1.5)                      # normal incidence: ((eta-1)/(eta+1))^2 = 0.04, straight through
    assert abs(r - 0.04) < 1e-6 and abs(ct + 1.0) < 1e-6 and eit == 1.5 and abs(eti - 1 / 1.5) < 1e-7
    r, ct, eit, eti = diel(-1.0, 1.5)                     # from inside: same reflectance, eta swapped, transmitted upwards
    assert abs(r - 0.04) < 1e-6 and abs(ct - 1.0) < 1e-6 and abs(eit - 1 / 1.5) < 1e-7 and eti == 1.5
    assert diel(-0.5, 1.5)[0] == 1.0                      # beyond the critical angle (sin = 0.866 > 1/1.5): total internal reflection
    brewster = float(np.cos(np.arctan(1.5)))
    r = diel(brewster, 1.5)[0]                            # Brewster angle: the p wave vanishes, r = a_s^2 / 2
    a_s = (brewster - 1.5 * np.sqrt(1 - (1 - brewster ** 2) / 2.25)) / (brewster + 1.5 * np.sqrt(1 - (1 - brewster ** 2) / 2.25))
    assert abs(r - 0.5 * a_s ** 2) < 1e-6
    assert diel(0.3, 1.0)[0] == 0.0 and diel(0.0, 1.5)[0] == 1.0          # index matched / grazing special cases
    for c in (1.0, 0.7, 0.2, 0.01):
        assert abs(L.orc_fresnel_conductor(c, 0.0, 1.0) - 1.0) < 1e-6     # eta = 0, k = 1 (the plugin's default): a perfect mirror
    assert abs(L.orc_fresnel_conductor(1.0, 1.5, 0.0) - 0.04) < 1e-6      # k = 0: the dielectric value at normal incidence
    assert 0.5 < L.orc_fresnel_conductor(1.0, 0.2, 3.9) < 1.0             # copper-like: highly reflective


def test_loader_semantics_of_the_specular_bsdfs(mi, orc):
    path = os.path.join(SCENES, "cornell_specular.xml")
    osc = orc.Scene(path, {})
    glass = [s for s in osc.flat.shapes if s["bsdf"] == 2][0]
    mirror = [s for s in osc.flat.shapes if s["bsdf"] == 1][0]
    assert abs(float(glass["diel_eta"]) - 1.5 / 1.000277) < 1e-6 and glass["twosided"] == 0
    assert mirror["twosided"] == 1 and np.allclose(mirror["cond_k"], [3.9, 2.45, 2.14])
    mi.load_file(path)                                                     # the product's loader accepts the same file
    text = open(path).read()
    with pytest.raises(mi.DtofError, match="Only materials without a transmission component can be nested"):
        mi.load_string(text.replace('<bsdf type="conductor">', '<bsdf type="dielectric">').replace(
            '<rgb name="eta" value="0.2, 0.92, 1.1" />', "").replace('<rgb name="k" value="3.9, 2.45, 2.14" />', ""))
    with pytest.raises(mi.DtofError, match="Unable to find an IOR value"):
        mi.load_string(text.replace('value="air"', 'value="unobtainium"'))
    with pytest.raises(mi.DtofError, match="named materials"):
        mi.load_string(text.replace('<rgb name="eta" value="0.2, 0.92, 1.1" />', '<string name="material" value="Cu" />').replace(
            '<rgb name="k" value="3.9, 2.45, 2.14" />', ""))
    with pytest.raises(mi.DtofError, match="unsupported BSDF plugin"):
        mi.load_string(text.replace('type="dielectric"', 'type="hair"'))


def mirror_room(mirror_bsdf):
    """the Cornell room with the area light; the back wall is either diffuse white or the given BSDF"""
    s = ms.HEADER.format(spp=16, res=32, tsm="antithetic", shift="0.5") + ms.SENSOR.format(fov="19.5", cam=ms.CAM)
    for b in ms.BSDFS:
        s += ms.bsdf(*b)
    s += mirror_bsdf
    for name, m, b in ms.WALLS:
        s += ms.rect(name, m, "M" if name == "BackWall" else b)
    return s + ms.AREA_LIGHT + "</scene>\n"


def test_oracle_physical_sanity(orc, tmp_path):
    """(1) a glass pane with index-matched glass (int_ior = ext_ior) is invisible; (2) a black conductor (specular_reflectance 0)
    kills every path that touches it; (3) a perfect mirror wall conserves energy: brighter than the black wall, and the mean over
    the room stays finite and positive; (4) radiance through a real glass pane = (1 - r)^2 + ... <= 1 of the unobstructed one."""
    integ = dict(type="path", max_depth=6)
    def render(xml, name, spp=128, depth=6):
        p = str(tmp_path / name)
        open(p, "w").write(xml)
        sc = orc.Scene(p, dict(resx=16, resy=16))
        pd = sc.params(integrator=dict(integ, max_depth=depth))
        return np.mean([sc.render(pd, seed=s, spp=spp, threads=NCPU)[0] for s in range(2)], axis=0)
    base = open(os.path.join(SCENES, "cornell_area.xml")).read()
    ref = render(base, "ref.xml")
    pane = ('\t<bsdf type="dielectric" id="G"><float name="int_ior" value="%s" /><float name="ext_ior" value="1.0" /></bsdf>\n'
            '\t<shape type="rectangle" id="Front"><transform name="to_world"><translate x="0" y="1" z="2.5" /></transform><ref id="G" /></shape>\n'
            '\t<shape type="rectangle" id="Back"><transform name="to_world"><rotate y="1" angle="180" /><translate x="0" y="1" z="2.4" /></transform>'
            '<ref id="G" /></shape>\n')   # a slab: entering through Front (normal towards the camera), leaving through Back (normal away)
    # every path of the room now starts with two extra (delta) vertices: depth 8 behind the slab == depth 6 without it
    same = render(base.replace("</scene>", pane % "1.0" + "</scene>"), "matched.xml", depth=8)
    assert abs(same.mean() - ref.mean()) < 0.03 * ref.mean()          # different random numbers (the pane consumes draws), same expectation
    glass = render(base.replace("</scene>", pane % "1.5" + "</scene>"), "glass.xml", depth=8)
    assert 0.80 * ref.mean() < glass.mean() < 1.0 * ref.mean()        # ~8 % reflected away at the two interfaces (and the 1/eta^2 radiance
                                                                      # compression inside the slab is undone on the way out), never brighter
    black = render(mirror_room('\t<bsdf type="twosided" id="M"><bsdf type="conductor"><rgb name="specular_reflectance" value="0" /></bsdf></bsdf>\n'), "black.xml")
    perfect = render(mirror_room('\t<bsdf type="twosided" id="M"><bsdf type="conductor" /></bsdf>\n'), "perfect.xml")
    white = render(mirror_room(ms.bsdf("M", "0.725, 0.71, 0.68")), "white.xml")
    assert black[6:10, 6:10].max() == 0.0                               # the wall itself is black
    assert perfect.mean() > 1.15 * black.mean() and np.isfinite(perfect).all()
    assert 0.6 * white.mean() < perfect.mean() < 1.6 * white.mean()


GPU_CASES = [("cornell_specular", os.path.join(SCENES, "cornell_specular.xml"), dict(resx=48, resy=48), 8, dict(type="path", max_depth=8)),
             ("cornell_specular_doppler", os.path.join(SCENES, "cornell_specular.xml"), dict(resx=32, resy=32, max_depth=6), 8, None),
             ("specular_rr", os.path.join(SCENES, "cornell_specular.xml"), dict(resx=24, resy=24), 8, dict(type="path", max_depth=-1, rr_depth=2)),
             ("mirror_wall_fused", None, dict(resx=32, resy=32), 8, dict(type="path", max_depth=5)),
             ("mirror_wall_doppler", None, dict(resx=32, resy=32, max_depth=5, time_sampling_method="stratified"), 8, None)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,path,params,spp,integ", GPU_CASES, ids=[c[0] for c in GPU_CASES])
def test_specular_scenes_are_bit_exact_per_lane(mi, orc, tmp_path, name, path, params, spp, integ):
    if path is None:     # rectangles only -> the fused pipeline, SPEC instantiation
        path = str(tmp_path / "mirror.xml")
        open(path, "w").write(mirror_room('\t<bsdf type="twosided" id="M"><bsdf type="conductor"><rgb name="eta" value="0.2, 0.92, 1.1" />'
                                          '<rgb name="k" value="3.9, 2.45, 2.14" /></bsdf></bsdf>\n'))
    sc, osc = mi.load_file(path, **params), orc.Scene(path, params)
    if integ:
        sc.set_integrator(integ)
    pd = osc.params(integrator=integ) if integ else osc.params()
    w, h = sc.size
    n = w * h * spp
    g = sc.sample_lanes(9, spp, 0, n)
    o = osc.render_lanes(pd, 9, spp, 0, n, threads=NCPU)
    for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
        assert np.array_equal(bits(g[k]), bits(o[k])), (name, k, int((bits(g[k]) != bits(o[k])).sum()))
    assert (g["rgb"] != 0).mean() > 0.05
    img = sc.render(seed=9, spp=spp)
    ref, _ = osc.render(pd, seed=9, spp=spp, threads=NCPU)
    assert float(np.abs(img - ref).max() / np.abs(ref).max()) <= 1e-5


# ------------------------------------------------------------------------------------------------ smooth plastic
def test_plastic_loader_constants_and_limits(mi, orc, tmp_path):
    """SmoothPlastic (plastic.cpp:167-217): the constructor constants computed by the product's loader are bit-identical to the
    oracle's; int_ior = ext_ior degenerates to a plain diffuse BSDF in expectation."""
    path = os.path.join(SCENES, "cornell_plastic.xml")
    sc, osc = mi.load_file(path), orc.Scene(path, {})
    rec = sc.export(9).reshape(-1, 24)
    pl = [(i, s) for i, s in enumerate(osc.flat.shapes) if s["bsdf"] == 3]
    assert len(pl) == 3
    for i, s in pl:
        assert rec[i, 0] == 3 and rec[i, 1] == 1 and np.array_equal(bits(rec[i, 4:7]), bits(s["plastic_params"]))
        assert np.array_equal(bits(rec[i, 7:10]), bits(s["reflectance"])) and bits(rec[i, 2]) == bits(np.float32(s["diel_eta"]))
    eta = np.float32(1.9) / np.float32(1.000277)
    assert abs(rec[pl[0][0], 4] - 1 / eta ** 2) < 1e-6 and 0.7 < rec[pl[0][0], 5] < 0.8      # fdr_int(1/1.9) ~ 0.76
    assert abs(rec[pl[0][0], 6] - 1.0 / (1.0 + (0.1 + 0.27 + 0.36) / 3)) < 1e-6               # s_mean / (d_mean + s_mean)
    # index-matched plastic == diffuse (fresnel 0, fdr_int ~ 0): same expectation, different sampling code
    base = open(os.path.join(SCENES, "cornell_area.xml")).read()
    plastic = base.replace('<bsdf type="twosided" id="FloorBSDF">\n\t\t<bsdf type="diffuse">\n\t\t\t<rgb name="reflectance" value="0.725, 0.71, 0.68" />',
                           '<bsdf type="twosided" id="FloorBSDF">\n\t\t<bsdf type="plastic">\n\t\t\t<rgb name="diffuse_reflectance" value="0.725, 0.71, 0.68" />'
                           '\n\t\t\t<float name="int_ior" value="1.0" />\n\t\t\t<float name="ext_ior" value="1.0" />')
    assert plastic != base
    p2 = str(tmp_path / "matched.xml")
    open(p2, "w").write(plastic)
    P, integ = dict(resx=16, resy=16), dict(type="path", max_depth=4)
    a, b = orc.Scene(os.path.join(SCENES, "cornell_area.xml"), P), orc.Scene(p2, P)
    ia = np.mean([a.render(a.params(integrator=integ), seed=s, spp=256, threads=NCPU)[0] for s in range(2)], axis=0)
    ib = np.mean([b.render(b.params(integrator=integ), seed=s, spp=256, threads=NCPU)[0] for s in range(2)], axis=0)
    assert abs(ia.mean() - ib.mean()) < 0.02 * ia.mean()


PLASTIC_CASES = [("plastic_doppler", dict(resx=48, resy=48), 8, None), ("plastic_path_depth6", dict(resx=32, resy=32), 8, dict(type="path", max_depth=6)),
                 ("plastic_nonlinear", dict(resx=24, resy=24), 8, dict(type="path", max_depth=4))]


@pytest.mark.gpu
@pytest.mark.parametrize("name,params,spp,integ", PLASTIC_CASES, ids=[c[0] for c in PLASTIC_CASES])
def test_plastic_scenes_are_bit_exact_per_lane(mi, orc, tmp_path, name, params, spp, integ):
    path = os.path.join(SCENES, "cornell_plastic.xml")
    if "nonlinear" in name:
        text = open(path).read().replace('<float name="int_ior" value="1.9" />', '<string name="int_ior" value="diamond" />\n\t\t\t<boolean name="nonlinear" value="true" />'
                                                                                 '\n\t\t\t<rgb name="specular_reflectance" value="0.9, 0.8, 0.7" />')
        path = str(tmp_path / "nl.xml")
        open(path, "w").write(text)
    sc, osc = mi.load_file(path, **params), orc.Scene(path, params)
    if integ:
        sc.set_integrator(integ)
    pd = osc.params(integrator=integ) if integ else osc.params()
    w, h = sc.size
    n = w * h * spp
    g = sc.sample_lanes(11, spp, 0, n)
    o = osc.render_lanes(pd, 11, spp, 0, n, threads=NCPU)
    for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
        assert np.array_equal(bits(g[k]), bits(o[k])), (name, k, int((bits(g[k]) != bits(o[k])).sum()))
    assert (g["rgb"] != 0).mean() > 0.3
    img = sc.render(seed=11, spp=spp)
    ref, _ = osc.render(pd, seed=11, spp=spp, threads=NCPU)
    assert float(np.abs(img - ref).max() / np.abs(ref).max()) <= 1e-5


# ------------------------------------------------------------------------------------------------ rough conductor (GGX)
def test_roughconductor_limits_and_loader(mi, orc, tmp_path):
    """RoughConductor with the GGX distribution (roughconductor.cpp, microfacet.h): (1) the loader record matches the oracle's;
    (2) alpha -> 1e-4 converges to the smooth conductor (glossy lobe with NEE + MIS vs a delta lobe: different estimators, same
    expectation); (3) a visibly rough wall stays finite and of the same order; (4) both distributions and both sampling modes (sample_visible) load."""
    path = os.path.join(SCENES, "cornell_rough.xml")
    sc, osc = mi.load_file(path), orc.Scene(path, {})
    rec = sc.export(9).reshape(-1, 24)
    rough = [(i, s) for i, s in enumerate(osc.flat.shapes) if s["bsdf"] == 4]
    assert len(rough) == 3
    for i, s in rough:
        assert rec[i, 0] == 4 and rec[i, 1] == 1 and bits(rec[i, 22]) == bits(np.float32(s["alpha_u"])) and bits(rec[i, 23]) == bits(np.float32(s["alpha_v"]))
        assert np.array_equal(bits(rec[i, 16:19]), bits(s["cond_eta"])) and np.array_equal(bits(rec[i, 19:22]), bits(s["cond_k"]))
    assert any(rec[i, 22] != rec[i, 23] for i, _ in rough)             # the brushed floor is anisotropic
    beck = mi.load_string(open(path).read().replace('value="ggx"', 'value="beckmann"'))     # the plugins' default distribution
    assert [beck.export(12)[i] for i, _ in rough] == [0.0] * 3 and [sc.export(12)[i] for i, _ in rough] == [1.0] * 3
    allnorm = mi.load_file(path, sample_visible="false")
    assert [allnorm.export(17)[i] for i, _ in rough] == [1.0] * len(rough) and not sc.export(17).any()
    with pytest.raises(mi.DtofError, match="invalid distribution"):
        mi.load_string(open(path).read().replace('value="ggx"', 'value="phong"'))
    with pytest.raises(mi.DtofError, match="both 'alpha_u' and 'alpha_v'"):
        mi.load_string(open(path).read().replace('<float name="alpha_v" value="0.3" />', ""))

    def render(bsdf_xml, name):
        p = str(tmp_path / name)
        open(p, "w").write(mirror_room(bsdf_xml))
        s = orc.Scene(p, dict(resx=16, resy=16))
        pd = s.params(integrator=dict(type="path", max_depth=5))
        return np.mean([s.render(pd, seed=k, spp=256, threads=NCPU)[0] for k in range(2)], axis=0)
    copper = '<rgb name="eta" value="0.2, 0.92, 1.1" /><rgb name="k" value="3.9, 2.45, 2.14" />'
    smooth = render('\t<bsdf type="twosided" id="M"><bsdf type="conductor">%s</bsdf></bsdf>\n' % copper, "smooth.xml")
    sharp = render('\t<bsdf type="twosided" id="M"><bsdf type="roughconductor"><string name="distribution" value="ggx" />'
                   '<float name="alpha" value="0.00001" />%s</bsdf></bsdf>\n' % copper, "sharp.xml")
    blurry = render('\t<bsdf type="twosided" id="M"><bsdf type="roughconductor"><string name="distribution" value="ggx" />'
                    '<float name="alpha" value="0.3" />%s</bsdf></bsdf>\n' % copper, "blurry.xml")
    assert abs(sharp.mean() - smooth.mean()) < 0.03 * smooth.mean(), (sharp.mean(), smooth.mean())
    assert np.isfinite(blurry).all() and 0.5 * smooth.mean() < blurry.mean() < 2.0 * smooth.mean()   # a rough wall also scatters the lamp towards the camera


ROUGH_CASES = [("rough_doppler", dict(resx=40, resy=40), 8, None), ("rough_path_depth6", dict(resx=32, resy=32), 8, dict(type="path", max_depth=6)),
               ("rough_rr", dict(resx=24, resy=24), 8, dict(type="path", max_depth=-1, rr_depth=2))]


@pytest.mark.gpu
@pytest.mark.parametrize("name,params,spp,integ", ROUGH_CASES, ids=[c[0] for c in ROUGH_CASES])
def test_rough_conductor_scenes_are_bit_exact_per_lane(mi, orc, name, params, spp, integ):
    path = os.path.join(SCENES, "cornell_rough.xml")
    sc, osc = mi.load_file(path, **params), orc.Scene(path, params)
    if integ:
        sc.set_integrator(integ)
    pd = osc.params(integrator=integ) if integ else osc.params()
    w, h = sc.size
    n = w * h * spp
    g = sc.sample_lanes(13, spp, 0, n)
    o = osc.render_lanes(pd, 13, spp, 0, n, threads=NCPU)
    for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
        assert np.array_equal(bits(g[k]), bits(o[k])), (name, k, int((bits(g[k]) != bits(o[k])).sum()))
    assert (g["rgb"] != 0).mean() > 0.3
    img = sc.render(seed=13, spp=spp)
    ref, _ = osc.render(pd, seed=13, spp=spp, threads=NCPU)
    assert float(np.abs(img - ref).max() / np.abs(ref).max()) <= 1e-5


# ------------------------------------------------------------------------------------------------ rough plastic (GGX)
def test_roughplastic_tables_loader_and_limits(mi, orc, tmp_path):
    """RoughPlastic (roughplastic.cpp:170-421): (1) Gauss-Legendre nodes (core/quad.h:27-86) against numpy's; (2) the
    transmittance table and internal reflectance the product's loader computes are bit-identical to the oracle's, physically
    ordered (more light enters at normal incidence, all values in (0, 1)); (3) alpha -> 0 approaches the smooth-plastic constants:
    T(mu) -> 1 - F(mu), internal reflectance -> fresnel_diffuse_reflectance(1 / eta); (4) loader errors as in the reference."""
    for n in (1, 2, 5, 32, 128):
        nodes, weights = np.zeros(n, np.float32), np.zeros(n, np.float32)
        orc.lib().orc_gauss_legendre(n, nodes.ctypes.data, weights.ctypes.data)
        ref = np.polynomial.legendre.leggauss(n)
        assert np.abs(ref[0] - nodes).max() < 1e-6 and np.abs(ref[1] - weights).max() < 1e-6 and abs(weights.sum() - 2) < 1e-5
    path = os.path.join(SCENES, "cornell_roughplastic.xml")
    sc, osc = mi.load_file(path), orc.Scene(path, {})
    rec, tables = sc.export(9).reshape(-1, 24), sc.export(10).reshape(-1, 64)
    rp = [(i, s) for i, s in enumerate(osc.flat.shapes) if s["bsdf"] == 5]
    assert len(rp) == 3 and tables.shape[0] == 3
    for k, (i, s) in enumerate(rp):
        assert rec[i, 0] == 5 and rec[i, 1] == 1 and np.array_equal(bits(rec[i, 4:7]), bits(s["plastic_params"]))
        assert bits(rec[i, 22]) == bits(np.float32(s["alpha_u"])) and np.array_equal(bits(tables[k]), bits(s["rough_table"]))
        t = s["rough_table"]
        assert 0 < t.min() and t.max() < 1 and t[-1] > t[0] and 0 < s["plastic_params"][1] < 1
    assert rec[rp[0][0], 3] == 1 and rec[rp[1][0], 3] == 0                      # the floor is nonlinear, the boxes are not
    assert abs(rec[rp[0][0], 6] - 0.85 / (0.55 + 0.85)) < 1e-6                 # s_mean / (d_mean + s_mean) with a specular_reflectance
    assert abs(rec[rp[1][0], 6] - 1.0 / (1.0 + (0.1 + 0.27 + 0.36) / 3)) < 1e-6   # s_mean = 1 without one
    eta = np.float32(1.49)
    table, ir = orc.rough_plastic_tables(1e-4, eta)
    smooth = np.zeros(3, np.float32)
    orc.lib().orc_plastic_params(C.c_float(eta), (C.c_float * 3)(.5, .5, .5), (C.c_float * 3)(1, 1, 1), smooth.ctypes.data)
    assert abs(ir - smooth[1]) < 0.03 * smooth[1], (ir, smooth[1])   # a 64-point mean against a fitted polynomial
    for i in (8, 32, 63):
        r = np.zeros(4, np.float32)
        orc.lib().orc_fresnel_dielectric(C.c_float(i / 63.0), C.c_float(eta), r.ctypes.data)
        assert abs(table[i] - (1 - r[0])) < 2e-3, (i, table[i], 1 - r[0])
    text = open(path).read()
    assert 0.0 in mi.load_string(text.replace('value="ggx"', 'value="beckmann"')).export(12)   # Beckmann loads (MicrofacetType 0)
    with pytest.raises(mi.DtofError, match="does not support anisotropic"):
        mi.load_string(text.replace('<float name="alpha" value="0.15" />', '<float name="alpha_u" value="0.15" /><float name="alpha_v" value="0.3" />'))
    with pytest.raises(mi.DtofError, match="must be positive and differ"):
        mi.load_string(text.replace('<float name="int_ior" value="1.9" />', '<float name="int_ior" value="1.5" /><float name="ext_ior" value="1.5" />'))


def test_roughplastic_energy_and_smooth_limit(orc, tmp_path):
    """A rough-plastic back wall (1) never reflects more than it receives: the room is darker than with a white diffuse wall of
    reflectance 1 and stays finite; (2) with alpha -> 1e-4 it matches the smooth `plastic` wall in expectation (glossy lobe with
    NEE + MIS vs a delta lobe: different estimators)."""
    def render(bsdf_xml, name):
        p = str(tmp_path / name)
        open(p, "w").write(mirror_room(bsdf_xml))
        s = orc.Scene(p, dict(resx=16, resy=16))
        pd = s.params(integrator=dict(type="path", max_depth=5))
        return np.mean([s.render(pd, seed=k, spp=256, threads=NCPU)[0] for k in range(2)], axis=0)
    body = '<rgb name="diffuse_reflectance" value="0.5, 0.4, 0.3" /><float name="int_ior" value="1.6" />'
    smooth = render('\t<bsdf type="twosided" id="M"><bsdf type="plastic">%s</bsdf></bsdf>\n' % body, "smooth.xml")
    sharp = render('\t<bsdf type="twosided" id="M"><bsdf type="roughplastic"><string name="distribution" value="ggx" />'
                   '<float name="alpha" value="0.00001" />%s</bsdf></bsdf>\n' % body, "sharp.xml")
    rough = render('\t<bsdf type="twosided" id="M"><bsdf type="roughplastic"><string name="distribution" value="ggx" />'
                   '<float name="alpha" value="0.4" />%s</bsdf></bsdf>\n' % body, "rough.xml")
    white = render('\t<bsdf type="twosided" id="M"><bsdf type="diffuse"><rgb name="reflectance" value="1, 1, 1" /></bsdf></bsdf>\n', "white.xml")
    assert abs(sharp.mean() - smooth.mean()) < 0.03 * smooth.mean(), (sharp.mean(), smooth.mean())
    assert np.isfinite(rough).all() and rough.min() >= 0 and rough.mean() < white.mean()
    assert 0.7 * smooth.mean() < rough.mean() < 1.3 * smooth.mean()


ROUGHPLASTIC_CASES = [("roughplastic_doppler", dict(resx=40, resy=40), 8, None),
                      ("roughplastic_path_depth6", dict(resx=32, resy=32), 8, dict(type="path", max_depth=6)),
                      ("roughplastic_rr", dict(resx=24, resy=24), 8, dict(type="path", max_depth=-1, rr_depth=2))]


@pytest.mark.gpu
@pytest.mark.parametrize("name,params,spp,integ", ROUGHPLASTIC_CASES, ids=[c[0] for c in ROUGHPLASTIC_CASES])
def test_rough_plastic_scenes_are_bit_exact_per_lane(mi, orc, name, params, spp, integ):
    path = os.path.join(SCENES, "cornell_roughplastic.xml")
    sc, osc = mi.load_file(path, **params), orc.Scene(path, params)
    if integ:
        sc.set_integrator(integ)
    pd = osc.params(integrator=integ) if integ else osc.params()
    w, h = sc.size
    n = w * h * spp
    g = sc.sample_lanes(17, spp, 0, n)
    o = osc.render_lanes(pd, 17, spp, 0, n, threads=NCPU)
    for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
        assert np.array_equal(bits(g[k]), bits(o[k])), (name, k, int((bits(g[k]) != bits(o[k])).sum()))
    assert (g["rgb"] != 0).mean() > 0.3
    img = sc.render(seed=17, spp=spp)
    ref, _ = osc.render(pd, seed=17, spp=spp, threads=NCPU)
    assert float(np.abs(img - ref).max() / np.abs(ref).max()) <= 1e-5


# ------------------------------------------------------------------------------------------------ thin dielectric
THIN_PANE = ('\t<bsdf type="thindielectric" id="T"><float name="int_ior" value="%s" /><float name="ext_ior" value="1.0" />%s</bsdf>\n'
             '\t<shape type="rectangle" id="Pane"><transform name="to_world"><translate x="0" y="1" z="2.5" /></transform><ref id="T" /></shape>\n')


def test_thindielectric_window(mi, orc, tmp_path):
    """ThinDielectric (thindielectric.cpp:137-226): one rectangle stands for a pane with both interfaces and all internal bounces,
    reflectance R' = 2r / (1 + r), transmission straight through.  (1) index-matched: invisible (one extra delta vertex per crossing);
    (2) eta = 1.5: darker than the open room by roughly the 7.7 % reflected away at normal incidence, never brighter; (3) a pane with
    specular_transmittance 0 in front of the camera blacks the image out, except for what it mirrors (nothing lit is in front of it);
    (4) twosided{thindielectric} is refused like twosided{dielectric}."""
    def render(xml, name, depth):
        p = str(tmp_path / name)
        open(p, "w").write(xml)
        sc = orc.Scene(p, dict(resx=16, resy=16))
        pd = sc.params(integrator=dict(type="path", max_depth=depth))
        return np.mean([sc.render(pd, seed=s, spp=128, threads=NCPU)[0] for s in range(2)], axis=0)
    base = open(os.path.join(SCENES, "cornell_area.xml")).read()
    ref = render(base, "ref.xml", 6)
    same = render(base.replace("</scene>", THIN_PANE % ("1.0", "") + "</scene>"), "matched.xml", 7)
    assert abs(same.mean() - ref.mean()) < 0.03 * ref.mean()
    glass = render(base.replace("</scene>", THIN_PANE % ("1.5", "") + "</scene>"), "glass.xml", 7)
    assert 0.85 * ref.mean() < glass.mean() < 1.0 * ref.mean()
    opaque = render(base.replace("</scene>", THIN_PANE % ("1.5", '<rgb name="specular_transmittance" value="0" />') + "</scene>"), "opaque.xml", 7)
    assert opaque.max() == 0.0
    text = base.replace("</scene>", THIN_PANE % ("1.5", "") + "</scene>")
    sc = mi.load_string(text)
    rec = sc.export(9).reshape(-1, 24)
    assert rec[-1, 0] == 6 and rec[-1, 1] == 0 and abs(rec[-1, 2] - 1.5) < 1e-6
    # an area emitter ON the pane loads on both loaders (round 4: the integrators' valid_ray flag is modelled; the lanes of such a scene are held against the
    # oracle in tests/test_mask.py::test_null_bsdf_and_emitters_on_null_shapes)
    lit = text.replace('<ref id="T" /></shape>', '<ref id="T" /><emitter type="area"><rgb name="radiance" value="1" /></emitter></shape>')
    assert mi.load_string(lit).info()["n_emitters"] == orc.Scene(lit, {}, is_string=True).c.n_emitters
    with pytest.raises(mi.DtofError, match="Only materials without a transmission component can be nested"):
        mi.load_string(text.replace('<bsdf type="thindielectric" id="T">', '<bsdf type="twosided" id="T"><bsdf type="thindielectric">').replace(
            '<float name="ext_ior" value="1.0" /></bsdf>', '<float name="ext_ior" value="1.0" /></bsdf></bsdf>'))


THIN_CASES = [("thin_pane_path", "cornell_area.xml", dict(resx=32, resy=32), 8, dict(type="path", max_depth=7)),       # rectangles only: fused pipeline
              ("thin_pane_doppler", "cornell_area.xml", dict(resx=24, resy=24, max_depth=6), 8, None),
              ("thin_pane_with_meshes_rr", "cornell_specular.xml", dict(resx=24, resy=24), 8, dict(type="path", max_depth=-1, rr_depth=3))]


@pytest.mark.gpu
@pytest.mark.parametrize("name,scene,params,spp,integ", THIN_CASES, ids=[c[0] for c in THIN_CASES])
def test_thindielectric_scenes_are_bit_exact_per_lane(mi, orc, tmp_path, name, scene, params, spp, integ):
    pane = THIN_PANE % ("1.5", '<rgb name="specular_reflectance" value="0.9, 0.95, 1.0" /><rgb name="specular_transmittance" value="0.95, 0.9, 0.85" />')
    path = str(tmp_path / "thin.xml")
    open(path, "w").write(open(os.path.join(SCENES, scene)).read().replace("</scene>", pane + "</scene>"))
    sc, osc = mi.load_file(path, **params), orc.Scene(path, params)
    if integ:
        sc.set_integrator(integ)
    pd = osc.params(integrator=integ) if integ else osc.params()
    w, h = sc.size
    n = w * h * spp
    g = sc.sample_lanes(19, spp, 0, n)
    o = osc.render_lanes(pd, 19, spp, 0, n, threads=NCPU)
    for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
        assert np.array_equal(bits(g[k]), bits(o[k])), (name, k, int((bits(g[k]) != bits(o[k])).sum()))
    assert (g["rgb"] != 0).mean() > 0.05
    img = sc.render(seed=19, spp=spp)
    ref, _ = osc.render(pd, seed=19, spp=spp, threads=NCPU)
    assert float(np.abs(img - ref).max() / np.abs(ref).max()) <= 5e-5


# ------------------------------------------------------------------------------------------------ rough dielectric (GGX)
FROSTED_PANE = ('\t<bsdf type="roughdielectric" id="G"><string name="distribution" value="ggx" /><float name="alpha" value="%s" />'
                '<float name="int_ior" value="%s" /><float name="ext_ior" value="1.0" /></bsdf>\n'
                '\t<shape type="rectangle" id="Front"><transform name="to_world"><translate x="0" y="1" z="2.5" /></transform><ref id="G" /></shape>\n'
                '\t<shape type="rectangle" id="Back"><transform name="to_world"><rotate y="1" angle="180" /><translate x="0" y="1" z="2.4" /></transform>'
                '<ref id="G" /></shape>\n')


def test_roughdielectric_limits_and_loader(mi, orc, tmp_path):
    """RoughDielectric with GGX (roughdielectric.cpp): (1) alpha -> 1e-4 approaches the smooth dielectric slab in expectation (glossy
    lobes with NEE + MIS against delta lobes); (2) a frosted slab neither creates energy nor blacks out: between 0.6 and 1.0 of the open room;
    (3) loader record and the reference's error messages; twosided{roughdielectric} is refused."""
    def render(xml, name, depth):
        p = str(tmp_path / name)
        open(p, "w").write(xml)
        sc = orc.Scene(p, dict(resx=16, resy=16))
        pd = sc.params(integrator=dict(type="path", max_depth=depth))
        return np.mean([sc.render(pd, seed=s, spp=128, threads=NCPU)[0] for s in range(2)], axis=0)
    base = open(os.path.join(SCENES, "cornell_area.xml")).read()
    ref = render(base, "ref.xml", 6)
    smooth = ('\t<bsdf type="dielectric" id="G"><float name="int_ior" value="1.5" /><float name="ext_ior" value="1.0" /></bsdf>\n' + FROSTED_PANE[FROSTED_PANE.index("\t<shape"):])
    glass = render(base.replace("</scene>", smooth + "</scene>"), "smooth.xml", 8)
    sharp = render(base.replace("</scene>", FROSTED_PANE % ("0.00001", "1.5") + "</scene>"), "sharp.xml", 8)
    assert abs(sharp.mean() - glass.mean()) < 0.04 * glass.mean(), (sharp.mean(), glass.mean())
    frosted = render(base.replace("</scene>", FROSTED_PANE % ("0.3", "1.5") + "</scene>"), "frosted.xml", 8)
    assert np.isfinite(frosted).all() and frosted.min() >= 0 and 0.6 * ref.mean() < frosted.mean() < 1.0 * ref.mean()
    path = os.path.join(SCENES, "cornell_frosted.xml")
    sc, osc = mi.load_file(path), orc.Scene(path, {})
    rec = sc.export(9).reshape(-1, 24)
    rd = [(i, s) for i, s in enumerate(osc.flat.shapes) if s["bsdf"] == 7]
    assert len(rd) == 2
    for i, s in rd:
        assert rec[i, 0] == 7 and rec[i, 1] == 0 and bits(rec[i, 2]) == bits(np.float32(s["diel_eta"]))
        assert bits(rec[i, 22]) == bits(np.float32(s["alpha_u"])) and bits(rec[i, 23]) == bits(np.float32(s["alpha_v"]))
        assert np.array_equal(bits(rec[i, 10:13]), bits(s["spec_refl"])) and np.array_equal(bits(rec[i, 13:16]), bits(s["spec_trans"]))
    assert any(rec[i, 22] != rec[i, 23] for i, _ in rd) and any(abs(rec[i, 2] - 2.419 / 1.000277) < 1e-5 for i, _ in rd)
    text = open(path).read()
    assert 0.0 in mi.load_string(text.replace('value="ggx"', 'value="beckmann"')).export(12)   # Beckmann loads (MicrofacetType 0)
    with pytest.raises(mi.DtofError, match="must be positive and differ"):
        mi.load_string(text.replace('<float name="int_ior" value="1.5" />', '<float name="int_ior" value="1.000277" />'))
    with pytest.raises(mi.DtofError, match="Only materials without a transmission component can be nested"):
        mi.load_string(text.replace('<bsdf type="roughdielectric" id="FrostedBSDF">', '<bsdf type="twosided" id="FrostedBSDF"><bsdf type="roughdielectric">').replace(
            '<string name="ext_ior" value="air" />\n\t</bsdf>', '<string name="ext_ior" value="air" />\n\t</bsdf></bsdf>', 1))


FROSTED_CASES = [("frosted_doppler", None, dict(resx=40, resy=40, max_depth=6), 8, None),
                 ("frosted_path_depth8", None, dict(resx=32, resy=32), 8, dict(type="path", max_depth=8)),
                 ("frosted_rr", None, dict(resx=24, resy=24), 8, dict(type="path", max_depth=-1, rr_depth=2)),
                 ("frosted_pane_fused", "pane", dict(resx=32, resy=32), 8, dict(type="path", max_depth=7))]


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,params,spp,integ", FROSTED_CASES, ids=[c[0] for c in FROSTED_CASES])
def test_rough_dielectric_scenes_are_bit_exact_per_lane(mi, orc, tmp_path, name, kind, params, spp, integ):
    path = os.path.join(SCENES, "cornell_frosted.xml")
    if kind == "pane":   # rectangles only: the fused pipeline
        path = str(tmp_path / "pane.xml")
        open(path, "w").write(open(os.path.join(SCENES, "cornell_area.xml")).read().replace("</scene>", FROSTED_PANE % ("0.2", "1.5") + "</scene>"))
    sc, osc = mi.load_file(path, **params), orc.Scene(path, params)
    if integ:
        sc.set_integrator(integ)
    pd = osc.params(integrator=integ) if integ else osc.params()
    w, h = sc.size
    n = w * h * spp
    g = sc.sample_lanes(23, spp, 0, n)
    o = osc.render_lanes(pd, 23, spp, 0, n, threads=NCPU)
    for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
        assert np.array_equal(bits(g[k]), bits(o[k])), (name, k, int((bits(g[k]) != bits(o[k])).sum()))
    assert (g["rgb"] != 0).mean() > 0.3
    img = sc.render(seed=23, spp=spp)
    ref, _ = osc.render(pd, seed=23, spp=spp, threads=NCPU)
    assert float(np.abs(img - ref).max() / np.abs(ref).max()) <= 5e-5


def test_microfacet_sampling_matches_its_density(orc):
    """MicrofacetDistribution::sample draws normals with the density ::pdf reports (what the reference checks with its chi^2 tests,
    src/render/tests/test_microfacet.py:288-309): for Beckmann and GGX, visible and plain sampling, the sample means of a few test
    functions agree with their quadrature against pdf().  This is the only pin of the Beckmann visible-normal inversion (erf / erfinv)."""
    L = orc.lib()
    rng = np.random.default_rng(7)
    for mf_type in (0, 1):
        for visible in (1, 0):
            for angle in (15.0, 80.0):
                au, av = np.float32(0.25), np.float32(0.4)
                wi = np.array([np.sin(np.radians(angle)), 0.0, np.cos(np.radians(angle))], np.float32)
                n = 40000
                u = rng.random((n, 2)).astype(np.float32)
                m = np.zeros((n, 4), np.float32)
                for i in range(n):
                    inp = np.array([wi[0], wi[1], wi[2], u[i, 0], u[i, 1]], np.float32)
                    L.orc_kat_microfacet(mf_type, au, av, visible, 3, inp.ctypes.data, m[i].ctypes.data)
                # the density sample() returns is the density pdf() reports
                chk = np.zeros(1, np.float32)
                for i in range(0, n, 997):
                    inp = np.concatenate([wi, m[i, :3]]).astype(np.float32)
                    L.orc_kat_microfacet(mf_type, au, av, visible, 1, inp.ctypes.data, chk.ctypes.data)
                    assert abs(chk[0] - m[i, 3]) <= 2e-4 * max(m[i, 3], 1e-3), (mf_type, visible, angle, chk[0], m[i, 3])
                # quadrature of pdf over the hemisphere of normals
                nt, npf = 300, 240
                ct = (np.arange(nt) + 0.5) / nt
                ph = (np.arange(npf) + 0.5) / npf * 2 * np.pi
                st = np.sqrt(1 - ct * ct)
                acc = np.zeros(4)
                for c, s_ in zip(ct, st):
                    for p in ph:
                        mm = np.array([s_ * np.cos(p), s_ * np.sin(p), c], np.float32)
                        inp = np.concatenate([wi, mm]).astype(np.float32)
                        L.orc_kat_microfacet(mf_type, au, av, visible, 1, inp.ctypes.data, chk.ctypes.data)
                        acc += chk[0] * np.array([1.0, mm[0], mm[1] * mm[1], mm[2]])
                acc *= (1.0 / nt) * (2 * np.pi / npf)
                assert abs(acc[0] - 1) < 2e-2, (mf_type, visible, angle, acc[0])           # a density
                est = np.array([1.0, m[:, 0].mean(), (m[:, 1] ** 2).mean(), m[:, 2].mean()])
                assert np.all(np.abs(est[1:] - acc[1:] / acc[0]) < 6e-3), (mf_type, visible, angle, est, acc)


def test_sampling_all_normals_is_consistent(orc):
    """sample_visible = false (roughconductor.cpp:260-265,405-409; roughplastic.cpp:413-417,467-470; roughdielectric.cpp:266-269,345-349,
    584-589): for every rough BSDF (a) the weight of a sample equals eval / pdf of the sampled direction, (b) the sampled directions follow pdf
    (the mean of 1 / pdf over the samples is the measure of the support: 2 pi for the reflecting ones), and (c) the albedo estimated with the
    two sampling modes agrees -- they are two estimators of the same integral."""
    import ctypes as C
    L = orc.lib()
    rng = np.random.default_rng(11)
    albedo = {}
    for scene, bsdf in (("cornell_rough.xml", 4), ("cornell_roughplastic.xml", 5), ("cornell_frosted.xml", 7)):
        for mode in ("true", "false"):
            osc = orc.Scene(os.path.join(SCENES, scene), dict(sample_visible=mode))
            i = [k for k, sh in enumerate(osc.flat.shapes) if sh["bsdf"] == bsdf][0]
            assert osc.flat.shapes[i]["sample_all"] == int(mode == "false")
            wi = np.float32([0.35, -0.2, 0.0]); wi[2] = np.sqrt(1 - wi[0] ** 2 - wi[1] ** 2)
            n, acc, inv, worst = 20000, 0.0, 0.0, 0.0
            for s3 in rng.random((n, 3)).astype(np.float32):
                out = np.zeros(13, np.float32)
                L.orc_kat_bsdf(C.byref(osc.c.shapes[i]), wi.ctypes.data, np.float32([0, 0, 1]).ctypes.data, s3.ctypes.data, out.ctypes.data)
                wo, bs_pdf, w = out[4:7].copy(), float(out[7]), out[10:13].astype(np.float64)
                if bs_pdf <= 0 or not w.any():
                    continue
                chk = np.zeros(13, np.float32)
                L.orc_kat_bsdf(C.byref(osc.c.shapes[i]), wi.ctypes.data, wo.ctypes.data, s3.ctypes.data, chk.ctypes.data)
                val, pdf = chk[0:3].astype(np.float64), float(chk[3])
                assert abs(pdf - bs_pdf) <= 2e-3 * max(pdf, bs_pdf), (scene, mode, pdf, bs_pdf)          # BSDF::pdf of the sampled direction is the sample's density
                # roughdielectric samples a distribution of scaled roughness but weights with the unscaled one (roughdielectric.cpp:266-269 vs
                # :345-349): there the weight is deliberately not eval / pdf
                if not (bsdf == 7 and mode == "false"):
                    worst = max(worst, float(np.abs(val / pdf - w).max() / max(np.abs(w).max(), 1e-6)))
                acc += w.mean(); inv += 1.0 / bs_pdf
            assert worst < 5e-3, (scene, mode, worst)
            albedo[scene, mode] = acc / n
            if bsdf != 7:
                assert abs(inv / n / (2 * np.pi) - 1) < 0.1, (scene, mode, inv / n)   # heavy-tailed estimator: loose bound
        a, b = albedo[scene, "true"], albedo[scene, "false"]
        assert 0.05 < a <= 1.05 and abs(a - b) < (0.06 if scene == "cornell_frosted.xml" else 0.04) * max(a, b), (scene, a, b)
