"""Area emitters with a textured `radiance` (src/emitters/area.cpp:129-176): the emitter is sampled THROUGH the texture
(Texture::sample_position -> Shape::eval_parameterization), evaluated at si.uv on a hit, and its MIS density is the texture's pdf_position.
DiscreteDistribution2D against the reference's own test (src/core/tests/test_distr_2d.py:165-179), both loaders (CPU), analytic checks on the GPU;
the per-lane parity is the `textured_light` configuration of tests/conftest.py (test_gpu_parity.py)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import SCENES

SCENE = '<scene version="3.0.0">%s</scene>'


def _kat_texture(orc, data, filt=0, wrap=0):
    d = np.asarray(data, np.float32)
    tex = dict(kind=1, filter=filt, wrap=wrap, channels=1 if d.ndim == 2 else 3, width=d.shape[1], height=d.shape[0], to_uv=np.array([1, 0, 0, 1], np.float32),
               color0=np.zeros(3, np.float32), color1=np.zeros(3, np.float32), data=np.ascontiguousarray(d.reshape(-1)), mean=float(d.mean()))
    class Keep:
        _keep = []
    k = Keep()
    return orc.Scene._make_texture(k, tex, distribution=True), k


def test_discrete_distribution_2d_reference_values(orc):
    """test_distr_2d.py:165-179 (test05_discrete_distribution_2d): DiscreteDistribution2D([[1, 2, 3], [0, 1, 3]]).sample -> (position, pdf, re-uniformised sample), atol 1e-6"""
    L = orc.lib()
    L.orc_kat_distr2d_sample.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    t, keep = _kat_texture(orc, [[1, 2, 3], [0, 1, 3]])
    def sample(x, y):
        out = np.zeros(5, np.float32)
        L.orc_kat_distr2d_sample(t, C.c_float(x), C.c_float(y), out.ctypes.data)
        return out
    for (x, y), (pos, pdf, rest) in [((0, 0), ([0, 0], .1, [0, 0])), ((1.0 / 6.0 - 1e-7, 0), ([0, 0], .1, [1, 0])), ((1.0 / 6.0 + 1e-7, 0), ([1, 0], .2, [0, 0])),
                                     ((1, 0), ([2, 0], .3, [1, 0])), ((0, 6 / 10 - 1e-7), ([0, 0], .1, [0, 1])), ((0, 6 / 10 + 1e-7), ([1, 1], .1, [0, 0]))]:
        got = sample(x, y)
        np.testing.assert_allclose(got, pos + [pdf] + rest, atol=1e-6)


def test_texture_sample_position_follows_its_density(orc):
    """BitmapTexture::sample_position / pdf_position (bitmap.cpp:450-528): nearest filter -- positions land in texels in proportion to their values and the reported
    density is pdf_position there; bilinear -- the density at the returned position equals pdf_position; a histogram of 200 000 samples follows the texel weights"""
    L = orc.lib()
    L.orc_kat_texture_sample_position.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    L.orc_kat_texture_pdf_position.argtypes = [C.c_void_p, C.c_float, C.c_float]; L.orc_kat_texture_pdf_position.restype = C.c_float
    data = np.array([[1, 4, 2, 0.5], [3, 0.25, 1, 6], [2, 2, 0, 1]], np.float32)
    rng = np.random.default_rng(7)
    for filt in (0, 1):
        t, keep = _kat_texture(orc, data, filt=filt)
        hist = np.zeros(data.shape)
        out = np.zeros(3, np.float32)
        for s in rng.random((200000 if filt == 0 else 20000, 2)):
            L.orc_kat_texture_sample_position(t, C.c_float(s[0]), C.c_float(s[1]), out.ctypes.data)
            assert 0 <= out[0] <= 1 and 0 <= out[1] <= 1
            if filt == 0:
                hist[min(int(out[1] * 3), 2), min(int(out[0] * 4), 3)] += 1
                assert abs(out[2] - L.orc_kat_texture_pdf_position(t, C.c_float(float(out[0])), C.c_float(float(out[1])))) <= 1e-5 * max(1.0, out[2])
        if filt == 0:
            np.testing.assert_allclose(hist / hist.sum(), data / data.sum(), atol=4e-3)


LIGHT = ('<shape type="rectangle"><transform name="to_world"><scale x="0.5" y="0.5"/><rotate x="1" angle="180"/><translate y="0" z="3"/></transform>'
         '<emitter type="area">%s</emitter></shape>')
FLOOR = '<shape type="rectangle"><transform name="to_world"><scale value="3"/></transform><bsdf type="diffuse"><rgb name="reflectance" value="0.5"/></bsdf></shape>'
SENSOR = ('<sensor type="perspective"><float name="fov" value="40"/><transform name="to_world"><lookat origin="0, -4, 2" target="0, 0, 0" up="0, 0, 1"/></transform>'
          '<film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/><rfilter type="box"/></film>'
          '<sampler type="independent"><integer name="sample_count" value="2048"/></sampler></sensor>')


def test_textured_radiance_loads_on_rectangles_only(mi, orc):
    tex = '<texture type="bitmap" name="radiance"><string name="filename" value="%s"/></texture>' % os.path.join(SCENES, "tex_rgb.png")
    xml = SCENE % (LIGHT % tex + LIGHT % '<rgb name="radiance" value="1, 2, 3"/>' + FLOOR)
    sc = mi.load_string(xml)
    assert sc.info()["n_emitters"] == 2
    assert np.asarray(sc.export(24), np.float32).tolist() == [0, -1, -1]          # per shape: index of the radiance texture
    fs = orc.Scene(xml, {}, is_string=True).flat
    assert [s["tex_radiance"] is not None for s in fs.shapes] == [True, False, False]
    for load in (lambda x: mi.load_string(x), lambda x: orc.Scene(x, {}, is_string=True)):
        with pytest.raises(Exception, match="rectangles only"):
            load(SCENE % ('<shape type="sphere"><emitter type="area">%s</emitter></shape>' % tex))


@pytest.mark.gpu
def test_a_uniform_checkerboard_light_is_the_constant_light(mi):
    """both colours equal: the emitter is 'spatially varying' for the code (uv sampling, |dp_du x dp_dv| densities) but constant in value -- the image is the constant
    light's up to rounding of the two density formulas"""
    def image(emitter):
        sc = mi.load_string(SCENE % ('<integrator type="path"><integer name="max_depth" value="3"/></integrator>' + SENSOR + LIGHT % emitter + FLOOR))
        return np.asarray(sc.render(seed=4))
    a = image('<rgb name="radiance" value="2, 3, 4"/>')
    b = image('<texture type="checkerboard" name="radiance"><rgb name="color0" value="2, 3, 4"/><rgb name="color1" value="2, 3, 4"/></texture>')
    assert a.max() > 0 and np.abs(a - b).max() <= 1e-4 * a.max()


@pytest.mark.gpu
def test_a_bitmap_light_emits_its_mean(mi):
    """far from a small emitter the irradiance is proportional to the mean radiance: the floor under a bitmap light against the floor under a constant light of the
    texture's mean colour (a gray file: one channel), within Monte Carlo noise and the near-field difference"""
    gray = os.path.join(SCENES, "tex_gray.png")
    def mean(emitter):
        sc = mi.load_string(SCENE % ('<integrator type="path"><integer name="max_depth" value="2"/></integrator>' + SENSOR + LIGHT % emitter + FLOOR))
        return float(np.asarray(sc.render(seed=6)).mean())
    tex = '<texture type="bitmap" name="radiance"><string name="filename" value="%s"/><boolean name="raw" value="true"/></texture>' % gray
    sys.path.insert(0, SCENES)
    import make_scenes   # the fixture's texel values
    rows = [[(x * x * 3 + y * 29 + 10) % 256 for x in range(8)] for y in range(8)]
    m = float(np.mean(rows)) / 255.0
    t, c = mean(tex), mean('<rgb name="radiance" value="%s"/>' % m)
    assert c > 0 and abs(t / c - 1.0) < 0.03
