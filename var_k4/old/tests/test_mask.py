"""The `mask` BSDF (src/bsdfs/mask.cpp): loader semantics on both loaders (CPU), analytic checks of the null interaction on the GPU.
The per-lane parity of a scene full of masks is the `masked` configuration of tests/conftest.py (test_gpu_parity.py)."""
import numpy as np
import pytest

SCENE = '<scene version="3.0.0">%s</scene>'
SHAPE = '<shape type="rectangle">%s</shape>'
DIFFUSE = '<bsdf type="diffuse"><rgb name="reflectance" value="0.2, 0.4, 0.6"/></bsdf>'


def both(mi, orc):
    return [("product", lambda xml: mi.load_string(xml)), ("oracle", lambda xml: orc.Scene(xml, {}, is_string=True))]


def test_mask_loads_with_constant_default_and_textured_opacity(mi, orc):
    """mask.cpp:93-117: `opacity` is a float or a texture (default 0.5) over exactly one nested BSDF, which may itself be two-sided"""
    xml = SCENE % (SHAPE % ('<bsdf type="mask"><float name="opacity" value="0.25"/>%s</bsdf>' % DIFFUSE)
                   + SHAPE % ('<bsdf type="mask"><bsdf type="twosided">%s</bsdf></bsdf>' % DIFFUSE)
                   + SHAPE % ('<bsdf type="mask"><texture type="checkerboard" name="opacity"><rgb name="color0" value="0.2"/><rgb name="color1" value="0.8"/></texture>'
                              '<bsdf type="conductor"/></bsdf>')
                   + SHAPE % DIFFUSE)
    sc = mi.load_string(xml)
    rec = np.asarray(sc.export(20), np.float32).reshape(-1, 3)          # masked, opacity, texture index
    np.testing.assert_allclose(rec[:, 0], [1, 1, 1, 0])
    np.testing.assert_allclose(rec[:, 1], [0.25, 0.5, 0.5, 1.0], atol=1e-7)   # the checkerboard's mean stands in for the constant
    assert rec[:, 2].tolist() == [-1, -1, 0, -1]
    bs = np.asarray(sc.export(9), np.float32).reshape(-1, 24)
    assert bs[:, 0].tolist() == [0, 0, 1, 0] and bs[:, 1].tolist() == [0, 1, 0, 0]      # the nested BSDF's kind and two-sidedness are kept
    fs = orc.Scene(xml, {}, is_string=True).flat
    assert [s["masked"] for s in fs.shapes] == [1, 1, 1, 0]
    np.testing.assert_allclose([float(s["opacity"]) for s in fs.shapes], [0.25, 0.5, 0.5, 1.0], atol=1e-7)
    assert [s["tex_opacity"] is not None for s in fs.shapes] == [False, False, True, False]
    assert [s["twosided"] for s in fs.shapes] == [0, 1, 0, 0]


@pytest.mark.parametrize("bsdf,message", [
    ('<bsdf type="mask">%s%s</bsdf>' % (DIFFUSE, DIFFUSE), "Cannot specify more than one child BSDF"),
    ('<bsdf type="mask"><float name="opacity" value="0.3"/></bsdf>', "Child BSDF not specified"),
    ('<bsdf type="twosided"><bsdf type="mask">%s</bsdf></bsdf>' % DIFFUSE, "Only materials without a transmission component can be nested"),
    ('<bsdf type="mask"><rgb name="opacity" value="0.3, 0.4, 0.5"/>%s</bsdf>' % DIFFUSE, 'rgb "opacity" is not supported'),
    ('<bsdf type="mask"><float name="opacity" value="0.3"/><float name="opaqueness" value="1"/>%s</bsdf>' % DIFFUSE, "opaqueness"),
])
def test_mask_errors(mi, orc, bsdf, message):
    for name, load in both(mi, orc):
        with pytest.raises(Exception, match=message):
            load(SCENE % (SHAPE % bsdf))


SENSOR = ('<sensor type="perspective"><float name="fov" value="20"/><transform name="to_world"><lookat origin="0, 0, 4" target="0, 0, 0" up="0, 1, 0"/></transform>'
          '<film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/><rfilter type="box"/></film>'
          '<sampler type="independent"><integer name="sample_count" value="4096"/></sampler></sensor>')
LIGHT = '<emitter type="point"><point name="position" value="0, 0, 4"/><rgb name="intensity" value="10"/></emitter>'
WALL = '<shape type="rectangle"><transform name="to_world"><scale value="3"/><translate z="-1"/></transform><bsdf type="diffuse"><rgb name="reflectance" value="0.5"/></bsdf></shape>'
VEIL = '<shape type="rectangle"><transform name="to_world"><scale value="3"/></transform><bsdf type="mask"><float name="opacity" value="%s"/><bsdf type="diffuse"><rgb name="reflectance" value="0.5"/></bsdf></bsdf></shape>'


@pytest.mark.gpu
def test_fully_transparent_and_fully_opaque_masks(mi):
    """opacity 0: the veil is not there for the camera path (every interaction is the null one) -- but it still blocks the light: occlusion tests do not look
    at BSDFs (Scene::ray_test), so with the light behind the camera the wall goes dark exactly as in the reference; opacity 1: the veil alone"""
    def image(xml):
        sc = mi.load_string(SCENE % ('<integrator type="path"><integer name="max_depth" value="3"/></integrator>' + SENSOR + LIGHT + xml))
        return np.asarray(sc.render(seed=1))
    wall = image(WALL)
    clear = image(WALL + VEIL % "0")
    solid = image(WALL + VEIL % "1")
    alone = image(VEIL % "1")
    assert wall.mean() > 1e-3
    assert np.all(clear == 0)                                # through the veil the camera sees the wall, whose light is blocked by the veil
    np.testing.assert_allclose(solid, alone, rtol=1e-6)      # an opaque mask hides the wall completely: same paths (the film sums them in another order)


@pytest.mark.gpu
def test_half_transparent_mask_is_the_opacity_weighted_veil(mi):
    """in expectation the direct light reflected by a veil of opacity a is a x the opaque veil's (eval is scaled by a; the paths that go through see a wall in
    the veil's shadow): mean(a = 0.5) / mean(a = 1) = 0.5 within Monte Carlo noise"""
    def mean(a):
        sc = mi.load_string(SCENE % ('<integrator type="path"><integer name="max_depth" value="2"/></integrator>' + SENSOR + LIGHT + WALL + VEIL % a))
        return float(np.asarray(sc.render(seed=5)).mean())
    full, half = mean("1"), mean("0.5")
    assert full > 1e-3 and abs(half / full - 0.5) < 0.02
