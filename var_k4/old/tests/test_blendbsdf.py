"""The `blendbsdf` BSDF (src/bsdfs/blendbsdf.cpp): loader semantics on both loaders (CPU), analytic checks on the GPU.
The per-lane parity of a scene full of blends is the `blend` configuration of tests/conftest.py (test_gpu_parity.py)."""
import numpy as np
import pytest

SCENE = '<scene version="3.0.0">%s</scene>'
SHAPE = '<shape type="rectangle">%s</shape>'
DIFFUSE = '<bsdf type="diffuse"><rgb name="reflectance" value="0.2, 0.4, 0.6"/></bsdf>'
METAL = '<bsdf type="roughconductor"><float name="alpha" value="0.3"/></bsdf>'


def both(mi, orc):
    return [("product", lambda xml: mi.load_string(xml)), ("oracle", lambda xml: orc.Scene(xml, {}, is_string=True))]


def blend(weight, a=DIFFUSE, b=METAL):
    return '<bsdf type="blendbsdf">%s%s%s</bsdf>' % (weight, a, b)


def test_blendbsdf_loads_with_constant_and_textured_weights(mi, orc):
    """blendbsdf.cpp:80-104: two nested BSDFs (each with its own adapters) and a weight; a twosided around the blend reaches both partners"""
    w = '<float name="weight" value="0.3"/>'
    tex = '<texture type="checkerboard" name="weight"><rgb name="color0" value="0.2"/><rgb name="color1" value="0.6"/></texture>'
    xml = SCENE % (SHAPE % blend(w) + SHAPE % ('<bsdf type="twosided">%s</bsdf>' % blend(tex))
                   + SHAPE % blend(w, '<bsdf type="twosided">%s</bsdf>' % DIFFUSE, '<bsdf type="dielectric"/>') + SHAPE % DIFFUSE)
    sc = mi.load_string(xml)
    rec = np.asarray(sc.export(23), np.float32).reshape(-1, 5)          # is a blend, weight, texture, kind of bsdf_1, bsdf_1 two-sided
    assert rec[:, 0].tolist() == [1, 1, 1, 0]
    np.testing.assert_allclose(rec[:3, 1], [0.3, 0.4, 0.3], atol=1e-7)
    assert rec[:, 2].tolist() == [-1, 0, -1, -1] and rec[:, 3].tolist() == [4, 4, 2, -1] and rec[:, 4].tolist() == [0, 1, 0, 0]
    bs = np.asarray(sc.export(9), np.float32).reshape(-1, 24)
    assert bs[:, 0].tolist() == [0, 0, 0, 0] and bs[:, 1].tolist() == [0, 1, 1, 0]      # bsdf_0: diffuse everywhere; two-sided where an adapter says so
    assert sc.info()["n_shapes"] == 4 + 3                                # one material-only record per blend behind the real shapes
    fs = orc.Scene(xml, {}, is_string=True).flat
    assert [s["blend_other"] is not None for s in fs.shapes] == [True, True, True, False]
    assert [s["blend_other"]["bsdf"] for s in fs.shapes[:3]] == [4, 4, 2] and [s["blend_other"]["twosided"] for s in fs.shapes[:3]] == [0, 1, 0]
    np.testing.assert_allclose([float(s["blend_weight"]) for s in fs.shapes[:3]], [0.3, 0.4, 0.3], atol=1e-7)


@pytest.mark.parametrize("bsdf,message", [
    (blend('<float name="weight" value="0.5"/>', DIFFUSE, METAL + DIFFUSE), "BlendBSDF: Cannot specify more than two child BSDFs"),
    ('<bsdf type="blendbsdf"><float name="weight" value="0.5"/>%s</bsdf>' % DIFFUSE, "BlendBSDF: Two child BSDFs must be specified"),
    (blend(""), 'Property "weight" has not been specified'),
    ('<bsdf type="twosided">%s</bsdf>' % blend('<float name="weight" value="0.5"/>', DIFFUSE, '<bsdf type="dielectric"/>'), "Only materials without a transmission component can be nested"),
    (blend('<rgb name="weight" value="0.3, 0.4, 0.5"/>'), 'rgb "weight" is not supported'),
    (blend('<float name="weight" value="0.5"/>', '<bsdf type="mask">%s</bsdf>' % DIFFUSE, METAL), "nested in a blendbsdf is not supported"),
    (blend('<float name="weight" value="0.5"/><float name="wieght" value="1"/>'), "wieght"),
])
def test_blendbsdf_errors(mi, orc, bsdf, message):
    for name, load in both(mi, orc):
        with pytest.raises(Exception, match=message):
            load(SCENE % (SHAPE % bsdf))


SENSOR = ('<sensor type="perspective"><float name="fov" value="30"/><transform name="to_world"><lookat origin="0.3, 0.2, 4" target="0, 0, 0" up="0, 1, 0"/></transform>'
          '<film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/><rfilter type="box"/></film>'
          '<sampler type="independent"><integer name="sample_count" value="512"/></sampler></sensor>')
LIGHT = '<emitter type="point"><point name="position" value="1, 1, 3"/><rgb name="intensity" value="10"/></emitter>'
FLOOR = '<shape type="rectangle"><transform name="to_world"><rotate x="1" angle="-90"/><scale value="4"/><translate y="-1"/></transform>%s</shape>' % DIFFUSE


def lanes(mi, bsdf, depth=4):
    sc = mi.load_string(SCENE % ('<integrator type="path"><integer name="max_depth" value="%d"/></integrator>' % depth + SENSOR + LIGHT + SHAPE % bsdf + FLOOR))
    return sc.sample_lanes(seed=4, spp=512, lane_begin=0, n=8 * 8 * 512)["rgb"]


@pytest.mark.gpu
def test_weights_zero_and_one_are_the_partners_alone(mi):
    """weight 0: every sample goes to bsdf_0 with sample1 unchanged and eval = eval_0 * 1 + eval_1 * 0; weight 1: bsdf_1 alone -- the lanes equal the plain BSDF's bit for bit"""
    plain_a, plain_b = lanes(mi, DIFFUSE), lanes(mi, METAL)
    assert np.array_equal(lanes(mi, blend('<float name="weight" value="0"/>')).view(np.uint32), plain_a.view(np.uint32))
    assert np.array_equal(lanes(mi, blend('<float name="weight" value="1"/>')).view(np.uint32), plain_b.view(np.uint32))
    assert not np.array_equal(plain_a, plain_b)


@pytest.mark.gpu
def test_a_blend_is_the_weighted_mean_of_its_partners_in_expectation(mi):
    """E[blend(w)] = (1 - w) E[a] + w E[b] for direct light (max_depth 2: emitter sampling only; the eval of the blend is the weighted sum)"""
    a, b = lanes(mi, DIFFUSE, 2).mean(), lanes(mi, METAL, 2).mean()
    m = lanes(mi, blend('<float name="weight" value="0.3"/>'), 2).mean()
    assert a > 0 and b > 0 and abs(a - b) > 0.05 * a
    assert abs(m - (0.7 * a + 0.3 * b)) < 1e-5 * max(a, b)      # the same emitter samples in all three renders: the identity holds per lane, up to rounding
