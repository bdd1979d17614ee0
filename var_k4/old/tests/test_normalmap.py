"""The `normalmap` BSDF (src/bsdfs/normalmap.cpp): loader semantics on both loaders (CPU), analytic checks on the GPU.
The per-lane parity of a scene with normal-mapped walls and boxes is the `normalmap` configuration of tests/conftest.py (test_gpu_parity.py)."""
import numpy as np
import pytest

SCENE = '<scene version="3.0.0">%s</scene>'
SHAPE = '<shape type="rectangle">%s</shape>'
DIFFUSE = '<bsdf type="diffuse"><rgb name="reflectance" value="0.2, 0.4, 0.6"/></bsdf>'
FLAT = '<texture type="checkerboard" name="normalmap"><rgb name="color0" value="0.5, 0.5, 1"/><rgb name="color1" value="0.5, 0.5, 1"/></texture>'


def both(mi, orc):
    return [("product", lambda xml: mi.load_string(xml)), ("oracle", lambda xml: orc.Scene(xml, {}, is_string=True))]


def test_normalmap_loads_plain_and_inside_the_adapters(mi, orc):
    """normalmap.cpp:84-108: exactly one nested BSDF and an RGB texture; twosided and mask go around it"""
    xml = SCENE % (SHAPE % ('<bsdf type="normalmap">%s%s</bsdf>' % (FLAT, DIFFUSE))
                   + SHAPE % ('<bsdf type="twosided"><bsdf type="normalmap">%s<bsdf type="roughconductor"/></bsdf></bsdf>' % FLAT)
                   + SHAPE % ('<bsdf type="mask"><bsdf type="twosided"><bsdf type="normalmap">%s%s</bsdf></bsdf></bsdf>' % (FLAT, DIFFUSE))
                   + SHAPE % DIFFUSE)
    sc = mi.load_string(xml)
    assert np.asarray(sc.export(21), np.float32).tolist() == [0, 1, 2, -1]            # index of each shape's normal map in the texture table
    bs = np.asarray(sc.export(9), np.float32).reshape(-1, 24)
    assert bs[:, 0].tolist() == [0, 4, 0, 0] and bs[:, 1].tolist() == [0, 1, 1, 0]
    assert np.asarray(sc.export(20), np.float32).reshape(-1, 3)[:, 0].tolist() == [0, 0, 1, 0]
    fs = orc.Scene(xml, {}, is_string=True).flat
    assert [s["tex_normal"] is not None for s in fs.shapes] == [True, True, True, False]
    assert [s["twosided"] for s in fs.shapes] == [0, 1, 1, 0] and [s["masked"] for s in fs.shapes] == [0, 0, 1, 0]


@pytest.mark.parametrize("bsdf,message", [
    ('<bsdf type="normalmap">%s%s%s</bsdf>' % (FLAT, DIFFUSE, DIFFUSE), "Only a single BSDF child object can be specified"),
    ('<bsdf type="normalmap">%s</bsdf>' % FLAT, "Exactly one BSDF child object must be specified"),
    ('<bsdf type="normalmap">%s</bsdf>' % DIFFUSE, 'Property "normalmap" has not been specified'),
    ('<bsdf type="normalmap">%s<bsdf type="twosided">%s</bsdf></bsdf>' % (FLAT, DIFFUSE), 'nested in a normalmap is not supported'),
    ('<bsdf type="normalmap">%s%s<float name="strength" value="2"/></bsdf>' % (FLAT, DIFFUSE), "strength"),
])
def test_normalmap_errors(mi, orc, bsdf, message):
    for name, load in both(mi, orc):
        with pytest.raises(Exception, match=message):
            load(SCENE % (SHAPE % bsdf))


SENSOR = ('<sensor type="perspective"><float name="fov" value="30"/><transform name="to_world"><lookat origin="0.3, 0.2, 4" target="0, 0, 0" up="0, 1, 0"/></transform>'
          '<film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/><rfilter type="box"/></film>'
          '<sampler type="independent"><integer name="sample_count" value="256"/></sampler></sensor>')
LIGHT = '<emitter type="point"><point name="position" value="1, 1, 3"/><rgb name="intensity" value="10"/></emitter>'
INTEGRATOR = '<integrator type="path"><integer name="max_depth" value="3"/></integrator>'


@pytest.mark.gpu
def test_a_flat_normal_map_on_an_axis_aligned_rectangle_changes_nothing(mi):
    """rgb (0.5, 0.5, 1) is the normal (0, 0, 1); on the untransformed rectangle dp_du = (2, 0, 0), so NormalMap::frame returns s = (1, 0, 0), t = (0, 1, 0),
    n = (0, 0, 1): the identity.  Every lane of the normal-mapped scene equals the plain one bit for bit."""
    def lanes(bsdf):
        sc = mi.load_string(SCENE % (INTEGRATOR + SENSOR + LIGHT + SHAPE % bsdf
                                     + '<shape type="rectangle"><transform name="to_world"><translate z="1"/><rotate y="1" angle="70"/><translate x="-1.5"/></transform>%s</shape>' % DIFFUSE))
        return sc.sample_lanes(seed=2, spp=256, lane_begin=0, n=8 * 8 * 256)
    plain, mapped = lanes(DIFFUSE), lanes('<bsdf type="normalmap">%s%s</bsdf>' % (FLAT, DIFFUSE))
    assert np.array_equal(plain["rgb"].view(np.uint32), mapped["rgb"].view(np.uint32))
    assert float(np.abs(plain["rgb"]).max()) > 0


@pytest.mark.gpu
def test_a_tilted_normal_changes_the_shading_and_leaks_are_cut(mi):
    """a constant normal tilted towards +x brightens a light from +x and darkens one from -x relative to the flat surface; a normal tilted beyond the light's
    grazing direction cuts the direct light completely (cos_theta(wo) * cos_theta(perturbed wo) <= 0 or the nested cosine <= 0)"""
    def mean(rgb, light_x):
        tex = '<texture type="checkerboard" name="normalmap"><rgb name="color0" value="%s"/><rgb name="color1" value="%s"/></texture>' % (rgb, rgb)
        sc = mi.load_string(SCENE % ('<integrator type="path"><integer name="max_depth" value="2"/></integrator>' + SENSOR
                                     + '<emitter type="point"><point name="position" value="%s, 0, 1"/><rgb name="intensity" value="10"/></emitter>' % light_x
                                     + SHAPE % ('<bsdf type="normalmap">%s%s</bsdf>' % (tex, DIFFUSE))))
        return float(np.asarray(sc.render(seed=3)).mean())
    flat_r, flat_l = mean("0.5, 0.5, 1", 3), mean("0.5, 0.5, 1", -3)
    tilt_r, tilt_l = mean("0.8, 0.5, 0.9", 3), mean("0.8, 0.5, 0.9", -3)
    assert flat_r > 0 and abs(flat_r / flat_l - 1) < 0.2
    assert tilt_r > 1.3 * flat_r and tilt_l < 0.7 * flat_l
    assert mean("1.0, 0.5, 0.55", -3) == 0.0           # the normal points along +x: a light from -x is below the perturbed horizon


# ------------------------------------------------------------------------------------------------ bumpmap (src/bsdfs/bumpmap.cpp)
def _png(path, rows):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes"))
    import make_scenes
    make_scenes.write_png(str(path), rows)
    return str(path)


def _bump(filename, scale, inner=DIFFUSE, extra=""):
    return ('<bsdf type="bumpmap"><float name="scale" value="%s"/><texture type="bitmap"><string name="filename" value="%s"/><boolean name="raw" value="true"/>%s</texture>%s</bsdf>'
            % (scale, filename, extra, inner))


def test_bumpmap_loads_and_refuses_what_the_reference_refuses(mi, orc, tmp_path):
    """bumpmap.cpp:84-112: exactly one nested BSDF, exactly one texture (under any name), `scale` (default 1)"""
    flat = _png(tmp_path / "flat.png", [[128] * 4] * 4)
    xml = SCENE % (SHAPE % _bump(flat, "0.3") + SHAPE % ('<bsdf type="twosided">%s</bsdf>' % _bump(flat, "2", '<bsdf type="conductor"/>')) + SHAPE % DIFFUSE)
    sc = mi.load_string(xml)
    assert np.asarray(sc.export(22), np.float32).reshape(-1, 2).tolist() == [[1, np.float32(0.3)], [1, 2], [0, 1]]
    assert np.asarray(sc.export(21), np.float32).tolist() == [0, 1, -1]
    fs = orc.Scene(xml, {}, is_string=True).flat
    assert [s["bumpmap"] for s in fs.shapes] == [1, 1, 0] and [s["twosided"] for s in fs.shapes] == [0, 1, 0]
    for bsdf, message in [
            ('<bsdf type="bumpmap">%s</bsdf>' % DIFFUSE, "Exactly one Texture child object must be specified"),
            ('<bsdf type="bumpmap"><texture type="bitmap"><string name="filename" value="%s"/></texture></bsdf>' % flat, "Exactly one BSDF child object must be specified"),
            (_bump(flat, "1", DIFFUSE + DIFFUSE), "Only a single BSDF child object can be specified"),
            (_bump(flat, "1", DIFFUSE + '<texture type="bitmap" name="b"><string name="filename" value="%s"/></texture>' % flat), "Only a single Texture child object can be specified"),
            (_bump(flat, "1", '<bsdf type="twosided">%s</bsdf>' % DIFFUSE), "nested in a bumpmap is not supported"),
            ('<bsdf type="bumpmap"><texture type="checkerboard" name="t"/>%s</bsdf>' % DIFFUSE, "must be a bitmap")]:
        for name, load in both(mi, orc):
            with pytest.raises(Exception, match=message):
                load(SCENE % (SHAPE % bsdf))


@pytest.mark.gpu
def test_a_constant_height_map_changes_nothing_and_a_ramp_tilts_the_normal(mi, tmp_path):
    """zero gradient: the bump-mapped normal is the geometric one, the frame the identity on the untransformed rectangle -- every lane equals the plain scene bit for
    bit; a height ramp rising with u tilts the normal towards -u: a light from -x brightens, one from +x darkens"""
    flat = _png(tmp_path / "flat.png", [[77] * 4] * 4)
    ramp = _png(tmp_path / "ramp.png", [[16 * x for x in range(16)]] * 4)
    def lanes(bsdf):
        sc = mi.load_string(SCENE % (INTEGRATOR + SENSOR + LIGHT + SHAPE % bsdf))
        return sc.sample_lanes(seed=2, spp=256, lane_begin=0, n=8 * 8 * 256)
    plain, mapped = lanes(DIFFUSE), lanes(_bump(flat, "5"))
    assert np.array_equal(plain["rgb"].view(np.uint32), mapped["rgb"].view(np.uint32)) and float(np.abs(plain["rgb"]).max()) > 0
    def mean(bsdf, light_x):
        sc = mi.load_string(SCENE % ('<integrator type="path"><integer name="max_depth" value="2"/></integrator>' + SENSOR
                                     + '<emitter type="point"><point name="position" value="%s, 0, 1"/><rgb name="intensity" value="10"/></emitter>' % light_x + SHAPE % bsdf))
        return float(np.asarray(sc.render(seed=3)).mean())
    clamp = '<string name="wrap_mode" value="clamp"/>'
    up = _bump(ramp, "1", extra=clamp)
    assert mean(up, -3) > 1.2 * mean(DIFFUSE, -3) and mean(up, 3) < 0.8 * mean(DIFFUSE, 3)
    down = _bump(ramp, "-1", extra=clamp)                                  # a negative scale turns the slope around
    assert mean(down, 3) > 1.2 * mean(DIFFUSE, 3) and mean(down, -3) < 0.8 * mean(DIFFUSE, -3)
