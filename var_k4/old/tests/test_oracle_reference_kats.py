"""The CPU oracle against the known answers of the reference's OWN unit tests (tests/golden/reference_kats.json.gz, harvested from
/root/reference by tests/golden/extract_reference_kats.py; values only).  This is what pins oracle == reference component by component:
every record is an assertion of the reference's test suite (file:line recorded), evaluated with the tolerance that assertion states."""
import collections
import ctypes as C

import numpy as np
import pytest

import refkat


class OracleBackend:
    """the facade's backend over oracle/libdtof_oracle.so"""

    def __init__(self, orc):
        self.orc, self.L = orc, orc.lib()

    def load_scene(self, xml):
        try:
            return self.orc.Scene(xml, is_string=True)
        except ValueError as e:
            raise refkat.Skip("loader: %s" % str(e)[:60])

    def ray_intersect(self, sc, o, d, t):
        out, ids = np.zeros(25, np.float32), np.zeros(3, np.int32)
        o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
        hit = self.L.orc_kat_ray_intersect(C.byref(sc.c), o.ctypes.data, d.ctypes.data, C.c_float(t), C.c_float(np.finfo(np.float32).max),
                                           out.ctypes.data, ids.ctypes.data)
        return bool(hit), out

    def ray_test(self, sc, o, d, t):
        o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
        fp = C.POINTER(C.c_float)
        return bool(self.L.orc_occluded(C.byref(sc.c), o.ctypes.data_as(fp), d.ctypes.data_as(fp), C.c_float(t), C.c_float(np.finfo(np.float32).max)))

    def shape_area(self, sc, i):
        return float(self.L.orc_kat_shape_area(C.byref(sc.c.shapes[i])))

    def sphere_sample_direction(self, sc, i, ref, sx, sy):
        out = np.zeros(11, np.float32)
        ref = np.ascontiguousarray(ref, np.float32)
        self.L.orc_kat_sphere_sample_direction(C.byref(sc.c.shapes[i]), ref.ctypes.data, C.c_float(sx), C.c_float(sy), out.ctypes.data)
        return out

    def bsdf(self, sc, i, wi, wo, s3):
        out = np.zeros(13, np.float32)
        wi, wo, s3 = (np.ascontiguousarray(x, np.float32) for x in (wi, wo, s3))
        self.L.orc_kat_bsdf(C.byref(sc.c.shapes[i]), wi.ctypes.data, wo.ctypes.data, s3.ctypes.data, out.ctypes.data)
        return out

    def filter_eval(self, kind, radius, stddev, B, Cc, x):
        return float(self.L.orc_kat_filter(kind, radius, stddev, B, Cc, x))

    def microfacet(self, type_, au, av, visible, fn, inp):
        out = np.zeros((len(inp), 4), np.float32)
        for i, row in enumerate(inp):
            row = np.ascontiguousarray(row, np.float32)
            self.L.orc_kat_microfacet(type_, au, av, int(visible), fn, row.ctypes.data, out[i].ctypes.data)
        return out

    def fresnel(self, c, eta):
        out = np.zeros(4, np.float32)
        self.L.orc_fresnel_dielectric(c, eta, out.ctypes.data)
        return out

    def tea_float32(self, v0, v1, rounds):
        return float(self.L.orc_tea_float32(v0, v1, rounds))

    def coordinate_system(self, n):
        out = np.zeros(6, np.float32)
        n = np.ascontiguousarray(n, np.float32)
        self.L.orc_kat_frame(n.ctypes.data, out.ctypes.data)
        return out[:3].copy(), out[3:].copy()

    def gauss_legendre(self, n):
        nodes, w = np.zeros(n, np.float32), np.zeros(n, np.float32)
        self.L.orc_gauss_legendre(n, nodes.ctypes.data, w.ctypes.data)
        return nodes, w

    def solve_quadratic(self, a, b, c):
        out = np.zeros(2, np.float64)
        ok = self.L.orc_kat_solve_quadratic(a, b, c, out.ctypes.data)
        return bool(ok), out[0], out[1]

    def warp(self, fn, sx, sy):
        inp, out = np.array([sx, sy, 0], np.float32), np.zeros(3, np.float32)
        self.L.orc_kat_warp(fn, inp.ctypes.data, out.ctypes.data)
        return out

    def splat(self, film, f, x, y, rgb):
        se = self.orc.OrcSensor()
        se.crop_w, se.crop_h, se.film_w, se.film_h = film.shape[1], film.shape[0], film.shape[1], film.shape[0]
        se.filter, se.filter_radius, se.filter_stddev, se.filter_b, se.filter_c = f.kind, f.radius, f.stddev, f.B, f.C
        rgb = np.asarray(rgb, np.float32)
        self.L.orc_kat_splat(C.byref(se), film.ctypes.data, C.c_float(x), C.c_float(y), rgb.ctypes.data)

    def sensor_info(self, sc):
        se = sc.flat.sensor
        return dict(shutter_open=float(se["shutter_open"]), shutter_close=float(se["shutter_close"]), focus_distance=float(se["focus_distance"]),
                    to_world=np.asarray(se["to_world"], np.float64).reshape(4, 4))

    def film_info(self, sc):
        se = sc.flat.sensor
        return dict(size=(int(se["film_w"]), int(se["film_h"])), crop_size=(int(se["crop_w"]), int(se["crop_h"])),
                    crop_offset=(int(se["crop_x"]), int(se["crop_y"])))

    def camera_ray(self, sc, px, py, ax=.5, ay=.5):
        out = np.zeros(7, np.float32)
        if sc.c.sensor.kind == 0:   # film position in pixels
            self.L.orc_camera_ray(C.byref(sc.c.sensor), px, py, out.ctypes.data_as(C.POINTER(C.c_float)))
        else:   # ThinLensCamera / OrthographicCamera: position sample in [0, 1]^2 of the crop window, aperture sample
            se = sc.c.sensor
            self.L.orc_camera_sample_ray(C.byref(se), (px - se.crop_x) / se.crop_w, (py - se.crop_y) / se.crop_h, ax, ay, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out[0:3].copy(), out[3:6].copy()


@pytest.fixture(scope="module")
def results(orc):
    return refkat.run_all(OracleBackend(orc))


# reference test file -> (SURVEY 8(a) rows it pins, minimum number of its assertions the oracle must reproduce)
PINNED = {
    "src/render/tests/test_microfacet.py": ("8(f)-3 rough BSDFs: MicrofacetDistribution eval / pdf / smith_g1 / sample, GGX and Beckmann", 20),
    "src/render/tests/test_fresnel.py": ("8(f)-3 fresnel", 10),
    "src/rfilters/tests/test_rfilter.py": ("I1 reconstruction filters", 8),
    "src/core/tests/test_warp.py": ("M1 / E1 warps", 10),
    "src/core/tests/test_random.py": ("S7 TEA", 8),
    "src/core/tests/test_frame.py": ("G4 coordinate_system", 3),
    "src/core/tests/test_quad.py": ("8(f)-3 Gauss-Legendre nodes of the roughplastic tables", 2),
    "src/shapes/tests/test_rectangle.py": ("G3 rectangle", 15),
    "src/shapes/tests/test_sphere.py": ("8(f)-3 sphere", 500),
    "src/shapes/tests/test_disk.py": ("8(f)-3 disk", 500),
    "src/shapes/tests/test_cube.py": ("G3 cube mesh", 100),
    "src/shapes/tests/test_cylinder.py": ("8(f)-3 cylinder", 300),
    "src/sensors/tests/test_perspective.py": ("C1 perspective camera", 20),
    "src/sensors/tests/test_orthographic.py": ("8(f) orthographic sensor: constructor, ray origins on the near plane, parallel directions", 40),
    "src/sensors/tests/test_thinlens.py": ("D2 aperture sample / thinlens camera: constructor, sample_ray with aperture samples, fov axes", 200),
    "src/render/tests/test_imageblock.py": ("I1 ImageBlock::put", 1),
    "src/bsdfs/tests/test_diffuse.py": ("M1 diffuse", 30),
    "src/bsdfs/tests/test_twosided.py": ("M1 twosided", 2),
    "src/bsdfs/tests/test_dielectric.py": ("8(f)-3 dielectric", 20),
    "src/films/tests/test_hdrfilm.py": ("I1 / X1 film size and crop window", 3),
}


def test_no_reference_assertion_fails(results):
    """every harvested assertion that the facade can evaluate holds, with the reference's own tolerance"""
    bad = {f: st["fail"][:8] for f, st in results.items() if st["fail"]}
    assert not bad, bad


@pytest.mark.parametrize("ref_file", sorted(PINNED))
def test_component_is_pinned(results, ref_file):
    what, need = PINNED[ref_file]
    st = results.get(ref_file)
    assert st is not None, "no records harvested from " + ref_file
    assert st["pass"] >= need, "%s: only %d assertions of %s reproduced (skips: %s)" % (what, st["pass"], ref_file, st["skip"].most_common(5))


def test_report(results, capsys):
    total = sum(st["pass"] for st in results.values())
    with capsys.disabled():
        print("\nreference known answers reproduced by the oracle: %d" % total)
        for f in sorted(results):
            st = results[f]
            print("  %-46s pass %5d  fail %3d  skip %5d  %s" % (f, st["pass"], len(st["fail"]), sum(st["skip"].values()),
                                                                 dict(st["passed_tests"])))
    assert total >= 1500


def test_instance_relations(results):
    """src/shapes/tests/test_instance.py asserts that an instanced shape behaves like the same shape placed directly (hit / miss, t, p,
    frame, dp_du, dp_dv, wi within 2e-2).  Rays through the exact edge of the primitive may differ in the last float32 bit of the two
    transform chains; the reference's own Embree path has the same property, so those are bounded instead of forbidden."""
    st = results["src/shapes/tests/test_instance.py"]
    assert st["pass"] >= 500, (st["pass"], st["skip"].most_common(4))
    assert len(st.get("edge", [])) <= 0.01 * st["pass"], st.get("edge")
