"""mitsuba3dopplertof_amd -- host-side mirror (ctypes over the C ABI in include/dtof.h) of the slice of
Mitsuba 3's Python interface that the Doppler-ToF hot path is driven through:

    import mitsuba3dopplertof_amd as mi
    scene = mi.load_file("scene.xml", resx=512, resy=512)            # mi.load_file (program_runner.py:142)
    integrator = mi.load_dict({'type': 'dopplertofpath', ...})      # mi.load_dict (program_runner.py:127-141)
    img = integrator.render(scene, seed=0, spp=64)                  # Integrator.render (integrator.h:74-79)

All compute happens in hand-written HIP kernels inside libdtof.so (csrc/); nothing here falls back to the
CPU: without the built library the import of any compute entry point raises, and without a GPU every
render call raises DtofError.
"""
import ctypes as C
import os

import numpy as np

__all__ = ["load_file", "load_string", "load_dict", "Scene", "Integrator", "Sampler", "DtofError", "render",
           "render_multi_pass", "to_tof_image", "lib_path", "ETimeSampling"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class DtofError(RuntimeError):
    pass


class ETimeSampling:   # include/mitsuba/render/sampler.h:27-34
    UNIFORM, STRATIFIED, ANTITHETIC, ANTITHETIC_MIRROR, PERIODIC, REGULAR = 0, 1, 2, 3, 4, 5


class _Stats(C.Structure):
    _fields_ = [("n_paths", C.c_uint64), ("n_bounces", C.c_uint64), ("n_shadow_rays", C.c_uint64),
                ("ms_total", C.c_double), ("ms_generate", C.c_double), ("ms_trace", C.c_double),
                ("ms_shade", C.c_double), ("ms_shadow", C.c_double), ("ms_splat", C.c_double),
                ("n_launches_trace", C.c_uint32), ("n_launches_shade", C.c_uint32), ("n_launches_shadow", C.c_uint32),
                ("n_batches", C.c_uint32), ("n_launches_first", C.c_uint32), ("ms_first", C.c_double),
                ("n_inline_iterations", C.c_uint32), ("n_bounces_inline", C.c_uint64), ("n_fused_splat_launches", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class _Info(C.Structure):
    _fields_ = [("film_width", C.c_int32), ("film_height", C.c_int32), ("crop_x", C.c_int32), ("crop_y", C.c_int32),
                ("crop_width", C.c_int32), ("crop_height", C.c_int32), ("sample_count", C.c_uint32),
                ("n_shapes", C.c_uint32), ("n_groups", C.c_uint32), ("n_objects", C.c_uint32), ("n_emitters", C.c_uint32),
                ("n_triangles", C.c_uint32), ("n_bvh_nodes", C.c_uint32), ("scene_blob_bytes", C.c_uint32),
                ("time", C.c_float), ("w_g", C.c_float), ("g_1", C.c_float), ("g_0", C.c_float), ("w_s", C.c_float),
                ("phase_offset", C.c_float), ("hetero_frequency", C.c_float), ("antithetic_shift", C.c_float),
                ("wave_type", C.c_int32), ("low_frequency_component_only", C.c_int32), ("time_sampling", C.c_int32),
                ("stratify_each_interval", C.c_int32), ("path_correlation_depth", C.c_uint32), ("max_depth", C.c_uint32),
                ("rr_depth", C.c_uint32), ("base_seed", C.c_uint32), ("time_correlate_number", C.c_int32),
                ("path_correlate_number", C.c_int32), ("bvh_stack_depth", C.c_uint32),
                ("filter_radius", C.c_float), ("filter_halo", C.c_int32), ("has_alpha", C.c_int32)]


def lib_path():
    return os.environ.get("DTOF_LIB") or os.path.join(_HERE, "libdtof.so")   # DTOF_LIB: A/B timing of two builds (tools/ab_time.sh)


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    try:
        # PyTorch wheels ship their own HIP runtime (torch/lib/libamdhip64.so).  One process can host only one
        # runtime, so when torch is installed it must be the first to load it; libdtof.so then binds to the same
        # runtime by soname.  (torch is plumbing for device buffers / torch.distributed, never compute.)
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise ImportError("%s is missing -- build it with `make -C %s/csrc` (or __graft_entry__.build()); "
                          "there is no CPU fallback" % (path, _HERE))
    L = C.CDLL(path)
    cpp = C.POINTER(C.c_char_p)
    vp = C.c_void_p
    L.dtof_version.restype = C.c_char_p
    L.dtof_last_error.restype = C.c_char_p
    L.dtof_scene_load_file.argtypes = [C.c_char_p, cpp, cpp, C.c_int, C.POINTER(vp)]
    L.dtof_scene_load_string.argtypes = [C.c_char_p, cpp, cpp, C.c_int, C.POINTER(vp)]
    L.dtof_scene_destroy.argtypes = [vp]
    L.dtof_scene_destroy.restype = None
    L.dtof_scene_set_integrator.argtypes = [vp, C.c_char_p, cpp, C.c_char_p, cpp, C.c_int]
    L.dtof_scene_set_sampler.argtypes = [vp, C.c_char_p, cpp, C.c_char_p, cpp, C.c_int]
    L.dtof_integrator_create.argtypes = [C.c_char_p, cpp, C.c_char_p, cpp, C.c_int, C.POINTER(C.c_void_p)]
    L.dtof_sampler_plugin_create.argtypes = [C.c_char_p, cpp, C.c_char_p, cpp, C.c_int, C.POINTER(C.c_void_p)]
    L.dtof_integrator_destroy.argtypes = [vp]; L.dtof_integrator_destroy.restype = None
    L.dtof_sampler_plugin_destroy.argtypes = [vp]; L.dtof_sampler_plugin_destroy.restype = None
    L.dtof_integrator_render.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.POINTER(_Stats)]
    L.dtof_scene_get_info.argtypes = [vp, C.POINTER(_Info)]
    L.dtof_scene_export.argtypes = [vp, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.dtof_render.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.POINTER(_Stats)]
    L.dtof_render_rows.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, vp, C.c_int, vp, C.POINTER(_Stats)]
    L.dtof_render_stripes.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int, vp, C.POINTER(_Stats)]
    L.dtof_develop.argtypes = [vp, vp, C.c_int64]
    L.dtof_render_offsets.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_int, vp, C.POINTER(_Stats)]
    L.dtof_cancel.argtypes = [vp]
    L.dtof_cancel.restype = None
    L.dtof_sample_lanes.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp]
    L.dtof_sample_lanes_valid.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp]
    L.dtof_develop_rgba.argtypes = [vp, vp, vp, C.c_int64]
    L.dtof_develop_on_stream.argtypes = [vp, vp, C.c_int64, vp]
    L.dtof_sampler_create.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.dtof_sampler_destroy.argtypes = [vp]
    L.dtof_sampler_destroy.restype = None
    L.dtof_sampler_seed.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.dtof_sampler_set_samples_per_wavefront.argtypes = [vp, C.c_uint32]
    L.dtof_sampler_advance.argtypes = [vp]
    L.dtof_sampler_next_1d.argtypes = [vp, vp]
    L.dtof_sampler_next_2d.argtypes = [vp, vp]
    L.dtof_sampler_next_1d_correlate.argtypes = [vp, vp, C.c_int, vp]
    L.dtof_sampler_next_2d_correlate.argtypes = [vp, vp, C.c_int, vp]
    L.dtof_sampler_next_1d_time.argtypes = [vp, C.c_int, C.c_float, C.c_int, vp]
    L.dtof_sampler_get_state.argtypes = [vp, vp]
    L.dtof_sampler_wavefront_size.argtypes = [vp]
    L.dtof_sampler_wavefront_size.restype = C.c_uint32
    L.dtof_sampler_sample_count.argtypes = [vp]
    L.dtof_sampler_sample_count.restype = C.c_uint32
    L.dtof_sampler_fork.argtypes = [vp, C.POINTER(vp)]
    L.dtof_sampler_clone.argtypes = [vp, C.POINTER(vp)]
    L.dtof_sampler_set_sample_count.argtypes = [vp, C.c_uint32]
    L.dtof_sampler_seeded.argtypes = [vp]
    L.dtof_eval_modulation.argtypes = [vp, C.c_int, vp, vp, vp, C.c_uint32]
    L.dtof_eval_component.argtypes = [C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_uint32]
    L.dtof_render_rows_async.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, vp, C.c_int, vp]
    L.dtof_clear_async.argtypes = [vp, vp, C.c_size_t]
    L.dtof_camera_rays.argtypes = [vp, C.c_uint32, vp, vp]
    L.dtof_bsdf_eval.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp]
    L.dtof_scene_set_stream.argtypes = [vp, vp]
    L.dtof_scene_set_film_layout.argtypes = [vp, C.c_int32, C.c_uint64]
    L.dtof_render_stripes_async.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int, vp]
    L.dtof_develop_async.argtypes = [vp, vp, vp, C.c_int64]
    L.dtof_async_collect.argtypes = [vp, vp, vp, C.c_uint32, vp]
    L.dtof_ray_intersect.argtypes = [vp, C.c_uint32, vp, vp, vp]
    L.dtof_ray_intersect_uv.argtypes = [vp, C.c_uint32, vp, vp, vp, vp]
    L.dtof_ray_test.argtypes = [vp, C.c_uint32, vp, vp]
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise DtofError(_lib().dtof_last_error().decode("utf-8", "replace"))


def _kv(params):
    names = [str(k).encode() for k in params]
    values = [str(v).encode() for v in params.values()]
    n = len(names)
    return (C.c_char_p * max(n, 1))(*names), (C.c_char_p * max(n, 1))(*values), n


def _plugin_args(d):
    """{'type': 'dopplertofpath', 'max_depth': 4, ...} -> (plugin, names, types, values, n) for the C ABI"""
    d = dict(d)
    plugin = d.pop("type", "")
    names, types, values = [], [], []
    for k, v in d.items():
        names.append(str(k).encode())
        if isinstance(v, (bool, np.bool_)):
            types.append("b"); values.append(b"true" if v else b"false")
        elif isinstance(v, (int, np.integer)):
            types.append("i"); values.append(str(int(v)).encode())
        elif isinstance(v, (float, np.floating)):
            types.append("f"); values.append(repr(float(v)).encode())
        elif isinstance(v, str):
            types.append("s"); values.append(v.encode())
        else:
            raise DtofError('unsupported value type for property "%s"' % k)
    n = len(names)
    return (plugin.encode(), (C.c_char_p * max(n, 1))(*names), "".join(types).encode(), (C.c_char_p * max(n, 1))(*values), n)


class Scene:
    """A loaded scene.xml (Scene + Sensor + Film + the integrator/sampler declared in it)."""

    def __init__(self, handle):
        self._h = handle
        self.last_stats = None

    def __del__(self):
        try:
            if self._h:
                _lib().dtof_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def info(self):
        i = _Info()
        _check(_lib().dtof_scene_get_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in i._fields_}

    @property
    def size(self):
        i = self.info()
        return i["crop_width"], i["crop_height"]

    def export(self, kind):
        n = C.c_size_t(0)
        _check(_lib().dtof_scene_export(self._h, kind, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        _check(_lib().dtof_scene_export(self._h, kind, out.ctypes.data, out.size, C.byref(n)))
        return out

    def set_integrator(self, props):
        _check(_lib().dtof_scene_set_integrator(self._h, *_plugin_args(props)))

    def set_sampler(self, props):
        _check(_lib().dtof_scene_set_sampler(self._h, *_plugin_args(props)))

    def render(self, seed=0, spp=0, offsets=None, sensor=0):
        """Developed image (H, W, 3) float32 -- (H, W, 4) for an rgba film; with `offsets` (list of hetero_offset values) -> (K, H, W, 3 | 4)."""
        w, h = self.size
        st = _Stats()
        ch = 4 if self.info()["has_alpha"] else 3
        if offsets is None:
            out = np.zeros((h, w, ch), np.float32)
            _check(_lib().dtof_render(self._h, sensor, seed, spp, out.ctypes.data, C.byref(st)))
        else:
            off = np.ascontiguousarray(offsets, dtype=np.float32)
            out = np.zeros((len(off), h, w, ch), np.float32)
            _check(_lib().dtof_render_offsets(self._h, seed, spp, off.ctypes.data, len(off), out.ctypes.data, C.byref(st)))
        self.last_stats = st.as_dict()
        return out

    def set_film_layout(self, planes, plane_stride_floats=0):
        """Declare the caller's device film for render_rows / render_stripes (dtof_scene_set_film_layout): `planes` RGBW planes, `plane_stride_floats` apart (0 = dense
        H * W * 4).  An rgba scene needs n_offsets + 1 planes -- the alpha film lies behind the colour films -- and is refused until they are declared."""
        _check(_lib().dtof_scene_set_film_layout(self._h, int(planes), int(plane_stride_floats)))

    def film_planes(self, n_offsets=1):
        """RGBW planes the device-film calls write for `n_offsets` batched offsets: one more for the alpha film of an rgba scene"""
        return max(int(n_offsets), 1) + (1 if self.info()["has_alpha"] else 0)

    def render_rows(self, d_film_ptr, seed, spp, row_begin, row_end, offsets=None):
        """Accumulate the undeveloped RGBW film of rows [row_begin,row_end) into a DEVICE buffer (int pointer) of film_planes() planes (set_film_layout for rgba scenes)."""
        st = _Stats()
        if offsets is None:
            _check(_lib().dtof_render_rows(self._h, seed, spp, row_begin, row_end, None, 0, d_film_ptr, C.byref(st)))
        else:
            off = np.ascontiguousarray(offsets, dtype=np.float32)
            _check(_lib().dtof_render_rows(self._h, seed, spp, row_begin, row_end, off.ctypes.data, len(off), d_film_ptr, C.byref(st)))
        self.last_stats = st.as_dict()
        return self.last_stats

    def render_rows_async(self, d_film_ptr, seed, spp, row_begin, row_end, offsets=None):
        """enqueue one frame on the scene's stream without waiting for it (dtof_render_rows_async); collect() waits and returns the timings"""
        offs = np.ascontiguousarray(offsets, np.float32) if offsets is not None else None
        _check(_lib().dtof_render_rows_async(self._h, seed, spp, row_begin, row_end, offs.ctypes.data if offs is not None else None, len(offs) if offs is not None else 0, d_film_ptr))

    def clear_async(self, d_ptr, nbytes):
        _check(_lib().dtof_clear_async(self._h, d_ptr, nbytes))

    def set_stream(self, stream_ptr):
        """enqueue on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream); 0 / None: back to the scene's own (dtof_scene_set_stream)"""
        _check(_lib().dtof_scene_set_stream(self._h, C.c_void_p(stream_ptr or None)))

    def render_stripes_async(self, d_film_ptr, seed, spp, first_row, stripe_rows, stripe_period, offsets=None):
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float32)
        _check(_lib().dtof_render_stripes_async(self._h, seed, spp, first_row, stripe_rows, stripe_period,
                                                None if off is None else off.ctypes.data, 0 if off is None else len(off), d_film_ptr))

    def develop_async(self, d_film_ptr, d_rgb_ptr, n_pixels):
        _check(_lib().dtof_develop_async(self._h, d_film_ptr, d_rgb_ptr, n_pixels))

    def collect(self, max_frames=4096):
        """wait for the frames enqueued by render_rows_async -> (summed stats dict, per-frame GPU milliseconds)"""
        st, ms, n = _Stats(), np.zeros(max_frames, np.float64), C.c_uint32(0)
        _check(_lib().dtof_async_collect(self._h, C.byref(st), ms.ctypes.data, max_frames, C.byref(n)))
        return st.as_dict(), ms[:min(n.value, max_frames)].copy()

    def render_stripes(self, d_film_ptr, seed, spp, first_row, stripe_rows, stripe_period, offsets=None):
        """Accumulate the rows of the stripes [first_row + k * stripe_period, ... + stripe_rows) (interleaved shard of one rank)."""
        st = _Stats()
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float32)
        _check(_lib().dtof_render_stripes(self._h, seed, spp, first_row, stripe_rows, stripe_period,
                                          None if off is None else off.ctypes.data, 0 if off is None else len(off), d_film_ptr, C.byref(st)))
        self.last_stats = st.as_dict()
        return self.last_stats

    def sample_lanes(self, seed, spp, lane_begin, n):
        out = np.zeros((n, 12), np.float32)
        valid = np.zeros(n, np.uint32)
        _check(_lib().dtof_sample_lanes_valid(self._h, seed, spp, lane_begin, n, out.ctypes.data, valid.ctypes.data))
        return {"sample_pos": out[:, 0:2], "time": out[:, 2], "ray_o": out[:, 3:6], "ray_d": out[:, 6:9], "rgb": out[:, 9:12], "valid": valid}

    def bsdf_eval(self, shape_index, queries):
        """BSDF::eval_pdf_sample of shape `shape_index` over an (n, 11) array of (wi, wo, sample1, sample2, uv) -> (n, 14): value[3], pdf, wo[3], pdf, eta, delta, weight[3], null"""
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, 11)
        out = np.zeros((len(q), 14), np.float32)
        _check(_lib().dtof_bsdf_eval(self._h, shape_index, len(q), q.ctypes.data, out.ctypes.data))
        return out

    def camera_rays(self, samples):
        """Sensor::sample_ray over an (n, 4) array of (position sample x, y in [0, 1]^2 of the crop window, aperture sample x, y) -> (origins, directions, maxt)"""
        s = np.ascontiguousarray(samples, np.float32).reshape(-1, 4)
        out = np.zeros((len(s), 7), np.float32)
        _check(_lib().dtof_camera_rays(self._h, len(s), s.ctypes.data, out.ctypes.data))
        return out[:, 0:3], out[:, 3:6], out[:, 6]

    def eval_modulation(self, mode, t, length=None):
        t = np.ascontiguousarray(t, np.float32)
        ln = np.ascontiguousarray(length, np.float32) if length is not None else None
        out = np.zeros_like(t)
        _check(_lib().dtof_eval_modulation(self._h, mode, t.ctypes.data, ln.ctypes.data if ln is not None else None,
                                           out.ctypes.data, t.size))
        return out

    @staticmethod
    def _rays(o, d, time, maxt):
        o, d = np.atleast_2d(np.asarray(o, np.float32)), np.atleast_2d(np.asarray(d, np.float32))
        n = max(len(o), len(d))
        rays = np.zeros((n, 8), np.float32)
        rays[:, 0:3], rays[:, 3:6], rays[:, 6] = o, d, time
        rays[:, 7] = np.finfo(np.float32).max if maxt is None else maxt
        return rays

    def ray_intersect(self, o, d, time=0.0, maxt=None):
        """Scene::ray_intersect over arrays of rays -> dict(t, p, n, sh_n, sh_s, sh_t, wi, ids, uv, prim_uv, prim_index, valid) (dtof_ray_intersect_uv)"""
        rays = self._rays(o, d, time, maxt)
        out, ids, uv = np.zeros((len(rays), 19), np.float32), np.zeros((len(rays), 3), np.int32), np.zeros((len(rays), 4), np.float32)
        _check(_lib().dtof_ray_intersect_uv(self._h, len(rays), rays.ctypes.data, out.ctypes.data, ids.ctypes.data, uv.ctypes.data))
        return {"t": out[:, 0], "p": out[:, 1:4], "n": out[:, 4:7], "sh_n": out[:, 7:10], "sh_s": out[:, 10:13], "sh_t": out[:, 13:16],
                "wi": out[:, 16:19], "ids": ids, "uv": uv[:, 0:2], "prim_uv": uv[:, 2:4], "prim_index": ids[:, 2], "valid": ids[:, 0] >= 0}

    def ray_test(self, o, d, time=0.0, maxt=None):
        """Scene::ray_test over arrays of rays -> bool array (dtof_ray_test)"""
        rays = self._rays(o, d, time, maxt)
        occ = np.zeros(len(rays), np.int32)
        _check(_lib().dtof_ray_test(self._h, len(rays), rays.ctypes.data, occ.ctypes.data))
        return occ != 0

    def cancel(self):
        _lib().dtof_cancel(self._h)


# DTOF_COMP_* of include/dtof.h
COMPONENTS = {"microfacet_eval": (0, 1), "microfacet_pdf": (1, 1), "microfacet_g1": (2, 1), "microfacet_sample": (3, 4), "fresnel": (4, 4),
              "fresnel_conductor": (5, 1), "rfilter": (6, 1), "warp_cosine_hemisphere": (7, 3), "warp_disk_concentric": (8, 2),
              "warp_uniform_triangle": (9, 2), "warp_uniform_sphere": (10, 3), "coordinate_system": (11, 6), "tea_float32": (12, 1), "math": (13, 1)}


def eval_component(name, inputs, params=()):
    """dtof_eval_component: one of the device functions the kernels are built from, over the rows of `inputs` (n x k float32)"""
    comp, n_out = COMPONENTS[name]
    x = np.ascontiguousarray(np.atleast_2d(np.asarray(inputs, np.float32)))
    p = np.ascontiguousarray(np.asarray(params, np.float32).reshape(-1))
    out = np.zeros((len(x), n_out), np.float32)
    _check(_lib().dtof_eval_component(comp, p.ctypes.data if p.size else None, p.size, x.ctypes.data, x.shape[1], out.ctypes.data, n_out, len(x)))
    return out


class Integrator:
    """What mi.load_dict({'type': 'dopplertofpath', ...}) returns: a plugin description that is bound to a
    scene at render time (the reference's integrator objects are scene-independent too)."""

    def __init__(self, props):
        self.props = dict(props)
        self._h = C.c_void_p()
        _check(_lib().dtof_integrator_create(*(_plugin_args(self.props) + (C.byref(self._h),))))   # constructor-time validation

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _LIB is not None:
            _LIB.dtof_integrator_destroy(self._h)
            self._h = C.c_void_p()

    def render(self, scene, seed=0, spp=0, sensor=0, offsets=None):
        if offsets is not None:
            scene.set_integrator(self.props)
            return scene.render(seed=seed, spp=spp, sensor=sensor, offsets=offsets)
        w, h = scene.size
        st, out = _Stats(), np.zeros((h, w, 4 if scene.info()["has_alpha"] else 3), np.float32)
        _check(_lib().dtof_integrator_render(self._h, None, scene._h, sensor, seed, spp, out.ctypes.data, C.byref(st)))
        scene.last_stats = st.as_dict()
        return out


def load_file(path, **params):
    h = C.c_void_p()
    names, values, n = _kv(params)
    _check(_lib().dtof_scene_load_file(os.fspath(path).encode(), names, values, n, C.byref(h)))
    return Scene(h)


def load_string(xml, **params):
    h = C.c_void_p()
    names, values, n = _kv(params)
    _check(_lib().dtof_scene_load_string(xml.encode(), names, values, n, C.byref(h)))
    return Scene(h)


def load_dict(d):
    t = d.get("type")
    if t in ("dopplertofpath", "path", "velocity"):
        return Integrator(d)
    raise DtofError('load_dict: unsupported plugin type "%s" (supported: dopplertofpath, path, velocity)' % t)


def render(scene, spp=0, seed=0, integrator=None, sensor=0):
    """mi.render(scene, spp=..., seed=..., integrator=...) (src/python/python/util.py)"""
    if integrator is not None:
        return integrator.render(scene, seed=seed, spp=spp, sensor=sensor)
    return scene.render(seed=seed, spp=spp, sensor=sensor)


def render_multi_pass(scene, integrator, total_spp, single_pass_spp=1024, show_progress=False):
    """doppler_tutorials/src/program_runner.py:11-31: mean of renders with seeds 0..n-1, each of
    min(single_pass_spp, total_spp) samples per pixel."""
    single = min(single_pass_spp, total_spp)
    n_pass = max(total_spp // single, 1)
    acc = None
    for i in range(n_pass):
        img = integrator.render(scene, seed=i, spp=single).astype(np.float32)
        acc = img if acc is None else acc + img
    return acc / np.float32(n_pass)


def to_tof_image(img, exposure_time=0.0015):
    """doppler_tutorials/src/utils/image_utils.py:20-31: luminance * exposure time"""
    img = np.asarray(img)
    return (0.2126 * img[..., 0] + 0.7152 * img[..., 1] + 0.0722 * img[..., 2]) * exposure_time


class Sampler:
    """Array-of-lanes `correlated` sampler living on the GPU (include/mitsuba/render/sampler.h:99-168)."""

    def __init__(self, sample_count=4, seed=0, time_correlate_number=2, path_correlate_number=None):
        self._h = C.c_void_p()
        pcn = time_correlate_number if path_correlate_number is None else path_correlate_number
        _check(_lib().dtof_sampler_create(sample_count, seed, time_correlate_number, pcn, C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                _lib().dtof_sampler_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def wavefront_size(self):
        return _lib().dtof_sampler_wavefront_size(self._h)

    def sample_count(self):
        return _lib().dtof_sampler_sample_count(self._h)

    def set_samples_per_wavefront(self, spw):
        _check(_lib().dtof_sampler_set_samples_per_wavefront(self._h, spw))

    def set_sample_count(self, spp):
        _check(_lib().dtof_sampler_set_sample_count(self._h, spp))

    def seeded(self):
        return bool(_lib().dtof_sampler_seeded(self._h))

    def _from_handle(self, fn):
        other = Sampler.__new__(Sampler)
        other._h = C.c_void_p()
        _check(fn(self._h, C.byref(other._h)))
        return other

    def fork(self):
        """same configuration, unseeded (src/samplers/correlated.cpp:25-32)"""
        return self._from_handle(_lib().dtof_sampler_fork)

    def clone(self):
        """same configuration and the same per-lane state (src/samplers/correlated.cpp:34-36)"""
        return self._from_handle(_lib().dtof_sampler_clone)

    def seed(self, seed, wavefront_size=0xffffffff):
        _check(_lib().dtof_sampler_seed(self._h, seed, wavefront_size))

    def advance(self):
        _check(_lib().dtof_sampler_advance(self._h))

    def _out(self, k=1):
        n = self.wavefront_size()
        return np.zeros((n, k) if k > 1 else n, np.float32)

    def next_1d(self):
        o = self._out()
        _check(_lib().dtof_sampler_next_1d(self._h, o.ctypes.data))
        return o

    def next_2d(self):
        o = self._out(2)
        _check(_lib().dtof_sampler_next_2d(self._h, o.ctypes.data))
        return o

    def _corr(self, correlate):
        if isinstance(correlate, (bool, int, np.bool_)):
            return None, int(bool(correlate)), None
        a = np.ascontiguousarray(correlate, dtype=np.uint8)
        return a.ctypes.data, 0, a

    def next_1d_correlate(self, correlate=False):
        o = self._out()
        p, allf, keep = self._corr(correlate)
        _check(_lib().dtof_sampler_next_1d_correlate(self._h, p, allf, o.ctypes.data))
        return o

    def next_2d_correlate(self, correlate=False):
        o = self._out(2)
        p, allf, keep = self._corr(correlate)
        _check(_lib().dtof_sampler_next_2d_correlate(self._h, p, allf, o.ctypes.data))
        return o

    def next_1d_time(self, strategy=ETimeSampling.UNIFORM, antithetic_shift=0.0, use_stratified_sampling_for_each_interval=False):
        o = self._out()
        _check(_lib().dtof_sampler_next_1d_time(self._h, int(strategy), float(antithetic_shift),
                                                int(bool(use_stratified_sampling_for_each_interval)), o.ctypes.data))
        return o

    def state(self):
        n = self.wavefront_size()
        o = np.zeros((n, 7), np.uint32)
        _check(_lib().dtof_sampler_get_state(self._h, o.ctypes.data))
        return o


# ------------------------------------------------------------------------------------------------ variant selection
# `import mitsuba as mi; mi.set_variant('cuda_rgb')` opens every tutorial script (program_runner.py:1-2).  This library has
# exactly one back end: RGB colour, float32 arithmetic, HIP kernels -- the counterpart of the reference's *_rgb variants.
_VARIANT = "hip_rgb"


def variants():
    return ["hip_rgb"]


def variant():
    return _VARIANT


def set_variant(*names):
    """Accepts the first usable of `names`; every scalar_/llvm_/cuda_ *_rgb variant of the reference maps onto hip_rgb
    (src/python/python/__init__.py: mi.set_variant).  Spectral, polarised, mono and double-precision variants do not exist here."""
    for n in names:
        if n == "hip_rgb" or (n.split("_", 1)[0] in ("scalar", "llvm", "cuda") and n.endswith("_rgb") and "_ad_" not in "_" + n.split("_", 1)[1] + "_"):
            return
    raise ImportError("Requested an unsupported variant \"%s\". The following variants are available: hip_rgb (the *_rgb variants of "
                      "the reference map onto it)." % ", ".join(names))
