"""ctypes glue around oracle/libdtof_oracle.so (TEST INFRASTRUCTURE ONLY -- see dtof_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import scene_xml

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

M16 = C.c_float * 16


class OrcTexture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("filter", C.c_int32), ("wrap", C.c_int32), ("channels", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("to_uv", C.c_float * 4), ("color0", C.c_float * 3), ("color1", C.c_float * 3), ("data", C.POINTER(C.c_float)),
                ("cond_cdf", C.POINTER(C.c_float)), ("marg_cdf", C.POINTER(C.c_float)), ("normalization", C.c_float), ("inv_normalization", C.c_float)]


class OrcShape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("twosided", C.c_int32), ("flip_normals", C.c_int32), ("face_normals", C.c_int32),
                ("reflectance", C.c_float * 3), ("to_world", M16), ("to_object", M16),
                ("n_vertices", C.c_int32), ("n_faces", C.c_int32),
                ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("texcoords", C.POINTER(C.c_float)), ("faces", C.POINTER(C.c_uint32)),
                ("emitter", C.c_int32), ("radiance", C.c_float * 3),
                ("area_pmf", C.POINTER(C.c_float)), ("area_cdf", C.POINTER(C.c_float)),
                ("area_sum", C.c_float), ("area_norm", C.c_float), ("area_lo", C.c_int32), ("area_hi", C.c_int32),
                ("center", C.c_float * 3), ("radius", C.c_float), ("sphere_inv_area", C.c_float),
                ("bsdf", C.c_int32), ("cond_eta", C.c_float * 3), ("cond_k", C.c_float * 3), ("spec_refl", C.c_float * 3),
                ("spec_trans", C.c_float * 3), ("diel_eta", C.c_float), ("nonlinear", C.c_int32),
                ("inv_eta_2", C.c_float), ("fdr_int", C.c_float), ("spec_sampling_weight", C.c_float),
                ("alpha_u", C.c_float), ("alpha_v", C.c_float), ("rough_table", C.POINTER(C.c_float)), ("tex_refl", C.POINTER(OrcTexture)), ("mf_type", C.c_int32), ("sample_all", C.c_int32),
                ("tex_spec", C.POINTER(OrcTexture)), ("tex_trans", C.POINTER(OrcTexture)), ("tex_alpha_u", C.POINTER(OrcTexture)), ("tex_alpha_v", C.POINTER(OrcTexture)),
                ("masked", C.c_int32), ("opacity", C.c_float), ("tex_opacity", C.POINTER(OrcTexture)), ("tex_normal", C.POINTER(OrcTexture)), ("bumpmap", C.c_int32), ("bump_scale", C.c_float), ("tex_radiance", C.POINTER(OrcTexture)),
                ("blend_other", C.c_void_p), ("blend_weight", C.c_float), ("tex_blend", C.POINTER(OrcTexture)), ("two_bsdfs", C.c_int32)]


class OrcGroup(C.Structure):
    _fields_ = [("first_shape", C.c_int32), ("n_shapes", C.c_int32)]


class OrcObject(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32), ("n_keys", C.c_int32),
                ("key_time", C.c_float * 2), ("key", (C.c_float * 16) * 2)]


class OrcEmitter(C.Structure):
    _fields_ = [("kind", C.c_int32), ("position", C.c_float * 3), ("intensity", C.c_float * 3), ("shape", C.c_int32),
                ("to_local", M16), ("cutoff_angle", C.c_float), ("cos_cutoff", C.c_float), ("cos_beam", C.c_float), ("inv_transition", C.c_float),
                ("bsphere", C.c_float * 4), ("envmap", C.c_void_p), ("env_to_world", M16)]


class OrcEnvmap(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("scale", C.c_float), ("data", C.POINTER(C.c_float)), ("n_levels", C.c_int32),
                ("level", C.POINTER(C.c_float) * 32), ("level_w", C.c_int32 * 32), ("level_size", C.c_int32 * 32),
                ("patch_size", C.c_float * 2), ("inv_patch_size", C.c_float * 2), ("max_patch", C.c_uint32 * 2)]


def envmap_export(em):
    """the tables of an OrcEmitter of kind 4 in the layout of the product's export kind 16"""
    e = C.cast(em.envmap, C.POINTER(OrcEnvmap)).contents
    out = [np.float32([e.w, e.h, e.n_levels, e.scale]), np.float32(list(em.bsphere)),
           np.float32(list(em.env_to_world)).reshape(4, 4)[:3].ravel(), np.float32(list(em.to_local)).reshape(4, 4)[:3].ravel(),
           np.ctypeslib.as_array(e.data, (e.w * e.h * 3,)).copy()]
    for k in range(e.n_levels):
        out += [np.float32([e.level_w[k], e.level_size[k]]), np.ctypeslib.as_array(e.level[k], (e.level_size[k],)).copy()]
    return np.concatenate(out)


class OrcSensor(C.Structure):
    _fields_ = [("to_world", M16), ("x_fov", C.c_float), ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("shutter_open", C.c_float), ("shutter_close", C.c_float),
                ("film_w", C.c_int32), ("film_h", C.c_int32),
                ("crop_x", C.c_int32), ("crop_y", C.c_int32), ("crop_w", C.c_int32), ("crop_h", C.c_int32),
                ("filter", C.c_int32), ("filter_radius", C.c_float), ("filter_stddev", C.c_float),
                ("filter_b", C.c_float), ("filter_c", C.c_float),
                ("kind", C.c_int32), ("aperture_radius", C.c_float), ("focus_distance", C.c_float)]


class OrcParams(C.Structure):
    _fields_ = [("time", C.c_float), ("w_g_mhz", C.c_float), ("g_1", C.c_float), ("g_0", C.c_float),
                ("w_s_mhz", C.c_float), ("phase_offset", C.c_float), ("hetero_frequency", C.c_float),
                ("wave_type", C.c_int32), ("low_frequency_component_only", C.c_int32),
                ("time_sampling", C.c_int32), ("antithetic_shift", C.c_float), ("stratify_each_interval", C.c_int32),
                ("path_correlation_depth", C.c_uint32), ("max_depth", C.c_uint32), ("rr_depth", C.c_uint32),
                ("hide_emitters", C.c_int32), ("base_seed", C.c_uint32),
                ("time_correlate_number", C.c_int32), ("path_correlate_number", C.c_int32), ("integrator", C.c_int32), ("sampler", C.c_int32), ("jitter", C.c_int32), ("samples_per_pass", C.c_uint32)]


class OrcScene(C.Structure):
    _fields_ = [("shapes", C.POINTER(OrcShape)), ("n_shapes", C.c_int32),
                ("groups", C.POINTER(OrcGroup)), ("n_groups", C.c_int32),
                ("objects", C.POINTER(OrcObject)), ("n_objects", C.c_int32),
                ("emitters", C.POINTER(OrcEmitter)), ("n_emitters", C.c_int32),
                ("sensor", OrcSensor)]


class OrcLane(C.Structure):
    _fields_ = [("sample_pos", C.c_float * 2), ("time", C.c_float), ("ray_o", C.c_float * 3), ("ray_d", C.c_float * 3),
                ("rgb", C.c_float * 3), ("path_length", C.c_float), ("depth", C.c_uint32), ("valid", C.c_uint32)]


LANE_DTYPE = np.dtype([("sample_pos", "<f4", 2), ("time", "<f4"), ("ray_o", "<f4", 3), ("ray_d", "<f4", 3),
                       ("rgb", "<f4", 3), ("path_length", "<f4"), ("depth", "<u4"), ("valid", "<u4")])


_ROUGH_CACHE = {}


def rough_plastic_tables(alpha, eta, mf_type=1):
    """(m_external_transmittance[64], m_internal_reflectance) of a roughplastic -- orc_roughplastic_tables"""
    key = (float(np.float32(alpha)), float(np.float32(eta)), int(mf_type))
    if key not in _ROUGH_CACHE:
        table, ir = np.zeros(64, np.float32), C.c_float()
        lib().orc_roughplastic_tables(int(mf_type), C.c_float(key[0]), C.c_float(key[1]), table.ctypes.data, C.byref(ir))
        _ROUGH_CACHE[key] = (table, np.float32(ir.value))
    return _ROUGH_CACHE[key]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, "libdtof_oracle.so")
    src = os.path.join(_HERE, "dtof_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src),
                                                                    os.path.getmtime(os.path.join(_HERE, "dtof_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdtof_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        # DTOF_ORACLE_LIB: another build of the same source (oracle/opcount.py: the block-counting build)
        L = C.CDLL(os.environ.get("DTOF_ORACLE_LIB") or build())
        L.orc_tea_float32.restype = C.c_float
        L.orc_tea_float32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.orc_tea32.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_pcg32_seed.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_pcg32_next_u32.restype = C.c_uint32
        L.orc_pcg32_next_u32.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
        L.orc_pcg32_next_f32.restype = C.c_float
        L.orc_pcg32_next_f32.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
        L.orc_permute_kensler.restype = C.c_uint32
        L.orc_permute_kensler.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_waveform.restype = C.c_float
        L.orc_waveform.argtypes = [C.c_float, C.c_int]
        L.orc_waveform_low_pass.restype = C.c_float
        L.orc_waveform_low_pass.argtypes = [C.c_float, C.c_int]
        L.orc_modulation_weight.restype = C.c_float
        L.orc_modulation_weight.argtypes = [C.POINTER(OrcParams), C.c_float, C.c_float]
        L.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_sampler_lane.argtypes = [C.POINTER(OrcParams), C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        L.orc_camera_ray.argtypes = [C.POINTER(OrcSensor), C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.orc_camera_sample_ray.argtypes = [C.POINTER(OrcSensor)] + [C.c_float] * 4 + [C.POINTER(C.c_float)]
        L.orc_intersect.restype = C.c_int
        L.orc_intersect.argtypes = [C.POINTER(OrcScene), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                                    C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.orc_occluded.restype = C.c_int
        L.orc_occluded.argtypes = [C.POINTER(OrcScene), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float]
        L.orc_render_lanes.argtypes = [C.POINTER(OrcScene), C.POINTER(OrcParams), C.c_uint32, C.c_uint32,
                                       C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        L.orc_render.restype = C.c_uint64
        L.orc_render.argtypes = [C.POINTER(OrcScene), C.POINTER(OrcParams), C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_int]
        L.orc_bake_cube.argtypes = [C.c_void_p] * 6
        L.orc_plastic_params.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_spot_params.argtypes = [C.c_float, C.c_float, C.c_void_p]
        L.orc_acos.restype = C.c_float
        L.orc_acos.argtypes = [C.c_float]
        L.orc_roughplastic_tables.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p, C.POINTER(C.c_float)]
        L.orc_gauss_legendre.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_fresnel_dielectric.argtypes = [C.c_float, C.c_float, C.c_void_p]
        L.orc_fresnel_conductor.restype = C.c_float
        L.orc_fresnel_conductor.argtypes = [C.c_float, C.c_float, C.c_float]
        L.orc_bake_sphere.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bake_cylinder.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mesh_area_table.restype = C.c_int
        L.orc_mesh_area_table.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bake_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                    C.c_void_p, C.c_void_p]
        L.orc_pass_layout.restype = C.c_int
        L.orc_pass_layout.argtypes = [C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        for nm in ("orc_expf", "orc_logf", "orc_tanf", "orc_erff", "orc_erfinvf"):
            getattr(L, nm).restype = C.c_float
            getattr(L, nm).argtypes = [C.c_float]
        L.orc_kat_microfacet.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_kat_filter.restype = C.c_float
        L.orc_kat_filter.argtypes = [C.c_int] + [C.c_float] * 5
        L.orc_kat_warp.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_kat_frame.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_kat_ray_intersect.restype = C.c_int
        L.orc_kat_ray_intersect.argtypes = [C.POINTER(OrcScene), C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_kat_bsdf.argtypes = [C.POINTER(OrcShape), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_kat_sphere_sample_direction.argtypes = [C.POINTER(OrcShape), C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_kat_shape_area.restype = C.c_float
        L.orc_kat_shape_area.argtypes = [C.POINTER(OrcShape)]
        L.orc_scene_bsphere.argtypes = [C.POINTER(OrcScene), C.c_void_p]
        L.orc_texture_eval.argtypes = [C.POINTER(OrcTexture), C.c_float, C.c_float, C.c_void_p]
        L.orc_kat_splat.argtypes = [C.POINTER(OrcSensor), C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_kat_solve_quadratic.restype = C.c_int
        L.orc_kat_solve_quadratic.argtypes = [C.c_double, C.c_double, C.c_double, C.c_void_p]
        _LIB = L
    return _LIB


def _m16(a):
    return M16(*np.asarray(a, dtype=np.float32).reshape(16).tolist())


def make_params(d):
    p = OrcParams()
    for name, _ in OrcParams._fields_:
        v = d.get(name, 0)
        setattr(p, name, v.item() if hasattr(v, "item") else v)
    return p


class Scene:
    """Oracle-side scene: FlatScene (scene_xml.load) marshalled into the C records."""

    def _fill_bsdf(self, o, s):
        """the BSDF fields of an OrcShape from a shape (or blend-partner) record of scene_xml"""
        L = lib()
        o.twosided = int(s["twosided"])
        o.reflectance = (C.c_float * 3)(*np.asarray(s["reflectance"], np.float32).tolist())
        o.bsdf = int(s.get("bsdf", 0))
        for key in ("cond_eta", "cond_k", "spec_refl", "spec_trans"):
            setattr(o, key, (C.c_float * 3)(*np.asarray(s.get(key, [0, 0, 0]), np.float32).tolist()))
        o.diel_eta = float(s.get("diel_eta", 1.0))
        o.alpha_u, o.alpha_v = float(s.get("alpha_u", 0.1)), float(s.get("alpha_v", 0.1))
        o.mf_type = int(s.get("mf_type", 1))
        o.sample_all = int(s.get("sample_all", 0))
        tex = s.get("tex_refl")
        if tex is not None:   # texture on the (diffuse) reflectance
            o.tex_refl = self._make_texture(tex)
        o.masked, o.opacity = int(s.get("masked", 0)), float(s.get("opacity", 1.0))   # the BSDF inside a `mask`
        o.bumpmap, o.bump_scale = int(s.get("bumpmap", 0)), float(s.get("bump_scale", 1.0))   # ... inside a `bumpmap` (tex_normal = the height texture)
        for key in ("tex_spec", "tex_trans", "tex_alpha_u", "tex_alpha_v", "tex_opacity", "tex_normal"):   # textures on the specular colours, the roughness, the mask's opacity
            if s.get(key) is not None:
                setattr(o, key, self._make_texture(s[key]))
        if o.bsdf == 3:   # plastic: SmoothPlastic::parameters_changed in C float32
            o.nonlinear = int(s.get("nonlinear", 0))
            out3 = (C.c_float * 3)()
            L.orc_plastic_params(C.c_float(o.diel_eta), o.reflectance, o.spec_refl, out3)
            o.inv_eta_2, o.fdr_int, o.spec_sampling_weight = out3[0], out3[1], out3[2]
            if tex is not None or s.get("spec_refl_mean") is not None:   # Texture::mean() of a textured slot is the texture's own mean (plastic.cpp:201-217)
                sp, d = np.asarray(s["spec_refl"], np.float32), np.asarray(s["reflectance"], np.float32)
                third = np.float32(1.0 / 3.0)
                s_mean = ((sp[0] + sp[1]) + sp[2]) * third if s.get("spec_refl_mean") is None else np.float32(s["spec_refl_mean"])
                d_mean = ((d[0] + d[1]) + d[2]) * third if tex is None else np.float32(tex["mean"])
                o.spec_sampling_weight = float(s_mean / (d_mean + s_mean))
            s["plastic_params"] = np.array([o.inv_eta_2, o.fdr_int, o.spec_sampling_weight], np.float32)
        if o.bsdf == 5:   # roughplastic: RoughPlastic::parameters_changed in C float32 (cached per (alpha, eta))
            o.nonlinear = int(s.get("nonlinear", 0))
            table, ir = rough_plastic_tables(o.alpha_u, o.diel_eta, o.mf_type)
            self._keep.append(table)
            o.rough_table = C.cast(table.ctypes.data, C.POINTER(C.c_float))
            eta = np.float32(o.diel_eta)
            o.inv_eta_2 = float(np.float32(1.0) / (eta * eta))
            d, sp = np.asarray(s["reflectance"], np.float32), np.asarray(s["spec_refl"], np.float32)
            third = np.float32(1.0 / 3.0)
            d_mean = ((d[0] + d[1]) + d[2]) * third if tex is None else np.float32(tex["mean"])
            s_mean = (((sp[0] + sp[1]) + sp[2]) * third if s.get("spec_refl_mean") is None else np.float32(s["spec_refl_mean"])) if s.get("has_spec_refl") else np.float32(1.0)
            o.fdr_int, o.spec_sampling_weight = float(ir), float(s_mean / (d_mean + s_mean))
            s["rough_table"], s["plastic_params"] = table, np.array([o.inv_eta_2, o.fdr_int, o.spec_sampling_weight], np.float32)
        if s.get("tex_radiance") is not None:   # textured radiance of the shape's area emitter: the texture with its sampling distribution
            o.tex_radiance = self._make_texture(s["tex_radiance"], distribution=True)
        if s.get("blend_other") is not None:   # blendbsdf: this record is bsdf_0, a second record carries bsdf_1
            other = OrcShape()
            self._fill_bsdf(other, s["blend_other"])
            self._keep.append(other)
            o.blend_other = C.addressof(other)
            o.blend_weight = float(s.get("blend_weight", 0.5))
            o.two_bsdfs = int(s.get("two_bsdfs", 0))
            if s.get("tex_blend") is not None:
                o.tex_blend = self._make_texture(s["tex_blend"])

    def _make_texture(self, tex, distribution=False):
        t = OrcTexture()
        if distribution and tex["data"] is not None:
            # DiscreteDistribution2D over the texels (distr_2d.h:92-117; BitmapTexture::rebuild_internals, bitmap.cpp:689-724): luminance of RGB texels in float32,
            # row-wise running sums and the running sum of the row totals accumulated in double, stored as float32
            d = np.asarray(tex["data"], np.float32).reshape(tex["height"], tex["width"], tex["channels"])
            imp = d[..., 0] if tex["channels"] == 1 else (d[..., 0] * np.float32(0.212671) + d[..., 1] * np.float32(0.715160)) + d[..., 2] * np.float32(0.072169)
            cond = np.cumsum(imp.astype(np.float64), axis=1)
            marg = np.cumsum(cond[:, -1])
            cond32, marg32 = np.ascontiguousarray(cond.astype(np.float32)), np.ascontiguousarray(marg.astype(np.float32))
            t.cond_cdf, t.marg_cdf = cond32.ctypes.data_as(C.POINTER(C.c_float)), marg32.ctypes.data_as(C.POINTER(C.c_float))
            t.inv_normalization, t.normalization = float(np.float32(marg[-1])), float(np.float32(1.0 / marg[-1]))
            self._keep += [cond32, marg32]
        t.kind, t.filter, t.wrap, t.channels, t.width, t.height = tex["kind"], tex["filter"], tex["wrap"], tex["channels"], tex["width"], tex["height"]
        t.to_uv = (C.c_float * 4)(*tex["to_uv"].tolist())
        t.color0, t.color1 = (C.c_float * 3)(*tex["color0"].tolist()), (C.c_float * 3)(*tex["color1"].tolist())
        if tex["data"] is not None:
            t.data = tex["data"].ctypes.data_as(C.POINTER(C.c_float))
        self._keep += [t, tex["data"]]
        return C.pointer(t)

    def __init__(self, source, params=None, is_string=False):
        self.flat = scene_xml.load(source, params, is_string)
        fs = self.flat
        self._keep = []
        self._envmaps = []
        L = lib()
        shapes = (OrcShape * max(1, len(fs.shapes)))()
        for i, s in enumerate(fs.shapes):
            o = shapes[i]
            o.kind, o.twosided, o.flip_normals, o.face_normals = s["kind"], s["twosided"], s["flip_normals"], s["face_normals"]
            o.reflectance = (C.c_float * 3)(*s["reflectance"].tolist())
            o.to_world, o.to_object = _m16(s["to_world"]), _m16(s["to_object"])
            o.emitter = int(s.get("emitter", 0))
            self._fill_bsdf(o, s)
            o.radiance = (C.c_float * 3)(*np.asarray(s.get("radiance", [0, 0, 0]), np.float32).tolist())
            if s["kind"] == 2:   # sphere: compose / decompose the transform in C float32 (orc_bake_sphere)
                tw = np.ascontiguousarray(s["to_world"], dtype=np.float32)
                to = np.ascontiguousarray(s["to_object"], dtype=np.float32)
                ctr = np.ascontiguousarray(s["sphere"]["center"], dtype=np.float32)
                comp, comp_inv, out8 = np.zeros(16, np.float32), np.zeros(16, np.float32), np.zeros(8, np.float32)
                L.orc_bake_sphere(tw.ctypes.data, to.ctypes.data, ctr.ctypes.data, C.c_float(float(s["sphere"]["radius"])),
                                  int(s["flip_normals"]), comp.ctypes.data, comp_inv.ctypes.data, out8.ctypes.data)
                s["to_world"], s["to_object"] = comp.reshape(4, 4), comp_inv.reshape(4, 4)
                s["sphere_baked"] = out8
                o.to_world, o.to_object = _m16(comp), _m16(comp_inv)
                o.center = (C.c_float * 3)(*out8[:3].tolist())
                o.radius, o.sphere_inv_area, o.flip_normals = float(out8[3]), float(out8[4]), int(out8[5])
            if s["kind"] == 4:   # cylinder: compose the transform in C float32 (orc_bake_cylinder)
                tw = np.ascontiguousarray(s["to_world"], dtype=np.float32)
                to = np.ascontiguousarray(s["to_object"], dtype=np.float32)
                cy = s["cylinder"]
                p0, p1 = np.ascontiguousarray(cy["p0"], np.float32), np.ascontiguousarray(cy["p1"], np.float32)
                comp, comp_inv, out8 = np.zeros(16, np.float32), np.zeros(16, np.float32), np.zeros(8, np.float32)
                L.orc_bake_cylinder(tw.ctypes.data, to.ctypes.data, p0.ctypes.data, p1.ctypes.data, C.c_float(float(cy["radius"])),
                                    int(s["flip_normals"]), comp.ctypes.data, comp_inv.ctypes.data, out8.ctypes.data)
                s["to_world"], s["to_object"] = comp.reshape(4, 4), comp_inv.reshape(4, 4)
                s["cylinder_baked"] = out8
                o.to_world, o.to_object = _m16(comp), _m16(comp_inv)
                o.radius, o.flip_normals = float(out8[0]), int(out8[3])
            if s["kind"] == 1 and s.get("mesh_raw") is not None:   # obj / ply: bake in C (orc_bake_mesh)
                raw = s["mesh_raw"]
                pin = np.ascontiguousarray(raw["positions"], dtype=np.float32).reshape(-1)
                faces = np.ascontiguousarray(raw["faces"], dtype=np.uint32).reshape(-1)
                nv, nf = pin.size // 3, faces.size // 3
                if nf and int(faces.max()) >= nv:
                    raise ValueError("mesh face references a vertex out of range")
                nin = None if raw["normals"] is None else np.ascontiguousarray(raw["normals"], dtype=np.float32).reshape(-1)
                pos = np.zeros(max(3 * nv, 1), np.float32)
                nrm = None if s["face_normals"] else np.zeros(max(3 * nv, 1), np.float32)
                tw = np.ascontiguousarray(s["to_world"], dtype=np.float32)
                to = np.ascontiguousarray(s["to_object"], dtype=np.float32)
                L.orc_bake_mesh(tw.ctypes.data, to.ctypes.data, nv, pin.ctypes.data, None if nin is None else nin.ctypes.data,
                                nf, faces.ctypes.data, int(s["face_normals"]), pos.ctypes.data, None if nrm is None else nrm.ctypes.data)
                uv = None if raw["texcoords"] is None else np.ascontiguousarray(raw["texcoords"], dtype=np.float32).reshape(-1)
                self._keep += [pos, nrm, uv, faces, pin, nin]
                s["positions"], s["normals"], s["texcoords"], s["faces"] = pos, nrm, uv, faces
                o.n_vertices, o.n_faces = nv, nf
                o.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
                if nrm is not None:
                    o.normals = nrm.ctypes.data_as(C.POINTER(C.c_float))
                if uv is not None:
                    o.texcoords = uv.ctypes.data_as(C.POINTER(C.c_float))
                o.faces = faces.ctypes.data_as(C.POINTER(C.c_uint32))
            elif s["kind"] == 1:   # cube
                pos, nrm = np.zeros(72, np.float32), np.zeros(72, np.float32)
                uv, faces = np.zeros(48, np.float32), np.zeros(36, np.uint32)
                tw = np.ascontiguousarray(s["to_world"], dtype=np.float32)
                to = np.ascontiguousarray(s["to_object"], dtype=np.float32)
                L.orc_bake_cube(tw.ctypes.data, to.ctypes.data, pos.ctypes.data, nrm.ctypes.data, uv.ctypes.data, faces.ctypes.data)
                self._keep += [pos, nrm, uv, faces]
                s["positions"], s["normals"], s["texcoords"], s["faces"] = pos, nrm, uv, faces
                o.n_vertices, o.n_faces = 24, 12
                o.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
                o.normals = nrm.ctypes.data_as(C.POINTER(C.c_float))
                o.texcoords = uv.ctypes.data_as(C.POINTER(C.c_float))
                o.faces = faces.ctypes.data_as(C.POINTER(C.c_uint32))
            if s["kind"] == 1 and o.emitter:   # Mesh::build_pmf (mesh.cpp:478-511)
                nf = int(o.n_faces)
                pmf, cdf = np.zeros(max(nf, 1), np.float32), np.zeros(max(nf, 1), np.float32)
                sm, nm, lo, hi = C.c_float(0), C.c_float(0), C.c_int32(0), C.c_int32(0)
                if L.orc_mesh_area_table(s["positions"].ctypes.data, nf, s["faces"].ctypes.data, pmf.ctypes.data, cdf.ctypes.data,
                                         C.byref(sm), C.byref(nm), C.byref(lo), C.byref(hi)) != 0:
                    raise ValueError("DiscreteDistribution: no probability mass found!")
                self._keep += [pmf, cdf]
                s["area_pmf"], s["area_cdf"] = pmf, cdf
                o.area_pmf, o.area_cdf = pmf.ctypes.data_as(C.POINTER(C.c_float)), cdf.ctypes.data_as(C.POINTER(C.c_float))
                o.area_sum, o.area_norm, o.area_lo, o.area_hi = sm.value, nm.value, lo.value, hi.value
        groups = (OrcGroup * max(1, len(fs.groups)))()
        for i, g in enumerate(fs.groups):
            groups[i].first_shape, groups[i].n_shapes = g["first_shape"], g["n_shapes"]
        objects = (OrcObject * max(1, len(fs.objects)))()
        for i, ob in enumerate(fs.objects):
            o = objects[i]
            o.kind, o.index, o.n_keys = ob["kind"], ob["index"], ob["n_keys"]
            o.key_time = (C.c_float * 2)(*np.asarray(ob["key_time"], np.float32).tolist())
            o.key[0], o.key[1] = _m16(ob["key"][0]), _m16(ob["key"][1])
        emitters = (OrcEmitter * max(1, len(fs.emitters)))()
        for i, e in enumerate(fs.emitters):
            emitters[i].kind = e["kind"]
            emitters[i].position = (C.c_float * 3)(*e["position"].tolist())
            emitters[i].intensity = (C.c_float * 3)(*e["intensity"].tolist())
            emitters[i].shape = int(e.get("shape", -1))
            if e["kind"] == 2:   # spot: constructor constants in C float32 (orc_spot_params)
                emitters[i].to_local = _m16(e["to_local"])
                out4 = (C.c_float * 4)()
                L.orc_spot_params(C.c_float(float(e["cutoff_deg"])), C.c_float(float(e["beam_deg"])), out4)
                emitters[i].cutoff_angle, emitters[i].cos_cutoff, emitters[i].cos_beam, emitters[i].inv_transition = out4[0], out4[1], out4[2], out4[3]
                e["spot_params"] = np.array(list(out4), np.float32)
        sc = OrcScene()
        sc.shapes, sc.n_shapes = shapes, len(fs.shapes)
        sc.groups, sc.n_groups = groups, len(fs.groups)
        sc.objects, sc.n_objects = objects, len(fs.objects)
        sc.emitters, sc.n_emitters = emitters, len(fs.emitters)
        se = fs.sensor
        if se is None:     # a scene without a sensor loads (as in the reference); there is nothing to render
            self.c = sc
            return
        sc.sensor.to_world = _m16(se["to_world"])
        for k in ("x_fov", "near_clip", "far_clip", "shutter_open", "shutter_close", "filter_radius", "filter_stddev", "filter_b", "filter_c"):
            setattr(sc.sensor, k, float(se[k]))
        sc.sensor.kind, sc.sensor.aperture_radius, sc.sensor.focus_distance = int(se.get("kind", 0)), float(se.get("aperture_radius", 0.0)), float(se.get("focus_distance", 0.0))
        for k in ("film_w", "film_h", "crop_x", "crop_y", "crop_w", "crop_h", "filter"):
            setattr(sc.sensor, k, int(se[k]))
        self._keep += [shapes, groups, objects, emitters]
        self.c = sc
        for i, e in enumerate(fs.emitters):
            if e["kind"] == 4:   # envmap: the tables of the constructor (envmap.cpp:130-224), built by the C side
                img = np.ascontiguousarray(e["image"], np.float32)
                L.orc_envmap_create2.restype = C.c_void_p
                L.orc_envmap_create2.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32]
                h = L.orc_envmap_create2(img.ctypes.data, img.shape[1], img.shape[0], C.c_float(float(e["scale"])), int(bool(e.get("mis_compensation", False))))
                emitters[i].envmap = h
                emitters[i].env_to_world = _m16(e["to_world"]); emitters[i].to_local = _m16(e["to_local"])
                self._envmaps.append(h)
            if e["kind"] in (3, 4, 5):   # environment / directional: ConstantBackgroundEmitter / EnvironmentMapEmitter::set_scene (constant.cpp:73-83, envmap.cpp:286-297)
                bs = (C.c_float * 4)()
                L.orc_scene_bsphere(C.byref(sc), bs)
                emitters[i].bsphere = bs
                e["bsphere"] = np.array(list(bs), np.float32)

    @property
    def size(self):
        return self.flat.sensor["crop_w"], self.flat.sensor["crop_h"]

    def params(self, integrator=None, sampler=None):
        """integrator/sampler: dict overriding the XML's plugin (mirrors mi.load_dict({...}))."""
        d = scene_xml.integrator_params(integrator if integrator is not None else self.flat.integrator,
                                        sampler if sampler is not None else self.flat.sampler)
        return d

    def render_lanes(self, pd, seed, spp, lane_begin, n, threads=1):
        out = np.zeros(n, dtype=LANE_DTYPE)
        p = make_params(pd)
        lib().orc_render_lanes(C.byref(self.c), C.byref(p), seed, spp, lane_begin, n, out.ctypes.data, threads)
        return out

    def render(self, pd, seed=0, spp=None, rows=None, threads=1, raw=False):
        spp = spp or pd["sample_count"]
        w, h = self.size
        film = np.zeros((h, w, 4), np.float32)
        img = np.zeros((h, w, 3), np.float32)
        r0, r1 = rows if rows else (0, h)
        p = make_params(pd)
        n = lib().orc_render(C.byref(self.c), C.byref(p), seed, spp, r0, r1, film.ctypes.data, img.ctypes.data, threads)
        return (film if raw else img), n

    def render_alpha(self, pd, seed=0, spp=None, threads=1):
        """the alpha channel of an rgba film (orc_render_alpha): weighted mean of the integrator's valid_ray flag"""
        spp = spp or pd["sample_count"]
        w, h = self.size
        alpha = np.zeros((h, w), np.float32)
        p = make_params(pd)
        L = lib()
        L.orc_render_alpha.restype = C.c_uint64
        L.orc_render_alpha.argtypes = [C.POINTER(OrcScene), C.POINTER(OrcParams), C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
        L.orc_render_alpha(C.byref(self.c), C.byref(p), seed, spp, alpha.ctypes.data, threads)
        return alpha

    def render_exact(self, pd, seed=0, spp=None, rows=None, threads=1):
        """the developed image with the splat terms summed in float64 (orc_render_exact): the order-independent value of the film"""
        spp = spp or pd["sample_count"]
        w, h = self.size
        film = np.zeros((h, w, 4), np.float64)
        img = np.zeros((h, w, 3), np.float32)
        r0, r1 = rows if rows else (0, h)
        p = make_params(pd)
        L = lib()
        L.orc_render_exact.restype = C.c_uint64
        L.orc_render_exact.argtypes = [C.POINTER(OrcScene), C.POINTER(OrcParams), C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int]
        n = L.orc_render_exact(C.byref(self.c), C.byref(p), seed, spp, r0, r1, film.ctypes.data, img.ctypes.data, threads)
        return img, n
