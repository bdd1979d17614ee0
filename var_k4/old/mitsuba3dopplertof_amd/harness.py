"""The experiment-driver contract of the reference's tutorials on top of the C ABI:
multi-pass seed averaging, the exact integrator dictionaries, ToF images and velocity maps.

Mirrors doppler_tutorials/src/program_runner.py:11-31 (mean over seeds 0..n-1 of min(1024,total)-spp passes), :33-80 (velocity /
radiance ground truth with the `velocity` / `path` integrators), :82-153 (Doppler render; `antithetic_shift` defaults to 0.5 for
"antithetic" and 0 otherwise) and doppler_tutorials/src/utils/image_utils.py:20-31,140-199 (luminance * exposure time; radial
velocity from the heterodyne / homodyne ratio)."""
import os

import numpy as np

from . import load_dict, load_file, render_multi_pass, to_tof_image   # noqa: F401
from .io import write_npy


def doppler_integrator_dict(wave_function_type="sinusoidal", low_frequency_component_only=True, hetero_frequency=1.0,
                            hetero_offset=0.0, time_sampling_method="antithetic", antithetic_shift=None,
                            path_correlation_depth=16, exposure_time=0.0015, w_g=30, max_depth=4,
                            use_stratified_sampling_for_each_interval=True):
    """The dictionary program_runner.py:127-141 hands to mi.load_dict."""
    if antithetic_shift is None:
        antithetic_shift = 0.5 if time_sampling_method == "antithetic" else 0.0
    return {"type": "dopplertofpath", "is_doppler_integrator": True, "max_depth": max_depth, "w_g": w_g, "time": exposure_time,
            "hetero_frequency": hetero_frequency, "hetero_offset": hetero_offset, "antithetic_shift": antithetic_shift,
            "time_sampling_method": time_sampling_method, "path_correlation_depth": path_correlation_depth,
            "low_frequency_component_only": low_frequency_component_only, "wave_function_type": wave_function_type,
            "use_stratified_sampling_for_each_interval": use_stratified_sampling_for_each_interval}


def _passes(total_spp):
    single = min(1024, total_spp)
    return single, max(total_spp // single, 1)


def run_scene_doppler_tof(scene, total_spp=1024, output_file=None, **integrator_kwargs):
    single, _ = _passes(total_spp)
    img = render_multi_pass(scene, load_dict(doppler_integrator_dict(**integrator_kwargs)), total_spp, single)
    if output_file:
        os.makedirs(os.path.dirname(os.path.abspath(output_file)), exist_ok=True)
        write_npy(output_file, img)
    return img


def run_scene_doppler_tof_offsets(scene, hetero_offsets, total_spp=1024, output_files=None, **integrator_kwargs):
    """The same as run_scene_doppler_tof for SEVERAL hetero_offset values of one otherwise identical setting: every traversal
    of the scene evaluates up to four modulation offsets at once (dtof_render_offsets; the paths do not depend on the offset,
    only the modulation weight does), so an 11-offset row of the experiment grids costs 3 traversals instead of 11.
    Returns the images in the order of `hetero_offsets`."""
    single, n_pass = _passes(total_spp)
    integrator_kwargs = dict(integrator_kwargs)
    integrator_kwargs.pop("hetero_offset", None)
    scene.set_integrator(doppler_integrator_dict(hetero_offset=0.0, **integrator_kwargs))
    images = []
    for g in range(0, len(hetero_offsets), 4):
        group = [float(o) for o in hetero_offsets[g:g + 4]]
        acc = None
        for i in range(n_pass):
            img = scene.render(seed=i, spp=single, offsets=group).astype(np.float32)
            acc = img if acc is None else acc + img
        images += list(acc / np.float32(n_pass))
    if output_files:
        for path, img in zip(output_files, images):
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            write_npy(path, img)
    return images


def run_scene_velocity(scene, total_spp=1024, output_file=None):
    single, _ = _passes(total_spp)
    img = render_multi_pass(scene, load_dict({"type": "velocity"}), total_spp, single)
    if output_file:
        write_npy(output_file, img)
    return img


def run_scene_radiance(scene, total_spp=1024, max_depth=4, output_file=None):
    single, _ = _passes(total_spp)
    img = render_multi_pass(scene, load_dict({"type": "path", "max_depth": max_depth}), total_spp, single)
    if output_file:
        write_npy(output_file, img)
    return img


def _velocity_from_ratio(ratio, exposure_time, w_g_mhz):
    """ratio = heterodyne / homodyne = dw T / (dw T - 1)  =>  dw = ratio / (T (ratio - 1)); v = -(c / 2) dw / w_g"""
    ratio = np.clip(ratio, -1.0, 0.999)
    delta_w = ratio * (1.0 / exposure_time) / (ratio - 1.0)
    return -(0.5 * delta_w * 3e8 / (w_g_mhz * 1e6))


def calc_velocity_from_homo_hetero(homodyne, heterodyne, exposure_time=0.0015, w_g=30):
    """image_utils.py:140-168 -- per-pixel ratio (0 where the homodyne image vanishes), then the closed form above."""
    homodyne, heterodyne = np.asarray(homodyne, np.float64), np.asarray(heterodyne, np.float64)
    ratio = np.divide(heterodyne, homodyne, out=np.zeros_like(homodyne), where=np.abs(homodyne) > 0)
    return _velocity_from_ratio(ratio, exposure_time, w_g)


def calc_velocity_from_homo_heteros(homodynes, heterodynes, exposure_time=0.0015, w_g=30):
    """image_utils.py:170-199 -- confidence-weighted (|homodyne| + 1e-5 T) mean of the ratios of several offset pairs."""
    num = den = 0.0
    for homodyne, heterodyne in zip(homodynes, heterodynes):
        homodyne, heterodyne = np.asarray(homodyne, np.float64), np.asarray(heterodyne, np.float64)
        ratio = np.divide(heterodyne, homodyne, out=np.zeros_like(homodyne), where=np.abs(homodyne) > 0)
        conf = np.abs(homodyne) + 1e-5 * 0.0015
        num, den = num + ratio * conf, den + conf
    return _velocity_from_ratio(num / den, exposure_time, w_g)
