"""Error tables and plots over the experiment grids written by `experiments.py`.

    python -m mitsuba3dopplertof_amd.analysis --expnumber 1 --basedir RUN_DIR [--scene_names cornell-box ...] [--grid 11] [--no_plots]

Counterpart of doppler_tutorials/src/main_plot.py (pure numpy + csv; matplotlib only for the figures, scipy only for their smoothing):
  * `export_error` (main_plot.py:20-76): for every (hetero_frequency, hetero_offset) of the grid and every experiment name, the error of
        <base_dir>/<scene>/<wave>/freq_%.3f_offset_%.3f/<expname>.npy   against   <reference_base_dir>/.../reference.npy
    (both scaled by the exposure time, as the reference does) -> `<output_base_dir>/<scene>/<wave>/result.csv` with the columns
    freq, offset, expname, MAE, RMSE, PSNR, RelativeMAE, RelativeRMSE, SNR.  PSNR is skimage.metrics.peak_signal_noise_ratio written out:
    10 log10(data_range^2 / MSE) with data_range = max - min of the reference image (computed in float64, as skimage does).
  * `error_curves` (main_plot.py:85-104): mean and sample standard deviation (pandas' ddof = 1) of one error over the other grid axis.
  * `show_image` (main_show_image.py:13-70): luminance images over the heterodyne frequency with their relative RMSE.
  * `plot_experiment(1 | 2 | 3, ...)` (main_plot.py:213-552): the experiment names of the three figures, their result tables, and -- when matplotlib
    is importable -- `plot_total.png`: error against the heterodyne frequency, one line per experiment name, +- one standard deviation.
"""
import argparse
import csv
import os

import numpy as np

COLUMNS = ("freq", "offset", "expname", "MAE", "RMSE", "PSNR", "RelativeMAE", "RelativeRMSE", "SNR")


def image_errors(image, reference):
    """the six error measures of main_plot.py:55-61 for one image pair (already scaled by the exposure time)"""
    image, reference = np.asarray(image), np.asarray(reference)
    diff = image - reference
    mae = float(np.mean(np.abs(diff)))
    rmse = float(np.sqrt(np.mean(diff ** 2)))
    ref_mean = float(np.mean(np.abs(reference)))
    # skimage.metrics.peak_signal_noise_ratio: float64 images, mean_squared_error, 10 * log10(data_range ** 2 / err)
    mse = float(np.mean((reference.astype(np.float64) - image.astype(np.float64)) ** 2))
    data_range = float(reference.max() - reference.min())
    with np.errstate(divide="ignore"):
        psnr = float(10 * np.log10(np.float64(data_range ** 2) / np.float64(mse)))
        rel_mae, rel_rmse = mae / ref_mean, rmse / ref_mean
        snr = float(-10 * np.log10(np.float64(rel_rmse)))
    return dict(MAE=mae, RMSE=rmse, PSNR=psnr, RelativeMAE=rel_mae, RelativeRMSE=rel_rmse, SNR=snr)


def export_error(base_dir, scene_name, expnames, output_base_dir, reference_base_dir=None, N_heterodyne_frequencies=10, N_heterodyne_offsets=10,
                 exposure_time=0.0015, exit_if_file_exists=False):
    """main_plot.py:20-76.  `scene_name` is "<scene>/<wave_function_type>" as the figures pass it.  Returns the rows written (or read back)."""
    out_dir = os.path.join(output_base_dir, scene_name)
    out_csv = os.path.join(out_dir, "result.csv")
    if os.path.exists(out_csv) and exit_if_file_exists:
        return read_result(out_csv)
    ref_root = os.path.join(reference_base_dir if reference_base_dir is not None else base_dir, scene_name)
    root = os.path.join(base_dir, scene_name)
    rows = []
    for freq in np.linspace(0.0, 1.0, N_heterodyne_frequencies + 1):
        for offset in np.linspace(0.0, 1.0, N_heterodyne_offsets + 1):
            cell = "freq_%.3f_offset_%.3f" % (freq, offset)
            reference = np.load(os.path.join(ref_root, cell, "reference.npy")) * exposure_time
            for expname in expnames:
                image = np.load(os.path.join(root, cell, "%s.npy" % expname)) * exposure_time
                rows.append(dict(freq=float(freq), offset=float(offset), expname=expname, **image_errors(image, reference)))
    os.makedirs(out_dir, exist_ok=True)
    with open(out_csv, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=COLUMNS)
        w.writeheader()
        for r in rows:
            w.writerow({k: (r[k] if k == "expname" else repr(float(r[k]))) for k in COLUMNS})
    return rows


def read_result(path):
    with open(path, newline="") as f:
        return [{k: (v if k == "expname" else float(v)) for k, v in r.items()} for r in csv.DictReader(f)]


def error_curves(rows, expnames, target="freq", error_type="RMSE", other_value="mean"):
    """main_plot.py:85-104: per experiment name the error against `target` ("freq" | "offset"): the mean over the other axis with its sample standard
    deviation (other_value="mean"), or the slice at one value of the other axis (std = 0).  -> {expname: (x, y, std)}"""
    other = "offset" if target == "freq" else "freq"
    out = {}
    for name in expnames:
        mine = [r for r in rows if r["expname"] == name]
        xs = sorted({r[target] for r in mine})
        ys, sd = [], []
        for x in xs:
            v = np.array([r[error_type] for r in mine if r[target] == x and (other_value == "mean" or np.isclose(r[other], other_value))], np.float64)
            ys.append(v.mean() if v.size else np.nan)
            sd.append(v.std(ddof=1) if (other_value == "mean" and v.size > 1) else 0.0)
        out[name] = (np.array(xs), np.array(ys), np.array(sd))
    return out


def experiment_expnames(expnumber, time_sampling_method="antithetic", grid=11):
    """the experiment names a figure compares and its results family / output folder (main_plot.py:220-252, 403-440, 499-506, 567-606)"""
    methods = ["uniform", "stratified", "antithetic", "antithetic_mirror"]
    if expnumber == 1:
        return ("time_spatial_sampling_comparison", "time_spatial_sampling_comparison_full_plot",
                ["%s_path_corr_depth_%d" % (t, s) for t in methods for s in (0, 16)])
    if expnumber == 2:
        names = []
        for t in methods:
            names.append("%s_path_corr_depth_16" % t)
            if t != "uniform":
                names.append("%s_path_corr_depth_16_no_further_stratification" % t)
        return "time_spatial_sampling_comparison", "further_stratificaion_comparison_plot", names      # (sic: the reference's folder name)
    if expnumber == 3:
        return ("antithetic_shift_comparison", "antithetic_shift_comparison_plot",
                ["%s_shift_%.1f" % (time_sampling_method, a) for a in np.linspace(0.0, 1.0, grid)])
    raise ValueError("expnumber must be 1, 2 or 3")


_COLORS = {"uniform": "k", "stratified": "r", "antithetic_mirror": "b", "antithetic": "g"}


def _style(expname):
    for m in ("antithetic_mirror", "antithetic", "stratified", "uniform"):
        if expname.startswith(m):
            dotted = expname.endswith("_no_further_stratification") or "_path_corr_depth_0" in expname
            return _COLORS[m], (":" if expname.endswith("_no_further_stratification") else "-." if dotted else "-")
    return "k", "-"


def plot_experiment(expnumber, basedir, scene_names=("cornell-box",), wave_function_types=("sinusoidal",), time_sampling_method="antithetic", grid=11,
                    error_types=("RMSE", "PSNR"), exposure_time=0.0015, make_plots=True, exit_if_file_exists=False, log=print):
    """result.csv for every scene / wave function of one figure, then (matplotlib permitting) plot_total.png.  Returns {scene/wave: rows}."""
    family, out_family, expnames = experiment_expnames(expnumber, time_sampling_method, grid)
    base_dir = os.path.join(basedir, "results", family)
    reference_base_dir = os.path.join(basedir, "results", "gt_images")
    output_base_dir = os.path.join(basedir, "results", out_family)
    tables = {}
    for s in scene_names:
        for wave in wave_function_types:
            key = "%s/%s" % (s, wave)
            tables[key] = export_error(base_dir, key, expnames, output_base_dir, reference_base_dir, grid - 1, grid - 1, exposure_time, exit_if_file_exists)
            log("%s: %d rows -> %s" % (key, len(tables[key]), os.path.join(output_base_dir, key, "result.csv")))
    if not make_plots:
        return tables
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        log("matplotlib is not installed: tables only")
        return tables
    keys = list(tables)
    fig, axis = plt.subplots(len(error_types), len(keys), figsize=(5 * len(keys), 4 * len(error_types)), squeeze=False)
    for i, key in enumerate(keys):
        for j, et in enumerate(error_types):
            ax = axis[j][i]
            for name, (x, y, sd) in error_curves(tables[key], expnames, "freq", et).items():
                color, ls = _style(name)
                if expnumber == 3:    # eleven shifts of one method: a colour ramp instead of the method colours
                    color, ls = plt.cm.viridis(expnames.index(name) / max(len(expnames) - 1, 1)), "-"
                xs, ys, ss = _smooth(x, y, sd)
                ax.plot(xs, ys, color=color, linestyle=ls, linewidth=2, label=name.replace("_", " "))
                ax.fill_between(xs, ys - ss, ys + ss, facecolor=color, alpha=0.2)
            ax.set_xlim(0.0, 1.0)
            ax.set_xlabel(r"$\omega_r$"); ax.set_ylabel(et); ax.set_title(key)
            if "Relative" in et:
                ax.set_yscale("log")
    axis[0][0].legend(fontsize=7)
    fig.tight_layout()
    out = os.path.join(output_base_dir, "plot_total.png" if expnumber != 3 else "plot_total_%s.png" % time_sampling_method)
    fig.savefig(out, dpi=150)
    plt.close(fig)
    log("wrote " + out)
    return tables


def _smooth(x, y, sd, n=100):
    """cubic B-spline through the grid points (main_plot.py:78-83,123-131); needs four points and scipy, otherwise the polyline"""
    try:
        from scipy.interpolate import make_interp_spline
    except ImportError:
        return x, y, sd
    if len(x) < 4 or not (np.isfinite(y).all() and np.isfinite(sd).all()):
        return x, y, sd
    xn = np.linspace(x.min(), x.max(), n)
    return xn, make_interp_spline(x, y, k=3)(xn), make_interp_spline(x, sd, k=3)(xn)


def show_image(expnames, base_dir, scene_name, output_base_dir, reference_base_dir=None, exposure_time=0.0015, grid=11, make_plots=True, log=print):
    """doppler_tutorials/src/main_show_image.py:13-70: the luminance images of every experiment name over the heterodyne frequencies at offset 0, each
    titled with its relative RMSE against the reference image (RMSE / RMS of the reference), colour range = the 10th .. 90th percentile of a column.
    Returns {expname: [relative RMSE per frequency]} and writes `<output_base_dir>/<scene_name>/image_over_w_r.png` when matplotlib is importable."""
    from .io import rgb2luminance
    freqs = np.linspace(0.0, 1.0, grid)
    root = os.path.join(base_dir, scene_name)
    ref_root = os.path.join(reference_base_dir if reference_base_dir is not None else base_dir, scene_name)
    rel = {n: [] for n in expnames}
    columns = []
    for f in freqs:
        cell = "freq_%.3f_offset_%.3f" % (f, 0.0)
        reference = rgb2luminance(np.load(os.path.join(ref_root, cell, "reference.npy")) * exposure_time)
        imgs = np.asarray([rgb2luminance(np.load(os.path.join(root, cell, "%s.npy" % n)) * exposure_time) for n in expnames])
        for n, im in zip(expnames, imgs):
            rel[n].append(float(np.sqrt(np.mean((im - reference) ** 2)) / np.sqrt(np.mean(reference ** 2))))
        columns.append((imgs, float(np.percentile(imgs, 10)), float(np.percentile(imgs, 90))))
    if make_plots:
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except ImportError:
            log("matplotlib is not installed: numbers only")
            return rel
        fig, axis = plt.subplots(len(expnames), len(freqs), figsize=(2 * len(freqs), 2 * len(expnames)), squeeze=False)
        for i, (imgs, vmin, vmax) in enumerate(columns):
            for j, n in enumerate(expnames):
                ax = axis[j][i]
                ax.imshow(imgs[j], vmin=vmin, vmax=vmax)
                ax.set_xticks([]); ax.set_yticks([]); ax.set_title("%.4f" % rel[n][i], fontsize=8)
        out_dir = os.path.join(output_base_dir, scene_name)
        os.makedirs(out_dir, exist_ok=True)
        fig.tight_layout(); fig.subplots_adjust(wspace=0.05, hspace=0.25)
        fig.savefig(os.path.join(out_dir, "image_over_w_r.png"), dpi=100)
        plt.close(fig)
        log("wrote " + os.path.join(out_dir, "image_over_w_r.png"))
    return rel


def main(argv=None):
    ap = argparse.ArgumentParser(description="error tables / plots of the experiment grids (doppler_tutorials/src/main_plot.py, main_show_image.py)")
    ap.add_argument("--show_images", action="store_true", help="the image strip of main_show_image.py instead of the error figures")
    ap.add_argument("--expnumber", type=int, default=1)
    ap.add_argument("--basedir", type=str, default="../")
    ap.add_argument("--scene_names", nargs="*", default=None)
    ap.add_argument("--wave_function_types", nargs="*", default=None)
    ap.add_argument("--grid", type=int, default=11)
    ap.add_argument("--no_plots", action="store_true")
    a = ap.parse_args(argv)
    # the reference's figures: 1 = six scenes, sinusoidal; 2 = cornell-box under the four wave functions; 3 = cornell-box, both antithetic methods
    scenes = a.scene_names or (["cornell-box", "bedroom", "kitchen", "living-room-2", "soccer-ball", "veach-ajar"] if a.expnumber == 1 else ["cornell-box"])
    waves = a.wave_function_types or (["sinusoidal", "rectangular", "triangular", "trapezoidal"] if a.expnumber == 2 else ["sinusoidal"])
    if a.show_images:   # main_show_image.py:72-95: four sampling methods of the first scene, sinusoidal
        names = ["%s_path_corr_depth_%d" % ts for ts in (("uniform", 0), ("stratified", 16), ("antithetic", 16), ("antithetic_mirror", 16))]
        show_image(names, os.path.join(a.basedir, "results", "time_spatial_sampling_comparison"), "%s/%s" % (scenes[0], waves[0]),
                   os.path.join(a.basedir, "results", "images_over_hetero_frequency"), os.path.join(a.basedir, "results", "gt_images"), grid=a.grid, make_plots=not a.no_plots)
        return 0
    for method in (("antithetic", "antithetic_mirror") if a.expnumber == 3 else ("antithetic",)):
        plot_experiment(a.expnumber, a.basedir, scenes, waves, method, a.grid, make_plots=not a.no_plots, exit_if_file_exists=a.expnumber == 1)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
