#!/usr/bin/env python3
"""Quick GPU-vs-oracle check + timing (development helper; the real tests are tests/ -m gpu)."""
import sys, time, json, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mitsuba3dopplertof_amd as mi
from oracle import orc

def lanes_check(xml, params, spp, seed=0, integ=None):
    sc = mi.load_file(xml, **params)
    if integ: sc.set_integrator(integ)
    osc = orc.Scene(xml, params)
    pd = osc.params(integrator=integ) if integ else osc.params()
    w, h = sc.size
    n = w * h * spp
    g = sc.sample_lanes(seed, spp, 0, n)
    o = osc.render_lanes(pd, seed, spp, 0, n, threads=os.cpu_count())
    res = {}
    for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
        a, b = g[k], o[k]
        eq = np.array_equal(a.view(np.uint32), np.ascontiguousarray(b).view(np.uint32))
        res[k] = (bool(eq), float(np.abs(a.astype(np.float64) - b).max()), int((a != b).sum()))
    bad = np.nonzero((g["rgb"] != o["rgb"]).any(axis=1))[0]
    print(os.path.basename(xml), params, "spp", spp, "lanes", n, res, "first bad lanes", bad[:8])
    t = time.time(); img = sc.render(seed=seed, spp=spp); dt = time.time() - t
    oimg, _ = osc.render(pd, seed=seed, spp=spp, threads=os.cpu_count())
    scale = np.abs(oimg).max()
    rel = np.abs(img - oimg).max() / max(scale, 1e-30)
    print("  image rel Linf (vs max|ref|):", rel, "stats", sc.last_stats)
    return res, rel

if __name__ == "__main__":
    print(mi._lib().dtof_version().decode())
    S = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scenes")
    lanes_check(S + "/cornell_boxes.xml", dict(resx=64, resy=64), 16)
    lanes_check(S + "/cornell_wall.xml", dict(resx=64, resy=64), 16)
    lanes_check(S + "/cornell_boxes.xml", dict(resx=48, resy=32, time_sampling_method="uniform"), 8)
    lanes_check(S + "/cornell_boxes.xml", dict(resx=32, resy=32, time_sampling_method="antithetic_mirror", antithetic_shift=0.0, wave_function_type="trapezoidal"), 16)
    lanes_check(S + "/domino_small.xml", dict(), 16)
    # timing: C2
    sc = mi.load_file(S + "/cornell_wall.xml")
    for i in range(3):
        t = time.time(); img = sc.render(seed=0, spp=64); dt = time.time() - t
        st = sc.last_stats
        loop = st["ms_trace"] + st["ms_shade"] + st["ms_shadow"]
        print("C2 512x512x64: wall %.1f ms, gpu total %.2f ms, Mpaths/s %.1f, loop %.2f ms, bytes-model GB/s %.1f" % (
            dt * 1e3, st["ms_total"], st["n_paths"] / st["ms_total"] / 1e3, loop, st["n_bounces"] * 412 / loop / 1e6), st)
