#!/usr/bin/env python3
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mitsuba3dopplertof_amd as mi
from oracle import orc
S = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scenes")
xml = S + "/domino.xml"
params = dict(resx=64, resy=64)
sc = mi.load_file(xml, **params); print(sc.info())
osc = orc.Scene(xml, params); pd = osc.params()
spp = 4; n = 64 * 64 * spp
t = time.time(); o = osc.render_lanes(pd, 0, spp, 0, n, threads=os.cpu_count()); print("oracle lanes %.1fs" % (time.time() - t))
g = sc.sample_lanes(0, spp, 0, n)
for k in ("sample_pos", "time", "ray_d", "rgb"):
    print(k, np.array_equal(g[k].view(np.uint32), np.ascontiguousarray(o[k]).view(np.uint32)), int((g[k] != o[k]).sum()))
full = mi.load_file(xml)
for i in range(3):
    t = time.time(); img = full.render(seed=0, spp=128); dt = time.time() - t
    st = full.last_stats; loop = st["ms_trace"] + st["ms_shade"] + st["ms_shadow"]
    print("domino 1024x1024x128: wall %.1f ms gpu %.1f ms Mpaths/s %.0f | gen %.1f trace %.1f shade %.1f shadow %.1f splat %.1f | bounces %d shadow %d batches %d model GB/s %.0f" % (
        dt * 1e3, st["ms_total"], st["n_paths"] / st["ms_total"] / 1e3, st["ms_generate"], st["ms_trace"], st["ms_shade"], st["ms_shadow"], st["ms_splat"],
        st["n_bounces"], st["n_shadow_rays"], st["n_batches"], st["n_bounces"] * 412 / loop / 1e6))
off = full.render(seed=0, spp=32, offsets=[0, .25, .5, .75]); st = full.last_stats
print("domino K=4 32spp: gpu %.1f ms Mpaths/s %.0f" % (st["ms_total"], st["n_paths"] / st["ms_total"] / 1e3), off.shape, np.isfinite(off).all())
