#!/usr/bin/env python3
"""A Domino field of n_side x n_side instances (the bench scene has 32 x 32 = 1 024: the largest TLAS the resident stage holds) -- what a frame costs beyond it (development helper).
   python tools/time_domino_n.py [n_side [res [spp]]]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_scenes
import mitsuba3dopplertof_amd as mi
n_side, res, spp = [int(x) for x in (sys.argv[1:4] + ["64", "1024", "32"][len(sys.argv) - 1:])]
d = tempfile.mkdtemp(prefix="dtof_domino_")
path = os.path.join(d, "s.xml")
open(path, "w").write(make_scenes.domino(n_side=n_side, res=res, spp=spp))
sc = mi.load_file(path)
info = sc.info()
best = None
for i in range(4):
    sc.render(seed=0, spp=spp); s = sc.last_stats
    if best is None or s["ms_total"] < best["ms_total"]: best = dict(s)
rays = best["n_bounces"] + best["n_shadow_rays"]
print("domino %d x %d: %d objects, %d nodes, blob %.2f MB | %dx%dx%d: total %.2f ms  first %.2f trace %.2f shade %.2f shadow %.2f | %.0f Mpaths/s  %.2f G rays/s  pipeline=%s" % (
    n_side, n_side, info["n_objects"], info["n_bvh_nodes"], info["scene_blob_bytes"] / 1e6, res, res, spp, best["ms_total"], best["ms_first"], best["ms_trace"], best["ms_shade"], best["ms_shadow"],
    best["n_paths"] / best["ms_total"] / 1e3, rays / best["ms_total"] / 1e6, os.environ.get("DTOF_PIPELINE", "auto")))
