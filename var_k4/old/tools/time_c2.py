#!/usr/bin/env python3
"""Times the C2 workload (cornell_wall 512x512x64) a few times and prints the per-stage breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mitsuba3dopplertof_amd as mi
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")); import make_scenes; make_scenes.ensure()
S = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell_wall.xml"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sc = mi.load_file(os.path.join(S, scene))
best = None
for i in range(6):
    t = time.time(); img = sc.render(seed=0, spp=spp); dt = time.time() - t
    st = sc.last_stats
    if best is None or st["ms_total"] < best["ms_total"]: best = dict(st, wall=dt * 1e3)
loop = best["ms_trace"] + best["ms_shade"] + best["ms_shadow"]
print("batch=%s %s: total %.2f ms (wall %.2f) Mpaths/s %.0f | gen %.2f trace %.2f shade %.2f shadow %.2f splat %.2f | loop %.2f ms model %.0f GB/s | batches %d" % (
    os.environ.get("DTOF_BATCH_LANES", "default"), scene, best["ms_total"], best["wall"], best["n_paths"] / best["ms_total"] / 1e3,
    best["ms_generate"], best["ms_trace"], best["ms_shade"], best["ms_shadow"], best["ms_splat"], loop, best["n_bounces"] * 412 / loop / 1e6, best["n_batches"]))
