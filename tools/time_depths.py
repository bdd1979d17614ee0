#!/usr/bin/env python3
"""What the later bounce iterations of a fused frame cost: the same scene rendered with max_depth = 2, 3, 4 ... (development helper).
   python tools/time_depths.py [scene.xml [spp [max_depths ...]]]      default: domino.xml 128 2 3 4"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_scenes; make_scenes.ensure()
import mitsuba3dopplertof_amd as mi
scene = sys.argv[1] if len(sys.argv) > 1 else "domino.xml"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
depths = [int(x) for x in sys.argv[3:]] or [2, 3, 4]
path = scene if os.path.exists(scene) else os.path.join(ROOT, "scenes", scene)
for md in depths:
    sc = mi.load_file(path, max_depth=md, wave_function_type="rectangular")
    best = None
    for i in range(4):
        sc.render(seed=0, spp=spp); s = sc.last_stats
        if best is None or s["ms_total"] < best["ms_total"]: best = dict(s)
    print("max_depth %d: total %.3f ms  shade %.3f  | paths %d  closest-hit rays %d  shadow rays %d" % (md, best["ms_total"], best["ms_shade"], best["n_paths"], best["n_bounces"], best["n_shadow_rays"]), flush=True)
