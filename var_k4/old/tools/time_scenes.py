#!/usr/bin/env python3
"""Frame time and stage breakdown of every generated scene at 512 x 512 x 64 spp (development helper: looks for slow paths)."""
import glob, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_scenes; make_scenes.ensure()
import mitsuba3dopplertof_amd as mi
for path in sorted(glob.glob(os.path.join(ROOT, "scenes", "*.xml"))):
    name = os.path.basename(path)
    if name.startswith("domino.xml"): continue
    try:
        sc = mi.load_file(path, resx=512, resy=512)
        T = []
        for i in range(6):
            sc.render(seed=0, spp=64); s = sc.last_stats
            T.append((s["ms_total"], s["ms_shade"], s["ms_trace"], s["ms_shadow"], s["ms_splat"], s["ms_generate"]))
        T = np.array(T[2:]).min(0)
        print("%-28s total %8.3f ms  shade %7.3f trace %7.3f shadow %7.3f splat %6.3f gen %6.3f  %7.0f Mpaths/s" % (name, *T, 512 * 512 * 64 / T[0] / 1e3))
    except Exception as e:
        print("%-28s %s" % (name, str(e)[:100]))
