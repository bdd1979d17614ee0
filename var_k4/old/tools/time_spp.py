#!/usr/bin/env python3
"""Frame time of cornell_wall 512x512 at sample counts that are not powers of two / small, per filter (development helper)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mitsuba3dopplertof_amd as mi
text = open(os.path.join(ROOT, "scenes", "cornell_wall.xml")).read()
a = text.index("<rfilter"); b = text.index("/>", a) + 2
for name, rf in (("tent", '<rfilter type="tent" />'), ("gaussian", '<rfilter type="gaussian" />')):
    for spp in (4, 8, 12, 48, 64, 100):
        sc = mi.load_string(text[:a] + rf + text[b:], spp=spp)
        T = []
        for i in range(8):
            sc.render(seed=0, spp=spp); s = sc.last_stats; T.append((s["ms_total"], s["ms_splat"]))
        T = np.array(T[3:]); print("%-10s spp %3d  total %8.3f ms  splat %8.3f ms   (%.1f Mpaths/s)" % (name, spp, T[:, 0].min(), T[:, 1].min(), 512 * 512 * spp / T[:, 0].min() / 1e3))
