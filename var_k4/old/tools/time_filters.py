#!/usr/bin/env python3
"""Frame time of the C2 workload under the reconstruction filters (development helper): tent (fast splat), gaussian / mitchell (generic splat)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mitsuba3dopplertof_amd as mi
text = open(os.path.join(ROOT, "scenes", "cornell_wall.xml")).read()
for name, rf in (("tent", '<rfilter type="tent" />'), ("gaussian", '<rfilter type="gaussian" />'), ("mitchell", '<rfilter type="mitchell" />'), ("box", '<rfilter type="box" />'), ("tent r=2", '<rfilter type="tent"><float name="radius" value="2"/></rfilter>')):
    a = text.index("<rfilter"); b = text.index("/>", a) + 2
    sc = mi.load_string(text[:a] + rf + text[b:])
    T = []
    for i in range(12):
        sc.render(seed=0, spp=0); s = sc.last_stats; T.append((s["ms_total"], s["ms_splat"]))
    T = np.array(T[3:]); print("%-10s total %.3f ms  splat %.3f ms" % (name, T[:, 0].min(), T[:, 1].min()))
