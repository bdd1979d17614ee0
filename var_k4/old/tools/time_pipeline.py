#!/usr/bin/env python3
"""Split vs fused pipeline on the Cornell scenes with meshes / spheres at 512 x 512 x 64 (development helper; DTOF_PIPELINE=fused|split overrides the automatic choice)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mitsuba3dopplertof_amd as mi
for name in (sys.argv[1:] or ("cornell_boxes.xml", "cornell_area.xml", "cornell_spheres.xml", "cornell_rough.xml", "cornell_textured.xml", "cornell_wall.xml")):
    sc = mi.load_file(os.path.join(ROOT, "scenes", name), resx=512, resy=512)
    T = []
    for i in range(6):
        sc.render(seed=0, spp=64); s = sc.last_stats
        T.append((s["ms_total"], s["ms_shade"], s["ms_trace"], s["ms_shadow"], s["ms_splat"], s["ms_generate"]))
    T = np.array(T[2:]).min(0)
    print("%-8s %-24s total %8.3f ms  shade %7.3f trace %7.3f shadow %7.3f splat %6.3f gen %6.3f" % (os.environ.get("DTOF_PIPELINE", "auto"), name, *T))
