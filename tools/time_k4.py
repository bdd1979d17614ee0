#!/usr/bin/env python3
"""Frame time of the K = 4 offset batch (four films in one traversal) on the scenes that take the every-BSDF kernels, 512 x 512 x 64 spp (development helper)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_scenes; make_scenes.ensure()
import mitsuba3dopplertof_amd as mi
for name in sys.argv[1:] or ["cornell_specular.xml", "cornell_rough.xml", "cornell_frosted.xml", "cornell_plastic.xml", "cornell_roughplastic.xml", "cornell_blend.xml", "cornell_env.xml",
                             "cornell_spot.xml", "cornell_sun.xml", "cornell_textured.xml", "cornell_boxes.xml"]:
    sc = mi.load_file(os.path.join(ROOT, "scenes", name), resx=512, resy=512)
    t = []
    for i in range(5):
        sc.render(seed=0, spp=64, offsets=[0.0, 0.25, 0.5, 0.75]); t.append(sc.last_stats["ms_total"])
    print("%-28s K = 4: %8.3f ms" % (name, min(t[1:])))
