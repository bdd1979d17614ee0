import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mitsuba3dopplertof_amd as mi
sc = mi.load_file(os.path.join(ROOT, "scenes", "domino.xml"), resx=1024, resy=1024, wave_function_type="rectangular")
for i in range(2):
    img = sc.render(seed=1, spp=128)
print(sc.info()["inline_choice"], sc.last_stats)
