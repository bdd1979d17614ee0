import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mitsuba3dopplertof_amd as mi
for scene, params, spp in (("domino.xml", dict(resx=1024, resy=1024, wave_function_type="rectangular"), 128), ("domino.xml", dict(resx=512, resy=512, wave_function_type="rectangular"), 32), ("cornell_wall.xml", dict(resx=256, resy=256), 16),
                           ("cornell_boxes.xml", dict(resx=256, resy=256), 16), ("open_veils.xml", dict(resx=128, resy=128), 16)):
    sc = mi.load_file(os.path.join(ROOT, "scenes", scene), **params)
    for i in range(3):
        t = time.time(); img = sc.render(seed=1, spp=spp); dt = time.time() - t
        inf = sc.info(); st = sc.last_stats
        print(scene, i, "choice", inf["inline_choice"], "survivors %.3f" % inf["survivors_after_first"], "ms_total %.3f" % st["ms_total"], "first %.3f" % st["ms_first"], "shade %.3f" % st["ms_shade"],
              "launches", st["n_launches_shade"], "inline iters", st["n_inline_iterations"], "bounces", st["n_bounces"], "checksum %.6f" % float(np.abs(img).sum()), flush=True)
    for forced in ("4", "1"):
        os.environ["DTOF_INLINE_ITERS"] = forced
        img = sc.render(seed=1, spp=spp); st = sc.last_stats
        print(scene, "forced", forced, "ms_total %.3f" % st["ms_total"], "first %.3f" % st["ms_first"], "shade %.3f" % st["ms_shade"], "launches", st["n_launches_shade"], "checksum %.6f" % float(np.abs(img).sum()), flush=True)
        del os.environ["DTOF_INLINE_ITERS"]
