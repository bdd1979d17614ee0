#!/usr/bin/env python3
"""Is the reference's own render (configs_example/scene.exr, decoded by tools/exr_piz.py into
tests/golden/reference_configs_example_scene_exr.npy) statistically one of OUR renders of the same scene?
Renders N seeds x 1024 spp on the GPU and places the reference among them for several summary statistics."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import mitsuba3dopplertof_amd as mi
ref = np.load(os.path.join(R, "tests/golden/reference_configs_example_scene_exr.npy")).astype(np.float64)
sc = mi.load_file(os.path.join(R, "scenes/cornell_boxes.xml"), resx=256, resy=256)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
imgs = np.stack([sc.render(seed=s, spp=1024).astype(np.float64) for s in range(N)])
mean, sd = imgs.mean(0), imgs.std(0, ddof=1)
def stats(a):
    return {"mean_R": a[..., 0].mean(), "mean_G": a[..., 1].mean(), "mean_B": a[..., 2].mean(),
            "tall_box_R": a[110:230, 60:130, 0].mean(), "short_box_R": a[180:245, 128:200, 0].mean(),
            "back_wall_R": a[20:100, 40:220, 0].mean(), "floor_R": a[246:256, 20:240, 0].mean(), "abs_mean": np.abs(a).mean()}
ours = [stats(i) for i in imgs]; r = stats(ref)
for k in r:
    v = np.array([o[k] for o in ours])
    print("%-12s ref % .4e  ours % .4e +- %.2e  -> z = % .2f" % (k, r[k], v.mean(), v.std(ddof=1), (r[k] - v.mean()) / v.std(ddof=1)))
z = (ref - mean) / np.sqrt(sd ** 2 * (1 + 1.0 / N) + (np.abs(ref) * 2 ** -11) ** 2 + 1e-30)
print("per-pixel z: mean %.3f std %.3f frac|z|<3 %.4f <4 %.4f <5 %.5f" % (z.mean(), z.std(), (abs(z) < 3).mean(), (abs(z) < 4).mean(), (abs(z) < 5).mean()))
zo = (imgs[0] - imgs[1:].mean(0)) / np.sqrt(imgs[1:].std(0, ddof=1) ** 2 * (1 + 1.0 / (N - 1)) + 1e-30)
print("control (our seed 0 vs the rest): mean %.3f std %.3f frac|z|<3 %.4f <4 %.4f" % (zo.mean(), zo.std(), (abs(zo) < 3).mean(), (abs(zo) < 4).mean()))
