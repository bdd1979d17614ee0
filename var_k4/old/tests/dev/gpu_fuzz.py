#!/usr/bin/env python3
"""Robustness fuzz (development helper, run through gpurun under `timeout`): numeric attribute values of the scene files are replaced
by extreme ones (0, -0, denormals, 1e30, overflow, nan, inf ...); every scene the loader accepts is rendered at 16x16x4.  The loader
must answer with DtofError or a scene, the renderer with an image (NaNs allowed) -- never a crash or a hang.
    python tests/dev/gpu_fuzz.py SEED COUNT [--no-render]"""
import os, random, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_scenes; make_scenes.ensure()
import numpy as np
import mitsuba3dopplertof_amd as mi

random.seed(int(sys.argv[1]))
render = "--no-render" not in sys.argv
names = ("cornell_boxes.xml", "cornell_wall.xml", "cornell_area.xml", "cornell_specular.xml", "cornell_roughplastic.xml", "cornell_sphere_light.xml",
         "cornell_rough.xml", "cornell_plastic.xml", "cornell_spheres.xml", "cornell_frosted.xml", "cornell_spot.xml", "cornell_disk.xml", "domino_small.xml")
texts = [open(os.path.join(ROOT, "scenes", n)).read() for n in names]
num = re.compile(r'-?\d+\.?\d*(?:e-?\d+)?')
vals = ['0', '-0', '1e-30', '1e30', '1e39', 'nan', 'inf', '-1', '4294967296', '1e-45', '0.5', '-1e39', '2', '1', '1e-8', '1e8', '3', '7']
ok = err = skipped = nonfinite = 0
t0 = time.time()
for it in range(int(sys.argv[2])):
    t = random.choice(texts)
    spans = [m.span() for m in num.finditer(t) if 'value=' in t[max(0, m.start() - 200):m.start()].split('<')[-1]]
    for a, b in sorted(random.sample(spans, random.randint(1, 3)), reverse=True):
        t = t[:a] + random.choice(vals) + t[b:]
    try:
        sc = mi.load_string(t, resx=16, resy=16)
    except mi.DtofError:
        err += 1
        continue
    info = sc.info()
    if not render or info["max_depth"] > 64 or info["film_width"] * info["film_height"] > 4096:
        skipped += 1
        continue
    try:
        img = sc.render(seed=it, spp=4)
        ok += 1
        nonfinite += int(not np.isfinite(img).all())
    except mi.DtofError:
        err += 1
print("rendered %d (non-finite images %d), rejected %d, skipped %d in %.1f s" % (ok, nonfinite, err, skipped, time.time() - t0))
