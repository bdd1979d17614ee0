#!/usr/bin/env python3
"""Generates the committed golden vectors from the CPU oracle (run from the repo root):

    python tests/golden/make_golden.py [--force]

The reference cannot be imported or built in the build container, so these vectors pin the
ORACLE's output (GPU-vs-oracle parity and oracle regression), not the reference's: for every parity
configuration of tests/conftest.py the developed image, the raw RGBW film and the first 512 lane records.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import CONFIGS, SCENES, GOLDEN   # noqa: E402
from oracle import orc                           # noqa: E402

force = "--force" in sys.argv
for name, xml, params, spp in CONFIGS:
    out = os.path.join(GOLDEN, name + ".npz")
    if os.path.exists(out) and not force:      # committed vectors are kept; --force regenerates all of them
        continue
    sc = orc.Scene(os.path.join(SCENES, xml), params)
    pd = sc.params()
    img, n = sc.render(pd, seed=3, spp=spp, threads=1)           # one thread: splat in lane order (bit-reproducible)
    film, _ = sc.render(pd, seed=3, spp=spp, threads=1, raw=True)
    lanes = sc.render_lanes(pd, 3, spp, 0, min(512, n), threads=1)
    np.savez_compressed(out, image=img, film=film,
                        lane_rgb=lanes["rgb"], lane_pos=lanes["sample_pos"], lane_time=lanes["time"],
                        lane_ray_o=lanes["ray_o"], lane_ray_d=lanes["ray_d"])
    print(name, img.shape, float(np.abs(img).max()))
