"""the GPU parity tests against a build of the kernels in which every automatic variable without an initialiser starts as a NaN / 0xAA pattern
(`make -C mitsuba3dopplertof_amd/csrc pattern` -> libdtof_pattern.so, built by __graft_entry__.build()): a result that depends on an uninitialised register shows
as a failing lane here although it may be right by accident in the regular build (the K = 4 films of round 3, profiles/r03_k4_uninitialised.txt).
The whole suite runs against that library with `tools/gpu_session.sh pattern`; this test runs its core -- the lane-parity configurations and the random scene sweep --
in a child process whose library is the pattern build."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mitsuba3dopplertof_amd", "libdtof_pattern.so")


@pytest.mark.gpu
def test_lane_parity_and_scene_sweep_on_the_pattern_initialised_build():
    if os.environ.get("DTOF_LIB"):
        pytest.skip("already running against a library variant")
    if not os.path.exists(LIB):
        pytest.skip("libdtof_pattern.so is not built (make -C mitsuba3dopplertof_amd/csrc pattern)")
    env = dict(os.environ, DTOF_LIB=LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_mask.py"), "-q", "-x", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k", "every_lane or random_scene or valid_ray or full_domino"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


STATS_LIB = os.path.join(ROOT, "mitsuba3dopplertof_amd", "libdtof_stats.so")
POISON_CHILD = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "scenes"))
import make_scenes; make_scenes.ensure()
import mitsuba3dopplertof_amd as mi
L = mi._lib()
out = (C.c_ulonglong * 16)()
L.dtof_debug_traversal_stats(out)           # reset
n_frames = 0
for pipeline in ("auto", "fused"):
    if pipeline != "auto":
        os.environ["DTOF_PIPELINE"] = pipeline
    # ragged segments on purpose: 13 x 11 pixels x 5 / 12 spp end in the middle of a wave, so lanes WITHOUT a path sit beside lanes with one
    for scene, params, spp in (("cornell_boxes.xml", dict(resx=13, resy=11), 5), ("cornell_specular.xml", dict(resx=13, resy=11, max_depth=8), 12),
                               ("cornell_area.xml", dict(resx=13, resy=11, max_depth=6), 5), ("domino.xml", dict(resx=37, resy=29), 6), ("cornell_wall.xml", dict(resx=13, resy=11), 12)):
        sc = mi.load_file(os.path.join(%(root)r, "scenes", scene), **params)
        a = sc.render(seed=1, spp=spp)
        b = sc.render(seed=1, spp=spp, offsets=[0.0, 0.25, 0.5, 0.75])
        assert np.isfinite(a).all() and np.isfinite(b).all()
        assert np.abs(b[0] - a).max() <= 1e-5 * max(np.abs(a).max(), 1e-20), scene      # film 0 of the K = 4 batch is the K = 1 film
        n_frames += 2
L.dtof_debug_traversal_stats(out)
print("POISON_HITS", int(out[14]), "RAYS", int(out[0]), "FRAMES", n_frames)
'''


@pytest.mark.gpu
def test_poisoned_inactive_lanes_never_reach_the_queues():
    """in the statistics build (make -C mitsuba3dopplertof_amd/csrc stats) every lane of k_shade that has no path carries a poison pattern in its path-state
    registers, and every place where path state leaves the registers counts it.  K = 1 and K = 4 frames with ragged last waves, all pipelines: the count stays 0."""
    if not os.path.exists(STATS_LIB):
        pytest.skip("libdtof_stats.so is not built (make -C mitsuba3dopplertof_amd/csrc stats)")
    env = dict(os.environ, DTOF_LIB=STATS_LIB)
    env.pop("DTOF_PIPELINE", None)
    r = subprocess.run([sys.executable, "-c", POISON_CHILD % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("POISON_HITS")][-1].split()
    assert int(line[1]) == 0 and int(line[3]) > 0, line
