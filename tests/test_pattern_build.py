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
