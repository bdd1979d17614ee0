#!/usr/bin/env python3
"""How the CPU oracle scales with host threads on this box (context for bench.py's cpu_baseline):
prints the CPUs the process may use and Mpaths/s of the C2 frame at several thread counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_scenes; make_scenes.ensure()
from oracle import orc
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
sc = orc.Scene(os.path.join(ROOT, "scenes", "cornell_wall.xml"), {})
pd = sc.params()
for nt in [int(x) for x in sys.argv[1:]] or [1, 8, 16, 32, 64, 128, 256]:
    rows = (0, 512) if nt >= 8 else (0, 32)
    t = time.time(); sc.render(pd, seed=0, spp=64, threads=nt, rows=rows); dt = time.time() - t
    print("threads %3d: %.2f Mpaths/s (%.2f s)" % (nt, (rows[1] - rows[0]) * 512 * 64 / dt / 1e6, dt), flush=True)
