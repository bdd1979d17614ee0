#!/usr/bin/env python3
"""A/B timing on one GPU box of several (library, environment) variants of the same workload, interleaved over several rounds.
usage: ab_env.py scene.xml [spp] -- name=[LIB.so][,ENV=VAL ...] ...   e.g.  ab_env.py cornell_wall.xml -- base=tools/ab/base.so new= memo0=,DTOF_INSTANCE_MEMO=0
Prints min / median of ms_total, ms_first (first-bounce kernel) and of the bounce-kernel launches (HIP events of the library)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]; k = args.index("--"); scene = args[0]; spp = int(args[1]) if k > 1 else 0
variants = []
for v in args[k + 1:]:
    name, rest = v.split("=", 1); parts = rest.split(",")
    env = dict(p.split("=", 1) for p in parts[1:] if p)
    if parts[0]: env["DTOF_LIB"] = os.path.join(ROOT, parts[0])
    variants.append((name, env))
code = ("import sys, numpy as np; sys.path.insert(0, %r); import mitsuba3dopplertof_amd as mi\n"
        "sc = mi.load_file(%r)\n"
        "T = []\n"
        "for i in range(30):\n"
        "    sc.render(seed=0, spp=%d); s = sc.last_stats; T.append((s['ms_total'], s['ms_first'], (s['ms_shade'] - s['ms_first']) / max(s['n_launches_shade'] - s['n_launches_first'], 1), s['ms_trace'], s['ms_shadow'], s['ms_splat']))\n"
        "T = np.array(T[5:]); print(' '.join('%%.3f/%%.3f' %% (T[:, j].min(), np.median(T[:, j])) for j in range(6)))\n" % (ROOT, os.path.join(ROOT, "scenes", scene), spp))
res = {n: [] for n, _ in variants}
for r in range(3):
    for n, env in variants:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env))
        res[n].append(out.stdout.strip() or out.stderr[-300:])
print("variant: per round min/median of  total | first | bounce launch | trace | shadow | splat  (ms)")
for n, _ in variants:
    for r in res[n]: print("%-14s %s" % (n, r))
