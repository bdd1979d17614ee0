#!/usr/bin/env python3
"""A/B timing of several builds of libdtof.so on the same GPU box: tools/ab/*.so, each in its own process (DTOF_LIB), several
interleaved rounds; prints min / median of the library's own frame time (HIP events)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, numpy as np; sys.path.insert(0, %r); import mitsuba3dopplertof_amd as mi\n"
        "sc = mi.load_file(%r)\n"
        "t = []\n"
        "for i in range(40):\n"
        "    sc.render(seed=0, spp=0); t.append(sc.last_stats['ms_total'])\n"
        "t = np.array(t[5:]); print('%%.3f %%.3f' %% (t.min(), np.median(t)))\n" % (ROOT, os.path.join(ROOT, "scenes", sys.argv[1] if len(sys.argv) > 1 else "cornell_wall.xml")))
libs = sorted(glob.glob(os.path.join(ROOT, "tools", "ab", "*.so")))
res = {l: [] for l in libs}
for r in range(4):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, DTOF_LIB=l))
        res[l].append(out.stdout.strip() or out.stderr[-300:])
for l in libs:
    print("%-28s %s" % (os.path.basename(l), "  ".join(res[l])))
