#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 --pmc output directories:
    python tools/pmc_counters.py gpurun_out/pmc_a [gpurun_out/pmc_b ...]"""
import csv, glob, os, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dtof::", "").replace("dtof::", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    c = acc[k]
    n = max(len(v) for v in c.values())
    waves = sum(c.get("SQ_WAVES", [0])) / max(len(c.get("SQ_WAVES", [1])), 1)
    print("%s  (%d launches)" % (k, n))
    for name in sorted(c):
        avg = sum(c[name]) / len(c[name])
        extra = "  per wave %.1f" % (avg / waves) if waves and name != "SQ_WAVES" else ""
        print("    %-28s %16.1f%s" % (name, avg, extra))
