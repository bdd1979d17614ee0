#!/usr/bin/env python3
"""Static per-source-line VALU attribution of one kernel in an assembly listing built with -gline-tables-only:
usage: isa_lines.py file.s <substring of the kernel symbol> [top N].  Weights: 1 per VALU instruction, 2 for the 8-cycle
transcendentals; lines are the innermost inlined callee's (file:line)."""
import re, sys, collections
txt = open(sys.argv[1]).read().split("\n"); key = sys.argv[2]; top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = {}; cur = None; inside = False; cnt = collections.Counter(); tot = 0
for ln in txt:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]; continue
    m = re.match(r"^([_A-Za-z]\w+):", ln)
    if m: inside = key in m.group(1); continue
    if not inside: continue
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", ln)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    m = re.match(r"^\s+(v_[a-z_0-9]+)\s", ln)
    if m:
        w = 2 if re.match(r"v_(rcp|rsq|sqrt|exp|log)", m.group(1)) else 1
        cnt[cur] += w; tot += w
print("total weighted VALU:", tot)
byfile = collections.Counter()
for (f, l), c in cnt.items(): byfile[f] += c
print(dict(byfile))
for (f, l), c in cnt.most_common(top): print("%5d  %s:%d" % (c, f, l))
