#!/usr/bin/env python3
"""Static list of the fast-form VALU instructions of one kernel that carry an SGPR operand (they issue at half rate on gfx950, profiles/r03_ubench_valu_rate.txt),
clustered by position in the listing.  usage: isa_sgpr_operands.py file.s <substring of the kernel symbol> [all]   (listing built with -gline-tables-only)"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n'); key = sys.argv[2]; every = len(sys.argv) > 3
inside = False; files = {}; cur = None; tot = 0
fast = re.compile(r'^v_(fma|fmac|mul|add|sub|subrev|mac)_f32' if not every else r'^v_(fma|fmac|mul|add|sub|subrev|mac)_f32|^v_mov_b32|^v_(add|sub|subrev)_u32|^v_(and|or|xor)_b32|^v_lshrrev_b32|^v_ashrrev_i32')
out = []
for i, ln in enumerate(txt):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]; continue
    m = re.match(r'^([_A-Za-z]\w+):', ln)
    if m: inside = key in m.group(1); continue
    if not inside: continue
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', ln)
    if m: cur = (files.get(int(m.group(1))), int(m.group(2))); continue
    m = re.match(r'^\s+(v_[a-z_0-9]+)\s+(.*)$', ln)
    if m:
        tot += 1
        if fast.match(m.group(1)) and re.search(r'(?<![a-z_\[])s\d+|s\[\d+', ','.join(m.group(2).split(';')[0].split(',')[1:])): out.append((i, cur, ln.strip()[:70]))
print("VALU instructions:", tot, " fast forms with an SGPR operand:", len(out))
prev = None
for i, c, l in out:
    if prev is None or i - prev > 40: print('----')
    print(i, c, l); prev = i
