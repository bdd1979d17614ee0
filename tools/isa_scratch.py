#!/usr/bin/env python3
"""Where a kernel's spill code sits: every scratch_load / scratch_store of one kernel in an assembly listing built with -gline-tables-only, with the loop nesting depth of
the instruction (loops = backward branches to a label) and the source line it is attributed to.  Spill code outside the loops costs once per path vertex; inside the
traversal loops it costs once per node step.
usage: isa_scratch.py file.s <substring of the kernel symbol> [top N]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split("\n"); key = sys.argv[2]; top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files = {}; body = []; inside = False; cur = None
for ln in txt:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]; continue
    m = re.match(r"^([_A-Za-z]\w+):", ln)
    if m and not m.group(1).startswith(".L"):
        if inside: break
        inside = key in m.group(1); continue
    if not inside: continue
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", ln)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    m = re.match(r"^(\.LBB\w+):", ln)
    if m: body.append(("label", m.group(1), None)); continue
    m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)", ln)
    if m and not m.group(1).startswith("."): body.append(("op", m.group(1), (m.group(2), cur)))
pos = {name: i for i, (kind, name, _) in enumerate(body) if kind == "label"}
loops = []   # (start index, end index) of every backward branch
for i, (kind, op, arg) in enumerate(body):
    if kind == "op" and op.startswith(("s_cbranch", "s_branch")):
        t = arg[0].split()[0] if arg[0] else ""
        if t in pos and pos[t] < i: loops.append((pos[t], i))
depth = [0] * len(body)
for a, b in loops:
    for i in range(a, b + 1): depth[i] += 1
by_depth = collections.Counter(); by_line = collections.Counter(); valu_depth = collections.Counter()
for i, (kind, op, arg) in enumerate(body):
    if kind != "op": continue
    if op.startswith("v_"): valu_depth[depth[i]] += 1
    if op.startswith("scratch_"):
        by_depth[(depth[i], "load" if "load" in op else "store")] += 1
        by_line[(depth[i], arg[1], "load" if "load" in op else "store")] += 1
print("loops (backward branches):", len(loops), " VALU instructions by loop depth:", dict(sorted(valu_depth.items())))
print("scratch instructions by loop depth:", {("depth %d %s" % k): v for k, v in sorted(by_depth.items())})
for (d, line, kind), c in sorted(by_line.items(), key=lambda kv: (-kv[0][0], -kv[1]))[:top]:
    print("  depth %d  %-5s x%-3d %s:%s" % (d, kind, c, line[0] if line else "?", line[1] if line else "?"))
# innermost loops: intervals that contain no other backward branch
inner = [(a, b) for (a, b) in loops if not any((c, d) != (a, b) and a <= c and d <= b for (c, d) in loops)]
print("innermost loops: VALU / LDS / VMEM / scratch instructions, dominant source lines")
for a, b in inner:
    c = collections.Counter(); lines = collections.Counter()
    for kind, op, arg in body[a:b + 1]:
        if kind != "op": continue
        c["valu"] += op.startswith("v_"); c["lds"] += op.startswith("ds_"); c["scratch"] += op.startswith("scratch_")
        c["vmem"] += op.startswith(("global_", "flat_", "buffer_"))
        if arg[1]: lines["%s:%d" % arg[1]] += 1
    if c["valu"] >= 8:
        print("  %4d valu %3d lds %3d vmem %3d scratch   %s" % (c["valu"], c["lds"], c["vmem"], c["scratch"], ", ".join(k for k, _ in lines.most_common(3))))
