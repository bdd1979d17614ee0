#!/usr/bin/env python3
"""Static instruction mix of the kernels in a gfx950 assembly listing (hipcc --cuda-device-only -S): per kernel the number of
VALU / SALU / LDS / VMEM / branch instructions, the slow VALU ops (v_rcp / v_sqrt / v_div_* / 64-bit multiplies), registers and
LDS.  Static counts are not executed counts, but for straight-line shading code they rank where the issue slots go.
usage: isa_count.py file.s [substring of the mangled kernel name ...]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split("\n")
filters = sys.argv[2:]
kern = None; stats = collections.OrderedDict()
for ln in txt:
    m = re.match(r"^([_A-Za-z]\w+):", ln)
    if m:
        kern = m.group(1); stats[kern] = collections.Counter(); continue
    if kern is None: continue
    s = ln.strip()
    if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
        pass
    m = re.match(r"^\s+([a-z_0-9]+)\s", ln)
    if not m:
        m2 = re.match(r"^\s*\.(vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count):\s*(\d+)", ln)
        m3 = re.match(r"^\s*; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", ln)
        if m3: stats[kern]["meta_" + m3.group(1)] = int(m3.group(2))
        continue
    op = m.group(1); c = stats[kern]
    if op.startswith("v_"):
        c["valu"] += 1
        if re.match(r"v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|exp|log|sin|cos)", op): c["valu_trans_div"] += 1
        if re.match(r"v_(mul_lo_u32|mul_hi_u32|mad_u64_u32|mad_i64_i32)", op): c["valu_mul32"] += 1
        if "f64" in op: c["valu_f64"] += 1
        if op.startswith("v_cndmask"): c["cndmask"] += 1
    elif op.startswith("s_"):
        if op.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
        elif op.startswith(("s_waitcnt", "s_nop")): c["wait"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"): c["smem"] += 1
        else: c["salu"] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        c["vmem"] += 1
        if op.startswith("scratch_"): c["scratch"] += 1
for k, c in stats.items():
    if filters and not all(f in k for f in filters): continue
    if not c["valu"] and not c["salu"]: continue
    print(k)
    print("   " + "  ".join("%s=%d" % kv for kv in sorted(c.items())))
