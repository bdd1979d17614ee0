#!/usr/bin/env python3
"""Prices the VALU instructions of the dominant kernels with the issue rates MEASURED on the MI355X (profiles/r03_ubench_valu_rate.txt,
tools/ubench/valu_rate.hip) and writes profiles/valu_cycle_model.json (build container: needs hipcc, no GPU).

The 78.6 T lane-instr/s "issue peak" bench.py prices `roofline.frac` against assumes one wave64 instruction per 2 cycles per SIMD.  The
microbenchmark shows that only a subset of the instruction forms issues at that rate on gfx950:
  * 2 cycles (2.25 nominal): v_fma / v_fmac / v_mul / v_add / v_sub _f32, v_mov_b32, v_add / v_sub _u32, v_and / v_or / v_xor _b32,
    v_lshrrev_b32, v_ashrrev_i32 -- with VGPR, inline-constant or literal operands;
  * 4 cycles (4.1 nominal): the SAME instructions with an SGPR operand, and v_min / v_max / v_med3, v_cmp*, v_cndmask, v_cvt*, v_floor / v_fract /
    v_ldexp, v_lshlrev_b32, every fused integer op (v_lshl_add, v_add3, v_xad, v_bfe, v_alignbit, v_bitop3, v_mad*), v_mul_lo / v_mul_hi,
    v_div_scale / v_div_fmas / v_div_fixup, v_readlane / v_readfirstlane, DPP moves, all packed-f32 and f64 arithmetic;
  * 8 cycles (8.2 nominal): v_rcp / v_sqrt / v_rsq / v_sin / v_cos / v_exp / v_log _f32.
So a kernel's VALU pipes are busy for  sum(count x cycles)  cycles, not for 2 x count.  This script disassembles the kernels (hipcc -S with the
Makefile's flags), classifies every VALU instruction of the named instantiations and reports the STATIC average cycles per VALU instruction; bench.py
multiplies the executed SQ_INSTS_VALU with it (`roofline.valu_busy_est`).  The static mix stands in for the dynamic one: the first-bounce kernels are
mostly straight-line code (C2 executes 4 040 lane slots per path through 4 564 static VALU instructions).

usage: python tools/valu_cycle_model.py [--asm FILE.s]
"""
import collections, json, os, re, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(HERE, "tools"))
from pmc_summary import kernel_sources_sha16, short   # noqa: E402

FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_xor_b32", "v_and_b32", "v_or_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mul_legacy_f32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}
QUARTER = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"}
C_FAST, C_SLOW, C_QUARTER = 2.25, 4.1, 8.2      # nominal 2.4 GHz cycles per wave64 instruction per SIMD, W >= 4 (the measured table)
# the kernels the bench configurations are dominated by (pmc_summary.short() names)
WANTED = {"c2": "_ZN4dtof7k_shadeILb1ELi2ELb0ELi1ELb0ELi0ELi0ELb0EEEvNS_9ShadeArgsE",
          "c3": "_ZN4dtof7k_shadeILb1ELi2ELb0ELi1ELb0ELi0ELi0ELb0EEEvNS_9ShadeArgsE",
          "c4": "_ZN4dtof7k_shadeILb0ELi2ELb0ELi1ELb1ELi0ELi16ELb0EEEvNS_9ShadeArgsE",
          "c5": "_ZN4dtof7k_shadeILb0ELi2ELb0ELi4ELb1ELi0ELi16ELb0EEEvNS_9ShadeArgsE"}


def classify(op, args):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in QUARTER:
        return "quarter", C_QUARTER
    srcs = [a.strip() for a in args.split(",")[1:]]
    sgpr = any(re.search(r"^-?\|?(s\d+|s\[\d+:\d+\]|vcc_lo|vcc_hi|exec_lo|exec_hi|m0)\|?$", a) for a in srcs)
    if base in FAST and not op.endswith(("_dpp", "_sdwa")):
        if sgpr:
            return "slow: SGPR operand", C_SLOW
        if base == "v_fmac_f32" and len(srcs) >= 2 and srcs[0] == srcs[1]:
            return "slow: v_fmac x,x", C_SLOW
        return "fast", C_FAST
    return "slow: " + base, C_SLOW


def model(asm_lines, symbol):
    start = next(i for i, l in enumerate(asm_lines) if l.startswith(symbol + ":"))
    n = collections.Counter(); cyc = collections.Counter(); salu = 0
    for l in asm_lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"\s+(v_[a-z0-9_]+)\s*(.*)", l)
        if m:
            cl, c = classify(m.group(1), m.group(2).split(";")[0])
            n[cl] += 1; cyc[cl] += c
        elif re.match(r"\s+s_[a-z0-9_]+", l) and not re.match(r"\s+s_(waitcnt|nop|endpgm|barrier|sleep)", l):
            salu += 1
    total_n, total_c = sum(n.values()), sum(cyc.values())
    top = sorted(cyc.items(), key=lambda kv: -kv[1])
    return {"valu_static": total_n, "salu_static": salu, "avg_cycles_per_valu": round(total_c / total_n, 4),
            "share_of_cycles": {"fast (2.25)": round(cyc["fast"] / total_c, 4),
                                "slow (4.1)": round(sum(v for k, v in cyc.items() if k.startswith("slow")) / total_c, 4),
                                "quarter (8.2)": round(cyc["quarter"] / total_c, 4)},
            "share_of_instructions": {"fast": round(n["fast"] / total_n, 4), "slow": round(sum(v for k, v in n.items() if k.startswith("slow")) / total_n, 4),
                                      "quarter": round(n["quarter"] / total_n, 4)},
            "largest_slow_classes": [{"class": k, "instructions": n[k], "share_of_cycles": round(v / total_c, 4)} for k, v in top if k != "fast"][:14]}


def main():
    asm = None
    if "--asm" in sys.argv:
        asm = sys.argv[sys.argv.index("--asm") + 1]
    else:
        tmp = tempfile.mkdtemp()
        csrc = os.path.join(HERE, "mitsuba3dopplertof_amd", "csrc")
        lines = []
        for tu in ("dtof_shade_plain", "dtof_shade_res0"):   # the translation units that hold the headline (C2) and the Domino (C4 / C5) first-bounce kernels
            asm = os.path.join(tmp, tu + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                            "-S", "--cuda-device-only", os.path.join(csrc, tu + ".hip"), "-o", asm], check=True, stderr=subprocess.DEVNULL)
            lines += open(asm).read().split("\n")
        asm = None
    if asm:
        lines = open(asm).read().split("\n")
    dem = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()
    out = {"csrc_sha16": kernel_sources_sha16(), "cycles": {"fast": C_FAST, "slow": C_SLOW, "quarter": C_QUARTER},
           "source": "static VALU instruction mix of the kernel x issue rates of profiles/r03_ubench_valu_rate.txt (tools/valu_cycle_model.py)", "configs": {}}
    for cfg, sym in WANTED.items():
        m = model(lines, sym); m["symbol"] = dem(sym); m["kernel"] = short(m["symbol"])
        out["configs"][cfg] = m
        print(cfg, m["symbol"], "avg cycles per VALU instruction:", m["avg_cycles_per_valu"], m["share_of_cycles"])
    json.dump(out, open(os.path.join(HERE, "profiles", "valu_cycle_model.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
