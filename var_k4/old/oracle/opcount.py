#!/usr/bin/env python3
"""Executed arithmetic of the CPU oracle, counted exactly: the algorithmic work of one path, to price the GPU kernels against ("make the
roofline algorithmic").  Test infrastructure, like the oracle itself: nothing in the product imports this.

How: oracle/dtof_oracle.c is compiled to LLVM IR (clang -O1, no inlining, no vectorisation, no unrolling, -ffp-contract=off as in the oracle's own
build), every basic block gets a 64-bit execution counter (three IR instructions inserted textually), the instrumented IR is linked into
oracle/_count/liborc_count.so and a sample of the workload is run through it on ONE thread.  executed ops = sum over blocks of (block count x the
arithmetic instructions the block holds), by category and by function.  Categories (one LLVM instruction = one op; an fma is one op):
    f32_addmul   fadd fsub fmul fneg                      f32_fma    llvm.fma / fmuladd
    f32_div      fdiv                                    f32_sqrt   sqrt calls / intrinsics
    f32_other    fcmp, select on floats, fabs, min / max, copysign, floor, conversions, other libm calls
    f64          every double-precision op (spheres, cylinders, loader-time tables)
    int          integer add / sub / mul / shifts / logic / icmp / integer select (TEA, PCG32, Kensler, lane -> pixel mapping)
Not counted: loads, stores, address arithmetic (getelementptr), phi, branches, calls of the oracle's own functions.

Function groups: `query` = everything under the closest-hit / occlusion queries (the oracle tests EVERY object for every ray: for scenes behind a BVH
this is not what a traversal needs -- the product's own traversal counters price that part), `path` = the rest of eval_lane (sampler, camera ray,
surface interaction, emitter sampling, BSDF, modulation weight, MIS, russian roulette).

usage: python oracle/opcount.py OUT.json scene.xml [spp [key=value ...]]      e.g.  oracle/opcount.py profiles/r03_oracle_opcount_c2.json cornell_wall.xml 8 resx=64 resy=64
"""
import ctypes as C, hashlib, json, os, re, subprocess, sys
from collections import defaultdict

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"
OUT_DIR = os.path.join(HERE, "oracle", "_count")
FLAGS = ["-O1", "-std=gnu11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-march=x86-64-v3", "-fno-inline", "-fno-vectorize", "-fno-slp-vectorize",
         "-fno-unroll-loops", "-pthread"]

F32_ADDMUL = {"fadd", "fsub", "fmul", "fneg"}
INT_OPS = {"add", "sub", "mul", "shl", "lshr", "ashr", "and", "or", "xor", "udiv", "urem", "sdiv", "srem", "icmp"}
CONV = {"sitofp", "uitofp", "fptosi", "fptoui", "fpext", "fptrunc", "bitcast", "trunc", "zext", "sext"}


def classify(line):
    """category of one IR instruction line, or None"""
    m = re.match(r"\s*(?:%[\w.]+\s*=\s*)?(?:tail |musttail |notail )?(\w+)\b(.*)", line)
    if not m:
        return None
    op, rest = m.group(1), m.group(2)
    is64 = bool(re.search(r"\bdouble\b", rest))
    if op in F32_ADDMUL:
        return "f64" if is64 else "f32_addmul"
    if op == "fdiv" or op == "frem":
        return "f64" if is64 else "f32_div"
    if op == "fcmp":
        return "f64" if is64 else "f32_other"
    if op == "select":
        if re.search(r"\b(float|double)\b", rest):
            return "f64" if is64 else "f32_other"
        return "int"
    if op in INT_OPS:
        return None if re.search(r"\bptr\b", rest) else "int"
    if op in CONV:
        if op in ("bitcast", "trunc", "zext", "sext"):
            return None          # free on the GPU (register reinterpretation / implicit)
        return "f64" if is64 and op not in ("fpext", "fptrunc") else "f32_other"
    if op == "call":
        c = re.search(r"@([\w.]+)\(", rest)
        if not c:
            return None
        name = c.group(1)
        if name.startswith("llvm.fma") or name.startswith("llvm.fmuladd") or name in ("fmaf", "fma"):
            return "f64" if "f64" in name or name == "fma" else "f32_fma"
        if name.startswith("llvm.sqrt") or name in ("sqrtf", "sqrt"):
            return "f64" if "f64" in name or name == "sqrt" else "f32_sqrt"
        if name.startswith(("llvm.fabs", "llvm.minnum", "llvm.maxnum", "llvm.copysign", "llvm.floor", "llvm.ceil", "llvm.trunc", "llvm.rint", "llvm.round", "llvm.minimum", "llvm.maximum")):
            return "f64" if "f64" in name else "f32_other"
        if name.startswith(("llvm.umin", "llvm.umax", "llvm.smin", "llvm.smax", "llvm.abs", "llvm.fshl", "llvm.fshr", "llvm.ctlz", "llvm.cttz", "llvm.ctpop", "llvm.bswap")):
            return "int"
        if name in ("fmodf", "floorf", "ceilf", "fabsf", "fminf", "fmaxf", "copysignf", "cosf", "sinf", "tanf", "acosf", "atan2f", "expf", "logf", "powf", "truncf", "roundf", "ldexpf", "frexpf"):
            return "f32_libm"
        if name in ("fmod", "floor", "ceil", "fabs", "fmin", "fmax", "copysign", "cos", "sin", "tan", "acos", "atan2", "exp", "log", "pow", "erf", "ldexp", "frexp"):
            return "f64"
        return None
    return None


def instrument(ll_text):
    """returns (instrumented IR, blocks): blocks[i] = (function, label, {category: n})"""
    out, blocks = [], []
    fn = None; pending = None    # pending: index of a block whose counter code still has to be placed (after its phis)
    lines = ll_text.split("\n")

    def counter_code(i):
        p = "getelementptr inbounds ([__NBLOCKS__ x i64], ptr @__orc_bb, i64 0, i64 %d)" % i
        return ["  %%__orc_c%d = load i64, ptr %s, align 8" % (i, p), "  %%__orc_d%d = add i64 %%__orc_c%d, 1" % (i, i), "  store i64 %%__orc_d%d, ptr %s, align 8" % (i, p)]
    for ln in lines:
        if ln.startswith("define "):
            fn = re.search(r"@([\w.]+)\(", ln).group(1)
            out.append(ln)
            blocks.append((fn, "entry", defaultdict(int))); pending = len(blocks) - 1
            continue
        if fn is None:
            out.append(ln); continue
        if ln.startswith("}"):
            fn = None; pending = None; out.append(ln); continue
        m = re.match(r"^([\w.]+):", ln)
        if m:
            out.append(ln)
            blocks.append((fn, m.group(1), defaultdict(int))); pending = len(blocks) - 1
            continue
        s = ln.strip()
        if pending is not None and s and not s.startswith(";"):
            if re.match(r"%[\w.]+\s*=\s*phi\b", s):
                out.append(ln); continue
            out.extend(counter_code(pending)); pending = None
        out.append(ln)
        if s and not s.startswith(";") and blocks:
            c = classify(ln)
            if c:
                blocks[-1][2][c] += 1
            cm = re.search(r"\bcall\b[^@]*@([\w.]+)\(", ln)
            if cm and not cm.group(1).startswith("llvm."):
                blocks[-1][2]["call:" + cm.group(1)] += 1
    text = "\n".join(out).replace("__NBLOCKS__", str(len(blocks)))
    # the counter array: appended after the target triple line
    decl = "@__orc_bb = dso_local global [%d x i64] zeroinitializer, align 16\n" % len(blocks)
    text = re.sub(r"(target triple = [^\n]*\n)", lambda m_: m_.group(1) + "\n" + decl, text, count=1)
    return text, blocks


def build():
    os.makedirs(OUT_DIR, exist_ok=True)
    src = os.path.join(HERE, "oracle", "dtof_oracle.c")
    ll, ill, so, meta = (os.path.join(OUT_DIR, n) for n in ("orc.ll", "orc_count.ll", "liborc_count.so", "blocks.json"))
    digest = hashlib.sha256(open(src, "rb").read() + open(os.path.join(HERE, "oracle", "dtof_oracle.h"), "rb").read() + open(__file__, "rb").read()).hexdigest()[:16]
    if os.path.exists(so) and os.path.exists(meta) and json.load(open(meta)).get("digest") == digest:
        return so, json.load(open(meta))
    subprocess.check_call([CLANG] + FLAGS + ["-S", "-emit-llvm", src, "-o", ll])
    text, blocks = instrument(open(ll).read())
    open(ill, "w").write(text)
    subprocess.check_call([CLANG, "-O1", "-fPIC", "-shared", "-pthread", "-Wno-override-module", ill, "-o", so, "-lm"])
    info = {"digest": digest, "blocks": [[f, l, dict(c)] for f, l, c in blocks]}
    json.dump(info, open(meta, "w"))
    return so, info


QUERY_ROOTS = ("scene_closest", "scene_occluded")   # Scene::ray_intersect / ray_test: everything executed under them is the `query` group


def main():
    out_path, scene = sys.argv[1], sys.argv[2]
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    params = dict(a.split("=", 1) for a in sys.argv[4:] if not a.startswith("--"))
    opts = dict(a[2:].split("=", 1) for a in sys.argv[4:] if a.startswith("--"))
    row_step, band = int(opts.get("row-step", 16)), int(opts.get("band", 1))     # sample: rows y with y % row_step < band, all their lanes, at the workload's own size and spp
    so, info = build()
    os.environ["DTOF_ORACLE_LIB"] = so
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "scenes"))
    import make_scenes; make_scenes.ensure()
    from oracle import orc
    path = scene if os.path.exists(scene) else os.path.join(HERE, "scenes", scene)
    osc = orc.Scene(path, params)
    pd = osc.params()
    spp = spp or pd["sample_count"]
    w, h = osc.size
    L = orc.lib()
    nb = len(info["blocks"])
    arr = (C.c_uint64 * nb).in_dll(L, "__orc_bb")
    before = list(arr)
    n = 0
    for y in range(0, h, row_step):
        rows = min(band, h - y)
        lanes = osc.render_lanes(pd, 0, spp, y * w * spp, rows * w * spp, threads=1)
        n += rows * w * spp
    after = list(arr)
    per_fn = defaultdict(lambda: defaultdict(int)); total = defaultdict(int); calls = {}
    edges = defaultdict(lambda: defaultdict(int))    # edges[callee][caller] = executed calls
    for i, (f, lbl, cats) in enumerate(info["blocks"]):
        cnt = after[i] - before[i]
        if lbl == "entry":
            calls[f] = cnt
        for c, k in cats.items():
            if c.startswith("call:"):
                edges[c[5:]][f] += cnt * k
            else:
                per_fn[f][c] += cnt * k; total[c] += cnt * k
    # share of a function's executions that happen under a ray query: 1 for the query roots, else the call-weighted mean over its callers (the oracle
    # has no recursion); a helper shared by both groups (dot products, transforms) is split by where its calls come from
    share = {}
    def query_share(f, depth=0):
        if f in share:
            return share[f]
        if f in QUERY_ROOTS:
            share[f] = 1.0; return 1.0
        callers = {c: k for c, k in edges.get(f, {}).items() if k and c != f}
        tot = sum(callers.values())
        share[f] = 0.0
        if tot and depth < 64:
            share[f] = sum(k * query_share(c, depth + 1) for c, k in callers.items()) / tot
        return share[f]
    groups = {"query": defaultdict(float), "path": defaultdict(float)}
    for f, cats in per_fn.items():
        q = query_share(f)
        for c, v in cats.items():
            groups["query"][c] += v * q; groups["path"][c] += v * (1.0 - q)
    assert lanes["rgb"].shape[0] > 0 and n > 0
    res = {
        "what": "executed arithmetic of oracle/dtof_oracle.c per path (exact basic-block counts x the arithmetic instructions of each block, clang -O1 IR, one op per instruction, fma = 1)",
        "scene": os.path.basename(path), "params": params, "spp": spp, "paths": n, "sample": "every lane of the rows y with y %% %d < %d of the %d x %d frame" % (row_step, band, w, h),
        "oracle_sha16": info["digest"],
        "ops_per_path": {c: round(v / n, 2) for c, v in sorted(total.items())},
        "ops_per_path_total": round(sum(total.values()) / n, 1),
        "groups_per_path": {g: {c: round(v / n, 2) for c, v in sorted(d.items())} for g, d in groups.items()},
        "groups_per_path_total": {g: round(sum(d.values()) / n, 1) for g, d in groups.items()},
        "calls_per_path": {f: round(c / n, 3) for f, c in sorted(calls.items(), key=lambda kv: -kv[1]) if c and c / n >= 0.01},
        "query_share_of_function": {f: round(q, 3) for f, q in sorted(share.items()) if 0.0 < q < 1.0},
        "ops_per_call": {f: round(sum(c.values()) / calls[f], 2) for f, c in sorted(per_fn.items()) if calls.get(f) and sum(c.values())},
        "top_functions_ops_per_path": {f: round(sum(c.values()) / n, 1) for f, c in sorted(per_fn.items(), key=lambda kv: -sum(kv[1].values()))[:25] if sum(c.values())},
    }
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
