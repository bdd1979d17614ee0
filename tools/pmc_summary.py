#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc passes of ONE bench configuration into profiles/roofline_traffic.json (merged: other configurations stay).

    python tools/pmc_summary.py OUT.json CONFIG FETCH_DIR WRITE_DIR VALU_DIR [FRAMES]

FRAMES: frames the profiled command rendered (warm-up + the synchronous counter frame + timed steps); with it every kernel also gets `launches_per_step`.

FETCH_DIR / WRITE_DIR / VALU_DIR: output directories of three separate passes (--pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
SQ_WAVES) over `bench.py --config CONFIG --no-extra --no-cpu-baseline`.  Per kernel and launch:
  * HBM bytes as prescribed by MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
    exactly half of the bytes of wide coalesced reads, so reads = 2 * FETCH_SIZE * 1024; WRITE_SIZE reads exactly for 16-B streaming stores and atomics;
  * executed VALU wave-instructions (x 64 = issued lane slots) and the active-lane ratio SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU).
The file is STAMPED with a hash of the kernel sources the counters were taken on (csrc_sha16, see kernel_sources_sha16); bench.py recomputes it and
reports `counters_stale: true` when the sources have changed since.  The full kernel symbol is kept beside the short name.
"""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("dtof_kernels.hip", "dtof_shade.h", "dtof_device.h", "dtof_kernels.h", "dtof_traverse.h", "dtof_sampling.h", "dtof_shading.h", "dtof_math.h", "dtof_scene.h", "dtof_render.hip", "Makefile")


def kernel_sources_sha16(root=HERE):
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(root, "mitsuba3dopplertof_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    m = re.search(r"k_shade<\w+, (\d), \w+, (\d+), (\w+), \w+(?:, (\d+))?(?:, \w+)?>", name)     # k_shade<LDS, MODE, AREA, KMAX, MESH, SPEC, RESW, RH16>
    if m:
        if m.group(1) == "2":      # the first-bounce instantiation (lane generation + primary ray + inline iterations)
            return "k_shade_first" + ("_mesh" if m.group(3) == "true" else "") + ("_k4" if m.group(2) != "1" else "") + ("_resident" if (m.group(4) or "0") != "0" else "")
        return "k_shade"
    for k in ("k_trace", "k_shadow", "k_generate", "k_splat_x8", "k_splat_pixel", "k_splat_tent3", "k_splat_generic", "k_develop", "k_sum_counts"):
        if k in name:
            return k
    return None


def main():
    out_path, config, d_fetch, d_write, d_valu = sys.argv[1:6]
    frames = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    fetch, write = collect(d_fetch, "FETCH_SIZE"), collect(d_write, "WRITE_SIZE")
    valu, thr, waves = collect(d_valu, "SQ_INSTS_VALU"), collect(d_valu, "SQ_THREAD_CYCLES_VALU"), collect(d_valu, "SQ_WAVES")
    kernels = {}
    for name in set(fetch) | set(write) | set(valu):
        k = short(name)
        if not k:
            continue
        f, w = fetch.get(name, []), write.get(name, [])
        fk, wk = sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)
        n_launches = max(len(f), len(w), len(valu.get(name, [])))
        rec = {"symbol": name, "launches_sampled": n_launches, "FETCH_SIZE_KiB_avg": round(fk, 1), "WRITE_SIZE_KiB_avg": round(wk, 1),
               "hbm_read_bytes_per_launch": int(2 * fk * 1024), "hbm_write_bytes_per_launch": int(wk * 1024), "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
        if valu.get(name):
            v = sum(valu[name]) / len(valu[name])
            rec["valu_wave_insts_per_launch"] = int(v)
            if thr.get(name):
                rec["active_lane_ratio"] = round(sum(thr[name]) / len(thr[name]) / (64.0 * v), 4)
            if waves.get(name):
                rec["waves_per_launch"] = int(sum(waves[name]) / len(waves[name]))
        if frames:
            rec["launches_per_step"] = round(n_launches / frames, 3)
        if k in kernels:      # several instantiations under one short name (the bounce kernels of different iterations): launch-weighted merge
            o, a, b = kernels[k], kernels[k]["launches_sampled"], n_launches
            for key in ("FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "hbm_bytes_per_launch", "valu_wave_insts_per_launch", "waves_per_launch"):
                if key in o and key in rec:
                    rec[key] = type(rec[key])((o[key] * a + rec[key] * b) / (a + b))
            rec["launches_sampled"] = a + b
            if frames:
                rec["launches_per_step"] = round((a + b) / frames, 3)
            rec["symbol"] = o["symbol"] + " | " + name
        kernels[k] = rec
    doc = {}
    if os.path.exists(out_path):
        try:
            doc = json.load(open(out_path))
        except Exception:
            doc = {}
    if "configs" not in doc:
        doc = {"configs": {}}
    doc["note"] = ("reads = 2*FETCH_SIZE*1024 (gfx950 half-count correction), writes = WRITE_SIZE*1024; valu_wave_insts = SQ_INSTS_VALU per launch; "
                   "active_lane_ratio = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU); one entry per bench configuration, separate --pmc passes")
    doc["configs"][config] = {"csrc_sha16": kernel_sources_sha16(), "kernels": kernels}
    json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(doc["configs"][config], indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
