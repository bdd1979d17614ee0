#!/bin/bash
# Collects the round's measurement evidence on the GPU box (run through gpurun from the repo root):   tools/profile_round.sh r05 [configs ...]
# For the reference's own example (c1), the headline config (c2), the same frame through the wavefront pipeline (c2_wavefront = bench.py --config c2 --pipeline split),
# the 256-spp Cornell config (c3) and the two Domino configs (c4, c5):
#   1. rocprofv3 --kernel-trace --stats of the bench command -> gpurun_out/<tag>_stats_<cfg>/  (+ the bench line measured under the profiler)
#   2. three separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES; never combined with other trace domains)
#      summarised into profiles/roofline_traffic.json (tools/pmc_summary.py; stamped with the hash of the kernel sources)
#   3. LAST, the plain bench lines (they read the counters of step 2)  -> gpurun_out/<tag>_profiles/<tag>_bench_<cfg>.json, <tag>_bench_n1.json (c2, the headline)
# plus the traversal counters of the Domino frame (needs mitsuba3dopplertof_amd/libdtof_stats.so: make -C mitsuba3dopplertof_amd/csrc stats).
# The profiles/ files written here come back with gpurun's merge only if they sit under gpurun_out/: the script leaves copies there (<tag>_profiles/).
set -u
tag=${1:-r05}; shift || true
configs=${*:-c1 c2 c2_wavefront c3 c4 c5}
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out" "$out/${tag}_profiles"; export TMPDIR=/tmp
steps_of() { case $1 in c1) echo "--steps 100 --warmup 3";; c2|c2_wavefront) echo "--steps 100 --warmup 3";; c3) echo "--steps 30 --warmup 2";; c4) echo "--steps 6 --warmup 1";; c5) echo "--steps 3 --warmup 1";; esac; }
for c in $configs; do
    steps=$(steps_of $c)
    # the --pmc passes render few frames: `frames` = warm-up + the synchronous counter frame + timed steps (pmc_summary.py turns launch counts into launches per step)
    case $c in c1|c2|c2_wavefront) short="--steps 3 --warmup 1"; frames=5;; c3) short="--steps 2 --warmup 1"; frames=4;; c4) short="--steps 2 --warmup 1"; frames=4;; c5) short="--steps 1 --warmup 1"; frames=3;; esac
    key=$c; cfg="--config $c"
    if [ "$c" = c2_wavefront ]; then cfg="--config c2 --pipeline split"; fi
    cd /tmp
    echo "[$c] rocprofv3 --kernel-trace --stats"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats_${c}" -o "$tag" -- python3 "$root/bench.py" $cfg --no-extra --no-cpu-baseline $steps \
        > "$out/${tag}_bench_${c}_under_rocprof.json" 2> "$out/${tag}_stats_${c}.log" || exit 1
    echo "[$c] pmc passes"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/${tag}_pmc_fetch_${c}" -- python3 "$root/bench.py" $cfg --no-extra --no-cpu-baseline $short > /dev/null 2> "$out/${tag}_pmc_fetch_${c}.log" || exit 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/${tag}_pmc_write_${c}" -- python3 "$root/bench.py" $cfg --no-extra --no-cpu-baseline $short > /dev/null 2> "$out/${tag}_pmc_write_${c}.log" || exit 1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d "$out/${tag}_pmc_valu_${c}" -- python3 "$root/bench.py" $cfg --no-extra --no-cpu-baseline $short > /dev/null 2> "$out/${tag}_pmc_valu_${c}.log" || exit 1
    cd "$root"
    python3 tools/pmc_summary.py profiles/roofline_traffic.json $c "$out/${tag}_pmc_fetch_${c}" "$out/${tag}_pmc_write_${c}" "$out/${tag}_pmc_valu_${c}" $frames > "$out/${tag}_pmc_summary_${c}.txt" || exit 1
    s=$(find "$out/${tag}_stats_${c}" -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp "$s" "$out/${tag}_profiles/${tag}_bench_${c}_kernel_stats.csv"
    cp "$out/${tag}_bench_${c}_under_rocprof.json" "$out/${tag}_profiles/"
    echo "== $c counters in place"
done
cp profiles/roofline_traffic.json "$out/${tag}_profiles/" 2>/dev/null
# SKIP_FINAL=1: counters only (a round's evidence does not fit one 20-minute gpurun call: first `SKIP_FINAL=1 tools/profile_round.sh r05 c1 c2 c2_wavefront c3`, copy
# gpurun_out/r05_profiles/roofline_traffic.json back into profiles/, then `tools/profile_round.sh r05 c4 c5`)
if [ -n "${SKIP_FINAL:-}" ]; then exit 0; fi
if [ -f mitsuba3dopplertof_amd/libdtof_stats.so ]; then
    python3 tools/traversal_stats.py domino.xml 128 wave_function_type=rectangular --pipeline fused --json profiles/${tag}_traversal_stats_c4.json > "$out/${tag}_traversal_stats_c4.txt" 2>&1
    cat "$out/${tag}_traversal_stats_c4.txt"
    cp profiles/${tag}_traversal_stats_c4.json "$out/${tag}_profiles/" 2>/dev/null
fi
python3 tools/algorithmic_ops.py > /dev/null 2>&1
cp profiles/roofline_traffic.json profiles/algorithmic_ops.json "$out/${tag}_profiles/" 2>/dev/null
# the bench lines LAST, now that the counters and the algorithmic figures of THIS build are in place (c2 with its extras and the CPU baseline = the headline line;
# the CPU oracle -- brute force over every object -- is timed on the headline workload only)
for c in c1 c3 c4 c5; do
    echo "[$c] bench line"
    python3 bench.py --config $c --no-extra --no-cpu-baseline $(steps_of $c) > "$out/${tag}_profiles/${tag}_bench_${c}.json" 2> "$out/${tag}_bench_${c}.err" || exit 1
    cut -c1-330 "$out/${tag}_profiles/${tag}_bench_${c}.json"
done
echo "[c2] headline bench line"
python3 bench.py > "$out/${tag}_profiles/${tag}_bench_n1.json" 2> "$out/${tag}_bench_n1.err"
cut -c1-1200 "$out/${tag}_profiles/${tag}_bench_n1.json"
