#!/bin/bash
# Copies what tools/profile_round.sh left under gpurun_out/ into profiles/ (tracked):  tools/collect_profiles.sh r01
set -eu
tag=${1:-r01}
out=gpurun_out
mkdir -p profiles
cp "$out/${tag}_bench_n1.json" "profiles/${tag}_bench_n1.json"
cp "$out/${tag}_bench_n1_under_rocprof.json" "profiles/${tag}_bench_n1_under_rocprof.json"
stats=$(find "$out/${tag}_stats" -name '*kernel_stats.csv' | head -1)
cp "$stats" "profiles/${tag}_bench_n1_kernel_stats.csv"
if [ -d "$out/${tag}_pmc_valu" ]; then python3 tools/pmc_summary.py "$out/${tag}_pmc_fetch" "$out/${tag}_pmc_write" profiles/roofline_traffic.json "$out/${tag}_pmc_valu"; else python3 tools/pmc_summary.py "$out/${tag}_pmc_fetch" "$out/${tag}_pmc_write" profiles/roofline_traffic.json; fi
ls -la profiles
# the other configs (tools/profile_configs.sh), when present
for c in c1 c3 c4 c5; do
    if [ -f "$out/${tag}_cfg_${c}_bench.json" ]; then
        cp "$out/${tag}_cfg_${c}_bench.json" "profiles/${tag}_bench_${c}.json"
        s=$(find "$out/${tag}_cfg_${c}_stats" -name '*kernel_stats.csv' | head -1)
        [ -n "$s" ] && cp "$s" "profiles/${tag}_bench_${c}_kernel_stats.csv"
    fi
done
if [ -f "$out/${tag}_cfg_mesh_time.txt" ]; then
    tail -1 "$out/${tag}_cfg_mesh_time.txt" > "profiles/${tag}_mesh_522k_tris.txt"
    s=$(find "$out/${tag}_cfg_mesh_stats" -name '*kernel_stats.csv' | head -1)
    [ -n "$s" ] && cp "$s" "profiles/${tag}_mesh_522k_tris_kernel_stats.csv"
fi
ls -la profiles
