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
python3 tools/pmc_summary.py "$out/${tag}_pmc_fetch" "$out/${tag}_pmc_write" profiles/roofline_traffic.json
ls -la profiles
