#!/bin/bash
# Collects the round's measurement evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# 1. plain bench line                      -> gpurun_out/<tag>_bench_n1.json
# 2. rocprofv3 --kernel-trace --stats      -> gpurun_out/<tag>_stats/   (+ the bench line measured under the profiler)
# 3. three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU; never combined with other trace domains)
# Afterwards, on the build host:  tools/collect_profiles.sh <tag>   copies the summaries into profiles/.
set -u
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/${tag}_bench_n1.json" 2> "$out/${tag}_bench_n1.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o "$tag" -- python3 "$root/bench.py" --no-cpu-baseline \
    > "$out/${tag}_bench_n1_under_rocprof.json" 2> "$out/${tag}_stats.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/${tag}_pmc_fetch" -- python3 "$root/bench.py" --no-cpu-baseline --steps 3 --warmup 1 \
    > /dev/null 2> "$out/${tag}_pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/${tag}_pmc_write" -- python3 "$root/bench.py" --no-cpu-baseline --steps 3 --warmup 1 \
    > /dev/null 2> "$out/${tag}_pmc_write.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d "$out/${tag}_pmc_valu" -- python3 "$root/bench.py" --no-cpu-baseline --steps 3 --warmup 1 \
    > /dev/null 2> "$out/${tag}_pmc_valu.log"
cd "$root"
cat "$out/${tag}_bench_n1.json"
ls "$out/${tag}_stats" | head
