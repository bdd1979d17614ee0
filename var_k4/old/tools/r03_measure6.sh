#!/bin/bash
# Round-3 sixth run: the default build (per-lane triangle loops, 16 resident waves): C5 with 12 / 16 waves, then the whole GPU test suite.
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
for w in 12 16; do
  DTOF_RESIDENT=$w python3 bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > $out/r03_bench_c5_res$w.json 2> $out/r03_bench_c5_res$w.err || exit 1
  echo "c5 resident $w:"; cut -c1-260 $out/r03_bench_c5_res$w.json
done
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/r03_gpu_suite.txt 2>&1; rc=$?
tail -8 $out/r03_gpu_suite.txt
exit $rc
