#!/bin/bash
# Round-3 second measurement: the resident first-bounce kernel (LDS-staged TLAS) -- parity on the Domino tests, A/B timing, fixed gather microbenchmark.
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "domino or c4" > $out/r03_res_parity.txt 2>&1; rc=$?
tail -5 $out/r03_res_parity.txt
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python3 tools/ab_env.py domino.xml -- off=,DTOF_RESIDENT=0 res12= res8=,DTOF_RESIDENT=8 > $out/r03_res_ab.txt 2>&1 || exit 1
cat $out/r03_res_ab.txt
timeout -k 10 300 tools/ubench/gather_nodes > $out/r03_gather_nodes2.txt 2>&1 || exit 1
grep "64 KB" $out/r03_gather_nodes2.txt
