#!/bin/bash
# Round-3 fourth run: 16 waves per resident block (128 VGPRs) vs 12, then the whole GPU test suite.
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 400 python3 tools/ab_env.py domino.xml -- res12= res16=,DTOF_RESIDENT=16 > $out/r03_res16_ab.txt 2>&1 || exit 1
cat $out/r03_res16_ab.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/r03_gpu_suite.txt 2>&1; rc=$?
tail -15 $out/r03_gpu_suite.txt
exit $rc
