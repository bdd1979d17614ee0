#!/bin/bash
# whole GPU test suite, then the round's profile collection
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/r03_gpu_suite.txt 2>&1; rc=$?
tail -6 $out/r03_gpu_suite.txt
[ $rc -ne 0 ] && exit 1
bash tools/profile_round.sh r03
