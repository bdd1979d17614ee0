#!/bin/bash
# Round-3 fifth run: cooperative triangle loops -- timing on Domino / Cornell boxes / mesh room, then the whole GPU test suite.
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "domino or c4 or boxes" > $out/r03_coop_parity.txt 2>&1; rc=$?
tail -5 $out/r03_coop_parity.txt
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python3 tools/ab_env.py domino.xml -- res12=,DTOF_RESIDENT=12 res16=,DTOF_RESIDENT=16 split=,DTOF_PIPELINE=split > $out/r03_coop_ab.txt 2>&1 || exit 1
cat $out/r03_coop_ab.txt
timeout -k 10 200 python3 tools/time_scenes.py > $out/r03_coop_scenes.txt 2>&1
tail -25 $out/r03_coop_scenes.txt
timeout -k 10 200 python3 tools/time_mesh.py > $out/r03_coop_mesh.txt 2>&1
tail -3 $out/r03_coop_mesh.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/r03_gpu_suite.txt 2>&1; rc=$?
tail -15 $out/r03_gpu_suite.txt
exit $rc
