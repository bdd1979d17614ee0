#!/bin/bash
# long sweeps on the final build: 4 x 300 random scene structures (both pipelines, K = 4 batches every third scene) and 200 random parameter combinations
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
DTOF_SCENE_SWEEP=300 DTOF_SWEEP=200 timeout -k 10 1100 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "random_scene_structures or random_parameter_combinations" -p no:cacheprovider > $out/r03_long_sweep.txt 2>&1
echo "rc=$?"; grep -E "^FAILED|passed|failed" $out/r03_long_sweep.txt | cut -c1-220 | tail -10
