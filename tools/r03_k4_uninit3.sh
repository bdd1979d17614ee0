#!/bin/bash
# third run: the RESIDENT every-BSDF K = 4 kernels at 12 waves per CU (3 per SIMD; the host forces 8 in the shipped build): v5 = shipped kernels,
# v6 = -ftrivial-auto-var-init=zero, v7 = =pattern
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
export DTOF_SCENE_SWEEP=40
for v in v5 v6 v7; do
    echo "== $v"
    DTOF_LIB=$root/var_k4/libdtof_$v.so timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "random_scene_structures or resident_stage_with_the_every or batched_offsets" -p no:cacheprovider > $out/r03_k4c_$v.txt 2>&1
    echo "rc=$?"; grep -E "^FAILED|passed|failed|^E  " $out/r03_k4c_$v.txt | cut -c1-260 | tail -14
done
