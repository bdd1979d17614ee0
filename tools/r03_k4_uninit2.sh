#!/bin/bash
# second run of tools/r03_k4_uninit.sh: the scene-structure sweep alone, 100 scenes per block
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
export DTOF_SCENE_SWEEP=100
for v in v2 v1 v3 v4; do
    echo "== $v"
    DTOF_LIB=$root/var_k4/libdtof_$v.so timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "random_scene_structures" -p no:cacheprovider > $out/r03_k4b_$v.txt 2>&1
    echo "rc=$?"; grep -E "^FAILED|passed|failed|^E  " $out/r03_k4b_$v.txt | cut -c1-260 | tail -14
done
