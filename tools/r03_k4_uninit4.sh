#!/bin/bash
# fourth run: the tree of commit 5f11a27 (where the K = 4 films went wrong) rebuilt at THREE waves per SIMD in var_k4/old: default flags, then zero- and pattern-initialised automatic variables
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cd $root/var_k4/old
for v in "" _zero _pattern; do
    echo "== old tree, libdtof$v.so"
    DTOF_LIB=$root/var_k4/old/mitsuba3dopplertof_amd/libdtof$v.so timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "random_scene_structures" -p no:cacheprovider > $out/r03_k4d$v.txt 2>&1
    echo "rc=$?"; grep -E "^FAILED|passed|failed|^E   +Assert|^E   +assert" $out/r03_k4d$v.txt | cut -c1-300 | tail -14
done
