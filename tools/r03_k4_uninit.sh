#!/bin/bash
# Round-3 run: what is at fault when the K = 4 every-BSDF fused kernels go wrong at three waves per SIMD?  Four builds of dtof_kernels.hip (var_k4/, not in the tree):
#   v1 = the shipped source (2 waves for SPEC && K > 1) with -ftrivial-auto-var-init=pattern: every automatic variable without an initialiser starts as 0xAA..; a read of one shows
#   v2 = 3 waves for those instantiations (the configuration that failed), default flags;  v3 = v2 + -ftrivial-auto-var-init=zero;  v4 = v2 + -ftrivial-auto-var-init=pattern
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
sel="random_scene_structures or batched_offsets or resident_stage_with_the_every or both_pipelines_reproduce"
for v in v1 v2 v3 v4; do
    echo "== $v"
    DTOF_LIB=$root/var_k4/libdtof_$v.so timeout -k 10 420 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "$sel" -p no:cacheprovider > $out/r03_k4_$v.txt 2>&1
    echo "rc=$?"; grep -E "^FAILED|passed|failed" $out/r03_k4_$v.txt | cut -c1-220 | tail -12
done
