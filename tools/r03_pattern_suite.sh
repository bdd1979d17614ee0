#!/bin/bash
# the whole GPU suite against a build of dtof_kernels.hip with -ftrivial-auto-var-init=pattern (var_k4/libdtof_pattern.so: tools/build_pattern_variant.sh): every automatic variable
# without an initialiser starts as 0xAA.. / NaN, so any result that depends on one shows
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
DTOF_LIB=$root/var_k4/libdtof_pattern.so DTOF_SCENE_SWEEP=40 timeout -k 10 1000 python3 -m pytest tests -q -m gpu -p no:cacheprovider > $out/r03_pattern_suite.txt 2>&1
echo "rc=$?"; grep -E "^FAILED|passed|failed" $out/r03_pattern_suite.txt | cut -c1-220 | tail -20
