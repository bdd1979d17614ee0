#!/bin/bash
# fifth run: which variable?  pattern-initialised builds of the failing tree with one group of declarations zeroed by hand (p1 cand, p2 rcur, p3 the ray / state registers, p4 the instance memo)
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cd $root/var_k4/old
for v in ${VARIANTS:-p1 p2 p3 p4}; do
    echo "== old tree, libdtof_$v.so"
    DTOF_LIB=$root/var_k4/old/mitsuba3dopplertof_amd/libdtof_$v.so timeout -k 10 300 python3 -m pytest "tests/test_gpu_parity.py::test_random_scene_structures[1]" -q -m gpu -p no:cacheprovider > $out/r03_k4e_$v.txt 2>&1
    echo "rc=$?"; grep -E "^FAILED|passed|failed|^E   +Assert|^E   +assert" $out/r03_k4e_$v.txt | cut -c1-200 | tail -6
done
