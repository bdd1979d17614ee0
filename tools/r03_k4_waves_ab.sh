#!/bin/bash
# K = 4 offset batches on the every-BSDF scenes: the build (three waves per SIMD again) against the same source with the two-wave exception (var_k4/libdtof_k4_two_waves.so)
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
echo "# two waves per SIMD for SPEC && K > 1 (the state before the initialisation fix)" > $out/r03_k4_waves_ab.txt
DTOF_LIB=$root/var_k4/libdtof_k4_two_waves.so timeout -k 10 300 python3 tools/time_k4.py >> $out/r03_k4_waves_ab.txt 2>/dev/null || exit 1
echo "# three waves per SIMD (the build)" >> $out/r03_k4_waves_ab.txt
timeout -k 10 300 python3 tools/time_k4.py >> $out/r03_k4_waves_ab.txt 2>/dev/null || exit 1
cat $out/r03_k4_waves_ab.txt
