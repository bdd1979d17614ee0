#!/bin/bash
# per-kernel durations of the mesh room with and without ray binning (rocprofv3 --kernel-trace --stats)
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
for setting in 0 0,3 2,3; do
    export DTOF_BIN_RAYS=$setting
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03_bin_stats_$setting -o bin -- python3 $root/tools/time_mesh.py > $out/r03_bin_stats_$setting.log 2>&1 || exit 1
    f=$(find $out/r03_bin_stats_$setting -name '*kernel_stats.csv' | head -1)
    echo "DTOF_BIN_RAYS=$setting"; cut -d, -f1-5 $f | cut -c1-200 | head -8
done
