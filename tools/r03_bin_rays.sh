#!/bin/bash
# Round-3 run: ray binning of the split pipeline (k_bin_rays) -- the invariance test, then the mesh room under each key setting.
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_meshes.py -x -q -m gpu > $out/r03_bin_tests.txt 2>&1 || { tail -30 $out/r03_bin_tests.txt; exit 1; }
tail -3 $out/r03_bin_tests.txt
: > $out/r03_bin_rays_ab.txt
for setting in 0 0,2 0,3 1,3 2,3 3,3 2,2 3,0 2,0; do
    echo "DTOF_BIN_RAYS=$setting" >> $out/r03_bin_rays_ab.txt
    DTOF_BIN_RAYS=$setting timeout -k 10 200 python3 tools/time_mesh.py >> $out/r03_bin_rays_ab.txt 2>&1 || exit 1
done
echo "depth 8:" >> $out/r03_bin_rays_ab.txt
cat $out/r03_bin_rays_ab.txt
