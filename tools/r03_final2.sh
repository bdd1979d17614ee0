#!/bin/bash
# the round's last run: a wide scene-structure sweep (K = 4 batches every third scene: the every-BSDF K = 4 kernels are back at three waves per SIMD), the whole GPU suite, the profile collection
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
DTOF_SCENE_SWEEP=60 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_scene_structures" -p no:cacheprovider > $out/r03_final_sweep.txt 2>&1; rc=$?
tail -3 $out/r03_final_sweep.txt
[ $rc -ne 0 ] && exit 1
bash tools/r03_final.sh
