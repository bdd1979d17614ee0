#!/bin/bash
# sixth run: the failing tree with the explicit initialisers (f1: default flags; f1p: everything else pattern-initialised), then the bench lines of THIS build (does the fix cost anything?)
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cd $root/var_k4/old
for v in f1 f1p; do
    echo "== old tree (3 waves), libdtof_$v.so"
    DTOF_LIB=$root/var_k4/old/mitsuba3dopplertof_amd/libdtof_$v.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k random_scene_structures -p no:cacheprovider > $out/r03_k4f_$v.txt 2>&1
    echo "rc=$?"; grep -E "^FAILED|passed|failed|^E   +assert" $out/r03_k4f_$v.txt | cut -c1-200 | tail -6
done
cd $root
for c in c2 c4 c5; do
    timeout -k 10 300 python3 bench.py --config $c --no-extra --no-cpu-baseline > $out/r03_fix_bench_$c.json 2> $out/r03_fix_bench_$c.err || exit 1
    python3 -c "import json,sys; d=json.load(open('$out/r03_fix_bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline'].get('kernel_ms_events'))"
done
