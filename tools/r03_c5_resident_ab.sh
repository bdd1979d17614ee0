#!/bin/bash
# C5 (Domino 1024^2 x 512 spp, four films) under DTOF_RESIDENT = 8 / 12 / 16 and C4 under 8 / 12 / 16: is the wave count of the resident stage still right on the final kernels?
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
: > $out/r03_c5_resident_ab.txt
for c in c5 c4; do
  for w in 8 12 16; do
    DTOF_RESIDENT=$w timeout -k 10 200 python3 bench.py --config $c --no-extra --no-cpu-baseline --steps 3 --warmup 1 > $out/tmp_bench.json 2>/dev/null || exit 1
    python3 -c "import json; d=json.load(open('$out/tmp_bench.json')); print('$c DTOF_RESIDENT=$w  %.2f ms per step  %.0f Mpaths/s' % (d['ms_per_step'], d['value']))" >> $out/r03_c5_resident_ab.txt
  done
done
rm -f $out/tmp_bench.json; cat $out/r03_c5_resident_ab.txt
