#!/bin/bash
# Instruction-cache counters of the dominant kernels (development helper; run through gpurun from the repo root):  tools/pmc_icache.sh [tag] [configs ...]
# The resident Domino kernel is ~70 KB of code for 16 waves per CU that sit in different parts of it (bounce code, TLAS loop, leaf code): is instruction fetch part of its waiting?
tag=${1:-r05}; shift || true; configs=${*:-c4 c2}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
for c in $configs; do
  rocprofv3 --pmc SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $out/${tag}_pmci1_$c -- python3 $root/bench.py --config $c --no-extra --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2> $out/${tag}_pmci1_$c.log || echo "pass 1 failed for $c (counter names?)"
  rocprofv3 --pmc SQ_WAVES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/${tag}_pmci2_$c -- python3 $root/bench.py --config $c --no-extra --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2> $out/${tag}_pmci2_$c.log || echo "pass 2 failed for $c"
  cd $root; echo "==== $c"; python3 tools/pmc_counters.py $out/${tag}_pmci1_$c $out/${tag}_pmci2_$c | grep -A12 "^k_shade"; cd /tmp
done
