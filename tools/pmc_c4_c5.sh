#!/bin/bash
# SQ / memory-path counters of the two Domino bench configurations side by side (development helper; run through gpurun from the repo root):  tools/pmc_c4_c5.sh [tag]
tag=${1:-r04}; root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
for c in c4 c5; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/${tag}_pmcx1_$c -- python3 $root/bench.py --config $c --no-extra --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2> $out/${tag}_pmcx1_$c.log || exit 1
  rocprofv3 --pmc SQ_WAVES TA_TA_BUSY_sum TD_TD_BUSY_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/${tag}_pmcx2_$c -- python3 $root/bench.py --config $c --no-extra --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2> $out/${tag}_pmcx2_$c.log || exit 1
  rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out/${tag}_pmcx3_$c -- python3 $root/bench.py --config $c --no-extra --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2> $out/${tag}_pmcx3_$c.log || exit 1
  cd $root; echo "==== $c"; python3 tools/pmc_counters.py $out/${tag}_pmcx1_$c $out/${tag}_pmcx2_$c $out/${tag}_pmcx3_$c | grep -A26 "^k_shade"; cd /tmp
done
