#!/bin/bash
# SQ / cache counter passes over the Domino frame (development helper):  tools/pmc_domino.sh [scene.xml [spp]]
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
s=${1:-domino.xml}; spp=${2:-16}
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/pmc_d1 -- python3 $root/tools/time_c2.py $s $spp > $out/pmc_d1.log 2>&1
rocprofv3 --pmc SQ_WAVES TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/pmc_d2 -- python3 $root/tools/time_c2.py $s $spp > $out/pmc_d2.log 2>&1
rocprofv3 --pmc FETCH_SIZE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $out/pmc_d3 -- python3 $root/tools/time_c2.py $s $spp > $out/pmc_d3.log 2>&1
cd $root
tail -1 $out/pmc_d1.log
python3 tools/pmc_counters.py $out/pmc_d1 $out/pmc_d2 $out/pmc_d3 | grep -A18 "k_trace\|k_shadow"
