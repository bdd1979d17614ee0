#!/bin/bash
# SQ / cache counter passes over tools/time_mesh.py (development helper):  tools/pmc_mesh.sh n_u n_v
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/pmc_m1 -- python3 $root/tools/time_mesh.py $1 $2 > $out/pmc_m1.log 2>&1
rocprofv3 --pmc SQ_WAVES TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_m2 -- python3 $root/tools/time_mesh.py $1 $2 > $out/pmc_m2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_m3 -- python3 $root/tools/time_mesh.py $1 $2 > $out/pmc_m3.log 2>&1
cd $root
python3 tools/pmc_counters.py $out/pmc_m1 $out/pmc_m2 $out/pmc_m3 | grep -A14 "k_trace\|k_shadow"
