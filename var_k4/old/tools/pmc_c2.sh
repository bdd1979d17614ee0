#!/bin/bash
# SQ counter passes over tools/time_c2.py (development helper):  tools/pmc_c2.sh [scene.xml]
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_c1 -- python3 $root/tools/time_c2.py ${1:-cornell_wall.xml} > $out/pmc_c1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_c2 -- python3 $root/tools/time_c2.py ${1:-cornell_wall.xml} > $out/pmc_c2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d $out/pmc_c3 -- python3 $root/tools/time_c2.py ${1:-cornell_wall.xml} > $out/pmc_c3.log 2>&1
cd $root
python3 tools/pmc_counters.py $out/pmc_c1 $out/pmc_c2 $out/pmc_c3 | grep -A26 "k_shade"
