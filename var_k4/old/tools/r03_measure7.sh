#!/bin/bash
# Round-3 seventh run: does the restructured call path (trace_rays, wave-uniform call sites) cost anything with the shared loops compiled out?  A/B against the
# library of commit df89278; mesh-room counters (is k_trace bound by the vector-memory path too?); then the GPU suite.
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 300 python3 tools/ab_env.py domino.xml -- old12=tools/ab/pre_coop.so,DTOF_RESIDENT=12 new12=,DTOF_RESIDENT=12 new16=,DTOF_RESIDENT=16 > $out/r03_callpath_ab_domino.txt 2>&1 || exit 1
cat $out/r03_callpath_ab_domino.txt
timeout -k 10 300 python3 tools/ab_env.py cornell_boxes.xml 64 -- old=tools/ab/pre_coop.so,resx=512 new= > $out/r03_callpath_ab_boxes.txt 2>&1 || exit 1
cat $out/r03_callpath_ab_boxes.txt
timeout -k 10 300 python3 tools/ab_env.py cornell_wall.xml -- old=tools/ab/pre_coop.so new= > $out/r03_callpath_ab_wall.txt 2>&1 || exit 1
cat $out/r03_callpath_ab_wall.txt
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r03_pmc_mesh_a -- python3 $root/tools/time_mesh.py > $out/r03_pmc_mesh_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/r03_pmc_mesh_b -- python3 $root/tools/time_mesh.py > $out/r03_pmc_mesh_b.log 2>&1 || exit 1
cd $root
python3 tools/pmc_counters.py $out/r03_pmc_mesh_a $out/r03_pmc_mesh_b > $out/r03_pmc_mesh_room.txt
grep -A 16 "k_trace\|k_shadow" $out/r03_pmc_mesh_room.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/r03_gpu_suite.txt 2>&1; rc=$?
tail -8 $out/r03_gpu_suite.txt
exit $rc
