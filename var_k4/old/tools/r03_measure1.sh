#!/bin/bash
# Round-3 first measurement (one gpurun call): divergent gather microbenchmark, Domino C4 pipeline variants, counters of the fused MESH kernel at 128 spp.
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 300 tools/ubench/gather_nodes > $out/r03_gather_nodes.txt 2>&1 || exit 1
echo "ubench done" ; tail -3 $out/r03_gather_nodes.txt
timeout -k 10 600 python3 tools/ab_env.py domino.xml -- base= it1=,DTOF_INLINE_ITERS=1 it2=,DTOF_INLINE_ITERS=2 split=,DTOF_PIPELINE=split > $out/r03_domino_variants.txt 2>&1 || exit 1
cat $out/r03_domino_variants.txt
python3 - > $out/r03_domino_stats.txt 2>&1 <<'EOF'
import os, sys
sys.path.insert(0, os.getcwd())
import mitsuba3dopplertof_amd as mi
for env in ({}, {"DTOF_INLINE_ITERS": "1"}):
    os.environ.update(env)
    sc = mi.load_file("scenes/domino.xml")
    sc.render(seed=0, spp=128); sc.render(seed=0, spp=128)
    print(env, sc.last_stats)
EOF
cat $out/r03_domino_stats.txt
cd /tmp
rocprofv3 --list-avail > $out/r03_list_avail.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/r03_pmc_a -- python3 $root/tools/time_c2.py domino.xml 128 > $out/r03_pmc_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/r03_pmc_b -- python3 $root/tools/time_c2.py domino.xml 128 > $out/r03_pmc_b.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/r03_pmc_c -- python3 $root/tools/time_c2.py domino.xml 128 > $out/r03_pmc_c.log 2>&1 || exit 1
cd $root
python3 tools/pmc_counters.py $out/r03_pmc_a $out/r03_pmc_b $out/r03_pmc_c > $out/r03_pmc_domino_fused.txt
cat $out/r03_pmc_domino_fused.txt
cd /tmp
rocprofv3 --pmc SQ_WAVES TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_GATE_EN2_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --kernel-trace --output-format csv -d $out/r03_pmc_d -- python3 $root/tools/time_c2.py domino.xml 128 > $out/r03_pmc_d.log 2>&1
cd $root
python3 tools/pmc_counters.py $out/r03_pmc_d > $out/r03_pmc_domino_fused_ta.txt 2>&1
cat $out/r03_pmc_domino_fused_ta.txt
