#!/bin/bash
# SQ / memory-path counters of the ray kernels on the mesh room (tools/time_mesh.py; development helper, run through gpurun from the repo root):  tools/pmc_mesh.sh [tag]
# Three separate --pmc passes (never combined with other trace domains), summarised per kernel by tools/pmc_counters.py.
tag=${1:-r05}; root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/${tag}_pmcm1 -- python3 $root/tools/time_mesh.py > $out/${tag}_pmcm1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES TA_TA_BUSY_sum TD_TD_BUSY_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/${tag}_pmcm2 -- python3 $root/tools/time_mesh.py > $out/${tag}_pmcm2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $out/${tag}_pmcm3 -- python3 $root/tools/time_mesh.py > $out/${tag}_pmcm3.log 2>&1 || exit 1
# (a pass of TCP_* counters -- TCP_TOTAL_CACHE_ACCESSES_sum, TCP_PENDING_STALL_CYCLES_sum ... -- aborts rocprofv3 on this image and leaves the run hanging: not collected)
cd $root; python3 tools/pmc_counters.py $out/${tag}_pmcm1 $out/${tag}_pmcm2 $out/${tag}_pmcm3 | grep -A34 "^k_trace\|^k_shadow" | cut -c1-200 | tee $out/${tag}_pmc_mesh.txt
