#!/bin/bash
# Round-3 third measurement: counters of the resident first-bounce kernel (Domino 128 spp), C5 timing resident vs classic, bench lines c4 / c5.
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/r03_pmc_ra -- python3 $root/tools/time_c2.py domino.xml 128 > $out/r03_pmc_ra.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES TCP_TOTAL_CACHE_ACCESSES_sum SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU TA_TA_BUSY_sum TD_TD_BUSY_sum --kernel-trace --output-format csv -d $out/r03_pmc_rb -- python3 $root/tools/time_c2.py domino.xml 128 > $out/r03_pmc_rb.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/r03_pmc_rc -- python3 $root/tools/time_c2.py domino.xml 128 > $out/r03_pmc_rc.log 2>&1 || exit 1
cd $root
python3 tools/pmc_counters.py $out/r03_pmc_ra $out/r03_pmc_rb $out/r03_pmc_rc > $out/r03_pmc_domino_resident.txt
grep -A 28 "k_shade" $out/r03_pmc_domino_resident.txt
python3 bench.py --config c4 --no-cpu-baseline --steps 5 --warmup 1 > $out/r03_bench_c4.json 2> $out/r03_bench_c4.err || exit 1
cut -c1-300 $out/r03_bench_c4.json
python3 bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > $out/r03_bench_c5.json 2> $out/r03_bench_c5.err || exit 1
cut -c1-300 $out/r03_bench_c5.json
DTOF_RESIDENT=0 python3 bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > $out/r03_bench_c5_classic.json 2> $out/r03_bench_c5_classic.err || exit 1
cut -c1-300 $out/r03_bench_c5_classic.json
