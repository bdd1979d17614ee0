#!/bin/bash
# One parametrised runner for the measurement sessions of a round (run through gpurun from the repo root; everything lands under gpurun_out/):
#
#   tools/gpu_session.sh suite [pytest args]                 the GPU test suite                                   -> gpurun_out/<tag>_suite.txt
#   tools/gpu_session.sh pattern [pytest args]               ... against the pattern-initialised build            -> gpurun_out/<tag>_pattern_suite.txt
#                                                            (make -C mitsuba3dopplertof_amd/csrc pattern: every uninitialised automatic variable is a NaN pattern)
#   tools/gpu_session.sh sweep N                             the random scene / parameter sweeps, N draws per block
#   tools/gpu_session.sh bench CONFIG [bench.py args]        one bench line                                       -> gpurun_out/<tag>_bench_CONFIG.json
#   tools/gpu_session.sh env CONFIG VAR v1 v2 ...            the same bench line under VAR=v1, VAR=v2, ...         -> gpurun_out/<tag>_env_CONFIG_VAR.txt
#                                                            (what the round-3 one-off scripts did: DTOF_RESIDENT = 8 12 16, DTOF_BATCH_LANES, DTOF_INLINE_ITERS ...)
#   tools/gpu_session.sh ab SCENE SPP name=[LIB.so][,ENV=VAL]...   interleaved A/B of library variants (tools/ab_env.py; variants: make -C ... variant NAME=x DEFS=...)
#
# TAG (environment, default r04) prefixes the output files.
set -u
tag=${TAG:-r04}; root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp
what=${1:-}; shift || true
case "$what" in
  suite)   timeout -k 10 1100 python3 -m pytest tests -q -m gpu -p no:cacheprovider "$@" > "$out/${tag}_suite.txt" 2>&1; echo "rc=$?"; grep -E "^FAILED|passed|failed" "$out/${tag}_suite.txt" | cut -c1-220 | tail -20 ;;
  pattern) [ -f mitsuba3dopplertof_amd/libdtof_pattern.so ] || { echo "build it first: make -C mitsuba3dopplertof_amd/csrc pattern"; exit 1; }
           DTOF_LIB=$root/mitsuba3dopplertof_amd/libdtof_pattern.so timeout -k 10 1100 python3 -m pytest tests -q -m gpu -p no:cacheprovider "$@" > "$out/${tag}_pattern_suite.txt" 2>&1
           echo "rc=$?"; grep -E "^FAILED|passed|failed" "$out/${tag}_pattern_suite.txt" | cut -c1-220 | tail -20 ;;
  sweep)   DTOF_SCENE_SWEEP=${1:-100} DTOF_SWEEP=${1:-100} timeout -k 10 1100 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "random" > "$out/${tag}_sweep.txt" 2>&1
           echo "rc=$?"; tail -3 "$out/${tag}_sweep.txt" ;;
  bench)   c=$1; shift; timeout -k 10 600 python3 bench.py --config "$c" "$@" > "$out/${tag}_bench_${c}.json" 2> "$out/${tag}_bench_${c}.err" || { tail -5 "$out/${tag}_bench_${c}.err"; exit 1; }
           python3 -c "import json; d=json.load(open('$out/${tag}_bench_${c}.json')); print('$c  %.4f ms per step (min %.4f)  %.0f Mpaths/s' % (d['ms_per_step'], d['ms_per_step_min'], d['value']))" ;;
  env)     c=$1; var=$2; shift 2; : > "$out/${tag}_env_${c}_${var}.txt"
           for v in "$@"; do
             env "$var=$v" timeout -k 10 400 python3 bench.py --config "$c" --no-extra --no-cpu-baseline > "$out/tmp_bench.json" 2>/dev/null || exit 1
             python3 -c "import json; d=json.load(open('$out/tmp_bench.json')); print('$c $var=$v  %.4f ms per step (min %.4f)  %.0f Mpaths/s' % (d['ms_per_step'], d['ms_per_step_min'], d['value']))" >> "$out/${tag}_env_${c}_${var}.txt"
           done
           rm -f "$out/tmp_bench.json"; cat "$out/${tag}_env_${c}_${var}.txt" ;;
  ab)      scene=$1; spp=$2; shift 2; python3 tools/ab_env.py "$scene" "$spp" -- "$@" | tee "$out/${tag}_ab_$(basename "$scene" .xml).txt" ;;
  *)       sed -n 2,16p "$0"; exit 1 ;;
esac
