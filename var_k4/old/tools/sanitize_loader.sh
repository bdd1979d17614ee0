#!/bin/bash
# CPU-only sanitizer run of the host side of the library (XML / obj / ply loaders, BVH builder, blob packing):
#   tools/sanitize_loader.sh [seed [count]]
# builds tests/dev/asan_loader.cpp with -fsanitize=address,undefined (host code only, no GPU needed), writes a corpus of damaged
# scene and mesh files with tests/dev/fuzz_corpus.py and feeds it through.  Prints every distinct sanitizer report (none expected).
set -eu
seed=${1:-1}; count=${2:-4000}
root=$(cd "$(dirname "$0")/.." && pwd)
work=$(mktemp -d /tmp/dtof_asan.XXXXXX)
cd "$root/mitsuba3dopplertof_amd/csrc"
/opt/rocm/bin/hipcc -x hip --offload-host-only -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -I. \
    scene_loader.cpp scene_build.cpp mesh_io.cpp image_io.cpp "$root/tests/dev/asan_loader.cpp" -o "$work/loader" -lz
python3 "$root/tests/dev/fuzz_corpus.py" "$seed" "$count" "$work/corpus"
cd "$work"
ls corpus/*.xml | xargs -n 200 ./loader > log.txt 2>&1 || true
grep "^ok" log.txt | awk '{o+=$2; e+=$4} END {print "loaded", o, "rejected", e}'
grep -v "^ok" log.txt | grep -v "^SUMMARY" | cut -c1-200 | sort | uniq -c | sort -rn | head -20
rm -rf "$work"
