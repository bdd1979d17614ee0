#!/bin/bash
# Round-3 eighth run: the quantised 4-wide BVH (libdtof_bvh4.so) -- parity, then A/B against the binary nodes on the mesh room, Domino and the Cornell scenes;
# C2 after restoring the call sites of the default build.
set -u
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
DTOF_LIB=$root/mitsuba3dopplertof_amd/libdtof_bvh4.so timeout -k 10 600 python3 -m pytest tests/test_meshes.py -x -q -m gpu > $out/r03_q4_parity.txt 2>&1; rc=$?
tail -5 $out/r03_q4_parity.txt
[ $rc -ne 0 ] && exit 1
for lib in libdtof.so libdtof_bvh4.so; do echo "== $lib"; DTOF_LIB=$root/mitsuba3dopplertof_amd/$lib timeout -k 10 200 python3 tools/time_mesh.py 2>&1 | tail -1; done > $out/r03_q4_mesh.txt
cat $out/r03_q4_mesh.txt
timeout -k 10 400 python3 tools/ab_env.py domino.xml -- bin= q4=mitsuba3dopplertof_amd/libdtof_bvh4.so binsplit=,DTOF_PIPELINE=split q4split=mitsuba3dopplertof_amd/libdtof_bvh4.so,DTOF_PIPELINE=split > $out/r03_q4_domino.txt 2>&1 || exit 1
cat $out/r03_q4_domino.txt
for lib in libdtof.so libdtof_bvh4.so; do echo "== $lib"; DTOF_LIB=$root/mitsuba3dopplertof_amd/$lib timeout -k 10 200 python3 tools/time_scenes.py 2>&1 | grep -v amdgpu; done > $out/r03_q4_scenes.txt
cat $out/r03_q4_scenes.txt
timeout -k 10 300 python3 tools/ab_env.py cornell_wall.xml -- old=tools/ab/pre_coop.so new= > $out/r03_callpath_ab_wall2.txt 2>&1 || exit 1
cat $out/r03_callpath_ab_wall2.txt
