#!/bin/bash
# Builds var_k4/libdtof_pattern.so (build container; ~4 min): dtof_kernels.hip compiled with -ftrivial-auto-var-init=pattern, linked with the objects of the regular build
# (run `make -C mitsuba3dopplertof_amd/csrc` first).  tools/r03_pattern_suite.sh runs the GPU suite against it; the directory is not tracked (*.so is git-ignored) but travels with gpurun.
set -eu
root=$(cd "$(dirname "$0")/.." && pwd); c=$root/mitsuba3dopplertof_amd/csrc; mkdir -p $root/var_k4
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-parameter -fno-slp-vectorize -ftrivial-auto-var-init=${1:-pattern} \
    -c $c/dtof_kernels.hip -o $root/var_k4/k.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $root/var_k4/libdtof_${1:-pattern}.so $root/var_k4/k.o $c/dtof_render.o $c/scene_loader.o $c/scene_build.o $c/mesh_io.o $c/image_io.o -lz
rm -f $root/var_k4/k.o; ls -la $root/var_k4
