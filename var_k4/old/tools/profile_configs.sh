#!/bin/bash
# Bench line + rocprofv3 kernel statistics of the other SURVEY 8(d) configs (c1, c3, c4, c5) and of the mesh workload, on one GPU
# (run through gpurun from the repo root):   tools/profile_configs.sh r01      -> gpurun_out/<tag>_cfg_*  (collect_profiles.sh copies them)
set -u
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
for c in c1 c3 c4 c5; do
    case $c in c4) steps="--steps 5 --warmup 1";; c5) steps="--steps 3 --warmup 1";; *) steps="--steps 20 --warmup 3";; esac
    python3 bench.py --config $c --no-cpu-baseline $steps > "$out/${tag}_cfg_${c}_bench.json" 2> "$out/${tag}_cfg_${c}_bench.err"
    cd /tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_cfg_${c}_stats" -o "$tag" -- python3 "$root/bench.py" --config $c --no-cpu-baseline $steps \
        > /dev/null 2> "$out/${tag}_cfg_${c}_stats.log"
    cd "$root"
done
python3 tools/time_mesh.py > "$out/${tag}_cfg_mesh_time.txt" 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_cfg_mesh_stats" -o "$tag" -- python3 "$root/tools/time_mesh.py" > /dev/null 2> "$out/${tag}_cfg_mesh_stats.log"
cd "$root"
for c in c1 c3 c4 c5; do echo "== $c"; cut -c1-400 "$out/${tag}_cfg_${c}_bench.json"; done
cat "$out/${tag}_cfg_mesh_time.txt" | tail -1
