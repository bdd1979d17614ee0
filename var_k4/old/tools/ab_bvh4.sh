set -u
out=gpurun_out/r02g_bvh4.txt
python -m pytest tests/test_gpu_parity.py tests/test_meshes.py tests/test_spheres.py -m gpu -x -q -k "domino or every_lane or mesh or blas or sphere or edge" > gpurun_out/r02g_tests.log 2>&1; tail -3 gpurun_out/r02g_tests.log
python tools/ab_env.py domino.xml 16 -- bvh4= bvh2=mitsuba3dopplertof_amd/libdtof_bvh2.so > $out 2>&1
python tools/ab_env.py cornell_boxes.xml 64 -- bvh4= bvh2=mitsuba3dopplertof_amd/libdtof_bvh2.so >> $out 2>&1
for l in "" mitsuba3dopplertof_amd/libdtof_bvh2.so; do DTOF_LIB=${l:+$PWD/$l} python tools/time_mesh.py >> $out 2>&1; done
python tools/traversal_stats.py domino.xml 4 >> $out 2>&1
grep -v amdgpu.ids $out
