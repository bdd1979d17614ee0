# A/B of the working tree's libdtof.so against tools/ab/base.so (built from HEAD) on the workloads the change can touch; plus the parity tests
set -u
out=gpurun_out/r02l_ab.txt
python -m pytest tests/test_gpu_parity.py tests/test_meshes.py tests/test_spheres.py tests/test_disks.py tests/test_emitters.py -m gpu -x -q > gpurun_out/r02l_tests.log 2>&1; tail -3 gpurun_out/r02l_tests.log
python tools/ab_env.py cornell_wall.xml -- base=tools/ab/base.so new= > $out 2>&1
python tools/ab_env.py domino.xml 16 -- base=tools/ab/base.so new= >> $out 2>&1
python tools/ab_env.py cornell_boxes.xml 64 -- base=tools/ab/base.so new= >> $out 2>&1
for l in tools/ab/base.so ""; do DTOF_LIB=${l:+$PWD/$l} python tools/time_mesh.py >> $out 2>&1; done
grep -v amdgpu.ids $out
