# development helper: block size of the unstaged traversal kernels on the large scenes (same box, interleaved)
for r in 1 2; do for b in 64 128 256; do echo "block $b"; DTOF_TRACE_BLOCK=$b python tools/time_c2.py domino.xml 2>&1 | tail -1; DTOF_TRACE_BLOCK=$b python tools/time_mesh.py 2>&1 | tail -1 | cut -c90-; done; done
