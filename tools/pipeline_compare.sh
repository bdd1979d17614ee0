for p in split fused; do for s in domino.xml; do DTOF_PIPELINE=$p python tools/time_c2.py $s 2>&1 | tail -1; done; done
for p in split fused; do DTOF_PIPELINE=$p python tools/time_mesh.py 2>&1 | tail -1; done
for p in split fused; do DTOF_PIPELINE=$p python tools/time_c2.py cornell_boxes.xml 64 2>&1 | tail -1; done
