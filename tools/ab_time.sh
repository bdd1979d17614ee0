#!/bin/bash
# A/B timing of several builds of libdtof.so on the same GPU box (tools/ab/*.so, selected through DTOF_LIB), interleaved twice.
for round in 1 2; do
  for lib in tools/ab/*.so; do
    echo "== $(basename $lib)"
    DTOF_LIB=$PWD/$lib python tools/time_c2.py cornell_wall.xml 2>/dev/null | tail -1
  done
done
