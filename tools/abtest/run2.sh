#!/bin/bash
# A/B of alternative builds of librmdf (tools/abtest/*.so): headline frame, one and two frames in flight, three repeats each
shopt -s nullglob
for lib in tools/abtest/*.so; do
  export RMDF_LIB=$PWD/$lib
  for s in 1 2; do
    python bench.py --no-cpu-baseline --no-secondary --streams $s --steps 100 2>/dev/null > /tmp/ab.json
    python tools/show_bench.py /tmp/ab.json | sed "s|/tmp/ab.json|$lib S=$s|"
  done
done
