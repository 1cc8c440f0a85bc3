#!/bin/bash
# A/B of alternative builds of librmdf (tools/abtest/*.so): headline frame, two frames in flight (the bench default is three: run4.sh / run5.sh), then one
shopt -s nullglob
for s in 2 1; do
  for lib in tools/abtest/*.so; do
    export RMDF_LIB=$PWD/$lib
    python bench.py --no-cpu-baseline --no-secondary --streams $s --steps 100 2>/dev/null > /tmp/ab.json
    python tools/show_bench.py /tmp/ab.json | sed "s|/tmp/ab.json|$lib S=$s|"
  done
done
