#!/bin/bash
# A/B of tools/abtest/*.so: headline frame static and animated (1/60 s and 0.1 s per frame), one frame at a time and three in flight
shopt -s nullglob
for lib in tools/abtest/*.so; do
  export RMDF_LIB=$PWD/$lib
  for s in 1 3; do
    for an in 0 0.0167 0.1; do
      python bench.py --no-cpu-baseline --no-secondary --no-animated --streams $s --steps 100 --animate $an 2>/dev/null > /tmp/ab.json
      python tools/show_bench.py /tmp/ab.json | sed "s|/tmp/ab.json|$lib S=$s animate=$an|"
    done
  done
done
