#!/bin/bash
# A/B of alternative builds of librmdf (tools/abtest/*.so) with RMDF_FLAGS=$1 (default 64 = split launches), 3 frames in flight
shopt -s nullglob
export RMDF_FLAGS=${1:-64}
for lib in tools/abtest/*.so; do
  export RMDF_LIB=$PWD/$lib
  python bench.py --no-cpu-baseline --no-secondary --steps 100 2>/dev/null > /tmp/ab.json
  python tools/show_bench.py /tmp/ab.json | sed "s|/tmp/ab.json|$lib flags $RMDF_FLAGS|"
done
