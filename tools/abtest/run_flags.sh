#!/bin/bash
# A/B of alternative builds of librmdf (tools/abtest/*.so) with RMDF_FLAGS=$1 (default 0; the product knows 4 / 16 / 32, a build
# patched with tools/abtest/split_generations.patch also 64), the bench's default frames in flight
shopt -s nullglob
export RMDF_FLAGS=${1:-0}
for lib in tools/abtest/*.so; do
  export RMDF_LIB=$PWD/$lib
  if python bench.py --no-cpu-baseline --no-secondary --steps 100 2>/tmp/ab.err > /tmp/ab.json; then
    python tools/show_bench.py /tmp/ab.json | sed "s|/tmp/ab.json|$lib flags $RMDF_FLAGS|"
  else
    echo "$lib flags $RMDF_FLAGS: bench.py failed:"; tail -3 /tmp/ab.err
  fi
done
