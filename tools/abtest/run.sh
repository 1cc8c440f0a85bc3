#!/bin/bash
# A/B of alternative builds of librmdf: put them next to this script as tools/abtest/<name>.so (e.g. built with other
# compiler flags: hipcc ... -o tools/abtest/maxilp.so) and run from the repo root on the GPU box.  Headline frame and Cornell,
# one frame in flight; `default` = the in-tree librmdf.so.
shopt -s nullglob
for lib in default tools/abtest/*.so; do
  if [ "$lib" = default ]; then unset RMDF_LIB; else export RMDF_LIB=$PWD/$lib; fi
  a=$(python bench.py --no-cpu-baseline --streams 1 --steps 100 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms_avg"])')
  b=$(python bench.py --scene 0 --width 1280 --height 720 --max-steps 128 --no-cpu-baseline --steps 100 --streams 1 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
  echo "$lib: mandelbulb $a ms, cornell $b ms"
done
