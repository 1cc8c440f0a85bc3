#!/bin/bash
# frames in flight on ONE GPU: the driver-style run (20 warm-up, 3 x K steps) for S = 1..6
for s in 1 2 3 4 6; do
  python bench.py --no-cpu-baseline --no-secondary --streams $s --steps ${1:-100} 2>/dev/null > /tmp/ab.json
  python tools/show_bench.py /tmp/ab.json | sed "s|/tmp/ab.json|S=$s|"
done
