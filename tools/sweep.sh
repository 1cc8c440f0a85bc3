#!/bin/bash
# usage: tools/sweep.sh "VAR=a VAR2=b" "VAR=c" ...   -> one bench line per env setting
for cfg in "$@"; do
  out=$(env $cfg python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$cfg => $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mpix/s", d["ms_per_step"], "ms; kernel", d["roofline"]["kernel_ms_avg"], d["roofline"]["kernel_ms_min"])')"
done
