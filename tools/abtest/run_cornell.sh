#!/bin/bash
# A/B of alternative builds of librmdf (tools/abtest/*.so) on the Cornell box (config 2): one frame at a time and two in flight
shopt -s nullglob
for lib in tools/abtest/*.so; do
  export RMDF_LIB=$PWD/$lib
  python bench.py --scene 0 --width 1280 --height 720 --max-steps 128 --no-cpu-baseline --no-secondary --steps 100 --streams 1 2>/dev/null > /tmp/ab1.json
  python bench.py --scene 0 --width 1280 --height 720 --max-steps 128 --no-cpu-baseline --no-secondary --steps 100 --streams 2 2>/dev/null > /tmp/ab2.json
  python - "$lib" <<'PY'
import json, sys
a, b = json.load(open('/tmp/ab1.json')), json.load(open('/tmp/ab2.json'))
print("%s: cornell %.4f ms one frame at a time (kernel %.4f), %.4f ms per frame with two in flight" % (sys.argv[1], a["ms_per_step"], a["roofline"]["kernel_ms_avg"], b["ms_per_step"]))
PY
done
