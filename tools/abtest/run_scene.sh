#!/bin/bash
# A/B of alternative builds of librmdf (tools/abtest/*.so) on another scene: run_scene.sh <scene> [w h steps]
shopt -s nullglob
sc=${1:-3}; w=${2:-1280}; h=${3:-720}; ms=${4:-128}
for lib in tools/abtest/*.so; do
  export RMDF_LIB=$PWD/$lib
  python bench.py --scene $sc --width $w --height $h --max-steps $ms --no-cpu-baseline --no-secondary --steps 50 --streams 1 2>/dev/null > /tmp/ab1.json
  python - "$lib" <<'PY'
import json, sys
a = json.load(open('/tmp/ab1.json'))
print("%s: %.4f ms one frame at a time (kernel %.4f)" % (sys.argv[1], a["ms_per_step"], a["roofline"]["kernel_ms_avg"]))
PY
done
