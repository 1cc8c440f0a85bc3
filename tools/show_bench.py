#!/usr/bin/env python3
"""Print the headline figures of bench.py JSON lines: show_bench.py file.json ..."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:                                  # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    r = d.get("roofline", {})
    print("%s: %.1f Mpix/s, %.4f ms/step %s, kernel avg %s ms (min %s), valu frac %s" % (
        f, d["value"], d["ms_per_step"], d.get("ms_per_step_blocks"), r.get("kernel_ms_avg"), r.get("kernel_ms_min"), r.get("frac")))
