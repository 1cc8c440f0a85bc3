#!/bin/bash
# Instruction-cache counters of the headline k_render (one PMC pass, one frame in flight).
export TMPDIR=/tmp RMDF_BENCH_MIN_WARM=0
out=gpurun_out/pmc_icache; rm -rf $out; mkdir -p $out
A="--steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary --streams 1"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/p1 -- python3 bench.py $A > /dev/null 2> $out/p1.log
tail -3 $out/p1.log
python3 - <<'PY'
import csv, glob, statistics
c = {}
for f in glob.glob("gpurun_out/pmc_icache/p1/**/*_counter_collection.csv", recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_render<2" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in per.items():
        c[k] = statistics.median(v.values())
for k in sorted(c):
    print("%-28s %16.0f" % (k, c[k]))
PY
