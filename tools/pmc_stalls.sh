#!/bin/bash
# Stall / instruction-mix counters of the headline k_render (two PMC passes, one frame in flight).
export TMPDIR=/tmp RMDF_BENCH_MIN_WARM=0
out=gpurun_out/pmc_stalls; rm -rf $out; mkdir -p $out
A="--steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary --streams 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d $out/p1 -- python3 bench.py $A > /dev/null 2> $out/p1.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH --output-format csv -d $out/p2 -- python3 bench.py $A > /dev/null 2> $out/p2.log
python3 - <<'PY'
import csv, glob, statistics
c = {}
for d in ("p1", "p2"):
    for f in glob.glob("gpurun_out/pmc_stalls/%s/**/*_counter_collection.csv" % d, recursive=True):
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
