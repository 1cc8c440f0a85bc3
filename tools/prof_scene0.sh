#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/prof_cornell
mkdir -p $out
A="--scene 0 --width 1280 --height 720 --max-steps 128 --no-cpu-baseline --steps 20 --warmup 2 --streams 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $out/p1 -- python3 bench.py $A > /dev/null 2> $out/p1.log
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $out/p2 -- python3 bench.py $A > /dev/null 2> $out/p2.log
python3 - <<'PY'
import csv, glob, statistics
for d in ("p1", "p2"):
    for f in glob.glob("gpurun_out/prof_cornell/%s/**/*_counter_collection.csv" % d, recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if "k_render<0" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for c, v in per.items():
            print(c, statistics.median(v.values()))
PY
tail -3 $out/p2.log
