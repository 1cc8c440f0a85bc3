#!/bin/bash
# PMC counters of k_render for one rectangle of the Cornell frame rendered alone (tools/cornell_rect_pmc.py): instruction counts and the
# average latency of its LDS / vector-memory instructions.   usage: tools/cornell_rect_pmc.sh x0 y0 x1 y1
export TMPDIR=/tmp
out=gpurun_out/rect_pmc; rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "VmemLatency" "LdsLatency"; do
  i=$((i + 1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 tools/cornell_rect_pmc.py $1 $2 $3 $4 > /dev/null 2> $out/p$i.log
done
python3 - <<'PY'
import csv, glob, statistics
c = {}
for f in glob.glob("gpurun_out/rect_pmc/p*/**/*_counter_collection.csv", recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_render<" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in per.items(): c[k] = statistics.median(v.values())
print({k: round(v, 1) for k, v in sorted(c.items())}, "kernel us ~ %.1f" % (c["GRBM_GUI_ACTIVE"] / 8 / 2400))
PY
