#!/bin/bash
# TCP / cache counters of a Cornell rectangle launch (rocprofv3 --pmc): usage tools/cornell_rect_tcp.sh
export TMPDIR=/tmp
out=gpurun_out/rect_tcp; rm -rf $out; mkdir -p $out
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 tools/cornell_rect_pmc.py 824 88 840 184 > /dev/null 2> $out/p$i.log
done
python3 - <<'PY'
import csv, glob, statistics
c={}
for f in glob.glob("gpurun_out/rect_tcp/p*/**/*_counter_collection.csv", recursive=True):
    per={}
    for r in csv.DictReader(open(f)):
        if "k_render<" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"],{}).setdefault(r["Dispatch_Id"],0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]]+=float(r["Counter_Value"])
    for k,v in per.items(): c[k]=(statistics.median(v.values()), list(v.values()))
for k,v in sorted(c.items()): print(k, v[0], [round(x) for x in v[1]])
PY
