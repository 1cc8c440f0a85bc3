#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/prof_shard
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 tools/rank_times.py > $out/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/prof_shard/p1/**/*_counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    grid = {}
    for r in csv.DictReader(open(f)):
        if "k_render<2" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); grid[r["Dispatch_Id"]] = r["Grid_Size"]
    by_grid = collections.defaultdict(list)
    for d, c in per.items(): by_grid[grid[d]].append(c["SQ_INSTS_VALU"])
    for g, v in sorted(by_grid.items(), key=lambda x: -int(x[0])):
        v.sort(); print("grid %s: %d launches, VALU instr median %.1f M (min %.1f max %.1f)" % (g, len(v), v[len(v)//2] / 1e6, v[0] / 1e6, v[-1] / 1e6))
PY
