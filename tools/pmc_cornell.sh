#!/bin/bash
# Where the Cornell kernel (config 2) waits: scalar-cache / instruction-cache hit rates, scalar and LDS issue counts, stall counters.
# One rocprofv3 --pmc pass per counter group (never combined with tracing); one frame in flight.   usage: tools/pmc_cornell.sh [tag]
export TMPDIR=/tmp RMDF_BENCH_MIN_WARM=0
tag=${1:-cornell}
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
A="--scene 0 --width 1280 --height 720 --max-steps 128 --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary --no-animated --streams 1"
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 bench.py $A > /dev/null 2> $out/p$i.log || tail -3 $out/p$i.log
done
python3 - $out <<'PY'
import csv, glob, statistics, sys
c = {}
for f in glob.glob(sys.argv[1] + "/p*/**/*_counter_collection.csv", recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_render<0" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in per.items():
        c[k] = statistics.median(v.values())
for k in sorted(c):
    print("%-28s %16.0f" % (k, c[k]))
PY
