#!/bin/bash
# PMC counters of k_prefilter (256x128, the four reference powers through rmdf_prefilter_env_powers = four kernels side by side).
# One rocprofv3 --pmc pass per group.   usage: tools/pmc_prefilter.sh
export TMPDIR=/tmp
out=gpurun_out/pmc_prefilter; rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 tools/prefilter_trace.py > /dev/null 2> $out/p$i.log || tail -3 $out/p$i.log
done
python3 - $out <<'PY'
import csv, glob, re, statistics, sys
c = {}
for f in glob.glob(sys.argv[1] + "/p*/**/*_counter_collection.csv", recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_prefilter" in r["Kernel_Name"]:
            m = re.search(r"k_prefilter(_chan)?<(-?\d+)", r["Kernel_Name"])
            key = "fused 1,8,64,512" if "k_prefilter_fused4" in r["Kernel_Name"] else ("chan" if m.group(1) else "one-wave") + " p=2^" + m.group(2)
            per.setdefault((key, r["Counter_Name"]), {}).setdefault(r["Dispatch_Id"], 0.0)
            per[(key, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in per.items():
        c[k] = statistics.median(v.values())
for k in sorted(c):
    print("%-18s %-26s %16.0f" % (k[0], k[1], c[k]))
# Which roof?  Per kernel: the share of each unit's cycles that the launch used (256 CUs, 1024 SIMDs, 8 XCDs; GRBM_GUI_ACTIVE is summed over
# the XCDs).  LDS array: SQ_LDS_IDX_ACTIVE = LDS-array cycles over all CUs (MI355X_MICROARCH.md "LDS": 256 B/clk/CU when every cycle is
# used); vector issue: one wave64 instruction per SIMD every 2 cycles; scalar issue: one instruction per CU and cycle.
print()
print("%-18s %9s %9s %11s %11s %11s %11s" % ("kernel", "ms@2.4GHz", "LDS array", "VALU issue", "SALU issue", "waves wait", "LDS instr"))
for name in sorted(set(k[0] for k in c)):
    g = lambda n: c.get((name, n), float("nan"))
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    if not cyc == cyc or cyc <= 0:
        continue
    print("%-18s %9.4f %9.3f %11.3f %11.3f %11.3f %11.0f" % (name, cyc / 2.4e6, g("SQ_LDS_IDX_ACTIVE") / (256 * cyc), g("SQ_INSTS_VALU") * 2 / (1024 * cyc),
          g("SQ_INSTS_SALU") / (256 * cyc), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else float("nan"), g("SQ_INSTS_LDS")))
PY
