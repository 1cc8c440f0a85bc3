#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/profile.sh into the files kept under profiles/.
usage: pmc_summary.py gpurun_out/prof_<tag> <tag>   -> gpurun_out/prof_<tag>/summary/{<tag>_*.csv, pmc_traffic.json}"""
import csv, glob, hashlib, json, os, shutil, statistics, sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)
KERNEL = "k_render<2"


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return f[0] if f else None


for name in ("trace_s1", "trace_default"):
    f = one(name + "/**/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, name[6:])))
for name in ("bench_s1.json", "bench_default.json"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        lines = [l for l in open(f) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, "%s_%s" % (tag, name)), "w").write(lines[-1])

counters = {}
# Cornell box (config 2): the SQ counters of k_render<0, ...>
f = one("pmc_sq_cornell/**/*_counter_collection.csv")
cornell = {}
if f:
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_render<0" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    cornell = {c: statistics.median(d.values()) for c, d in per.items()}

for grp in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = one(grp + "/**/*_counter_collection.csv")
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]]
    # keep the judged evidence small: the k_render rows only
    with open(os.path.join(dst, "%s_%s.csv" % (tag, grp)), "w", newline="") as o:
        wr = csv.writer(o)
        wr.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count",
                     "SGPR_Count", "Counter_Name", "Counter_Value"])
        for r in rows:
            wr.writerow([r["Dispatch_Id"], r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"],
                         r["Accum_VGPR_Count"], r["SGPR_Count"], r["Counter_Name"], r["Counter_Value"]])
    per = {}
    for r in rows:
        per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in per.items():
        v = list(d.values())
        counters[c] = {"n": len(v), "median": statistics.median(v), "min": min(v), "max": max(v)}
    if rows:
        counters.setdefault("_resources", {"VGPR_Count": rows[0]["VGPR_Count"], "Accum_VGPR_Count": rows[0]["Accum_VGPR_Count"],
                                           "SGPR_Count": rows[0]["SGPR_Count"], "LDS_Block_Size": rows[0]["LDS_Block_Size"],
                                           "Workgroup_Size": rows[0]["Workgroup_Size"], "Grid_Size": rows[0]["Grid_Size"]})

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ray-marching-distance-fields_amd", "librmdf.so")
out = {"workload": [2, 1920, 1080, 256], "kernel": "rmdf::k_render<2, true, 0> (scene 2, MERGE, OUT_RGBA8)",
       "lib_sha256": hashlib.sha256(open(LIB, "rb").read()).hexdigest(),
       "command": "tools/profile.sh %s (rocprofv3 --pmc <one group per run> -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --streams 1)" % tag,
       "counters_per_launch": counters}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    fetch = counters["FETCH_SIZE"]["median"] * 1024.0
    write = counters["WRITE_SIZE"]["median"] * 1024.0
    out.update({"fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
                "notes": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch (median over the k_render dispatches). MI355X_MICROARCH.md: on gfx950 "
                         "FETCH_SIZE under-reports wide (16 B/lane) coalesced streams by 2x; this kernel reads 8-byte texels and scalar "
                         "constants, a width the guide calls uncalibrated, so the raw value is reported uncorrected. WRITE_SIZE is exact for "
                         "full-line stores (the LDS-staged 128-byte rows). Algorithmic bytes per launch: 8.29 MB RGBA8 store + 2.15 MB "
                         "cube-map read = 10.44 MB."})
if "SQ_INSTS_VALU" in counters and "SQ_THREAD_CYCLES_VALU" in counters:
    out["valu"] = {"SQ_INSTS_VALU_per_launch": counters["SQ_INSTS_VALU"]["median"],
                   "lane_utilisation": counters["SQ_THREAD_CYCLES_VALU"]["median"] / (64.0 * counters["SQ_ACTIVE_INST_VALU"]["median"])
                   if "SQ_ACTIVE_INST_VALU" in counters else None}
if cornell:
    out["cornell_1280x720_m128"] = {"counters_per_launch_median": cornell,
                                    "lane_utilisation": cornell.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * cornell["SQ_ACTIVE_INST_VALU"]) if cornell.get("SQ_ACTIVE_INST_VALU") else None}
# VGPR_Count / SGPR_Count as rocprofv3 prints them are allocation figures of the dispatch packet, not the compiler's register counts
if "_resources" in counters:
    counters["_resources"]["note"] = ("as printed by rocprofv3 for the dispatch (allocation units of the packet); the compiler's counts are in "
                                      "profiles/*_kernel_resources.txt (tools/resources.sh: hipcc's own remarks for every k_render instantiation)")
json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)


# The schedule the bench times (frames in flight on several streams): cut the kernel trace into the timed blocks at the marker
# dispatches (bench.py, RMDF_BENCH_MARK=1: k_resolve_box2 before the opening barrier of every block and after the last one) and
# report, per block, (last k_render end - first k_render start) / number of k_render dispatches.
def schedule(name):
    f = one(name + "/**/*_kernel_trace.csv")
    if not f:
        return None
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # (the 2x2 marker launches are one workgroup; config 4's resolve in the secondary workloads is a real one)
    marks = [int(r["Start_Timestamp"]) for r in rows if "k_resolve_box2" in r["Kernel_Name"]
             and int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) <= 1024]
    if len(marks) < 2:
        return None
    ren = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if KERNEL in r["Kernel_Name"] and "true, 0>" in r["Kernel_Name"]]
    blocks = []
    for a, b in zip(marks[:-1], marks[1:]):
        d = [x for x in ren if a <= x[0] < b]
        if d:
            span = max(e for _, e in d) - min(st for st, _ in d)
            busy = sum(e - st for st, e in d)
            blocks.append({"k_render_dispatches": len(d), "span_ms": round(span / 1e6, 4),
                           "span_ms_per_frame": round(span / 1e6 / len(d), 5),
                           "sum_of_kernel_durations_ms_per_frame": round(busy / 1e6 / len(d), 5)})
    return blocks


for name in ("trace_s1", "trace_default"):
    b = schedule(name)
    if b:
        bj = os.path.join(src, "bench_%s.json" % name[6:])
        line = [l for l in open(bj) if l.startswith("{")][-1] if os.path.exists(bj) else None
        bench = json.loads(line) if line else {}
        json.dump({"command": "RMDF_BENCH_MARK=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline"
                              + (" --streams 1" if name == "trace_s1" else ""),
                   "what": "timed blocks of the run cut out of the kernel trace at the marker dispatches: first k_render<2, true, 0> start -> last "
                           "end, per frame.  With frames in flight the kernels overlap, so the per-kernel durations of the same trace "
                           "(kernel_stats) do NOT add up to the block; span / frames is the figure bench.py's ms_per_step reproduces "
                           "(plus the host's barrier + synchronize around the block). The first blocks are warm-up / animated blocks as "
                           "bench.py orders them: `repeats` timed blocks, then (animate 0) one short and one full animated block.",
                   "blocks": b,
                   "bench_line_of_the_same_run": {k: bench.get(k) for k in ("ms_per_step", "ms_per_step_blocks", "device_span_ms_per_step",
                                                                            "ms_per_step_animated", "value", "value_animated")}},
                  open(os.path.join(dst, "%s_schedule_%s.json" % (tag, name[6:])), "w"), indent=1)
print(json.dumps({k: (v["median"] if isinstance(v, dict) and "median" in v else v) for k, v in counters.items()}, indent=1))
