#!/bin/bash
# PMC snapshot of k_render for one scene: VALU issue rate, lane utilisation, wait fraction.   usage: tools/prof_scene.sh <scene> [w h steps]
export TMPDIR=/tmp
sc=${1:-0}; w=${2:-1280}; h=${3:-720}; ms=${4:-128}
out=gpurun_out/prof_scene$sc
rm -rf $out; mkdir -p $out
tm=0.0; [ "$sc" = 1 ] && tm=2.5; [ "$sc" = 3 ] && tm=3.0      # the committed views bench.py's secondary workloads time (tests/golden/full_size_digests.json)
A="--scene $sc --time $tm --pmc off --width $w --height $h --max-steps $ms --no-cpu-baseline --no-secondary --repeats 1 --steps 20 --warmup 2 --streams 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $out/p1 -- python3 bench.py $A > /dev/null 2> $out/p1.log
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $out/p2 -- python3 bench.py $A > /dev/null 2> $out/p2.log
python3 - $sc <<'PY'
import csv, glob, statistics, sys
sc = sys.argv[1]
c = {}
for d in ("p1", "p2"):
    for f in glob.glob("gpurun_out/prof_scene%s/%s/**/*_counter_collection.csv" % (sc, d), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if "k_render<" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, v in per.items():
            c[k] = statistics.median(v.values())
ms = c["GRBM_GUI_ACTIVE"] / 8 / 2.4e6
import json, os
name = {"0": "config2_cornell_1280x720_m128", "1": "scene1_detest_1280x720_m128", "3": "scene3_mbgeneral_1280x720_m128"}.get(sc)
import hashlib
lib_sha = hashlib.sha256(open("ray-marching-distance-fields_amd/librmdf.so", "rb").read()).hexdigest()
if name:
    # bench.py reads the instruction count of the secondary scenes from here (profiles/scene_pmc.json after copying)
    fn = "gpurun_out/scene_pmc.json"
    d = json.load(open(fn)) if os.path.exists(fn) else {}
    d[name] = {"SQ_INSTS_VALU_per_launch": c["SQ_INSTS_VALU"], "lane_utilisation": round(c["SQ_THREAD_CYCLES_VALU"] / 64 / c["SQ_ACTIVE_INST_VALU"], 3),
               "kernel_ms_from_GRBM_GUI_ACTIVE": round(ms, 4), "lib_sha256": lib_sha, "source": "tools/prof_scene.sh %s (rocprofv3 --pmc, median per launch)" % sc}
    json.dump(d, open(fn, "w"), indent=1, sort_keys=True)
print("scene %s: kernel ~%.3f ms; VALU instr %.1f M -> %.2f G/s/SIMD (0.93 = issue peak); lane utilisation %.2f; waves waiting %.0f %% of their cycles; SALU %.1f M, SMEM %.2f M, VMEM reads %.2f M, LDS %.2f M" %
      (sc, ms, c["SQ_INSTS_VALU"] / 1e6, c["SQ_INSTS_VALU"] / 1024 / (ms * 1e-3) / 1e9, c["SQ_THREAD_CYCLES_VALU"] / 64 / c["SQ_ACTIVE_INST_VALU"],
       100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_INSTS_SALU"] / 1e6, c["SQ_INSTS_SMEM"] / 1e6, c.get("SQ_INSTS_VMEM_RD", 0) / 1e6, c.get("SQ_INSTS_LDS", 0) / 1e6))
PY
