#!/bin/bash
# Upper bound of what staging env-map texels could save: the product build against a build whose texel reads touch no memory
# (tools/abtest/ab_no_texel_fetch.so, built with -DRMDF_AB_NO_TEXEL_FETCH).  Kernel time (bench) and FETCH_SIZE (PMC).
export TMPDIR=/tmp
for lib in tools/abtest/ab_product.so tools/abtest/ab_no_texel_fetch.so; do
  export RMDF_LIB=$PWD/$lib
  for s in 1 2; do
    python bench.py --no-cpu-baseline --no-secondary --streams $s --steps 100 2>/dev/null > /tmp/ab.json
    python tools/show_bench.py /tmp/ab.json | sed "s|/tmp/ab.json|$lib S=$s|"
  done
  d=gpurun_out/texel_ab/$(basename $lib .so); rm -rf $d; mkdir -p $d
  RMDF_BENCH_MIN_WARM=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d -- python3 bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary --streams 1 > /dev/null 2> $d/log.txt
  python3 - $d <<'PY'
import csv, glob, statistics, sys
v = {}
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render<2" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            v[r["Dispatch_Id"]] = v.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
print("   FETCH_SIZE per k_render launch (median, KiB):", statistics.median(v.values()) if v else None)
PY
done
