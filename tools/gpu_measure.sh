#!/bin/bash
# gpu_measure.sh -- what a GPU call does FIRST, in this order (VERDICT r05 "Next round" 3): the tier as the driver runs it, the bench as the
# driver runs it, the rocprofv3 summaries + PMC passes that name the SHIPPED library's sha, the per-scene timings.  Nothing in here
# faults a GPU on purpose, loops a tier several workers wide, or re-executes a process that has touched the GPU.
#   gpurun --timeout 2400 -- tools/gpu_measure.sh <tag> [tier|bench|prof|scenes|sweep ...]     (no step list = all, in that order)
# Output: gpurun_out/<tag>/ and gpurun_out/prof_<tag>/ (copy what is judged into profiles/).
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=${1:-r06}; shift || true
steps=${*:-tier bench prof scenes sweep}
o=gpurun_out/$tag; mkdir -p $o
sha256sum ray-marching-distance-fields_amd/librmdf.so ray-marching-distance-fields_amd/librmdf_xcheck.so > $o/lib_sha256.txt; cat $o/lib_sha256.txt
step() { echo "==== $(date +%T) $*"; }
for s in $steps; do case $s in
tier)  step "the tier as the driver runs it (one worker, -x)"
       timeout 1500 python -m pytest tests -m gpu -x -q > $o/tier.txt 2>&1; echo "rc=$?" >> $o/tier.txt; tail -4 $o/tier.txt ;;
bench) step "bench.py, default (what the driver records)"
       timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "rc=$?"; python tools/show_bench.py $o/bench_default.json 2>/dev/null | head -40 ;;
prof)  step "rocprofv3 kernel trace + PMC passes of the shipped library"
       timeout 1500 tools/profile.sh $tag > $o/profile.log 2>&1; tail -8 $o/profile.log
       for sc in 0 1 3; do timeout 300 tools/prof_scene.sh $sc > $o/prof_scene$sc.txt 2>&1; tail -1 $o/prof_scene$sc.txt; done ;;
scenes) step "scene kernels one frame at a time: shipped library, then the A/B builds"
       for l in default tools/abtest/noxl.so tools/abtest/xl4.so tools/abtest/sharedb.so tools/abtest/r04.so; do
         [ $l = default ] || [ -f $l ] || continue
         if [ $l = default ]; then unset RMDF_LIB; else export RMDF_LIB=$PWD/$l; fi; timeout 200 python tools/scene_times.py 60; done > $o/scene_times.txt 2>&1; unset RMDF_LIB; cat $o/scene_times.txt ;;
sweep) step "whole-frame host call: every band count / hand-over mode"
       timeout 500 python tools/whole_frame_sweep.py 40 > $o/wf_sweep.txt 2>&1; cat $o/wf_sweep.txt ;;
*) echo "unknown step $s" ;;
esac; done
step done
