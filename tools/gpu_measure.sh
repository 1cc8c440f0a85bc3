#!/bin/bash
# gpu_measure.sh -- what a GPU call does, MEASUREMENTS FIRST (VERDICT r05 "Next round" 3; replaces round 5's tools/pending_gpu.sh):
#   1 tier    the -m gpu tier exactly as the driver runs it (one worker, -x)
#   2 bench   bench.py with no flags (what the driver records)
#   3 prof    rocprofv3 --kernel-trace --stats of the bench + the PMC passes (tools/profile.sh, tools/prof_scene.sh 0/1/3): the files that
#             name the SHIPPED library's sha256 (profiles/pmc_traffic.json, scene_pmc.json, r06_kernel_stats_*.csv)
#   4 scenes  every FragmentShader value one frame at a time: shipped library, then the A/B builds (tools/abtest/rebuild_all.sh) and r04.so
#   5 sweep   the viewer's whole-frame host call by band count / hand-over mode (modes 2, 3: librmdf_xcheck.so) ; mirror16 ; copynt
#   6 prefilter  which unit bounds the one-power lobe prefilter (tools/pmc_prefilter.sh) ; the ring form (librmdf_xcheck.so) against the default
#   7 unverified  the tests of code no GPU has run yet (RMDF_TEST_UNVERIFIED=1) EXCEPT the deliberate overrun
# Nothing above faults a GPU on purpose, runs a tier several workers wide, or re-executes a process that has touched the GPU (every
# measurement is a fresh child of this shell).  What does -- the deliberate overrun of the electric-fence allocator, the stand-alone
# reproducer of round 5's memory fault, the tier in a loop (at most 4 wide) -- runs ONLY when `--faults` is given, and last.  The driver
# never runs this script.
#   gpurun --timeout 2700 -- tools/gpu_measure.sh <tag> [step ...] [--faults]      (no step list = 1 .. 7 in that order)
# Output: gpurun_out/<tag>/ and gpurun_out/prof_<tag>/ (copy what is judged into profiles/ and commit).
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=${1:-r06}; shift || true
faults=0; steps=""
for a in "$@"; do if [ "$a" = --faults ]; then faults=1; else steps="$steps $a"; fi; done
[ -z "$steps" ] && [ $faults = 0 ] && steps="tier bench prof scenes sweep prefilter unverified"
o=gpurun_out/$tag; mkdir -p $o
XLIB=$PWD/ray-marching-distance-fields_amd/librmdf_xcheck.so
sha256sum ray-marching-distance-fields_amd/librmdf.so $XLIB > $o/lib_sha256.txt; cat $o/lib_sha256.txt
step() { echo "==== $(date +%T) $*"; }
for s in $steps; do case $s in
tier)  step "1 the tier as the driver runs it (one worker, -x)"
       timeout 1500 python -m pytest tests -m gpu -x -q > $o/tier.txt 2>&1; echo "rc=$?" >> $o/tier.txt; tail -4 $o/tier.txt ;;
bench) step "2 bench.py, default (what the driver records)"
       timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "rc=$?"; python tools/show_bench.py $o/bench_default.json 2>/dev/null | head -40 ;;
prof)  step "3 rocprofv3 kernel trace + PMC passes of the shipped library"
       timeout 1500 tools/profile.sh $tag > $o/profile.log 2>&1; tail -8 $o/profile.log
       for sc in 0 1 3; do timeout 300 tools/prof_scene.sh $sc > $o/prof_scene$sc.txt 2>&1; tail -1 $o/prof_scene$sc.txt; done ;;
scenes) step "4 scene kernels one frame at a time: shipped library, then the A/B builds"
       [ -f tools/abtest/mt56.so ] || tools/abtest/rebuild_all.sh > $o/abtest_build.log 2>&1      # (git-ignored binaries: a fresh checkout builds them here, ~1 min)
       for l in default tools/abtest/mt48.so tools/abtest/mt56.so tools/abtest/noxl.so tools/abtest/xl4.so tools/abtest/sharedb.so tools/abtest/r04.so; do
         [ $l = default ] || [ -f $l ] || continue
         if [ $l = default ]; then unset RMDF_LIB; else export RMDF_LIB=$PWD/$l; fi; timeout 200 python tools/scene_times.py 60; done > $o/scene_times.txt 2>&1; unset RMDF_LIB; cat $o/scene_times.txt ;;
sweep) step "5 whole-frame host call: every band count / hand-over mode"
       timeout 600 python tools/whole_frame_sweep.py 40 > $o/wf_sweep.txt 2>&1; cat $o/wf_sweep.txt ;;
mirror16) step "5b the same with 16-byte mirror stores (A/B build)"
       RMDF_LIB=$PWD/tools/abtest/mirror16.so timeout 600 python tools/whole_frame_sweep.py 40 product > $o/wf_sweep_mirror16.txt 2>&1; cat $o/wf_sweep_mirror16.txt ;;
copynt) step "5c the same with streaming-store host copies (librmdf_xcheck.so, RMDF_COPY_NT=1)"
       RMDF_COPY_NT=1 RMDF_LIB=$XLIB timeout 600 python tools/whole_frame_sweep.py 40 product > $o/wf_sweep_copy_nt.txt 2>&1; cat $o/wf_sweep_copy_nt.txt ;;
prefilter) step "6 prefilter: which unit; the barrier-free ring form against the default"
       timeout 600 tools/pmc_prefilter.sh > $o/prefilter_pmc.txt 2>&1; tail -12 $o/prefilter_pmc.txt
       timeout 120 python tools/prefilter_time.py > $o/prefilter_time_default.txt 2>&1
       RMDF_LIB=$XLIB timeout 120 python tools/prefilter_time.py > $o/prefilter_time_xcheck_default.txt 2>&1
       RMDF_LIB=$XLIB RMDF_PREFILTER_RING=1 timeout 120 python tools/prefilter_time.py > $o/prefilter_time_ring.txt 2>&1
       tail -6 $o/prefilter_time_default.txt $o/prefilter_time_xcheck_default.txt $o/prefilter_time_ring.txt ;;
unverified) step "7 tests of code no GPU has run yet (no deliberate fault among them)"
       RMDF_TEST_UNVERIFIED=1 timeout 1500 python -m pytest tests -m gpu -q -k "(guard and not overrun) or no_kernel_touches or eight_ranks or whole_frame_host_call or ring_form or eight_lane_tail" > $o/unverified.txt 2>&1; tail -15 $o/unverified.txt ;;
*) echo "unknown step $s" ;;
esac; done
if [ $faults = 1 ]; then
  step "F1 the electric fence catches a deliberate overrun (ONE GPU memory fault, in a child process)"
  RMDF_TEST_UNVERIFIED=1 timeout 600 python -m pytest tests -m gpu -q -k "fence_does_catch_an_overrun" > $o/faults_overrun.txt 2>&1; tail -5 $o/faults_overrun.txt
  step "F2 the stand-alone reproducer of round 5's fault (stops at its first fault)"
  (cd tools/repro && timeout 120 ./stale_register) > $o/repro_stale_register_rocm72.txt 2>&1; tail -5 $o/repro_stale_register_rocm72.txt
  step "F3 the tier in a loop, 4 wide x 4 (fresh processes, own copies of the tree)"
  tools/faultlog/loop.sh r06 - 4 4 0
fi
step done
