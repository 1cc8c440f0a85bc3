#!/bin/bash
# pending_gpu.sh -- everything round 5 wrote after GPU access was closed from outside the build, in the order it should run on the first
# GPU call that is allowed again.  From the repo root: gpurun --timeout 3000 -- tools/pending_gpu.sh [quick]
# Output: gpurun_out/r05p/ (copy what is judged into profiles/).  Each step has its own timeout; `quick` stops after step 5.
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
o=gpurun_out/r05p; mkdir -p $o
step() { echo "==== $(date +%T) $*"; }
step "1. the tier as the driver runs it"
timeout 900 python -m pytest tests -m gpu -x -q > $o/tier.txt 2>&1; tail -3 $o/tier.txt
step "2. bench.py, default"
timeout 600 python bench.py > $o/bench_default.json 2> $o/bench_default.err; python tools/show_bench.py $o/bench_default.json 2>/dev/null | head -30
step "3. scene kernels: default / without the eight-lane Cornell tail / four lanes per ray / round-4 library"
for l in default tools/abtest/noxl.so tools/abtest/xl4.so tools/abtest/sharedb.so tools/abtest/r04.so; do if [ $l = default ]; then unset RMDF_LIB; else export RMDF_LIB=$PWD/$l; fi; timeout 200 python tools/scene_times.py 60; done > $o/scene_times.txt 2>&1; unset RMDF_LIB; cat $o/scene_times.txt
step "4. whole-frame host call: every band count and hand-over mode; the 16-byte mirror stores"
timeout 500 python tools/whole_frame_sweep.py 40 > $o/wf_sweep.txt 2>&1; cat $o/wf_sweep.txt
RMDF_LIB=$PWD/tools/abtest/mirror16.so timeout 500 python tools/whole_frame_sweep.py 40 > $o/wf_sweep_mirror16.txt 2>&1; cat $o/wf_sweep_mirror16.txt
RMDF_COPY_NT=1 timeout 500 python tools/whole_frame_sweep.py 40 > $o/wf_sweep_copy_nt.txt 2>&1; cat $o/wf_sweep_copy_nt.txt
step "5. the tests that have never run (electric fence incl. ONE deliberate fault, 8-rank bench against the RCCL double, one-launch bands)"
RMDF_TEST_UNVERIFIED=1 timeout 1200 python -m pytest tests -m gpu -q -k "guard or eight_ranks or whole_frame_host_call or ring_form" > $o/unverified.txt 2>&1; tail -15 $o/unverified.txt
[ "${1:-}" = quick ] && exit 0
step "6. the stand-alone reproducer of the fault (stops at its first fault)"
(cd tools/repro && timeout 120 ./stale_register) > $o/repro_stale_register_rocm72.txt 2>&1; cat $o/repro_stale_register_rocm72.txt
(cd tools/repro && LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib timeout 120 ./stale_register) > $o/repro_stale_register_torch.txt 2>&1; cat $o/repro_stale_register_torch.txt
step "7. prefilter: which unit; the barrier-free ring form against the default"
timeout 600 tools/pmc_prefilter.sh > $o/prefilter_pmc.txt 2>&1; tail -12 $o/prefilter_pmc.txt
timeout 120 python tools/prefilter_time.py > $o/prefilter_time_default.txt 2>&1; RMDF_PREFILTER_RING=1 timeout 120 python tools/prefilter_time.py > $o/prefilter_time_ring.txt 2>&1; tail -6 $o/prefilter_time_default.txt $o/prefilter_time_ring.txt
step "8. the round's rocprofv3 summaries"
timeout 1500 tools/profile.sh r05 > $o/profile.log 2>&1; tail -5 $o/profile.log
timeout 300 tools/prof_scene.sh 0 > $o/prof_scene0.txt 2>&1; timeout 300 tools/prof_scene.sh 1 > $o/prof_scene1.txt 2>&1; timeout 300 tools/prof_scene.sh 3 > $o/prof_scene3.txt 2>&1
step "9. the fixed split build: the remaining tier runs to 150 (8 side by side x 8)"
tools/faultlog/loop.sh fixed2 build_variants/split_fixed 8 8 0
