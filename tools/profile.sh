#!/bin/bash
# Collect the judged profiles on the GPU box (run from the repo root through gpurun):
#   tools/profile.sh <tag>        e.g. tools/profile.sh r02
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (headline, secondary workloads included: Cornell config 2 and the
#    lobe prefilter of config 5) with ONE frame in flight (per-kernel durations comparable with bench.py's HIP-event figure)
#    and with the default frames in flight;
# 2. PMC passes of the headline kernel, one counter group per run (never combined with tracing): FETCH_SIZE, WRITE_SIZE, SQ.
# Raw output lands in gpurun_out/prof_<tag>/; tools/pmc_summary.py condenses it into gpurun_out/prof_<tag>/summary/
# (copy to profiles/ + commit by hand; pmc_traffic.json records the sha256 of the librmdf.so it was measured with).
set -u
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
args="--steps 100 --warmup 10 --no-cpu-baseline --pmc off"
export RMDF_BENCH_MARK=1     # marker dispatches around the timed blocks (tools/pmc_summary.py cuts the trace there)
# (--no-secondary --no-animated: k_render<2, true, 0> is then launched for the headline frame only, so its row of the stats file IS the
#  headline kernel's average; Cornell, the prefilter and the animated block are in the default run's files)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_s1" -- python3 bench.py $args --streams 1 --no-secondary --no-animated > "$out/bench_s1.json" 2> "$out/trace_s1.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_default" -- python3 bench.py $args > "$out/bench_default.json" 2> "$out/trace_default.log"
export RMDF_BENCH_MIN_WARM=0
unset RMDF_BENCH_MARK
pmc_args="--steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary --streams 1 --pmc off"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py $pmc_args > /dev/null 2> "$out/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py $pmc_args > /dev/null 2> "$out/pmc_write.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_sq" -- python3 bench.py $pmc_args > /dev/null 2> "$out/pmc_sq.log"
# Cornell (config 2): SQ counters of its kernel
c2="--scene 0 --width 1280 --height 720 --max-steps 128 $pmc_args"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_SALU --output-format csv -d "$out/pmc_sq_cornell" -- python3 bench.py $c2 > /dev/null 2> "$out/pmc_sq_cornell.log"
python3 tools/pmc_summary.py "$out" "$tag"
