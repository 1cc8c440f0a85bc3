#!/bin/bash
# Collect the judged profiles of the headline bench on the GPU box (run from the repo root through gpurun):
#   tools/profile.sh <tag>        e.g. tools/profile.sh r01c
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py` with ONE frame in flight (per-kernel durations that are
#    comparable with bench.py's HIP-event figure) and with the default frames in flight;
# 2. PMC passes, one counter group per run (never combined with tracing): FETCH_SIZE, WRITE_SIZE, SQ counters.
# Raw output lands in gpurun_out/prof_<tag>/; tools/pmc_summary.py condenses it into profiles/ (copy + commit by hand).
set -u
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
args="--steps 100 --warmup 10 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_s1" -- python3 bench.py $args --streams 1 > "$out/bench_s1.json" 2> "$out/trace_s1.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_default" -- python3 bench.py $args > "$out/bench_default.json" 2> "$out/trace_default.log"
pmc_args="--steps 20 --warmup 2 --no-cpu-baseline --streams 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py $pmc_args > /dev/null 2> "$out/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py $pmc_args > /dev/null 2> "$out/pmc_write.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_sq" -- python3 bench.py $pmc_args > /dev/null 2> "$out/pmc_sq.log"
python3 tools/pmc_summary.py "$out" "$tag"
