#!/bin/bash
# Sample the GPU clocks / power with rocm-smi while the headline bench runs (is the kernel power- or clock-limited?).
( for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clock_watch.log &
W=$!
python bench.py --no-cpu-baseline --no-secondary --steps 2000 --repeats 3 2>/dev/null > /tmp/cw.json
wait $W
python tools/show_bench.py /tmp/cw.json
cat gpurun_out/clock_watch.log | head -14
