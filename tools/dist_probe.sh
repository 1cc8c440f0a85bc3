#!/bin/bash
# What a 1-GPU box can say about the N > 1 path of bench.py: (1) the whole sharded path over the library's RCCL exchange with ONE rank
# (RMDF_BENCH_FORCE_DIST=1): frames verified against the committed digest, deal verified by the library, host enqueue time per step;
# (2) four real processes sharing the GPU (RMDF_BENCH_SHARE_GPU=1, gloo through host staging), started by bench.py's own launcher.
# The rates it prints mean nothing for scaling (one GPU does all the work); the host enqueue time and the verdicts do.
RMDF_BENCH_FORCE_DIST=1 python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('1 rank rccl:', d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step; host enqueue', d['host_enqueue_ms_per_step_rank0'], 'ms; exchange', d['config']['exchange_ms'], 'shard render', d['config']['shard_render_ms'], '|', d['config']['exchanged_frames_verified'], '|', d['config']['tile_deal'][-90:])"
RMDF_BENCH_SHARE_GPU=1 python bench.py --gpus 4 --steps 50 --no-cpu-baseline 2>/tmp/e4.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('4 ranks sharing the GPU:', d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step; host enqueue', d['host_enqueue_ms_per_step_rank0'], 'ms |', d['config']['exchanged_frames_verified'], '| ranks', d['n_gpus'])"
tail -2 /tmp/e4.txt
