#!/usr/bin/env python3
"""One GPU standing in for rank r of N (no gather): per-frame time of the shard render, one frame at a time and with S
frames in flight.  Shows what bounds the N-GPU strong-scaling run: the longest ray's serial chain (one frame at a time)
vs the shard's share of the work (frames in flight).  Measurement aid."""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rmdf_amd

dev = torch.device("cuda", 0)
sr = rmdf_amd.ShaderRenderer(0)
sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
w, h, ms = 1920, 1080, 256
out = {}
for n in (1, 2, 4, 8):
    slots = rmdf_amd.shard_slots(n)
    for S in (1, 2, 4, 8):
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        bufs = [torch.zeros((slots, h // 8, w // 8), dtype=torch.int32, device=dev) for _ in range(S)]
        worst = host = 0.0
        for r in sorted({0, n // 2, n - 1}):
            def frame(i):
                k = i % S
                sr.render_shard_device(2, w, h, 0.0, ms, r, n, bufs[k].data_ptr(), stream=streams[k].cuda_stream)
            for i in range(4 * S):
                frame(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 200
            for i in range(K):
                frame(i)
            enq = (time.perf_counter() - t0) / K * 1e3
            torch.cuda.synchronize()
            worst = max(worst, (time.perf_counter() - t0) / K * 1e3)
            host = max(host, enq)
        out["N=%d S=%d" % (n, S)] = round(worst, 4)
        print("N=%d ranks, %d frame(s) in flight: slowest rank %.4f ms/frame (host enqueue %.4f) -> %.0f Mpixels/s if the gather hides" %
              (n, S, worst, host, w * h / 1e6 / (worst * 1e-3)), flush=True)
print(json.dumps(out))
sr.close()
