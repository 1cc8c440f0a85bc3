#!/usr/bin/env python3
"""Lobe prefilter timings at 256x128 (BASELINE config 5) on the GPU: each reference power alone, the four powers on four
streams, and the host-pointer entry that runs them concurrently."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rmdf_amd
sr = rmdf_amd.ShaderRenderer(0)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 128)
src = np.random.RandomState(3).uniform(0, 4, (h, w, 3)).astype(np.float32)
d_src = torch.from_numpy(src).cuda()
outs = [torch.empty_like(d_src) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
P = (1.0, 8.0, 64.0, 512.0)
torch.cuda.synchronize()
for p, o in zip(P, outs):
    sr.prefilter_env_device(d_src.data_ptr(), w, h, p, o.data_ptr(), stream=streams[0].cuda_stream)
torch.cuda.synchronize()
for p, o in zip(P, outs):
    t0 = time.perf_counter()
    for _ in range(5):
        sr.prefilter_env_device(d_src.data_ptr(), w, h, p, o.data_ptr(), stream=streams[0].cuda_stream)
    torch.cuda.synchronize()
    print("power %5.0f alone: %.3f ms" % (p, (time.perf_counter() - t0) / 5 * 1e3))
t0 = time.perf_counter()
for _ in range(5):
    for k, (p, o) in enumerate(zip(P, outs)):
        sr.prefilter_env_device(d_src.data_ptr(), w, h, p, o.data_ptr(), stream=streams[k].cuda_stream)
torch.cuda.synchronize()
print("four powers on four streams: %.3f ms per set" % ((time.perf_counter() - t0) / 5 * 1e3))
sr.prefilter_env_powers(src, P)
t0 = time.perf_counter()
for _ in range(5):
    sr.prefilter_env_powers(src, P)
print("rmdf_prefilter_env_powers (host in/out): %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
for PP in ((1.0, 8.0), (1.0, 8.0, 64.0)):
    sr.prefilter_env_powers(src, PP)
    t0 = time.perf_counter()
    for _ in range(5):
        sr.prefilter_env_powers(src, PP)
    print("rmdf_prefilter_env_powers %r (host in/out): %.3f ms" % (PP, (time.perf_counter() - t0) / 5 * 1e3))
