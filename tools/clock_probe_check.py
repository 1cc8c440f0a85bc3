#!/usr/bin/env python3
"""Shader clock (rmdf_probe_shader_clock: s_memtime against s_memrealtime on one wave of a highest-priority stream) idle and while
the headline frame renders continuously on three streams: 100 us windows at growing delays after the load starts."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, rmdf_amd
dev = torch.device("cuda", 0)
sr = rmdf_amd.ShaderRenderer(0); sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
S = 3
streams = [torch.cuda.Stream(dev) for _ in range(S)]
bufs = [torch.zeros((1080, 1920), dtype=torch.int32, device=dev) for _ in range(S)]
def frames(n):
    for i in range(n):
        sr.render_rect_device(2, 1920, 1080, 0.0, 256, (0, 0, 1920, 1080), d_rgba8=bufs[i % S].data_ptr(), stream=streams[i % S].cuda_stream)
frames(30); torch.cuda.synchronize()
print("idle:", [round(sr.probe_shader_clock(300.0)) for _ in range(3)])
for rep in range(2):
    time.sleep(0.05)
    frames(400)                                   # ~160 ms of work
    t0 = time.perf_counter()
    out = []
    for k in range(40):
        mhz = sr.probe_shader_clock(100.0)
        out.append("%.1f ms: %.0f" % ((time.perf_counter() - t0) * 1e3, mhz))
        time.sleep(0.002 if k > 8 else 0.0)
    torch.cuda.synchronize()
    print("under load (time since the probes began: MHz):", "  ".join(out), " | frames done at %.1f ms" % ((time.perf_counter() - t0) * 1e3))
