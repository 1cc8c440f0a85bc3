#!/usr/bin/env python3
"""Timings of the secondary BASELINE.json configurations (device-resident unless noted)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rmdf_amd

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream); sp = stream.cuda_stream
sr = rmdf_amd.ShaderRenderer(0)
sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
out = {}

def time_render(scene, w, h, ms, t=0.0, n=50):
    frame = torch.empty((h, w), dtype=torch.int32, device=dev)
    for _ in range(5):
        sr.render_rect_device(scene, w, h, t, ms, (0, 0, w, h), d_rgba8=frame.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(n):
        sr.render_rect_device(scene, w, h, t, ms, (0, 0, w, h), d_rgba8=frame.data_ptr(), stream=sp)
    e1.record(stream); torch.cuda.synchronize()
    ms_ = e0.elapsed_time(e1) / n
    return ms_, w * h / 1e6 / (ms_ * 1e-3)

m, r = time_render(0, 1280, 720, 128)
out["C2 CornellBox 1280x720/128"] = {"ms": round(m, 4), "Mpixels/s": round(r, 1)}
m, r = time_render(2, 1920, 1080, 256)
out["C3 Mandelbulb 1920x1080/256"] = {"ms": round(m, 4), "Mpixels/s": round(r, 1)}
# the reference's tiled dispatch: 64 drawShaderTile calls per frame, each returning the whole accumulated frame to a host buffer
fbv = np.empty(1920 * 1080, np.uint32)
for idx in range(64):
    sr.draw_shader_tile(2, idx, 1920, 1080, 0.0, fbv, max_steps=256)
t0 = time.perf_counter()
for idx in range(64):
    sr.draw_shader_tile(2, idx, 1920, 1080, 0.0, fbv, max_steps=256)
tt = time.perf_counter() - t0
out["C3 tiled: 64 drawShaderTile calls (each copies the whole frame to the host, as fillFrameBuffer's orphaned PBO requires)"] = {
    "ms per frame of 64 tiles": round(tt * 1e3, 2), "ms per call": round(tt / 64 * 1e3, 3), "Mpixels/s": round(1920 * 1080 / 1e6 / tt, 1)}
t0 = time.perf_counter()
for _ in range(20):
    sr.draw_shader_tile(2, None, 1920, 1080, 0.0, fbv, max_steps=256)
tt = (time.perf_counter() - t0) / 20
out["C3 untiled through the host-buffer boundary (drawShaderTile Nothing)"] = {"ms": round(tt * 1e3, 3), "Mpixels/s": round(1920 * 1080 / 1e6 / tt, 1)}
m, r = time_render(2, 7680, 4320, 256, n=5)
out["C4 Mandelbulb 7680x4320 rays (4 rays/px of 3840x2160), 1 GPU, no resolve"] = {"ms": round(m, 3), "Mrays/s": round(r, 1)}
# C5: env prefilter: synthetic 2048x1024 latlong -> resize 256 -> 4 powers (host buffers in/out, PCIe included)
rng = np.random.RandomState(0)
big = np.exp(rng.uniform(-3, 3, (1024, 2048, 3))).astype(np.float32)
t0 = time.perf_counter(); small = sr.resize_latlong(big, 256); t1 = time.perf_counter()
ts = []
for p in (1.0, 8.0, 64.0, 512.0):
    ta = time.perf_counter(); sr.prefilter_env(small, p); ts.append(time.perf_counter() - ta)
ta = time.perf_counter(); sr.prefilter_env_powers(small, (1.0, 8.0, 64.0, 512.0)); t4 = time.perf_counter() - ta
out["C5 prefilter 2048x1024 -> 256x128, powers 1/8/64/512 (host in/out)"] = {
    "resize_ms": round((t1 - t0) * 1e3, 2), "per_power_ms": [round(x * 1e3, 2) for x in ts],
    "four_powers_concurrent_ms": round(t4 * 1e3, 2), "G pair-terms/s": round(4 * (256 * 128) ** 2 / t4 / 1e9, 1)}
# (the CPU baselines of these paths are bench.py's cpu_baseline / cpu_reference_paths leg: only that leg may run the oracle)
print(json.dumps(out, indent=1))
sr.close()
