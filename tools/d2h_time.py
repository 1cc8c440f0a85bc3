#!/usr/bin/env python3
"""Boundary path (rmdf_render_tile into an unregistered host buffer): ms per whole-frame call, headline frame and Cornell box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rmdf_amd
sr = rmdf_amd.ShaderRenderer(0); sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
w, h = 1920, 1080
host = np.empty(w * h, np.uint32)
ref = sr.render(2, w, h, 0.0, max_steps=256)["rgba8"]
for scene, ww, hh, ms in ((2, 1920, 1080, 256), (0, 1280, 720, 128)):
    hb = np.empty(ww * hh, np.uint32)
    for _ in range(5): sr.draw_shader_tile(scene, None, ww, hh, 0.0, hb, max_steps=ms)
    t0 = time.perf_counter()
    for _ in range(20): sr.draw_shader_tile(scene, None, ww, hh, 0.0, hb, max_steps=ms)
    dt = (time.perf_counter() - t0) / 20
    print("scene %d %dx%d: %.4f ms per call, %.1f Mpixels/s" % (scene, ww, hh, dt * 1e3, ww * hh / 1e6 / dt))
sr.draw_shader_tile(2, None, w, h, 0.0, host, max_steps=256)
print("equal:", np.array_equal(host.reshape(h, w), ref))
