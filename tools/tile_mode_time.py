#!/usr/bin/env python3
"""Tile mode through the boundary: 64 rmdf_render_tile calls per 1920x1080 frame, each handing back the whole frame (wall clock)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rmdf_amd
sr = rmdf_amd.ShaderRenderer(0, copy_threads=int(sys.argv[1]) if len(sys.argv) > 1 else 0); sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
host = np.empty(1920*1080, np.uint32)
full = np.empty(1920*1080, np.uint32)
sr.draw_shader_tile(2, None, 1920, 1080, 0.0, full, max_steps=256)
for t in range(64): sr.draw_shader_tile(2, t, 1920, 1080, 0.0, host, max_steps=256)
print("tiled == full:", np.array_equal(host, full))
for rep in range(3):
    t0 = time.perf_counter()
    for f in range(2):
        for t in range(64): sr.draw_shader_tile(2, 64*(f+1)+t, 1920, 1080, 0.0, host, max_steps=256)
    dt = (time.perf_counter()-t0)/128*1e3
    print("ms per tile call %.4f  per frame %.2f" % (dt, dt*64))
print("tiled == full:", np.array_equal(host, full))
