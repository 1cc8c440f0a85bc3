#!/usr/bin/env python3
"""Tile mode through the boundary: 64 rmdf_render_tile calls per frame, each handing back the whole frame (wall clock).
usage: tile_mode_time.py [copy_threads=0 [w=1920 h=1080]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rmdf_amd
sr = rmdf_amd.ShaderRenderer(0, copy_threads=int(sys.argv[1]) if len(sys.argv) > 1 else 0); sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
host = np.empty(W*H, np.uint32)
full = np.empty(W*H, np.uint32)
sr.draw_shader_tile(2, None, W, H, 0.0, full, max_steps=256)
for t in range(64): sr.draw_shader_tile(2, t, W, H, 0.0, host, max_steps=256)
print("tiled == full:", np.array_equal(host, full))
for rep in range(3):
    t0 = time.perf_counter()
    for f in range(2):
        for t in range(64): sr.draw_shader_tile(2, 64*(f+1)+t, W, H, 0.0, host, max_steps=256)
    dt = (time.perf_counter()-t0)/128*1e3
    print("ms per tile call %.4f  per frame %.2f" % (dt, dt*64))
print("tiled == full:", np.array_equal(host, full))
