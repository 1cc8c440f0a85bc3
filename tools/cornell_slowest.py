#!/usr/bin/env python3
"""Cornell box (config 2), raster dispatch order (so that a wave's index says where it is): the waves with the longest march, with
their packet coordinates.  Cross-check build.  Measurement aid.   usage: cornell_slowest.py [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmdf_amd
W, H, MS = 1280, 720, 128
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sr = rmdf_amd.ShaderRenderer(0, flags=4, xcheck=True)          # 4 = RMDF_FLAG_RASTER_ORDER
sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
fb = np.empty(W * H, np.uint32)
for _ in range(2): sr.draw_shader_tile(0, None, W, H, 0.0, fb, max_steps=MS)
sr.debug_march_stats(True)
sr.draw_shader_tile(0, None, W, H, 0.0, fb, max_steps=MS)
st = sr.debug_march_stats(True, 32768).astype(np.float64).reshape(-1, 8)
march = st[:, 2] / 100.0
order = np.argsort(march)[::-1][:n]
gx = W // 32
for wid in order:
    strip, wave = wid // 4, wid % 4
    bx, by = strip % gx, strip // gx
    print("wave %5d: packet x %4d..%4d y %3d..%3d  lane-0 steps %3d  march %.1f us = %.2f us per lane-0 step, total %.1f us" %
          (wid, bx * 32 + wave * 8, bx * 32 + wave * 8 + 7, by * 8, by * 8 + 7, st[wid, 0], march[wid], march[wid] / max(1.0, st[wid, 0]), (st[wid, 7] - st[wid, 6]) / 100.0))
