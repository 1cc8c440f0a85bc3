#!/usr/bin/env python3
"""Cornell box (config 2): microseconds per march step of the long waves, in the full frame (machine crowded) and in single tiles of the
same frame (the same packets with the machine nearly empty).  Cross-check build (wave timeline in p.dbg).  Measurement aid.
usage: cornell_step_time.py [min_steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmdf_amd
W, H, MS = 1280, 720, 128
min_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sr = rmdf_amd.ShaderRenderer(0, flags=0, xcheck=True)
sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
fb = np.empty(W * H, np.uint32)

def timeline(tile):
    for _ in range(2): sr.draw_shader_tile(0, tile, W, H, 0.0, fb, max_steps=MS)
    sr.debug_march_stats(True)
    sr.draw_shader_tile(0, tile, W, H, 0.0, fb, max_steps=MS)
    st = sr.debug_march_stats(True, 32768).astype(np.float64).reshape(-1, 8)
    st = st[st[:, 7] > 0]
    t0 = st[:, 6].min()
    return st, (st[:, 6] - t0) / 100.0, (st[:, 7] - t0) / 100.0

def report(name, st, b, e):
    steps, march = st[:, 0], st[:, 2] / 100.0
    sel = steps >= min_steps
    if not sel.any():
        print("%s: no wave with lane-0 steps >= %d (waves %d, span %.1f us)" % (name, min_steps, len(st), e.max())); return
    us = march[sel] / steps[sel]
    print("%s: waves %d, span %.1f us; %d waves with lane-0 steps >= %d: steps mean %.0f max %.0f, march %.1f us mean, %.2f us per step (p10 %.2f p90 %.2f)" %
          (name, len(st), e.max(), sel.sum(), min_steps, steps[sel].mean(), steps[sel].max(), march[sel].mean(), us.mean(), *np.percentile(us, [10, 90])))

st, b, e = timeline(None)
report("full frame", st, b, e)
for tile in range(64):
    st, b, e = timeline(tile)
    if (st[:, 0] >= min_steps).any(): report("tile %2d alone" % tile, st, b, e)
