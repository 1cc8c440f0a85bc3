#!/usr/bin/env python3
"""Wave timeline of the nested-loop kernel (all waves of the headline frame): measurement aid.
usage: nested_timeline.py [flags [scene w h max_steps]]   (flags = rmdf_config.reserved[0], e.g. 4 = raster order, 16 = no pooling)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmdf_amd
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
scene, w, h, ms = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (2, 1920, 1080, 256)
sr = rmdf_amd.ShaderRenderer(0, flags=flags, xcheck=True)
sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
fb = np.empty(w * h, np.uint32)
sr.draw_shader_tile(scene, None, w, h, 0.0, fb, max_steps=ms)
sr.draw_shader_tile(scene, None, w, h, 0.0, fb, max_steps=ms)
sr.debug_march_stats(True)
sr.draw_shader_tile(scene, None, w, h, 0.0, fb, max_steps=ms)
st = sr.debug_march_stats(True, 32768).astype(np.float64).reshape(-1, 8)
st = st[st[:, 7] > 0]
t0 = st[:, 6].min()
b, e = (st[:, 6] - t0) / 100.0, (st[:, 7] - t0) / 100.0
d = e - b
print("flags %d: waves recorded %d, span %.1f us" % (flags, len(st), e.max()))
print("wave duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f ; sum %.3e us (= %.1f us x 8192 wave slots)" %
      (d.mean(), *np.percentile(d, [50, 90, 99]), d.max(), d.sum(), d.sum() / 8192))
long_w = d > 20.0
if long_w.any():
    # shader clock while the kernel runs: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz), waves that lived > 20 us
    print("shader clock during the launch: %.0f MHz (s_memtime / s_memrealtime over %d waves)" %
          ((st[long_w, 5] - st[long_w, 4]).sum() / (st[long_w, 7] - st[long_w, 6]).sum() * 100.0, long_w.sum()))
print("start time us: p50 %.1f p90 %.1f max %.1f" % (*np.percentile(b, [50, 90]), b.max()))
for q in range(25, int(e.max()) + 25, 25):
    print("  resident at t=%d us: %d waves" % (q, ((b <= q) & (e > q)).sum()))
idx = np.argsort(d)[-5:]
for i in idx:
    print("  long wave: start %.1f dur %.1f lane0 steps %d; march ended after %.1f us" % (b[i], d[i], st[i, 0], st[i, 2] / 100.0))
if scene == 0:
    tm = st[:, 2] / 100.0
    print("Cornell: march phase per wave: mean %.1f us p99 %.1f max %.1f; after-march (normal, AO, shade) mean %.1f us p99 %.1f max %.1f" %
          (tm.mean(), np.percentile(tm, 99), tm.max(), (d - tm).mean(), np.percentile(d - tm, 99), (d - tm).max()))
