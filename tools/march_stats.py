#!/usr/bin/env python3
"""Print per-wave scheduling statistics of the Mandelbulb march kernel (measurement aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmdf_amd

w, h, ms = 1920, 1080, 256
sr = rmdf_amd.ShaderRenderer(0, flags=rmdf_amd.FLAG_FLAT_MARCH)
sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
fb = np.empty(w * h, np.uint32)
sr.draw_shader_tile(2, None, w, h, 0.0, fb, max_steps=ms)
sr.debug_march_stats(True)
sr.draw_shader_tile(2, None, w, h, 0.0, fb, max_steps=ms)
nw = 256 * 4 * 2
st = sr.debug_march_stats(True, nw).astype(np.float64)
st = st[st[:, 7] > 0]
t0 = st[:, 6].min()
dur = (st[:, 7] - st[:, 6]) / 100.0            # us
endt = (st[:, 7] - t0) / 100.0
print("waves", len(st), "kernel span us", endt.max())
for name, col in (("iter passes", 0), ("march tails", 1), ("shade tails", 2), ("refill rounds", 3)):
    c = st[:, col]
    print("%-14s mean %8.1f min %8.0f max %8.0f sum %.3e" % (name, c.mean(), c.min(), c.max(), c.sum()))
print("iter lane utilisation %.3f" % (st[:, 4].sum() / (64 * st[:, 0].sum())))
print("march tail utilisation %.3f" % (st[:, 5].sum() / (64 * st[:, 1].sum())))
print("wave duration us: mean %.1f min %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), dur.min(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print("wave end time us: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (*np.percentile(endt, [10, 50, 90, 99]), endt.max()))
passes = st[:, 0] + st[:, 1] + st[:, 2]
print("us per pass (all kinds): mean %.3f" % (dur.sum() / passes.sum()))
tot = dur.sum()
for name, col, cnt in (("refill", 9, 0), ("iterate", 10, 0), ("push", 11, 0), ("march tail", 12, 1), ("shade tail", 13, 2), ("init", 14, 3)):
    print("%-10s: %7.1f ns per execution, share of wave time %.3f" % (name, st[:, col].sum() * 10.0 / max(1, st[:, cnt].sum()), st[:, col].sum() / 100.0 / tot))
late = np.argsort(endt)[-5:]
for i in late:
    print("late wave: iter %d mtail %d stail %d refill %d dur %.1f end %.1f" % (st[i, 0], st[i, 1], st[i, 2], st[i, 3], dur[i], endt[i]))
sr.close()
