#!/usr/bin/env python3
"""Divergence statistics of the nested-loop march (8x8 packets): measurement aid."""
import os, sys
import numpy as np
os.environ["RMDF_NESTED_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmdf_amd
w, h, ms = 1920, 1080, 256
sr = rmdf_amd.ShaderRenderer(0, xcheck=True)
sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
sr.debug_march_stats(True)
fb = np.empty(w * h, np.uint32)
sr.draw_shader_tile(2, None, w, h, 0.0, fb, max_steps=ms)
st2 = sr.debug_march_stats(True, 2)
st = st2[0]
hist = st2[0][8:15].astype(float)
print("inner passes by active-lane count [1-2, 3-4, 5-8, 9-16, 17-32, 33-48, 49-64]:", (hist / hist.sum()).round(4), "total %.3e" % hist.sum())
wp, li, ws, lst, lh, nw = [float(x) for x in st[:6]]
print("waves %d  wave inner passes (lower bound) %.3e  lane iterations %.3e  -> inner-loop lane utilisation <= %.3f" % (nw, wp, li, li / (64 * wp)))
print("wave march steps %.3e  lane steps %.3e -> march-loop lane utilisation %.3f" % (ws, lst, lst / (64 * ws)))
print("hits %d" % lh)
