#!/usr/bin/env python3
"""Lobe prefilter at sizes where launch / copy overheads vanish: seconds per power and pair-terms/s.  Measurement aid."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmdf_amd
sr = rmdf_amd.ShaderRenderer(0)
rng = np.random.RandomState(0)
for (w, h) in ((256, 128), (512, 256)):
    img = np.exp(rng.uniform(-3, 3, (h, w, 3))).astype(np.float32)
    for p in (1.0, 8.0, 64.0, 512.0, 3.0):
        sr.prefilter_env(img, p)
        t0 = time.perf_counter(); sr.prefilter_env(img, p); dt = time.perf_counter() - t0
        print("%dx%d power %5.1f: %8.3f ms  %7.1f G pair-terms/s" % (w, h, p, dt * 1e3, (w * h) ** 2 / dt / 1e9), flush=True)
sr.close()
