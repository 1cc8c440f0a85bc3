#!/usr/bin/env python3
"""Run under `rocprofv3 --kernel-trace`: rmdf_prefilter_env_powers (the four reference powers of a 256x128 map, host in / out) a few
times, so that the trace shows whether the four k_prefilter launches overlap."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rmdf_amd
sr = rmdf_amd.ShaderRenderer(0)
src = np.random.RandomState(3).uniform(0, 4, (128, 256, 3)).astype(np.float32)
for _ in range(4):
    sr.prefilter_env_powers(src, (1.0, 8.0, 64.0, 512.0))       # the reference's four powers: one launch (k_prefilter_fused4)
if os.environ.get("RMDF_PREFILTER_NO_FUSED"):
    pass                                                         # (with that switch: the one-wave kernel, four launches side by side)
for _ in range(3):
    for p in (1.0, 8.0, 64.0, 512.0):
        sr.prefilter_env(src, p)                                 # a power alone: k_prefilter_chan
sr.close()
