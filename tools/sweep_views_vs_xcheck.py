#!/usr/bin/env python3
"""Product frames (short quotients in the shading tail and in exp / acos / atan) against the three-kernel schedule of librmdf_xcheck
(RMDF_FLAG_PIPELINE: the compiler's divisions everywhere in the tail) over many views of every scene: float colour, steps and iteration
planes must agree bit for bit.   usage: sweep_views_vs_xcheck.py [views_per_scene]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmdf_amd
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 24
a = rmdf_amd.ShaderRenderer(0)
a.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
b = rmdf_amd.ShaderRenderer(0, flags=rmdf_amd.FLAG_PIPELINE)
b.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
print("shading self-test:", a.selftest_shading_math().tolist(), " pinned-math self-test:", a.selftest_pinned_math().tolist())
bad = 0
for scene, (w, h, ms) in ((2, (1920, 1080, 256)), (0, (1280, 720, 128)), (1, (1280, 720, 128)), (3, (1280, 720, 128))):
    diff = 0
    for i in range(nv):
        t = i * 1.37 + (0.0 if i % 3 else 0.5)                     # the orbit camera of main(): a different view every time
        fa, fb = a.render(scene, w, h, t, max_steps=ms), b.render(scene, w, h, t, max_steps=ms)
        for k in ("rgba_f32", "rgba8", "steps", "iters"):
            x, y = fa[k], fb[k]
            same = (x.view(np.uint32) == y.view(np.uint32)) if x.dtype == np.float32 else (x == y)
            if x.dtype == np.float32:
                same = same | (np.isnan(x) & np.isnan(y))
            n = int((~same).sum())
            if n:
                diff += n
                print("  scene %d t=%.2f plane %s: %d differing entries" % (scene, t, k, n))
    print("scene %d: %d views of %dx%d @%d steps, differing entries: %d" % (scene, nv, w, h, ms, diff))
    bad += diff
sys.exit(1 if bad else 0)
