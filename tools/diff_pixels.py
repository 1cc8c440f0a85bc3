#!/usr/bin/env python3
"""Where does a product frame differ from the alternative schedule's (librmdf_xcheck, FLAG_FLAT_MARCH)?  Diagnostic.
usage: diff_pixels.py [scene w h max_steps time]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmdf_amd
scene, w, h, ms = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (2, 1920, 1080, 256)
t = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
a = rmdf_amd.ShaderRenderer(0)
a.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
print("shading selftest:", a.selftest_shading_math().tolist())
b = rmdf_amd.ShaderRenderer(0, flags=rmdf_amd.FLAG_FLAT_MARCH, xcheck=True)
b.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
fa, fb = a.render(scene, w, h, t, max_steps=ms), b.render(scene, w, h, t, max_steps=ms)
for k in fa:
    if isinstance(fa[k], np.ndarray):
        d = np.argwhere(fa[k].reshape(h, w, -1) != fb[k].reshape(h, w, -1))
        print(k, "differing entries:", len(d))
        for y, x, c in d[:12]:
            print("   px (%d, %d) ch %d: product %r alt %r ; steps %s iters %s" % (x, y, c, fa[k].reshape(h, w, -1)[y, x, c], fb[k].reshape(h, w, -1)[y, x, c],
                  fa.get("steps", np.zeros((h, w))).reshape(h, w)[y, x], fa.get("iters", np.zeros((h, w))).reshape(h, w)[y, x]))
