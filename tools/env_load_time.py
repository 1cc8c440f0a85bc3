#!/usr/bin/env python3
"""rmdf_load_env_hdr: wall clock of the cache-miss path (fresh directory) and of the cache-hit path."""
import os, sys, time, shutil, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmdf_amd
sr = rmdf_amd.ShaderRenderer(0)
for k in range(3):
    d = tempfile.mkdtemp()
    p = os.path.join(d, "uffizi_512.hdr")
    shutil.copy(rmdf_amd.DEFAULT_ENV_HDR, p)
    t0 = time.perf_counter(); sr.load_env_hdr(p); t1 = time.perf_counter()
    sr.load_env_hdr(p); t2 = time.perf_counter()
    print("cache miss %.2f ms, cache hit %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
