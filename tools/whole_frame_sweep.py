#!/usr/bin/env python3
"""rmdf_render_tile(tile_idx = -1, pageable pointer) -- the reference viewer's per-frame call: ms per call by row-band count
(rmdf_config.reserved[2]), by the way a band's rows reach the host (reserved[3]: 0 copy behind each band's kernel, 1 the band kernels' mirror stores, 2 one launch with mirror stores and band flags, 3 the same dispatched band by band)
and by host copy threads (reserved[1]).  usage: whole_frame_sweep.py [reps [product]]   ("product": hand-over modes 0 and 1 only -- the modes the library
named by RMDF_LIB, or librmdf.so, has; modes 2 and 3 always run on librmdf_xcheck.so)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rmdf_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
product_only = len(sys.argv) > 2 and sys.argv[2] == "product"
cases = ((2, 1920, 1080, 256), (0, 1280, 720, 128))
def run(**kw):
    # (hand-over modes 2 and 3 -- one launch, band flags -- exist in the cross-check build only)
    sr = rmdf_amd.ShaderRenderer(0, xcheck=kw.get("frame_mirror", 0) >= 2, **kw); sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
    out = []
    for scene, w, h, ms in cases:
        hb = np.empty(w * h, np.uint32)
        for _ in range(8): sr.draw_shader_tile(scene, None, w, h, 0.0, hb, max_steps=ms)
        best = 1e9; tot = 0.0
        for blk in range(3):
            t0 = time.perf_counter()
            for _ in range(reps): sr.draw_shader_tile(scene, None, w, h, 0.0, hb, max_steps=ms)
            dt = (time.perf_counter() - t0) / reps; best = min(best, dt); tot += dt
        out.append((best * 1e3, tot / 3 * 1e3, w * h / 1e6 / best))
    sr.close()
    return out
print("bands mirror threads | headline 1080p: best ms, mean ms, Mpixels/s | Cornell 720p: best ms, mean ms, Mpixels/s")
for threads in (0, 8, 32):
    for mirror in (0, 1, 2, 3):
        for bands in (1, 2, 3, 4, 6, 8, 12, 16):
            if threads and bands not in (1, 4, 8): continue
            if mirror < 2 and bands > 4: continue
            if mirror >= 2 and product_only: continue
            r = run(frame_bands=bands, frame_mirror=mirror, copy_threads=threads)
            print("%5d %6d %7d | %.4f %.4f %8.1f | %.4f %.4f %8.1f" % (bands, mirror, threads, r[0][0], r[0][1], r[0][2], r[1][0], r[1][1], r[1][2]), flush=True)
