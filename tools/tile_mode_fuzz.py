#!/usr/bin/env python3
"""Fuzz of the boundary's tile mode (rmdf_render_tile with tile_idx >= 0, round 4: tiles rendered ahead of their calls, shadow frame,
copy threads) against a model of what the reference's accumulating frame buffer holds: random sequences of sequential tiles, jumps,
repeats, shader changes, whole-frame calls and size changes; after EVERY call the caller's buffer must equal the model.
usage: tile_mode_fuzz.py [calls=3000] [seed=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rmdf_amd

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sr = rmdf_amd.ShaderRenderer(0)
sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
ref = rmdf_amd.ShaderRenderer(0)                      # a second renderer supplies the expected full frames
ref.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
SIZES = ((128, 72), (96, 40), (250, 131))             # the last one: 8 divides neither side (the plain tile path)
MS = 48
cache = {}


def full(scene, w, h, t):
    key = (scene, w, h, float(np.float32(t)))
    if key not in cache:
        if len(cache) > 400:
            cache.clear()
        cache[key] = ref.render(scene, w, h, t, max_steps=MS, want_f32=False)["rgba8"]
    return cache[key]


w, h = SIZES[0]
expect = np.full((h, w), 0xFF000000, np.uint32)       # a fresh renderer's frame is cleared
buf = np.empty(w * h, np.uint32)
latched_t, latched = 0.0, False
idx = 0
stats = {"tile": 0, "whole": 0, "resize": 0, "jump": 0, "shader": 0}
scene = 2
hist = []
for c in range(calls):
    u = rng.uniform()
    if u < 0.02:                                        # another size: the frame is cleared, the next call latches
        nw, nh = SIZES[rng.randint(len(SIZES))]
        if (nw, nh) != (w, h):
            w, h = nw, nh
            expect = np.full((h, w), 0xFF000000, np.uint32)
            buf = np.empty(w * h, np.uint32)
            latched = False
            stats["resize"] += 1
            hist.append("resize %dx%d" % (w, h))
    if u > 0.97:                                        # a whole frame
        t = float(np.float32(rng.uniform(0.0, 8.0)))
        sr.draw_shader_tile(scene, None, w, h, t, buf, max_steps=MS)
        expect = full(scene, w, h, t).copy()
        latched_t, latched = t, True
        stats["whole"] += 1
        hist.append("whole s%d t%.3f" % (scene, t))
    else:
        v = rng.uniform()
        if v < 0.85:
            idx += 1                                    # the viewer's pattern: the next tile
        elif v < 0.93:
            idx = int(rng.randint(0, 256)); stats["jump"] += 1
        # else: repeat the same tile
        if rng.uniform() < 0.03:
            scene = int(rng.choice([0, 1, 2])); stats["shader"] += 1
        t = float(np.float32(rng.uniform(0.0, 8.0)))
        if idx % 64 == 0 or not latched:                # the first tile latches the time (ShaderRendering.hs:162-176); so does a new size
            latched_t, latched = t, True
        sr.draw_shader_tile(scene, idx, w, h, t, buf, max_steps=MS)
        x0, y0, x1, y1 = rmdf_amd.tile_rect(idx, w, h)
        expect[y0:y1, x0:x1] = full(scene, w, h, latched_t)[y0:y1, x0:x1]
        stats["tile"] += 1
        hist.append("tile s%d idx%d t%.3f latched %.3f rect %s" % (scene, idx, t, latched_t, (x0, y0, x1, y1)))
    if not np.array_equal(buf.reshape(h, w), expect):
        bad = np.argwhere(buf.reshape(h, w) != expect)
        print("MISMATCH at call %d (scene %d, idx %d, %dx%d): %d pixels, first at %s" % (c, scene, idx, w, h, len(bad), bad[0]))
        print("  last operations:", *hist[-8:], sep="\n    ")
        g = buf.reshape(h, w)
        print("  got[0,0] %08x expect[0,0] %08x; got == cleared frame on %d px; got == full(latched) on %d px; expect == cleared on %d px" % (
            g[0, 0], expect[0, 0], int((g == 0xFF000000).sum()), int((g == full(scene, w, h, latched_t)).sum()), int((expect == 0xFF000000).sum())))
        sys.exit(1)
print("tile-mode fuzz: %d calls equal to the model (%s)" % (calls, stats))
