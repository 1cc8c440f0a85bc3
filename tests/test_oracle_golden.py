"""CPU tier: the oracle against the committed golden vectors (tests/golden/make_fixtures.py).

The reference ships no vectors of its own, so these pin the ORACLE: any change of
compiler flags, libm or source that alters its output shows up here before it can silently move the target
the HIP kernels are compared with."""
import glob
import os
import re

import numpy as np
import pytest

from conftest import ENV_CACHE, GOLD, rel_err

CASES = sorted(glob.glob(os.path.join(GOLD, "render_s*_*.npz")))


def _parse(fn):
    m = re.match(r"render_s(\d)_(\d+)x(\d+)_t(\d+)p(\d+)_m(\d+)\.npz", os.path.basename(fn))
    return int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4) + "." + m.group(5)), int(m.group(6))


@pytest.mark.parametrize("fn", CASES, ids=[os.path.basename(c)[:-4] for c in CASES])
def test_render_matches_golden(orc, env_oracle, fn):
    scene, w, h, t, ms = _parse(fn)
    g = np.load(fn)
    r = orc.render(scene, w, h, t, ms, env_oracle)
    assert np.array_equal(r["steps"], g["steps"])          # bit-exact step counts and hit mask
    assert np.array_equal(r["iters"], g["iters"])          # bit-exact escape-iteration counts
    assert np.array_equal(r["rgba8"], g["rgba8"])
    assert np.array_equal(r["rgba_f32"].view(np.uint32), g["rgba_f32"].view(np.uint32))
    # (the fixtures hold the first six counters; tri_inside joined the struct in round 5, behind them)
    assert list(r["counters"].values())[:6] == [int(x) for x in g["counters"]]


def test_fixture_grid_256x144_pins_the_oracle(orc):
    """SURVEY 8c's fixture grid (every FragmentShader value at in_time 0, 1, 2.5, 7 at 256x144) as digests: the oracle must reproduce its
    own committed planes from the committed cube maps (the GPU tier holds the HIP planes to the same digests)."""
    import hashlib
    import json
    grid = json.load(open(os.path.join(GOLD, "grid_256x144_digests.json")))
    assert len(grid) == 16
    z = np.load(os.path.join(GOLD, "env_cubes_uffizi.npz"))
    env = orc.EnvSet(z["refl"], z["cos1"], z["cos8"])
    for name, d in sorted(grid.items()):
        r = orc.render(d["scene"], d["w"], d["h"], d["time"], d["max_steps"], env)
        for k in ("steps", "iters", "rgba8", "rgba_f32"):
            assert hashlib.sha256(np.ascontiguousarray(r[k]).tobytes()).hexdigest() == d["sha256"][k], (name, k)


def test_render_is_thread_count_invariant(orc, env_oracle):
    a = orc.render(orc.SCENE_MB_POWER8, 64, 36, 0.0, 256, env_oracle, nthreads=1)
    b = orc.render(orc.SCENE_MB_POWER8, 64, 36, 0.0, 256, env_oracle, nthreads=5)
    assert np.array_equal(a["rgba_f32"].view(np.uint32), b["rgba_f32"].view(np.uint32))
    assert a["counters"] == b["counters"]


def test_tile_render_equals_full_frame(orc, env_oracle, rmdf):
    """drawShaderTile tiles (ShaderRendering.hs:183-193) reproduce the full-frame pixels, including tiles
    that start on odd rows (36/8 = 4.5 -> centre-inside rule; helper pixels feed the quad derivatives)."""
    w, h = 64, 36
    full = orc.render(orc.SCENE_MB_POWER8, w, h, 1.0, 256, env_oracle)
    acc = np.zeros((h, w), np.uint32)
    cover = np.zeros((h, w), np.int32)
    for idx in range(64):
        x0, y0, x1, y1 = rmdf.tile_rect(idx, w, h)
        r = orc.render(orc.SCENE_MB_POWER8, w, h, 1.0, 256, env_oracle, rect=(x0, y0, x1, y1))
        acc[y0:y1, x0:x1] = r["rgba8"][y0:y1, x0:x1]
        cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all()                 # the 64 tiles partition the frame
    assert np.array_equal(acc, full["rgba8"])


def test_env_vectors(orc, env_latlongs):
    g = np.load(os.path.join(GOLD, "env_vectors.npz"))
    faces = orc.latlong_to_cube(orc.build_test_latlong())
    assert np.array_equal(faces.astype(np.float16), g["test_faces"])
    for (u, v), rgb in zip(g["bil_uv"], g["bil_rgb"]):
        assert np.array_equal(orc.pixel_at_bilinear(env_latlongs["refl"], u, v), rgb)
    small = orc.resize_hdr(env_latlongs["refl"], 32)
    assert np.array_equal(small, g["small32"])
    for i, p in enumerate((1.0, 8.0)):
        # the vector was written with the literal libm powf form; the pinned cos^p stays within 1e-6 of it
        assert np.array_equal(orc.cosine_convolve(small, p, pow_mode=0), g["prefilter32"][i])
        assert rel_err(orc.cosine_convolve(small, p, pow_mode=1), g["prefilter32"][i]).max() <= 1e-6
    uf = orc.latlong_to_cube(env_latlongs["refl"])
    assert np.array_equal(uf[:, ::17, ::17], g["uffizi_faces_sample"])
    assert np.array_equal(orc.cube_pad_f16(uf)[:, :3, :3], g["uffizi_padded_corner"])


def test_committed_cache_files_are_the_oracle_prefilter(orc, env_latlongs, rmdf):
    """tests/golden/env_cache/uffizi_512_cache_pow_*.hdr are resizeHDRImage 256 -> cosineConvolveHDREnvMap -> RGBE of the
    ORACLE (ShaderRendering.hs:131-149) -- the expected output of the product's cache-miss path.  Recompute all four with
    the pinned cos^p (a few seconds on 8 cores) and compare the file bytes; then bound the distance of the pin from the
    literal libm powf call: <= 1e-6 relative before RGBE, RGBE bytes equal except +-1 mantissa step on <= 1e-4 of them."""
    small = orc.resize_hdr(env_latlongs["refl"], 256)
    assert small.shape == (128, 256, 3)
    for p in (1.0, 8.0, 64.0, 512.0):
        data = open(os.path.join(ENV_CACHE, "uffizi_512_cache_pow_%s.hdr" % repr(p)), "rb").read()
        pinned = orc.cosine_convolve(small, p, pow_mode=1)
        assert orc.hdr_encode(pinned) == data, p
        img = orc.hdr_decode(data)
        assert img.shape == (128, 256, 3)
        # RGBE encode -> decode is idempotent on already-quantised data
        assert np.array_equal(orc.hdr_decode(orc.hdr_encode(img)), img)
        # the cache is a cosine-weighted average of non-negative radiance: bounded by the source range
        assert img.min() >= 0 and img.max() <= small.max()
        if p in (8.0, 512.0):
            libm = orc.cosine_convolve(small, p, pow_mode=0)
            assert rel_err(pinned, libm).max() <= 1e-6
            a, b = np.frombuffer(orc.hdr_encode(libm), np.uint8), np.frombuffer(data, np.uint8)
            d = np.abs(a.astype(np.int32) - b.astype(np.int32))
            assert d.max() <= 1 and (d != 0).mean() <= 1e-4


def test_fractal_vectors(orc):
    g = np.load(os.path.join(GOLD, "fractal_vectors.npz"))
    for tick, tn in ((0.0, "0p0"), (3.7, "3p7")):
        for smooth in (0, 1):
            assert np.array_equal(orc.julia_animated(64, 64, smooth, tick), g["julia_t%s_s%d" % (tn, smooth)])
    for smooth in (0, 1):
        assert np.array_equal(orc.mandelbrot(96, 64, smooth), g["mandelbrot_s%d" % smooth])
