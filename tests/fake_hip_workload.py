#!/usr/bin/env python3
"""The library's host paths against the HIP test double (tests/fake_hip.cpp) -- no GPU.  Run in a child process with
LD_PRELOAD=tests/libfake_hip.so (tests/test_host_logic.py: test_host_paths_against_the_hip_double; tools/asan_host.sh adds the
AddressSanitizer build).  The double's "kernels" write a hash of (pixel, frame, scene, camera, step limit, cube-map contents) where the
real kernels write colours, so what is checked here is the HOST's work: that every way of asking for a frame hands back the same
frame, in the right place, touching nothing else; that tiles, bands, shards and cache files end up where they belong; that error paths
return errors; that nothing leaks.  Nothing here says anything about a kernel.
usage: fake_hip_workload.py [xcheck] [quick] [only=whole|tiles|shards|env|random|calls|args|leaks|exchange ...]"""
import ctypes as C
import os
import shutil
import sys
import tempfile
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rmdf_amd as rmdf                                      # noqa: E402

if os.environ.get("FAKE_HIP_WORKLOAD_XCHECK_LIB"):             # tools/asan_host.sh: the sanitizer build of the cross-check library
    rmdf.XCHECK_LIB_PATH = os.environ["FAKE_HIP_WORKLOAD_XCHECK_LIB"]
XCHECK = "xcheck" in sys.argv[1:]
QUICK = "quick" in sys.argv[1:]
FAKE = C.CDLL(os.environ.get("FAKE_HIP_LIB", os.path.join(ROOT, "tests", "libfake_hip.so")))  # (already in the process: LD_PRELOAD)
FAKE.fake_hip_counters.argtypes = [C.c_void_p]
FAKE.fake_hip_fail_malloc_in.argtypes = [C.c_longlong]
CANARY = 0xDEADBEEF


def counters():
    a = (C.c_longlong * 8)()
    FAKE.fake_hip_counters(a)
    return dict(zip(("dev", "host", "streams", "events", "launches", "unknown", "copied", "kernels"), a))


def set_env(sr, seed, sizes=(16, 8, 8)):
    rng = np.random.default_rng(seed)
    for slot, W in zip((rmdf.ENV_REFLECTION, rmdf.ENV_COS_1, rmdf.ENV_COS_8), sizes):
        sr.set_env_cube(slot, rng.uniform(0.0, 2.0, (6, W, W, 3)).astype(np.float32))


def whole(sr, scene, w, h, t, ms, pad=32):
    fb = np.full(w * h + 2 * pad, CANARY, np.uint32)
    sr.draw_shader_tile(scene, None, w, h, t, fb[pad:pad + w * h], max_steps=ms)
    assert (fb[:pad] == CANARY).all() and (fb[pad + w * h:] == CANARY).all(), "whole-frame call wrote outside the frame"
    return fb[pad:pad + w * h].reshape(h, w).copy()


def section_whole_frame_paths():
    sizes = [(2, 1920, 1080, 64), (0, 1283, 721, 32), (2, 200, 100, 64), (3, 600, 599, 24), (1, 33, 17, 16), (2, 8, 8, 8), (2, 1, 1, 4)]
    if QUICK:
        sizes = sizes[1:]
    ref = {}
    for bands, mode in [(0, 0), (1, 0), (16, 0), (5, 1), (2, 1), (1, 2), (4, 2), (7, 3), (16, 3)]:
        if mode >= 2 and not XCHECK:
            # the one-launch hand-over has had no green run on hardware: the product library refuses it (RMDF_E_UNSUPPORTED)
            try:
                rmdf.ShaderRenderer(0, frame_bands=bands, frame_mirror=mode)
                raise AssertionError("librmdf.so accepted rmdf_config.reserved[3] = %d" % mode)
            except rmdf.RmdfError as e:
                assert e.code == -6, e.code
            continue
        sr = rmdf.ShaderRenderer(0, xcheck=XCHECK, frame_bands=bands, frame_mirror=mode)
        set_env(sr, 1)
        for key in sizes:
            scene, w, h, ms = key
            planes = sr.render(scene, w, h, 0.7, max_steps=ms)
            ref.setdefault(key, planes["rgba8"].copy())
            assert np.array_equal(planes["rgba8"], ref[key]), (bands, mode, key)
            assert (planes["rgba8"] >> 24 == 0xFF).all() and (planes["steps"] & 0x7fff < ms).all()
            for rep in range(3):                              # (from the second call on the strips are dispatched in cost order)
                assert np.array_equal(whole(sr, scene, w, h, 0.7, ms), ref[key]), (bands, mode, key, rep)
            tiled = np.zeros(w * h, np.uint32)
            sr.draw_shader_tile(scene, 9, w, h, 0.7, tiled, max_steps=ms)       # not tile 0: keeps the latched frame
            assert np.array_equal(tiled.reshape(h, w), ref[key]), (bands, mode, key, "tile after whole frame")
            other = whole(sr, scene, w, h, 1.9, ms)           # another camera: another frame
            assert w * h < 64 or not np.array_equal(other, ref[key])
        sr.close()
    for bad in (dict(frame_bands=17), dict(frame_bands=-1), dict(frame_mirror=4), dict(copy_threads=65)):
        try:
            rmdf.ShaderRenderer(0, xcheck=XCHECK, **bad)
            raise AssertionError("accepted %r" % bad)
        except rmdf.RmdfError:
            pass
    print("ok whole-frame paths (%d configurations x %d frames)" % (9, len(sizes)), flush=True)


def section_tile_mode():
    for threads in (1, 3, 0, 64):
        sr = rmdf.ShaderRenderer(0, xcheck=XCHECK, copy_threads=threads)
        set_env(sr, 2)
        for (scene, w, h, ms) in ((2, 640, 480, 64), (0, 333, 187, 32), (2, 40, 24, 16), (3, 1920, 1080, 32)):
            if QUICK and w > 1000:
                continue
            full = whole(sr, scene, w, h, 1.5, ms)
            fb = np.full(w * h + 64, CANARY, np.uint32)
            for idx in range(64):
                sr.draw_shader_tile(scene, idx, w, h, 1.5 if idx == 0 else 99.0 + idx, fb[32:32 + w * h], max_steps=ms)   # time latches on tile 0
                assert (fb[:32] == CANARY).all() and (fb[32 + w * h:] == CANARY).all()
            assert np.array_equal(fb[32:32 + w * h].reshape(h, w), full), (threads, scene, w, h)
            # tiles in another order, with jumps: every call returns the accumulated frame
            order = list(np.random.default_rng(w).permutation(64))
            order.remove(0)
            fb[:] = 0
            sr.draw_shader_tile(scene, 0, w, h, 2.5, fb[32:32 + w * h], max_steps=ms)
            for idx in order:
                sr.draw_shader_tile(scene, int(idx), w, h, 0.0, fb[32:32 + w * h], max_steps=ms)
            assert np.array_equal(fb[32:32 + w * h].reshape(h, w), whole(sr, scene, w, h, 2.5, ms)), (threads, scene, w, h, "permuted")
        sr.close()
    # the environment changes between two tile calls of one frame: tiles issued ahead of their calls must not show the old one
    sr = rmdf.ShaderRenderer(0, xcheck=XCHECK)
    w, h, ms = 640, 360, 32
    set_env(sr, 3)
    a = whole(sr, 2, w, h, 0.3, ms)
    set_env(sr, 4)
    b = whole(sr, 2, w, h, 0.3, ms)
    assert not np.array_equal(a, b)
    for swap_at in (1, 20, 62, 63):
        set_env(sr, 3)
        fb = np.zeros(w * h, np.uint32)
        want = np.zeros((h, w), np.uint32)
        for idx in range(64):
            if idx == swap_at:
                set_env(sr, 4)
            sr.draw_shader_tile(2, idx, w, h, 0.3, fb, max_steps=ms)
            x0, y0, x1, y1 = rmdf.tile_rect(idx, w, h)
            want[y0:y1, x0:x1] = (a if idx < swap_at else b)[y0:y1, x0:x1]
            assert np.array_equal(fb.reshape(h, w)[y0:y1, x0:x1], want[y0:y1, x0:x1]), (swap_at, idx)
        assert np.array_equal(fb.reshape(h, w), want), swap_at
    # a registered buffer (bookkeeping only since round 5) behaves like any other; unregistering twice is an error
    big = np.zeros(w * h + 4096, np.uint32)
    sr.register_host_buffer(big)
    sr.register_host_buffer(big)
    view = big[1024:1024 + w * h]
    sr.draw_shader_tile(2, None, w, h, 0.3, view, max_steps=ms)
    assert np.array_equal(view.reshape(h, w), b) and (big[:1024] == 0).all() and (big[1024 + w * h:] == 0).all()
    sr.unregister_host_buffer(big)
    try:
        sr.unregister_host_buffer(big)
        raise AssertionError("second unregister accepted")
    except rmdf.RmdfError:
        pass
    sr.close()
    print("ok tile mode, environment swap, registered buffer", flush=True)


def section_shards():
    sr = rmdf.ShaderRenderer(0, xcheck=XCHECK)
    set_env(sr, 5)
    for (w, h, ms) in ((640, 360, 32), (1920, 1080, 32), (64, 64, 8)):
        if QUICK and w > 1000:
            continue
        full = whole(sr, 2, w, h, 0.0, ms)
        for costs in (None, sr.probe_tile_costs(2, w, h, 0.0, ms)):
            sr.set_shard_costs(costs)
            sr.set_shard_root_handicap(0.0 if costs is None else 0.25)
            for n in (1, 2, 3, 8, 64):
                slots = rmdf.shard_slots(n)
                gathered = np.full((n, slots, h // 8, w // 8), CANARY, np.uint32)
                seen = []
                for r in range(n):
                    seen += list(sr.shard_tiles(r, n))
                    sr.render_shard_device(2, w, h, 0.0, ms, r, n, gathered[r].ctypes.data)
                assert sorted(seen) == list(range(64)), (n, seen)
                frame = np.full(w * h + 64, CANARY, np.uint32)
                sr.assemble_shards_device(w, h, n, gathered.ctypes.data, frame[32:].ctypes.data)
                sr.synchronize()
                assert np.array_equal(frame[32:32 + w * h].reshape(h, w), full), (w, h, n, costs is not None)
                assert (frame[:32] == CANARY).all() and (frame[32 + w * h:] == CANARY).all()
        sr.set_shard_costs(None)
        sr.set_shard_root_handicap(0.0)
    # frames in flight on CALLER streams (bench.py's schedule; the strip-order tables are per stream, the least recently used set is
    # recycled beyond 32 streams): rectangles and shards on 5, then 40 streams, interleaved, every result == the frame
    FAKE.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    FAKE.hipStreamDestroy.argtypes = [C.c_void_p]
    FAKE.hipStreamSynchronize.argtypes = [C.c_void_p]
    w, h, ms = 640, 360, 16
    full = {t: whole(sr, 2, w, h, t, ms) for t in (0.0, 1.0, 2.5)}
    for nstreams in (5, 40):
        streams = []
        for _ in range(nstreams):
            st = C.c_void_p()
            assert FAKE.hipStreamCreateWithFlags(C.byref(st), 1) == 0
            streams.append(st)
        n = 8
        slots = rmdf.shard_slots(n)
        bufs = [np.zeros((h, w), np.uint32) for _ in streams]
        shards = [np.zeros((slots, h // 8, w // 8), np.uint32) for _ in streams]
        for rep in range(3):                                   # (from the second round on: cost-ordered dispatch from each stream's own table)
            for k, st in enumerate(streams):
                t = (0.0, 1.0, 2.5)[k % 3]
                sr.render_rect_device(2, w, h, t, ms, (0, 0, w, h), d_rgba8=bufs[k].ctypes.data, stream=st.value)
                sr.render_shard_device(2, w, h, t, ms, k % n, n, shards[k].ctypes.data, stream=st.value)
            for k, st in enumerate(streams):
                assert FAKE.hipStreamSynchronize(st) == 0
                t = (0.0, 1.0, 2.5)[k % 3]
                assert np.array_equal(bufs[k], full[t]), (nstreams, rep, k)
                for slot, idx in enumerate(sr.shard_tiles(k % n, n)):
                    x0, y0, x1, y1 = rmdf.tile_rect(idx, w, h)
                    assert np.array_equal(shards[k][slot], full[t][y0:y1, x0:x1]), (nstreams, rep, k, slot)
        for st in streams:
            assert FAKE.hipStreamDestroy(st) == 0
    # supersampling: 2x2 rays per pixel, one mip level
    for (w, h) in ((320, 180), (64, 40)):
        got = sr.render_supersampled(2, w, h, 1, 0.0, max_steps=16)
        hi = whole(sr, 2, 2 * w, 2 * h, 0.0, 16).view(np.uint8).reshape(2 * h, 2 * w, 4).astype(np.uint32)
        want = ((hi[0::2, 0::2] + hi[0::2, 1::2] + hi[1::2, 0::2] + hi[1::2, 1::2] + 2) >> 2).astype(np.uint8).view(np.uint32).reshape(h, w)
        assert np.array_equal(got, want), (w, h)
    sr.close()
    print("ok shards (static and cost-aware deal; 1, 2, 3, 8, 64 ranks), frames in flight on 5 and 40 caller streams, supersampling", flush=True)


def section_env_pipeline():
    sr = rmdf.ShaderRenderer(0, xcheck=XCHECK)
    rng = np.random.default_rng(6)
    for (lw, lh) in ((96, 48), (51, 25), (512, 256)):
        ll = rng.uniform(0.0, 3.0, (lh, lw, 3)).astype(np.float32)
        sr.set_env_latlong(rmdf.ENV_REFLECTION, ll)
        padded = sr.get_env_cube_padded(rmdf.ENV_REFLECTION)
        assert padded.shape[0] == 6 and padded.shape[1] == padded.shape[2] == lw // 3 + 2, padded.shape
        small = sr.resize_latlong(ll, 24)
        assert small.shape[1] == 24 and small.shape[2] == 3
    for (w, h) in ((256, 128), (100, 37), (8, 3), (300, 150)):
        src = rng.uniform(0.0, 3.0, (h, w, 3)).astype(np.float32)
        outs = sr.prefilter_env_powers(src, (1.0, 8.0, 64.0, 512.0))
        assert len(outs) == 4 and all(o.shape == (h, w, 3) for o in outs)
        for p, o in zip((1.0, 8.0, 64.0, 512.0), outs):
            assert np.array_equal(sr.prefilter_env(src, p), o), (w, h, p)       # one power alone: another kernel, the same (stand-in) map
        assert sr.prefilter_env(src, 3.5).shape == (h, w, 3)
        assert len(sr.prefilter_env_powers(src, (8.0, 512.0))) == 2
    # the cache files of withShaderRenderer's pipeline: built on the first load (private name + rename), read on the second
    with tempfile.TemporaryDirectory() as d:
        hdr = os.path.join(d, "probe.hdr")
        shutil.copy(rmdf.DEFAULT_ENV_HDR, hdr)
        before = counters()["launches"]
        sr.load_env_hdr(hdr)
        first = counters()["launches"] - before
        files = sorted(os.listdir(d))
        assert len(files) == 5 and not [f for f in files if ".tmp" in f], files
        a = whole(sr, 2, 200, 100, 0.0, 16)
        sizes = {f: os.path.getsize(os.path.join(d, f)) for f in files}
        before = counters()["launches"]
        sr.load_env_hdr(hdr)
        second = counters()["launches"] - before
        assert second < first, (first, second)               # no prefilter launches the second time
        assert np.array_equal(whole(sr, 2, 200, 100, 0.0, 16), a)
        assert sizes == {f: os.path.getsize(os.path.join(d, f)) for f in sorted(os.listdir(d))}
        # a damaged cache file is an error the caller sees, not a crash
        victim = [f for f in files if "cache" in f][0]
        open(os.path.join(d, victim), "wb").write(b"#?RADIANCE\n\n-Y 128 +X 256\n" + b"\x02\x02\x01\x00" + b"\xff" * 40)
        try:
            sr.load_env_hdr(hdr)
            damaged = "accepted (rebuilt)"
        except rmdf.RmdfError as e:
            damaged = "error %d" % e.code
        for missing in (os.path.join(d, "nope.hdr"), d):
            try:
                sr.load_env_hdr(missing)
                raise AssertionError("loaded %s" % missing)
            except rmdf.RmdfError:
                pass
    sr.close()
    # a renderer without an environment refuses to render
    sr = rmdf.ShaderRenderer(0, xcheck=XCHECK)
    try:
        sr.render(2, 64, 64, 0.0)
        raise AssertionError("rendered without an environment")
    except rmdf.RmdfError as e:
        assert "environment" in str(e)
    for bad in ((7, 64, 64, 16), (2, 0, 64, 16), (2, 64, 40000, 16), (2, 64, 64, 40000)):
        set_env(sr, 1)
        try:
            sr.render(bad[0], bad[1], bad[2], 0.0, max_steps=bad[3])
            raise AssertionError("accepted %r" % (bad,))
        except (rmdf.RmdfError, ValueError):
            pass
    sr.close()
    print("ok env pipeline, cache files (damaged cache: %s), argument errors" % damaged, flush=True)


def section_random_call_sequences():
    """A model of what the boundary promises (ShaderRendering.hs:162-193 as include/rmdf.h states it: time, step limit and size latch on tile 0
    or a whole-frame call; a new size clears the accumulating frame; the shader value and the environment are those of the call; every
    call hands back the whole accumulating frame) against the library, over random call sequences: tiles in any order and repeated, whole
    frames through the fast path and through the plane-writing path, size changes, environment changes, shader changes -- with tile jobs
    issued ahead, band hand-overs and the shadow frame all in play.  The expected pixels come from a second renderer that only ever
    renders whole frames through the plain path."""
    rng = np.random.default_rng(11)
    ref = rmdf.ShaderRenderer(0, xcheck=XCHECK, frame_bands=1, frame_mirror=0)
    cache = {}

    def want_frame(scene, w, h, t, ms, seed):
        key = (scene, w, h, t, ms, seed)
        if key not in cache:
            if cache.get("seed") != seed:
                set_env(ref, seed)
                cache["seed"] = seed
            cache[key] = ref.render(scene, w, h, t, max_steps=ms, want_f32=False)["rgba8"].copy()
        return cache[key]

    sizes = [(128, 72), (64, 40), (200, 100), (33, 17), (640, 360)]
    nops = 0
    for trial in range(6 if QUICK else 16):
        cfg = dict(frame_bands=int(rng.integers(0, 6)), frame_mirror=int(rng.integers(0, 4 if XCHECK else 2)), copy_threads=int(rng.choice([0, 1, 3])))
        sr = rmdf.ShaderRenderer(0, xcheck=XCHECK, **cfg)
        seed = int(rng.integers(100, 104))
        set_env(sr, seed)
        cur, lt, lms, model = None, None, None, None
        for step in range(60 if QUICK else 120):
            op = rng.choice(["tile", "tile", "tile", "tile", "next", "whole", "planes", "planes_tile", "env", "size"])
            scene = int(rng.integers(0, 4))
            t = float(rng.choice([0.0, 1.0, 2.5, 7.0]))
            ms = int(rng.choice([8, 16, 33]))
            w, h = cur if cur and op != "size" else sizes[int(rng.integers(0, len(sizes)))]
            if op == "env":
                seed = int(rng.integers(100, 104))
                set_env(sr, seed)
                continue
            if op == "size":
                op = "tile"
            if op == "next":                                   # the tile after the last one: the pattern the jobs issued ahead are made for
                op, idx = "tile", (locals().get("last_idx", -1) + 1) % 128
            else:
                idx = int(rng.integers(0, 128))
            whole_call = op in ("whole", "planes")
            first = whole_call or idx % 64 == 0
            if first or cur is None or (w, h) != cur:
                if cur is None or (w, h) != cur:
                    model = np.full((h, w), 0xFF000000, np.uint32)
                cur, lt, lms = (w, h), t, ms
            src = want_frame(scene, w, h, lt, lms, seed)
            if whole_call:
                model[:] = src
            else:
                x0, y0, x1, y1 = rmdf.tile_rect(idx, w, h)
                model[y0:y1, x0:x1] = src[y0:y1, x0:x1]
                last_idx = idx
            if op in ("tile", "whole"):
                fb = np.full(w * h + 16, CANARY, np.uint32)
                sr.draw_shader_tile(scene, None if whole_call else idx, w, h, t, fb[8:8 + w * h], max_steps=ms)
                assert (fb[:8] == CANARY).all() and (fb[8 + w * h:] == CANARY).all()
                got = fb[8:8 + w * h].reshape(h, w)
            else:
                got = sr.render(scene, w, h, t, max_steps=ms, tile_idx=None if whole_call else idx, want_f32=bool(rng.integers(0, 2)))["rgba8"]
            assert np.array_equal(got, model), (trial, step, op, scene, idx, (w, h), t, ms, cfg)
            nops += 1
        sr.close()
    ref.close()
    print("ok %d random calls against the model of the boundary (tiles / whole frames / planes, size, shader and environment changes)" % nops, flush=True)


def section_runtime_calls_per_frame():
    """What a frame costs the host in HIP runtime calls (each is microseconds of host time; the band sweep of round 5 lost to them): counted
    by the double, held to a budget so that a change that adds calls to the per-frame paths shows up on the CPU tier."""
    FAKE.fake_hip_calls.argtypes = [C.c_char_p, C.c_int, C.c_int]

    def calls():
        b = C.create_string_buffer(16384)
        FAKE.fake_hip_calls(b, 16384, 1)
        d = {k: int(v) for k, v in (kv.split("=") for kv in b.value.decode().split(";") if kv)}
        return sum(v for k, v in d.items() if k not in ("hipEventQuery", "hipStreamQuery")), d      # (polls depend on timing)

    w, h, n = 1920, 1080, 10
    fb = np.zeros(w * h, np.uint32)
    rows = []
    for name, cfg, budget in (("default (two mirror bands)", dict(), 14), ("one mirror band", dict(frame_bands=1, frame_mirror=1), 8),
                              ("two bands, copies behind the launches", dict(frame_bands=2, frame_mirror=0), 16),
                              ("eight mirror bands", dict(frame_bands=8, frame_mirror=1), 34)) + \
                             ((("one launch, eight flagged bands", dict(frame_bands=8, frame_mirror=2), 5),) if XCHECK else ()):
        sr = rmdf.ShaderRenderer(0, xcheck=XCHECK, **cfg)
        set_env(sr, 1)
        for _ in range(3):
            sr.draw_shader_tile(2, None, w, h, 0.0, fb, max_steps=8)
        calls()
        for _ in range(n):
            sr.draw_shader_tile(2, None, w, h, 0.0, fb, max_steps=8)
        per_frame, d = calls()
        assert per_frame <= budget * n, (name, per_frame / n, d)
        rows.append("%s %.1f" % (name, per_frame / n))
        if not cfg:
            for rep in range(2):
                for idx in range(64):
                    sr.draw_shader_tile(2, idx, w, h, 0.0, fb, max_steps=8)
                per_tile, d = calls()
            assert per_tile <= 8 * 64, (per_tile / 64.0, d)
            rows.append("tile call %.1f" % (per_tile / 64.0))
        sr.close()
    print("ok runtime calls per 1080p frame: " + "; ".join(rows), flush=True)


def section_argument_sweep():
    """Every entry point of the ABI whose first parameter is the ctx, with a valid ctx and null pointers / empty or unusable paths / zero,
    one, negative and large scalars for everything else (four variants each): an error code or a harmless success -- never a crash, never
    a write through a null pointer -- and the renderer still works afterwards.  (Entry points without a ctx get all-zero arguments.)"""
    L = rmdf.load_library(xcheck=XCHECK)
    names = list(rmdf.ABI_SYMBOLS) + (list(rmdf.XCHECK_SYMBOLS) if XCHECK else [])
    no_ctx = {"rmdf_create", "rmdf_create_ex", "rmdf_get_cornell_vertices", "rmdf_get_shader_constants", "rmdf_debug_cornell_table", "rmdf_debug_hdr_decode",
              "rmdf_debug_hdr_encode", "rmdf_comm_get_unique_id", "rmdf_shard_tiles", "rmdf_save_png", "rmdf_debug_cornell_masks", "rmdf_debug_cube_uv_table",
              "rmdf_debug_lobe_tables", "rmdf_debug_camera", "rmdf_is_tile_idx_first_tile", "rmdf_is_tile_idx_last_tile"}
    is_ptr = lambda t: t is C.c_void_p or t is C.c_char_p or (isinstance(t, type) and issubclass(t, C._Pointer))
    sr = rmdf.ShaderRenderer(0, xcheck=XCHECK)
    set_env(sr, 1)
    calls = 0
    for n in names:
        f = getattr(L, n)
        if n in ("rmdf_destroy", "rmdf_last_error") or f.argtypes is None:
            continue
        if n in no_ctx:
            if n in ("rmdf_comm_get_unique_id",) and not os.environ.get("RMDF_RCCL_LIB"):
                continue                                        # (would load the real RCCL)
            f(*[(0.0 if t in (C.c_double, C.c_float) else (None if is_ptr(t) else 0)) for t in f.argtypes])
            calls += 1
            continue
        if n.startswith("rmdf_comm_") and not os.environ.get("RMDF_RCCL_LIB"):
            continue
        assert f.argtypes[0] is C.c_void_p, n
        for variant in range(4):
            args = [sr._ctx]
            for t in f.argtypes[1:]:
                if t in (C.c_double, C.c_float):
                    args.append([0.0, 1.0, -1.0, 1e30][variant])
                elif t is C.c_char_p:
                    args.append([None, b"", b"/nonexistent/x", b"/"][variant])
                elif is_ptr(t):
                    args.append(None)
                elif t is C.c_size_t:
                    args.append([0, 1, 64, 4096][variant])
                else:
                    args.append([0, 1, -1, 64][variant])
            f(*args)
            calls += 1
    fb = np.zeros(200 * 100, np.uint32)
    sr.draw_shader_tile(2, None, 200, 100, 0.0, fb, max_steps=8)
    assert (fb >> 24 == 0xFF).all()
    sr.close()
    print("ok %d calls with null pointers and out-of-range scalars: errors, no crash" % calls, flush=True)


def section_leaks_and_failed_allocations():
    assert counters()["dev"] == 0 and counters()["host"] == 0 and counters()["streams"] == 0 and counters()["events"] == 0, counters()
    for _ in range(3):
        sr = rmdf.ShaderRenderer(0, xcheck=XCHECK)
        set_env(sr, 7)
        whole(sr, 2, 640, 360, 0.0, 16)
        sr.close()
        c = counters()
        assert c["dev"] == 0 and c["host"] == 0 and c["streams"] == 0 and c["events"] == 0, c
    # the n-th allocation fails: an error, not a crash; and everything is given back
    failed = 0
    for n in range(1, 60 if not QUICK else 25):
        FAKE.fake_hip_fail_malloc_in(n)
        sr = None
        try:
            sr = rmdf.ShaderRenderer(0, xcheck=XCHECK)
            set_env(sr, 7)
            sr.render(0, 200, 100, 0.0, max_steps=8)
            whole(sr, 2, 1920, 1080, 0.0, 8)
            fb = np.zeros(200 * 100, np.uint32)
            for idx in range(6):
                sr.draw_shader_tile(2, idx, 200, 100, 0.0, fb, max_steps=8)
            sr.prefilter_env(np.ones((16, 32, 3), np.float32), 8.0)
        except rmdf.RmdfError as e:
            failed += 1
            assert e.code in (-2, -7, -3), (n, e.code, str(e))
        finally:
            FAKE.fake_hip_fail_malloc_in(0)
            if sr is not None:
                sr.close()
        c = counters()
        assert c["dev"] == 0 and c["host"] == 0 and c["streams"] == 0 and c["events"] == 0, (n, c)
    assert failed >= 10, failed
    print("ok no leaks over create / destroy; %d injected allocation failures: an error each, nothing left behind" % failed, flush=True)


def section_exchange():
    """N ranks as N threads of this process, each with its own renderer and its own rank of one communicator of the RCCL double
    (tests/fake_rccl.c, honoured by the cross-check library only)."""
    if not XCHECK or not os.environ.get("RMDF_RCCL_LIB"):
        print("skip exchange (needs the cross-check library and RMDF_RCCL_LIB)", flush=True)
        return
    w, h, ms, S = 640, 360, 16, 3
    for n in (2, 3, 8):
        uid = rmdf.comm_get_unique_id(xcheck=True)
        errors, results = [], {}

        def rank_main(rank):
            try:
                sr = rmdf.ShaderRenderer(0, xcheck=True)
                set_env(sr, 8)
                sr.comm_init(uid, rank, n)
                assert sr.comm_info() == (rank, n)
                slots = rmdf.shard_slots(n)
                gath = [np.zeros((n, slots, h // 8, w // 8), np.uint32) if rank == 0 else None for _ in range(S)]
                shard = [gath[k][0] if rank == 0 else np.zeros((slots, h // 8, w // 8), np.uint32) for k in range(S)]
                frame = [np.zeros((h, w), np.uint32) if rank == 0 else None for _ in range(S)]
                single = sr.render(2, w, h, 0.0, max_steps=ms, want_f32=False)["rgba8"].copy() if rank == 0 else None

                def frames(times, tag, check=True):
                    for i, t in enumerate(times):
                        k = i % S
                        sr.render_frame_sharded_device(2, w, h, t, ms, shard[k].ctypes.data, gath[k].ctypes.data if rank == 0 else 0,
                                                       frame[k].ctypes.data if rank == 0 else 0)
                    sr.synchronize()
                    if rank == 0 and check:
                        for i, t in enumerate(times):
                            if i >= len(times) - S and t == 0.0:
                                assert np.array_equal(frame[i % S], single), "%s: frame %d differs from the single launch" % (tag, i)

                frames([0.0] * (2 * S), "static deal")
                cost = sr.probe_tile_costs(2, w, h, 0.0, ms)
                sr.set_shard_costs(cost)
                sr.set_shard_root_handicap(0.25)
                sr.comm_verify_deal()
                frames([0.0] * S, "verified cost-aware deal")
                bad = np.array(cost, np.float32).copy()
                if rank == n - 1:
                    bad[::3] *= 7.0
                sr.set_shard_costs(bad)
                try:
                    sr.comm_verify_deal()
                    raise AssertionError("rmdf_comm_verify_deal accepted different deals")
                except rmdf.RmdfError as e:
                    assert e.code == -8, str(e)
                frames([0.0], "mixed deals", check=False)
                sr.set_shard_costs(cost)
                sr.comm_verify_deal()
                frames([2.5] + [0.0] * S, "second verified deal")
                sr.comm_destroy()
                sr.close()
                results[rank] = True
            except BaseException as e:                          # noqa: BLE001
                import traceback
                errors.append((rank, traceback.format_exc()))

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(300)
        assert not errors and len(results) == n, errors[:2]
    c = counters()
    assert c["dev"] == 0 and c["host"] == 0 and c["streams"] == 0 and c["events"] == 0, c
    print("ok exchange with 2, 3 and 8 ranks against the RCCL double (deal check equal / unequal, frames in flight)", flush=True)


def main():
    assert os.environ.get("LD_PRELOAD", "").find("libfake_hip") >= 0, "run with LD_PRELOAD=tests/libfake_hip.so"
    if os.environ.get("FAKE_HIP_WORKLOAD_WATCHDOG_S"):          # a hang: say where every thread stands, then give up
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FAKE_HIP_WORKLOAD_WATCHDOG_S"]), exit=True)
    sr = rmdf.ShaderRenderer(0, xcheck=XCHECK)
    assert "fake_hip" in sr.device_info()[0], sr.device_info()
    sr.close()
    only = [a[5:] for a in sys.argv[1:] if a.startswith("only=")]
    for name, fn in (("whole", section_whole_frame_paths), ("tiles", section_tile_mode), ("shards", section_shards), ("env", section_env_pipeline),
                     ("random", section_random_call_sequences), ("calls", section_runtime_calls_per_frame),
                     ("args", section_argument_sweep), ("leaks", section_leaks_and_failed_allocations), ("exchange", section_exchange)):
        if not only or name in only:
            fn()
    c = counters()
    assert c["unknown"] == 0 or XCHECK, c
    print("done: %d launches through the double, %.1f MB moved by its copy calls, %d kernels registered, %d launches of kernels it has no stand-in for"
          % (c["launches"], c["copied"] / 1e6, c["kernels"], c["unknown"]), flush=True)


if __name__ == "__main__":
    main()
