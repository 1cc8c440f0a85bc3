#!/usr/bin/env python3
"""The library's device work under the electric-fence allocator (tests/test_gpu_guard.py runs this in a child process with
RMDF_GUARD_ALLOC=end and =start: every device allocation of librmdf_xcheck.so then ends -- or starts -- at an unmapped page, so a kernel
that touches one element outside a buffer it was given dies of a GPU memory fault instead of reading a neighbour).  Every kernel of the
library runs here at sizes that do NOT round to anything convenient: ragged strips, odd widths, face sizes and map sizes whose byte counts
end in the middle of a 16-byte group.  Prints `guard workload ok` at the end."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rmdf_amd                                             # noqa: E402


def main():
    assert os.environ.get("RMDF_GUARD_ALLOC") in ("end", "start")
    # On the CPU tier the same workload runs against the HIP test double (tests/fake_hip.cpp, LD_PRELOAD): its virtual-memory calls are
    # mmap / mprotect, so the fence is real there too -- it then checks the HOST's buffer sizes against what the kernels' stand-ins write
    # and read, and the allocator's own bookkeeping (tests/test_host_logic.py: test_guarded_allocations_against_the_hip_double).
    double = "libfake_hip" in os.environ.get("LD_PRELOAD", "")
    # ... and with FAKE_HIP_EMULATE=1 the double RUNS the kernels' source (tests/kernel_on_host.cpp): then the fence checks what it was built to
    # check -- the KERNELS' own global accesses, every one of them, against the exact size of every device buffer -- on the CPU.  The emulator
    # is slow: the 256 x 128 prefilters are left to the cache files and to smaller maps of the same shapes.
    emulated = double and os.environ.get("FAKE_HIP_EMULATE", "0") not in ("", "0")
    import ctypes as C
    import shutil
    import tempfile
    tmpdir = tempfile.mkdtemp() if double else None
    hdr = rmdf_amd.DEFAULT_ENV_HDR
    if double:                                               # (the double's prefilter is a stand-in: its cache files must not land in the tree)
        hdr = os.path.join(tmpdir, os.path.basename(rmdf_amd.DEFAULT_ENV_HDR) if emulated else "probe.hdr")
        shutil.copy(rmdf_amd.DEFAULT_ENV_HDR, hdr)
        if emulated:
            cache = os.path.join(ROOT, "tests", "golden", "env_cache")
            for f in os.listdir(cache):
                shutil.copy(os.path.join(cache, f), tmpdir)
    rng = np.random.RandomState(7)
    sr = rmdf_amd.ShaderRenderer(0, xcheck=True)
    L = rmdf_amd.load_library(True)
    held = []

    def dmalloc(nbytes):
        """device memory from the library's own allocator (rmdf_device_malloc -> dev_malloc: fenced like everything else)"""
        p = C.c_void_p()
        assert L.rmdf_device_malloc(sr.handle, nbytes, C.byref(p)) == 0
        held.append(p)
        return p.value
    # the whole env pipeline: decode, resize (k_resize_latlong), the fused four-power prefilter, RGBE caches, k_latlong_to_cube, k_cube_upload
    sr.load_env_hdr(hdr)
    # prefilter kernels: the channel-split form, the one-wave form (odd width, wide map reading its table through L2), other powers
    for (w, h) in (((36, 10), (100, 3), (252, 5), (8, 3), (260, 4), (1100, 2)) if emulated else ((256, 128), (100, 37), (252, 5), (8, 3), (260, 20), (1100, 6))):
        src = rng.uniform(0.0, 4.0, (h, w, 3)).astype(np.float32)
        for pw in ((1.0, 8.0, 64.0, 512.0), (8.0,), (2.5, 64.0), (1.0, 8.0, 64.0, 512.0, 3.0)):
            sr.prefilter_env_powers(src, pw)
    # lat/long maps whose sizes are not 2:1, faces of odd sizes
    for (w, h) in ((96, 48), (101, 37), (33, 70)):
        ll = rng.uniform(0.0, 2.0, (h, w, 3)).astype(np.float32)
        sr.resize_latlong(ll, 29)
        for slot in (rmdf_amd.ENV_COS_64, rmdf_amd.ENV_COS_512):
            sr.set_env_latlong(slot, ll)
    for fw in (1, 2, 7, 85):
        sr.set_env_cube(rmdf_amd.ENV_COS_64, rng.uniform(0.0, 2.0, (6, fw, fw, 3)).astype(np.float32))
        sr.get_env_cube_padded(rmdf_amd.ENV_COS_64)
    # every scene, every output variant: planes, RGBA8 only (row bands, every way the rows reach the host), tiles, a registered buffer
    for scene, ms in ((0, 40), (1, 40), (2, 64), (3, 24)):
        for (w, h) in (((133, 87), (33, 17), (1, 1), (250, 9)) if emulated else ((333, 187), (64, 64), (33, 17), (1, 1), (250, 9))):
            sr.render(scene, w, h, 0.4, max_steps=ms)
            fb = np.zeros(w * h, np.uint32)
            sr.draw_shader_tile(scene, None, w, h, 0.4, fb, max_steps=ms)
            for idx in range(0, 64, 13 if emulated else 5):
                sr.draw_shader_tile(scene, idx, w, h, 0.4, fb, max_steps=ms)
    for bands, mode in ((3, 0), (4, 1), (5, 2), (16, 3)):
        r2 = rmdf_amd.ShaderRenderer(0, xcheck=True, frame_bands=bands, frame_mirror=mode)
        for slot in (rmdf_amd.ENV_REFLECTION, rmdf_amd.ENV_COS_1, rmdf_amd.ENV_COS_8):
            r2.set_env_cube(slot, rng.uniform(0.0, 2.0, (6, 9, 9, 3)).astype(np.float32))
        for (w, h) in (((323, 181),) if emulated else ((1283, 721), (1920, 1080))):
            fb = np.zeros(w * h, np.uint32)
            for _ in range(2):
                r2.draw_shader_tile(2, None, w, h, 0.1, fb, max_steps=32)
        r2.close()
    big = np.zeros(640 * 360 + 4096, np.uint32)
    sr.register_host_buffer(big)
    sr.draw_shader_tile(2, None, 640, 360, 0.0, big[1021:1021 + 640 * 360], max_steps=32)       # not 16-byte aligned: the narrow mirror stores
    sr.draw_shader_tile(2, None, 640, 360, 0.0, big[1024:1024 + 640 * 360], max_steps=32)
    sr.unregister_host_buffer(big)
    # the alternative schedule of the cross-check build (its G-buffer and work counter)
    alt = rmdf_amd.ShaderRenderer(0, flags=rmdf_amd.FLAG_FLAT_MARCH, xcheck=True)
    alt.load_env_hdr(hdr)
    for (w, h) in ((333, 187), (33, 17)):
        alt.render(2, w, h, 0.4, max_steps=64)
    alt.close()
    # shards, the assembly, the resolve, the supersampled frame, the cost probe, frames on caller streams
    w, h = 640, 360
    for n in (1, 3, 8):
        slots = rmdf_amd.shard_slots(n)
        tile = (h // 8) * (w // 8) * 4
        gath = dmalloc(n * slots * tile)
        frame = dmalloc(w * h * 4)
        for r in range(n):
            sr.render_shard_device(2, w, h, 0.0, 48, r, n, gath + r * slots * tile)
        sr.assemble_shards_device(w, h, n, gath, frame)
        sr.synchronize()
    sr.render_supersampled(2, 160, 90, 2, 0.0, max_steps=32)
    sr.probe_tile_costs(2, 1920, 1080, 0.0, 64)                  # (renders its own 256 x 144 probe frame)
    rect = dmalloc(187 * 333 * 4)
    sr.render_rect_device(0, 333, 187, 0.0, 32, (5, 3, 301, 180), d_rgba8=rect)
    sr.synchronize()
    # the self-tests' kernels (their own cube map, the Cornell table)
    if not emulated or os.environ.get("RMDF_TEST_SLOW") == "1":  # (100 s on the emulator: exhaustive over the shading functions' inputs)
        assert sum(sr.selftest_shading_math()) == 0
    for p in held:
        assert L.rmdf_device_free(sr.handle, p) == 0
    sr.close()
    if tmpdir:
        shutil.rmtree(tmpdir, ignore_errors=True)
    print("guard workload ok")


if __name__ == "__main__":
    main()
