"""CPU tier (round 6): the render kernel's SOURCE, executed, against the oracle -- every plane, bit for bit, without a GPU.

tests/kernel_on_host.cpp compiles csrc/rmdf_render.hip as it is -- render_body with its march loops, the workgroup pooling of the last
rays through LDS mailboxes, the AO straggler queue, the eight-lanes-per-ray Cornell tail with its DPP minima, the quad exchanges, the
LDS-staged stores, the strip-cost reduction, k_order_blocks, and the library's own launch code (launch_render: grid, variant choice,
strip order) -- for the CPU and runs it under a SIMT emulator (tests/koh_shim/hip/hip_runtime.h): one fiber per lane, __ballot / __shfl /
DPP / readfirstlane as true 64-lane collectives, __syncthreads as a workgroup barrier, __shared__ as per-workgroup storage.  Frames are
small (the emulator is ~10^4 x slower than the GPU), the comparison is total: steps, hit mask, escape-iteration counts, float colour bits,
RGBA8.  Also held against the COMMITTED golden frames and, directly, against what the reference's own fragment.shd produced on SwiftShader.

What this adds to tests/test_device_source_on_host.py (per-lane arithmetic): the wave-level program -- which ray marches where, what the
mailboxes and queues hand over, which lanes' results reach which pixel.  What it cannot say: anything about the code generator, about
the hardware's memory model (acquire / release pairs are plain accesses between fibers) or about time.  Those stay the GPU tier's."""
import ctypes as C
import glob
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLD, ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
THREADS = min(8, os.cpu_count() or 1)


class Frame(C.Structure):
    _fields_ = [("scene", C.c_int), ("w", C.c_int), ("h", C.c_int), ("x0", C.c_int), ("y0", C.c_int), ("x1", C.c_int), ("y1", C.c_int), ("max_steps", C.c_int),
                ("cam", C.c_float * 12), ("fov_xs", C.c_float), ("time", C.c_float), ("no_merge", C.c_int), ("no_prune", C.c_int),
                ("env_refl", C.c_void_p), ("env_cos1", C.c_void_p), ("env_cos8", C.c_void_p), ("w_refl", C.c_int), ("w_cos1", C.c_int), ("w_cos8", C.c_int),
                ("cornell_tri", C.c_void_p), ("cornell_tab", C.c_void_p), ("cornell_grid", C.c_void_p),
                ("rgba8", C.c_void_p), ("rgba8_mirror", C.c_void_p), ("rgba_f32", C.c_void_p), ("steps", C.c_void_p), ("iters", C.c_void_p),
                ("block_order", C.c_void_p), ("block_cost", C.c_void_p), ("n_shard_tiles", C.c_int), ("shard_tile", C.c_ubyte * 64),
                ("threads", C.c_int), ("seed_mode", C.c_int),
                ("band_count", C.c_void_p), ("band_flag", C.c_void_p), ("band_seq", C.c_uint), ("band_strip_rows", C.c_int)]


class Emulated:
    """the kernel source + the inputs rmdf_create / fill_params would give it (all host-built: librmdf_xcheck.so's host-only accessors)"""

    @staticmethod
    def build(variants):
        """compile the harness for every (defines, tag) that is missing or older than its sources -- all at once (each takes ~30 s)"""
        tdir = os.path.join(ROOT, "tests")
        src = os.path.join(tdir, "kernel_on_host.cpp")
        csrc = os.path.join(ROOT, "ray-marching-distance-fields_amd", "csrc")
        deps = [src] + [os.path.join(csrc, f) for f in ("rmdf_render.hip", "rmdf_env.hip", "rmdf_util.hip", "rmdf_device.hpp", "rmdf_internal.hpp")] + \
               [os.path.join(tdir, "koh_shim", "hip", f) for f in ("hip_runtime.h", "hip_fp16.h")]
        newest = max(os.path.getmtime(d) for d in deps)
        fma = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
        procs = []
        for defines, tag in variants:
            so = os.path.join(tdir, "libkernel_on_host%s.so" % tag)
            if os.path.exists(so) and os.path.getmtime(so) >= newest:
                continue
            cmd = [CLANG, "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-Wall", "-Wno-unused-function",
                   "-Wno-unknown-attributes", "-Wno-unused-variable", "-Wl,-Bsymbolic"] + fma + list(defines) + ["-x", "c++", "-I", os.path.join(tdir, "koh_shim"), "-I", csrc, src, "-o", so]
            # (-Bsymbolic: the emulated kernels carry the names of librmdf.so's kernel stubs -- that is how launches find them; in a host that LINKS
            #  librmdf.so the stubs sit in the global scope and would capture the emulator's own calls to its kernels)
            procs.append((tag, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for tag, pr in procs:
            out = pr.communicate()[0]
            assert pr.returncode == 0, "kernel_on_host%s: %s" % (tag, out[-3000:])

    def __init__(self, rmdf, env_oracle, defines=(), tag=""):
        Emulated.build([(defines, tag)])
        so = os.path.join(ROOT, "tests", "libkernel_on_host%s.so" % tag)
        self.K = K = C.CDLL(so)
        assert self.K.koh_frame_size() == C.sizeof(Frame)
        vp, i = C.c_void_p, C.c_int
        K.koh_order_blocks.argtypes = [vp, i, vp, i]
        if "-DKOH_RENDER_ONLY" not in defines:
            K.koh_cube_upload.argtypes = [vp, i, vp, i]
            K.koh_latlong_to_cube.argtypes = [vp, i, i, vp, vp, i]
            K.koh_resize_latlong.argtypes = [vp, i, i, i, i, vp, i]
            K.koh_resolve_box2.argtypes = [vp, i, i, vp, i]
        rmdf.build()
        X = rmdf.load_library(xcheck=True)
        self.X = X
        X.rmdf_debug_cornell_table.argtypes = [C.c_void_p] * 3
        X.rmdf_debug_cornell_bounds.argtypes = [C.c_void_p]
        X.rmdf_debug_cornell_masks.argtypes = [C.c_int, C.c_int, C.c_void_p]
        X.rmdf_get_cornell_vertices.argtypes = [C.c_void_p]
        X.rmdf_debug_camera.argtypes = [C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        self.tab = np.zeros(32 * 44 + 256, np.float32)
        assert X.rmdf_debug_cornell_table(self.tab.ctypes.data, None, None) == 0 and X.rmdf_debug_cornell_bounds(self.tab[32 * 44:].ctypes.data) == 0
        self.fine = np.zeros(64 ** 3, np.uint32)
        assert X.rmdf_debug_cornell_masks(64, 0, self.fine.ctypes.data) == 0
        self.tri = np.zeros(96 * 3, np.float32)
        assert X.rmdf_get_cornell_vertices(self.tri.ctypes.data) == 0
        self.cubes = [np.ascontiguousarray(c).view(np.uint16) for c in (env_oracle.reflection, env_oracle.cos_1, env_oracle.cos_8)]

    def frame(self, scene, w, h, t, ms, no_merge=0, no_prune=0, rect=None, seed_mode=0):
        f = Frame()
        f.scene, f.w, f.h, f.max_steps, f.time = scene, w, h, ms, t
        f.x0, f.y0, f.x1, f.y1 = rect or (0, 0, w, h)
        cam, fov = np.zeros(12, np.float32), C.c_float()
        assert self.X.rmdf_debug_camera(scene, C.c_float(t), cam.ctypes.data, C.byref(fov)) == 0
        for i in range(12):
            f.cam[i] = cam[i]
        f.fov_xs, f.no_merge, f.no_prune = fov.value, no_merge, no_prune
        f.env_refl, f.env_cos1, f.env_cos8 = [c.ctypes.data for c in self.cubes]
        f.w_refl, f.w_cos1, f.w_cos8 = [c.shape[1] - 2 for c in self.cubes]
        f.cornell_tri, f.cornell_tab = self.tri.ctypes.data, self.tab.ctypes.data
        f.cornell_grid = None if no_prune else self.fine.ctypes.data
        f.threads, f.seed_mode = THREADS, seed_mode
        return f

    def render(self, scene, w, h, t, ms, planes=True, mirror=False, **kw):
        f = self.frame(scene, w, h, t, ms, **kw)
        out = {"rgba8": np.full((h, w), 0xDEADBEEF, np.uint32)}
        f.rgba8 = out["rgba8"].ctypes.data
        if planes:
            out.update(rgba_f32=np.zeros((h, w, 4), np.float32), steps=np.zeros((h, w), np.uint16), iters=np.zeros((h, w), np.uint16))
            f.rgba_f32, f.steps, f.iters = out["rgba_f32"].ctypes.data, out["steps"].ctypes.data, out["iters"].ctypes.data
        if mirror:
            out["mirror"] = np.full((h, w), 0xDEADBEEF, np.uint32)
            f.rgba8_mirror = out["mirror"].ctypes.data
        assert self.K.koh_render(C.byref(f)) == 0
        return out

    def counts(self):
        c = (C.c_ulonglong * 6)()
        self.K.koh_take_counts(c)
        return dict(zip(("ballot", "shfl", "readfirstlane", "dpp", "polled_load", "syncthreads"), [int(x) for x in c]))


AB_BUILDS = [("-DRMDF_AB_XL_G=4", "_xl4", "four lanes per ray in the Cornell tail (sixteen rays per wave, two DPP steps)"),
             ("-DRMDF_AB_SHARED_BOUNDS", "_sharedb", "one pass of bound tests serves the normal's four sample points"),
             ("-DRMDF_AB_NO_XL", "_noxl", "the Cornell march without the lanes-per-ray tail"),
             ("-DRMDF_AB_MIRROR16", "_mirror16", "mirror stores as one wave's 16-byte stores"),
             ("-DRMDF_AB_MERGE_T=48", "_mt48", "workgroup pooling of the last rays at <= 48 live rays (tools/emulated_schedule.py predicts -2.8 % instructions)"),
             ("-DRMDF_AB_MERGE_T=56", "_mt56", "workgroup pooling of the last rays at <= 56 live rays (predicted -3.6 %)")]


@pytest.fixture(scope="module")
def emu(rmdf, env_oracle):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++")
    Emulated.build([((), ""), (("-DRMDF_XCHECK",), "_xcheck")] + [((d, "-DKOH_RENDER_ONLY"), t) for d, t, _ in AB_BUILDS])       # every build this module needs, side by side
    return Emulated(rmdf, env_oracle)


def assert_same_frame(got, ref, where=""):
    assert np.array_equal(got["steps"], ref["steps"]), "steps / hit mask differ " + where
    assert np.array_equal(got["iters"], ref["iters"]), "escape-iteration counts differ " + where
    assert np.array_equal(got["rgba_f32"].view(np.uint32), ref["rgba_f32"].view(np.uint32)), "float colour bits differ " + where
    assert np.array_equal(got["rgba8"], ref["rgba8"]), "RGBA8 differs " + where


@pytest.mark.parametrize("scene,ms", [(2, 256), (0, 128), (1, 128), (3, 128)])
@pytest.mark.parametrize("no_merge", [0, 1], ids=["product", "no_pooling"])
def test_small_frames_of_the_kernel_source_equal_the_oracle(emu, orc, env_oracle, scene, ms, no_merge):
    """the GPU tier's test_small_frames_vs_oracle, on the emulator: four camera times per scene, the product's variant (pooled march + AO
    queue for scenes 1-3, the eight-lane tail for the Cornell box) and RMDF_FLAG_NO_MERGE's; EVERY plane bit-equal (the GPU tier allows
    the colour 1e-4)"""
    emu.counts()
    for t in (0.0, 1.0, 2.5, 7.0):
        assert_same_frame(emu.render(scene, 64, 36, t, ms, no_merge=no_merge), orc.render(scene, 64, 36, t, ms, env_oracle), "scene %d t %.1f" % (scene, t))
    c = emu.counts()
    if scene != 0 and not no_merge:
        assert c["polled_load"] > 0 and c["readfirstlane"] > 0, c          # the pooled march's hand-over DID run (mailbox polls, host election)
    if scene == 0:
        assert c["dpp"] > 0, c                                             # the eight-lanes-per-ray tail DID run (DPP minima)


CASES = sorted(glob.glob(os.path.join(GOLD, "render_s*_64x36_*.npz")))


@pytest.mark.parametrize("fn", CASES, ids=[os.path.basename(c)[:-4] for c in CASES])
def test_kernel_source_vs_committed_golden_frames(emu, fn):
    """the committed golden frames (tests/golden/render_*.npz, written by the oracle in round 1 and pinned since) -- no oracle call here"""
    m = re.match(r"render_s(\d)_(\d+)x(\d+)_t(\d+)p(\d+)_m(\d+)\.npz", os.path.basename(fn))
    scene, w, h, t, ms = int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4) + "." + m.group(5)), int(m.group(6))
    g = np.load(fn)
    got = emu.render(scene, w, h, t, ms)
    assert np.array_equal(got["steps"], g["steps"]) and np.array_equal(got["iters"], g["iters"])
    assert np.array_equal(got["rgba8"], g["rgba8"])
    a, b = got["rgba_f32"].astype(np.float64), g["rgba_f32"].astype(np.float64)
    assert (np.abs(a - b) <= 1e-4 * np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)).all()


@pytest.mark.parametrize("scene,ms", [(2, 256), (0, 128)])
def test_headline_scenes_at_256x144(emu, orc, env_oracle, scene, ms):
    """the two BASELINE scenes at 256 x 144: 180 workgroups, long marches, mailboxes that fill, stragglers queued"""
    assert_same_frame(emu.render(scene, 256, 144, 0.0, ms), orc.render(scene, 256, 144, 0.0, ms, env_oracle), "scene %d" % scene)


def test_rectangles_tiles_ragged_sizes_and_outputs(emu, orc, env_oracle, rmdf):
    """what the C ABI's entry points ask of the kernel: the reference's tile rectangles of a frame 8 does not divide (helper pixels beyond the
    rectangle are computed, not written), a frame of odd size, the RGBA8-only and the mirror-store variants, Cornell without pruning"""
    w, h, ms = 100, 52, 64
    for scene in (2, 0):
        ref = orc.render(scene, w, h, 0.7, ms, env_oracle)
        for tile in (0, 9, 27, 63):
            x0, y0, x1, y1 = rmdf.tile_rect(tile, w, h)
            got = emu.render(scene, w, h, 0.7, ms, rect=(x0, y0, x1, y1))
            for k in ("steps", "iters", "rgba8"):
                assert np.array_equal(got[k][y0:y1, x0:x1], ref[k][y0:y1, x0:x1]), (scene, tile, k)
            outside = np.ones((h, w), bool)
            outside[y0:y1, x0:x1] = False
            assert (got["rgba8"][outside] == 0xDEADBEEF).all() and not got["steps"][outside].any(), "the kernel wrote outside its rectangle"
        odd = emu.render(scene, 37, 23, 0.7, ms)
        assert_same_frame(odd, orc.render(scene, 37, 23, 0.7, ms, env_oracle), "37 x 23")
        only8 = emu.render(scene, w, h, 0.7, ms, planes=False, mirror=True)
        assert np.array_equal(only8["rgba8"], ref["rgba8"]) and np.array_equal(only8["mirror"], ref["rgba8"])
    assert_same_frame(emu.render(0, w, h, 0.7, ms, no_prune=1), orc.render(0, w, h, 0.7, ms, env_oracle), "Cornell, NO_PRUNE")


def test_strip_order_and_cost_feedback(emu, orc, env_oracle):
    """the cost-ordered dispatch: a frame rendered in ANY strip order is the same frame, the kernel writes a cost per strip, and
    k_order_blocks (1024 lanes, LDS histogram, emulated too) turns the costs into a permutation, costliest bins first"""
    scene, w, h, ms = 2, 128, 72, 128
    ref = orc.render(scene, w, h, 0.0, ms, env_oracle)
    f = emu.frame(scene, w, h, 0.0, ms)
    n = emu.K.koh_grid_blocks(C.byref(f))
    assert n == 4 * 9
    cost = np.zeros(n, np.uint32)
    out = {k: np.zeros((h, w) + s, d) for k, s, d in (("rgba8", (), np.uint32), ("rgba_f32", (4,), np.float32), ("steps", (), np.uint16), ("iters", (), np.uint16))}
    f.rgba8, f.rgba_f32, f.steps, f.iters = [out[k].ctypes.data for k in ("rgba8", "rgba_f32", "steps", "iters")]
    f.block_cost = cost.ctypes.data
    assert emu.K.koh_render(C.byref(f)) == 0
    assert_same_frame(out, ref, "raster order")
    pix = (ref["iters"].astype(np.int64) + (ref["steps"] & 0x7FFF)).reshape(9, 8, 4, 32).max(axis=(1, 3)).reshape(-1)
    assert np.array_equal(cost, pix.astype(np.uint32)), "strip cost = the largest (iterations + steps) of its pixels"
    order = np.zeros(n, np.uint32)
    emu.K.koh_order_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    assert emu.K.koh_order_blocks(cost.ctypes.data, n, order.ctypes.data, 1) == 0
    assert sorted(order.tolist()) == list(range(n)), "not a permutation"

    def bin_of(c):
        if c < 8:
            return c
        e = int(c).bit_length() - 1
        return min(255, (e - 2) * 8 + ((int(c) >> (e - 3)) & 7))
    bins = [bin_of(int(cost[i])) for i in order]
    assert bins == sorted(bins, reverse=True), "k_order_blocks: not in descending cost bins"
    for k in out:
        out[k][...] = 0
    f.block_order = order.ctypes.data
    assert emu.K.koh_render(C.byref(f)) == 0
    assert_same_frame(out, ref, "cost order")


def test_shard_launch_renders_its_tiles_into_packed_slots(emu, orc, env_oracle, rmdf):
    """rmdf_render_shard_device's launch: grid.z = the rank's tiles, each into its packed slot of (w/8) x (h/8) pixels"""
    scene, w, h, ms = 2, 128, 72, 64
    ref = orc.render(scene, w, h, 0.0, ms, env_oracle)["rgba8"]
    tiles = [63, 0, 28, 35, 7]
    f = emu.frame(scene, w, h, 0.0, ms)
    f.n_shard_tiles = len(tiles)
    for i, t in enumerate(tiles):
        f.shard_tile[i] = t
    packed = np.zeros((len(tiles), h // 8, w // 8), np.uint32)
    f.rgba8 = packed.ctypes.data
    assert emu.K.koh_render(C.byref(f)) == 0
    for i, t in enumerate(tiles):
        x0, y0, x1, y1 = rmdf.tile_rect(t, w, h)
        assert np.array_equal(packed[i], ref[y0:y1, x0:x1]), t


SS_CASES = sorted(f for f in glob.glob(os.path.join(GOLD, "swiftshader_s[0-2]_9*.npz")) + glob.glob(os.path.join(GOLD, "swiftshader_s[0-2]_1[29]*.npz")) if not f.endswith("_gbuf.npz"))


@pytest.mark.parametrize("fn", SS_CASES, ids=[os.path.basename(c)[:-4] for c in SS_CASES])
def test_kernel_source_vs_the_reference_shader_on_swiftshader(emu, fn):
    """the GPU tier's test_hip_planes_vs_reference_shader_fixtures on the emulator: the kernel source's steps / hit / escape-iteration
    planes against what the reference's OWN fragment.shd produced on SwiftShader (tests/golden/swiftshader_*.npz; the oracle is not
    involved), with that test's bars"""
    m = re.match(r"swiftshader_s(\d)_(\d+)x(\d+)_t(\d+)p(\d+)_m(\d+)\.npz", os.path.basename(fn))
    scene, w, h, t, ms = int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4) + "." + m.group(5)), int(m.group(6))
    g = np.load(fn)
    got = emu.render(scene, w, h, t, ms)
    hit = (got["steps"] >> 15).astype(bool)
    ds = np.abs((got["steps"] & 0x7FFF).astype(int) - g["steps"].astype(int))
    assert np.array_equal(hit, g["hit"])
    assert (ds > 0).sum() <= 8 and ds.max() <= 1
    di = got["iters"].astype(int) - g["iters"].astype(int)
    if scene == 2:
        assert not di[~hit].any()
        assert (di[hit] != 0).mean() < 0.05
        assert abs(int(got["iters"].sum()) - int(g["iters"].sum())) < 1e-3 * int(g["iters"].sum())
    else:
        assert not got["iters"].any() and not g["iters"].any()


def test_frames_under_perturbed_hardware_seeds(emu, orc, env_oracle):
    """every emulated v_rsq / v_rcp / v_sqrt result moved one ulp at random (seed mode 3): the frame keeps its hit mask and all but a
    handful of its step / iteration counts (the inputs whose exact roots rest on the hardware's own seeds: test_device_source_on_host.py)"""
    for scene, ms in ((2, 256), (0, 128)):
        ref = orc.render(scene, 64, 36, 0.0, ms, env_oracle)
        got = emu.render(scene, 64, 36, 0.0, ms, seed_mode=3)
        assert np.array_equal(got["steps"] >> 15, ref["steps"] >> 15)
        assert (got["steps"] != ref["steps"]).mean() < 0.01 and (got["iters"] != ref["iters"]).mean() < 0.01
        d = np.abs(got["rgba8"].view(np.uint8).astype(int) - ref["rgba8"].view(np.uint8).astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.01


# ---- the env-map kernels (csrc/rmdf_env.hip) and the small ones (csrc/rmdf_util.hip), through the library's own launchers ---------------------

def _synthetic_latlong(w, h, seed):
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.empty((h, w, 3), np.float32)
    for k in range(3):
        lobes = sum(a * np.exp(-(((x - cx) / sx) ** 2 + ((y - cy) / sy) ** 2))
                    for a, cx, cy, sx, sy in zip(rng.uniform(2, 40, 5), rng.uniform(0, w, 5), rng.uniform(0, h, 5), rng.uniform(3, max(4, w / 8), 5), rng.uniform(3, max(4, h / 8), 5)))
        img[..., k] = 0.2 + 0.8 * (1.0 - y / h) ** (k + 1) + lobes + rng.uniform(0, 0.05, (h, w))
    return img.astype(np.float32)


def test_latlong_to_cube_and_rgb16f_upload_kernels_equal_the_oracle(emu, orc, env_latlongs):
    """k_latlong_to_cube on the host-built (u, v) table (rmdf_debug_cube_uv_table: what rmdf_set_env_latlong uploads) and k_cube_upload (RNE
    to RGB16F + the seamless border): every texel of the 512-wide reflection map's cube, of a 256-wide lobe map's, and of two odd sizes"""
    X, K = emu.X, emu.K
    X.rmdf_debug_cube_uv_table.argtypes = [C.c_int, C.c_void_p]
    for ll in (env_latlongs["refl"], env_latlongs["cos8"], _synthetic_latlong(100, 37, 1), orc.build_test_latlong()):
        ll = np.ascontiguousarray(ll, np.float32)
        h, w, _ = ll.shape
        cw = w // 3
        uv = np.zeros(6 * cw * cw * 2, np.float32)
        assert X.rmdf_debug_cube_uv_table(cw, uv.ctypes.data) == 0
        faces = np.zeros((6, cw, cw, 3), np.float32)
        assert K.koh_latlong_to_cube(ll.ctypes.data, w, h, uv.ctypes.data, faces.ctypes.data, THREADS) == 0
        ref = orc.latlong_to_cube(ll)
        assert np.array_equal(faces.view(np.uint32), ref.view(np.uint32)), (w, h)
        padded = np.zeros((6, cw + 2, cw + 2, 4), np.uint16)
        assert K.koh_cube_upload(faces.ctypes.data, cw, padded.ctypes.data, THREADS) == 0
        assert np.array_equal(padded, orc.cube_pad_f16(ref)), (w, h)


def test_resize_kernel_equals_the_oracle(emu, orc, env_latlongs):
    """k_resize_latlong (resizeHDRImage): the probe to the reference's 256, to other widths, from a map that is not 2:1"""
    for src, dstw in ((env_latlongs["refl"], 256), (env_latlongs["refl"], 128), (env_latlongs["refl"], 100), (_synthetic_latlong(100, 37, 2), 32)):
        src = np.ascontiguousarray(src, np.float32)
        ref = orc.resize_hdr(src, dstw)
        out = np.zeros_like(ref)
        assert emu.K.koh_resize_latlong(src.ctypes.data, src.shape[1], src.shape[0], dstw, ref.shape[0], out.ctypes.data, THREADS) == 0
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), dstw


@pytest.mark.parametrize("w,h", [(32, 16), (36, 10), (17, 6), (8, 3), (4, 2), (252, 5), (300, 4)])
def test_lobe_prefilter_kernels_equal_the_oracle(emu, orc, env_latlongs, w, h):
    """cosineConvolveHDREnvMap's kernels on the host-built lobe tables (rmdf_debug_lobe_tables): a reference power alone as the launcher picks
    the form (k_prefilter_chan -- producer waves, three summing waves, DPP row broadcasts -- for widths 4 divides up to 256, the one-wave
    k_prefilter with the table in LDS or read through global memory otherwise), side by side with others (split_ok = 0: the one-wave form),
    and the four powers in ONE launch (k_prefilter_fused4): bit-equal to the oracle's pinned form"""
    X, K = emu.X, emu.K
    X.rmdf_debug_lobe_tables.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    src = np.ascontiguousarray(orc.resize_hdr(env_latlongs["refl"], w) if (w, h) == (32, 16) else _synthetic_latlong(w, h, 7), np.float32)
    assert src.shape == (h, w, 3)
    lut = np.zeros(((w + 63) // 64) * w * 64, np.float32)
    tcs = np.zeros(2 * h, np.float32)
    assert X.rmdf_debug_lobe_tables(w, h, lut.ctypes.data, tcs.ctypes.data) == 0
    K.koh_prefilter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    refs = {p: orc.cosine_convolve(src, p, pow_mode=1) for p in (1.0, 8.0, 64.0, 512.0)}
    for p, ref in refs.items():
        for split_ok in (1, 0):
            out = np.zeros_like(src)
            assert K.koh_prefilter(src.ctypes.data, w, h, p, lut.ctypes.data, tcs.ctypes.data, out.ctypes.data, split_ok, THREADS) == 0
            assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (p, split_ok, float(np.abs(out - ref).max()))
    if w <= 256 and w % 4 == 0:
        outs = [np.zeros_like(src) for _ in range(4)]
        K.koh_prefilter_fused4.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p] + [C.c_void_p] * 4 + [C.c_int]
        assert K.koh_prefilter_fused4(src.ctypes.data, w, h, lut.ctypes.data, tcs.ctypes.data, *[o.ctypes.data for o in outs], THREADS) == 0
        for o, p in zip(outs, (1.0, 8.0, 64.0, 512.0)):
            assert np.array_equal(o.view(np.uint32), refs[p].view(np.uint32)), ("fused", p)
    c = emu.counts()
    if w <= 256 and w % 4 == 0 and w >= 16:
        assert c["dpp"] > 0, c                                    # the producer / summing-wave forms DID run (row broadcasts)


def test_resolve_assemble_and_fill_kernels(emu, orc, rmdf):
    """k_resolve_box2 (the super-sampling resolve) against the oracle's; k_assemble_shards (both forms: 16-byte and 4-byte) against the
    package's host assembly for 1, 2, 3 and 8 ranks; k_fill_u32"""
    K = emu.K
    rng = np.random.RandomState(5)
    for sw, sh in ((64, 36), (130, 6), (2, 2)):
        src = rng.randint(0, 2 ** 32, (sh, sw), dtype=np.uint64).astype(np.uint32)
        dst = np.zeros((sh // 2, sw // 2), np.uint32)
        assert K.koh_resolve_box2(src.ctypes.data, sw, sh, dst.ctypes.data, THREADS) == 0
        assert np.array_equal(dst, orc.resolve_box2(src))
    K.koh_fill_u32.argtypes = [C.c_void_p, C.c_uint, C.c_size_t, C.c_int]
    buf = np.zeros(100003, np.uint32)
    assert K.koh_fill_u32(buf.ctypes.data, 0xFF203040, buf.size - 3, THREADS) == 0
    assert (buf[:-3] == 0xFF203040).all() and not buf[-3:].any()
    K.koh_assemble_shards.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    for (w, h) in ((128, 72), (104, 40)):                         # tile widths 16 (the uint4 form) and 13 (the scalar form)
        for n in (1, 2, 3, 8):
            slots = rmdf.shard_slots(n)
            gathered = rng.randint(0, 2 ** 32, (n, slots, h // 8, w // 8), dtype=np.uint64).astype(np.uint32)
            where = np.zeros(64, np.uint16)
            for r in range(n):
                for s, t in enumerate(rmdf.shard_tiles(r, n)):
                    where[t] = (r << 8) | s
            frame = np.zeros((h, w), np.uint32)
            assert K.koh_assemble_shards(gathered.ctypes.data, frame.ctypes.data, w, h, n, where.ctypes.data, THREADS) == 0
            assert np.array_equal(frame, rmdf.assemble_shards_host(gathered, w, h, n)), (w, h, n)


# ---- the A/B builds of the render kernel (tools/abtest/*.so on the GPU): written in round 5, never run on hardware -- here: do they render the same frames? ----

@pytest.mark.parametrize("define,tag,what", AB_BUILDS, ids=[t[1:] for _, t, _ in AB_BUILDS])
def test_ab_builds_of_the_render_kernel_render_the_same_frames(emu, rmdf, orc, env_oracle, define, tag, what):
    """Each A/B build claims "bit-identical frames, maybe faster".  The second half needs a GPU; the first is checked here, before any
    GPU minute is spent on timing it: Cornell frames at four times + a 256 x 144 one, a Mandelbulb frame, tile rectangles with the mirror
    output (a variant whose frame differs from the oracle's is wrong, whatever its speed)."""
    if not os.path.exists(CLANG):
        pytest.skip("no clang++")
    e = Emulated(rmdf, env_oracle, defines=(define, "-DKOH_RENDER_ONLY"), tag=tag)
    e.counts()
    for t in (0.0, 1.0, 2.5, 7.0):
        assert_same_frame(e.render(0, 64, 36, t, 128), orc.render(0, 64, 36, t, 128, env_oracle), "%s: Cornell t %.1f" % (what, t))
    assert_same_frame(e.render(0, 256, 144, 0.0, 128), orc.render(0, 256, 144, 0.0, 128, env_oracle), what + ": Cornell 256 x 144")
    assert_same_frame(e.render(2, 64, 36, 0.0, 256), orc.render(2, 64, 36, 0.0, 256, env_oracle), what + ": Mandelbulb")
    c = e.counts()
    assert (c["dpp"] > 0) == (tag != "_noxl"), c
    if tag in ("_mt48", "_mt56"):
        # ... and BASELINE config 3 at FULL size: the sha256 of the RGBA8 frame == the committed oracle digest
        import hashlib
        import json
        want = json.load(open(os.path.join(GOLD, "full_size_digests.json")))["config3_mandelbulb8_1920x1080_m256"]["sha256"]["rgba8"]
        assert hashlib.sha256(e.render(2, 1920, 1080, 0.0, 256, planes=False)["rgba8"].tobytes()).hexdigest() == want, what
        for scene, ms in ((2, 256), (1, 128), (3, 128)):          # the scenes that pool: every time, and one larger frame
            for t in (1.0, 2.5, 7.0):
                assert_same_frame(e.render(scene, 64, 36, t, ms), orc.render(scene, 64, 36, t, ms, env_oracle), "%s: scene %d t %.1f" % (what, scene, t))
        assert_same_frame(e.render(2, 256, 144, 0.0, 256), orc.render(2, 256, 144, 0.0, 256, env_oracle), what + ": Mandelbulb 256 x 144")
    w, h = 100, 52
    for scene in (0, 2):
        ref = orc.render(scene, w, h, 0.7, 64, env_oracle)["rgba8"]
        whole = e.render(scene, w, h, 0.7, 64, planes=False, mirror=True)
        assert np.array_equal(whole["rgba8"], ref) and np.array_equal(whole["mirror"], ref), (what, scene)
        for tile in (0, 27, 63):
            x0, y0, x1, y1 = rmdf.tile_rect(tile, w, h)
            got = e.render(scene, w, h, 0.7, 64, planes=False, mirror=True, rect=(x0, y0, x1, y1))
            outside = np.ones((h, w), bool)
            outside[y0:y1, x0:x1] = False
            for k in ("rgba8", "mirror"):
                assert np.array_equal(got[k][y0:y1, x0:x1], ref[y0:y1, x0:x1]) and (got[k][outside] == 0xDEADBEEF).all(), (what, scene, tile, k)


# ---- what lives in librmdf_xcheck.so because no GPU has run it: does it at least compute the right thing? ------------------------------------

@pytest.fixture(scope="module")
def xemu(emu, rmdf, env_oracle):
    os.environ["RMDF_PREFILTER_RING"] = "1"          # read once per process by the cross-check build's launcher: the ring form where it applies
    return Emulated(rmdf, env_oracle, defines=("-DRMDF_XCHECK",), tag="_xcheck")


def test_one_launch_band_handover_of_the_cross_check_build(xemu, orc, env_oracle):
    """rmdf_config.reserved[3] = 2, 3 (round 5, never run on hardware): ONE launch whose OUT_MIRROR kernels count their workgroups in per band
    and flag a band when its last strip's stores are out; the strips ordered costliest first, then band by band from the outside in
    (k_order_blocks_bands).  Emulated: every band's flag carries the frame's sequence number afterwards, the counters are back at zero for the
    next frame, frame and mirror equal the oracle's -- in raster order and in the band-aware order, which is a permutation that does put the
    costly strips first and the outer bands before the inner ones."""
    scene, w, h, ms = 2, 128, 72, 128
    ref = orc.render(scene, w, h, 0.0, ms, env_oracle)["rgba8"]
    f = xemu.frame(scene, w, h, 0.0, ms)
    n, gx, rows = xemu.K.koh_grid_blocks(C.byref(f)), 4, 9
    assert n == gx * rows
    bsr, nb = 2, 5                                                        # bands of two strip rows: five bands, the last one a single row
    count, flag = np.zeros(16, np.uint32), np.zeros(16, np.uint32)
    frame, mirror, cost = np.zeros((h, w), np.uint32), np.zeros((h, w), np.uint32), np.zeros(n, np.uint32)
    f.rgba8, f.rgba8_mirror, f.block_cost = frame.ctypes.data, mirror.ctypes.data, cost.ctypes.data
    f.band_count, f.band_flag, f.band_seq, f.band_strip_rows = count.ctypes.data, flag.ctypes.data, 7, bsr
    assert xemu.K.koh_render(C.byref(f)) == 0
    assert np.array_equal(frame, ref) and np.array_equal(mirror, ref)
    assert (flag[:nb] == 7).all() and not flag[nb:].any() and not count.any(), (flag, count)
    order = np.zeros(n, np.uint32)
    xemu.K.koh_order_blocks_bands.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    assert xemu.K.koh_order_blocks_bands(cost.ctypes.data, n, order.ctypes.data, gx, bsr, nb, 1) == 0
    assert sorted(order.tolist()) == list(range(n))
    def bin_of(c):
        if c < 8:
            return c
        e = int(c).bit_length() - 1
        return min(255, (e - 2) * 8 + ((int(c) >> (e - 3)) & 7))
    maxbin = max(bin_of(int(c)) for c in cost)
    assert maxbin > 16, "a frame whose strips differ in cost"

    def key(i):                                                            # the kernel's sort key, restated: descending
        b = bin_of(int(cost[i]))
        if b >= maxbin - 8:
            return 128 + (b >> 1)                                          # within a factor two of the costliest strip: first, by cost
        band = (int(i) // gx) // bsr
        from_edge = 2 * band if band < nb - 1 - band else 2 * (nb - 1 - band) + 1
        return 127 - from_edge                                             # then band by band, outer bands first
    keys = [key(i) for i in order]
    assert keys == sorted(keys, reverse=True) and len(set(keys)) > 2, keys
    frame[...] = 0; mirror[...] = 0; flag[...] = 0
    f.block_order, f.band_seq = order.ctypes.data, 8
    assert xemu.K.koh_render(C.byref(f)) == 0
    assert np.array_equal(frame, ref) and np.array_equal(mirror, ref) and (flag[:nb] == 8).all() and not count.any()


@pytest.mark.parametrize("w,h", [(128, 4), (252, 5), (256, 3), (132, 9)])
def test_ring_form_of_the_prefilter_in_the_cross_check_build(xemu, orc, w, h):
    """k_prefilter_ring (RMDF_PREFILTER_RING=1, librmdf_xcheck.so; round 5, never run on hardware): producers and summing waves that never meet
    at a barrier -- factors through a ring of four chunk buffers with `filled` / `drained` counters in LDS.  Emulated (the counters are
    polled with the wave-uniform load, the waves really do run ahead of each other): every reference power bit-equal to the oracle's."""
    X, K = xemu.X, xemu.K
    X.rmdf_debug_lobe_tables.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    src = _synthetic_latlong(w, h, 9)
    lut, tcs = np.zeros(((w + 63) // 64) * w * 64, np.float32), np.zeros(2 * h, np.float32)
    assert X.rmdf_debug_lobe_tables(w, h, lut.ctypes.data, tcs.ctypes.data) == 0
    K.koh_prefilter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    xemu.counts()
    for p in (1.0, 8.0, 64.0, 512.0):
        out = np.zeros_like(src)
        assert K.koh_prefilter(src.ctypes.data, w, h, p, lut.ctypes.data, tcs.ctypes.data, out.ctypes.data, 1, THREADS) == 0
        ref = orc.cosine_convolve(src, p, pow_mode=1)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (p, float(np.abs(out - ref).max()))
    assert xemu.counts()["polled_load"] > 0, "the ring form did not run (no counter was polled)"


@pytest.mark.skipif(os.environ.get("RMDF_TEST_SLOW") != "1", reason="a build of its own (a minute): RMDF_TEST_SLOW=1 (clean when last run)")
def test_kernel_source_under_the_undefined_behaviour_sanitizer(rmdf, orc, env_oracle, tmp_path):
    """The same kernel source built with -fsanitize=undefined (minimal runtime: it prints `ubsan: <kind>` on stderr) and executed by the emulator: every scene,
    pooled and not, the env kernels through the launchers -- no shift, signed-overflow, misaligned-access, float-cast or bounds report, and the frames still
    equal the oracle's.  (GPU AddressSanitizer is not available on this pool; this is what a sanitizer can say about the kernels on the CPU.)"""
    import subprocess
    import sys
    rt = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.ubsan_minimal-x86_64.a"
    if not (os.path.exists(CLANG) and os.path.exists(rt)):
        pytest.skip("no clang++ / ubsan runtime")
    tdir = os.path.join(ROOT, "tests")
    so = os.path.join(tdir, "libkernel_on_host_ubsan.so")
    subprocess.check_call([CLANG, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-pthread", "-fsanitize=undefined",
                           "-fsanitize-minimal-runtime", "-fno-sanitize=vptr,function", "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-unused-variable",
                           "-x", "c++", "-I", os.path.join(tdir, "koh_shim"), "-I", os.path.join(ROOT, "ray-marching-distance-fields_amd", "csrc"),
                           os.path.join(tdir, "kernel_on_host.cpp"), "-x", "none", rt, "-o", so, "-ldl", "-lpthread"])
    code = '''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, rmdf_amd
from oracle import orc
import test_kernel_source_on_host as T
orc.build()
rd = lambda fn: orc.hdr_decode(open(fn, "rb").read())
cache = os.path.join(%r, "tests", "golden", "env_cache")
ll = {"refl": rd(rmdf_amd.DEFAULT_ENV_HDR), "cos1": rd(cache + "/uffizi_512_cache_pow_1.0.hdr"), "cos8": rd(cache + "/uffizi_512_cache_pow_8.0.hdr")}
env = orc.EnvSet(*(orc.cube_pad_f16(orc.latlong_to_cube(ll[k])) for k in ("refl", "cos1", "cos8")))
os.utime(%r, None)
e = T.Emulated(rmdf_amd, env, defines=("-DUBSAN",), tag="_ubsan")
for scene, ms in ((2, 256), (0, 128), (1, 128), (3, 128)):
    for nm in (0, 1):
        got, ref = e.render(scene, 96, 54, 2.5, ms, no_merge=nm), orc.render(scene, 96, 54, 2.5, ms, env)
        assert np.array_equal(got["rgba8"], ref["rgba8"]) and np.array_equal(got["iters"], ref["iters"]) and np.array_equal(got["steps"], ref["steps"]), (scene, nm)
print("frames ok")
''' % (ROOT, tdir, ROOT, so)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500)
    os.remove(so)
    assert r.returncode == 0 and "frames ok" in r.stdout and "ubsan:" not in r.stderr, (r.stdout[-1000:], r.stderr[-3000:])
