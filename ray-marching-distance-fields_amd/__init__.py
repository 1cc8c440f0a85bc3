"""Host-side mirror of the reference's `ShaderRendering` module over librmdf.so.

The reference (Haskell) exposes, for the per-pixel ray-march path,

    withShaderRenderer :: FilePath -> FilePath -> (ShaderRenderer -> IO a) -> IO a
    drawShaderTile     :: ShaderRenderer -> FragmentShader -> Maybe Int -> Int -> Int -> Double -> IO ()
    data FragmentShader = FSDECornellBoxShader | FSDETestShader | FSMBPower8Shader | FSMBGeneralShader
    isTileIdxFirstTile, isTileIdxLastTile :: Int -> Bool          (ShaderRendering.hs:4-11)

and gets CPU-rendered pixels on screen through `FrameBuffer.fillFrameBuffer`
(FrameBuffer.hs:117-158).  This module keeps those names and argument meanings
(snake_case) on top of the C ABI in include/rmdf.h; everything is computed by the
gfx950 kernels in librmdf.so.  There is NO CPU fallback: if the library or a GPU is
missing, loading / `with_shader_renderer` raises `RmdfError`.

The directory name contains hyphens, so import it with
    importlib.import_module("ray-marching-distance-fields_amd")
or through the `rmdf_amd` shim at the repository root.
"""
import ctypes as C
import enum
import os
import subprocess
from contextlib import contextmanager

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librmdf.so")
XCHECK_LIB_PATH = os.path.join(_HERE, "librmdf_xcheck.so")     # cross-check build: product sources + alternative schedules
CSRC = os.path.join(_HERE, "csrc")
DATA_DIR = os.path.join(_HERE, "data")
# (RMDF_ENV_HDR: another light probe -- or, for the CPU tier's runs against the HIP test double, a private copy of the shipped one, so that
# the stand-in cache files of those runs never sit next to it)
DEFAULT_ENV_HDR = os.environ.get("RMDF_ENV_HDR") or os.path.join(DATA_DIR, "latlong_envmaps", "uffizi_512.hdr")

TILES_X, TILES_Y, N_TILES = 8, 8, 64          # ShaderRendering.hs:49-52
ENV_REFLECTION, ENV_COS_1, ENV_COS_8, ENV_COS_64, ENV_COS_512 = range(5)
FLAG_RASTER_ORDER, FLAG_NO_MERGE, FLAG_NO_PRUNE = 4, 16, 32          # rmdf.h RMDF_FLAG_*
FLAG_NESTED_LOOPS, FLAG_FLAT_MARCH = 1, 2                            # rmdf_xcheck.h: librmdf_xcheck.so only
FLAG_FORCE_WRITTEN = 64                                              # rmdf_xcheck.h: every folded Mandelbulb pass takes its written fall-back
COMM_ID_BYTES = 128

_ERRORS = {-1: "RMDF_E_INVALID", -2: "RMDF_E_NO_DEVICE", -3: "RMDF_E_HIP", -4: "RMDF_E_IO",
           -5: "RMDF_E_NO_ENV", -6: "RMDF_E_UNSUPPORTED", -7: "RMDF_E_NOMEM", -8: "RMDF_E_COMM"}

# every symbol include/rmdf.h declares
ABI_SYMBOLS = (
    "rmdf_create", "rmdf_destroy", "rmdf_last_error", "rmdf_load_env_hdr", "rmdf_set_env_latlong",
    "rmdf_set_env_cube", "rmdf_get_env_cube_padded", "rmdf_resize_latlong", "rmdf_prefilter_env",
    "rmdf_is_tile_idx_first_tile", "rmdf_is_tile_idx_last_tile", "rmdf_render_tile", "rmdf_render_tile_ex",
    "rmdf_render_rect_device", "rmdf_render_shard_device", "rmdf_assemble_shards_device", "rmdf_synchronize",
    "rmdf_device_info", "rmdf_resolve_box2_device", "rmdf_render_supersampled",
    "rmdf_selftest_exact_math", "rmdf_shard_tiles", "rmdf_probe_tile_costs", "rmdf_set_shard_costs",
    "rmdf_get_shard_tiles", "rmdf_save_png", "rmdf_register_host_buffer", "rmdf_unregister_host_buffer",
    "rmdf_selftest_pinned_math", "rmdf_selftest_shading_math", "rmdf_set_shard_root_handicap", "rmdf_create_ex", "rmdf_prefilter_env_powers",
    "rmdf_prefilter_env_device", "rmdf_comm_get_unique_id", "rmdf_comm_init", "rmdf_comm_destroy", "rmdf_comm_info",
    "rmdf_gather_shards_device", "rmdf_comm_verify_deal", "rmdf_render_frame_sharded_device", "rmdf_device_malloc", "rmdf_device_free",
    "rmdf_copy_to_host", "rmdf_probe_shader_clock", "rmdf_comm_selftest_loopback", "rmdf_get_cornell_vertices",
    "rmdf_get_shader_constants",
)
XCHECK_SYMBOLS = ("rmdf_debug_march_stats", "rmdf_debug_cornell_masks", "rmdf_debug_cornell_table", "rmdf_debug_cornell_bounds", "rmdf_debug_cube_uv_table",
                  "rmdf_debug_lobe_tables", "rmdf_debug_camera", "rmdf_debug_hdr_decode", "rmdf_debug_hdr_encode")      # include/rmdf_xcheck.h


class RmdfError(RuntimeError):
    """The `Left String` / `traceAndThrow` of the reference (ShaderRendering.hs:110)."""

    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (_ERRORS.get(code, "RMDF_E_?"), code, message))
        self.code = code


class FragmentShader(enum.IntEnum):
    """`data FragmentShader`, ShaderRendering.hs:46-47 (Enum order)."""
    FSDECornellBoxShader = 0
    FSDETestShader = 1
    FSMBPower8Shader = 2
    FSMBGeneralShader = 3


def is_tile_idx_first_tile(idx):   # ShaderRendering.hs:57-58
    return idx % N_TILES == 0


def is_tile_idx_last_tile(idx):    # ShaderRendering.hs:54-55
    return idx % N_TILES == N_TILES - 1


def tile_rect(tile_idx, w, h):
    """Pixel rectangle [x0,x1) x [y0,y1) of tile `tile_idx` (ShaderRendering.hs:183-193); ty counts from the bottom."""
    midx = tile_idx % N_TILES
    tx, ty = midx % TILES_X, midx // TILES_X
    return ((2 * tx * w + 7) // 16, (2 * ty * h + 7) // 16, (2 * (tx + 1) * w + 7) // 16, (2 * (ty + 1) * h + 7) // 16)


def _source_files():
    """The SOURCES the two libraries are built from: csrc/Makefile, csrc/**/*.{hip,hpp,cpp} outside the object directories, the two
    public headers.  Build products (csrc/build*/), caches and editor droppings never take part: `make clean`, or a pytest run
    started inside csrc/, must not turn a current library into a "stale" one (and send a GPU box looking for hipcc)."""
    files = []
    for r, ds, fs in os.walk(CSRC):
        ds[:] = sorted(d for d in ds if not d.startswith("build") and not d.startswith(".") and d != "__pycache__")
        files += [os.path.join(r, f) for f in fs if f == "Makefile" or f.endswith((".hip", ".hpp", ".cpp", ".h"))]
    files.sort()
    return files + [os.path.join(_HERE, "..", "include", h) for h in ("rmdf.h", "rmdf_xcheck.h")]


def _source_digest():
    """sha256 over the contents of everything the two libraries are built from (_source_files)."""
    import hashlib
    h = hashlib.sha256()
    for fn in _source_files():
        h.update(os.path.relpath(fn, _HERE).encode() + b"\0")
        with open(fn, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile librmdf.so and librmdf_xcheck.so for gfx950 with hipcc (cross-compiles without a GPU).  Serialised with a file
    lock: several ranks of one node may call this at the same time.  Whether the libraries are current is decided by a digest
    of the source CONTENTS kept next to them (librmdf.stamp), not by file times: a copied tree (the GPU box's snapshot) keeps
    the prebuilt libraries instead of rebuilding them because a copy happened to reorder mtimes."""
    import fcntl
    stamp = os.path.join(_HERE, "librmdf.stamp")

    def stale():
        if not (os.path.exists(LIB_PATH) and os.path.exists(XCHECK_LIB_PATH) and os.path.exists(stamp)):
            return True
        return open(stamp).read().strip() != _source_digest()
    if not (force or stale()):
        return LIB_PATH
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or stale():
                out = None if verbose else subprocess.DEVNULL
                # the digest is the authority, not make's file times: a tree restored with its mtimes (cp -p, rsync) would make a
                # plain `make` a no-op and the new digest would then bless stale libraries
                subprocess.check_call(["make", "-B", "-j4", "-C", CSRC], stdout=out)
                with open(stamp + ".tmp", "w") as f:
                    f.write(_source_digest() + "\n")
                os.replace(stamp + ".tmp", stamp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


_libs = {}


def load_library(xcheck=False):
    """dlopen librmdf.so (xcheck=True: librmdf_xcheck.so) and declare the prototypes.  Raises if the HIP library is missing."""
    if xcheck in _libs:
        return _libs[xcheck]
    # RMDF_LIB: measurement knob, an alternative build of the product sources (tools/abtest)
    path = XCHECK_LIB_PATH if xcheck else os.environ.get("RMDF_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise RmdfError(-2, "%s is not built (run __graft_entry__.build() or `make -C %s`); "
                            "there is no CPU fallback" % (os.path.basename(path), CSRC))
    L = C.CDLL(path)
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    L.rmdf_create.argtypes = [C.POINTER(vp), vp]
    L.rmdf_create_ex.argtypes = [C.POINTER(vp), vp, C.c_char_p, C.c_size_t]
    L.rmdf_destroy.argtypes = [vp]
    L.rmdf_destroy.restype = None
    L.rmdf_last_error.argtypes = [vp]
    L.rmdf_last_error.restype = C.c_char_p
    L.rmdf_load_env_hdr.argtypes = [vp, C.c_char_p]
    L.rmdf_set_env_latlong.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int]
    L.rmdf_set_env_cube.argtypes = [vp, C.c_int, vp, C.c_int]
    L.rmdf_get_env_cube_padded.argtypes = [vp, C.c_int, vp, ip]
    L.rmdf_resize_latlong.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, ip]
    L.rmdf_prefilter_env.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, vp]
    L.rmdf_prefilter_env_powers.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]
    L.rmdf_prefilter_env_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp]
    L.rmdf_comm_get_unique_id.argtypes = [vp]
    L.rmdf_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.rmdf_comm_destroy.argtypes = [vp]
    L.rmdf_comm_info.argtypes = [vp, ip, ip]
    L.rmdf_gather_shards_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.rmdf_comm_verify_deal.argtypes = [vp, vp]
    L.rmdf_get_cornell_vertices.argtypes = [vp]
    L.rmdf_get_shader_constants.argtypes = [vp, vp, C.c_int]
    L.rmdf_comm_selftest_loopback.argtypes = [vp, C.c_size_t, vp, C.POINTER(C.c_uint64)]
    L.rmdf_render_frame_sharded_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp, vp]
    L.rmdf_is_tile_idx_first_tile.argtypes = [C.c_int]
    L.rmdf_is_tile_idx_last_tile.argtypes = [C.c_int]
    L.rmdf_render_tile.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp]
    L.rmdf_render_tile_ex.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp, vp]
    L.rmdf_render_rect_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int] + [C.c_int] * 4 + [vp] * 5
    L.rmdf_render_shard_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, vp, vp]
    L.rmdf_shard_tiles.argtypes = [C.c_int, C.c_int, ip]
    L.rmdf_save_png.argtypes = [C.c_char_p, vp, C.c_int, C.c_int]
    L.rmdf_register_host_buffer.argtypes = [vp, vp, C.c_size_t]
    L.rmdf_unregister_host_buffer.argtypes = [vp, vp]
    L.rmdf_probe_tile_costs.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp]
    L.rmdf_set_shard_costs.argtypes = [vp, vp]
    L.rmdf_get_shard_tiles.argtypes = [vp, C.c_int, C.c_int, ip]
    L.rmdf_set_shard_root_handicap.argtypes = [vp, C.c_float]
    L.rmdf_assemble_shards_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.rmdf_synchronize.argtypes = [vp, vp]
    L.rmdf_probe_shader_clock.argtypes = [vp, C.c_double, C.POINTER(C.c_double)]
    L.rmdf_device_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.rmdf_device_free.argtypes = [vp, vp]
    L.rmdf_copy_to_host.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.rmdf_device_info.argtypes = [vp, C.c_char_p, C.c_int, ip]
    if xcheck:
        L.rmdf_debug_march_stats.argtypes = [vp, C.c_int, vp, C.c_int]
        L.rmdf_debug_cornell_masks.argtypes = [C.c_int, C.c_int, vp]
        L.rmdf_debug_cornell_table.argtypes = [vp, vp, vp]
        L.rmdf_debug_cube_uv_table.argtypes = [C.c_int, vp]
        L.rmdf_debug_lobe_tables.argtypes = [C.c_int, C.c_int, vp, vp]
        L.rmdf_debug_camera.argtypes = [C.c_int, C.c_float, vp, vp]
        L.rmdf_debug_hdr_decode.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_size_t]
        L.rmdf_debug_hdr_encode.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
        L.rmdf_debug_hdr_encode.restype = C.c_long
    L.rmdf_selftest_exact_math.argtypes = [vp, vp]
    L.rmdf_selftest_pinned_math.argtypes = [vp, vp]
    L.rmdf_selftest_shading_math.argtypes = [vp, vp]
    L.rmdf_resolve_box2_device.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    L.rmdf_render_supersampled.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp]
    _libs[xcheck] = L
    return L


class _Config(C.Structure):
    _fields_ = [("device", C.c_int), ("reserved", C.c_int * 7)]


def _ptr(a):
    return None if a is None else a.ctypes.data


class ShaderRenderer:
    """The `ShaderRenderer` record (ShaderRendering.hs:36-44): owns the device-side env cube maps,
    the Cornell geometry table and the accumulating frame."""

    def __init__(self, device=0, flags=0, xcheck=False, copy_threads=0, frame_bands=0, frame_mirror=0):
        """xcheck=True (implied by FLAG_FLAT_MARCH / FLAG_FORCE_WRITTEN): run on librmdf_xcheck.so, the cross-check build."""
        self.xcheck = bool(xcheck or (flags & (FLAG_FLAT_MARCH | FLAG_FORCE_WRITTEN)))
        self._lib = load_library(self.xcheck)
        self._ctx = C.c_void_p()
        cfg = _Config(device=device)
        cfg.reserved[0] = flags
        cfg.reserved[1] = copy_threads             # host threads of the frame copies (0 = library default)
        cfg.reserved[2] = frame_bands              # whole-frame host calls: row bands in flight (0 = library default, 1 = one launch)
        cfg.reserved[3] = frame_mirror             # ... how their rows reach the host: rmdf.h (0 .. 3)
        err = C.create_string_buffer(1024)
        rc = self._lib.rmdf_create_ex(C.byref(self._ctx), C.byref(cfg), err, 1024)
        if rc != 0:
            raise RmdfError(rc, err.value.decode())

    # -- lifetime --------------------------------------------------------------------------
    def close(self):
        if self._ctx:
            self._lib.rmdf_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RmdfError(rc, (self._lib.rmdf_last_error(self._ctx) or b"").decode())

    @property
    def handle(self):
        return self._ctx

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int()
        self._check(self._lib.rmdf_device_info(self._ctx, name, 256, C.byref(cus)))
        return name.value.decode(), cus.value

    # -- environment maps -------------------------------------------------------------------
    def load_env_hdr(self, path):
        """ShaderRendering.hs:67-91 for one latlong .hdr (cache files are built on the GPU if missing)."""
        self._check(self._lib.rmdf_load_env_hdr(self._ctx, os.fsencode(path)))

    def set_env_latlong(self, slot, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32)
        h, w, _ = rgb.shape
        self._check(self._lib.rmdf_set_env_latlong(self._ctx, slot, rgb.ctypes.data, w, h))

    def set_env_cube(self, slot, faces):
        faces = np.ascontiguousarray(faces, np.float32)
        assert faces.ndim == 4 and faces.shape[0] == 6 and faces.shape[1] == faces.shape[2] and faces.shape[3] == 3
        self._check(self._lib.rmdf_set_env_cube(self._ctx, slot, faces.ctypes.data, faces.shape[1]))

    def get_env_cube_padded(self, slot):
        w = C.c_int()
        self._check(self._lib.rmdf_get_env_cube_padded(self._ctx, slot, None, C.byref(w)))
        out = np.empty((6, w.value + 2, w.value + 2, 4), np.uint16)
        self._check(self._lib.rmdf_get_env_cube_padded(self._ctx, slot, out.ctypes.data, C.byref(w)))
        return out

    def resize_latlong(self, rgb, dstw):
        rgb = np.ascontiguousarray(rgb, np.float32)
        h, w, _ = rgb.shape
        dh = C.c_int()
        self._check(self._lib.rmdf_resize_latlong(self._ctx, rgb.ctypes.data, w, h, dstw, None, C.byref(dh)))
        out = np.empty((dh.value, dstw, 3), np.float32)
        self._check(self._lib.rmdf_resize_latlong(self._ctx, rgb.ctypes.data, w, h, dstw, out.ctypes.data, C.byref(dh)))
        return out

    def prefilter_env(self, rgb, power):
        rgb = np.ascontiguousarray(rgb, np.float32)
        h, w, _ = rgb.shape
        out = np.empty_like(rgb)
        self._check(self._lib.rmdf_prefilter_env(self._ctx, rgb.ctypes.data, w, h, float(power), out.ctypes.data))
        return out

    def prefilter_env_powers(self, rgb, powers):
        """Several lobe powers of one map, concurrently (the reference's mapConcurrently): (len(powers), h, w, 3)."""
        rgb = np.ascontiguousarray(rgb, np.float32)
        h, w, _ = rgb.shape
        pw = np.ascontiguousarray(powers, np.float32)
        out = np.empty((pw.size, h, w, 3), np.float32)
        self._check(self._lib.rmdf_prefilter_env_powers(self._ctx, rgb.ctypes.data, w, h, pw.ctypes.data, pw.size, out.ctypes.data))
        return out

    def prefilter_env_device(self, d_rgb, w, h, power, d_out, stream=0):
        self._check(self._lib.rmdf_prefilter_env_device(self._ctx, d_rgb, w, h, float(power), d_out, stream or None))

    # -- multi-GPU exchange (RCCL behind the C ABI) ----------------------------------------------
    def comm_init(self, unique_id, rank, nranks):
        """Collective: join the job's RCCL communicator.  unique_id = the 128 bytes rank 0 got from comm_get_unique_id()."""
        assert len(unique_id) == COMM_ID_BYTES
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._check(self._lib.rmdf_comm_init(self._ctx, buf, rank, nranks))

    def comm_destroy(self):
        self._check(self._lib.rmdf_comm_destroy(self._ctx))

    def comm_info(self):
        r, n = C.c_int(), C.c_int()
        self._check(self._lib.rmdf_comm_info(self._ctx, C.byref(r), C.byref(n)))
        return r.value, n.value

    def comm_selftest_loopback(self, nbytes, stream=0):
        """The exchange's own RCCL calls against this rank itself (grouped ncclRecv from self + ncclSend to self of nbytes on
        `stream`), compared word for word; returns the number of differing words (raises RMDF_E_COMM if any)."""
        bad = C.c_uint64(0)
        self._check(self._lib.rmdf_comm_selftest_loopback(self._ctx, int(nbytes), stream or None, C.byref(bad)))
        return bad.value

    def comm_verify_deal(self, stream=0):
        """COLLECTIVE: every rank of the communicator calls it after its last set_shard_costs / set_shard_root_handicap; raises on
        every rank if the ranks hold different deals (frames would be assembled from the wrong tiles; the sizes on the wire never depend on the deal)."""
        self._check(self._lib.rmdf_comm_verify_deal(self._ctx, stream or None))

    def gather_shards_device(self, w, h, d_shard, d_gathered=0, stream=0):
        self._check(self._lib.rmdf_gather_shards_device(self._ctx, w, h, d_shard, d_gathered or None, stream or None))

    def render_frame_sharded_device(self, shd_enum, w, h, time, max_steps, d_shard, d_gathered=0, d_frame=0, stream=0):
        """One frame of the multi-GPU path on `stream`: this rank's shard, the RCCL gather, rank 0's assembly."""
        self._check(self._lib.rmdf_render_frame_sharded_device(self._ctx, int(shd_enum), w, h, float(time), max_steps, d_shard,
                                                               d_gathered or None, d_frame or None, stream or None))

    def register_host_buffer(self, arr):
        """Declare a numpy frame buffer that is reused from frame to frame.  A hint only since round 5 (rmdf.h): the library no longer maps
        caller memory into the GPU's address space; every buffer takes the staged whole-frame path."""
        self._check(self._lib.rmdf_register_host_buffer(self._ctx, arr.ctypes.data, arr.nbytes))

    def unregister_host_buffer(self, arr):
        self._check(self._lib.rmdf_unregister_host_buffer(self._ctx, arr.ctypes.data))

    # -- rendering ----------------------------------------------------------------------------
    def draw_shader_tile(self, shd_enum, tile_idx, w, h, time, fb_vec, max_steps=128):
        """drawShaderTile sr shdEnum tileIdx w h time, writing into the `MVector Word32` that
        fillFrameBuffer hands out.  tile_idx None = `Nothing` (whole frame)."""
        assert fb_vec.dtype == np.uint32 and fb_vec.size == w * h and fb_vec.flags["C_CONTIGUOUS"]
        self._check(self._lib.rmdf_render_tile(self._ctx, int(shd_enum), -1 if tile_idx is None else int(tile_idx),
                                               w, h, float(time), max_steps, fb_vec.ctypes.data))

    def render(self, shd_enum, w, h, time, max_steps=128, tile_idx=None, want_f32=True):
        """Rich form for parity tests: dict(rgba8, rgba_f32, steps, iters), arrays (h, w[, 4]), row 0 = bottom."""
        rgba8 = np.empty((h, w), np.uint32)
        f32 = np.empty((h, w, 4), np.float32) if want_f32 else None
        steps = np.empty((h, w), np.uint16)
        iters = np.empty((h, w), np.uint16)
        self._check(self._lib.rmdf_render_tile_ex(self._ctx, int(shd_enum), -1 if tile_idx is None else int(tile_idx),
                                                  w, h, float(time), max_steps, rgba8.ctypes.data, _ptr(f32),
                                                  steps.ctypes.data, iters.ctypes.data))
        return {"rgba8": rgba8, "rgba_f32": f32, "steps": steps, "iters": iters}

    # -- device-resident forms (pointers are plain integers, e.g. torch.Tensor.data_ptr()) -------
    def render_rect_device(self, shd_enum, w, h, time, max_steps, rect, d_rgba8=0, d_rgba_f32=0, d_steps=0,
                           d_iters=0, stream=0):
        x0, y0, x1, y1 = rect
        self._check(self._lib.rmdf_render_rect_device(self._ctx, int(shd_enum), w, h, float(time), max_steps,
                                                      x0, y0, x1, y1, d_rgba8 or None, d_rgba_f32 or None,
                                                      d_steps or None, d_iters or None, stream or None))

    def render_shard_device(self, shd_enum, w, h, time, max_steps, rank, nranks, d_packed_rgba8, stream=0):
        self._check(self._lib.rmdf_render_shard_device(self._ctx, int(shd_enum), w, h, float(time), max_steps,
                                                       rank, nranks, d_packed_rgba8, stream or None))

    def probe_tile_costs(self, shd_enum, w, h, time, max_steps):
        """Per-tile work of this view measured on a 256 x ~144 probe frame (64 float32; identical on every rank)."""
        cost = np.zeros(64, np.float32)
        self._check(self._lib.rmdf_probe_tile_costs(self._ctx, int(shd_enum), w, h, float(time), max_steps, cost.ctypes.data))
        return cost

    def set_shard_costs(self, cost):
        """Deal the tiles to the ranks longest-processing-time-first on `cost` (None = static deal).  All ranks of a
        job must set the same costs."""
        if cost is not None:
            cost = np.ascontiguousarray(cost, np.float32)
            assert cost.size == 64
        self._check(self._lib.rmdf_set_shard_costs(self._ctx, _ptr(cost)))

    def set_shard_root_handicap(self, fraction):
        """Start rank 0 of the cost-aware deal at fraction x (total cost / nranks): it also receives and assembles."""
        self._check(self._lib.rmdf_set_shard_root_handicap(self._ctx, float(fraction)))

    def shard_tiles(self, rank, nranks):
        """The deal in effect on this renderer (static, or by the costs set with set_shard_costs)."""
        buf = (C.c_int * 64)()
        n = self._lib.rmdf_get_shard_tiles(self._ctx, rank, nranks, buf)
        self._check(n if n < 0 else 0)
        return list(buf[:n])

    def assemble_shards_device(self, w, h, nranks, d_gathered, d_frame_rgba8, stream=0):
        self._check(self._lib.rmdf_assemble_shards_device(self._ctx, w, h, nranks, d_gathered, d_frame_rgba8,
                                                          stream or None))

    def resolve_box2_device(self, d_src, sw, sh, d_dst, stream=0):
        self._check(self._lib.rmdf_resolve_box2_device(self._ctx, d_src, sw, sh, d_dst, stream or None))

    def render_supersampled(self, shd_enum, w, h, levels, time, max_steps=128):
        """Frame-buffer scale 2**levels (App.hs:105-106) + mip-chain resolve: returns (h, w) uint32."""
        out = np.empty((h, w), np.uint32)
        self._check(self._lib.rmdf_render_supersampled(self._ctx, int(shd_enum), w, h, levels, float(time), max_steps,
                                                       out.ctypes.data))
        return out

    def selftest_exact_math(self):
        """Mismatch counts (sqrt, rcp, log, rsqrt, Cornell division, Mandelbulb bailout test / in-loop sqrt / in-loop rsqrt) of
        the short exact sequences vs the compiler's, all 2^32 inputs; [8] folded vs written Mandelbulb estimates (2^28 points), [9] how many
        of those took the written fall-back (informational)."""
        out = np.zeros(10, np.uint64)
        self._check(self._lib.rmdf_selftest_exact_math(self._ctx, out.ctypes.data))
        return out

    def probe_shader_clock(self, spin_us=300.0):
        """MHz of the shader engines over the next `spin_us` microseconds (under whatever load other streams provide)."""
        mhz = C.c_double(0.0)
        self._check(self._lib.rmdf_probe_shader_clock(self._ctx, float(spin_us), C.byref(mhz)))
        return mhz.value

    def selftest_pinned_math(self):
        """Mismatch counts (exp, acos, atan, sin, cos, atan2, pow) of the straight-line device forms vs the branchy ones."""
        out = np.zeros(7, np.uint64)
        self._check(self._lib.rmdf_selftest_pinned_math(self._ctx, out.ctypes.data))
        return out

    def selftest_shading_math(self):
        """Mismatch counts (quotient, AO term, fresnel, cube-map lookup, generate_ray's quotients over every legal frame size) of the
        shading tail's short quotients vs the compiler's division."""
        out = np.zeros(5, np.uint64)
        self._check(self._lib.rmdf_selftest_shading_math(self._ctx, out.ctypes.data))
        return out

    def debug_march_stats(self, enable=True, read_waves=0):
        """Per-wave counters of the march kernels (rmdf_xcheck.h, xcheck renderers only); returns (n, 16) uint64 or None."""
        if not self.xcheck:
            raise RmdfError(-6, "debug_march_stats needs ShaderRenderer(..., xcheck=True)")
        out = np.zeros((read_waves, 16), np.uint64) if read_waves else None
        self._check(self._lib.rmdf_debug_march_stats(self._ctx, int(enable), _ptr(out), read_waves))
        return out

    def synchronize(self, stream=0):
        self._check(self._lib.rmdf_synchronize(self._ctx, stream or None))


_SHIPPED_PROBE = object()


@contextmanager
def with_shader_renderer(refl_map_fn=_SHIPPED_PROBE, device=0):
    """withShaderRenderer shdFn reflMapFn (ShaderRendering.hs:60-110) as a bracket.  There is no
    shader file: the kernels are compiled ahead of time for gfx950.  Default: the shipped light probe
    (DEFAULT_ENV_HDR as it stands when the call is made); refl_map_fn=None skips the env-map load (set
    the cube maps yourself)."""
    if refl_map_fn is _SHIPPED_PROBE:
        refl_map_fn = DEFAULT_ENV_HDR
    sr = ShaderRenderer(device)
    try:
        if refl_map_fn is not None:
            sr.load_env_hdr(refl_map_fn)
        yield sr
    finally:
        sr.close()


def comm_get_unique_id(xcheck=False):
    """Rank 0: the RCCL unique id (128 bytes) the other ranks need for ShaderRenderer.comm_init (ship it by any channel)."""
    L = load_library(xcheck)
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = L.rmdf_comm_get_unique_id(buf)
    if rc != 0:
        raise RmdfError(rc, (L.rmdf_last_error(None) or b"").decode())
    return buf.raw


def cornell_vertices():
    """The 96 x 3 triangle vertices the kernels' Cornell box is built from (mkCornellBoxVerticesTex, CornellBox.hs:21-46).  Host only."""
    out = np.empty((96, 3), np.float32)
    rc = load_library().rmdf_get_cornell_vertices(out.ctypes.data)
    if rc != 0:
        raise RmdfError(rc, "rmdf_get_cornell_vertices")
    return out


def shader_constants():
    """{name: value} of the fragment.shd constants the kernels use (rmdf_get_shader_constants).  Host only."""
    L = load_library()
    n = L.rmdf_get_shader_constants(None, None, 0)
    names = (C.c_char_p * n)()
    vals = (C.c_float * n)()
    L.rmdf_get_shader_constants(names, vals, n)
    return {names[i].decode(): float(vals[i]) for i in range(n)}


class FrameBuffer:
    """The slice of `FrameBuffer` (FrameBuffer.hs) the boundary needs: a w*h Word32 buffer handed to a
    filler (fillFrameBuffer, :117-158) and the PNG screenshot (saveFrameBufferToPNG, :215-228)."""

    def __init__(self, w, h):
        self.w, self.h = w, h
        self.vec = np.full(w * h, 0xFF000000, np.uint32)     # cleared to opaque black, :109-111

    def fill_frame_buffer(self, f):
        """fillFrameBuffer fb (\\w h vec -> ...)"""
        return f(self.w, self.h, self.vec)

    def to_image_rows_top_down(self):
        """Rows flipped and alpha forced to 0xFF exactly like saveFrameBufferToPNG (:222-227)."""
        rgba = self.vec.view(np.uint8).reshape(self.h, self.w, 4)[::-1].copy()
        rgba[..., 3] = 0xFF
        return rgba

    def save_png(self, fn):
        """saveFrameBufferToPNG fb fn, written by the library's own encoder (rmdf_save_png)."""
        L = load_library()
        rc = L.rmdf_save_png(os.fsencode(fn), self.vec.ctypes.data, self.w, self.h)
        if rc != 0:
            raise RmdfError(rc, (L.rmdf_last_error(None) or b"").decode())


# --- multi-GPU tile sharding: pure index arithmetic, testable without a GPU ------

def _ring_order():
    """The 64 tiles sorted by squared distance from the frame centre (nearest first, idx order among equals)."""
    d2 = lambda idx: (2 * (idx % 8) - 7) ** 2 + (2 * (idx // 8) - 7) ** 2
    return sorted(range(N_TILES), key=lambda idx: (d2(idx), idx))


def shard_tiles(rank, nranks):
    """Tiles rank `rank` of `nranks` renders, in slot order -- the same deal as rmdf_shard_tiles (rmdf.h): tiles sorted by
    distance from the frame centre, dealt boustrophedon (0..n-1, n-1..0, ...), so every rank gets near and far tiles (the
    scenes are centred: cost falls off with the distance from the centre)."""
    order = _ring_order()
    out = []
    for j, idx in enumerate(order):
        rnd, pos = divmod(j, nranks)
        if (nranks - 1 - pos if rnd & 1 else pos) == rank:
            out.append(idx)
    return out


def shard_tiles_by_cost(rank, nranks, cost, root_handicap=0.0):
    """The cost-aware deal of rmdf_set_shard_costs restated: tiles in descending cost order (idx order among equals),
    each to the least loaded rank that still has a free slot (lowest rank among equals); rank 0 starts at
    root_handicap x (total cost / nranks) (rmdf_set_shard_root_handicap)."""
    cost = [float(np.float32(c)) for c in cost]
    order = sorted(range(N_TILES), key=lambda i: (-cost[i], i))
    cap = shard_slots(nranks)
    load, used, out = [0.0] * nranks, [0] * nranks, []
    load[0] = float(np.float32(root_handicap)) * sum(cost) / nranks
    for idx in order:
        best = min((r for r in range(nranks) if used[r] < cap), key=lambda r: (load[r], r))
        load[best] += cost[idx]
        used[best] += 1
        if best == rank:
            out.append(idx)
    return out


def shard_tiles_abi(rank, nranks):
    """rmdf_shard_tiles through the library (host-only arithmetic, works without a GPU)."""
    buf = (C.c_int * 64)()
    n = load_library().rmdf_shard_tiles(rank, nranks, buf)
    if n < 0:
        raise RmdfError(n, "rmdf_shard_tiles")
    return list(buf[:n])


def shard_slots(nranks):
    """Slots every rank's packed shard buffer has (the gather needs equal sizes)."""
    return (N_TILES + nranks - 1) // nranks


def assemble_shards_host(gathered, w, h, nranks, tiles_of=None):
    """Reference (numpy) statement of rmdf_assemble_shards_device, for the gloo tests.
    gathered: (nranks, slots, h/8, w/8) uint32; tiles_of(rank, nranks) = the deal (default: the static one)."""
    tw, th = w // 8, h // 8
    frame = np.zeros((h, w), np.uint32)
    tiles_of = tiles_of or shard_tiles
    for r in range(nranks):
        for slot, idx in enumerate(tiles_of(r, nranks)):
            tx, ty = idx % 8, idx // 8
            frame[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = gathered[r, slot]
    return frame
