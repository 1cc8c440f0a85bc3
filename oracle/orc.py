"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE, not product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

SCENE_CORNELL, SCENE_DETEST, SCENE_MB_POWER8, SCENE_MB_GENERAL = 0, 1, 2, 3


def build(force=False):
    """gcc the oracle (serialised with a file lock: test workers may race)."""
    import fcntl
    src = [os.path.join(_HERE, f) for f in ("rmdf_oracle.c", "rmdf_oracle.h", "Makefile")]

    def stale():
        return not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src)
    if not (force or stale()):
        return _LIB
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or stale():
                subprocess.check_call(["make"] + (["-B"] if force else []) + ["-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB


class Cube(C.Structure):
    _fields_ = [("W", C.c_int), ("padded", C.c_void_p)]


class Frame(C.Structure):
    _fields_ = [("scene", C.c_int), ("w", C.c_int), ("h", C.c_int), ("time", C.c_float),
                ("max_steps", C.c_int), ("env_reflection", Cube), ("env_cos_1", Cube), ("env_cos_8", Cube)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("de_evals", "triplex_iters", "march_steps", "hit_pixels",
                                          "sphere_pixels", "pixels", "tri_inside")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        fp, ip, u8p = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
        L.orc_fov_xs.restype = C.c_float
        for n in ("orc_logf", "orc_expf", "orc_sinf", "orc_cosf", "orc_acosf", "orc_general_power"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        for n in ("orc_powf", "orc_atan2f"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float, C.c_float]
        L.orc_triplex_pow.argtypes = [fp, C.c_float, fp]
        L.orc_de.restype = C.c_float
        L.orc_de.argtypes = [C.c_int, C.c_float, fp]
        L.orc_fresnel_conductor.restype = C.c_float
        L.orc_fresnel_conductor.argtypes = [C.c_float] * 3
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_ray_sphere.argtypes = [fp, fp, C.c_float, fp, fp]
        L.orc_render.argtypes = [C.POINTER(Frame)] + [C.c_int] * 4 + [C.c_void_p] * 4 + [C.POINTER(Counters), C.c_int]
        L.orc_render_ex.argtypes = [C.POINTER(Frame)] + [C.c_int] * 4 + [C.c_void_p] * 5 + [C.POINTER(Counters), C.c_int]
        L.orc_shade_gbuffer.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_hdr_decode.argtypes = [u8p, C.c_long, ip, ip, C.c_void_p]
        L.orc_hdr_encode.restype = C.c_long
        L.orc_hdr_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_rgbe_encode.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        L.orc_rgbe_decode.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        L.orc_latlong_to_cube.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_cube_pad_f16.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_resize_hdr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_cosine_convolve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int]
        L.orc_pixel_at_bilinear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, fp]
        L.orc_cube_pixel_to_dir.argtypes = [C.c_int] * 4 + [fp]
        L.orc_cube_sample.argtypes = [C.POINTER(Cube), fp, C.c_int, fp]
        L.orc_julia_animated.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int]
        L.orc_mandelbrot.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_make_n_segments.argtypes = [C.c_int, C.c_int, C.c_int, ip]
        L.orc_camera.argtypes = [C.c_int, C.c_float, fp]
        L.orc_triplex_pow8.argtypes = [fp, fp]
        L.orc_cornell_vertices.argtypes = [fp]
        L.orc_build_test_latlong.argtypes = [C.c_void_p]
        L.orc_resolve_box2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


# ---- scalar probes --------------------------------------------------------------

def logf(x): return lib().orc_logf(float(x))
def expf(x): return lib().orc_expf(float(x))
def powf(x, y): return lib().orc_powf(float(x), float(y))
def sinf(x): return lib().orc_sinf(float(x))
def cosf(x): return lib().orc_cosf(float(x))
def acosf(x): return lib().orc_acosf(float(x))
def atan2f(y, x): return lib().orc_atan2f(float(y), float(x))
def general_power(t): return lib().orc_general_power(float(t))


def triplex_pow(w, power):
    out = (C.c_float * 3)()
    lib().orc_triplex_pow(_f3(w), float(power), out)
    return np.array(out[:], dtype=np.float32)
def de(scene, pos, time=0.0): return lib().orc_de(scene, float(time), _f3(pos))
def fresnel_conductor(cosi, eta, k): return lib().orc_fresnel_conductor(cosi, eta, k)
def fov_xs(): return lib().orc_fov_xs()


def triplex_pow8(w):
    out = (C.c_float * 3)()
    lib().orc_triplex_pow8(_f3(w), out)
    return np.array(out[:], dtype=np.float32)


def camera(scene, time):
    out = (C.c_float * 12)()
    lib().orc_camera(scene, float(time), out)
    return np.array(out[:], dtype=np.float32)


def ray_sphere(o, d, R):
    tmin, tmax = C.c_float(), C.c_float()
    hit = lib().orc_ray_sphere(_f3(o), _f3(d), float(R), C.byref(tmin), C.byref(tmax))
    return bool(hit), tmin.value, tmax.value


def cornell_vertices():
    out = (C.c_float * 288)()
    lib().orc_cornell_vertices(out)
    return np.array(out[:], dtype=np.float32).reshape(96, 3)


def shader_constants():
    """{name: value} of the fragment.shd constants the oracle uses (orc_shader_constants)."""
    L = lib()
    L.orc_shader_constants.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    n = L.orc_shader_constants(None, None, 0)
    names = (C.c_char_p * n)()
    vals = (C.c_float * n)()
    L.orc_shader_constants(names, vals, n)
    return {names[i].decode(): float(vals[i]) for i in range(n)}


def make_n_segments(nseg, low, high):
    out = (C.c_int * (2 * max(nseg, 1)))()
    n = lib().orc_make_n_segments(nseg, low, high, out)
    return [(out[2 * i], out[2 * i + 1]) for i in range(n)]


def num_processors():
    return lib().orc_num_processors()


# ---- env-map data prep ------------------------------------------------------------

def hdr_decode(data: bytes):
    w, h = C.c_int(), C.c_int()
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    if lib().orc_hdr_decode(buf, len(data), C.byref(w), C.byref(h), None) != 0:
        raise ValueError("not a Radiance RGBE file")
    out = np.empty((h.value, w.value, 3), np.float32)
    rc = lib().orc_hdr_decode(buf, len(data), C.byref(w), C.byref(h), out.ctypes.data)
    if rc != 0:
        raise ValueError("truncated Radiance file (rc=%d)" % rc)
    return out


def hdr_encode(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    buf = np.empty(64 + 4 * w * h, np.uint8)
    n = lib().orc_hdr_encode(rgb.ctypes.data, w, h, buf.ctypes.data)
    return buf[:n].tobytes()


def rgbe_roundtrip(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    n = rgb.size // 3
    e = np.empty((n, 4), np.uint8)
    lib().orc_rgbe_encode(rgb.ctypes.data, n, e.ctypes.data)
    out = np.empty_like(rgb)
    lib().orc_rgbe_decode(e.ctypes.data, n, out.ctypes.data)
    return out


def latlong_to_cube(latlong, nthreads=0):
    latlong = np.ascontiguousarray(latlong, np.float32)
    h, w, _ = latlong.shape
    cw = w // 3
    faces = np.empty((6, cw, cw, 3), np.float32)
    lib().orc_latlong_to_cube(latlong.ctypes.data, w, h, faces.ctypes.data, nthreads)
    return faces


def cube_pad_f16(faces):
    faces = np.ascontiguousarray(faces, np.float32)
    cw = faces.shape[1]
    out = np.empty((6, cw + 2, cw + 2, 4), np.uint16)
    lib().orc_cube_pad_f16(faces.ctypes.data, cw, out.ctypes.data)
    return out


def build_test_latlong():
    out = np.empty((256, 512, 3), np.float32)
    lib().orc_build_test_latlong(out.ctypes.data)
    return out


def resize_hdr(src, dstw):
    src = np.ascontiguousarray(src, np.float32)
    sh, sw, _ = src.shape
    dh = lib().orc_resize_hdr(src.ctypes.data, sw, sh, dstw, None)
    out = np.empty((dh, dstw, 3), np.float32)
    lib().orc_resize_hdr(src.ctypes.data, sw, sh, dstw, out.ctypes.data)
    return out


def cosine_convolve(src, power, nthreads=0, pow_mode=1):
    """pow_mode 1 (default): cos^p by the pinned binary64 squaring chain for the reference's powers 1/8/64/512;
    pow_mode 0: libm powf, the reference's literal call (not correctly rounded, libm-build dependent)."""
    src = np.ascontiguousarray(src, np.float32)
    h, w, _ = src.shape
    out = np.empty_like(src)
    lib().orc_cosine_convolve(src.ctypes.data, w, h, float(power), out.ctypes.data, nthreads, int(pow_mode))
    return out


def env_pipeline(hdr_bytes, powers=(1.0, 8.0), nthreads=0, pow_mode=1):
    """The env part of withShaderRenderer (ShaderRendering.hs:65-91) in the oracle, cache miss: decode the reflection map,
    resizeHDRImage to 256, cosineConvolveHDREnvMap per power, Radiance RGBE write + reload.  Returns
    (latlongs, cache_files): latlongs = {"refl": ..., "cos1": ..., "cos8": ..., ...}, cache_files = {power: file bytes}."""
    refl = hdr_decode(hdr_bytes)
    small = resize_hdr(refl, 256)
    lat, files = {"refl": refl}, {}
    for p in powers:
        files[float(p)] = hdr_encode(cosine_convolve(small, p, nthreads=nthreads, pow_mode=pow_mode))
        lat["cos%d" % int(p)] = hdr_decode(files[float(p)])
    return lat, files


def pixel_at_bilinear(img, u, v):
    img = np.ascontiguousarray(img, np.float32)
    h, w, _ = img.shape
    out = (C.c_float * 3)()
    lib().orc_pixel_at_bilinear(img.ctypes.data, w, h, float(u), float(v), out)
    return np.array(out[:], np.float32)


def cube_pixel_to_dir(face, w, x, y):
    out = (C.c_float * 3)()
    lib().orc_cube_pixel_to_dir(face, w, x, y, out)
    return np.array(out[:], np.float32)


def cube_sample(padded, direction, linear):
    padded = np.ascontiguousarray(padded, np.uint16)
    c = Cube(padded.shape[1] - 2, padded.ctypes.data)
    out = (C.c_float * 3)()
    lib().orc_cube_sample(C.byref(c), _f3(direction), int(linear), out)
    return np.array(out[:], np.float32)


# ---- 2-D fractals -----------------------------------------------------------------

def julia_animated(w, h, smooth, tick, nthreads=0):
    fb = np.empty((h, w), np.uint32)
    lib().orc_julia_animated(w, h, fb.ctypes.data, int(smooth), float(tick), nthreads)
    return fb


def mandelbrot(w, h, smooth):
    fb = np.empty((h, w), np.uint32)
    lib().orc_mandelbrot(w, h, fb.ctypes.data, int(smooth))
    return fb


def resolve_box2(src):
    src = np.ascontiguousarray(src, np.uint32)
    sh, sw = src.shape
    dst = np.empty((sh // 2, sw // 2), np.uint32)
    if lib().orc_resolve_box2(src.ctypes.data, sw, sh, dst.ctypes.data) != 0:
        raise ValueError("resolve needs even sizes")
    return dst


# ---- the renderer -------------------------------------------------------------------

class EnvSet:
    """The three cube maps the shader samples, as padded RGB16F arrays."""

    def __init__(self, reflection, cos_1, cos_8):
        self.reflection = np.ascontiguousarray(reflection, np.uint16)
        self.cos_1 = np.ascontiguousarray(cos_1, np.uint16)
        self.cos_8 = np.ascontiguousarray(cos_8, np.uint16)

    @staticmethod
    def from_latlongs(refl_ll, cos1_ll, cos8_ll):
        return EnvSet(*(cube_pad_f16(latlong_to_cube(x)) for x in (refl_ll, cos1_ll, cos8_ll)))


def render(scene, w, h, time, max_steps, env: EnvSet, rect=None, nthreads=0, want_f32=True):
    """Returns dict(rgba_f32, rgba8, steps, iters, iters_march, counters); arrays are (h, w[,4]), row 0 = bottom."""
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, w, h)
    f = Frame()
    f.scene, f.w, f.h, f.time, f.max_steps = scene, w, h, float(time), max_steps
    for name in ("reflection", "cos_1", "cos_8"):
        arr = getattr(env, name)
        setattr(f, "env_" + name, Cube(arr.shape[1] - 2, arr.ctypes.data))
    rgba = np.zeros((h, w, 4), np.float32) if want_f32 else None
    rgba8 = np.zeros((h, w), np.uint32)
    steps = np.zeros((h, w), np.uint16)
    iters = np.zeros((h, w), np.uint16)
    iters_march = np.zeros((h, w), np.uint16)
    ctr = Counters()
    rc = lib().orc_render_ex(C.byref(f), x0, y0, x1, y1, rgba.ctypes.data if want_f32 else None,
                             rgba8.ctypes.data, steps.ctypes.data, iters.ctypes.data, iters_march.ctypes.data, C.byref(ctr), nthreads)
    if rc != 0:
        raise RuntimeError("orc_render failed rc=%d" % rc)
    return {"rgba_f32": rgba, "rgba8": rgba8, "steps": steps, "iters": iters, "iters_march": iters_march, "counters": ctr.as_dict()}


def shade_gbuffer(scene, w, h, time, max_steps, env: EnvSet, nao, hit):
    """Shading alone: colour (h, w, 4) from per-pixel hit flags and (normal, ao) -- see orc_shade_gbuffer."""
    f = Frame()
    f.scene, f.w, f.h, f.time, f.max_steps = scene, w, h, float(time), max_steps
    for name in ("reflection", "cos_1", "cos_8"):
        arr = getattr(env, name)
        setattr(f, "env_" + name, Cube(arr.shape[1] - 2, arr.ctypes.data))
    nao = np.ascontiguousarray(nao, np.float32)
    hit = np.ascontiguousarray(hit, np.uint8)
    out = np.zeros((h, w, 4), np.float32)
    rc = lib().orc_shade_gbuffer(C.byref(f), nao.ctypes.data, hit.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_shade_gbuffer failed rc=%d" % rc)
    return out
