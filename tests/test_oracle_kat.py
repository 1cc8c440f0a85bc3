"""CPU tier: analytic known-answer tests for the oracle (tier C).

The reference has no tests to borrow, so each restated function is checked against an independent
closed form or a float64 numpy statement of the same mathematics."""
import math
import struct

import numpy as np
import pytest


def ulp_diff(a, b):
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


# ---- pinned transcendental functions -------------------------------------------------------

def test_log_exp_pow_accuracy(orc):
    rng = np.random.RandomState(7)
    xs = np.concatenate([np.exp(rng.uniform(-80, 80, 4000)), rng.uniform(0.5, 2.0, 4000), [1.0, 4.0, 4.0000005, 1e-40, 3e38]]).astype(np.float32)
    got = np.array([orc.logf(x) for x in xs], np.float32)
    ref = np.log(xs.astype(np.float64)).astype(np.float32)
    near_one = np.abs(ref) < 1e-3
    assert ulp_diff(got[~near_one], ref[~near_one]).max() <= 1
    assert np.abs(got[near_one].astype(np.float64) - np.log(xs[near_one].astype(np.float64))).max() < 2e-10
    es = rng.uniform(-80, 80, 6000).astype(np.float32)
    got = np.array([orc.expf(x) for x in es], np.float32)
    assert ulp_diff(got, np.exp(es.astype(np.float64)).astype(np.float32)).max() <= 1
    # the gamma curve pow(c, 1/2.2) over the colour range
    cs = np.exp(rng.uniform(-12, 5, 4000)).astype(np.float32)
    ig = np.float32(1.0) / np.float32(2.2)
    got = np.array([orc.powf(c, ig) for c in cs], np.float64)
    ref = cs.astype(np.float64) ** float(ig)
    assert (np.abs(got - ref) / ref).max() < 6e-7


def test_log_exp_special_values(orc):
    assert orc.logf(0.0) == -math.inf and math.isnan(orc.logf(-1.0)) and orc.logf(math.inf) == math.inf
    assert orc.logf(1.0) == 0.0
    assert orc.expf(0.0) == 1.0 and orc.expf(-100.0) == 0.0 and orc.expf(89.0) == math.inf
    assert orc.powf(0.0, 0.4545) == 0.0 and orc.powf(-1.0, 0.4545) == 0.0 and orc.powf(float("nan"), 0.4545) == 0.0
    assert orc.powf(math.inf, 0.4545) == math.inf


# ---- triplex algebra and the Mandelbulb DE ---------------------------------------------------

def triplex_pow_spherical(w, power):
    """fragment.shd:42-72 in float64: the general power the closed form must agree with."""
    x, y, z = (float(c) for c in w)
    r = math.sqrt(x * x + y * y + z * z)
    theta, phi = math.acos(z / r), math.atan2(y, x)
    zr = r ** power
    theta, phi = theta * power, phi * power
    return np.array([zr * math.sin(theta) * math.cos(phi), zr * math.sin(theta) * math.sin(phi), zr * math.cos(theta)])


def test_triplex_pow8_matches_spherical_form(orc):
    rng = np.random.RandomState(3)
    for _ in range(500):
        w = rng.uniform(-1.3, 1.3, 3).astype(np.float32)
        got = orc.triplex_pow8(w).astype(np.float64)
        ref = triplex_pow_spherical(w, 8.0)
        scale = max(1e-12, np.abs(ref).max())
        assert np.abs(got - ref).max() / scale < 2e-5


def test_triplex_pow8_nan_on_the_polar_axis(orc):
    """H6: x = y = 0 gives inversesqrt(0) = inf and 0*inf = NaN in the first two components."""
    out = orc.triplex_pow8([0.0, 0.0, 0.7])
    assert np.isnan(out[0]) and np.isnan(out[1]) and np.isfinite(out[2])


def test_mandelbulb_de_outside_bailout(orc):
    """|pos| > 4: the loop breaks at i = 0 with r = |pos|, dr = 1 -> DE = 0.5*log(r)*r."""
    for p in ([5.0, 0.0, 0.0], [3.0, 3.0, 3.0], [0.0, -4.5, 1.0]):
        r = np.float32(math.sqrt(sum(np.float32(c) * np.float32(c) for c in p)))
        ref = 0.5 * math.log(float(r)) * float(r)
        assert abs(orc.de(orc.SCENE_MB_POWER8, p) - ref) / ref < 3e-7


def test_mandelbulb_de_is_a_conservative_distance(orc):
    """Marching from outside along a ray must approach the surface monotonically without tunnelling far
    inside: DE(p) <= distance to the point where a fine march hits the surface (+ slack)."""
    o = np.array([0.0, 0.0, 1.14], np.float32)
    d = np.array([0.0, 0.0, -1.0], np.float32)
    t = 0.0
    for _ in range(300):
        dist = orc.de(orc.SCENE_MB_POWER8, o + np.float32(t) * d)
        assert dist > -1e-3
        if dist < 1e-4:
            break
        t += dist
    assert 0.0 < t < 1.14           # converged on a surface before the centre


# ---- geometry helpers ---------------------------------------------------------------------------

def test_ray_sphere_closed_form(orc):
    hit, tmin, tmax = orc.ray_sphere([0, 0, 3], [0, 0, -1], 1.15)
    assert hit and abs(tmin - 1.85) < 1e-6 and abs(tmax - 4.15) < 1e-6
    hit, _, _ = orc.ray_sphere([0, 2, 3], [0, 0, -1], 1.15)
    assert not hit
    hit, tmin, tmax = orc.ray_sphere([0, 0, 0], [1, 0, 0], 1.0)      # origin inside
    assert hit and abs(tmin + 1.0) < 1e-6 and abs(tmax - 1.0) < 1e-6


def test_fresnel_conductor(orc):
    def ref(c, eta, k):
        tmp = (eta * eta + k * k) * c * c
        rp = (tmp - 2 * eta * c + 1) / (tmp + 2 * eta * c + 1)
        tf = eta * eta + k * k
        rs = (tf - 2 * eta * c + c * c) / (tf + 2 * eta * c + c * c)
        return (rp + rs) / 2
    for c in (1.0, 0.7, 0.3, 0.05, 0.0):
        assert abs(orc.fresnel_conductor(c, 0.4, 0.8) - ref(c, np.float32(0.4), np.float32(0.8))) < 1e-6
    assert abs(orc.fresnel_conductor(0.0, 0.4, 0.8) - 1.0) < 1e-6    # grazing incidence reflects everything


def test_camera(orc):
    cam = orc.camera(orc.SCENE_MB_POWER8, 0.0)
    eye = cam[9:12]
    assert abs(np.linalg.norm(eye) - 2.414213562373095) < 1e-6
    assert np.allclose(eye, np.array([0, 1, 1]) / math.sqrt(2) * 2.414213562373095, atol=1e-6)
    x, y, z = cam[0:3], cam[3:6], cam[6:9]
    for a, b in ((x, y), (x, z), (y, z)):
        assert abs(float(np.dot(a, b))) < 1e-6
    assert np.allclose(np.cross(x, y), z, atol=1e-6)                  # right-handed
    cam = orc.camera(orc.SCENE_CORNELL, 0.0)
    assert np.allclose(cam[9:12], [0.0, 0.4, -2.0], atol=1e-7)
    assert abs(orc.fov_xs() - math.tan(math.radians(67.5) / 2)) < 1e-6


def test_cornell_geometry(orc):
    v = orc.cornell_vertices()
    assert v.shape == (96, 3)
    assert np.linalg.norm(v, axis=1).max() <= 0.99 + 1e-6             # scaled into the unit sphere
    # triangulation (q0,q1,q3),(q3,q1,q2): vertex 2 == vertex 3, vertex 1 == vertex 4 of each quad
    q = v.reshape(16, 6, 3)
    assert np.array_equal(q[:, 2], q[:, 3]) and np.array_equal(q[:, 1], q[:, 4])
    # DE against a float64 brute-force point-triangle distance
    tris = v.reshape(32, 3, 3).astype(np.float64)

    def seg(a, b, p):
        ab = b - a
        t = np.clip(np.dot(p - a, ab) / np.dot(ab, ab), 0, 1)
        return np.linalg.norm(p - (a + t * ab))

    def tri_dist(p, a, b, c):
        n = np.cross(b - a, c - a)
        n /= np.linalg.norm(n)
        q_ = p - np.dot(p - a, n) * n
        e0, e1, e2 = c - a, b - a, q_ - a
        d00, d01, d02, d11, d12 = e0 @ e0, e0 @ e1, e0 @ e2, e1 @ e1, e1 @ e2
        inv = 1 / (d00 * d11 - d01 * d01)
        u, w = (d11 * d02 - d01 * d12) * inv, (d00 * d12 - d01 * d02) * inv
        if u >= 0 and w >= 0 and u + w < 1:
            return abs(np.dot(p - a, n))
        return min(seg(a, b, p), seg(a, c, p), seg(b, c, p))

    rng = np.random.RandomState(11)
    for _ in range(60):
        p = rng.uniform(-0.6, 0.6, 3)
        ref = min(tri_dist(p, *t) for t in tris)
        assert abs(orc.de(orc.SCENE_CORNELL, p.astype(np.float32)) - ref) < 2e-6


# ---- half floats / cube maps ----------------------------------------------------------------------

def test_f16_conversions_exhaustive(orc):
    L = orc.lib()
    halves = np.arange(65536, dtype=np.uint16)
    ref = halves.view(np.float16).astype(np.float32)
    got = np.array([L.orc_f16_to_f32(int(h)) for h in halves], np.float32)
    assert np.array_equal(got.view(np.uint32)[~np.isnan(ref)], ref.view(np.uint32)[~np.isnan(ref)])
    rng = np.random.RandomState(5)
    xs = np.concatenate([rng.uniform(-70000, 70000, 20000), np.exp(rng.uniform(-30, 12, 20000)),
                         [0.0, 65504.0, 65519.9, 65520.0, 6.1e-5, 5.96e-8, 2.98e-8, 2.9802325e-8, 1e-10]]).astype(np.float32)
    with np.errstate(over="ignore"):
        ref16 = xs.astype(np.float16).view(np.uint16)
    got16 = np.array([L.orc_f32_to_f16(float(x)) for x in xs], np.uint16)
    assert np.array_equal(got16, ref16)                               # numpy casts RNE too


def test_cube_pixel_to_dir_round_trip(orc):
    """cubeMapPixelToDir (HDREnvMap.hs:76-87) is the inverse of GL face selection: sampling the
    cube at a texel-centre direction returns exactly that texel, NEAREST and LINEAR alike."""
    W = 9
    faces = np.zeros((6, W, W, 3), np.float32)
    for f in range(6):
        for y in range(W):
            for x in range(W):
                faces[f, y, x] = (f + 1, x + 1, y + 1)
    pad = orc.cube_pad_f16(faces)
    for f in range(6):
        for y in range(W):
            for x in range(W):
                d = orc.cube_pixel_to_dir(f, W, x, y)
                assert np.array_equal(orc.cube_sample(pad, d, 0), faces[f, y, x])
                assert np.allclose(orc.cube_sample(pad, d, 1), faces[f, y, x], atol=2e-3)


def test_cube_seamless_border(orc):
    """A cube that encodes a smooth function of direction stays smooth across every edge and corner under
    LINEAR filtering (GL_TEXTURE_CUBE_MAP_SEAMLESS, HDREnvMap.hs:126)."""
    W = 16
    faces = np.zeros((6, W, W, 3), np.float32)
    for f in range(6):
        for y in range(W):
            for x in range(W):
                faces[f, y, x] = orc.cube_pixel_to_dir(f, W, x, y) * 0.5 + 0.5
    pad = orc.cube_pad_f16(faces)
    rng = np.random.RandomState(2)
    worst = 0.0
    for _ in range(3000):
        d = rng.normal(size=3).astype(np.float32)
        d /= np.linalg.norm(d)
        got = orc.cube_sample(pad, d, 1)
        worst = max(worst, np.abs(got - (d * 0.5 + 0.5)).max())
    assert worst < 0.02            # bilinear reconstruction error of a 16^2 face, no seam spikes
    # constant cube -> constant everywhere, corners included
    pad1 = orc.cube_pad_f16(np.full((6, 4, 4, 3), 0.25, np.float32))
    assert (pad1[..., :3] == np.float16(0.25).view(np.uint16)).all()


def test_bilinear_quirks(orc):
    """pixelAtBilinear keeps the reference's `mod (w-1)` wrap and `min (h-1)` clamp (HDREnvMap.hs:100-104)."""
    img = np.zeros((4, 6, 3), np.float32)
    img[..., 0] = np.arange(6)[None, :]
    img[..., 1] = np.arange(4)[:, None]
    assert np.allclose(orc.pixel_at_bilinear(img, 0.0, 0.0), [0, 0, 0])
    assert np.allclose(orc.pixel_at_bilinear(img, 0.5, 0.5), [2.5, 1.5, 0])
    # u = 1: x = w-1 = 5, xp1 = 6 mod 5 = 1 (sic), weight on it is 0
    assert np.allclose(orc.pixel_at_bilinear(img, 1.0, 1.0), [5, 3, 0])
    # x = 4: xp1 = 5 mod 5 = 0 (sic, not 5): halfway between column 4 and column 0
    assert np.allclose(orc.pixel_at_bilinear(img, 0.9, 0.0), [2.0, 0, 0])


def test_prefilter_constant_environment(orc):
    """Constant radiance c: dst = c * sum(sin(theta_y) * cos^p) / count over the positive-cosine samples
    (HDREnvMap.hs:232-253) -- evaluate that closed form in float64."""
    w, h = 16, 8
    src = np.full((h, w, 3), 2.0, np.float32)
    for p in (1.0, 8.0):
        out = orc.cosine_convolve(src, p)
        th = np.arange(h) / (h - 1) * math.pi
        ph = np.arange(w) / (w - 1) * 2 * math.pi
        for dy in (0, 3, 7):
            for dx in (0, 5):
                ca = math.cos(th[dy]) * np.cos(th)[:, None] + math.sin(th[dy]) * np.sin(th)[:, None] * np.cos(np.abs(ph[dx] - ph))[None, :]
                m = ca > 0
                ref = 2.0 * (np.sin(th)[:, None] * np.where(m, ca, 0) ** p)[m].sum() / m.sum()
                assert abs(out[dy, dx, 0] - ref) < 2e-4 * max(1.0, abs(ref))


def test_rgbe_round_trip(orc):
    rng = np.random.RandomState(9)
    rgb = np.exp(rng.uniform(-8, 4, (500, 3))).astype(np.float32)
    back = orc.rgbe_roundtrip(rgb)
    mx = rgb.max(axis=1, keepdims=True)
    assert (np.abs(back - rgb) <= mx / 128.0).all()                    # 8-bit mantissa shared exponent
    assert np.array_equal(orc.rgbe_roundtrip(np.zeros((4, 3), np.float32)) < 1e-40, np.ones((4, 3), bool))


def test_hdr_rle_and_flat_decode_agree(orc):
    rng = np.random.RandomState(4)
    w, h = 16, 3
    rgbe = rng.randint(0, 255, (h, w, 4)).astype(np.uint8)
    rgbe[1, 3:12] = rgbe[1, 3]                                          # a run
    header = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)
    flat = header + rgbe.tobytes()
    rle = bytearray(header)
    for y in range(h):
        rle += bytes([2, 2, w >> 8, w & 255])
        for ch in range(4):
            row = rgbe[y, :, ch]
            x = 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 3:
                    rle += bytes([128 + run, int(row[x])])
                    x += run
                else:
                    rle += bytes([1, int(row[x])])
                    x += 1
    a, b = orc.hdr_decode(flat), orc.hdr_decode(bytes(rle))
    assert a.shape == (h, w, 3) and np.array_equal(a, b)
    with pytest.raises(ValueError):
        orc.hdr_decode(flat[:-5])
    with pytest.raises(ValueError):
        orc.hdr_decode(b"not an hdr file at all")


def test_resize_of_a_map_that_is_not_2_to_1_stays_inside_the_image(orc):
    """resizeHDRImage (HDREnvMap.hs:169-195) of 1024x510 -> 256: dsth = round(127.5) = 128 (half to even), so the last rows' taps ask
    pixelAtBilinear for source rows 510/511 -- past the image, where the reference's unsafePixelAt reads whatever follows.  Pin: the
    integer texel is clamped into the image (oracle and device kernel alike).  The map is embedded in a larger allocation filled with
    NaN behind it: no NaN may come out, and the last output rows equal the resize of a copy extended by replicated rows."""
    rng = np.random.RandomState(11)
    h, w = 510, 1024
    arena = np.full((h + 8, w, 3), np.nan, np.float32)
    arena[:h] = rng.uniform(0.0, 4.0, (h, w, 3)).astype(np.float32)
    out = orc.resize_hdr(arena[:h], 256)
    assert out.shape == (128, 256, 3) and np.isfinite(out).all()
    # rows whose taps stay inside are untouched by the pin: identical to resizing with the map's true geometry cut differently
    again = orc.resize_hdr(arena[:h].copy(), 256)
    assert np.array_equal(out.view(np.uint32), again.view(np.uint32))
    # u, v outside [0, 1] directly
    img = arena[:h]
    for (u, v) in ((1.2, 0.5), (-0.2, 0.5), (0.5, 1.01), (0.5, -0.3), (1.5, 1.5)):
        assert np.isfinite(orc.pixel_at_bilinear(img, u, v)).all()
    assert np.array_equal(orc.pixel_at_bilinear(img, 0.5, 1.01), orc.pixel_at_bilinear(img, 0.5, 1.0))


def test_resize_dimensions(orc):
    src = np.ones((256, 512, 3), np.float32)
    assert orc.resize_hdr(src, 256).shape == (128, 256, 3)
    assert np.allclose(orc.resize_hdr(src, 256), 1.0)
    assert orc.resize_hdr(np.ones((5, 9, 3), np.float32), 4).shape == (2, 4, 3)   # round(5/9*4 = 2.22) = 2


# ---- ConcurrentSegments / Fractal2D -------------------------------------------------------------

def test_make_n_segments(orc):
    assert orc.make_n_segments(4, 0, 0) == [] and orc.make_n_segments(0, 0, 10) == []
    assert orc.make_n_segments(1, 3, 9) == [(3, 9)]
    assert orc.make_n_segments(3, 0, 10) == [(0, 3), (3, 6), (6, 10)]     # remainder goes to the last segment
    assert orc.make_n_segments(8, 0, 5) == [(i, i + 1) for i in range(5)]  # nseg clamped to the range
    for n, lo, hi in ((7, 2, 101), (8, 0, 1080), (3, -5, 4)):
        segs = orc.make_n_segments(n, lo, hi)
        assert segs[0][0] == lo and segs[-1][1] == hi
        assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))


def test_julia_known_points(orc):
    """tick 0: c = (sin 0 * 0.7, cos 0 * 0.7) = (0, 0.7).  Corner pixels start far outside -> escape quickly;
    the picture is point-symmetric under z -> -z up to the half-pixel grid offset."""
    fb = orc.julia_animated(64, 64, 0, 0.0)
    assert fb.dtype == np.uint32 and (fb & 0xFFFF00FF == 0).all()          # green channel only, alpha 0
    g = (fb >> 8) & 0xFF
    assert g[0, 0] <= 2 * 255 // 40                                        # |z0|^2 = 2*1.45^2 < 16 escapes in ~2 steps
    assert g.max() == 255                                                  # interior reaches maxIter -> 255
    # independent float32 escape-time iteration for a few pixels
    for (px, py) in ((10, 20), (32, 32), (50, 7), (63, 63)):
        zr = np.float32(px) / np.float32(64) * np.float32(2.9) * np.float32(1.0) - np.float32(1.45) * np.float32(1.0)
        zi = np.float32(py) / np.float32(64) * np.float32(2.9) - np.float32(1.45)
        cr, ci = np.float32(0.0) * np.float32(0.7), np.float32(1.0) * np.float32(0.7)
        it = 0
        while it < 40 and zr * zr + zi * zi <= np.float32(16):
            nr, ni = (zr * zr - zi * zi) + cr, (zr * zi + zi * zr) + ci
            if nr == zr and ni == zi:
                it = 40
                break
            zr, zi, it = nr, ni, it + 1
        assert g[py, px] == int(np.float32(it) / np.float32(40) * np.float32(255))


def test_mandelbrot_known_points(orc):
    fb = orc.mandelbrot(96, 64, 0)
    g = (fb >> 8) & 0xFF
    # c = 0 (x = 0, y = 0) is interior: 1-cycle detection fires -> maxIter -> 255
    # x = (px/96)*2*1.5 + xshift, xshift = -2 - (3-2.5)/2 = -2.25 -> px = 72 ; y = 0 -> py = 32
    assert g[32, 72] == 255
    assert g[0, 0] < 40                                                    # far corner escapes at once
    sm = (orc.mandelbrot(96, 64, 1) >> 8) & 0xFF
    # smooth = iter - log2(ln|z|^2) with |z|^2 in (16, ~(16 + 2.5)^2]  ->  1.47 .. 2.55 iterations lower
    d = (g.astype(int) - sm.astype(int))[g < 255]
    assert (d >= 0).all() and (d <= int(2.6 * 255 / 40) + 1).all()


# ---- pinned trigonometry of the general-power variants (fragment.shd:42-72, 116-119) ------------------

def test_pinned_trig_accuracy(orc):
    rng = np.random.RandomState(21)
    xs = rng.uniform(-25, 25, 8000).astype(np.float32)                  # theta*power, phi*power <= pi*6.5
    s = np.array([orc.sinf(x) for x in xs], np.float32)
    c = np.array([orc.cosf(x) for x in xs], np.float32)
    assert np.abs(s - np.sin(xs.astype(np.float64))).max() < 1.5e-7
    assert np.abs(c - np.cos(xs.astype(np.float64))).max() < 1.5e-7
    xa = np.concatenate([rng.uniform(-1, 1, 8000), [-1.0, 1.0, 0.0, 0.5, -0.5, 1e-9]]).astype(np.float32)
    a = np.array([orc.acosf(x) for x in xa], np.float32)
    assert ulp_diff(a, np.arccos(xa.astype(np.float64)).astype(np.float32)).max() <= 1
    yy, xx = rng.normal(size=8000).astype(np.float32), rng.normal(size=8000).astype(np.float32)
    t = np.array([orc.atan2f(y, x) for y, x in zip(yy, xx)], np.float32)
    assert ulp_diff(t, np.arctan2(yy.astype(np.float64), xx.astype(np.float64)).astype(np.float32)).max() <= 1
    assert orc.atan2f(0.0, -1.0) == np.float32(np.pi) and orc.atan2f(1.0, 0.0) == np.float32(np.pi / 2)
    assert math.isnan(orc.acosf(1.5)) and math.isnan(orc.sinf(math.inf))


def test_general_power_schedule(orc):
    """fragment.shd:116-119: triangle wave 2 .. 6.5 .. 2 with period 18 s"""
    for t, p in ((0.0, 2.0), (1.0, 2.5), (9.0, 6.5), (10.0, 6.0), (18.0, 2.0), (20.0, 3.0), (27.0, 6.5)):
        assert abs(orc.general_power(t) - p) < 1e-6


def test_general_triplex_pow_matches_power8_closed_form(orc):
    """triplex_pow(w, 8) (spherical form, pinned trig) and triplex_pow8(w) (closed form) describe the same map."""
    rng = np.random.RandomState(8)
    for _ in range(300):
        w = rng.uniform(-1.2, 1.2, 3).astype(np.float32)
        a, b = orc.triplex_pow(w, 8.0).astype(np.float64), orc.triplex_pow8(w).astype(np.float64)
        assert np.abs(a - b).max() / max(1e-9, np.abs(b).max()) < 3e-5


def test_de_test_scene_known_points(orc):
    """FSDETestShader (fragment.shd:447-456): far from the tori and boxes the exponential smooth-min collapses to the
    nearest primitive; at the origin that is the sphere of radius 0.4."""
    assert abs(orc.de(orc.SCENE_DETEST, [0.0, 0.0, 0.0]) - (-0.4)) < 2e-3
    assert abs(orc.de(orc.SCENE_DETEST, [0.3, 0.3, 0.3]) - (math.sqrt(0.27) - 0.4)) < 2e-3
    # on the x axis at 0.9 the box arm along x (half-length 0.8, rounding 0.03 + 0.06) and the tori compete; the
    # value must be a lower bound of the true distance and within the smooth-min blending width log(n)/64
    d = orc.de(orc.SCENE_DETEST, [0.9, 0.0, 0.0])
    true = min(0.9 - 0.8 - 0.03, abs(0.9 - 0.85) - 0.1)      # box end cap vs. the two tori through (0.85, 0, 0)
    assert true - math.log(6) / 64 - 1e-3 <= d <= true + 1e-3


def test_resolve_box2(orc):
    rng = np.random.RandomState(3)
    src = rng.randint(0, 2**32, (6, 8), dtype=np.uint64).astype(np.uint32)
    got = orc.resolve_box2(src)
    b = src.view(np.uint8).reshape(6, 8, 4).astype(np.int64)
    ref = ((b[0::2, 0::2] + b[0::2, 1::2] + b[1::2, 0::2] + b[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(got.view(np.uint8).reshape(3, 4, 4), ref)
    assert np.array_equal(orc.resolve_box2(np.full((4, 4), 0xFF102030, np.uint32)), np.full((2, 2), 0xFF102030, np.uint32))
    with pytest.raises(ValueError):
        orc.resolve_box2(np.zeros((3, 4), np.uint32))
