"""CPU tier: a SECOND, independent reading of the reference's environment-map preparation, written in numpy straight from the
Haskell text -- `HDREnvMap.hs:76-113` (cubeMapPixelToDir, pixelAtBilinear), `:118-163` (latLongHDREnvMapToCubeMap), `:169-195`
(resizeHDRImage), `:217-254` (cosineConvolveHDREnvMap) and `CoordTransf.hs:35-70` -- and compared with the C oracle.

Why: GHC is not in the image, so these paths have no reference run (parity unpinned), and the SwiftShader fixtures are
rendered from ORACLE-built cube maps, so they cannot see an error in this part of the oracle either.  Two restatements made
independently from the same source do not share typos: the index quirks (`xp1 = (x+1) mod (w-1)`, `yp1 = min (h-1) (y+1)`, texel
centres scaled by `w-1`), the face orientation table, the phi wrap-arounds, the tap pattern of the resize and the summation order and
divisor of the convolution are all exercised here.  Float32 arithmetic throughout, one rounding per operation; libm functions come from
numpy here and from glibc there, so values may differ in the last bits: tolerances are stated per test (bilinear gathers amplify an
ulp of (u, v) by the texel contrast).
"""
import numpy as np

f32 = np.float32
PI = f32(np.pi)


def _pixel_at_bilinear(img, u, v):
    """HDREnvMap.hs:91-113"""
    h, w, _ = img.shape
    upx = f32(u * f32(w - 1))
    upy = f32(v * f32(h - 1))
    x, y = int(np.floor(upx)), int(np.floor(upy))
    xp1 = (x + 1) % (w - 1)
    yp1 = min(h - 1, y + 1)
    ur, vr = f32(upx - f32(x)), f32(upy - f32(y))
    uo, vo = f32(f32(1) - ur), f32(f32(1) - vr)
    t = lambda xc, yc: img[yc, xc].astype(f32)
    top = (t(x, y) * uo + t(xp1, y) * ur).astype(f32)
    bot = (t(x, yp1) * uo + t(xp1, yp1) * ur).astype(f32)
    return (top * vo + bot * vr).astype(f32)


def _cube_pixel_to_dir(face, w, x, y):
    """HDREnvMap.hs:76-87 (faces in GL order +X -X +Y -Y +Z -Z) followed by Linear.normalize"""
    vw = f32(f32(f32(f32(x) + f32(0.5)) / f32(w)) * f32(2) - f32(1))
    vh = f32(f32(f32(f32(y) + f32(0.5)) / f32(w)) * f32(2) - f32(1))
    d = [(f32(1), -vh, -vw), (f32(-1), -vh, vw), (vw, f32(1), vh), (vw, f32(-1), -vh), (vw, -vh, f32(1)), (-vw, -vh, f32(-1))][face]
    d = np.array(d, f32)
    n2 = f32(f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])      # quadrance: sum of the products in order
    return (d / np.sqrt(n2, dtype=f32)).astype(f32)             # (Linear's near-unit short cut cannot trigger: |d| >= 1, == 1 only at a face centre no texel hits)


def _dir_to_uv(d):
    """worldToLocal (CoordTransf.hs:46-50), cartesianToSpherical (:35-44), sphericalToEnvironmentUV (:60-70)"""
    lx, ly, lz = d[0], -d[2], d[1]                               # dot with x = (1,0,0), y = (0,0,-1), n = (0,1,0)
    theta = np.arccos(np.clip(lz, f32(-1), f32(1)), dtype=f32)
    phi = np.arctan2(ly, lx, dtype=f32)
    two_pi = f32(f32(2) * PI)
    if phi < 0:
        phi = f32(phi + two_pi)
    if phi == two_pi:
        phi = f32(0)
    phi1 = f32(phi + f32(PI / f32(2)))
    phi2 = f32(phi1 - two_pi) if phi1 > two_pi else phi1
    phi3 = f32(two_pi - phi2)
    return f32(phi3 / f32(PI * f32(2))), f32(theta / PI)


def test_latlong_to_cube_second_reading(orc):
    rng = np.random.RandomState(11)
    for (w, h) in ((24, 12), (33, 14)):                           # w div 3 = 8 and 11 (w not divisible by 3: the faces ignore the rest)
        lat = rng.uniform(0.0, 2.0, (h, w, 3)).astype(f32)
        cw = w // 3
        want = np.empty((6, cw, cw, 3), f32)
        for face in range(6):
            for y in range(cw):
                for x in range(cw):
                    u, v = _dir_to_uv(_cube_pixel_to_dir(face, cw, x, y))
                    want[face, y, x] = _pixel_at_bilinear(lat, u, v)
        got = orc.latlong_to_cube(lat)
        assert got.shape == want.shape
        d = np.abs(got.astype(np.float64) - want)
        # an ulp of (u, v) times (w - 1) texels times the contrast of random texels (~2): a few 1e-5 at most; most texels are equal
        assert d.max() < 2e-4, d.max()
        assert (d == 0).mean() > 0.5, (d == 0).mean()
        assert np.median(d) < 1e-6


def test_resize_hdr_second_reading(orc):
    rng = np.random.RandomState(12)
    for (sw, sh, dw) in ((40, 20, 16), (37, 19, 8), (16, 8, 16)):
        src = rng.uniform(0.0, 3.0, (sh, sw, 3)).astype(f32)
        dh = int(np.rint(f32(f32(f32(sh) / f32(sw)) * f32(dw))))    # Haskell round = half-to-even
        scale = f32(f32(sw) / f32(dw))
        taps = int(np.ceil(scale))
        ntaps = f32(taps * taps)
        step = f32(scale / f32(taps))
        want = np.empty((dh, dw, 3), f32)
        for dy in range(dh):
            for dx in range(dw):
                sx1, sy1 = f32(f32(dx) * scale), f32(f32(dy) * scale)
                acc = np.zeros(3, f32)
                for ty in range(taps):
                    for tx in range(taps):
                        srcx = f32(sx1 + f32(f32(tx) * step))
                        srcy = f32(sy1 + f32(f32(ty) * step))
                        acc = (acc + _pixel_at_bilinear(src, f32(srcx / f32(sw - 1)), f32(srcy / f32(sh - 1)))).astype(f32)
                want[dy, dx] = acc / ntaps
        got = orc.resize_hdr(src, dw)
        assert got.shape == want.shape
        # +, *, / only: the two readings must agree to the bit
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()


def test_cosine_convolve_second_reading(orc):
    rng = np.random.RandomState(13)
    w, h = 16, 8
    src = rng.uniform(0.0, 4.0, (h, w, 3)).astype(f32)
    px_theta = lambda p: f32(f32(f32(p) / f32(h - 1)) * PI)
    px_phi = lambda p: f32(f32(f32(f32(p) / f32(w - 1)) * f32(2)) * PI)
    for power in (1.0, 8.0, 3.0):
        want = np.empty_like(src)
        for dy in range(h):
            tl = px_theta(dy)
            tlc, tls = np.cos(tl, dtype=f32), np.sin(tl, dtype=f32)
            for dx in range(w):
                pl = px_phi(dx)
                lut = [np.cos(np.abs(f32(pl - px_phi(x))), dtype=f32) for x in range(w)]
                ar = ag = ab = n = f32(0)
                for y in range(h):
                    tp = px_theta(y)
                    tpc, tps = np.cos(tp, dtype=f32), np.sin(tp, dtype=f32)
                    for x in range(w):
                        r, g, b = src[y, x]
                        ca = f32(f32(tlc * tpc) + f32(f32(tls * tps) * lut[x]))
                        if ca > 0:
                            fac = f32(tps * np.power(ca, f32(power), dtype=f32))
                            ar, ag, ab, n = f32(ar + f32(r * fac)), f32(ag + f32(g * fac)), f32(ab + f32(b * fac)), f32(n + f32(1))
                want[dy, dx] = (ar / n, ag / n, ab / n)
        got = orc.cosine_convolve(src, power, pow_mode=0)            # the reference's literal `**` = libm powf
        rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-6)
        # sin / cos / pow of numpy against glibc: last-bit differences of the factors, averaged over ~64 terms per texel
        assert rel.max() < 5e-6, (power, rel.max())
        if power in (1.0, 8.0):                                     # the pinned squaring chain stays within the same distance of both
            pin = orc.cosine_convolve(src, power, pow_mode=1)
            assert (np.abs(pin.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-6)).max() < 5e-6


# ---- Fractal2D.hs (BASELINE config 1, the CPU plumbing path): the same second reading ------------------------------------------

def _escape_time(zr, zi, cr, ci, max_iter=40):
    """`go` of Fractal2D.hs:46-52 / 85-90, all pixels at once: (iCnt, escZ)"""
    zr, zi = zr.astype(f32).copy(), zi.astype(f32).copy()
    it = np.zeros(zr.shape, np.int32)
    done = np.zeros(zr.shape, bool)
    cnt = np.zeros(zr.shape, np.int32)
    er, ei = np.zeros_like(zr), np.zeros_like(zi)
    for _ in range(max_iter + 1):
        mag = (zr * zr + zi * zi).astype(f32)
        stop = ~done & ((it == max_iter) | (mag > f32(16)))
        cnt[stop], er[stop], ei[stop] = it[stop], zr[stop], zi[stop]
        done |= stop
        nr = ((zr * zr).astype(f32) - (zi * zi).astype(f32) + cr).astype(f32)        # (a :+ b) * (a :+ b) + c in Float
        ni = ((zr * zi).astype(f32) + (zi * zr).astype(f32) + ci).astype(f32)
        cyc = ~done & (nr == zr) & (ni == zi)                                        # 1-cycle: (maxIter, z)
        cnt[cyc], er[cyc], ei[cyc] = max_iter, zr[cyc], zi[cyc]
        done |= cyc
        go = ~done
        zr, zi = np.where(go, nr, zr), np.where(go, ni, zi)
        it = it + go.astype(np.int32)
    assert done.all()
    return cnt, er, ei


def _to_green(cnt, er, ei, smooth, max_iter=40):
    if smooth:
        with np.errstate(invalid="ignore", divide="ignore"):
            mag = (er * er + ei * ei).astype(f32)
            frac = np.maximum(f32(0), cnt.astype(f32) - (np.log(np.log(mag, dtype=f32), dtype=f32) / np.log(f32(2))).astype(f32)).astype(f32)
        v = np.where(cnt == max_iter, f32(max_iter), frac).astype(f32)
    else:
        v = cnt.astype(f32)
    return (np.trunc((v / f32(max_iter)).astype(f32) * f32(255)).astype(np.uint32)) << 8


def _julia(w, h, smooth, tick):
    ft = f32(tick)
    frac = lambda x: f32(x - np.trunc(x))                                          # snd . properFraction
    s1, s2, s3 = frac(f32(ft / f32(17))), frac(f32(ft / f32(61))), frac(f32(ft / f32(71)))
    two_pi = f32(f32(s1 * f32(2)) * PI)
    jr = f32(np.sin(two_pi, dtype=f32) * max(f32(0.7), s2))
    ji = f32(np.cos(two_pi, dtype=f32) * max(f32(0.7), s3))
    fw, fh = f32(w), f32(h)
    ratio = f32(fw / fh)
    xshift = f32(f32(1.45) * ratio)
    px, py = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32))
    y = ((py / fh).astype(f32) * f32(2.9) - f32(1.45)).astype(f32)
    x = (((px / fw).astype(f32) * f32(2.9)).astype(f32) * ratio - xshift).astype(f32)
    return _to_green(*_escape_time(x, y, jr, ji), smooth)


def _mandelbrot(w, h, smooth):
    fw, fh = f32(w), f32(h)
    ratio = f32(fw / fh)
    xshift = f32(f32(-2) - f32(f32(f32(2) * ratio - f32(2.5)) * f32(0.5)))
    px, py = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32))
    y = ((py / fh).astype(f32) * f32(2) - f32(1)).astype(f32)
    x = (((px / fw).astype(f32) * f32(2)).astype(f32) * ratio + xshift).astype(f32)
    return _to_green(*_escape_time(np.zeros_like(x), np.zeros_like(y), x, y), smooth)


def test_fractal2d_second_reading(orc):
    for (w, h) in ((64, 64), (96, 40)):
        for tick in (0.0, 3.7, 123.4):
            got = orc.julia_animated(w, h, 0, tick).reshape(h, w)
            assert np.array_equal(got, _julia(w, h, False, tick)), (w, h, tick)          # iteration counts: identical
            gs = orc.julia_animated(w, h, 1, tick).reshape(h, w).astype(np.int64) >> 8
            ws = _julia(w, h, True, tick).astype(np.int64) >> 8
            assert np.abs(gs - ws).max() <= 1 and (gs != ws).mean() < 5e-3               # smooth: log of numpy vs glibc, at truncation steps
        assert np.array_equal(orc.mandelbrot(w, h, 0).reshape(h, w), _mandelbrot(w, h, False))
        gs = orc.mandelbrot(w, h, 1).reshape(h, w).astype(np.int64) >> 8
        ws = _mandelbrot(w, h, True).astype(np.int64) >> 8
        assert np.abs(gs - ws).max() <= 1 and (gs != ws).mean() < 5e-3
