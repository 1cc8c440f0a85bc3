#!/usr/bin/env python3
"""Generate the committed fixtures from the CPU oracle (run in the build container).

    python tests/golden/make_fixtures.py [--caches] [--renders] [--env] [--fractals]

* --caches   tests/golden/env_cache/uffizi_512_cache_pow_{1,8,64,512}.0.hdr: the pre-convolved environment maps the
             reference keeps next to its .hdr (buildPreConvolvedHDREnvMapCache, ShaderRendering.hs:131-149), produced by
             the ORACLE's resizeHDRImage + cosineConvolveHDREnvMap (pinned cos^p) + RGBE encode.  They are the expected
             output of the product's rmdf_load_env_hdr cache-miss path (tests/test_gpu_env.py), not product data: the
             product ships uffizi_512.hdr alone and builds its caches on the GPU at first load.
* --cubes    tests/golden/env_cubes_uffizi.npz: the oracle-built padded RGB16F cube maps of the three sampled slots (so
             that the full-size digests below do not depend on the libm of the box that checks them).
* --digests  tests/golden/full_size_digests.json: sha256 of the oracle's rgba8 / steps / iters planes of BASELINE
             configs 2 and 3 at full size (1280x720 @128 Cornell, 1920x1080 @256 Mandelbulb), from those cube maps.
* --grid     tests/golden/grid_256x144_digests.json: sha256 of the oracle's planes for every FragmentShader value at in_time 0, 1, 2.5 and 7
             at 256x144 (SURVEY 8c's fixture grid), from the committed cube maps.
* --digest4  adds BASELINE config 4 (7680x4320 rays @256 -> box-resolved 3840x2160) to that file (~4 min on 8 cores).
* --renders  tests/golden/render_<scene>_<w>x<h>_t<time>.npz : oracle float RGBA / RGBA8 / steps / iters.
* --env      tests/golden/env_*.npz : cube faces for the procedural test env map, pixelAtBilinear probes,
             a 32x16 prefilter case.
* --fractals tests/golden/julia_*.npz, mandelbrot_*.npz.

The reference has no test vectors of its own; these files pin the ORACLE so that a change
of compiler, libm or source that alters its output is caught on the CPU tier.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ENVDIR = os.path.join(GOLD, "env_cache")
HDR = os.path.join(ROOT, "ray-marching-distance-fields_amd", "data", "latlong_envmaps", "uffizi_512.hdr")
POWERS = (1.0, 8.0, 64.0, 512.0)
RENDER_CASES = [(scene, w, h, t, ms)
                for scene, ms in ((orc.SCENE_MB_POWER8, 256), (orc.SCENE_CORNELL, 128))
                for (w, h) in ((64, 36),)
                for t in (0.0, 1.0, 2.5, 7.0)] + \
               [(orc.SCENE_MB_POWER8, 256, 144, 0.0, 256), (orc.SCENE_CORNELL, 256, 144, 0.0, 128),
                (orc.SCENE_MB_POWER8, 64, 36, 0.0, 128)] + \
               [(scene, 64, 36, t, 128) for scene in (orc.SCENE_DETEST, orc.SCENE_MB_GENERAL) for t in (0.0, 2.5, 11.0)]


def cache_name(power):
    return os.path.join(ENVDIR, "uffizi_512_cache_pow_%s.hdr" % repr(float(power)))


def load_env():
    refl = orc.hdr_decode(open(HDR, "rb").read())
    c1 = orc.hdr_decode(open(cache_name(1.0), "rb").read())
    c8 = orc.hdr_decode(open(cache_name(8.0), "rb").read())
    return orc.EnvSet.from_latlongs(refl, c1, c8)


def render_name(scene, w, h, t, ms):
    return os.path.join(GOLD, "render_s%d_%dx%d_t%s_m%d.npz" % (scene, w, h, ("%.1f" % t).replace(".", "p"), ms))


def main():
    ap = argparse.ArgumentParser()
    for f in ("caches", "renders", "env", "fractals", "cubes", "digests", "digest4", "grid"):
        ap.add_argument("--" + f, action="store_true")
    a = ap.parse_args()
    if not (a.caches or a.renders or a.env or a.fractals or a.cubes or a.digests or a.digest4 or a.grid):
        a.caches = a.renders = a.env = a.fractals = a.cubes = a.digests = True

    if a.caches:
        refl = orc.hdr_decode(open(HDR, "rb").read())
        small = orc.resize_hdr(refl, 256)
        for p in POWERS:
            conv = orc.cosine_convolve(small, p)
            open(cache_name(p), "wb").write(orc.hdr_encode(conv))
            print("wrote", cache_name(p))

    if a.cubes:
        env = load_env()
        np.savez_compressed(os.path.join(GOLD, "env_cubes_uffizi.npz"), refl=env.reflection, cos1=env.cos_1, cos8=env.cos_8)
        print("wrote env_cubes_uffizi.npz")

    if a.digests:
        import hashlib
        import json
        z = np.load(os.path.join(GOLD, "env_cubes_uffizi.npz"))
        env = orc.EnvSet(z["refl"], z["cos1"], z["cos8"])
        out = {}
        for name, (scene, w, h, t, ms) in (("config2_cornell_1280x720_m128", (orc.SCENE_CORNELL, 1280, 720, 0.0, 128)),
                                            ("config3_mandelbulb8_1920x1080_m256", (orc.SCENE_MB_POWER8, 1920, 1080, 0.0, 256)),
                                            ("detest_1280x720_t2p5_m128", (orc.SCENE_DETEST, 1280, 720, 2.5, 128)),
                                            ("mbgeneral_1280x720_t3p0_m128", (orc.SCENE_MB_GENERAL, 1280, 720, 3.0, 128)),
                                            ("mandelbulb8_1920x1080_t7p0_m256", (orc.SCENE_MB_POWER8, 1920, 1080, 7.0, 256))):
            r = orc.render(scene, w, h, t, ms, env, want_f32=False)
            out[name] = {"scene": scene, "w": w, "h": h, "time": t, "max_steps": ms,
                         "sha256": {k: hashlib.sha256(np.ascontiguousarray(r[k]).tobytes()).hexdigest() for k in ("rgba8", "steps", "iters")},
                         "counters": r["counters"]}
            print(name, out[name])
        fn = os.path.join(GOLD, "full_size_digests.json")
        if os.path.exists(fn):                                  # entries written by --digest4 stay
            out = dict(json.load(open(fn)), **out)
        json.dump(out, open(fn, "w"), indent=1, sort_keys=True)

    if a.grid:
        # SURVEY 8c's fixture grid at 256x144 -- every FragmentShader value at in_time 0, 1, 2.5 and 7 -- as digests of the oracle's
        # planes (the full arrays of two of them are committed as render_*_256x144_*.npz; sixteen would be 8 MB)
        import hashlib
        import json
        z = np.load(os.path.join(GOLD, "env_cubes_uffizi.npz"))
        env = orc.EnvSet(z["refl"], z["cos1"], z["cos8"])
        out = {}
        for scene, ms in ((orc.SCENE_CORNELL, 128), (orc.SCENE_DETEST, 128), (orc.SCENE_MB_POWER8, 256), (orc.SCENE_MB_GENERAL, 128)):
            for t in (0.0, 1.0, 2.5, 7.0):
                r = orc.render(scene, 256, 144, t, ms, env)
                out["s%d_t%s_m%d" % (scene, ("%.1f" % t).replace(".", "p"), ms)] = {
                    "scene": scene, "w": 256, "h": 144, "time": t, "max_steps": ms,
                    "sha256": {k: hashlib.sha256(np.ascontiguousarray(r[k]).tobytes()).hexdigest() for k in ("rgba8", "steps", "iters", "rgba_f32")},
                    "hit_pixels": int((r["steps"] >> 15).sum())}
        json.dump(out, open(os.path.join(GOLD, "grid_256x144_digests.json"), "w"), indent=1, sort_keys=True)
        print("wrote grid_256x144_digests.json (%d frames)" % len(out))

    if a.digest4:
        # BASELINE config 4 on every pixel: 7680x4320 rays @256 (frame-buffer scale 2, App.hs:105-106,131-133), one mip level of the
        # RGBA8 frame (FrameBuffer.hs:153-154,187-195) -> 3840x2160.  ~4 minutes on 8 cores; added to the existing digest file.
        import hashlib
        import json
        z = np.load(os.path.join(GOLD, "env_cubes_uffizi.npz"))
        env = orc.EnvSet(z["refl"], z["cos1"], z["cos8"])
        r = orc.render(orc.SCENE_MB_POWER8, 7680, 4320, 0.0, 256, env, want_f32=False)
        res = orc.resolve_box2(r["rgba8"])
        assert res.shape == (2160, 3840)
        fn = os.path.join(GOLD, "full_size_digests.json")
        out = json.load(open(fn))
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        out["config4_mandelbulb8_3840x2160_x4rays_m256"] = {
            "scene": orc.SCENE_MB_POWER8, "w": 3840, "h": 2160, "supersample_levels": 1, "time": 0.0, "max_steps": 256,
            "sha256": {"rgba8_resolved_3840x2160": sha(res), "rgba8_rays_7680x4320": sha(r["rgba8"]),
                       "steps_rays_7680x4320": sha(r["steps"]), "iters_rays_7680x4320": sha(r["iters"])},
            "counters": r["counters"]}
        print(out["config4_mandelbulb8_3840x2160_x4rays_m256"])
        json.dump(out, open(fn, "w"), indent=1, sort_keys=True)

    if a.renders:
        env = load_env()
        for (scene, w, h, t, ms) in RENDER_CASES:
            r = orc.render(scene, w, h, t, ms, env)
            np.savez_compressed(render_name(scene, w, h, t, ms), rgba_f32=r["rgba_f32"], rgba8=r["rgba8"],
                                steps=r["steps"], iters=r["iters"],
                                counters=np.array(list(r["counters"].values()), np.uint64))
            print("wrote", render_name(scene, w, h, t, ms), r["counters"])

    if a.env:
        test_ll = orc.build_test_latlong()
        faces = orc.latlong_to_cube(test_ll)
        refl = orc.hdr_decode(open(HDR, "rb").read())
        rng = np.random.RandomState(1234)
        uv = rng.rand(64, 2).astype(np.float32)
        bil = np.stack([orc.pixel_at_bilinear(refl, u, v) for u, v in uv])
        small = orc.resize_hdr(refl, 32)
        pre = np.stack([orc.cosine_convolve(small, p) for p in (1.0, 8.0)])
        ufaces = orc.latlong_to_cube(refl)
        np.savez_compressed(os.path.join(GOLD, "env_vectors.npz"), test_faces=faces.astype(np.float16),
                            bil_uv=uv, bil_rgb=bil, small32=small, prefilter32=pre,
                            uffizi_faces_sample=ufaces[:, ::17, ::17], uffizi_padded_corner=orc.cube_pad_f16(ufaces)[:, :3, :3])
        print("wrote env_vectors.npz")

    if a.fractals:
        out = {}
        for tick in (0.0, 3.7):
            for smooth in (0, 1):
                out["julia_t%s_s%d" % (("%.1f" % tick).replace(".", "p"), smooth)] = orc.julia_animated(64, 64, smooth, tick)
        for smooth in (0, 1):
            out["mandelbrot_s%d" % smooth] = orc.mandelbrot(96, 64, smooth)
        np.savez_compressed(os.path.join(GOLD, "fractal_vectors.npz"), **out)
        print("wrote fractal_vectors.npz")


if __name__ == "__main__":
    main()
