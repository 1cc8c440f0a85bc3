#!/usr/bin/env python3
"""Tier-B cross-check vectors: the REFERENCE's fragment.shd executed on SwiftShader (GLES 3.0, software).

Runs only in the build container: it reads /root/reference/fragment.shd at run time, patches it mechanically for
GLSL ES 3.00 (never writing the patched text anywhere), renders into an RGBA32F target with the environment cube
maps / Cornell geometry produced by the oracle's data-prep restatement, and stores the resulting float images as
tests/golden/swiftshader_*.npz.  tests/test_oracle_vs_glsl.py then compares the CPU oracle with these images.

The patch:
  * header `#version 300 es` + highp precision, the variant #defines of ShaderRendering.hs:119-122
  * sampler1D -> sampler2D and ivec2 coordinates in the three texelFetch calls (fragment.shd:16,404-406)
  * int -> float literals on the 24 lines where GLSL 3.30 converts implicitly and GLSL ES does not
  * MAX_STEPS set to the requested value; the march loop counter exported through a global so it can be written to the
    alpha channel (alpha is constant 1 in the original);
  * a second program per variant ("counters") adds `g_iters += 1.0` behind `w += pos;` in de_mandelbulb's loop
    (fragment.shd:134-152) and writes (g_iters, g_steps, g_hit, 1) instead of the colour: the per-pixel total of Mandelbulb
    iterations that ran triplex_pow -- over the march, the four normal taps and the two AO taps -- i.e. the escape-iteration
    counts the north star wants bit-exact, read back from the reference shader itself (g_iters_march: the part spent inside
    ray_march, snapshotted at its three exits).
  * a third program ("gbuffer") exports what the shader's hit branch feeds into its shading: `g_n = isec_n; g_ao = ao;` behind
    `float ao = distance_ao(isec_pos, isec_n);` (fragment.shd:769) and writes (g_n, g_ao) instead of the colour.  Given these,
    shading is a short deterministic function -- Fresnel, reflect, three cube-map lookups, gamma -- so
    tests/test_oracle_vs_glsl.py can compare the oracle's SHADING with the reference shader's colour tightly, with the chaotic
    eps = 1e-5 normal differentiation taken out of the comparison (frames *_gbuf.npz, float32 colour kept).
SwiftShader's float math is its own (its inversesqrt/pow/log/exp are approximations, its texture filter uses
fixed-point weights), so this is a tolerance-level cross-check of the RESTATEMENT, not a bit-level oracle.
"""
import ctypes as C
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

REF_SHADER = "/root/reference/fragment.shd"
SS_DIR = "/usr/local/lib/python3.10/dist-packages/kaleido/executable/bin/swiftshader/"
GOLD = os.path.join(ROOT, "tests", "golden")
HDR = os.path.join(ROOT, "ray-marching-distance-fields_amd", "data", "latlong_envmaps", "uffizi_512.hdr")
ENVDIR = os.path.join(GOLD, "env_cache")

INT_LITERAL_LINES = [113, 116, 118, 119, 121, 318, 340, 345, 363, 400, 487, 537, 557, 561, 588, 657, 723, 770, 808, 867,
                     888, 889, 890]
VARIANTS = {0: "#define CORNELL_BOX_SCENE\n", 1: "", 2: "#define MANDELBULB_SCENE\n#define POWER8\n", 3: "#define MANDELBULB_SCENE\n"}


def patched_shader(scene, max_steps, counters=False, gbuffer=False):
    lines = open(REF_SHADER).read().split("\n")
    lit = re.compile(r"(?<![\w.])(\d+)(?![\w.])")
    for ln in INT_LITERAL_LINES + (list(range(448, 457)) if scene == 1 else []):
        lines[ln - 1] = lit.sub(r"\1.0", lines[ln - 1])
    lines[522 - 1] = lines[522 - 1].replace("pow(i + 1.0", "pow(float(i) + 1.0")
    src = "\n".join(lines)
    src = src.replace("uniform sampler1D cornell_geom;", "uniform highp sampler2D cornell_geom;")
    src = re.sub(r"texelFetch\(cornell_geom, (i \* 3 \+ \d), 0\)", r"texelFetch(cornell_geom, ivec2(\1, 0), 0)", src)
    src = src.replace("const int   MAX_STEPS = 128;", "const int   MAX_STEPS = %d;" % max_steps)
    # export the loop counter: alpha = steps (+ 0.5 if hit)
    src = src.replace("out vec4 frag_color;", "out vec4 frag_color;\nfloat g_steps = 0.0;\nfloat g_hit = 0.0;")
    src = src.replace("        if (t > tspheremax) // Left bounding sphere?\n            return false;",
                      "        if (t > tspheremax) // Left bounding sphere?\n        { g_steps = float(steps); return false; }")
    src = src.replace("            step_gradient = 1.0 - float(steps) / float(MAX_STEPS);\n            return true;",
                      "            step_gradient = 1.0 - float(steps) / float(MAX_STEPS);\n            g_steps = float(steps); g_hit = 1.0; return true;")
    src = src.replace("    }\n\n    return false;\n}\n\nvec3 soft_lam", "    }\n\n    g_steps = float(MAX_STEPS); return false;\n}\n\nvec3 soft_lam")
    if gbuffer:
        assert src.count("        float ao = distance_ao(isec_pos, isec_n);\n") == 1
        src = src.replace("out vec4 frag_color;", "out vec4 frag_color;\nvec3 g_n = vec3(0.0);\nfloat g_ao = 0.0;", 1)
        src = src.replace("        float ao = distance_ao(isec_pos, isec_n);\n", "        float ao = distance_ao(isec_pos, isec_n); g_n = isec_n; g_ao = ao;\n")
        src = src.replace("    frag_color = vec4(gamma, 1);", "    frag_color = vec4(g_n, g_ao + 0.0 * gamma.x);")
    elif counters:
        assert src.count("        w += pos;\n") == 1
        src = src.replace("out vec4 frag_color;", "out vec4 frag_color;\nfloat g_iters = 0.0;\nfloat g_iters_march = 0.0;", 1)
        src = src.replace("        w += pos;\n", "        w += pos; g_iters += 1.0;\n")
        # snapshot at the three exits of ray_march (where g_steps is set): iterations spent in the march alone
        assert src.count("g_steps = float(") == 3
        src = src.replace("g_steps = float(", "g_iters_march = g_iters; g_steps = float(")
        src = src.replace("    frag_color = vec4(gamma, 1);", "    frag_color = vec4(g_iters, g_steps + 0.5 * g_hit, g_iters_march, 1.0 + 0.0 * gamma.x);")
    else:
        src = src.replace("    frag_color = vec4(gamma, 1);", "    frag_color = vec4(gamma, g_steps + 0.5 * g_hit);")
    header = "#version 300 es\nprecision highp float;\nprecision highp int;\nprecision highp samplerCube;\n" + VARIANTS[scene]
    return header + src


VS = """#version 300 es
precision highp float;
uniform vec4 quad;
void main()
{
    vec2 v[4] = vec2[4](vec2(quad.x, quad.y), vec2(quad.z, quad.y), vec2(quad.x, quad.w), vec2(quad.z, quad.w));
    gl_Position = vec4(v[gl_VertexID], 0.0, 1.0);
}
"""

# GL / EGL constants
EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT, EGL_NONE = 0x3033, 0x0001, 0x3040, 0x0040, 0x3038
EGL_WIDTH, EGL_HEIGHT, EGL_OPENGL_ES_API, EGL_CONTEXT_CLIENT_VERSION = 0x3057, 0x3056, 0x30A0, 0x3098
GL_TEXTURE_2D, GL_TEXTURE_CUBE_MAP, GL_TEXTURE_CUBE_MAP_POSITIVE_X = 0x0DE1, 0x8513, 0x8515
GL_RGBA32F, GL_RGBA16F, GL_RGB32F, GL_RGBA, GL_RGB, GL_FLOAT = 0x8814, 0x881A, 0x8815, 0x1908, 0x1907, 0x1406
GL_TEXTURE_MIN_FILTER, GL_TEXTURE_MAG_FILTER, GL_NEAREST, GL_LINEAR = 0x2801, 0x2800, 0x2600, 0x2601
GL_TEXTURE_WRAP_S, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE = 0x2802, 0x2803, 0x812F
GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_FRAMEBUFFER_COMPLETE = 0x8D40, 0x8CE0, 0x8CD5
GL_VERTEX_SHADER, GL_FRAGMENT_SHADER, GL_COMPILE_STATUS, GL_LINK_STATUS = 0x8B31, 0x8B30, 0x8B81, 0x8B82
GL_TRIANGLE_STRIP, GL_TEXTURE0, GL_UNPACK_ALIGNMENT, GL_PACK_ALIGNMENT = 0x0005, 0x84C0, 0x0CF5, 0x0D05


class GLES:
    def __init__(self):
        self.gl = C.CDLL(SS_DIR + "libGLESv2.so", mode=C.RTLD_GLOBAL)
        self.egl = C.CDLL(SS_DIR + "libEGL.so", mode=C.RTLD_GLOBAL)
        e = self.egl
        e.eglGetDisplay.restype = C.c_void_p
        e.eglGetDisplay.argtypes = [C.c_void_p]
        dpy = C.c_void_p(e.eglGetDisplay(None))
        assert e.eglInitialize(dpy, None, None)
        cfg_attr = (C.c_int * 5)(EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT, EGL_NONE)
        cfg, n = C.c_void_p(), C.c_int()
        assert e.eglChooseConfig(dpy, cfg_attr, C.byref(cfg), 1, C.byref(n)) and n.value == 1
        e.eglCreatePbufferSurface.restype = C.c_void_p
        surf = C.c_void_p(e.eglCreatePbufferSurface(dpy, cfg, (C.c_int * 5)(EGL_WIDTH, 16, EGL_HEIGHT, 16, EGL_NONE)))
        assert e.eglBindAPI(EGL_OPENGL_ES_API)
        e.eglCreateContext.restype = C.c_void_p
        ctx = C.c_void_p(e.eglCreateContext(dpy, cfg, None, (C.c_int * 3)(EGL_CONTEXT_CLIENT_VERSION, 3, EGL_NONE)))
        assert ctx.value and e.eglMakeCurrent(dpy, surf, surf, ctx)
        g = self.gl
        g.glGetString.restype = C.c_char_p
        self.version = g.glGetString(0x1F02).decode()
        g.glGetUniformLocation.argtypes = [C.c_uint, C.c_char_p]
        g.glUniform1f.argtypes = [C.c_int, C.c_float]
        g.glUniform4f.argtypes = [C.c_int] + [C.c_float] * 4

    def shader(self, kind, src):
        g = self.gl
        s = g.glCreateShader(kind)
        b = src.encode()
        g.glShaderSource(s, 1, C.byref(C.c_char_p(b)), None)
        g.glCompileShader(s)
        ok = C.c_int()
        g.glGetShaderiv(s, GL_COMPILE_STATUS, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(8192)
            g.glGetShaderInfoLog(s, 8192, None, log)
            raise RuntimeError("shader compile failed:\n" + log.value.decode())
        return s

    def program(self, fs_src):
        g = self.gl
        p = g.glCreateProgram()
        g.glAttachShader(p, self.shader(GL_VERTEX_SHADER, VS))
        g.glAttachShader(p, self.shader(GL_FRAGMENT_SHADER, fs_src))
        g.glLinkProgram(p)
        ok = C.c_int()
        g.glGetProgramiv(p, GL_LINK_STATUS, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(8192)
            g.glGetProgramInfoLog(p, 8192, None, log)
            raise RuntimeError("link failed:\n" + log.value.decode())
        return p

    def cube(self, faces_f32):
        """faces (6, W, W, 3) float32 -> RGBA16F cube map, MIN=NEAREST / MAG=LINEAR (TFMagOnly, GLHelpers.hs:105-106)"""
        g = self.gl
        t = C.c_uint()
        g.glGenTextures(1, C.byref(t))
        g.glBindTexture(GL_TEXTURE_CUBE_MAP, t)
        g.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)
        W = faces_f32.shape[1]
        for f in range(6):
            rgba = np.concatenate([faces_f32[f], np.ones((W, W, 1), np.float32)], axis=2).astype(np.float32).copy()
            g.glTexImage2D(GL_TEXTURE_CUBE_MAP_POSITIVE_X + f, 0, GL_RGBA16F, W, W, 0, GL_RGBA, GL_FLOAT, rgba.ctypes.data_as(C.c_void_p))
        g.glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_MIN_FILTER, GL_NEAREST)
        g.glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_MAG_FILTER, GL_LINEAR)
        return t

    def geom(self, verts):
        g = self.gl
        t = C.c_uint()
        g.glGenTextures(1, C.byref(t))
        g.glBindTexture(GL_TEXTURE_2D, t)
        g.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)
        v = np.ascontiguousarray(verts, np.float32)
        g.glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB32F, 96, 1, 0, GL_RGB, GL_FLOAT, v.ctypes.data_as(C.c_void_p))
        g.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST)
        g.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST)
        return t

    def render(self, prog, w, h, time, textures):
        g = self.gl
        fbo, tex = C.c_uint(), C.c_uint()
        g.glGenTextures(1, C.byref(tex))
        g.glBindTexture(GL_TEXTURE_2D, tex)
        g.glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, None)
        g.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST)
        g.glGenFramebuffers(1, C.byref(fbo))
        g.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
        g.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex, 0)
        assert g.glCheckFramebufferStatus(GL_FRAMEBUFFER) == GL_FRAMEBUFFER_COMPLETE
        g.glViewport(0, 0, w, h)
        g.glUseProgram(prog)
        vao = C.c_uint()
        g.glGenVertexArrays(1, C.byref(vao))
        g.glBindVertexArray(vao)
        g.glUniform1f(g.glGetUniformLocation(prog, b"in_screen_wdh"), float(w))
        g.glUniform1f(g.glGetUniformLocation(prog, b"in_screen_hgt"), float(h))
        g.glUniform1f(g.glGetUniformLocation(prog, b"in_time"), float(time))
        g.glUniform4f(g.glGetUniformLocation(prog, b"quad"), -1.0, -1.0, 1.0, 1.0)
        for unit, (name, target, t) in enumerate(textures):
            g.glActiveTexture(GL_TEXTURE0 + unit)
            g.glBindTexture(target, t)
            loc = g.glGetUniformLocation(prog, name)
            if loc >= 0:
                g.glUniform1i(loc, unit)
        g.glDrawArrays(GL_TRIANGLE_STRIP, 0, 4)
        g.glFinish()
        out = np.empty((h, w, 4), np.float32)
        g.glPixelStorei(GL_PACK_ALIGNMENT, 1)
        g.glReadPixels(0, 0, w, h, GL_RGBA, GL_FLOAT, out.ctypes.data_as(C.c_void_p))
        assert g.glGetError() == 0
        g.glDeleteFramebuffers(1, C.byref(fbo))
        g.glDeleteTextures(1, C.byref(tex))
        return out     # rows bottom-up like gl_FragCoord


CASES = [(2, 96, 54, 0.0, 128), (2, 96, 54, 2.5, 256), (2, 192, 108, 0.0, 256), (2, 480, 270, 0.0, 256),
         (0, 96, 54, 0.0, 128), (0, 128, 72, 1.0, 128), (0, 320, 180, 0.0, 128),
         (1, 96, 54, 0.0, 128), (1, 320, 180, 3.0, 128), (3, 96, 54, 0.0, 128), (3, 320, 180, 3.0, 128), (3, 160, 90, 11.0, 128)]


GBUF_CASES = [(2, 192, 108, 0.0, 256), (2, 96, 54, 2.5, 256), (0, 128, 72, 1.0, 128), (1, 96, 54, 0.0, 128)]


def main():
    rd = lambda n: orc.hdr_decode(open(os.path.join(ENVDIR, n), "rb").read())
    faces = {"env_reflection": orc.latlong_to_cube(orc.hdr_decode(open(HDR, "rb").read())),
             "env_cos_1": orc.latlong_to_cube(rd("uffizi_512_cache_pow_1.0.hdr")),
             "env_cos_8": orc.latlong_to_cube(rd("uffizi_512_cache_pow_8.0.hdr"))}
    gl = GLES()
    print(gl.version)
    tex = [(k.encode(), GL_TEXTURE_CUBE_MAP, gl.cube(v)) for k, v in faces.items()]
    tex.append((b"cornell_geom", GL_TEXTURE_2D, gl.geom(orc.cornell_vertices())))
    progs = {}
    for (scene, w, h, t, ms) in CASES:
        key = (scene, ms)
        if key not in progs:
            progs[key] = (gl.program(patched_shader(scene, ms)), gl.program(patched_shader(scene, ms, counters=True)))
        img = gl.render(progs[key][0], w, h, t, tex)
        cnt = gl.render(progs[key][1], w, h, t, tex)
        iters = np.rint(cnt[..., 0]).astype(np.uint32)
        iters_march = np.rint(cnt[..., 2]).astype(np.uint32)
        assert np.array_equal(cnt[..., 0], iters) and iters.max() < 65536 and np.array_equal(cnt[..., 2], iters_march)
        # the counters program marches exactly like the colour program
        assert np.array_equal(cnt[..., 1], img[..., 3])
        fn = os.path.join(GOLD, "swiftshader_s%d_%dx%d_t%s_m%d.npz" % (scene, w, h, ("%.1f" % t).replace(".", "p"), ms))
        alpha = img[..., 3]
        steps = np.floor(alpha).astype(np.uint16)
        hit = (alpha - np.floor(alpha)) > 0.25
        # float16 keeps the files small; 11 significant bits are ample for the statistical colour checks, the
        # tight background check uses the float32 copy of the bottom and top 8 rows
        np.savez_compressed(fn, rgb16=img[..., :3].astype(np.float16), steps=steps, hit=hit, iters=iters.astype(np.uint16), iters_march=iters_march.astype(np.uint16),
                            rows_f32=np.concatenate([img[:8, :, :3], img[-8:, :, :3]]).astype(np.float32))
        print("wrote", fn, "hit fraction %.4f" % hit.mean(), "max steps", steps.max(), "iterations", int(iters.sum()))
        if (scene, w, h, t, ms) in GBUF_CASES:
            gprog = gl.program(patched_shader(scene, ms, gbuffer=True))
            gb = gl.render(gprog, w, h, t, tex)
            fn2 = fn[:-4] + "_gbuf.npz"
            np.savez_compressed(fn2, nao=gb.astype(np.float32), rgb=img[..., :3].astype(np.float32), hit=hit, steps=steps)
            print("wrote", fn2)


if __name__ == "__main__":
    main()
