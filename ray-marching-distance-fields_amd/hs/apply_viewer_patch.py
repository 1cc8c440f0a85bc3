#!/usr/bin/env python3
"""INTEGRATION.md section 3 as a checked transformation: apply_viewer_patch.py <reference checkout> <output dir>

Reads App.hs, Main.hs and rmdf.cabal of a checkout of blitzcode/ray-marching-distance-fields, makes the edits INTEGRATION.md describes in
prose -- every edit anchored on a short fragment that must occur EXACTLY the stated number of times, or the script stops and says which
one drifted -- and writes the three patched files plus a copy of RmdfFFI.hs into <output dir> (never into the checkout, never into this
repository: the result contains the reference's source).  What it cannot do here is compile the result: the build image has no GHC.

Edits:
  rmdf.cabal  other-modules gets RmdfFFI; extra-libraries: rmdf (+ an extra-lib-dirs line to fill in)
  App.hs      import RmdfFFI; AppEnv gets `_aeHR :: HipRenderer` next to `_aeSR`; in `draw` the four shader modes move from
              `drawFB ... drawShader` (GL draws into the FBO) to `fillFB ... hipShader` (librmdf fills the vector fillFrameBuffer hands out)
  Main.hs     the renderer bracket opens `withHipRenderer reflMapFn` inside `withShaderRenderer`
"""
import os
import re
import shutil
import sys


class Drift(Exception):
    pass


def sub_exact(text, pattern, repl, count, what):
    """re.subn with the number of matches it must find"""
    new, n = re.subn(pattern, repl, text, flags=re.M)
    if n != count:
        raise Drift("%s: expected %d occurrence(s) of /%s/, found %d" % (what, count, pattern, n))
    return new


def patch_app(src):
    s = src
    s = sub_exact(s, r"^(import ShaderRendering.*)$", r"\1\nimport RmdfFFI", 1, "App.hs: the ShaderRendering import")
    s = sub_exact(s, r"^(\s*), _aeSR(\s*):: ShaderRenderer\s*$", r"\1, _aeSR\2:: ShaderRenderer\n\1, _aeHR\2:: HipRenderer", 1,
                  "App.hs: AppEnv's _aeSR field")
    # the helper bound next to drawShader, and the four shader modes
    s = sub_exact(s, r"^(\s*)drawShader shd w h = drawShaderTile _aeSR shd tileIdx w h _asCurTick\s*$",
                  r"\1drawShader shd w h = drawShaderTile _aeSR shd tileIdx w h _asCurTick\n"
                  r"\1hipShader shd w h vec = drawHipTile _aeHR shd tileIdx w h _asCurTick vec >>= either (traceS TLError) return", 1,
                  "App.hs: draw's drawShader helper")
    s = sub_exact(s, r"-> drawFB \$ \\w h\s+-> drawShader (FS\w+)\s+w h", r"-> fillFB $ \\w h fbVec -> hipShader \1 w h fbVec", 4,
                  "App.hs: the four shader modes of draw")
    return s


def patch_main(src):
    return sub_exact(src, r"^(\s*)withShaderRenderer shdFn reflMapFn \$ \\_aeSR -> do\s*$",
                     r"\1withShaderRenderer shdFn reflMapFn $ \\_aeSR ->\n\1withHipRenderer reflMapFn $ \\_aeHR -> do", 1,
                     "Main.hs: the withShaderRenderer bracket")


def patch_cabal(src):
    s = sub_exact(src, r"^(\s*)-- other-modules:\s*$", r"\1other-modules:       RmdfFFI\n\1extra-libraries:     rmdf\n\1-- extra-lib-dirs:   <directory of librmdf.so>", 1,
                  "rmdf.cabal: the (commented) other-modules line")
    return s


def main():
    ref, out = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    if os.path.commonpath([os.path.abspath(out), os.path.dirname(os.path.dirname(here))]) == os.path.dirname(os.path.dirname(here)):
        raise SystemExit("the output directory must lie outside this repository (the result contains the reference's source)")
    os.makedirs(out, exist_ok=True)
    for name, fn in (("App.hs", patch_app), ("Main.hs", patch_main), ("rmdf.cabal", patch_cabal)):
        src = open(os.path.join(ref, name)).read()
        new = fn(src)
        open(os.path.join(out, name), "w").write(new)
        changed = sum(1 for a, b in zip(src.splitlines(), new.splitlines()) if a != b) + abs(len(new.splitlines()) - len(src.splitlines()))
        print("%s: patched (%d lines before, %d after)" % (name, len(src.splitlines()), len(new.splitlines())))
    shutil.copy(os.path.join(here, "RmdfFFI.hs"), os.path.join(out, "RmdfFFI.hs"))
    print("wrote %s" % out)


if __name__ == "__main__":
    try:
        main()
    except Drift as e:
        raise SystemExit("the reference has drifted from what INTEGRATION.md describes -- %s" % e)
