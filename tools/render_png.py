#!/usr/bin/env python3
"""Render one frame of every FragmentShader value through the host-buffer boundary and save it the way the viewer's
screenshot key does (FrameBuffer.saveFrameBufferToPNG -> rmdf_save_png).   usage: render_png.py [outdir] [w h time]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmdf_amd
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/shots"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (960, 540)
t = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
os.makedirs(out, exist_ok=True)
with rmdf_amd.with_shader_renderer() as sr:
    for shd in rmdf_amd.FragmentShader:
        fb = rmdf_amd.FrameBuffer(w, h)
        fb.fill_frame_buffer(lambda w_, h_, vec: sr.draw_shader_tile(shd, None, w_, h_, t, vec, max_steps=256 if shd == 2 else 128))
        fn = os.path.join(out, "%s_%dx%d_t%g.png" % (shd.name, w, h, t))
        fb.save_png(fn)
        print(fn)
