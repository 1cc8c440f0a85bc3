#!/usr/bin/env python3
"""Render one rectangle of the Cornell frame (config 2) a few times -- to be run under rocprofv3 --pmc: instruction counts of the few
waves of a slow region.  usage: cornell_rect_pmc.py x0 y0 x1 y1"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmdf_amd
x0, y0, x1, y1 = (int(v) for v in sys.argv[1:5])
W, H, MS = 1280, 720, 128
sr = rmdf_amd.ShaderRenderer(0)
sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
buf = torch.zeros(W * H, dtype=torch.int32, device="cuda")
for _ in range(5):
    sr.render_rect_device(0, W, H, 0.0, MS, (x0, y0, x1, y1), d_rgba8=buf.data_ptr())
sr.synchronize()
