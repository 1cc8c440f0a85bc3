#!/usr/bin/env python3
"""Kernel time of every FragmentShader value at its committed view, one frame at a time (HIP events around each launch, after a warm-up
that brings the shader clock up) and two frames in flight; for A/B of library builds: RMDF_LIB=<path> tools/scene_times.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hashlib, json
import numpy as np, torch, rmdf_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
sr = rmdf_amd.ShaderRenderer(0); sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
print("library:", os.environ.get("RMDF_LIB", rmdf_amd.LIB_PATH))
DIG = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "full_size_digests.json")))
DNAME = {0: "config2_cornell_1280x720_m128", 1: "detest_1280x720_t2p5_m128", 2: "config3_mandelbulb8_1920x1080_m256", 3: "mbgeneral_1280x720_t3p0_m128"}
for name, sc, w, h, ms, t in (("config2 cornell", 0, 1280, 720, 128, 0.0), ("scene1 detest", 1, 1280, 720, 128, 2.5), ("headline mb8", 2, 1920, 1080, 256, 0.0), ("scene3 mbgeneral", 3, 1280, 720, 128, 3.0)):
    fb = [torch.empty((h, w), dtype=torch.int32, device=dev) for _ in range(2)]
    one = lambda k=0, st=s0: sr.render_rect_device(sc, w, h, t, ms, (0, 0, w, h), d_rgba8=fb[k].data_ptr(), stream=st.cuda_stream)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10): one()
        torch.cuda.synchronize(dev)
    best = []
    for blk in range(3):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in evs:
            e0.record(s0); one(); e1.record(s0)
        torch.cuda.synchronize(dev)
        best.append(float(np.mean([a.elapsed_time(b) for a, b in evs])))
    two = []
    for blk in range(3):
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for i in range(2 * reps): one(i & 1, (s0, s1)[i & 1])
        torch.cuda.synchronize(dev); two.append((time.perf_counter() - t0) / (2 * reps) * 1e3)
    torch.cuda.synchronize(dev)
    ok = hashlib.sha256(fb[0].cpu().numpy().tobytes()).hexdigest() == DIG[DNAME[sc]]["sha256"]["rgba8"]      # an A/B build must still render the committed frame
    mp = w * h / 1e6
    print("%-18s %dx%d @%d: one at a time %.4f ms (%.0f Mpixels/s; blocks %s); two in flight %.4f ms per frame (%.0f Mpixels/s); frame %s the committed digest" % (
        name, w, h, ms, sorted(best)[1], mp / (sorted(best)[1] * 1e-3), " ".join("%.4f" % b for b in best), sorted(two)[1], mp / (sorted(two)[1] * 1e-3), "==" if ok else "DIFFERS FROM"), flush=True)
