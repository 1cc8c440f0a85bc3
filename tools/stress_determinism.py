#!/usr/bin/env python3
"""Race hunt for the ray pooling / frames-in-flight machinery: many frames on several streams, every one compared on the
device with the first (scene 2 and 0, two sizes).  Prints the number of differing frames (must be 0)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rmdf_amd
dev = torch.device("cuda", 0)
sr = rmdf_amd.ShaderRenderer(0); sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for (scene, w, h, ms) in ((2, 1920, 1080, 256), (2, 1000, 562, 256), (0, 1280, 720, 128), (3, 640, 360, 128)):
    S = 4
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    bufs = [torch.zeros((h, w), dtype=torch.int32, device=dev) for _ in range(S)]
    ref = torch.zeros((h, w), dtype=torch.int32, device=dev)
    sr.render_rect_device(scene, w, h, 0.0, ms, (0, 0, w, h), d_rgba8=ref.data_ptr(), stream=streams[0].cuda_stream)
    torch.cuda.synchronize()
    bad = 0
    for i in range(n):
        k = i % S
        with torch.cuda.stream(streams[k]):
            if i >= S:
                bad += int(not torch.equal(bufs[k], ref))          # checks frame i - S (synchronises with its stream)
            bufs[k].zero_()
            sr.render_rect_device(scene, w, h, 0.0, ms, (0, 0, w, h), d_rgba8=bufs[k].data_ptr(), stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
    bad += sum(int(not torch.equal(b, ref)) for b in bufs)
    print("scene %d %dx%d: %d frames on %d streams, %d differ" % (scene, w, h, n, S, bad), flush=True)
sr.close()
