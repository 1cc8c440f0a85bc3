import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, "/root/repo")
import torch, rmdf_amd
dev = torch.device("cuda", 0)
sr = rmdf_amd.ShaderRenderer(0); sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
w, h, ms = 1920, 1080, 256
for n, costed in ((4, False), (8, False), (4, True), (8, True)):
    sr.set_shard_costs(sr.probe_tile_costs(2, w, h, 0.0, ms) if costed else None)
    slots = rmdf_amd.shard_slots(n); S = 8
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    bufs = [torch.zeros((slots, h // 8, w // 8), dtype=torch.int32, device=dev) for _ in range(S)]
    res = []
    for r in range(n):
        for i in range(32): sr.render_shard_device(2, w, h, 0.0, ms, r, n, bufs[i % S].data_ptr(), stream=streams[i % S].cuda_stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(400): sr.render_shard_device(2, w, h, 0.0, ms, r, n, bufs[i % S].data_ptr(), stream=streams[i % S].cuda_stream)
        torch.cuda.synchronize(); res.append(round((time.perf_counter() - t0) / 400 * 1e3, 4))
    print(n, "LPT" if costed else "static", res, "sum", round(sum(res), 3), "max/mean", round(max(res) * n / sum(res), 3))
