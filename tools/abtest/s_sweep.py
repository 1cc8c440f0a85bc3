import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]
sys.path.insert(0, "/root/repo")
import torch, rmdf_amd
dev = torch.device("cuda", 0)
sr = rmdf_amd.ShaderRenderer(0); sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
w, h, ms, n = 1920, 1080, 256, 8
sr.set_shard_costs(sr.probe_tile_costs(2, w, h, 0.0, ms))
slots = rmdf_amd.shard_slots(n)
for S in (8,):
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    bufs = [torch.zeros((slots, h // 8, w // 8), dtype=torch.int32, device=dev) for _ in range(S)]
    r = 2
    for i in range(4 * S): sr.render_shard_device(2, w, h, 0.0, ms, r, n, bufs[i % S].data_ptr(), stream=streams[i % S].cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(800): sr.render_shard_device(2, w, h, 0.0, ms, r, n, bufs[i % S].data_ptr(), stream=streams[i % S].cuda_stream)
    torch.cuda.synchronize()
    print("HWQ", sys.argv[1], "S", S, round((time.perf_counter() - t0) / 800 * 1e3, 4), flush=True)
