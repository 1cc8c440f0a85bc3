import sys, time
sys.path.insert(0, "/root/repo")
import torch, rmdf_amd
dev = torch.device("cuda", 0)
sr = rmdf_amd.ShaderRenderer(0)
st = torch.cuda.Stream(dev); sp = st.cuda_stream
for (sw, sh) in ((7680, 4320), (3840, 2160)):
    src = torch.randint(0, 2**31 - 1, (sh, sw), dtype=torch.int32, device=dev)
    dst = torch.empty((sh // 2, sw // 2), dtype=torch.int32, device=dev)
    for _ in range(5): sr.resolve_box2_device(src.data_ptr(), sw, sh, dst.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(50): sr.resolve_box2_device(src.data_ptr(), sw, sh, dst.data_ptr(), stream=sp)
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("%dx%d -> resolve %.4f ms, %.0f GB/s (algorithmic 5 B per source pixel / 20 B per output pixel)" % (sw, sh, ms, sw * sh * 5 / ms / 1e6))
# assemble
w, h, n = 1920, 1080, 8
g = torch.zeros((n, 8, h // 8, w // 8), dtype=torch.int32, device=dev); f = torch.zeros((h, w), dtype=torch.int32, device=dev)
for _ in range(5): sr.assemble_shards_device(w, h, n, g.data_ptr(), f.data_ptr(), stream=sp)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(50): sr.assemble_shards_device(w, h, n, g.data_ptr(), f.data_ptr(), stream=sp)
e1.record(st); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print("assemble 1920x1080: %.4f ms, %.0f GB/s" % (ms, w * h * 8 / ms / 1e6))
