#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X sphere tracer.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): Mpixels/s, Mandelbulb power-8, 1920x1080, 256 max march steps, uffizi_512.hdr
environment, in_time = 0.  One "step" = one full frame of the hot path (ray generation, bounding-sphere
clip, march loop, finite-difference normal, distance AO, prefiltered-env-map shading, gamma, RGBA8 pack)
with the cube maps already resident in HBM; the frame stays in HBM (the PCIe-inclusive rate is reported
separately as `d2h_inclusive_mpixels_s`, never as `value`).

N > 1 (launched by torch.distributed.run, one rank per GPU): the reference's 64 tiles are dealt to the ranks
interleaved (tile idx mod N), each rank renders its shard, ONE gather over RCCL/xGMI brings the shards to
rank 0, which scatters them to frame positions.  Per-GPU work shrinks as N grows: "scaling": "strong".

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (k_render): algorithmic HBM bytes per
launch / its average duration measured with HIP events on the launch stream.  The path is FP32-VALU bound by
construction, so `roofline.frac` is expected to be ~1 %; the VALU-side figure is reported next
to it in `valu_roofline`.  `cpu_baseline` = the CPU oracle (a port of the reference's shader to C) on the host
cores of this box, reported, never the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# frames in flight live on separate HIP streams; give the runtime enough hardware queues for them (read at HIP init).
# Measured with 8 frames in flight (one GPU standing in for a rank of 8): 0.109 ms per frame with 8 queues -- streams share
# queues -- 0.069 ms with 12 or more.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_TLANEOPS = 78.6       # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (157.3 TFLOPS only if every op were an FMA)


# ---- helpers shared with the CPU-tier tests --------------------------------------------------------

def load_oracle_env(orc):
    """Oracle-built cube maps of the shipped uffizi_512 + cache files (checker side only)."""
    import rmdf_amd
    d = os.path.join(rmdf_amd.DATA_DIR, "latlong_envmaps")
    rd = lambda n: orc.hdr_decode(open(os.path.join(d, n), "rb").read())
    return orc.EnvSet.from_latlongs(rd("uffizi_512.hdr"), rd("uffizi_512_cache_pow_1.0.hdr"),
                                    rd("uffizi_512_cache_pow_8.0.hdr"))


def load_oracle_faces(orc):
    """Oracle-built float32 cube faces of the same files (input shared by both sides in --check)."""
    import rmdf_amd
    d = os.path.join(rmdf_amd.DATA_DIR, "latlong_envmaps")
    rd = lambda n: orc.hdr_decode(open(os.path.join(d, n), "rb").read())
    return {"refl": orc.latlong_to_cube(rd("uffizi_512.hdr")), "cos1": orc.latlong_to_cube(rd("uffizi_512_cache_pow_1.0.hdr")),
            "cos8": orc.latlong_to_cube(rd("uffizi_512_cache_pow_8.0.hdr"))}


def gather_shards(shard, rank, world, dist, out=None, out_list=None):
    """The single exchange step of the path: gather every rank's packed tile shard on rank 0.
    shard: (slots, th, tw) int32 tensor.  Returns (world, slots, th, tw) on rank 0, None elsewhere.
    out_list = list(out.unbind(0)), precomputed by callers that gather every frame."""
    import torch
    if not (dist.is_available() and dist.is_initialized()):
        return shard.unsqueeze(0)
    if shard.is_cuda and dist.get_backend() == "gloo":
        # test aid (RMDF_BENCH_SHARE_GPU): gloo gathers host tensors; stage through the host, synchronously
        host = shard.cpu()
        if rank == 0:
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.gather(host, parts, dst=0)
            if out is None:
                out = torch.empty((world,) + tuple(shard.shape), dtype=shard.dtype, device=shard.device)
            out.copy_(torch.stack(parts))
            return out
        dist.gather(host, None, dst=0)
        return None
    if rank == 0:
        if out is None:
            out = torch.empty((world,) + tuple(shard.shape), dtype=shard.dtype, device=shard.device)
            out_list = None
        dist.gather(shard, out_list if out_list is not None else list(out.unbind(0)), dst=0)
        return out
    dist.gather(shard, None, dst=0)
    return None


def ranks_agree_on_deal(deal, dist, device):
    """deal = [tiles of rank 0, tiles of rank 1, ...] as THIS rank computed it.  True iff every rank holds rank 0's deal
    (one broadcast + one all-reduce, outside the timed region)."""
    import torch
    mine = torch.tensor([t for tiles in deal for t in (list(tiles) + [-1] * 64)[:64]], dtype=torch.int32, device=device)
    ref0 = mine.clone()
    dist.broadcast(ref0, src=0)
    agree = torch.tensor([1 if bool((ref0 == mine).all()) else 0], dtype=torch.int32, device=device)
    dist.all_reduce(agree, op=dist.ReduceOp.MIN)
    return int(agree.item()) == 1


def max_over_ranks(seconds, dist, device):
    import torch
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def flops_model(c):
    """F = 79 I + 11 E + 9 S + 150 H + 30 P  (as-written IEEE op counts)."""
    return 79 * c["triplex_iters"] + 11 * c["de_evals"] + 9 * c["march_steps"] + 150 * c["hit_pixels"] + 30 * c["pixels"]


# ---- the benchmark ----------------------------------------------------------------------------------

def main():
    # stdout carries exactly ONE line, the JSON result: everything else that writes to fd 1 (RCCL's version banner,
    # library chatter) is sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--max-steps", type=int, default=256)
    ap.add_argument("--time", type=float, default=0.0)
    ap.add_argument("--scene", type=int, default=2, help="FragmentShader enum (2 = FSMBPower8Shader)")
    ap.add_argument("--supersample", type=int, default=0, help="mip levels of super-sampling: rays = (w<<L) x (h<<L), "
                    "resolved on the GPU before the gather (BASELINE config 4: --width 3840 --height 2160 --supersample 1)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("RMDF_BENCH_STREAMS", "0")),
                    help="frames kept in flight (one HIP stream + buffer set each); 1 = one frame at a time; "
                         "0 = default: 2 on one GPU, 8 on N GPUs")
    ap.add_argument("--animate", type=float, default=0.0, help="advance in_time by this many seconds per frame (the viewer's "
                    "animation: the cost-ordered dispatch then works from the previous frame's costs of a slightly different view)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="also compare the frame with the oracle (slow)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import rmdf_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    # RMDF_BENCH_SHARE_GPU=1 (test aid for a 1-GPU box): every rank uses cuda:0 and the exchange goes over gloo through
    # host staging -- RCCL cannot put two ranks on one device.  It exercises the multi-rank logic (probe determinism across
    # processes, deal agreement, per-rank shards, assembly), not the transport; the numbers it prints mean nothing.
    share_gpu = os.environ.get("RMDF_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if share_gpu else dev          # where the small control-plane tensors live
    # the N > 1 path (shard render, RCCL gather, assemble); RMDF_BENCH_FORCE_DIST=1 runs it with world size 1 so that a
    # 1-GPU box can smoke-test it
    sharded = world > 1 or os.environ.get("RMDF_BENCH_FORCE_DIST") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # one rank builds (a no-op when librmdf.so is current), the others wait: N ranks must not run make at once
    if local_rank == 0:
        rmdf_amd.build()
    if sharded:
        dist.barrier()
    w, h, ms, scene = a.width, a.height, a.max_steps, a.scene
    sr = rmdf_amd.ShaderRenderer(local_rank, flags=int(os.environ.get("RMDF_FLAGS", "0")))
    if a.check:
        # strict comparison: both sides get the oracle-built float32 cube faces (the device's own latlong -> cube uses the
        # device libm where the oracle uses glibc: a few texels differ by one f16 ulp, see tests/test_gpu_parity.py)
        from oracle import orc as _orc
        _env = load_oracle_faces(_orc)
        for slot, k in ((rmdf_amd.ENV_REFLECTION, "refl"), (rmdf_amd.ENV_COS_1, "cos1"), (rmdf_amd.ENV_COS_8, "cos8")):
            sr.set_env_cube(slot, _env[k])
    else:
        sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
    dev_name, cus = sr.device_info()
    deal = "single GPU"
    if sharded:
        # cost-aware deal of the 64 tiles: every rank probes the view at 256 x 144 on its own GPU (bit-reproducible
        # kernels -> identical costs, no exchange) and deals longest-processing-time-first.  Outside the timed region
        # the ranks compare their deals once; any disagreement falls back to the static deal on all of them.
        sr.set_shard_costs(sr.probe_tile_costs(a.scene, a.width << a.supersample, a.height << a.supersample, a.time, a.max_steps))
        # rank 0 also receives 63/64 of every frame and assembles it: about 10 us per frame next to 68 us of render at N = 8
        # (measured on one GPU: the N > 1 path with world size 1 against the plain path), less in proportion at smaller N.
        # The deal therefore starts rank 0 with that share of load.  RMDF_ROOT_HANDICAP overrides (a fraction of a rank's share).
        handicap = float(os.environ.get("RMDF_ROOT_HANDICAP", min(0.25, 0.015 * world)))
        sr.set_shard_root_handicap(handicap)
        agree = ranks_agree_on_deal([sr.shard_tiles(r, world) for r in range(world)], dist, cdev)
        if agree:
            deal = "cost-aware (probe frame, LPT, rank 0 handicap %.3f)" % handicap
        else:
            sr.set_shard_costs(None)
            sr.set_shard_root_handicap(0.0)
            deal = "static (ranks disagreed on the probed costs)"
    # Frames are independent, so S of them are kept in flight: frame i goes to HIP stream i % S (dedicated, non-null
    # streams; kernels, the RCCL call and the timing events of a frame all go on its stream) and owns buffer set
    # i % S.  On one GPU this overlaps the thin tail of a frame -- the launch cannot end before its longest ray has
    # finished a ~0.45 ms serial chain -- with the bulk of the next; on N GPUs it also overlaps the gather of frame i
    # with the render of frame i+1.  --streams 1 = strictly one frame at a time.
    S = a.streams if a.streams > 0 else (2 if not sharded else 8)
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    stream = streams[0]
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream
    assert all(st.cuda_stream != 0 for st in streams)

    L = a.supersample
    rw, rh = w << L, h << L                                   # ray grid
    i32 = dict(dtype=torch.int32, device=dev)
    frames = [torch.empty((h, w), **i32) for _ in range(S)]
    if not sharded:
        bigs = [torch.empty((rh, rw), **i32) for _ in range(S)] if L else frames
        tmps = [torch.empty((rh // 2, rw // 2), **i32) if L > 1 else None for _ in range(S)]
    else:
        slots = rmdf_amd.shard_slots(world)
        gathereds = [torch.zeros((world, slots, h // 8, w // 8), **i32) if rank == 0 else None for _ in range(S)]
        gather_lists = [list(g.unbind(0)) if g is not None else None for g in gathereds]
        # rank 0 renders straight into its own slot of the gather buffer (the gather's copy of the root's part is then a
        # copy onto itself)
        shards = [gathereds[k][0] if rank == 0 else torch.zeros((slots, h // 8, w // 8), **i32) for k in range(S)]
        bigs = [torch.zeros((slots, rh // 8, rw // 8), **i32) for _ in range(S)] if L else shards
        tmps = [torch.empty((slots, rh // 16, rw // 16), **i32) if L > 1 else None for _ in range(S)]
    frame = frames[0]

    def resolve(src, sw, sh, dst, tmp, sp):
        """`L` box-filter levels from src (sw x sh) into dst, ping-ponging through tmp"""
        cur, cw, ch = src, sw, sh
        for lvl in range(L):
            out = dst if lvl == L - 1 else (tmp if cur is not tmp else src)
            sr.resolve_box2_device(cur.data_ptr(), cw, ch, out.data_ptr(), stream=sp)
            cur, cw, ch = out, cw // 2, ch // 2

    def render_only(k=0, t=a.time):
        sp = streams[k].cuda_stream
        if not sharded:
            sr.render_rect_device(scene, rw, rh, t, ms, (0, 0, rw, rh), d_rgba8=bigs[k].data_ptr(), stream=sp)
        else:
            sr.render_shard_device(scene, rw, rh, t, ms, rank, world, bigs[k].data_ptr(), stream=sp)

    def step(i=0):
        k = i % S
        sp = streams[k].cuda_stream
        with torch.cuda.stream(streams[k]):                   # the RCCL call orders itself against the current stream
            render_only(k, a.time + a.animate * i)
            if not sharded:
                if L:
                    resolve(bigs[k], rw, rh, frames[k], tmps[k], sp)
            else:
                if L:
                    resolve(bigs[k], rw // 8, slots * (rh // 8), shards[k], tmps[k], sp)
                g = gather_shards(shards[k], rank, world, dist, out=gathereds[k], out_list=gather_lists[k])
                if rank == 0:
                    sr.assemble_shards_device(w, h, world, g.data_ptr(), frames[k].data_ptr(), stream=sp)

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(a.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    host_enqueue_ms = (time.perf_counter() - t0) / a.steps * 1e3      # host time to issue one step (this rank)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, dist, cdev)
    ms_per_step = dt / a.steps * 1e3
    mpix = w * h / 1e6
    value = mpix / (dt / a.steps)

    # dominant kernel alone: HIP events on the launch stream around each launch (this rank's share)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(a.steps, 100))]
    for e0, e1 in evs:
        e0.record(stream)
        render_only()
        e1.record(stream)
    torch.cuda.synchronize(dev)
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))
    kern_ms_min = float(np.min([e0.elapsed_time(e1) for e0, e1 in evs]))

    result = None
    if rank == 0:
        # PCIe-inclusive rate (host buffer hand-over as the boundary does it) -- informational
        host = np.empty(w * h, np.uint32)
        if L == 0:
            sr.draw_shader_tile(scene, None, w, h, a.time, host, max_steps=ms)
        else:
            sr.render_supersampled(scene, w, h, L, a.time, max_steps=ms)
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            if L == 0:
                sr.draw_shader_tile(scene, None, w, h, a.time, host, max_steps=ms)
            else:
                sr.render_supersampled(scene, w, h, L, a.time, max_steps=ms)
        d2h_rate = mpix / ((time.perf_counter() - t1) / reps)
        d2h_registered = None
        if L == 0:
            # the same hand-over into a buffer the caller registered once (rmdf_register_host_buffer): the render kernel
            # writes it over PCIe while it renders
            sr.register_host_buffer(host)
            sr.draw_shader_tile(scene, None, w, h, a.time, host, max_steps=ms)
            t1 = time.perf_counter()
            for _ in range(reps):
                sr.draw_shader_tile(scene, None, w, h, a.time, host, max_steps=ms)
            d2h_registered = mpix / ((time.perf_counter() - t1) / reps)
            sr.unregister_host_buffer(host)

        env_bytes = 6 * 172 * 172 * 8 + 2 * 6 * 87 * 87 * 8               # padded RGB16F cube maps read once
        px_this_launch = rw * rh if not sharded else len(sr.shard_tiles(rank, world)) * (rw // 8) * (rh // 8)
        algo_bytes = px_this_launch * 4 + env_bytes                        # RGBA8 store + env read
        achieved_gbs = algo_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        valu_instr = None           # SQ_INSTS_VALU per launch, from the committed PMC pass (profiles/pmc_traffic.json)
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj) and not sharded and L == 0:
            try:
                t = json.load(open(tj))
                if t.get("workload") == [scene, w, h, ms]:
                    traffic = t.get("hbm_bytes_per_launch")
                    valu_instr = (t.get("valu") or {}).get("SQ_INSTS_VALU_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "Mpixels/s, Mandelbulb power-8 1920x1080 @256 steps; 1/2/4/8 GPU",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "FragmentShader %d (2 = FSMBPower8Shader) %dx%d%s, max_steps %d, in_time %.1f, uffizi_512.hdr env, "
                                   "full frame -> RGBA8 resident in HBM" % (scene, w, h, (" x %d rays/px, box-resolved on the GPU" % (4 ** L)) if L else "", ms, a.time),
                       "supersample_levels": L, "mrays_per_s": round(value * 4 ** L, 2),
                       "scene": scene, "width": w, "height": h, "max_steps": ms,
                       "parallelism": ("1 GPU, one launch per frame" if not sharded else
                                       "64 tiles dealt to %d %s + one %s gather per frame" %
                                       (world, "ranks sharing one GPU (test aid)" if share_gpu else "GPUs", "gloo (host-staged)" if share_gpu else "RCCL")) +
                                      ", %d frame(s) in flight" % S,
                       "frames_in_flight": S, "tile_deal": deal, "animate_dt": a.animate,
                       "device": dev_name, "compute_units": cus},
            "roofline": {"bound": "hbm", "kernel": "k_render<2>", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms_avg": round(kern_ms, 4),
                         "kernel_ms_min": round(kern_ms_min, 4),
                         "note": "VALU-bound path (SURVEY 8d): HBM fraction is a sanity figure, see valu_roofline"},
            "d2h_inclusive_mpixels_s": round(d2h_rate, 2),
            "d2h_inclusive_registered_buffer_mpixels_s": None if d2h_registered is None else round(d2h_registered, 2), "host_enqueue_ms_per_step_rank0": round(host_enqueue_ms, 4),
        }
        if ((not a.no_cpu_baseline and world == 1) or a.check) and L == 0:      # CPU baselines: rank 0 at N = 1 only
            from oracle import orc
            env = load_oracle_env(orc)
            cores = orc.num_processors()
            tc = time.perf_counter()
            ref = orc.render(scene, w, h, a.time, ms, env, nthreads=cores, want_f32=False)
            cpu_dt = time.perf_counter() - tc
            ctr = ref["counters"]
            F = flops_model(ctr)
            if not sharded:
                result["valu_roofline"] = {"achieved": round(F / (kern_ms * 1e-3) / 1e12, 3), "peak": VALU_PEAK_TLANEOPS,
                                           "unit": "T lane-ops/s (as-written IEEE ops, no FMA contraction)",
                                           "frac": round(F / (kern_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS, 4),
                                           "flop_per_frame": F, "counters": ctr}
                if valu_instr:
                    # what actually bounds the kernel: VALU instruction issue.  Peak = the rate a stream of simple VALU
                    # instructions sustains on this chip (tools/ubench/valu_rates: 0.93 G wave-instructions/s/SIMD)
                    simds = cus * 4
                    rate = valu_instr / simds / (kern_ms * 1e-3) / 1e9
                    result["valu_roofline"]["issue"] = {"valu_instructions_per_launch_pmc": valu_instr, "simds": simds,
                                                        "achieved": round(rate, 3), "peak": 0.93, "frac": round(rate / 0.93, 3),
                                                        "unit": "G wave-instructions/s/SIMD"}
            result["cpu_baseline"] = {"value": round(mpix / cpu_dt, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
                                      "sample": "1 full frame %dx%d of the same workload, CPU oracle (C port of fragment.shd), "
                                                "row segments over all host cores as ConcurrentSegments does" % (w, h)}
            # the reference's own CPU paths (BASELINE.json north_star: "timed on the same box's host cores ... as the
            # reported, non-target baseline"), as their C restatements with the reference's threading model
            # (SURVEY.md 8d): Fractal2D.juliaAnimated over all cores in row segments; cosineConvolveHDREnvMap with one
            # thread per power (ShaderRendering.hs:142) -- one power timed on one thread -- and split over all cores
            d = os.path.join(rmdf_amd.DATA_DIR, "latlong_envmaps")
            small = orc.resize_hdr(orc.hdr_decode(open(os.path.join(d, "uffizi_512.hdr"), "rb").read()), 256)
            tj0 = time.perf_counter(); orc.julia_animated(512, 512, 0, 0.0); tj = time.perf_counter() - tj0
            tc0 = time.perf_counter(); orc.cosine_convolve(small, 8.0, nthreads=1); tc1 = time.perf_counter() - tc0
            tc0 = time.perf_counter(); orc.cosine_convolve(small, 8.0, nthreads=0); tca = time.perf_counter() - tc0
            tg0 = time.perf_counter(); sr.prefilter_env(small, 8.0); tg = time.perf_counter() - tg0
            result["cpu_reference_paths"] = {
                "cores": cores, "kind": "port",
                "julia_animated_512x512_ms": round(tj * 1e3, 2), "julia_animated_mpixels_s": round(0.262144 / tj, 1),
                "cosine_convolve_256x128_power8_one_thread_s": round(tc1, 3),
                "cosine_convolve_256x128_power8_all_cores_s": round(tca, 3),
                "gpu_prefilter_256x128_power8_ms_host_in_out": round(tg * 1e3, 3)}
            if a.check:
                result["check_rgba8_equal"] = all(bool(np.array_equal(f.cpu().numpy().view(np.uint32), ref["rgba8"]))
                                                  for f in frames)
        os.write(json_fd, (json.dumps(result) + "\n").encode())

    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    sr.close()
    return result


if __name__ == "__main__":
    main()
