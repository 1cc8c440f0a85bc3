#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X sphere tracer.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): Mpixels/s, Mandelbulb power-8, 1920x1080, 256 max march steps, uffizi_512.hdr
environment, in_time = 0.  One "step" = one full frame of the hot path (ray generation, bounding-sphere
clip, march loop, finite-difference normal, distance AO, prefiltered-env-map shading, gamma, RGBA8 pack)
with the cube maps already resident in HBM -- built by the product's own env pipeline (rmdf_load_env_hdr: GPU resize,
GPU lobe prefilter, RGBE cache files, GPU cube conversion) before the timed region; the frame stays in HBM (the
PCIe-inclusive rate is reported separately as `d2h_inclusive_mpixels_s`, never as `value`).

Timing: after the W warm-up steps the run keeps stepping until at least 0.3 s have passed (clocks ramp up on a fresh box),
then times `repeats` (3) blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both sides with the MAX over
ranks taken; `ms_per_step` / `value` are those of the MEDIAN block, the others are listed in `ms_per_step_blocks`.

N > 1 (launched by torch.distributed.run, one rank per GPU): the reference's 64 tiles are dealt to the ranks by measured
cost (a probe frame rendered on every rank, longest-processing-time-first, rank 0 handicapped by its measured receive +
assemble time), each rank renders its shard, ONE gather over RCCL/xGMI -- issued by the library itself
(rmdf_render_frame_sharded_device: grouped ncclRecv fan-in on rank 0; torch.distributed only carries the 128-byte
unique id and the control-plane reductions) -- brings the shards to rank 0, which scatters them to frame positions.
Per-GPU work shrinks as N grows: "scaling": "strong".

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (k_render) against the roof that binds it, FP32
vector-ALU issue: as-written IEEE operations of the frame (counted by the instrumented oracle) / the
kernel's average duration, measured with HIP events on the launch stream, against 78.6 T lane-ops/s.  `hbm_roofline` is the
same kernel's algorithmic HBM bytes against 8 TB/s (the north star asks for it; ~0.3 % by construction).  `cpu_baseline` =
the CPU oracle (a port of the reference's shader to C) on the host cores of this box, reported, never the target.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# frames in flight live on separate HIP streams and need more than the runtime's 4 hardware queues (read at HIP init).
# rmdf_create sets this too, but torch initialises HIP first in this process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_TLANEOPS = 78.6       # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (157.3 TFLOPS only if every op were an FMA)
VALU_ISSUE_PEAK = 1.0           # G wave-instructions/s/SIMD a pure v_mul stream sustains after the clock has ramped (tools/ubench/valu_rates,
                                # "sustained": 0.98-1.03 at 2.3-2.4 GHz = one per 2.2-2.4 cycles; 0.93-0.97 in a 1 ms launch from idle at ~2.0 GHz)
VALU_ISSUE_SPEC = 1.2           # the guide's figure: one wave64 instruction per SIMD-32 every 2 cycles, at 2.4 GHz (MI355X_MICROARCH.md)
MIN_WARM_SECONDS = float(os.environ.get("RMDF_BENCH_MIN_WARM", "0.3"))   # profiling runs shorten it

# as-written operation counters of the headline frame (scene 2, 1920x1080, in_time 0, 256 steps), counted by the
# instrumented oracle (tests/golden/full_size_digests.json holds the same numbers; `--check` or the cpu_baseline leg
# recount them live).  They depend on the geometry only, not on the environment map.
HEADLINE_COUNTERS = {"de_evals": 39739969, "triplex_iters": 128870630, "march_steps": 32356849, "hit_pixels": 1230520,
                     "sphere_pixels": 1534400, "pixels": 2073600}


# ---- helpers shared with the CPU-tier tests --------------------------------------------------------

_ORACLE_ENV = {}


def oracle_env_latlongs(orc):
    """The ORACLE's env pipeline on the shipped uffizi_512.hdr (checker side only): reflection map + lobe maps 1 and 8 after
    resize, prefilter and the RGBE round trip.  Cached per process."""
    if "lat" not in _ORACLE_ENV:
        import rmdf_amd
        _ORACLE_ENV["lat"] = orc.env_pipeline(open(rmdf_amd.DEFAULT_ENV_HDR, "rb").read(), powers=(1.0, 8.0))[0]
    return _ORACLE_ENV["lat"]


def load_oracle_env(orc):
    """Oracle-built cube maps of the shipped uffizi_512 (checker side only)."""
    if "env" not in _ORACLE_ENV:
        lat = oracle_env_latlongs(orc)
        _ORACLE_ENV["env"] = orc.EnvSet.from_latlongs(lat["refl"], lat["cos1"], lat["cos8"])
    return _ORACLE_ENV["env"]


def gather_shards(shard, rank, world, dist, out=None, out_list=None):
    """The exchange step over torch.distributed (fallback of the library's own RCCL gather, and the gloo transport of the
    CPU-tier tests): gather every rank's packed tile shard on rank 0.
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


class Watchdog:
    """A collective that never completes (a peer that died, a mismatched send / receive) would leave the driver waiting for its own
    timeout with nothing on stderr.  arm(what) starts a countdown, disarm() stops it; when it runs out the thread says which rank
    was stuck in what and ends the process with exit code 3 (os._exit: no re-exec, no atexit handlers that could block again)."""

    def __init__(self, rank, seconds):
        import threading
        self.rank, self.seconds, self.what, self.deadline = rank, seconds, None, None
        self.lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, what, seconds=None, on_timeout=None):
        """on_timeout (optional): called instead of the exit when the countdown runs out -- for legs behind the timed region whose loss must
        not cost the line (it should write the line and end the process itself)"""
        with self.lock:
            self.what, self.deadline, self.on_timeout = what, time.monotonic() + (seconds or self.seconds), on_timeout

    def disarm(self):
        with self.lock:
            self.what, self.deadline, self.on_timeout = None, None, None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                what, dl, cb = self.what, self.deadline, getattr(self, "on_timeout", None)
            if dl is not None and time.monotonic() > dl and cb is not None:
                sys.stderr.write("bench.py watchdog: rank %d has been stuck in %r -- the line goes out without that leg\n" % (self.rank, what))
                sys.stderr.flush()
                cb()
                os._exit(0)
            if dl is not None and time.monotonic() > dl:
                sys.stderr.write("bench.py watchdog: rank %d has been stuck in %r for more than %.0f s -- giving up (exit 3)\n" % (self.rank, what, self.seconds))
                sys.stderr.flush()
                os._exit(3)


def flops_model(c):
    """F = 79 I + 11 E + 9 S + 150 H + 30 P  (as-written IEEE op counts)."""
    return 79 * c["triplex_iters"] + 11 * c["de_evals"] + 9 * c["march_steps"] + 150 * c["hit_pixels"] + 30 * c["pixels"]


# As-written operation counts of the other three FragmentShader values, in SURVEY.md 8(d)'s style (every IEEE operation 1, and sqrt,
# inversesqrt, /, pow, log, exp, sin, cos, acos, atan each 1 as well; min / max / abs / compare 1, clamp 2; texel fetches 0).
#   Cornell box (fragment.shd:312-411): compute_barycentric = 3 vector differences (9) + 5 dot products (25) + inv_denom (4) + u, v (8) +
#     the prism test (4) = 50; prism branch: point on plane (17) + distance (9) = 26 -> 76 per triangle; edge branch: 3 x
#     line_seg_min_dist_sq (36: ab 3, len_sq 5, p - a 3, dot 5, / 1, clamp 2, a + t ab 6, two p - proj 6, dot 5) + 2 min + sqrt = 111
#     -> 161 per triangle; + 32 min per estimate.  T_in (estimate, triangle) pairs take the prism branch (oracle counter tri_inside), the
#     other 32 E - T_in the edge branch.  Hit pixel: distance_ao has four taps instead of two -> 170 instead of 150.
#   DE test scene (fragment.shd:21-33, 413-456): de_sphere 7 + 3 x de_torus (10) + 3 x de_rounded_box (16) + 5 x smin (2 mul, 2 exp,
#     add, log, neg, / = 8) + 1 min = 126 per estimate.
#   General-power Mandelbulb (fragment.shd:42-72, 101-158): one full iteration = length 6 + compare 1 + triplex_pow 22
#     (cartesian_to_spherical: length 6, / 1, acos 1, atan 1; pow 1; two angle products 2; spherical_to_cartesian: 3 sin, 2 cos, 5 mul)
#     + w += pos 3 + the running derivative 5 (power - 1, pow, 2 mul, add) = 37; breaking iteration 7 + estimate tail 4 = 11 per estimate.
# Counters I, E, S, H, P, T_in: the instrumented oracle's, committed with the full-size digests (tests/golden/full_size_digests.json).
# NOTE what the fraction means for these scenes: a transcendental priced at ONE operation costs this implementation (and any other
# without hardware acos / atan / sincos / pow at full precision) dozens of instructions, so scenes 1 and 3 sit far below the roof by
# construction of the count; and the Cornell box's count is the REFERENCE's work -- all 32 triangles per estimate -- of which the pruned
# estimate evaluates ~1.1 (bit-identical result), so its fraction exceeds 1.  `issue_frac_of_spec_peak` (instructions actually issued,
# from the committed PMC pass) is the figure that says how busy the vector pipes are.
def secondary_ops(scene, c):
    if scene == 0:
        t_in = c["tri_inside"]
        return 76 * t_in + 161 * (32 * c["de_evals"] - t_in) + 32 * c["de_evals"] + 9 * c["march_steps"] + 170 * c["hit_pixels"] + 30 * c["pixels"]
    if scene == 1:
        return 126 * c["de_evals"] + 9 * c["march_steps"] + 150 * c["hit_pixels"] + 30 * c["pixels"]
    if scene == 3:
        return 37 * c["triplex_iters"] + 11 * c["de_evals"] + 9 * c["march_steps"] + 150 * c["hit_pixels"] + 30 * c["pixels"]
    return flops_model(c)


SECONDARY_FORMULA = {0: "76 T_in + 161 (32 E - T_in) + 32 E + 9 S + 170 H + 30 P", 1: "126 E + 9 S + 150 H + 30 P",
                     3: "37 I + 11 E + 9 S + 150 H + 30 P", 2: "79 I + 11 E + 9 S + 150 H + 30 P"}


def secondary_roofline(digest_name, scene, kernel_ms, valu_instr, cus):
    """roofline object of a secondary scene: as-written operations of the committed view / measured kernel time against 78.6 T lane-ops/s"""
    try:
        c = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size_digests.json")))[digest_name]["counters"]
    except Exception as e:                                      # noqa: BLE001
        return {"bound": "valu", "achieved": None, "frac": None, "note": "no committed counters for %s: %s" % (digest_name, e)}
    F = secondary_ops(scene, c)
    ach = F / (kernel_ms * 1e-3) / 1e12
    r = {"bound": "valu", "achieved": round(ach, 3), "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s", "frac": round(ach / VALU_PEAK_TLANEOPS, 4),
         "ops_per_launch": F, "formula": SECONDARY_FORMULA[scene], "op_counters": c,
         "op_counters_source": "instrumented oracle, committed with the digests (tests/golden/full_size_digests.json: %s)" % digest_name,
         "traffic": None}
    if valu_instr:
        rate = valu_instr / (cus * 4) / (kernel_ms * 1e-3) / 1e9
        r["issue_g_wave_instr_s_simd"] = round(rate, 3)
        r["issue_frac_of_spec_peak"] = round(rate / VALU_ISSUE_SPEC, 3)
    if scene == 0:
        r["note"] = ("as-written count = the reference's 32 full triangle distances per estimate; the pruned per-lane estimate evaluates ~1.1 of "
                     "them (bit-identical minimum), so frac > 1 measures the algorithmic saving, not pipe utilisation: see issue_frac_of_spec_peak")
    elif scene in (1, 3):
        r["note"] = ("exp / log / pow / sin / cos / acos / atan priced at ONE operation each (SURVEY 8d's convention) while each costs dozens of "
                     "instructions at the pinned precision: frac is low by construction of the count; issue_frac_of_spec_peak says how busy the pipes are")
    return r


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


_PMC_OTHER_BUILD = {}       # the committed counter passes when they belong to ANOTHER build of the library (reported apart, labelled)


def achievable_issue_roof():
    """profiles/r06_march_loop_classes.json: the headline kernel's ISA priced in vector-port issue slots (tools/isa/march_loop_classes.py,
    CPU-only), or None"""
    fn = os.path.join(ROOT, "profiles", "r06_march_loop_classes.json")
    try:
        d = json.load(open(fn))
        return d if all(k in d for k in ("t_spec_ms", "t_sust_ms", "vslots", "pass_instr", "full_slots")) else None
    except Exception:                                           # noqa: BLE001
        return None


_LIVE_SCENE_PMC = {}       # scene name -> SQ_INSTS_VALU per launch, collected by this run (live_pmc_passes)


def live_pmc_passes(lib_path, scenes=True, budget_s=240.0):
    """The counter passes of tools/profile.sh + tools/prof_scene.sh, run NOW as child processes of this one -- `rocprofv3 --pmc <one group>
    -- python3 bench.py ...`, never combined with tracing, one pass at a time, after the timed region -- for a build of the library that
    profiles/ holds no passes of (round 5 ended without GPU access: the line the driver records would otherwise carry `traffic: null`).
    Returns (record like profiles/pmc_traffic.json, source text) or (None, why not).  The raw output stays under gpurun_out/prof_live/."""
    import shutil
    import subprocess
    rp = shutil.which("rocprofv3")
    if not rp:
        return None, "no rocprofv3 on PATH"
    out = os.path.join(ROOT, "gpurun_out", "prof_live")
    try:
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
    except OSError:
        import tempfile
        out = tempfile.mkdtemp(prefix="rmdf_prof_live_")
    env = dict(os.environ, RMDF_BENCH_PMC_CHILD="1", RMDF_BENCH_MIN_WARM="0", TMPDIR="/tmp")
    env.pop("RMDF_BENCH_MARK", None)
    me = os.environ.get("RMDF_BENCH_SELF", os.path.abspath(__file__))      # (the CPU tier's dry run names its own harness)
    common = ["--steps", "20", "--warmup", "2", "--repeats", "1", "--no-cpu-baseline", "--no-secondary", "--no-animated", "--streams", "1", "--pmc", "off"]
    t0, notes = time.monotonic(), []

    def one_pass(name, counters, extra):
        left = budget_s - (time.monotonic() - t0)
        if left < 20:
            notes.append("%s skipped (time budget)" % name)
            return False
        cmd = [rp, "--pmc"] + counters + ["--output-format", "csv", "-d", os.path.join(out, name), "--", sys.executable, me] + common + extra
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=min(left, 120))
            open(os.path.join(out, name + ".log"), "w").write(r.stderr[-4000:])
            if r.returncode != 0:
                notes.append("%s rc=%d" % (name, r.returncode))
            return r.returncode == 0
        except Exception as e:                                  # noqa: BLE001
            notes.append("%s: %s" % (name, e))
            return False

    ok = one_pass("pmc_fetch", ["FETCH_SIZE"], [])
    ok = one_pass("pmc_write", ["WRITE_SIZE"], []) and ok
    one_pass("pmc_sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"], [])
    if scenes:
        import csv
        import glob
        import statistics
        for name, sc in (("config2_cornell_1280x720_m128", 0), ("scene1_detest_1280x720_m128", 1), ("scene3_mbgeneral_1280x720_m128", 3)):
            d = "pmc_scene%d" % sc
            if not one_pass(d, ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
                            ["--scene", str(sc), "--width", "1280", "--height", "720", "--max-steps", "128"] + (["--time", {1: "2.5", 3: "3.0"}[sc]] if sc else [])):
                continue
            per = {}
            for f in glob.glob(os.path.join(out, d, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "k_render<%d" % sc in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
                        per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            if per:
                _LIVE_SCENE_PMC[name] = statistics.median(per.values())
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), out, "live"], capture_output=True, text=True, timeout=120)
        rec = json.load(open(os.path.join(out, "summary", "pmc_traffic.json")))
    except Exception as e:                                      # noqa: BLE001
        return None, "live counter passes: no summary (%s; %s)" % (e, "; ".join(notes))
    if rec.get("lib_sha256") != sha256_file(lib_path) or "hbm_bytes_per_launch" not in rec and "valu" not in rec:
        return None, "live counter passes: nothing usable (%s)" % "; ".join(notes)
    rec["command"] = "bench.py --pmc auto: child runs `rocprofv3 --pmc <one group per run> -- python3 bench.py --steps 20 --warmup 2 --streams 1 ...` after the timed region"
    return rec, ("collected LIVE by this run: child processes under rocprofv3 --pmc, one counter group per pass, after the timed region (%.0f s; librmdf.so sha256 %s...%s); "
                 "raw output in gpurun_out/prof_live/" % (time.monotonic() - t0, sha256_file(lib_path)[:12], ("; " + "; ".join(notes)) if notes else ""))


def pmc_record(lib_path, workload):
    """PMC figures of the dominant kernel from the committed counter passes (profiles/pmc_traffic.json, written by
    tools/pmc_summary.py on the GPU box).  They are measurements of ONE build: used only when the file names this very
    librmdf.so (sha256) and this workload, and labelled with where they come from."""
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tj):
        return None, "no profiles/pmc_traffic.json"
    try:
        t = json.load(open(tj))
    except Exception as e:                                      # noqa: BLE001
        return None, "unreadable profiles/pmc_traffic.json: %s" % e
    if t.get("workload") != workload:
        return None, "profiles/pmc_traffic.json is for workload %s" % t.get("workload")
    have = sha256_file(lib_path)
    if t.get("lib_sha256") != have:
        _PMC_OTHER_BUILD["record"] = t
        return None, "profiles/pmc_traffic.json was collected with another build of librmdf.so (%s..., this is %s...)" % (
            str(t.get("lib_sha256"))[:12], have[:12])
    return t, "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this build, librmdf.so sha256 %s...)" % have[:12]


# ---- the benchmark ----------------------------------------------------------------------------------

def self_launch(n_gpus, json_fd):
    """Start the N ranks of this benchmark as child processes (python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>) and relay the one JSON line rank 0 prints.  Returns the exit code: the launcher's, or 1 if no JSON line came back."""
    import socket
    import subprocess
    import torch                                            # device_count() does not initialise the GPU on this image
    have = torch.cuda.device_count()
    if have < n_gpus and os.environ.get("RMDF_BENCH_SHARE_GPU") != "1":
        sys.stderr.write("bench.py --gpus %d: this node shows %d GPU(s)\n" % (n_gpus, have))
        return 2
    env = dict(os.environ)
    env.pop("RMDF_BENCH_SELF_LAUNCH", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: the only kind this pool's host driver supports
    env.setdefault("OMP_NUM_THREADS", "1")
    rc, line = 1, None
    for attempt in range(4):
        # a free rendezvous port -- free NOW: somebody else may take it before the launcher's store binds it (seen once in 30 runs of
        # the test tier: EADDRINUSE), so a launch that dies of exactly that, before any rank has run, is repeated on another port
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write("bench.py: no launcher in the environment -- starting %d rank(s): %s\n" % (n_gpus, " ".join(cmd)))
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        import threading
        err_tail = []
        def pump(src=child.stderr):
            for raw in src:
                txt = raw.decode("utf-8", "replace")
                sys.stderr.write(txt)
                err_tail.append(txt)
                del err_tail[:-200]
        th = threading.Thread(target=pump, daemon=True)
        th.start()
        line = None
        for raw in child.stdout:
            txt = raw.decode("utf-8", "replace")
            if txt.lstrip().startswith("{") and line is None:
                try:
                    json.loads(txt)
                    line = txt
                    continue
                except ValueError:
                    pass
            sys.stderr.write(txt)                           # anything else a rank wrote to fd 1
        rc = child.wait()
        th.join(timeout=5)
        if rc != 0 and line is None and any("EADDRINUSE" in t or "address already in use" in t.lower() for t in err_tail):
            sys.stderr.write("bench.py: rendezvous port %d was taken in the meantime -- launching again on another\n" % port)
            continue
        break
    if line is not None and rc == 0:
        os.write(json_fd, line.encode())
        return 0
    sys.stderr.write("bench.py: the %d-rank run ended with exit code %d%s\n" % (n_gpus, rc, "" if line is not None else " and printed no JSON line"))
    return rc if rc != 0 else 1


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
    ap.add_argument("--repeats", type=int, default=3, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--max-steps", type=int, default=256)
    ap.add_argument("--time", type=float, default=0.0)
    ap.add_argument("--scene", type=int, default=2, help="FragmentShader enum (2 = FSMBPower8Shader)")
    ap.add_argument("--supersample", type=int, default=0, help="mip levels of super-sampling: rays = (w<<L) x (h<<L), "
                    "resolved on the GPU before the gather (BASELINE config 4: --width 3840 --height 2160 --supersample 1)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("RMDF_BENCH_STREAMS", "0")),
                    help="frames kept in flight (one HIP stream + buffer set each); 1 = one frame at a time; "
                         "0 = default: 3 on one GPU, 8 on N GPUs")
    ap.add_argument("--animate", type=float, default=0.0, help="advance in_time by this many seconds per frame (the viewer's "
                    "animation: the cost-ordered dispatch then works from the previous frame's costs of a slightly different view)")
    ap.add_argument("--no-animated", action="store_true", help="skip the extra animated block behind `value_animated` (profiling runs whose "
                    "per-kernel averages should cover the headline frame only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pmc", choices=("auto", "off"), default=os.environ.get("RMDF_BENCH_PMC", "auto"),
                    help="auto: when profiles/ holds no counter passes of THIS build of librmdf.so, collect them now -- child runs of this script "
                         "under `rocprofv3 --pmc <one group>` after the timed region (N = 1 only; about a minute); off: never")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (Cornell config 2, env prefilter config 5)")
    ap.add_argument("--check", action="store_true", help="also compare the frame with the oracle (slow)")
    a = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It starts `python -m torch.distributed.run`
    # with one rank per GPU as a CHILD (nothing here has touched the GPU, and nothing is exec'ed), relays rank 0's JSON line and
    # exits with the child's code.  RMDF_BENCH_SELF_LAUNCH=1 forces that path for --gpus 1 too (GPU-tier test on a 1-GPU box).
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or os.environ.get("RMDF_BENCH_SELF_LAUNCH") == "1"):
        raise SystemExit(self_launch(a.gpus, json_fd))

    import torch
    import torch.distributed as dist
    import rmdf_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    # RMDF_BENCH_SHARE_GPU=1 (test aid for a 1-GPU box): every rank uses cuda:0 and the exchange goes over gloo through
    # host staging -- RCCL cannot put two ranks on one device.  It exercises the multi-rank logic (probe determinism across
    # processes, deal agreement, per-rank shards, assembly), not the transport; the numbers it prints mean nothing.
    share_gpu = os.environ.get("RMDF_BENCH_SHARE_GPU") == "1"
    # ... and with RMDF_RCCL_LIB naming the test double of RCCL (tests/libfake_rccl.so, honoured by librmdf_xcheck.so only) the exchange
    # itself runs through the library's own calls -- rmdf_comm_init, the peers' ncclSend, the root's grouped ncclRecv, rmdf_comm_verify_deal,
    # S frames in flight on one communicator -- with N processes on this one GPU.  Readiness evidence, not a scaling number.
    rccl_double = share_gpu and bool(os.environ.get("RMDF_RCCL_LIB"))
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if share_gpu else dev          # where the small control-plane tensors live
    # the N > 1 path (shard render, RCCL gather, assemble); RMDF_BENCH_FORCE_DIST=1 runs it with world size 1 so that a
    # 1-GPU box can smoke-test it
    sharded = world > 1 or os.environ.get("RMDF_BENCH_FORCE_DIST") == "1"
    dog = Watchdog(rank, float(os.environ.get("RMDF_BENCH_WATCHDOG_S", "180")))
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # (the rendezvous also waits for ranks that are still importing torch on a cold box: minutes, not a hang)
        dog.arm("torch.distributed.init_process_group", seconds=max(dog.seconds, 600.0))
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dog.disarm()

    # one rank builds (a no-op when librmdf.so is current), the others wait: N ranks must not run make at once
    if local_rank == 0:
        rmdf_amd.build()
    if sharded:
        dist.barrier()
    w, h, ms, scene = a.width, a.height, a.max_steps, a.scene
    lib_path = os.environ.get("RMDF_LIB", rmdf_amd.LIB_PATH)
    sr = rmdf_amd.ShaderRenderer(local_rank, flags=int(os.environ.get("RMDF_FLAGS", "0")), xcheck=rccl_double)
    if rccl_double:
        lib_path = rmdf_amd.XCHECK_LIB_PATH
    # the product's own env pipeline (cache files are built on the GPU at first load; local rank 0 first, so that the
    # other ranks find complete files)
    t_env0 = time.perf_counter()
    if local_rank == 0 or share_gpu:
        sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
    if sharded:
        dist.barrier()
    if not (local_rank == 0 or share_gpu):
        sr.load_env_hdr(rmdf_amd.DEFAULT_ENV_HDR)
    env_load_s = time.perf_counter() - t_env0
    dev_name, cus = sr.device_info()

    # Frames are independent, so S of them are kept in flight: frame i goes to HIP stream i % S (dedicated, non-null
    # streams; kernels, the RCCL calls and the timing events of a frame all go on its stream) and owns buffer set
    # i % S.  On one GPU this overlaps the thin tail of a frame -- the launch cannot end before its longest ray has
    # finished a ~0.45 ms serial chain -- with the bulk of the next; on N GPUs it also overlaps the gather of frame i
    # with the render of frame i+1.  --streams 1 = strictly one frame at a time.
    S = a.streams if a.streams > 0 else (3 if not sharded else 8)
    if int(os.environ.get("RMDF_FLAGS", "0")) & rmdf_amd.FLAG_FLAT_MARCH:
        # the alternative schedule of librmdf_xcheck.so keeps one scratch set per ctx and renders on the ctx stream only
        raise SystemExit("RMDF_FLAGS selects the alternative schedule (librmdf_xcheck.so): it supports neither frames in flight nor "
                         "caller streams; time it with tools/bench_configs.py / the cross-check tests, not with bench.py")
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    stream = streams[0]
    torch.cuda.set_stream(stream)
    assert all(st.cuda_stream != 0 for st in streams)

    L = a.supersample
    rw, rh = w << L, h << L                                   # ray grid
    i32 = dict(dtype=torch.int32, device=dev)
    frames = [torch.empty((h, w), **i32) for _ in range(S)]
    exchange = "none (single GPU)"
    rccl_ranks = 0
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
        # The exchange behind the C ABI: rank 0 draws an RCCL unique id, torch.distributed ships the 128 bytes, every rank
        # joins the library's communicator.  Any failure (all ranks decide together) falls back to dist.gather.
        use_abi_comm = (not share_gpu or rccl_double) and os.environ.get("RMDF_BENCH_TORCH_GATHER") != "1"
        if use_abi_comm:
            # every step that can fail on ONE rank is followed by an all-reduce of the outcome before the next collective
            # starts, so a rank that cannot load RCCL makes all ranks fall back together instead of leaving the others waiting
            def all_ok(ok):
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return int(flag.item()) == 1
            uid = torch.zeros(rmdf_amd.COMM_ID_BYTES, dtype=torch.uint8, device=cdev)
            torch.cuda.synchronize()
            try:
                # The exchange's own calls against this rank itself, BEFORE it joins the job's communicator: the ctx has none yet,
                # so the library runs its grouped ncclRecv + ncclSend of one shard's size on a private one-rank communicator --
                # nothing a peer could hold up -- and compares the bytes (the lines that move the tiles have no other coverage on
                # a single GPU).  A failure makes all ranks fall back to the torch.distributed gather together.
                dog.arm("rmdf_comm_selftest_loopback (private one-rank communicator)")
                sr.comm_selftest_loopback(slots * (h // 8) * (w // 8) * 4, stream=streams[0].cuda_stream)
                dog.disarm()
                my_id = rmdf_amd.comm_get_unique_id(xcheck=rccl_double)   # on every rank: proves librccl loads here (only rank 0's is used)
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(my_id), dtype=torch.uint8))
                ok = True
            except Exception as e:                              # noqa: BLE001
                print("rank %d: rmdf_comm_selftest_loopback / rmdf_comm_get_unique_id failed (%s)" % (rank, e), file=sys.stderr)
                ok = False
            use_abi_comm = all_ok(ok)
            if use_abi_comm:
                dist.broadcast(uid, src=0)
                try:
                    dog.arm("rmdf_comm_init (ncclCommInitRank, %d ranks)" % world)
                    sr.comm_init(bytes(uid.cpu().numpy().tobytes()), rank, world)
                    dog.disarm()
                    ok = True
                except Exception as e:                          # noqa: BLE001
                    print("rank %d: rmdf_comm_init failed (%s)" % (rank, e), file=sys.stderr)
                    ok = False
                use_abi_comm = all_ok(ok)
                if not use_abi_comm and ok:
                    sr.comm_destroy()
            if not use_abi_comm:
                print("rank %d: falling back to the torch.distributed gather" % rank, file=sys.stderr)
        dog.disarm()
        if use_abi_comm:
            rccl_ranks = sr.comm_info()[1]
            exchange = "librmdf: rmdf_render_frame_sharded_device (grouped ncclSend/ncclRecv fan-in to rank 0), RCCL communicator of %d ranks" % rccl_ranks
            if rccl_double:
                exchange = ("librmdf_xcheck: rmdf_render_frame_sharded_device with %d ranks SHARING ONE GPU over a TEST DOUBLE of RCCL (tests/fake_rccl.c, "
                            "bytes through /dev/shm): the library's N > 1 code paths run end to end; the figures of this run say nothing about xGMI or scaling" % rccl_ranks)
        else:
            exchange = "torch.distributed gather (%s)" % ("gloo, host-staged: test aid" if share_gpu else "nccl = RCCL")
    frame = frames[0]
    torch.cuda.synchronize(dev)                              # the zero fills above ran on torch's stream, not on the frame streams

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

    def exchange_only(k=0):
        sp = streams[k].cuda_stream
        if use_abi_comm:
            sr.gather_shards_device(w, h, shards[k].data_ptr(), gathereds[k].data_ptr() if rank == 0 else 0, stream=sp)
            g = gathereds[k]
        else:
            with torch.cuda.stream(streams[k]):                # the RCCL call orders itself against the current stream
                g = gather_shards(shards[k], rank, world, dist, out=gathereds[k], out_list=gather_lists[k])
        if rank == 0:
            sr.assemble_shards_device(w, h, world, g.data_ptr(), frames[k].data_ptr(), stream=sp)

    def step(i=0):
        step_at(i, a.time + a.animate * i)

    def step_at(i, t):
        k = i % S
        sp = streams[k].cuda_stream
        if not sharded:
            render_only(k, t)
            if L:
                resolve(bigs[k], rw, rh, frames[k], tmps[k], sp)
        elif use_abi_comm and not L:
            sr.render_frame_sharded_device(scene, w, h, t, ms, shards[k].data_ptr(), gathereds[k].data_ptr() if rank == 0 else 0,
                                           frames[k].data_ptr() if rank == 0 else 0, stream=sp)
        else:
            render_only(k, t)
            if L:
                resolve(bigs[k], rw // 8, slots * (rh // 8), shards[k], tmps[k], sp)
            exchange_only(k)

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def event_ms(fn, reps):
        """average GPU time of fn() on stream 0 (HIP events on the stream the work is launched on)"""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in evs:
            e0.record(stream)
            fn()
            e1.record(stream)
        torch.cuda.synchronize(dev)
        ts = [e0.elapsed_time(e1) for e0, e1 in evs]
        return float(np.mean(ts)), float(np.min(ts))

    deal = "single GPU"
    shard_split = None
    if sharded:
        # cost-aware deal of the 64 tiles: every rank probes the view at 256 x 144 on its own GPU (bit-reproducible
        # kernels -> identical costs, no exchange) and deals longest-processing-time-first.  Outside the timed region
        # the ranks compare their deals once; any disagreement falls back to the static deal on all of them.
        sr.set_shard_costs(sr.probe_tile_costs(a.scene, a.width << a.supersample, a.height << a.supersample, a.time, a.max_steps))
        # Rank 0 also receives (world-1)/world of every frame and assembles it.  Measured here, before the timed region:
        # the exchange + assembly alone on rank 0's stream (all ranks send at once, so the root sees transfer + launch cost,
        # not waiting) against the mean render time of a rank's shard; the deal starts rank 0 with that share of load.
        # RMDF_ROOT_HANDICAP overrides (a fraction of a rank's fair share).
        dog.arm("the first exchange of the run (shard render + gather to rank 0 + assemble, 3 frames)")
        for _ in range(3):
            render_only(0)
            exchange_only(0)
        barrier()
        dog.disarm()
        t_render = event_ms(lambda: render_only(0), 10)[0]
        barrier()
        t_exch = event_ms(lambda: exchange_only(0), 10)[0]
        barrier()
        tt = torch.tensor([t_render, t_exch if rank == 0 else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        mean_render, root_exchange = float(tt[0].item()) / world, float(tt[1].item())
        mm = torch.tensor([t_render, -t_render], dtype=torch.float64, device=cdev)
        dist.all_reduce(mm, op=dist.ReduceOp.MAX)
        shard_split = {"shard_render_ms_mean_over_ranks": round(mean_render, 4), "exchange_plus_assemble_ms_rank0": round(root_exchange, 4),
                       "shard_render_ms_min_over_ranks": round(-float(mm[1].item()), 4), "shard_render_ms_max_over_ranks": round(float(mm[0].item()), 4),
                       "note": "measured before the timed region, one frame at a time on stream 0: every rank's shard render (HIP events), and on "
                               "rank 0 the gather + assembly alone while all peers send at once"}
        measured = min(0.5, root_exchange / max(mean_render, 1e-6))
        handicap = float(os.environ.get("RMDF_ROOT_HANDICAP", measured))
        sr.set_shard_root_handicap(handicap)
        agree = ranks_agree_on_deal([sr.shard_tiles(r, world) for r in range(world)], dist, cdev)
        verified = False
        if agree and use_abi_comm:
            # the library's own collective check (rmdf_comm_verify_deal: fingerprints to rank 0, verdict back); the verdict is the same on
            # every rank (the sizes on the wire never depend on the deal: whole fixed-size slots)
            dog.arm("rmdf_comm_verify_deal")
            try:
                sr.comm_verify_deal(stream=streams[0].cuda_stream)
                verified = True
            except Exception as e:                              # noqa: BLE001
                print("rank %d: %s" % (rank, e), file=sys.stderr)
                agree = False
            dog.disarm()
        if agree:
            deal = "cost-aware (probe frame, LPT, rank 0 handicap %.3f = measured exchange+assemble %.4f ms / mean shard render %.4f ms%s)%s" % (
                handicap, root_exchange, mean_render, ", overridden by RMDF_ROOT_HANDICAP" if "RMDF_ROOT_HANDICAP" in os.environ else "",
                ", deal verified by the library (rmdf_comm_verify_deal)" if verified else "")
        else:
            sr.set_shard_costs(None)
            sr.set_shard_root_handicap(0.0)
            deal = "static (ranks disagreed on the probed costs)"

    # N > 1: before anything is timed, the frames the exchange assembles are compared with the committed digest of this very frame
    # (tests/golden/full_size_digests.json: sha256 of the oracle's RGBA8 plane) -- first one frame at a time, then with all S frame
    # streams in flight on the one communicator.  Frames that differ with S in flight but not alone drop the run to ONE frame in
    # flight (one stream on the communicator) and the JSON line says so; frames that differ even alone end the run: a wrong frame
    # is not timed.
    frames_verified = None
    if sharded:
        want = None
        if [scene, w, h, ms, L] == [2, 1920, 1080, 256, 0] and a.time == 0.0 and a.animate == 0.0:
            try:
                want = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size_digests.json")))["config3_mandelbulb8_1920x1080_m256"]["sha256"]["rgba8"]
            except Exception as e:                              # noqa: BLE001
                print("no committed digest for the exchanged frame: %s" % e, file=sys.stderr)

        def frames_ok(ks):
            ok = 1
            if rank == 0 and want is not None:
                for k in ks:
                    got = hashlib.sha256(frames[k].cpu().numpy().tobytes()).hexdigest()
                    if got != want:
                        print("exchanged frame of stream %d differs from the committed digest (%s... != %s...)" % (k, got[:12], want[:12]), file=sys.stderr)
                        ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1
        if want is not None:
            dog.arm("verification of the exchanged frames against the committed digest")
            for k in range(S):
                frames[k].zero_() if rank == 0 else None
            torch.cuda.synchronize(dev)
            step(0)
            barrier()
            alone_ok = frames_ok([0])
            if not alone_ok:
                raise SystemExit("bench.py: the frame assembled from %d shard(s) differs from the committed digest -- not timing wrong frames" % world)
            for i in range(2 * S):
                step(i)
            barrier()
            if frames_ok(range(S)):
                frames_verified = "%d exchanged frame(s) in flight == committed sha256 of the frame" % S
            else:
                S = 1
                frames_verified = "frames differed with several in flight on one communicator: dropped to ONE frame in flight (alone == committed sha256)"
                print("rank %d: %s" % (rank, frames_verified), file=sys.stderr)
            dog.disarm()

    # warm-up: W steps, then keep stepping until MIN_WARM_SECONDS have passed (all ranks decide together)
    def dog_block(what, n_steps):
        # re-armed per block, scaled with its size: a long block is not a hung collective (10 ms per step is 25x the slowest step measured)
        if sharded:
            dog.arm(what, seconds=max(dog.seconds, 60.0 + 0.01 * n_steps))
    dog_block("warm-up", a.warmup)
    for i in range(a.warmup):
        step(i)
    barrier()
    t_w0 = time.perf_counter()
    extra_warm = 0
    while True:
        dog_block("warm-up", max(a.warmup, 10))
        for i in range(max(a.warmup, 10)):
            step(i)
        extra_warm += max(a.warmup, 10)
        barrier()
        if max_over_ranks(time.perf_counter() - t_w0, dist, cdev) >= MIN_WARM_SECONDS or extra_warm >= 100000:
            break
    dog.disarm()
    # RMDF_BENCH_MARK=1 (tools/profile.sh): a one-thread marker kernel (k_resolve_box2 on a 2x2 image, used by nothing else in this
    # run) is launched BEFORE the opening barrier of every timed block and after the closing barrier of the last one, so that a
    # rocprofv3 kernel trace of this command can be cut into the timed blocks (tools/pmc_summary.py: first dispatch start -> last
    # dispatch end of a block / steps = the figure `ms_per_step` should reproduce).  Nothing is added inside the timed region.
    mark = None
    if os.environ.get("RMDF_BENCH_MARK") == "1" and rank == 0:
        m_src, m_dst = torch.zeros((2, 2), **i32), torch.zeros((1, 1), **i32)
        torch.cuda.synchronize(dev)
        mark = lambda: sr.resolve_box2_device(m_src.data_ptr(), 2, 2, m_dst.data_ptr(), stream=streams[0].cuda_stream)

    def timed_block(n_steps, first_i=0, dt_anim=None):
        """EXACTLY n_steps steps between barrier + synchronize on both sides.  Returns (wall seconds, max over ranks; host seconds
        this rank spent issuing; device span in ms: first stream-start event -> last stream-end event of the block)."""
        if mark:
            mark()
        dog_block("a timed block of %d steps" % n_steps, n_steps)
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
        barrier()
        t0 = time.perf_counter()
        for k in range(S):
            ev0[k].record(streams[k])
        for i in range(n_steps):
            step(first_i + i) if dt_anim is None else step_at(first_i + i, a.time + dt_anim * (first_i + i))
        t_issue = time.perf_counter() - t0
        for k in range(S):
            ev1[k].record(streams[k])
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0, dist, cdev)
        dog.disarm()
        span = max(e0.elapsed_time(e1) for e0 in ev0 for e1 in ev1)
        return wall, t_issue, span

    blocks, host_enqueue, spans = [], [], []
    for _ in range(max(1, a.repeats)):
        wall, t_issue, span = timed_block(a.steps)
        blocks.append(wall)
        host_enqueue.append(t_issue / a.steps * 1e3)                             # host time to issue one step (this rank)
        spans.append(span / a.steps)
    # the same block with the viewer's animation (in_time advances 1/60 s per frame): the cost-ordered dispatch then works from
    # the costs of a slightly different view instead of a perfect table
    anim_ms = None
    if a.animate == 0.0 and not a.no_animated:
        timed_block(max(a.warmup, 10), dt_anim=1.0 / 60.0)
        anim_ms = timed_block(a.steps, dt_anim=1.0 / 60.0)[0] / a.steps * 1e3
        for i in range(S):                                         # every frame buffer holds the in_time = a.time frame again (--check)
            step(i)
        barrier()
    if mark:
        mark()
        torch.cuda.synchronize(dev)
    dog.disarm()
    order = sorted(range(len(blocks)), key=lambda i: blocks[i])
    med = order[len(order) // 2]
    dt = blocks[med]
    ms_per_step = dt / a.steps * 1e3
    mpix = w * h / 1e6
    value = mpix / (dt / a.steps)

    # dominant kernel alone: HIP events on the launch stream around each launch (this rank's share)
    kern_ms, kern_ms_min = event_ms(lambda: render_only(0), min(max(a.steps, 20), 100))

    # the clock the shader engines sustain under this load (outside the timed region): a one-wave probe of the library stamps
    # shader cycles against the 100 MHz real-time counter while ~30 more frames run on the frame streams
    clock_mhz = None
    if not sharded:
        try:
            for i in range(10):
                step(i)
            probes = []
            for _ in range(3):
                for i in range(12):
                    step(i)
                probes.append(sr.probe_shader_clock(300.0))
            torch.cuda.synchronize(dev)
            clock_mhz = float(np.median(probes))
        except Exception as e:                                      # noqa: BLE001
            print("shader clock probe failed: %s" % e, file=sys.stderr)

    result = None
    if rank == 0:
        # PCIe-inclusive rate (host buffer hand-over as the boundary does it) -- informational
        host = np.empty(w * h, np.uint32)
        if L == 0:
            sr.draw_shader_tile(scene, None, w, h, a.time, host, max_steps=ms)
        else:
            sr.render_supersampled(scene, w, h, L, a.time, max_steps=ms)
        # (three blocks, the median: five calls in a row right after the timed region were within +-8 % of each other from run to run)
        reps = 20 if L == 0 else 5
        blocks = []
        for _ in range(3):
            t1 = time.perf_counter()
            for _ in range(reps):
                if L == 0:
                    sr.draw_shader_tile(scene, None, w, h, a.time, host, max_steps=ms)
                else:
                    sr.render_supersampled(scene, w, h, L, a.time, max_steps=ms)
            blocks.append((time.perf_counter() - t1) / reps)
        d2h_rate = mpix / sorted(blocks)[1]
        # (rounds 2-4 also timed the call into a buffer registered with rmdf_register_host_buffer -- hipHostRegister on the caller's pages,
        # written by the render kernel directly: 4050-4310 Mpixels/s.  Round 5 retired that mapping with the GPU memory fault it was part
        # of (NOTEBOOK.md A.5); registration is bookkeeping now and every buffer takes the path timed above.)
        d2h_registered = None

        env_bytes = 6 * 172 * 172 * 8 + 2 * 6 * 87 * 87 * 8               # padded RGB16F cube maps read once
        px_this_launch = rw * rh if not sharded else len(sr.shard_tiles(rank, world)) * (rw // 8) * (rh // 8)
        algo_bytes = px_this_launch * 4 + env_bytes                        # RGBA8 store + env read
        achieved_gbs = algo_bytes / (kern_ms * 1e-3) / 1e9
        headline = [scene, w, h, ms] == [2, 1920, 1080, 256] and a.time == 0.0 and L == 0
        pmc, pmc_src = (None, "not collected for this workload")
        if not sharded and L == 0:
            pmc, pmc_src = pmc_record(lib_path, [scene, w, h, ms])
            under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
            if (pmc is None and headline and a.pmc == "auto" and world == 1 and not os.environ.get("RMDF_BENCH_PMC_CHILD")
                    and not os.environ.get("RMDF_LIB") and not a.no_secondary and not under_profiler):      # (never a profiler inside a profiler)
                # no counter passes of THIS build are committed: collect them now (the GPU is idle: the timed region is over)
                # (every child has its own timeout and the passes share a 240 s budget: no watchdog needed, and none that could cost the line)
                try:
                    live, live_src = live_pmc_passes(lib_path)
                except Exception as e:                              # noqa: BLE001
                    live, live_src = None, "live counter passes failed: %s" % e
                if live is not None:
                    pmc, pmc_src = live, live_src
                else:
                    pmc_src = pmc_src + "; " + live_src
        kname = "k_render<%d, %s, OUT_RGBA8>" % (scene, "true" if scene != 0 and not (int(os.environ.get("RMDF_FLAGS", "0")) & 16) else "false")
        result = {
            "metric": "Mpixels/s, Mandelbulb power-8 1920x1080 @256 steps; 1/2/4/8 GPU",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "no input data beyond the scene: the Mandelbulb is procedural; environment = the reference's own light probe "
                    "uffizi_512.hdr (shipped, sha256 ef959a2b...), prefiltered and converted by the product's pipeline",
            "config": {"workload": "FragmentShader %d (2 = FSMBPower8Shader) %dx%d%s, max_steps %d, in_time %.1f, uffizi_512.hdr env "
                                   "(cube maps built by the product's own env pipeline), full frame -> RGBA8 resident in HBM" %
                                   (scene, w, h, (" x %d rays/px, box-resolved on the GPU" % (4 ** L)) if L else "", ms, a.time),
                       "supersample_levels": L, "mrays_per_s": round(value * 4 ** L, 2),
                       "scene": scene, "width": w, "height": h, "max_steps": ms,
                       "parallelism": ("1 GPU, one launch per frame" if not sharded else
                                       "64 tiles dealt to %d %s + one gather per frame" %
                                       (world, "ranks sharing one GPU (test aid)" if share_gpu else "GPUs")) +
                                      ", %d frame(s) in flight" % S,
                       "exchange": exchange, "rccl_ranks": rccl_ranks,
                       "exchange_ms": None if shard_split is None else shard_split["exchange_plus_assemble_ms_rank0"],
                       "shard_render_ms": None if shard_split is None else shard_split["shard_render_ms_mean_over_ranks"],
                       "shard_render_ms_min": None if shard_split is None else shard_split["shard_render_ms_min_over_ranks"],
                       "shard_render_ms_max": None if shard_split is None else shard_split["shard_render_ms_max_over_ranks"],
                       "frames_in_flight": S, "exchanged_frames_verified": frames_verified, "tile_deal": deal, "animate_dt": a.animate,
                       "device": dev_name, "compute_units": cus},
            "repeats": len(blocks), "ms_per_step_blocks": [round(b / a.steps * 1e3, 4) for b in blocks],
            "ms_per_step_min": round(min(blocks) / a.steps * 1e3, 4),
            "device_span_ms_per_step": round(spans[med], 4),
            "device_span_note": "the median block again, measured on the device: HIP events recorded on every frame stream at the start and "
                                "at the end of the block, latest end - earliest start, / steps (rank 0's streams); the tracked counterpart "
                                "from a rocprofv3 kernel trace of this command is profiles/*_schedule_default.json",
            "value_animated": None if anim_ms is None else round(mpix / (anim_ms * 1e-3), 2),
            "ms_per_step_animated": None if anim_ms is None else round(anim_ms, 4),
            "animated_note": "one more block of `steps` steps with in_time advancing 1/60 s per frame (the viewer's animation): the cost-"
                             "ordered dispatch then runs on the previous frame's costs of a slightly different view; `value` renders the "
                             "same frame every step",
            "timing_note": "value / ms_per_step = the median of `repeats` blocks of exactly `steps` steps, each bracketed by "
                           "barrier + synchronize; %d + %d untimed warm-up steps (>= %.1f s) ran before the first block" % (a.warmup, extra_warm, MIN_WARM_SECONDS),
            "hbm_roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 6),
                             "algorithmic_bytes_per_launch": algo_bytes,
                             "traffic": None if not pmc else pmc.get("hbm_bytes_per_launch"), "traffic_source": pmc_src,
                             "note": "asked for by the north star; the path is VALU-bound (SURVEY 8d), so this is a sanity figure"},
            "d2h_inclusive_mpixels_s": round(d2h_rate, 2),
            "d2h_inclusive_registered_buffer_mpixels_s": None if d2h_registered is None else round(d2h_registered, 2),
            "host_enqueue_ms_per_step_rank0": round(float(np.median(host_enqueue)), 4),
            "env_pipeline_load_s": round(env_load_s, 3),
        }
        ctr, ctr_src = (dict(HEADLINE_COUNTERS), "instrumented oracle, committed (tests/golden/full_size_digests.json)") if headline else (None, None)
        ref = None
        if ((not a.no_cpu_baseline and world == 1) or a.check) and L == 0:      # CPU baselines: rank 0 at N = 1 only
            from oracle import orc
            env = load_oracle_env(orc)
            cores = orc.num_processors()
            tc = time.perf_counter()
            ref = orc.render(scene, w, h, a.time, ms, env, nthreads=cores, want_f32=False)
            cpu_dt = time.perf_counter() - tc
            ctr, ctr_src = ref["counters"], "instrumented oracle, counted in this run"
            result["cpu_baseline"] = {"value": round(mpix / cpu_dt, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
                                      "sample": "1 full frame %dx%d of the same workload, CPU oracle (C port of fragment.shd), "
                                                "row segments over all host cores as ConcurrentSegments does" % (w, h)}
        if ctr is not None and not sharded and L == 0:
            F = flops_model(ctr)
            ach = F / (kern_ms * 1e-3) / 1e12
            result["roofline"] = {"bound": "valu", "kernel": kname, "achieved": round(ach, 3), "peak": VALU_PEAK_TLANEOPS,
                                  "unit": "T lane-ops/s", "frac": round(ach / VALU_PEAK_TLANEOPS, 4),
                                  "traffic": None if not pmc else pmc.get("hbm_bytes_per_launch"), "traffic_source": pmc_src,
                                  "kernel_ms_avg": round(kern_ms, 4), "kernel_ms_min": round(kern_ms_min, 4),
                                  "shader_clock_mhz_under_load": None if clock_mhz is None else round(clock_mhz, 0),
                                  "frac_at_measured_clock": None if clock_mhz is None else round(ach / (VALU_PEAK_TLANEOPS * clock_mhz / 2400.0), 4),
                                  "clock_note": "peak is priced at 2.4 GHz; the shader engines sustain less under this load (rmdf_probe_shader_clock: "
                                                "s_memtime against s_memrealtime on one wave beside the running frames, median of 3 probes of 0.3 ms)",
                                  "kernel_ms_note": "HIP events on the launch stream around one frame's launches, one frame at a time: k_render "
                                                    "plus the single-workgroup k_order_blocks (~0.009 ms) that follows it; the rocprofv3 "
                                                    "average of k_render alone is in profiles/*_kernel_stats_s1.csv (same command with --streams 1)",
                                  "ops_per_launch": F, "op_counters": ctr, "op_counters_source": ctr_src,
                                  "note": "binding roof = FP32 vector-ALU issue (SURVEY 8d): as-written IEEE operations (sqrt, 1/sqrt, "
                                          "log, pow, / each 1; no FMA contraction by the parity contract) per second against 256 CU x "
                                          "4 SIMD x 32 lanes x 2.4 GHz; HBM side in hbm_roofline"}
            ach_roof = achievable_issue_roof()
            if headline and ach_roof:
                # (B) of profiles/r06_march_loop_classes.txt: the instruction stream this kernel has to issue -- its ISA priced in issue slots
                # with the per-form costs measured on MI355X -- on 64 live lanes, against the as-written roof (A) above
                result["roofline"]["achievable_issue"] = {
                    "ms_at_spec_issue_rate": round(ach_roof["t_spec_ms"], 4), "ms_at_sustained_v_mul_rate": round(ach_roof["t_sust_ms"], 4),
                    "frac_of_spec": round(ach_roof["t_spec_ms"] / kern_ms, 4), "frac_of_sustained": round(ach_roof["t_sust_ms"] / kern_ms, 4),
                    "vector_slots_per_iteration_pass": ach_roof["vslots"], "vector_instructions_per_iteration_pass": ach_roof["pass_instr"],
                    "as_written_ops_per_iteration_pass": 79, "wave_slots_at_full_lanes": round(ach_roof["full_slots"]),
                    "source": "profiles/r06_march_loop_classes.json (tools/isa/march_loop_classes.py)",
                    "note": "what is left between `frac_of_sustained` and 1 is lane utilisation (rays that have ended while their packet "
                            "marches on); between this roof and `peak` the exact roots, guards and the pinned log of the parity contract"}
            if not pmc and _PMC_OTHER_BUILD.get("record"):
                # no counter passes exist for THIS build (round 5: GPU access closed before the round's profiling run).  The committed ones
                # are round 4's; that build's k_render<2, true, 0> has the same march loop instruction for instruction (only the shading
                # tail lost 21 instructions: exp's scaling became one v_ldexp_f32), so they are printed -- apart from `traffic`, which stays
                # null because it would not be a measurement of this library.
                o = _PMC_OTHER_BUILD["record"]
                result["roofline"]["counters_of_previous_build"] = {
                    "lib_sha256": o.get("lib_sha256"), "hbm_bytes_per_launch": o.get("hbm_bytes_per_launch"),
                    "valu_instructions_per_launch": (o.get("valu") or {}).get("SQ_INSTS_VALU_per_launch"),
                    "lane_utilisation": (o.get("valu") or {}).get("lane_utilisation"),
                    "note": "rocprofv3 --pmc passes of the round-4 library (profiles/pmc_traffic.json); not of the library this run timed"}
            valu_instr = None if not pmc else (pmc.get("valu") or {}).get("SQ_INSTS_VALU_per_launch")
            if valu_instr:
                simds = cus * 4
                rate = valu_instr / simds / (kern_ms * 1e-3) / 1e9
                result["roofline"]["issue"] = {"valu_instructions_per_launch": valu_instr, "source": pmc_src, "simds": simds,
                                               "achieved": round(rate, 3), "peak": VALU_ISSUE_SPEC, "frac": round(rate / VALU_ISSUE_SPEC, 3),
                                               "measured_v_mul_stream": VALU_ISSUE_PEAK, "frac_of_measured_v_mul_stream": round(rate / VALU_ISSUE_PEAK, 3),
                                               "frac_of_spec_peak_at_measured_clock": None if clock_mhz is None else round(rate / (clock_mhz / 2000.0), 3),
                                               "peak_note": "`peak` = the documented one wave64 instruction per SIMD-32 every 2 cycles at 2.4 GHz (round 5: "
                                                            "the figure the fraction is taken against); `measured_v_mul_stream` = what a pure v_mul_f32 stream was "
                                                            "measured to sustain on this chip (tools/ubench/valu_rates: 0.82-1.0, one per 2.2-2.4 cycles), kept as a "
                                                            "note -- the kernel's mix has issued above it, so it is not a ceiling",
                                               "unit": "G wave-instructions/s/SIMD",
                                               "achieved_frames_in_flight": round(valu_instr / simds / (ms_per_step * 1e-3) / 1e9, 3),
                                               "frames_in_flight_note": "the same instruction count over ms_per_step (the schedule `value` is measured "
                                                                        "on): with frames in flight the vector pipes issue at this rate",
                                               "lane_utilisation": (pmc.get("valu") or {}).get("lane_utilisation")}
        elif ctr is not None and sharded and L == 0:
            # N GPUs: the frame's as-written operations against the job's N vector-ALU roofs, at the whole-job rate the timed region
            # measured (the per-launch kernel time of one rank's shard is kernel_ms_avg; sharding adds helper rows and the exchange)
            F = flops_model(ctr)
            ach = F / (ms_per_step * 1e-3) / 1e12
            result["roofline"] = {"bound": "valu", "kernel": kname + " (one shard launch per rank and frame)", "achieved": round(ach, 3),
                                  "peak": round(VALU_PEAK_TLANEOPS * world, 1), "unit": "T lane-ops/s", "frac": round(ach / (VALU_PEAK_TLANEOPS * world), 4),
                                  "traffic": None, "kernel_ms_avg": round(kern_ms, 4), "kernel_ms_min": round(kern_ms_min, 4),
                                  "ops_per_launch": F, "op_counters": ctr, "op_counters_source": ctr_src,
                                  "exchange": shard_split,
                                  "note": "aggregate: as-written IEEE operations of the whole frame / ms_per_step against n_gpus x 78.6 T "
                                          "lane-ops/s; kernel_ms_avg = rank 0's shard launch alone (HIP events)"}
        else:
            result["roofline"] = {"bound": "valu", "kernel": kname, "achieved": None, "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s",
                                  "frac": None, "traffic": None, "kernel_ms_avg": round(kern_ms, 4), "kernel_ms_min": round(kern_ms_min, 4),
                                  "exchange": shard_split,
                                  "note": "operation counters are only known for the headline workload (or after the "
                                          "cpu_baseline leg counted them)"}
        if not a.no_secondary and world == 1 and not sharded and L == 0:
            # (the extra workloads never cost the headline line: a failure in them is reported in the line instead)
            try:
                result["secondary"] = secondary_workloads(sr, torch, dev, stream, cus, streams)
            except Exception as e:                                  # noqa: BLE001
                result["secondary"] = None
                result["secondary_error"] = "%s: %s" % (type(e).__name__, e)
        if ref is not None:
            from oracle import orc
            # the reference's own CPU paths (BASELINE.json north_star: "timed on the same box's host cores ... as the
            # reported, non-target baseline"), as their C restatements with the reference's threading model
            # (SURVEY.md 8d): Fractal2D.juliaAnimated over all cores in row segments; cosineConvolveHDREnvMap with one
            # thread per power (ShaderRendering.hs:142) -- one power timed on one thread -- and split over all cores
            small = orc.resize_hdr(oracle_env_latlongs(orc)["refl"], 256)
            tj0 = time.perf_counter(); orc.julia_animated(512, 512, 0, 0.0); tj = time.perf_counter() - tj0
            tj0 = time.perf_counter(); orc.julia_animated(512, 512, 1, 0.0); tjs = time.perf_counter() - tj0
            tj0 = time.perf_counter(); orc.julia_animated(1920, 1080, 0, 0.0); tjb = time.perf_counter() - tj0
            tj0 = time.perf_counter(); orc.julia_animated(1920, 1080, 1, 0.0); tjbs = time.perf_counter() - tj0
            refl_ll = oracle_env_latlongs(orc)["refl"]
            orc.latlong_to_cube(refl_ll)                           # first call builds nothing persistent, but warms the pool
            tq0 = time.perf_counter(); orc.latlong_to_cube(refl_ll); tq = time.perf_counter() - tq0
            tc0 = time.perf_counter(); orc.cosine_convolve(small, 8.0, nthreads=1, pow_mode=0); tc1 = time.perf_counter() - tc0
            tc0 = time.perf_counter(); orc.cosine_convolve(small, 8.0, nthreads=0, pow_mode=0); tca = time.perf_counter() - tc0
            result["cpu_reference_paths"] = {
                "cores": orc.num_processors(), "kind": "port",
                "julia_animated_512x512_ms": round(tj * 1e3, 2), "julia_animated_mpixels_s": round(0.262144 / tj, 1),
                "julia_animated_512x512_smooth_ms": round(tjs * 1e3, 2),
                "julia_animated_1920x1080_ms": round(tjb * 1e3, 2), "julia_animated_1920x1080_smooth_ms": round(tjbs * 1e3, 2),
                "latlong_to_cube_512x256_to_6x170x170_all_cores_ms": round(tq * 1e3, 2),
                "note": "C restatements with the reference's threading: juliaAnimated and latLongHDREnvMapToCubeMap in nproc row "
                        "segments (ConcurrentSegments.hs:14-28, Fractal2D.hs:98, HDREnvMap.hs:139); cosineConvolveHDREnvMap one power "
                        "per thread (ShaderRendering.hs:142) and, labelled, split over all cores; literal libm powf (pow_mode 0)",
                "cosine_convolve_256x128_power8_one_thread_s": round(tc1, 3),
                "cosine_convolve_256x128_power8_all_cores_s": round(tca, 3)}
            if a.check:
                result["check_rgba8_equal"] = all(bool(np.array_equal(f.cpu().numpy().view(np.uint32), ref["rgba8"]))
                                                  for f in frames)
        if not a.no_secondary and world == 1 and not sharded and L == 0 and w % 8 == 0 and h % 8 == 0:
            # N = 1 carries the N > 1 line's exchange fields too (rccl_ranks, shard_render_ms min / max, exchange_ms), so that the driver's
            # N = 1 SCALE line can be read field by field beside the others: the same library calls -- shard render of all 64 tiles, gather
            # (the root's own part, in its slot), assembly -- on a ONE-rank RCCL communicator.  LAST, behind everything else the line holds,
            # and never fatal: an error is reported in the line, and should RCCL hang, the watchdog writes the line without this leg.
            def line_without_the_leg():
                result["config"]["one_rank_exchange_error"] = "timed out (60 s): the leg was abandoned, everything else in this line was complete"
                os.write(json_fd, (json.dumps(result) + "\n").encode())
            dog.arm("the one-rank exchange leg (after the timed region)", 60, on_timeout=line_without_the_leg)
            try:
                result["config"].update(one_rank_exchange_leg(sr, torch, dev, stream, event_ms, scene, w, h, ms, a.time,
                                                              frames[0] if a.animate == 0.0 else None, None, xcheck=rccl_double))
            except Exception as e:                                  # noqa: BLE001
                result["config"]["one_rank_exchange_error"] = "%s: %s" % (type(e).__name__, e)
            dog.disarm()
        os.write(json_fd, (json.dumps(result) + "\n").encode())

    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    sr.close()
    return result


def scene_pmc(name, lib_path=None):
    """VALU instruction count of a secondary scene from the committed counter pass (profiles/scene_pmc.json, written by tools/prof_scene.sh),
    or None -- also None when the pass does not name THIS build of librmdf.so (round 5 rewrote the kernels of scenes 1 and 3: round 4's
    instruction counts over the new kernels' times would be a figure of nothing)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "scene_pmc.json"))).get(name, {})
        import rmdf_amd
        if rec.get("lib_sha256") != sha256_file(lib_path or os.environ.get("RMDF_LIB", rmdf_amd.LIB_PATH)):
            return _LIVE_SCENE_PMC.get(name)                    # (collected by this very run, if it did: live_pmc_passes)
        return rec.get("SQ_INSTS_VALU_per_launch")
    except Exception:                                           # noqa: BLE001
        return _LIVE_SCENE_PMC.get(name)


def one_rank_exchange_leg(sr, torch, dev, stream, event_ms, scene, w, h, ms, t, plain_frame, dog=None, xcheck=False):
    """The sharded frame path with a communicator of ONE rank (all one GPU can hold), timed one frame at a time with HIP events on the
    launch stream; the assembled frame must equal the plain launch's.  Returns the exchange fields of the N > 1 line for N = 1."""
    import rmdf_amd
    i32 = dict(dtype=torch.int32, device=dev)
    slots = rmdf_amd.shard_slots(1)
    gathered = torch.zeros((1, slots, h // 8, w // 8), **i32)
    frame = torch.zeros((h, w), **i32)
    torch.cuda.synchronize(dev)                                  # the fills ran on torch's stream; the library's calls below do not wait for it
    sp = stream.cuda_stream
    own = sr.comm_info()[1] == 0
    if own:
        if dog: dog.arm("rmdf_comm_init (one-rank communicator, N = 1 exchange leg)")
        sr.comm_init(rmdf_amd.comm_get_unique_id(xcheck=xcheck), 0, 1)
        if dog: dog.disarm()
    try:
        shard = gathered[0]                                      # the root renders straight into its slot, as at N > 1
        render = lambda: sr.render_shard_device(scene, w, h, t, ms, 0, 1, shard.data_ptr(), stream=sp)

        def exch():
            sr.gather_shards_device(w, h, shard.data_ptr(), gathered.data_ptr(), stream=sp)
            sr.assemble_shards_device(w, h, 1, gathered.data_ptr(), frame.data_ptr(), stream=sp)
        whole = lambda: sr.render_frame_sharded_device(scene, w, h, t, ms, shard.data_ptr(), gathered.data_ptr(), frame.data_ptr(), stream=sp)
        for _ in range(3):
            whole()
        torch.cuda.synchronize(dev)
        same = None if plain_frame is None else bool(torch.equal(frame, plain_frame))
        t_r = event_ms(render, 10)
        t_x = event_ms(exch, 10)
        t_w = event_ms(whole, 10)
        return {"rccl_ranks": sr.comm_info()[1], "exchange_ms": round(t_x[0], 4), "shard_render_ms": round(t_r[0], 4),
                "shard_render_ms_min": round(t_r[0], 4), "shard_render_ms_max": round(t_r[0], 4),
                "sharded_frame_ms": round(t_w[0], 4), "sharded_frame_equals_plain_frame": same,
                "exchange_fields_note": "N = 1: measured AFTER the timed region on a one-rank RCCL communicator (rmdf_render_frame_sharded_device: all 64 "
                                        "tiles as one shard, the root's own part gathered in its slot, k_assemble_shards); `value` is the plain launch"}
    finally:
        if own:
            sr.comm_destroy()


def secondary_workloads(sr, torch, dev, stream, cus, streams=()):
    """The other configurations SURVEY.md 8(d) lists, timed after the headline run (device-resident unless said otherwise): config 2
    (Cornell box 1280x720 @128), config 5 (lobe prefilter 256x128, the four powers), config 4 on one GPU (3840x2160 output from
    7680x4320 rays, box-resolved on the GPU), the reference's 64-tile mode through the boundary call (rmdf_render_tile into a host
    buffer, 64 calls per frame), and the headline scene at in_time 1, 2.5 and 7."""
    out = {}
    fb = torch.empty((720, 1280), dtype=torch.int32, device=dev)
    sp = stream.cuda_stream

    def ev(fn, reps):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in evs:
            e0.record(stream)
            fn()
            e1.record(stream)
        torch.cuda.synchronize(dev)
        return float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))
    cornell = lambda: sr.render_rect_device(0, 1280, 720, 0.0, 128, (0, 0, 1280, 720), d_rgba8=fb.data_ptr(), stream=sp)

    def warm(fn, seconds=0.25):
        """keep launching until `seconds` have passed: the shader clock needs ~20 ms of continuous load to reach its steady 2.4 GHz
        (a run of a few milliseconds after a pause sees 2.06-2.1 GHz), and the headline figure is measured warm too"""
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                fn()
            torch.cuda.synchronize(dev)
    warm(cornell)
    t = ev(cornell, 50)
    # the same with two frames in flight on two streams (what the headline figure does): the launch is 14 400 waves = 1.76
    # fillings of the machine, so the thin end of one frame overlaps the start of the next
    fb2 = torch.empty((720, 1280), dtype=torch.int32, device=dev)
    # the second frame stream: one of the headline run's frame streams when there is one (they are known to sit on hardware queues of
    # their own; a stream created this late can end up sharing a queue with `stream`: 0.18 instead of 0.14 ms per frame, the
    # "bimodal" figure of round 2)
    st2 = streams[1] if len(streams) > 1 else torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    bufs, sps = (fb, fb2), (sp, st2.cuda_stream)
    n2 = 200
    ctr = [0]

    def two():
        i = ctr[0]; ctr[0] += 1
        sr.render_rect_device(0, 1280, 720, 0.0, 128, (0, 0, 1280, 720), d_rgba8=bufs[i & 1].data_ptr(), stream=sps[i & 1])
    warm(two)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(n2):
        sr.render_rect_device(0, 1280, 720, 0.0, 128, (0, 0, 1280, 720), d_rgba8=bufs[i & 1].data_ptr(), stream=sps[i & 1])
    torch.cuda.synchronize(dev)
    t2 = (time.perf_counter() - t0) / n2 * 1e3
    out["config2_cornell_1280x720_m128"] = {"kernel_ms_avg": round(t, 4), "mpixels_s": round(0.9216 / (t * 1e-3), 1),
                                            "two_frames_in_flight_ms_per_frame": round(t2, 4),
                                            "two_frames_in_flight_mpixels_s": round(0.9216 / (t2 * 1e-3), 1),
                                            "roofline": secondary_roofline("config2_cornell_1280x720_m128", 0, t, scene_pmc("config2_cornell_1280x720_m128"), cus)}
    # config 5: the prefilter kernel, 256x128, each reference power alone, then the four concurrently through the host entry
    rng = np.random.RandomState(3)
    src = rng.uniform(0.0, 4.0, (128, 256, 3)).astype(np.float32)
    d_src = torch.from_numpy(src).to(dev)
    d_out = torch.empty_like(d_src)
    pair_terms = (256 * 128) ** 2
    per = {}
    for p in (1.0, 8.0, 64.0, 512.0):
        f = lambda p=p: sr.prefilter_env_device(d_src.data_ptr(), 256, 128, p, d_out.data_ptr(), stream=sp)
        f(); f()
        t = ev(f, 5)
        per["power_%d" % int(p)] = {"kernel_ms": round(t, 3), "g_pair_terms_s": round(pair_terms / (t * 1e-3) / 1e9, 1)}
    sr.prefilter_env_powers(src, (1.0, 8.0, 64.0, 512.0))           # first call: the ctx allocates its scratch and tables
    t0 = time.perf_counter()
    for _ in range(3):
        sr.prefilter_env_powers(src, (1.0, 8.0, 64.0, 512.0))
    four = (time.perf_counter() - t0) / 3
    # config 4 on ONE GPU: frame-buffer scale 2 (App.hs:105-106,131-133) = 7680x4320 rays, one launch, then one mip level of the
    # RGBA8 frame (k_resolve_box2).  Two frames in flight like the headline's schedule would need 2 x 133 MB more; one at a time.
    big = torch.empty((4320, 7680), dtype=torch.int32, device=dev)
    fb4 = torch.empty((2160, 3840), dtype=torch.int32, device=dev)
    c4_render = lambda: sr.render_rect_device(2, 7680, 4320, 0.0, 256, (0, 0, 7680, 4320), d_rgba8=big.data_ptr(), stream=sp)
    c4_resolve = lambda: sr.resolve_box2_device(big.data_ptr(), 7680, 4320, fb4.data_ptr(), stream=sp)
    for _ in range(3):
        c4_render(); c4_resolve()
    t_r, t_s = ev(c4_render, 8), ev(c4_resolve, 8)
    out["config4_mandelbulb_3840x2160_x4rays_one_gpu"] = {
        "render_ms": round(t_r, 4), "resolve_ms": round(t_s, 4), "ms_per_frame": round(t_r + t_s, 4),
        "mrays_s": round(33.1776 / ((t_r + t_s) * 1e-3), 1), "mpixels_s": round(8.2944 / ((t_r + t_s) * 1e-3), 1),
        "note": "BASELINE config 4's work on one GPU (the 8-GPU form shards the 64 tiles and resolves before the gather); "
                "7680x4320 rays @256 steps in one launch + k_resolve_box2, HIP events, one frame at a time"}
    del big, fb4
    # the reference's tiled dispatch (ShaderRendering.hs:181-193): 64 drawShaderTile calls per frame through the boundary,
    # each returning the whole accumulated frame into the caller's Word32 buffer (the PBO is orphaned every call)
    host = np.empty(1920 * 1080, np.uint32)
    for tidx in range(64):
        sr.draw_shader_tile(2, tidx, 1920, 1080, 0.0, host, max_steps=256)
    t0 = time.perf_counter()
    n_tf = 2
    for f in range(n_tf):
        for tidx in range(64):
            sr.draw_shader_tile(2, 64 * (f + 1) + tidx, 1920, 1080, 0.0, host, max_steps=256)
    t_tile = (time.perf_counter() - t0) / (64 * n_tf) * 1e3
    out["tile_mode_64_calls_1920x1080"] = {"ms_per_tile_call": round(t_tile, 4), "ms_per_frame": round(64 * t_tile, 3),
                                           "mpixels_s": round(2.0736 / (64 * t_tile * 1e-3), 1),
                                           "note": "rmdf_render_tile with tile_idx 0..63, host buffer handed back whole on every call "
                                                   "(FrameBuffer.hs:129,207-213), wall clock, PCIe included.  Round 4: the next three tiles of "
                                                   "the latched frame are rendered ahead of their calls into scratch tiles (mirrored into page-"
                                                   "locked host memory by the kernel), the caller's buffer is filled from a page-locked shadow "
                                                   "of the frame by host threads: only the new tile crosses PCIe (round 3: 21-22 ms per frame)"}
    # the other two FragmentShader values (ShaderRendering.hs:46-47,119-122) at the size and view of their committed digests
    for name, sc, tv in (("scene1_detest_1280x720_m128", 1, 2.5), ("scene3_mbgeneral_1280x720_m128", 3, 3.0)):
        fs = lambda sc=sc, tv=tv: sr.render_rect_device(sc, 1280, 720, tv, 128, (0, 0, 1280, 720), d_rgba8=fb.data_ptr(), stream=sp)
        warm(fs, 0.1)
        tt = ev(fs, 20)
        # ... and with two frames in flight on two streams, as for the Cornell box above
        n2s = 60
        for i in range(6):
            sr.render_rect_device(sc, 1280, 720, tv, 128, (0, 0, 1280, 720), d_rgba8=bufs[i & 1].data_ptr(), stream=sps[i & 1])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(n2s):
            sr.render_rect_device(sc, 1280, 720, tv, 128, (0, 0, 1280, 720), d_rgba8=bufs[i & 1].data_ptr(), stream=sps[i & 1])
        torch.cuda.synchronize(dev)
        t2s = (time.perf_counter() - t0) / n2s * 1e3
        out[name] = {"kernel_ms_avg": round(tt, 4), "mpixels_s": round(0.9216 / (tt * 1e-3), 1), "in_time": tv,
                     "two_frames_in_flight_ms_per_frame": round(t2s, 4), "two_frames_in_flight_mpixels_s": round(0.9216 / (t2s * 1e-3), 1),
                     "valu_instructions_per_launch": scene_pmc(name), "note": "one frame at a time, HIP events; instruction count from profiles/ (rocprofv3 --pmc of this scene) when present",
                     "roofline": secondary_roofline({1: "detest_1280x720_t2p5_m128", 3: "mbgeneral_1280x720_t3p0_m128"}[sc], sc, tt, scene_pmc(name), cus)}
    # the headline scene from the SURVEY's other camera times
    fbv = torch.empty((1080, 1920), dtype=torch.int32, device=dev)
    views = {}
    for tv in (1.0, 2.5, 7.0):
        fv = lambda tv=tv: sr.render_rect_device(2, 1920, 1080, tv, 256, (0, 0, 1920, 1080), d_rgba8=fbv.data_ptr(), stream=sp)
        for _ in range(5):
            fv()
        tt = ev(fv, 20)
        views["in_time_%g" % tv] = {"kernel_ms_avg": round(tt, 4), "mpixels_s": round(2.0736 / (tt * 1e-3), 1)}
    out["headline_scene_other_views_1920x1080_m256"] = views
    out["config5_lobe_prefilter_256x128"] = {"per_power": per, "four_powers_concurrent_host_in_out_ms": round(four * 1e3, 3),
                                             "pair_terms_per_power": pair_terms,
                                             "note": "one lane per destination texel sums the source serially in the reference's order "
                                                     "(bit-exact).  A power alone: the factor sin*cos^p of every (destination, source) pair is "
                                                     "computed by producer waves and handed through LDS to three summing waves, one per colour channel "
                                                     "(5632 waves instead of 512; k_prefilter_chan).  The reference's four powers at once (mapConcurrently): ONE launch "
                                                     "-- the squaring chains nest, so producers compute all four factors from one cosine and feed "
                                                     "four summing waves (k_prefilter_fused4)"}
    return out


if __name__ == "__main__":
    main()
