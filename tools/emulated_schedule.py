#!/usr/bin/env python3
"""emulated_schedule.py [w h] -- lane utilisation of the headline kernel's Mandelbulb estimates under the schedule the kernel SOURCE really runs,
from the SIMT emulator (tests/kernel_on_host.cpp; CPU only, ~30 s per 1920 x 1080 frame on eight cores).  The emulator counts, per wave and
per mb8_iterate_t call, the iteration passes of every lane; a wave pays its slowest lane 64 lanes wide.  Cost model = tools/ubench/sched_sim's
(round 4): 87 vector instructions per pass + 100 per estimate.  Printed for the product's schedule (workgroup pooling at <= 32 live rays + AO
queue), for other pooling thresholds and for RMDF_FLAG_NO_MERGE (plain 8 x 8 packets) -- next to round 4's figures from the GPU trace replay
(profiles/r04_sched_regroup.txt: 0.645 plain, 0.741 pooled) and the PMC's whole-kernel 0.735."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rmdf_amd
from oracle import orc
import test_kernel_source_on_host as T

CORNELL = len(sys.argv) > 1 and sys.argv[1] == "cornell"         # emulated_schedule.py cornell: the Cornell box's builds instead (config 2)
if CORNELL:
    sys.argv.pop(1)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else ((1280, 720) if CORNELL else (1920, 1080))
orc.build()
rd = lambda fn: orc.hdr_decode(open(fn, "rb").read())
cache = os.path.join(ROOT, "tests", "golden", "env_cache")
ll = {"refl": rd(rmdf_amd.DEFAULT_ENV_HDR), "cos1": rd(cache + "/uffizi_512_cache_pow_1.0.hdr"), "cos8": rd(cache + "/uffizi_512_cache_pow_8.0.hdr")}
env = orc.EnvSet(*(orc.cube_pad_f16(orc.latlong_to_cube(ll[k])) for k in ("refl", "cos1", "cos8")))
if CORNELL:
    # Instruction CHAINS of the Cornell box (BASELINE config 2), where one frame at a time waits for its slowest wave: every point-triangle
    # distance priced at 130 vector instructions, every bound test at 13 (rmdf_device.hpp: RMDF_EMU_COST); a wave's chain = its slowest lane
    # between collectives, summed over the launch.  A count of ISSUED instructions -- it does not see dependent-chain latency, which is what
    # the lanes-per-ray tail was built to shorten (a lone wave issues one instruction per ~5 cycles when each waits for the one before).
    variants = [((), "", "product: eight lanes per ray once <= 8 rays are live")] + [((d, "-DKOH_RENDER_ONLY"), t, what) for d, t, what in T.AB_BUILDS if t in ("_xl4", "_noxl", "_sharedb")]
    T.Emulated.build([(d, t) for d, t, _ in variants])
    print("Cornell box %d x %d, 128 steps: instruction chains (130 per point-triangle distance, 13 per bound test)" % (w, h))
    print("%-76s %16s %14s %8s" % ("build", "longest wave, k", "all waves, M", "util"))
    ch, base = (C.c_ulonglong * 3)(), None
    for d, t, what in variants:
        e = T.Emulated(rmdf_amd, env, defines=d, tag=t)
        e.K.koh_take_chains(ch)
        e.render(0, w, h, 0.0, 128, planes=False)
        e.K.koh_take_chains(ch)
        mx, sm, us = [int(x) for x in ch]
        base = base or (mx, sm)
        print("%-76s %16.1f %14.1f %8.3f   longest %+.1f %%, total %+.1f %%" % (what[:76], mx / 1e3, sm / 1e6, us / (64.0 * sm), 100.0 * (mx - base[0]) / base[0], 100.0 * (sm - base[1]) / base[1]), flush=True)
    sys.exit(0)
emu = T.Emulated(rmdf_amd, env)
sched = (C.c_ulonglong * 4)()
print("headline frame %d x %d, 256 steps: Mandelbulb-estimate lane-slots (87 per pass + 100 per estimate; a wave pays its slowest lane x 64)" % (w, h))
print("%-44s %10s %12s %12s %8s %10s" % ("schedule", "seconds", "useful M", "issued M", "util", "wave-passes M"))
base = None
for name, nm in (("product: pooling at <= 32 live rays", 0), ("pooling at <= 16", 16), ("pooling at <= 8", 8), ("no pooling (RMDF_FLAG_NO_MERGE)", 1)):
    emu.K.koh_take_schedule(sched)
    t0 = time.time()
    out = emu.render(2, w, h, 0.0, 256, planes=False, no_merge=nm)
    dt = time.time() - t0
    emu.K.koh_take_schedule(sched)
    useful, slots, wp, lp = [int(x) for x in sched]
    if base is None:
        base = slots
    print("%-44s %10.1f %12.1f %12.1f %8.3f %10.2f   issued vs product %+.1f %%" % (name, dt, useful / 1e6 / 64, slots / 1e6 / 64, useful / slots, wp / 1e6, 100.0 * (slots - base) / base), flush=True)
