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

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
orc.build()
rd = lambda fn: orc.hdr_decode(open(fn, "rb").read())
cache = os.path.join(ROOT, "tests", "golden", "env_cache")
ll = {"refl": rd(rmdf_amd.DEFAULT_ENV_HDR), "cos1": rd(cache + "/uffizi_512_cache_pow_1.0.hdr"), "cos8": rd(cache + "/uffizi_512_cache_pow_8.0.hdr")}
env = orc.EnvSet(*(orc.cube_pad_f16(orc.latlong_to_cube(ll[k])) for k in ("refl", "cos1", "cos8")))
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
