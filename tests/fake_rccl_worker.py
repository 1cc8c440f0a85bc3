#!/usr/bin/env python3
"""One rank of the exchange step's N > 1 tests on ONE GPU (tests/test_gpu_parity.py: test_exchange_with_*_ranks_against_the_rccl_double).
N of these processes share cuda:0; librmdf_xcheck.so loads tests/libfake_rccl.so instead of RCCL (RMDF_RCCL_LIB), so the peer branches of
rmdf_gather_shards_device / rmdf_comm_verify_deal / rmdf_render_frame_sharded_device run with real peers.
usage: fake_rccl_worker.py <rank> <nranks> <id file> <w> <h> <max_steps> <frames in flight>
Prints one line `rank r ok <sha256 of the last assembled frame or ->` and exits 0, or a traceback and 1."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                # noqa: E402  (before librmdf: one HIP runtime per process)
if "libfake_hip" in os.environ.get("LD_PRELOAD", "") and os.environ.get("FAKE_HIP_EMULATE", "0") not in ("", "0"):
    # the N ranks on EMULATED devices (the HIP double running the kernels' source: tests/test_emulated_gpu_tier.py): torch's device buffers and
    # streams are the stand-ins over the double
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_cuda_standins                              # noqa: E402
    torch_cuda_standins.install()
import rmdf_amd                                             # noqa: E402


def main():
    rank, n, idfile, w, h, ms, S = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    assert os.environ.get("RMDF_RCCL_LIB"), "the worker is for the test double only"
    sr = rmdf_amd.ShaderRenderer(0, xcheck=True)
    z = np.load(os.path.join(ROOT, "tests", "golden", "env_cubes_uffizi.npz"))
    for slot, k in ((rmdf_amd.ENV_REFLECTION, "refl"), (rmdf_amd.ENV_COS_1, "cos1"), (rmdf_amd.ENV_COS_8, "cos8")):
        sr.set_env_cube(slot, z[k].view(np.float16)[:, 1:-1, 1:-1, :3].astype(np.float32))
    # rank 0 draws the id through the library (rmdf_comm_get_unique_id -> the double's ncclGetUniqueId), the others read it
    if rank == 0:
        uid = rmdf_amd.comm_get_unique_id(xcheck=True)
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 120, "no unique id from rank 0"
            time.sleep(0.01)
        uid = open(idfile, "rb").read()
    sr.comm_init(uid, rank, n)
    assert sr.comm_info() == (rank, n)
    dev = torch.device("cuda", 0)
    slots = rmdf_amd.shard_slots(n)
    i32 = dict(dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    gath = [torch.zeros((n, slots, h // 8, w // 8), **i32) if rank == 0 else None for _ in range(S)]
    shard = [gath[k][0] if rank == 0 else torch.zeros((slots, h // 8, w // 8), **i32) for k in range(S)]
    frame = [torch.zeros((h, w), **i32) if rank == 0 else None for _ in range(S)]
    torch.cuda.synchronize(dev)     # the fills run on torch's stream; the library's non-blocking streams do not wait for them
    single = sr.render(2, w, h, 0.0, max_steps=ms, want_f32=False)["rgba8"] if rank == 0 else None

    def frames(times, tag, check=True):
        """len(times) frames in flight on one communicator, frame i on stream i % S; rank 0 checks each against the single launch"""
        for i, t in enumerate(times):
            k = i % S
            sr.render_frame_sharded_device(2, w, h, t, ms, shard[k].data_ptr(), gath[k].data_ptr() if rank == 0 else 0,
                                           frame[k].data_ptr() if rank == 0 else 0, stream=streams[k].cuda_stream)
        torch.cuda.synchronize(dev)
        if rank == 0 and check:
            for i, t in enumerate(times):
                if i >= len(times) - S and t == 0.0:
                    got = frame[i % S].cpu().numpy().view(np.uint32)
                    assert np.array_equal(got, single), "%s: frame %d differs from the single launch" % (tag, i)

    # 1. the static deal, S frames in flight twice over
    frames([0.0] * (2 * S), "static deal")
    # 2. a cost-aware deal: every rank probes the same costs (deterministic kernel), sets them, and the COLLECTIVE check agrees
    cost = sr.probe_tile_costs(2, w, h, 0.0, ms)
    sr.set_shard_costs(cost)
    sr.set_shard_root_handicap(0.25)
    sr.comm_verify_deal()
    mine = sr.shard_tiles(rank, n)
    assert 0 < len(mine) <= slots
    frames([0.0] * S, "verified cost-aware deal")
    # 3. ONE rank holds other costs: every rank gets RMDF_E_COMM from the check, nobody hangs
    bad = np.array(cost, np.float32).copy()
    if rank == n - 1:
        bad[::3] *= 7.0
    sr.set_shard_costs(bad)
    try:
        sr.comm_verify_deal()
        raise AssertionError("rmdf_comm_verify_deal accepted different deals")
    except rmdf_amd.RmdfError as e:
        assert e.code == -8 and "different tile deals" in str(e), str(e)
    # ... and the exchange itself still runs to the end: the sizes on the wire never depend on the deal (the frame is whatever the mixed deals give)
    frames([0.0], "mixed deals", check=False)
    # 4. back to one deal everywhere; verified again; a frame at another time in between
    sr.set_shard_costs(cost)
    sr.comm_verify_deal()
    frames([2.5] + [0.0] * S, "second verified deal")
    sha = hashlib.sha256(frame[0].cpu().numpy().tobytes()).hexdigest() if rank == 0 else "-"
    sr.comm_destroy()
    sr.close()
    print("rank %d ok %s" % (rank, sha), flush=True)


if __name__ == "__main__":
    main()
