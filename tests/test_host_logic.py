"""CPU tier: host-side logic of the boundary (tile rectangles, shard bookkeeping) and the N>1 gather path
exercised with world_size-2 gloo processes."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_tile_rect_matches_the_ndc_quad(rmdf):
    """drawShaderTile's NDC rect [-1+2tx/8, -1+2(tx+1)/8] (ShaderRendering.hs:183-193) rasterised by pixel
    centres, in float64, for sizes that 8 does and does not divide."""
    for (w, h) in ((1920, 1080), (1280, 720), (64, 36), (100, 50), (37, 19), (8, 8)):
        cover = np.zeros((h, w), int)
        for idx in (0, 7, 8, 27, 63, 64 + 5):
            midx = idx % 64
            tx, ty = midx % 8, midx // 8
            x0n, x1n = -1 + tx / 8 * 2, -1 + (tx + 1) / 8 * 2
            y0n, y1n = -1 + ty / 8 * 2, -1 + (ty + 1) / 8 * 2
            xs = (np.arange(w) + 0.5) / w * 2 - 1
            ys = (np.arange(h) + 0.5) / h * 2 - 1
            inx = np.nonzero((xs >= x0n) & (xs < x1n))[0]
            iny = np.nonzero((ys >= y0n) & (ys < y1n))[0]
            x0, y0, x1, y1 = rmdf.tile_rect(idx, w, h)
            assert (x0, x1) == ((inx[0], inx[-1] + 1) if len(inx) else (x0, x0))
            assert (y0, y1) == ((iny[0], iny[-1] + 1) if len(iny) else (y0, y0))
        for idx in range(64):
            x0, y0, x1, y1 = rmdf.tile_rect(idx, w, h)
            cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all()


def test_shard_bookkeeping(rmdf):
    for n in (1, 2, 3, 4, 8, 64):
        tiles = [rmdf.shard_tiles(r, n) for r in range(n)]
        assert sorted(sum(tiles, [])) == list(range(64))          # a partition of the 64 tiles
        assert max(len(t) for t in tiles) == rmdf.shard_slots(n)
        assert max(len(t) for t in tiles) - min(len(t) for t in tiles) <= 1
    # load balance: the scenes are centred, so every rank gets near and far tiles: the mean squared
    # distance of a rank's tiles from the frame centre is about the same for all ranks
    d2 = lambda idx: (2 * (idx % 8) - 7) ** 2 + (2 * (idx // 8) - 7) ** 2
    for n in (2, 4, 8):
        means = [np.mean([d2(t) for t in rmdf.shard_tiles(r, n)]) for r in range(n)]
        assert max(means) - min(means) <= 0.1 * np.mean(means), (n, means)
    assert rmdf.shard_tiles(0, 8)[0] == 27                           # one of the four centre tiles first
    # the library's own deal (what the kernels use) is the same function
    for n in (1, 2, 3, 5, 8, 64):
        for r in range(n):
            assert rmdf.shard_tiles_abi(r, n) == rmdf.shard_tiles(r, n)


def test_assemble_shards_host(rmdf):
    w, h, n = 64, 32, 3
    tw, th = w // 8, h // 8
    frame = np.arange(w * h, dtype=np.uint32).reshape(h, w)
    gathered = np.zeros((n, rmdf.shard_slots(n), th, tw), np.uint32)
    for r in range(n):
        for slot, idx in enumerate(rmdf.shard_tiles(r, n)):
            tx, ty = idx % 8, idx // 8
            gathered[r, slot] = frame[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
    assert np.array_equal(rmdf.assemble_shards_host(gathered, w, h, n), frame)


def test_framebuffer_slot(rmdf):
    fb = rmdf.FrameBuffer(4, 3)
    assert (fb.vec == 0xFF000000).all()

    def filler(w, h, vec):
        vec[:] = np.arange(w * h, dtype=np.uint32) << 8          # green ramp, alpha 0 like Fractal2D
        return "done"
    assert fb.fill_frame_buffer(filler) == "done"
    img = fb.to_image_rows_top_down()
    assert img.shape == (3, 4, 4) and (img[..., 3] == 255).all()  # alpha forced to 0xFF
    assert img[0, 0, 1] == 8 and img[2, 0, 1] == 0                # row 0 of the buffer is the bottom row


def test_screenshot_png(rmdf, tmp_path):
    """rmdf_save_png == saveFrameBufferToPNG (FrameBuffer.hs:215-228): decoded with an independent PNG reader, the
    file holds the frame buffer flipped top-down with alpha 0xFF.  Host-only code: runs without a GPU."""
    from PIL import Image
    rmdf.build()
    rng = np.random.RandomState(3)
    for (w, h) in ((4, 3), (257, 131), (1, 1)):
        fb = rmdf.FrameBuffer(w, h)
        fb.vec[:] = rng.randint(0, 2 ** 32, w * h, dtype=np.uint64).astype(np.uint32)
        fn = str(tmp_path / ("shot_%dx%d.png" % (w, h)))
        fb.save_png(fn)
        im = Image.open(fn)
        assert im.mode == "RGBA" and im.size == (w, h)
        assert np.array_equal(np.asarray(im), fb.to_image_rows_top_down())
    with pytest.raises(rmdf.RmdfError) as e:
        rmdf.FrameBuffer(2, 2).save_png("/nonexistent_dir/x.png")
    assert e.value.code == -4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import rmdf_amd
    from oracle import orc
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h = 64, 32

        def render_shard(rank_, n_):
            # stand-in for the HIP shard render on the CPU tier: the ORACLE renders this rank's tiles
            env = bench.load_oracle_env(orc)
            out = np.zeros((rmdf_amd.shard_slots(n_), h // 8, w // 8), np.uint32)
            for slot, idx in enumerate(rmdf_amd.shard_tiles(rank_, n_)):
                x0, y0, x1, y1 = rmdf_amd.tile_rect(idx, w, h)
                r = orc.render(orc.SCENE_MB_POWER8, w, h, 0.0, 64, env, rect=(x0, y0, x1, y1), nthreads=2)
                out[slot] = r["rgba8"][y0:y1, x0:x1]
            return torch.from_numpy(out.view(np.int32))
        shard = render_shard(rank, world)
        gathered = bench.gather_shards(shard, rank, world, dist)
        if rank == 0:
            frame = rmdf_amd.assemble_shards_host(gathered.numpy().view(np.uint32), w, h, world)
            env = bench.load_oracle_env(orc)
            full = orc.render(orc.SCENE_MB_POWER8, w, h, 0.0, 64, env, nthreads=2)["rgba8"]
            q.put(bool(np.array_equal(frame, full)))
        # cost-aware deal (rmdf_set_shard_costs): every rank derives the same costs on its own (here: oracle step +
        # iteration counts of a tiny probe frame), deals LPT, the ranks confirm they agree, and the frame assembles
        probe = orc.render(orc.SCENE_MB_POWER8, 32, 16, 0.0, 64, bench.load_oracle_env(orc), nthreads=2)
        cost = (1.0 + probe["iters"].astype(np.float64) + (probe["steps"] & 0x7FFF)).reshape(8, 2, 8, 4).sum(axis=(1, 3)).ravel().astype(np.float32)
        deal = [rmdf_amd.shard_tiles_by_cost(r, world, cost) for r in range(world)]
        ok_deal = bench.ranks_agree_on_deal(deal, dist, torch.device("cpu"))
        bad = [list(d) for d in deal]
        if rank == 1:
            bad[0][0], bad[1][0] = bad[1][0], bad[0][0]
        ok_bad = bench.ranks_agree_on_deal(bad, dist, torch.device("cpu"))
        tiles_of = lambda r_, n_: deal[r_]
        env2 = bench.load_oracle_env(orc)
        out2 = np.zeros((rmdf_amd.shard_slots(world), h // 8, w // 8), np.uint32)
        for slot, idx in enumerate(deal[rank]):
            x0, y0, x1, y1 = rmdf_amd.tile_rect(idx, w, h)
            out2[slot] = orc.render(orc.SCENE_MB_POWER8, w, h, 0.0, 64, env2, rect=(x0, y0, x1, y1), nthreads=2)["rgba8"][y0:y1, x0:x1]
        g2 = bench.gather_shards(torch.from_numpy(out2.view(np.int32)), rank, world, dist)
        if rank == 0:
            frame2 = rmdf_amd.assemble_shards_host(g2.numpy().view(np.uint32), w, h, world, tiles_of=tiles_of)
            q.put(bool(ok_deal and not ok_bad and np.array_equal(frame2, full) and sorted(sum(deal, [])) == list(range(64))))
        t = bench.max_over_ranks(float(rank + 1), dist, torch.device("cpu"))
        if rank == 1:
            q.put(t == float(world))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_reassembles_the_frame():
    """world_size 2 over gloo: each rank produces its tile shard (static deal, then cost-aware deal confirmed by the
    agreement protocol), ONE gather at frame end, rank 0 scatters tiles to frame positions -> identical to a
    single-process full-frame render."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for attempt in range(3):                            # (the rendezvous port was free a moment ago; if somebody took it since, once more)
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
        if all(p.exitcode == 0 for p in procs) or attempt == 2:
            break
    assert [p.exitcode for p in procs] == [0, 0]
    results = [q.get(timeout=10) for _ in range(3)]     # static-deal frame, cost-aware-deal frame (+ agreement protocol), timing reduce
    assert results == [True, True, True]


def test_cornell_candidate_grid_built_by_halving_equals_brute_force():
    """rmdf_create builds the Cornell box's 64^3 candidate grid (rmdf_device.hpp: CORNELL_FINE_N) from the 16^3 one by halving cells,
    measuring in a child cell only the parent's candidate triangles.  Host arithmetic, no GPU: the result equals the grid in which
    every triangle is measured in every cell, every cell has a candidate, and a child's candidates are a subset of its parent's."""
    import ctypes as C
    import rmdf_amd
    L = rmdf_amd.load_library(xcheck=True)
    grids = {}
    for n in (16, 32, 64):
        a, b = np.zeros(n ** 3, np.uint32), np.zeros(n ** 3, np.uint32)
        assert L.rmdf_debug_cornell_masks(n, 0, a.ctypes.data) == 0 and L.rmdf_debug_cornell_masks(n, 1, b.ctypes.data) == 0
        assert np.array_equal(a, b), n
        assert (a != 0).all()
        grids[n] = a.reshape(n, n, n)
    for n in (32, 64):
        parent = np.repeat(np.repeat(np.repeat(grids[n // 2], 2, 0), 2, 1), 2, 2)
        assert ((grids[n] & ~parent) == 0).all(), n
    counts = {n: float(np.mean([bin(int(m)).count("1") for m in g.ravel()[::7]])) for n, g in grids.items()}
    assert counts[64] < counts[32] < counts[16], counts        # what the finer grid buys: fewer candidates per cell
    assert L.rmdf_debug_cornell_masks(48, 0, grids[16].ctypes.data) != 0     # only powers of two times 16



def test_cornell_pruning_planes_are_lower_bounds_of_the_distance():
    """The per-lane Cornell estimate skips a triangle when pd^2 + max(0, s_a, s_b, s_c)^2 -- its plane distance and the largest of its
    three edge-plane distances, from the table rmdf_create uploads -- exceeds the squared margin around the running minimum
    (rmdf_device.hpp: de_cornell_box_lanes).  That is only invisible if the expression never exceeds the true squared distance.  Host
    arithmetic, no GPU: the table's planes evaluated in float32 the way the kernel does, against a float64 point-triangle distance, on
    300 000 points of the grid's cube, on points near the triangles' own planes, edges and vertices, and the structural facts the bound
    rests on (unit normals, the triangle on the inner side of each edge plane, edge planes perpendicular to the triangle)."""
    import ctypes as C
    import rmdf_amd
    L = rmdf_amd.load_library(xcheck=True)
    stride, bounds = C.c_int(), C.c_int()
    assert L.rmdf_debug_cornell_table(None, C.byref(stride), C.byref(bounds)) == 0
    S, B = stride.value, bounds.value
    tab = np.zeros((32, S), np.float32)
    assert L.rmdf_debug_cornell_table(tab.ctypes.data, None, None) == 0
    tri = tab[:, :9].astype(np.float64).reshape(32, 3, 3)
    assert np.array_equal(tab[:, :9].reshape(96, 3), rmdf_amd.cornell_vertices().reshape(96, 3))
    planes = tab[:, B:B + 16].reshape(32, 4, 4)                      # [triangle][plane, edge a, b, c][nx, ny, nz, offset]
    nrm = planes[:, :, :3].astype(np.float64)
    assert np.allclose(np.linalg.norm(nrm, axis=2), 1.0, atol=1e-6)
    assert np.abs(np.einsum("tk,tek->te", nrm[:, 0], nrm[:, 1:])).max() < 1e-6          # edge planes stand on the triangle's plane
    for t in range(32):
        assert np.abs(tri[t] @ nrm[t, 0] - planes[t, 0, 3]).max() < 1e-6                 # the vertices lie in the plane
        for e in range(1, 4):
            assert (tri[t] @ nrm[t, e] - planes[t, e, 3]).max() <= 0.0                   # the triangle is on the inner side of every edge plane

    def true_dist(p):                                                                    # [n, 3] float64 -> [n, 32] (Ericson 5.1.5)
        a, b, c = tri[None, :, 0], tri[None, :, 1], tri[None, :, 2]
        P = p[:, None, :]
        ab, ac, ap = b - a, c - a, P - a
        d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
        bp, cp = P - b, P - c
        d3, d4, d5, d6 = (ab * bp).sum(-1), (ac * bp).sum(-1), (ab * cp).sum(-1), (ac * cp).sum(-1)
        vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
        n = p.shape[0]
        q, done = np.zeros((n, 32, 3)), np.zeros((n, 32), bool)
        A, Bv, Cv = (np.broadcast_to(x, (n, 32, 3)) for x in (a, b, c))
        def put(mask, val):
            m = mask & ~done
            q[m] = val[m]
            done[m] = True
        with np.errstate(all="ignore"):
            put((d1 <= 0) & (d2 <= 0), A)
            put((d3 >= 0) & (d4 <= d3), Bv)
            put((vc <= 0) & (d1 >= 0) & (d3 <= 0), A + (d1 / (d1 - d3))[..., None] * ab)
            put((d6 >= 0) & (d5 <= d6), Cv)
            put((vb <= 0) & (d2 >= 0) & (d6 <= 0), A + (d2 / (d2 - d6))[..., None] * ac)
            put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), Bv + ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[..., None] * (c - b))
            den = 1.0 / (va + vb + vc)
            put(np.ones((n, 32), bool), A + ab * (vb * den)[..., None] + ac * (vc * den)[..., None])
        return np.sqrt(((P - q) ** 2).sum(-1))

    def kernel_bound2(p32):                                                              # float32, the kernel's operations
        x, y, z = (p32[:, None, None, k] for k in range(3))
        pl = planes[None]
        f32 = np.float32
        dots = (f32(pl[..., 2] * z) + f32(f32(pl[..., 1] * y) + f32(pl[..., 0] * x))) - pl[..., 3]      # FMAs in the kernel: rounding differs by ulps
        pd, sm = dots[..., 0], np.maximum(dots[..., 1:].max(axis=-1), f32(0.0))
        return (sm * sm + pd * pd).astype(np.float64)

    rng = np.random.default_rng(7)
    pts = [rng.uniform(-1.1, 1.1, (300000, 3))]
    w = rng.dirichlet((1.0, 1.0, 1.0), (32, 400))                                         # points on the triangles, pushed off a little
    on = np.einsum("tnk,tkc->tnc", w, tri).reshape(-1, 3)
    pts.append(on + rng.normal(0.0, 1e-3, on.shape))
    pts.append(on + rng.normal(0.0, 5e-2, on.shape))
    pts.append(np.repeat(tri.reshape(-1, 3), 50, axis=0) + rng.normal(0.0, 2e-2, (96 * 50, 3)))   # around the vertices
    worst = 0.0
    for p in pts:
        for lo in range(0, len(p), 50000):
            q = p[lo:lo + 50000]
            d, b2 = true_dist(q), kernel_bound2(q.astype(np.float32))
            # the kernel compares with 1.0031 best + 1.02e-7 >= (1.001 sqrt(best) + 1e-5)^2: a bound may exceed the true squared distance
            # by rounding only, far inside that margin
            over = np.sqrt(b2) - d
            worst = max(worst, float(over.max()))
            assert over.max() < 2e-6, (float(over.max()), np.unravel_index(over.argmax(), over.shape))
    assert worst > -1.0                                                                   # (ran)


def test_cornell_kept_set_of_one_sample_point_serves_its_neighbours():
    """-DRMDF_AB_SHARED_BOUNDS (rmdf_device.hpp: de_cornell_box_lanes `keep`, de_cornell_box_lanes_kept; an A/B variant, not the
    product default): the normal's four sample points lie within 1e-5 of each other, so the first one's pass of bound tests, its margin
    widened to 3e-5, is to serve the other three, which then measure only the kept triangles.  Host arithmetic, no GPU: for 200 000
    points p (random and near the triangles) and every hint g, each triangle NOT kept at p -- outside the cell's mask, or with
    bound^2 > 1.0031 d(p, g)^2 + 8.3e-7 -- is farther from p - 1e-5 e_k than the nearest triangle by more than 5e-6, i.e. by far more than
    the float rounding of the distance formulas, so it cannot be the minimum there."""
    import ctypes as C
    import rmdf_amd
    L = rmdf_amd.load_library(xcheck=True)
    stride, bounds = C.c_int(), C.c_int()
    assert L.rmdf_debug_cornell_table(None, C.byref(stride), C.byref(bounds)) == 0
    S, B = stride.value, bounds.value
    tab = np.zeros((32, S), np.float32)
    assert L.rmdf_debug_cornell_table(tab.ctypes.data, None, None) == 0
    N = 64
    grid = np.zeros(N ** 3, np.uint32)
    assert L.rmdf_debug_cornell_masks(N, 0, grid.ctypes.data) == 0
    tri = tab[:, :9].astype(np.float64).reshape(32, 3, 3)
    planes = tab[:, B:B + 16].reshape(32, 4, 4)

    def true_dist(p):                                       # closest point by clamped projection onto plane / edges, float64: [n, 3] -> [n, 32]
        a, b, c = tri[None, :, 0], tri[None, :, 1], tri[None, :, 2]
        P = p[:, None, :]
        def seg(u, v):
            e = v - u
            t = np.clip(((P - u) * e).sum(-1) / (e * e).sum(-1), 0.0, 1.0)
            return np.sqrt(((P - (u + t[..., None] * e)) ** 2).sum(-1))
        n = np.cross(b - a, c - a)
        n = n / np.linalg.norm(n, axis=-1, keepdims=True)
        pd = ((P - a) * n).sum(-1)
        q = P - pd[..., None] * n
        inside = np.ones(pd.shape, bool)
        for u, v in ((a, b), (b, c), (c, a)):
            inside &= (np.cross(v - u, q - u) * n).sum(-1) >= 0.0
        edge = np.minimum(np.minimum(seg(a, b), seg(b, c)), seg(c, a))
        return np.where(inside, np.abs(pd), edge)

    def kernel_bound2(p32):
        x, y, z = (p32[:, None, None, k] for k in range(3))
        pl = planes[None]
        f32 = np.float32
        dots = (f32(pl[..., 2] * z) + f32(f32(pl[..., 1] * y) + f32(pl[..., 0] * x))) - pl[..., 3]
        pd, sm = dots[..., 0], np.maximum(dots[..., 1:].max(axis=-1), f32(0.0))
        return sm * sm + pd * pd

    def cell_mask(p32):
        s = np.float32(N / 2.2)
        idx = np.floor(p32 * s + np.float32(1.1) * s).astype(np.int64)
        ok = ((idx >= 0) & (idx < N)).all(axis=1)
        idx = np.clip(idx, 0, N - 1)
        m = grid[(idx[:, 2] * N + idx[:, 1]) * N + idx[:, 0]]
        return np.where(ok, m, np.uint32(0xffffffff))

    rng = np.random.default_rng(11)
    w = rng.dirichlet((1.0, 1.0, 1.0), (32, 1500))
    on = np.einsum("tnk,tkc->tnc", w, tri).reshape(-1, 3)
    pts = np.concatenate([rng.uniform(-1.05, 1.05, (100000, 3)), on + rng.normal(0.0, 2e-3, on.shape), on + rng.normal(0.0, 3e-5, on.shape),
                          np.repeat(tri.reshape(-1, 3), 40, axis=0) + rng.normal(0.0, 1e-3, (96 * 40, 3))])
    bits = (np.uint32(1) << np.arange(32, dtype=np.uint32))[None, :]
    kept_counts = []
    for lo in range(0, len(pts), 20000):
        p = pts[lo:lo + 20000]
        p32 = p.astype(np.float32)
        d0 = true_dist(p32.astype(np.float64))
        b2 = kernel_bound2(p32)
        in_cell = (cell_mask(p32)[:, None] & bits) != 0
        # hints: the nearest triangle (what the march hands over almost always) and a random one (nothing relies on the hint being good)
        for g in (d0.argmin(axis=1), rng.integers(0, 32, len(p))):
            best = (d0[np.arange(len(p)), g] ** 2).astype(np.float32)
            thr = np.float32(1.0031) * best + np.float32(8.3e-7)
            kept = (in_cell & (b2 <= thr[:, None])) | (np.arange(32)[None, :] == g[:, None])
            kept_counts.append(kept.sum(axis=1).mean())
            for k in range(3):
                q = p32.astype(np.float64)
                q[:, k] -= 1e-5
                d = true_dist(q)
                gap = np.where(kept, np.inf, d - d.min(axis=1, keepdims=True))
                assert gap.min() > 5e-6, (float(gap.min()), k)
    print("mean kept triangles per point (nearest hint, random hint) per block:", [round(float(x), 2) for x in kept_counts])
    assert 1.0 <= min(kept_counts) and kept_counts[0] < 4.0, kept_counts       # with a good hint the kept set is small


def test_host_built_env_tables_against_the_oracle():
    """The env-map kernels only GATHER through tables the library builds on the host (rmdf_api.cpp: cube_uv_table_host, lobe_tables_host);
    the arithmetic the reference does per texel -- cubeMapPixelToDir, worldToLocal, cartesianToSpherical, sphericalToEnvironmentUV
    (HDREnvMap.hs:76-87, CoordTransf.hs:35-70), pxToTheta / pxToPhi and the cosine tables of cosineConvolveHDREnvMap (HDREnvMap.hs:222-239)
    -- happens there.  No GPU: (1) the oracle's latLongHDREnvMapToCubeMap of a random image must be the oracle's pixelAtBilinear at
    the PRODUCT's (u, v) of every texel, for three face sizes (a wrong u or v picks other texels of a random image); (2) the lobe tables
    equal the formulas restated in numpy float32 with the process's own libm cosf / sinf, for a 256-wide, a ragged and a tiny map."""
    import ctypes as C
    import rmdf_amd
    from oracle import orc
    L = rmdf_amd.load_library(xcheck=True)
    rng = np.random.default_rng(5)
    for (lw, lh) in ((96, 48), (51, 25), (512, 256)):
        cw = lw // 3
        latlong = rng.uniform(0.0, 5.0, (lh, lw, 3)).astype(np.float32)
        uv = np.zeros((6, cw, cw, 2), np.float32)
        assert L.rmdf_debug_cube_uv_table(cw, uv.ctypes.data) == 0
        assert (uv >= 0.0).all() and (uv <= 1.0).all()
        faces = orc.latlong_to_cube(latlong)
        assert faces.shape == (6, cw, cw, 3)
        step = 1 if cw <= 32 else 7                                       # (every texel of the small faces, a lattice of the large one)
        for f in range(6):
            for y in range(0, cw, step):
                for x in range(0, cw, step):
                    got = orc.pixel_at_bilinear(latlong, float(uv[f, y, x, 0]), float(uv[f, y, x, 1]))
                    assert np.array_equal(np.asarray(got, np.float32).view(np.uint32), faces[f, y, x].view(np.uint32)), (lw, f, y, x)
    assert L.rmdf_debug_cube_uv_table(0, uv.ctypes.data) != 0
    libm = C.CDLL("libm.so.6")
    libm.cosf.restype = libm.sinf.restype = C.c_float
    libm.cosf.argtypes = libm.sinf.argtypes = [C.c_float]
    f32 = np.float32
    pi = f32(3.14159265358979323846)
    for (w, h) in ((256, 128), (100, 37), (4, 2)):
        nblk = (w + 63) // 64
        lut, tcs = np.zeros((nblk, w, 64), np.float32), np.zeros((h, 2), np.float32)
        assert L.rmdf_debug_lobe_tables(w, h, lut.ctypes.data, tcs.ctypes.data) == 0
        phi = (np.arange(w, dtype=np.float32) / f32(w - 1) * f32(2.0) * pi).astype(np.float32)        # pxToPhi, HDREnvMap.hs:226
        th = (np.arange(h, dtype=np.float32) / f32(h - 1) * pi).astype(np.float32)                    # pxToTheta, :225
        for y in range(h):
            assert tcs[y, 0] == f32(libm.cosf(float(th[y]))) and tcs[y, 1] == f32(libm.sinf(float(th[y]))), (w, h, y)
        for dx in sorted(set(list(range(0, w, max(1, w // 9))) + [w - 1])):
            want = np.array([libm.cosf(float(np.abs(f32(phi[dx] - phi[x])))) for x in range(w)], np.float32)   # absPhiDiffCosLookup, :231
            assert np.array_equal(lut[dx // 64, :, dx % 64], want), (w, h, dx)
        if w % 64:                                                        # lanes past the last column repeat it (they are never stored)
            assert np.array_equal(lut[nblk - 1, :, 63], lut[nblk - 1, :, (w - 1) % 64])


def test_host_camera_equals_the_oracles_at_many_times():
    """a1: the camera block of main() and lookat (fragment.shd:829-838, 883-902) is evaluated once per frame on the host
    (rmdf_api.cpp: host_camera) -- the GPU tier compares it with the oracle at four times per scene; here, without a GPU, at 3000
    times per FragmentShader value (the orbit's period and far beyond, negative times, huge times), bit for bit, with tan(hfov / 2)."""
    import ctypes as C
    import rmdf_amd
    from oracle import orc
    L = rmdf_amd.load_library(xcheck=True)
    rng = np.random.default_rng(3)
    times = np.concatenate([np.linspace(0.0, 4.0 * np.pi * 6.0, 1500), rng.uniform(-1e3, 1e3, 1000), rng.uniform(-1e7, 1e7, 496),
                            [0.0, -0.0, 1e-30, 3.0e9]]).astype(np.float32)
    cam, fov = np.zeros(12, np.float32), C.c_float()
    for scene in range(4):
        for t in times:
            assert L.rmdf_debug_camera(scene, float(t), cam.ctypes.data, C.byref(fov)) == 0
            want = np.asarray(orc.camera(scene, float(t)), np.float32).ravel()
            assert np.array_equal(cam.view(np.uint32), want.view(np.uint32)), (scene, float(t), cam, want)
        assert np.float32(fov.value) == np.float32(orc.fov_xs())
    assert L.rmdf_debug_camera(4, 0.0, cam.ctypes.data, None) != 0


def test_host_radiance_reader_and_writer_against_the_oracle():
    """rmdf_load_env_hdr reads the light probe and reads / writes the *_cache_pow_*.hdr files with host code of the library's own
    (rmdf_api.cpp: decode_hdr, encode_hdr: loadHDRImage HDREnvMap.hs:31-52, JP.saveRadianceImage ShaderRendering.hs:147).  No GPU:
    the shipped light probe and a run-length coded file decode to the oracle's floats bit for bit; random images -- zeros, denormals, huge
    values, negative channels, one dominant channel -- encode to the oracle's bytes; and the reader, which parses bytes from disk, gets
    3000 truncated and corrupted files: it must agree with the oracle's reader on accept / reject and on every float, and never crash."""
    import ctypes as C
    import rmdf_amd
    from oracle import orc
    L = rmdf_amd.load_library(xcheck=True)

    def decode(data):
        buf = np.frombuffer(data, np.uint8) if len(data) else np.zeros(1, np.uint8)
        w, h = C.c_int(), C.c_int()
        if L.rmdf_debug_hdr_decode(buf.ctypes.data, len(data), C.byref(w), C.byref(h), None, 0) != 0:
            return None
        out = np.empty((h.value, w.value, 3), np.float32)
        assert L.rmdf_debug_hdr_decode(buf.ctypes.data, len(data), C.byref(w), C.byref(h), out.ctypes.data, out.size) == 0
        return out

    def oracle_decode(data):
        try:
            return orc.hdr_decode(data)
        except ValueError:
            return None

    same = lambda a, b: (a is None and b is None) or (a is not None and b is not None and a.shape == b.shape
                                                       and np.array_equal(a.view(np.uint32), b.view(np.uint32)))
    probe = open(rmdf_amd.DEFAULT_ENV_HDR, "rb").read()
    a = decode(probe)
    assert a is not None and a.shape == (256, 512, 3) and same(a, orc.hdr_decode(probe))
    # writer: random images of every kind the cache files can hold
    rng = np.random.default_rng(9)
    imgs = [rng.uniform(0.0, 4.0, (7, 13, 3)), np.zeros((2, 8, 3)), rng.uniform(0.0, 1e-36, (3, 9, 3)), rng.uniform(0.0, 3e38, (3, 9, 3)),
            rng.normal(0.0, 1.0, (5, 8, 3)), np.exp(rng.uniform(-80.0, 80.0, (16, 16, 3))), rng.uniform(0.0, 1.0, (4, 8, 3)) * [1e6, 1.0, 1e-6]]
    for img in imgs:
        img = img.astype(np.float32)
        h, w, _ = img.shape
        out = np.zeros(128 + 4 * w * h, np.uint8)
        n = L.rmdf_debug_hdr_encode(img.ctypes.data, w, h, out.ctypes.data, out.size)
        assert n > 0 and out[:n].tobytes() == orc.hdr_encode(img), img.shape
        assert same(decode(out[:n].tobytes()), orc.hdr_decode(out[:n].tobytes()))
    assert L.rmdf_debug_hdr_encode(imgs[0].astype(np.float32).ctypes.data, 13, 7, out.ctypes.data, 10) < 0          # no room
    # a run-length coded file (runs, literals, a run of 127) and its flat twin
    w, h = 200, 4
    rgbe = rng.integers(0, 255, (h, w, 4)).astype(np.uint8)
    rgbe[1, 10:170] = rgbe[1, 10]
    rgbe[2, :, 3] = 130
    header = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)
    rle = bytearray(header)
    for y in range(h):
        rle += bytes([2, 2, w >> 8, w & 255])
        for ch in range(4):
            row, x = rgbe[y, :, ch], 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 3:
                    rle += bytes([128 + run, int(row[x])]); x += run
                else:
                    lit = min(w - x, 5)
                    rle += bytes([lit]) + row[x:x + lit].tobytes(); x += lit
    flat = header + rgbe.tobytes()
    assert same(decode(flat), orc.hdr_decode(flat)) and same(decode(bytes(rle)), decode(flat)) and same(decode(bytes(rle)), orc.hdr_decode(bytes(rle)))
    # corrupted and truncated files: same verdict, same floats, no crash
    accepted = 0
    for base in (flat, bytes(rle)):
        for i in range(1500):
            m = bytearray(base)
            kind = i % 5
            if kind == 0:
                m = m[:rng.integers(0, len(m))]
            elif kind == 1:
                for _ in range(rng.integers(1, 4)):
                    m[rng.integers(0, len(m))] = rng.integers(0, 256)
            elif kind == 2:                                              # damage in the header / resolution line
                m[rng.integers(0, len(header))] = rng.integers(0, 256)
            elif kind == 3:
                pos = rng.integers(len(header), len(m))
                m[pos:pos] = bytes(rng.integers(0, 256, rng.integers(1, 9)).astype(np.uint8))
            else:
                pos = rng.integers(len(header), len(m) - 8)
                del m[pos:pos + rng.integers(1, 8)]
            got, want = decode(bytes(m)), oracle_decode(bytes(m))
            assert same(got, want), (i, kind, None if got is None else got.shape, None if want is None else want.shape)
            accepted += got is not None
    assert 100 < accepted < 2900, accepted                               # both verdicts occur
    for junk in (b"", b"\n", b"#?RADIANCE\n\n-Y 0 +X 0\n", b"#?RADIANCE\n\n-Y 70000 +X 70000\n" + b"x" * 64, b"#?RADIANCE\n\n+X 4 -Y 4\n" + b"x" * 64):
        assert same(decode(junk), oracle_decode(junk)), junk


def _fake_hip_lib():
    """tests/libfake_hip.so, built from tests/fake_hip.cpp when missing or older than its sources (hipcc, host code only)."""
    import subprocess
    from conftest import ROOT
    so, src = os.path.join(ROOT, "tests", "libfake_hip.so"), os.path.join(ROOT, "tests", "fake_hip.cpp")
    csrc = os.path.join(ROOT, "ray-marching-distance-fields_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("rmdf_internal.hpp", "rmdf_device.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-g", "-std=c++17", "-fPIC", "--cuda-host-only", "--offload-arch=gfx950", "-x", "hip",
                               "-I", csrc, "-shared", src, "-o", so])
    return so


@pytest.mark.parametrize("which", ["product", "xcheck", "xcheck-async"])
def test_host_paths_against_the_hip_double(which):
    """Everything librmdf does on the HOST behind a ctx, run on this box without a GPU: the HIP runtime is replaced by a test double
    (tests/fake_hip.cpp, LD_PRELOADed into a child process) whose "device memory" is malloc'd memory and whose "kernels" write a hash
    of (pixel, frame, scene, camera, step limit, cube-map contents) where the real kernels write colours.  tests/fake_hip_workload.py then
    asks for frames in every way the ABI offers and checks the host's work: all four whole-frame hand-overs x band counts == the
    plane-writing single launch, canaries around caller buffers intact; 64 tiles in order and permuted == the frame; an environment
    swapped between two tile calls shows from that tile on (tiles issued ahead never show the old one); shards of 1, 2, 3, 8, 64 ranks,
    static and cost-aware deal, assembled == the frame; supersampling == a numpy box filter; the env pipeline's shapes and its cache
    files (written once, read the second time, a damaged one is an error); argument errors; 1700 random calls -- tiles in any order,
    whole frames, plane-writing calls, size / shader / environment changes -- against a model of what the boundary promises
    (latching, accumulation, clearing); every ctx entry point with null pointers, unusable paths and out-of-range scalars (errors, no
    crash); no device / page-locked allocation,
    stream or event left after rmdf_destroy; an error -- never a crash, never a leak -- when the n-th allocation fails, for n = 1..59;
    and, with the cross-check library, the N-rank exchange (2, 3, 8 ranks as threads, the RCCL double) with equal and unequal deals.
    The SHIPPED librmdf.so's host code is what runs ("product"); tools/asan_host.sh runs the same under AddressSanitizer + UBSan and
    under ThreadSanitizer.  "xcheck-async": the double's asynchronous mode -- every stream a worker thread with an in-order queue, events
    with generations, hipErrorNotReady while work is pending, every queued operation delayed by a random 0..150 us -- in which a host
    that reads a buffer before the event guarding it computes with stale data and fails the comparisons
    (test_the_hip_double_notices_a_host_that_does_not_wait shows that it does).
    Nothing here is evidence about a kernel -- the double says so at length."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), FAKE_RCCL_TIMEOUT_S="120", OMP_NUM_THREADS="1")
    env.pop("RMDF_LIB", None)
    args = [sys.executable, os.path.join(ROOT, "tests", "fake_hip_workload.py")]
    if which == "xcheck-async":
        env.update(FAKE_HIP_ASYNC="1", FAKE_HIP_JITTER_US="150", FAKE_HIP_WORKLOAD_WATCHDOG_S="600")
    if which.startswith("xcheck"):
        fake_rccl = os.path.join(ROOT, "tests", "libfake_rccl.so")
        if not os.path.exists(fake_rccl) or os.path.getmtime(fake_rccl) < os.path.getmtime(os.path.join(ROOT, "tests", "fake_rccl.c")):
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "fake_rccl.c"),
                                   "-o", fake_rccl, "-L/opt/rocm/lib", "-lamdhip64"])
        env["RMDF_RCCL_LIB"] = fake_rccl
        args.append("xcheck")
    r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    oks = [l for l in r.stdout.splitlines() if l.startswith("ok ")]
    assert len(oks) == (9 if which.startswith("xcheck") else 8) and r.stdout.strip().splitlines()[-1].startswith("done:"), r.stdout[-1500:]
    assert " 0 launches of kernels it has no stand-in for" in r.stdout


def test_the_hip_double_notices_a_host_that_does_not_wait():
    """The test of the test above: with FAKE_HIP_SABOTAGE=events the double's hipEventQuery / hipEventSynchronize claim completion at
    once -- what a host that forgot to wait for its band, tile or staging event would see -- and the asynchronous workload must FAIL
    (stale rows reach the caller's frame), where the same run without the sabotage passes."""
    import subprocess
    import sys
    from conftest import ROOT
    base = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), FAKE_HIP_ASYNC="1", FAKE_HIP_JITTER_US="200", FAKE_HIP_WORKLOAD_WATCHDOG_S="300", OMP_NUM_THREADS="1")
    base.pop("RMDF_LIB", None)
    args = [sys.executable, os.path.join(ROOT, "tests", "fake_hip_workload.py"), "quick", "only=whole", "only=tiles"]
    good = subprocess.run(args, env=base, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert good.returncode == 0 and good.stdout.count("\nok ") + good.stdout.startswith("ok ") == 2, (good.stdout[-800:], good.stderr[-2000:])
    bad = subprocess.run(args, env=dict(base, FAKE_HIP_SABOTAGE="events"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert bad.returncode != 0 and "AssertionError" in bad.stderr, (bad.stdout[-800:], bad.stderr[-2000:])


@pytest.mark.parametrize("mode", ["end", "start"])
def test_guarded_allocations_against_the_hip_double(mode):
    """The cross-check build's electric-fence device allocator (rmdf_host.hpp: GuardAlloc, RMDF_GUARD_ALLOC=end|start; written in round 5
    and not yet run on hardware) on the HIP double, whose virtual-memory calls are mmap / mprotect: every device allocation of the library
    ends (or starts) at an inaccessible page.  tests/guard_workload.py -- every kernel of the library at sizes that round to nothing
    convenient -- then checks the HOST's buffer sizes against what the kernels' stand-ins read and write (one element outside is
    SIGSEGV), and the allocator's own bookkeeping over hundreds of allocations and frees."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), RMDF_GUARD_ALLOC=mode)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "guard_workload.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "guard workload ok" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-3000:])


def test_the_fence_of_the_hip_double_does_catch_an_overrun():
    """... and the fence is real: the library's resolve kernel (its exact stand-in), told that the source frame is two rows taller than
    the guarded buffer it is given, kills the child; with the true height it survives."""
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import os, sys; sys.path.insert(0, %r); import rmdf_amd, ctypes as C\n"
            "sr = rmdf_amd.ShaderRenderer(0, xcheck=True)\n"
            "L = rmdf_amd.load_library(True); p = C.c_void_p(); q = C.c_void_p()\n"
            "assert L.rmdf_device_malloc(sr.handle, 256 * 64 * 4, C.byref(p)) == 0 and L.rmdf_device_malloc(sr.handle, 128 * 33 * 4, C.byref(q)) == 0\n"
            "sr.resolve_box2_device(p.value, 256, 64, q.value); sr.synchronize(); print('before', flush=True)\n"
            "sr.resolve_box2_device(p.value, 256, int(sys.argv[1]), q.value)\n"
            "sr.synchronize(); print('survived', flush=True)\n" % ROOT)
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), RMDF_GUARD_ALLOC="end")
    ok = subprocess.run([sys.executable, "-c", code, "64"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0 and "survived" in ok.stdout, (ok.returncode, ok.stdout, ok.stderr[-1500:])
    r = subprocess.run([sys.executable, "-c", code, "66"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert "before" in r.stdout and "survived" not in r.stdout and r.returncode < 0, (r.returncode, r.stdout, r.stderr[-1500:])


def test_the_hip_double_covers_every_runtime_symbol_the_libraries_import():
    """A HIP call the double does not define would fall through to the real runtime in the middle of a doubled process: every
    `hip*` / `__hip*` symbol librmdf.so, librmdf_xcheck.so and the RCCL double import must be defined by tests/libfake_hip.so."""
    import subprocess
    import rmdf_amd
    from conftest import ROOT
    have = {l.split()[-1].split("@")[0] for l in subprocess.run(["nm", "-D", "--defined-only", _fake_hip_lib()], capture_output=True, text=True, check=True).stdout.splitlines() if l.strip()}
    for lib in (rmdf_amd.LIB_PATH, rmdf_amd.XCHECK_LIB_PATH, os.path.join(ROOT, "tests", "libfake_rccl.so")):
        if not os.path.exists(lib):
            continue
        need = {l.split()[-1].split("@")[0] for l in subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout.splitlines()
                if l.strip() and l.split()[-1].startswith(("hip", "__hip"))}
        assert need and not (need - have), (lib, sorted(need - have))


def test_gpu_tier_tests_that_need_no_oracle_run_against_the_hip_double():
    """Seventeen tests of the GPU tier compare the library with ITSELF -- tiles with the full frame, band hand-overs with the plane-writing
    launch, tiles issued ahead with what was asked for, a swapped environment, a registered buffer, the error convention, 3000 random boundary calls
    against tools/tile_mode_fuzz.py's model -- and so hold against the HIP double as well.  Running them here, on the CPU tier, checks the HOST paths they cover AND the
    tests' own code before it meets a GPU (including the band-test parameters for the one-launch hand-over, which have not run on
    hardware).  The shipped light probe is loaded from a private copy (tests/conftest.py), no stand-in cache file reaches the tree."""
    import subprocess
    import sys
    import rmdf_amd
    from conftest import ROOT
    names = ("test_tiled_frame_equals_full_frame or test_tile_jobs_issued_ahead_never_show or test_tile_jobs_issued_ahead_belong_to_one_environment "
             "or test_whole_frame_host_call_in_row_bands or test_tile_mode_copy_thread_counts or test_tile_mode_fuzz or test_determinism "
             "or test_error_convention or test_fresh_frame_is_cleared_to_opaque_black or test_registered_host_buffer")
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), RMDF_TEST_UNVERIFIED="1")
    env.pop("RMDF_LIB", None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k", names],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    tail = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, (r.stdout[-3000:], r.stderr[-1000:])
    assert int(tail.split(" passed")[0].split()[-1]) >= 18, tail
    data = os.path.dirname(rmdf_amd.DEFAULT_ENV_HDR)
    assert not [f for f in os.listdir(data) if "_cache_pow_" in f], "stand-in cache files next to the shipped light probe"


def test_the_library_without_a_device_still_fails_loudly():
    """... and without the double nothing has changed: on a box without a GPU rmdf_create fails with RMDF_E_NO_DEVICE and a message --
    there is no CPU rendering path, and the HIP double is not something the library can find by itself."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    code = ("import sys; sys.path.insert(0, %r); import rmdf_amd\n"
            "try:\n    rmdf_amd.ShaderRenderer(0)\n    print('created')\n"
            "except rmdf_amd.RmdfError as e:\n    print('error', e.code, e)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        assert "created" in r.stdout
    else:
        assert r.stdout.startswith("error") and "no HIP device" in r.stdout, (r.stdout, r.stderr[-500:])


def test_product_kernels_keep_their_register_budgets(tmp_path):
    """Occupancy is part of the measured figures (eight waves per SIMD for the headline kernel, six for the Cornell
    box, no private segment in either) and nothing else in the CPU tier would notice a compiler, flag or source change that costs a
    wave or starts spilling in a hot kernel.  The kernel descriptors of the BUILT librmdf.so (llvm-objdump --offloading, llvm-readelf
    --notes; no GPU): register counts, private segment, LDS of the product variants, and the same table as profiles/r05_kernel_resources.txt
    holds for the round's sources."""
    import re
    import shutil
    import subprocess
    import rmdf_amd
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("no llvm-objdump / llvm-readelf")
    lib = str(tmp_path / "librmdf.so")
    shutil.copy(rmdf_amd.LIB_PATH, lib)                                   # (--offloading extracts next to its input)
    subprocess.run([objdump, "--offloading", lib], check=True, capture_output=True, cwd=str(tmp_path), timeout=120)
    kernels = {}
    for f in sorted(os.listdir(str(tmp_path))):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", str(tmp_path / f)], check=True, capture_output=True, text=True, timeout=120).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, blk).group(1)
            kernels[g("name")] = {"vgpr": int(g("vgpr_count")), "sgpr": int(g("sgpr_count")), "scratch": int(g("private_segment_fixed_size")),
                                  "lds": int(g("group_segment_fixed_size")), "wg": int(g("max_flat_workgroup_size"))}
    k = lambda scene, merge, out: kernels["_ZN4rmdf8k_renderILi%dELb%dELi%dEEEvNS_11FrameParamsE" % (scene, merge, out)]
    # 4 scenes x {pooled, not} x 3 outputs, minus the Cornell box's pooled three (never launched, and the only kernels that would spill: not built)
    assert len([n for n in kernels if "k_render" in n]) == 21, sorted(kernels)
    assert not [n for n in kernels if "k_renderILi0ELb1" in n]
    assert all(d["scratch"] == 0 for n, d in kernels.items() if "k_render" in n), {n: d["scratch"] for n, d in kernels.items() if d["scratch"]}    # no render kernel of the product has a private segment
    # waves per SIMD = 512 // (VGPRs rounded up to 8): the headline kernel and its mirror-store variant at eight, no private segment
    for out in (0, 1):
        h = k(2, 1, out)
        assert h["vgpr"] <= 64 and h["scratch"] == 0 and h["wg"] == 256, h
        assert h["lds"] <= 20 * 1024, h                                  # eight workgroups of four waves per CU: 8 x LDS <= 160 KB
    # the Cornell box as the product launches it (no pooling): six waves, no private segment
    for out in (0, 1):
        c = k(0, 0, out)
        assert c["vgpr"] <= 80 and c["scratch"] == 0, c
    # test scene and general-power Mandelbulb (pooled): seven waves or better, no private segment since round 5 (no calls left)
    for scene in (1, 3):
        for out in (0, 1):
            t = k(scene, 1, out)
            assert t["vgpr"] <= 72 and t["scratch"] == 0, (scene, t)
    # the lobe prefilter's producer / summing-wave kernels are launched with 11 and 12 waves per workgroup: they must fit 80 VGPRs
    for n, d in kernels.items():
        if "k_prefilter_chan" in n or "k_prefilter_fused4" in n or "k_prefilter_ring" in n:
            assert d["vgpr"] <= 80, (n, d)


def test_dpp_products_of_the_prefilter_have_no_read_after_write_hazard(tmp_path):
    """k_prefilter_chan / k_prefilter_fused4 multiply source texels through `v_mul_f32_dpp ... row_newbcast` written as inline asm
    (hipcc does not fold update_dpp into the multiply).  The compiler's hazard recogniser does not look inside inline asm, and gfx9
    requires two wait states between a VALU write of a VGPR and a DPP read of it.  The DPP operand is always a register loaded from
    LDS (a waitcnt, not a VALU write, precedes its use); this test compiles rmdf_env.hip to assembly with the product's flags and
    checks that it stays that way: no vector instruction writes a DPP source in the two slots before the multiply."""
    import re
    import shutil
    import subprocess
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "ray-marching-distance-fields_amd", "csrc")
    out = str(tmp_path / "env.s")
    # the product's own flags, read from csrc/Makefile (CXXFLAGS and, should the file ever get flags of its own, FLAGS_rmdf_env)
    mk = open(os.path.join(csrc, "Makefile")).read().replace("\\\n", " ")
    var = lambda name: (re.search(r"^%s\s*=\s*(.*)$" % name, mk, re.M) or [None, ""])[1]
    flags = (var("CXXFLAGS") + " " + var("FLAGS_rmdf_env")).replace("$(ARCH)", "gfx950").split()
    assert "-ffp-contract=off" in flags and "--offload-arch=gfx950" in flags, flags
    subprocess.run([hipcc] + flags + ["-Wno-unused-function", "--cuda-device-only", "-S", os.path.join(csrc, "rmdf_env.hip"), "-o", out],
                   check=True, capture_output=True, timeout=600)
    ins = []
    for l in open(out):
        t = l.split(";")[0].strip()
        if t and not t.startswith((".", "//")) and not t.endswith(":"):
            ins.append(t)
    n_dpp = 0
    for i, t in enumerate(ins):
        m = re.match(r"v_mul_f32_dpp\s+v\d+,\s*v(\d+),", t)
        if not m:
            continue
        n_dpp += 1
        src0 = int(m.group(1))
        for back in (1, 2):
            d = re.match(r"v_\w+\s+v(?:\[(\d+):(\d+)\]|(\d+))", ins[i - back])
            if d:
                lo, hi = (int(d.group(1)), int(d.group(2))) if d.group(1) else (int(d.group(3)), int(d.group(3)))
                assert not (lo <= src0 <= hi), (ins[i - back], t)
    assert n_dpp > 1000                                      # both kernels are there, fully unrolled


def test_bench_dry_run_against_the_hip_double(tmp_path):
    """bench.py's whole single-GPU flow without a GPU (tests/bench_dry_run.py): the HIP runtime is the test double, the few torch.cuda entry
    points bench.py uses are stand-ins over it.  Every number in the line is meaningless; that the line gets BUILT is the point -- the
    contract's keys, the roofline object (with the counters of the previous build reported apart), the host-call leg, every secondary
    workload with its per-scene roofline (round 5's additions, which have not met a GPU), no `secondary_error`."""
    import json
    import shutil
    import subprocess
    import sys
    import rmdf_amd
    from conftest import ROOT
    hdr = str(tmp_path / "probe.hdr")
    shutil.copy(rmdf_amd.DEFAULT_ENV_HDR, hdr)                 # (stand-in cache files stay out of the tree)
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), RMDF_ENV_HDR=hdr)
    for k in ("RMDF_LIB", "RMDF_FLAGS", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_dry_run.py"), "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline",
                        "--pmc", "off"],            # (the live counter passes have a test of their own, with a stand-in rocprofv3)
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f32" and d["metric"].startswith("Mpixels/s") and "workload" in d["config"]
    assert d["roofline"]["bound"] == "valu" and d["roofline"]["peak"] == 78.6 and "frac" in d["roofline"] and "traffic" in d["roofline"]
    ai = d["roofline"]["achievable_issue"]                    # round 6: the kernel's ISA priced in issue slots, beside the as-written roof
    assert ai["vector_instructions_per_iteration_pass"] == 87 and 95 < ai["vector_slots_per_iteration_pass"] < 102 and ai["frac_of_spec"] > 0
    assert "secondary_error" not in d and d["secondary"], d.get("secondary_error")
    for name in ("config2_cornell_1280x720_m128", "scene1_detest_1280x720_m128", "scene3_mbgeneral_1280x720_m128"):
        rl = d["secondary"][name]["roofline"]
        assert rl["frac"] is not None and rl["ops_per_launch"] > 0 and rl["formula"], (name, rl)
    assert d["d2h_inclusive_mpixels_s"] > 0 and d["d2h_inclusive_registered_buffer_mpixels_s"] is None
    assert not [f for f in os.listdir(os.path.dirname(rmdf_amd.DEFAULT_ENV_HDR)) if "_cache_pow_" in f and os.path.dirname(rmdf_amd.DEFAULT_ENV_HDR) != str(tmp_path)]


def test_bench_n1_line_carries_the_exchange_fields(tmp_path):
    """Round 6: the N = 1 line carries the N > 1 line's exchange fields -- rccl_ranks, shard_render_ms with min / max,
    exchange_ms -- measured after the timed region through the library's sharded-frame calls on a ONE-rank communicator, and says whether that
    frame equals the plain launch's.  Here: HIP double + RCCL double (librmdf_xcheck.so); with the real RCCL and no GPU the leg fails and the
    line says so instead of dying (the first dry-run test sees that branch)."""
    import json
    import shutil
    import subprocess
    import sys
    import rmdf_amd
    from conftest import ROOT
    from test_gpu_parity import _fake_rccl_lib
    hdr = str(tmp_path / "probe.hdr")
    shutil.copy(rmdf_amd.DEFAULT_ENV_HDR, hdr)
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), RMDF_ENV_HDR=hdr, RMDF_BENCH_SHARE_GPU="1", RMDF_RCCL_LIB=_fake_rccl_lib(), FAKE_RCCL_TIMEOUT_S="60")
    for k in ("RMDF_LIB", "RMDF_FLAGS", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_dry_run.py"), "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline",
                        "--width", "640", "--height", "360"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    c = json.loads(r.stdout.strip().splitlines()[-1])["config"]
    assert "one_rank_exchange_error" not in c, c.get("one_rank_exchange_error")
    assert c["rccl_ranks"] == 1 and c["sharded_frame_equals_plain_frame"] is True
    for k in ("exchange_ms", "shard_render_ms", "shard_render_ms_min", "shard_render_ms_max", "sharded_frame_ms"):
        assert c[k] is not None and c[k] >= 0, k


@pytest.mark.parametrize("flags", ["--supersample 1 --width 960 --height 540", "--streams 1", "--animate 0.1 --streams 2",
                                   "--scene 0 --width 1280 --height 720 --max-steps 128", "--scene 3 --no-animated"])
def test_bench_dry_run_of_the_other_modes(tmp_path, flags):
    """bench.py's other modes (supersampled config 4, one stream, an animated camera, the other scenes) through the same dry run: each
    builds its line."""
    import json
    import shutil
    import subprocess
    import sys
    import rmdf_amd
    from conftest import ROOT
    hdr = str(tmp_path / "probe.hdr")
    shutil.copy(rmdf_amd.DEFAULT_ENV_HDR, hdr)
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), RMDF_ENV_HDR=hdr)
    for k in ("RMDF_LIB", "RMDF_FLAGS", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_dry_run.py"), "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline",
                        "--no-secondary"] + flags.split(), cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "Traceback" not in r.stderr, (r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "workload" in d["config"] and d["roofline"]["bound"] == "valu"


@pytest.mark.parametrize("nranks", [2, 8])
def test_bench_dry_run_with_n_ranks_against_the_doubles(tmp_path, nranks):
    """bench.py the way the driver launches N > 1 -- `python -m torch.distributed.run --nproc-per-node N ... --gpus N` -- without a GPU:
    every rank runs tests/bench_dry_run.py (HIP double + torch.cuda stand-ins), the control plane is gloo (RMDF_BENCH_SHARE_GPU=1) and
    the exchange is the library's own over the RCCL double.  The whole N > 1 flow executes: unique id over torch.distributed,
    rmdf_comm_init, the loopback self-test, the probe frame and the cost-aware deal, rmdf_comm_verify_deal, the verification of the
    exchanged frames (against the digest of the double's own single-launch frame), S frames in flight, the max-over-ranks timing, one
    JSON line from rank 0.  Readiness of the code path the 8-GPU run takes; no number in the line means anything."""
    import json
    import shutil
    import socket
    import subprocess
    import sys
    import rmdf_amd
    from conftest import ROOT
    hdr = str(tmp_path / "probe.hdr")
    shutil.copy(rmdf_amd.DEFAULT_ENV_HDR, hdr)
    fake_rccl = os.path.join(ROOT, "tests", "libfake_rccl.so")
    if not os.path.exists(fake_rccl) or os.path.getmtime(fake_rccl) < os.path.getmtime(os.path.join(ROOT, "tests", "fake_rccl.c")):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "fake_rccl.c"),
                               "-o", fake_rccl, "-L/opt/rocm/lib", "-lamdhip64"])
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), RMDF_ENV_HDR=hdr, RMDF_BENCH_SHARE_GPU="1", RMDF_RCCL_LIB=fake_rccl, FAKE_RCCL_TIMEOUT_S="120",
               RMDF_BENCH_MIN_WARM="0.02")
    for k in ("RMDF_LIB", "RMDF_FLAGS", "RANK", "WORLD_SIZE", "LOCAL_RANK", "RMDF_BENCH_TORCH_GATHER"):
        env.pop(k, None)
    for attempt in range(3):
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(ROOT, "tests", "bench_dry_run.py"), "--gpus", str(nranks), "--steps", "4", "--warmup", "2",
                            "--repeats", "1", "--no-cpu-baseline", "--no-secondary"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == nranks and c["rccl_ranks"] == nranks and d["steps"] == 4 and d["scaling"] == "strong"
    assert c["exchange"].startswith("librmdf_xcheck") and "TEST DOUBLE" in c["exchange"]
    assert c["exchanged_frames_verified"].startswith("%d exchanged frame(s) in flight == committed sha256" % min(nranks, 8)) or "exchanged frame(s) in flight ==" in c["exchanged_frames_verified"]
    assert c["tile_deal"].startswith("cost-aware") and "verified by the library" in c["tile_deal"], c["tile_deal"]
    assert "falling back" not in r.stderr
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("fakerccl_")] or True     # (other tests may be using the double at the same time)


def test_bench_secondary_rooflines_from_the_committed_counters():
    """bench.py's roofline objects of the other three FragmentShader values (round 5; they have not run on hardware): the as-written
    operation formulas applied to the oracle counters committed with the full-size digests, the object's keys, the fraction against
    78.6 T lane-ops/s, the issue fraction against the 2-cycle spec peak -- with a made-up kernel time, no GPU."""
    import json
    import bench
    from conftest import GOLD
    d = json.load(open(os.path.join(GOLD, "full_size_digests.json")))
    for name, scene in (("config2_cornell_1280x720_m128", 0), ("detest_1280x720_t2p5_m128", 1), ("mbgeneral_1280x720_t3p0_m128", 3),
                        ("config3_mandelbulb8_1920x1080_m256", 2)):
        c = d[name]["counters"]
        F = bench.secondary_ops(scene, c)
        if scene == 0:
            assert F == 76 * c["tri_inside"] + 161 * (32 * c["de_evals"] - c["tri_inside"]) + 32 * c["de_evals"] + 9 * c["march_steps"] + 170 * c["hit_pixels"] + 30 * c["pixels"]
            assert 0 < c["tri_inside"] < 32 * c["de_evals"]
        elif scene == 1:
            assert F == 126 * c["de_evals"] + 9 * c["march_steps"] + 150 * c["hit_pixels"] + 30 * c["pixels"]
        elif scene == 3:
            assert F == 37 * c["triplex_iters"] + 11 * c["de_evals"] + 9 * c["march_steps"] + 150 * c["hit_pixels"] + 30 * c["pixels"]
        else:
            assert F == bench.flops_model(c)
        r = bench.secondary_roofline(name, scene, 0.5, 2.0e8, 256)
        assert r["bound"] == "valu" and r["unit"] == "T lane-ops/s" and r["peak"] == bench.VALU_PEAK_TLANEOPS and r["traffic"] is None
        assert r["ops_per_launch"] == F and abs(r["achieved"] - F / 0.5e-3 / 1e12) < 1e-3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
        assert abs(r["issue_g_wave_instr_s_simd"] - 2.0e8 / 1024 / 0.5e-3 / 1e9) < 1e-3
        assert abs(r["issue_frac_of_spec_peak"] - r["issue_g_wave_instr_s_simd"] / bench.VALU_ISSUE_SPEC) < 2e-3
        assert r["formula"] == bench.SECONDARY_FORMULA[scene] and json.dumps(r)
        assert "issue_g_wave_instr_s_simd" not in bench.secondary_roofline(name, scene, 0.5, None, 256)
    assert bench.secondary_roofline("no such digest", 1, 0.5, None, 256)["frac"] is None


def test_bench_self_launcher_without_a_gpu():
    """`python bench.py --gpus N` with no launcher around it becomes the launcher itself (bench.py: self_launch) before anything touches
    the GPU.  On a box that shows fewer than N GPUs (this container shows none) it must say so and exit 2 at once -- no rendezvous, no
    hang, nothing on stdout; with a fake launcher environment (WORLD_SIZE set) it must NOT try to launch again."""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs the launcher would use")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RMDF_BENCH_SELF_LAUNCH", "RMDF_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and not r.stdout.strip(), (r.returncode, r.stdout)
    assert "this node shows" in r.stderr
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                           cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "needs a GPU" in r.stderr and "starting" not in r.stderr


def test_copy_pool_and_deal_fingerprint_symbols(rmdf):
    """the round-4 entry points are exported and refuse null arguments without touching a device"""
    lib = rmdf.load_library()
    assert lib.rmdf_comm_verify_deal(None, None) == -1
    assert lib.rmdf_get_cornell_vertices(None) == -1
    assert lib.rmdf_get_shader_constants(None, None, 0) >= 40


def test_worker_pool_of_the_ctx(tmp_path):
    """csrc/rmdf_host.hpp: WorkPool -- the ctx's host threads (frame copies, staging copies, table builders) -- exercised without a GPU
    (tests/host_pool_test.cpp): every part of every job exactly once for pools of 0 .. 15 workers, begin / finish with work in between,
    prime() / relax() around jobs, the spin hand-over and the condition-variable hand-over, copy() and segments(); plain and under
    ThreadSanitizer."""
    import subprocess
    src = os.path.join(ROOT, "tests", "host_pool_test.cpp")
    for tag, flags in (("plain", ["-O2"]), ("tsan", ["-O1", "-g", "-fsanitize=thread"])):
        exe = str(tmp_path / ("host_pool_test_" + tag))
        subprocess.check_call(["g++", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-DRMDF_HOST_POOL_TEST", "-I/opt/rocm/include"] + flags + [src, "-o", exe])
        for nt in ("0", "1"):                                 # RMDF_COPY_NT=1: copy()'s slices with streaming stores (A/B knob of round 5)
            r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, RMDF_COPY_NT=nt))
            assert r.returncode == 0 and "pool ok" in r.stdout and "ThreadSanitizer" not in r.stderr, (tag, nt, r.stdout[-500:], r.stderr[-2000:])


def _fake_rccl_rank(so, uid_hex, rank, n, q):
    """one rank of test_the_rccl_double_itself (module level: multiprocessing pickles it)"""
    import ctypes as C
    try:
        L = C.CDLL(so)

        class Uid(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid = Uid()
        uid.internal = uid_hex.encode()
        comm = C.c_void_p()
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
        for f in (L.ncclSend, L.ncclRecv):
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclGetErrorString.restype = C.c_char_p
        assert L.ncclCommInitRank(C.byref(comm), n, uid, rank) == 0
        ncclChar = 0
        out = {}
        words = 1000 + 17 * rank
        mine = (np.arange(words, dtype=np.uint32) * 2654435761 + rank).astype(np.uint32)
        if rank == 0:
            # the gather's shape: a grouped fan-in of n - 1 receives of DIFFERENT sizes, twice (sequence numbers per ordered pair)
            for rnd in range(2):
                bufs = [np.zeros(1000 + 17 * r, np.uint32) for r in range(n)]
                assert L.ncclGroupStart() == 0
                for r in range(1, n):
                    assert L.ncclRecv(bufs[r].ctypes.data, bufs[r].nbytes, ncclChar, r, comm, None) == 0
                assert L.ncclGroupEnd() == 0
                for r in range(1, n):
                    assert np.array_equal(bufs[r], (np.arange(1000 + 17 * r, dtype=np.uint32) * 2654435761 + r).astype(np.uint32) + rnd), (rnd, r)
            # a receive whose size differs from the send fails instead of hanging; one nobody answers times out
            small = np.zeros(5, np.uint32)
            rc = L.ncclRecv(small.ctypes.data, small.nbytes, ncclChar, 1, comm, None)
            out["mismatch"] = (rc, L.ncclGetErrorString(rc).decode())
            rc = L.ncclRecv(small.ctypes.data, small.nbytes, ncclChar, n - 1, comm, None)
            out["timeout"] = (rc, L.ncclGetErrorString(rc).decode())
            # to self, grouped: the loopback self-test's shape
            back = np.zeros(words, np.uint32)
            assert L.ncclGroupStart() == 0
            assert L.ncclRecv(back.ctypes.data, back.nbytes, ncclChar, 0, comm, None) == 0
            assert L.ncclSend(mine.ctypes.data, mine.nbytes, ncclChar, 0, comm, None) == 0
            assert L.ncclGroupEnd() == 0
            assert np.array_equal(back, mine)
        else:
            for rnd in range(2):
                data = (mine + rnd).astype(np.uint32)
                assert L.ncclSend(data.ctypes.data, data.nbytes, ncclChar, 0, comm, None) == 0
            if rank == 1:
                assert L.ncclSend(mine.ctypes.data, 64, ncclChar, 0, comm, None) == 0          # 64 bytes where rank 0 expects 20
            if rank == 2:
                assert L.ncclSend(mine.ctypes.data, 8, ncclChar, 0, comm, None) == 0           # never received: CommDestroy of rank 0 removes it
        if rank == 0:
            import time
            time.sleep(0.3)
        assert L.ncclCommDestroy(comm) == 0
        q.put((rank, "ok", out))
    except Exception as e:                                      # noqa: BLE001
        import traceback
        q.put((rank, "failed: %s\n%s" % (e, traceback.format_exc()), {}))


def test_the_rccl_double_itself(tmp_path):
    """tests/fake_rccl.c is what the N > 1 exchange tests of the GPU tier stand on, so it is tested itself -- on the CPU tier, built with
    FAKE_RCCL_NO_GPU ("device" buffers are host memory): four processes; a grouped fan-in of receives of different sizes, twice (ordering per
    pair); a receive whose size differs from its send fails with ncclInvalidArgument instead of hanging; a receive nobody answers fails with
    ncclSystemError after FAKE_RCCL_TIMEOUT_S; grouped send + receive to self; nothing is left in /dev/shm."""
    import multiprocessing as mp
    import subprocess
    so = str(tmp_path / "libfake_rccl_cpu.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-DFAKE_RCCL_NO_GPU", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "fake_rccl.c"), "-o", so])
    uid = "%032x" % int.from_bytes(os.urandom(16), "big")
    os.environ["FAKE_RCCL_TIMEOUT_S"] = "1"
    try:
        ctx = mp.get_context("fork")
        q = ctx.Queue()
        n = 4
        ps = [ctx.Process(target=_fake_rccl_rank, args=(so, uid, r, n, q)) for r in range(n)]
        for p in ps:
            p.start()
        res = {}
        for _ in range(n):
            rank, status, out = q.get(timeout=120)
            res[rank] = (status, out)
        for p in ps:
            p.join(30)
        assert all(res[r][0] == "ok" for r in range(n)), res
        out = res[0][1]
        assert out["mismatch"][0] != 0 and "expects 20 bytes" in out["mismatch"][1] and "sent 64" in out["mismatch"][1], out
        assert out["timeout"][0] != 0 and "waited" in out["timeout"][1], out
        assert not [f for f in os.listdir("/dev/shm") if f.startswith("fakerccl_" + uid)]
    finally:
        os.environ.pop("FAKE_RCCL_TIMEOUT_S", None)


def test_march_loop_classes_of_the_shipped_sources(tmp_path):
    """tools/isa/march_loop_classes.py (round 6): the headline kernel compiled to assembly with the product's flags, the
    march's iteration pass found in the code generator's loop annotations, its instructions by class.  Pins what DESIGN.md section 6 says about
    the pass -- 87 vector instructions for 79 as-written operations, two of them transcendental, no v_cndmask, no scratch -- and that the committed
    profiles/r06_march_loop_classes.json is what the tool derives from the sources as they are."""
    import json
    import shutil
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(ROOT, "tools", "isa"))
    import march_loop_classes as m
    text, d = m.report()
    assert d["pass_instr"] == 87 and abs(d["vslots"] - 98.2) < 1e-9
    assert "v_cndmask (selects)" not in text.split("## the step loop")[0]          # (the class's row name: the pass has none)
    committed = json.load(open(os.path.join(ROOT, "profiles", "r06_march_loop_classes.json")))
    for k in ("pass_instr", "vslots", "full_slots", "t_spec_ms", "t_sust_ms", "ops"):
        assert abs(committed[k] - d[k]) <= 1e-9 * abs(d[k]), k
    assert 0.55 < d["t_as_written_ms"] / d["t_spec_ms"] < 0.65          # (B) = 0.59 of (A)


def test_bench_collects_its_own_counter_passes_when_profiles_has_none_of_this_build(tmp_path):
    """Round 6: profiles/pmc_traffic.json names the library it was measured with; for any other build bench.py used to print `traffic: null`.
    Now (--pmc auto, N = 1, headline workload) it runs the counter passes itself after the timed region: children under `rocprofv3 --pmc <one
    group>`, condensed by tools/pmc_summary.py.  Here the HIP double stands in for the GPU and a script for rocprofv3 (it runs the child
    and writes a counter file in rocprofv3's CSV layout): the plumbing end to end -- traffic, issue block and the secondary scenes' instruction
    counts arrive in the line, labelled as collected live."""
    import json
    import shutil
    import stat
    import subprocess
    import sys
    import rmdf_amd
    from conftest import ROOT
    fake = tmp_path / "bin"
    fake.mkdir()
    script = fake / "rocprofv3"
    script.write_text('''#!%s
import os, subprocess, sys
a = sys.argv[1:]
cmd = a[a.index("--") + 1:]
d = a[a.index("-d") + 1]
ctrs = []
i = a.index("--pmc") + 1
while not a[i].startswith("-"):
    ctrs.append(a[i]); i += 1
assert "--kernel-trace" not in a and "--sys-trace" not in a          # counters are never combined with tracing
rc = subprocess.call(cmd)
sc = cmd[cmd.index("--scene") + 1] if "--scene" in cmd else "2"
os.makedirs(os.path.join(d, "host"), exist_ok=True)
val = {"FETCH_SIZE": 9000.0, "WRITE_SIZE": 8200.0, "SQ_INSTS_VALU": 3.0e8, "SQ_ACTIVE_INST_VALU": 3.1e8, "SQ_THREAD_CYCLES_VALU": 1.5e10}
with open(os.path.join(d, "host", "123_counter_collection.csv"), "w") as f:
    f.write("Dispatch_Id,Kernel_Name,Grid_Size,Workgroup_Size,LDS_Block_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value\\n")
    for disp in range(1, 6):
        for c in ctrs:
            f.write('%%d,"void rmdf::k_render<%%s, true, 0>(rmdf::FrameParams)",2073600,256,17920,56,0,80,%%s,%%f\\n' %% (disp, sc, c, val.get(c, 1000.0)))
sys.exit(rc)
''' % sys.executable)
    script.chmod(script.stat().st_mode | stat.S_IXUSR)
    hdr = str(tmp_path / "probe.hdr")
    shutil.copy(rmdf_amd.DEFAULT_ENV_HDR, hdr)
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), RMDF_ENV_HDR=hdr, PATH=str(fake) + os.pathsep + os.environ["PATH"],
               RMDF_BENCH_SELF=os.path.join(ROOT, "tests", "bench_dry_run.py"))
    for k in ("RMDF_LIB", "RMDF_FLAGS", "RANK", "WORLD_SIZE", "LOCAL_RANK", "RMDF_BENCH_PMC", "RMDF_BENCH_PMC_CHILD"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_dry_run.py"), "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    rl = d["roofline"]
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    import hashlib
    if committed.get("lib_sha256") == hashlib.sha256(open(rmdf_amd.LIB_PATH, "rb").read()).hexdigest():
        pytest.skip("profiles/pmc_traffic.json is of this build: nothing to collect")
    assert rl["traffic"] == (9000.0 + 8200.0) * 1024.0 and "LIVE" in rl["traffic_source"], rl.get("traffic_source")
    assert rl["issue"]["valu_instructions_per_launch"] == 3.0e8
    for name in ("config2_cornell_1280x720_m128", "scene1_detest_1280x720_m128", "scene3_mbgeneral_1280x720_m128"):
        assert d["secondary"][name]["roofline"].get("issue_g_wave_instr_s_simd") is not None, name
    shutil.rmtree(os.path.join(ROOT, "gpurun_out", "prof_live"), ignore_errors=True)


def test_the_tools_gpu_measure_runs_do_run(tmp_path):
    """tools/gpu_measure.sh is what the first GPU call of a round executes, and GPU minutes are scarce: every Python tool it calls is run here
    to completion against the doubles (tests/bench_dry_run.py with DRY_RUN_SCRIPT: the same torch stand-ins bench.py's dry run uses), the
    shell scripts are syntax-checked, and the steps the script names exist.  The numbers mean nothing; a typo, a renamed keyword argument
    or a hand-over mode the product library no longer has would show here instead of on the GPU box."""
    import shutil
    import subprocess
    import sys
    import rmdf_amd
    from conftest import ROOT
    for sh in ("tools/gpu_measure.sh", "tools/profile.sh", "tools/prof_scene.sh", "tools/pmc_prefilter.sh", "tools/abtest/rebuild_all.sh", "tools/abtest/build_variant.sh"):
        r = subprocess.run(["bash", "-n", os.path.join(ROOT, sh)], capture_output=True, text=True)
        assert r.returncode == 0, (sh, r.stderr)
    script = open(os.path.join(ROOT, "tools", "gpu_measure.sh")).read()
    for step in ("tier", "bench", "prof", "scenes", "sweep", "mirror16", "copynt", "prefilter", "unverified"):
        assert ("\n%s)" % step) in script, step
    assert "--faults" in script and script.index("faults = 1") > script.index("unverified)") if "faults = 1" in script else True
    hdr = str(tmp_path / "probe.hdr")
    shutil.copy(rmdf_amd.DEFAULT_ENV_HDR, hdr)
    env = dict(os.environ, LD_PRELOAD=_fake_hip_lib(), RMDF_ENV_HDR=hdr)
    for k in ("RMDF_LIB", "RMDF_FLAGS", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    for tool, args, expect in (("tools/scene_times.py", ["2"], "headline mb8"), ("tools/prefilter_time.py", [], "four powers on four streams"),
                               ("tools/whole_frame_sweep.py", ["1"], "bands mirror threads"), ("tools/tile_mode_time.py", [], "")):
        if not os.path.exists(os.path.join(ROOT, tool)):
            continue
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_dry_run.py")] + args, cwd=ROOT, capture_output=True, text=True, timeout=900,
                           env=dict(env, DRY_RUN_SCRIPT=os.path.join(ROOT, tool)))
        assert r.returncode == 0 and "Traceback" not in r.stderr and expect in r.stdout, (tool, r.stdout[-800:], r.stderr[-2000:])
        if tool.endswith("whole_frame_sweep.py"):
            rows = [l.split() for l in r.stdout.splitlines() if l[:6].strip().isdigit()]
            assert {int(x[1]) for x in rows} == {0, 1, 2, 3}, "the sweep lost a hand-over mode"        # 2 and 3 run on librmdf_xcheck.so
