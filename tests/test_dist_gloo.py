"""CPU, world_size 2, gloo: the N>1 host path -- parameter broadcast and frame sharding."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from pysp_amd.multi_gpu import broadcast_params, frames_for_rank
        if rank == 0:
            from pysp_amd.colorize.transform import final_matrix
            from pysp_amd.synth import default_wb
            wbobj = default_wb()
            wb, M = broadcast_params(wbobj.get_reciprocal_multipliers(), final_matrix(wbobj.get_matrix()))
        else:
            wb, M = broadcast_params(np.zeros(3, np.float32), np.zeros((3, 3)))
        q.put((rank, wb.tobytes(), M.tobytes(), frames_for_rank(64, rank, world)))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, wb0, m0, f0), (_, wb1, m1, f1) = res
    assert wb0 == wb1 and m0 == m1                         # every rank holds rank 0's block, bit for bit
    wb = np.frombuffer(wb0, np.float32)
    assert np.array_equal(wb, (1.0 / np.array([0.5, 1.0, 0.7], np.float32)))
    assert sorted(f0 + f1) == list(range(64)) and not set(f0) & set(f1) and len(f0) == len(f1) == 32


def test_sharding_properties():
    from pysp_amd.multi_gpu import frames_for_rank, pack_params, unpack_params
    for n in (0, 1, 7, 64):
        for world in (1, 2, 8):
            parts = [frames_for_rank(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        frames_for_rank(4, 2, 2)
    wb = np.array([2.0, 1.0, 1.4285715], np.float32); M = np.random.default_rng(0).standard_normal((3, 3))
    w2, m2 = unpack_params(pack_params(wb, M))
    assert np.array_equal(w2, wb) and np.array_equal(m2, M)


def test_band_ranges():
    from pysp_amd.multi_gpu import band_ranges
    for H, n, halo in ((8736, 8, 20), (600, 8, 20), (10, 8, 4), (2, 3, 0)):
        bands = band_ranges(H, n, halo)
        assert bands[0][0] == 0 and bands[-1][1] == H
        for (y0, y1, r0, r1), nxt in zip(bands, bands[1:] + [None]):
            assert y0 % 2 == 0 and y1 % 2 == 0 and y0 < y1 and r0 % 2 == 0 and r1 % 2 == 0
            assert r0 == max(0, y0 - halo) and r1 == min(H, y1 + halo)
            if nxt:
                assert nxt[0] == y1
    with pytest.raises(ValueError):
        band_ranges(7, 2, 2)


# ---- band-parallel (BASELINE config 5): row exchange between demosaic and warp -----------------------

def test_plan_row_exchange():
    from pysp_amd.multi_gpu import plan_row_exchange
    bands = [(0, 10), (10, 20), (20, 30)]
    needs = [(0, 13), (8, 22), (19, 30)]
    plan = plan_row_exchange(bands, needs)
    assert plan == [(1, 0, 10, 13), (0, 1, 8, 10), (2, 1, 20, 22), (1, 2, 19, 20)]
    assert plan_row_exchange(bands, bands) == []                       # identity warp: nothing moves
    assert plan_row_exchange(bands, [(0, 30)] * 3) == [(1, 0, 10, 20), (2, 0, 20, 30), (0, 1, 0, 10), (2, 1, 20, 30), (0, 2, 0, 10), (1, 2, 10, 20)]
    assert plan_row_exchange(bands, [(0, 0), (0, 0), (25, 26)]) == []  # empty needs, needs inside the own band


def _exchange_worker(rank, world, port, q, mode):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from pysp_amd.multi_gpu import allgather_bands, exchange_rows, plan_row_exchange
        H, W = 24, 5
        truth = torch.arange(H * W * 3, dtype=torch.float32).reshape(H, W, 3)
        bands = [(0, 8), (8, 16), (16, 24)] if mode != "ragged" else [(0, 6), (6, 18), (18, 24)]
        y0, y1 = bands[rank]
        full = torch.full((H, W, 3), float("nan"))
        full[y0:y1] = truth[y0:y1]
        if mode == "needed":
            needs = [(0, 11), (5, 17), (13, 24)]
            exchange_rows(full, plan_row_exchange(bands, needs), rank, via_host=(rank % 2 == 0))   # both landing paths
            s0, s1 = needs[rank]
            lo, hi = min(s0, y0), max(s1, y1)
            ok = bool(torch.equal(full[lo:hi], truth[lo:hi])) and bool(torch.isnan(full[:lo]).all()) and bool(torch.isnan(full[hi:]).all())
        else:
            allgather_bands(full, bands, rank, via_host=(mode == "ragged" and rank == 1))
            ok = bool(torch.equal(full, truth))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["needed", "allgather", "ragged"])
def test_row_exchange_world3(mode):
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True), (2, True)]


class _OraclePipe:
    """Stand-in for DevicePipeline with the CPU oracle as the compute: rehearses the orchestration of
    demosaic_warp_banded (band cut, halo, exchange plan, row-limited warp) without a GPU."""
    def __init__(self, orc):
        import torch
        self.torch, self.orc, self.device = torch, orc, torch.device("cpu")

    def sync(self):
        pass

    def demosaic(self, bayer, wb, M, quality, hdr, stages, out=None):
        res = self.torch.from_numpy(self.orc.demosaic_ahd(bayer.numpy(), wb, M, hdr, stages))
        if out is None:
            return res
        out.copy_(res)            # halo rows included, as the device kernels write them (multi_gpu: the exchange overwrites the ones the warp reads)
        return out

    def _cells(self, H, W, coeffs, centre, scale):
        rows = []
        for c in range(3):
            k = [float(v) for v in np.asarray(coeffs, np.float64).reshape(3, 6)[c]]
            tab = self.orc.warp_table(*k, W, H, float(centre[0]), float(centre[1]), float(scale))
            my = np.clip(tab[..., 1], 0, H - 1)
            rows.append((np.rint(my * np.float32(32)).astype(np.int64) >> 5) - 3)
        return np.stack(rows)                                            # first tap row per channel and pixel

    def warp_source_rows(self, H, W, coeffs, centre, row0, row1, scale=1.0):
        iy = self._cells(H, W, coeffs, centre, scale)[:, row0:row1]
        return max(0, int(iy.min())), min(H - 1, int(iy.max()) + 7) + 1

    def warp_rows(self, rgb, coeffs, centre, row0, row1, out, scale=1.0):
        res = self.orc.warp_rectilinear(rgb.numpy().copy(), coeffs, centre, scale)
        out[row0:row1] = self.torch.from_numpy(res[row0:row1])
        return out


def _banded_worker(rank, world, port, q, exchange):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import oracle as orc
        from pysp_amd import multi_gpu
        from pysp_amd.colorize.transform import final_matrix
        from pysp_amd.synth import default_wb, rggb_frame
        import torch
        H, W, stages = 96, 64, 1
        bayer = rggb_frame(H, W, 1234)
        wbobj = default_wb()
        wb, M = wbobj.get_reciprocal_multipliers(), final_matrix(wbobj.get_matrix())
        coeffs = np.array([[1.0, 0.05, 0.01, 0, 0, 0], [1.0, 0, 0, 0, 0, 0], [1.0, -0.05, 0.01, 0, 0.002, 0]])
        real_empty = torch.empty
        torch.empty = lambda *a, **k: real_empty(*a, **k).fill_(float("nan")) if k.get("dtype", None) == torch.float32 else real_empty(*a, **k)
        try:
            y0, y1, band = multi_gpu.demosaic_warp_banded(_OraclePipe(orc), bayer, wb, M, coeffs, (0.5, 0.5), stages=stages, rank=rank, world=world,
                                                          exchange=exchange)
        finally:
            torch.empty = real_empty
        ref = orc.warp_rectilinear(orc.demosaic_ahd(bayer, wb, M, False, stages), coeffs, (0.5, 0.5), 1.0)
        q.put((rank, y0, y1, bool(np.array_equal(band.numpy(), ref[y0:y1]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["needed", "allgather"])
def test_banded_demosaic_warp_world2_oracle_compute(exchange):
    # rows a band does not need stay NaN in its frame buffer: the output must still equal the whole-frame result
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_banded_worker, args=(r, world, port, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, 0, 48, True), (1, 48, 96, True)]


# ---- BASELINE config 5 at its real geometry, 8 ranks: the band plan and the row-exchange plan ---------------
def _warp_rows_needed(H, W, coeffs, centre, y0, y1, scale=1.0):
    """Source rows [s0, s1) that output rows [y0, y1) of the per-channel WarpRectilinear read: dng_warp_rectilinear_coords.pyx:18-40 restated
    for the rows of one band (float64; +-1 row of margin covers the float32 table's rounding), clipped like chan_distortion_corr.py:95-96,
    Lanczos-4 support of cv2.remap (first tap row = cell - 3, eight rows)."""
    xs = np.arange(W, dtype=np.float64)[None, :]
    ys = np.arange(y0, y1, dtype=np.float64)[:, None]
    cx, cy = (W - 1) * centre[0], (H - 1) * centre[1]
    m = np.sqrt(max(abs(-cx), abs(W - 1 - cx)) ** 2 + max(abs(-cy), abs(H - 1 - cy)) ** 2)
    dx, dy = (xs - cx) / m, (ys - cy) / m
    r2 = dx * dx + dy * dy
    lo, hi = H, 0
    for kr0, kr1, kr2, kr3, kt0, kt1 in np.asarray(coeffs, np.float64).reshape(3, 6):
        f = kr0 + kr1 * r2 + kr2 * r2 * r2 + kr3 * r2 * r2 * r2
        dyt = 2.0 * kt1 * dx * dy + kt0 * (r2 + 2.0 * dy * dy)        # tangential part of the y coordinate (DNG 1.4 WarpRectilinear)
        yp = cy + m * (dy * f + dyt)
        my = np.clip(ys + (yp - ys) * scale, 0, H - 1)
        cell = np.floor(my).astype(np.int64)
        lo, hi = min(lo, int(cell.min()) - 3 - 1), max(hi, int(cell.max()) + 4 + 1 + 1)
    return max(0, lo), min(H, hi)


def _plan8_worker(rank, world, port, q):
    import hashlib
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from pysp_amd.multi_gpu import BandPlan, plan_row_exchange
        H, W, stages = 8736, 11648, 3                                   # BASELINE config 5
        coeffs = [[1.0, 0.01, 0.002, 0.0, 0.0, 0.0], [1.0, 0.0, 0.002, 0.0, 0.0, 0.0], [1.0, -0.01, 0.002, 0.0, 0.0, 0.0]]
        plan = BandPlan(H, W, world, rank, stages)
        s0, s1 = _warp_rows_needed(H, W, coeffs, (0.5, 0.5), plan.y0, plan.y1)
        mine = torch.tensor([s0, s1], dtype=torch.int64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                                    # what demosaic_warp_banded_dev does with the device's bounds
        needs = [(int(t[0]), int(t[1])) for t in every]
        xfers = plan_row_exchange(plan.bands, needs)
        digest = int(hashlib.sha256(repr(xfers).encode()).hexdigest()[:12], 16)
        all_d = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(all_d, torch.tensor([digest], dtype=torch.int64))
        same_plan = len({int(d[0]) for d in all_d}) == 1
        recv = sum(r1 - r0 for src, dst, r0, r1 in xfers if dst == rank)
        sent = sum(r1 - r0 for src, dst, r0, r1 in xfers if src == rank)
        q.put((rank, plan.y0, plan.y1, plan.r0, plan.r1, needs[rank], recv, sent, same_plan,
               max((r1 - r0 for _, _, r0, r1 in xfers), default=0), max((abs(s - d) for s, d, _, _ in xfers), default=0)))
    finally:
        dist.destroy_process_group()


def test_config5_band_and_exchange_plan_world8_real_geometry():
    """8 gloo ranks, the 8736 x 11648 frame of BASELINE config 5 with its warp coefficients: bands tile the frame on even rows with the
    AHD(3) halo of 20 rows; every rank derives the same exchange plan; rows move between neighbouring bands only, at most 60 per
    transfer, and each rank receives a small fraction of what the all-gather would deliver."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_plan8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    H = 8736
    assert res[0][1] == 0 and res[-1][2] == H
    for i, (rank, y0, y1, r0, r1, need, recv, sent, same_plan, biggest, reach) in enumerate(res):
        assert rank == i and y0 % 2 == 0 and y1 % 2 == 0 and (i == 0 or y0 == res[i - 1][2])
        assert r0 == max(0, y0 - 20) and r1 == min(H, y1 + 20)                          # ahd_halo_rows(3) = 20
        assert same_plan
        assert 0 <= need[0] < need[1] <= H and need[0] < y1 and need[1] > y0              # what a band's warp reads overlaps the band itself
        assert biggest <= 60 and reach <= 1                                             # neighbours only, <= 60 rows per transfer
        assert recv <= 120 and recv < (H - (y1 - y0)) // 50                             # all-gather: H - band rows received per rank
    assert sum(r[6] for r in res) == sum(r[7] for r in res) > 0


def test_rccl_rendezvous_without_torch_or_gpu():
    """pysp_amd._rccl.exchange_unique_id: rank 0 serves the 128-byte ncclUniqueId over TCP (MASTER_ADDR / a port derived from MASTER_PORT), every rank
    returns the same bytes -- late joiners retry until the server is up.  Plain sockets: this part of the torch-free RCCL path runs anywhere."""
    import socket
    import threading
    import time
    from pysp_amd._rccl import NCCL_UNIQUE_ID_BYTES, exchange_unique_id
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    payload = bytes(range(128))
    assert len(payload) == NCCL_UNIQUE_ID_BYTES
    got = {}

    def run(rank, delay):
        time.sleep(delay)
        got[rank] = exchange_unique_id(rank, 4, payload if rank == 0 else None, "127.0.0.1", port, timeout=30.0)
    ts = [threading.Thread(target=run, args=(r, d)) for r, d in ((1, 0.0), (2, 0.0), (0, 0.3), (3, 0.5))]      # two ranks knock before rank 0 listens
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert got == {r: payload for r in range(4)}
    assert exchange_unique_id(0, 1, payload) == payload
