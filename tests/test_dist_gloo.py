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

    def demosaic(self, bayer, wb, M, quality, hdr, stages):
        return self.torch.from_numpy(self.orc.demosaic_ahd(bayer.numpy(), wb, M, hdr, stages))

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
