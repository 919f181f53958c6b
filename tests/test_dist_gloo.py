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
