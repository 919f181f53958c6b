"""GPU (-m gpu): checks added in round 5.  Same bars as tests/test_gpu_parity.py (bit-exact vs the oracle unless a tolerance is stated)."""
import numpy as np
import pytest

from conftest import D65_XY, MULT, XYZ2CAM

pytestmark = pytest.mark.gpu


def _wbM(orc):
    return (1.0 / MULT).astype(np.float32), orc.final_matrix(XYZ2CAM, orc.xy_to_XYZ(D65_XY))


# frame sizes around every boundary of the streaming select kernel: one head pass (<= 14 quad rows), head + chained passes (14 + 16 m rows and one more / fewer),
# the flush pass (a column whose last row is the stashed one: 14 + 16 m + 16 rows), several column tiles, partial tiles right and below, many chunks per column
STREAM_SIZES = [(8, 8), (10, 30), (28, 28), (30, 30), (32, 60), (58, 34), (60, 28), (62, 90), (64, 56), (92, 40), (94, 118), (120, 176), (124, 30),
                (126, 62), (188, 64), (250, 318), (316, 58), (638, 126), (1260, 92)]


def test_streaming_select_equals_tile_select_and_oracle(orc):
    """the persistent, column-streaming form of the AHD select kernel (k_ahd_select_stream: carried Lab rows and votes, the last quad row of a
    pass selected one pass later from an LDS stash, chunk queues per XCD) returns the same bits as the tile form on every size class, for scene and noise
    content (noise sends waves through the float form of the vote), with 0 and 1 median stages, with and without the colour tail, float32 and uint16 input --
    and both equal the oracle (debayer/ahd.py:14-170, debayer/ahd_homogeneity_cython.pyx:22-58)."""
    import torch
    from pysp_amd import _lib
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import random_frame, rggb_frame
    wb, M = _wbM(orc)
    pipe = DevicePipeline(0)
    ctx = pipe.ctx
    ctx.set_lab_layout("packed")
    try:
        for n, (H, W) in enumerate(STREAM_SIZES):
            for kind in ("scene", "noise"):
                f = rggb_frame(H, W, 1000 + n) if kind == "scene" else random_frame(H, W, 7 + n)
                d = torch.from_numpy(f).cuda()
                for stages, tail in ((0, 0), (1, 2), (0, 2)):
                    ctx.set_select_form("tile")
                    a = pipe.batch([d], wb, M, _lib.QUALITY_BEST, False, stages, tail)[0].cpu().numpy()
                    ctx.set_select_form("stream")
                    b = pipe.batch([d], wb, M, _lib.QUALITY_BEST, False, stages, tail)[0].cpu().numpy()
                    assert np.array_equal(a, b, equal_nan=True), (H, W, kind, stages, tail, int((a != b).sum()), np.argwhere((a != b).any(axis=2))[:5].tolist())
                if kind == "scene" and n % 3 == 0:               # uint16 mosaic, bayer_normalize fused into the loader
                    u = torch.from_numpy(np.clip(np.round(f * 15359.0 + 512.0), 0, 16383).astype(np.uint16).view(np.int16)).cuda()
                    ctx.set_select_form("tile")
                    a = pipe.raw_u16_to_rgb(u, [512.0] * 4, [15871.0] * 4, wb, M, _lib.QUALITY_BEST, 1, 2).cpu().numpy()
                    ctx.set_select_form("stream")
                    b = pipe.raw_u16_to_rgb(u, [512.0] * 4, [15871.0] * 4, wb, M, _lib.QUALITY_BEST, 1, 2).cpu().numpy()
                    assert np.array_equal(a, b, equal_nan=True), (H, W, "uint16")
                if H * W <= 120 * 176:
                    want = orc.demosaic_ahd(f, wb, M, False, 0)
                    ctx.set_select_form("stream")
                    got = pipe.demosaic(d, wb, M, _lib.QUALITY_BEST, False, 0).cpu().numpy()
                    assert np.array_equal(got, want, equal_nan=True), (H, W, kind)
        # the same call again and again on one context: the queues reset themselves at the end of every launch
        f = rggb_frame(250, 318, 5)
        d = torch.from_numpy(f).cuda()
        ctx.set_select_form("tile")
        a = pipe.demosaic(d, wb, M, _lib.QUALITY_BEST, False, 1).cpu().numpy()
        ctx.set_select_form("stream")
        for _ in range(5):
            assert np.array_equal(pipe.demosaic(d, wb, M, _lib.QUALITY_BEST, False, 1).cpu().numpy(), a)
        # the HDR metric and the float-plane layout have no streaming instance: the call falls back to the tile kernel, same results as ever
        want = orc.demosaic_ahd(f * np.float32(3.0), wb, M, True, 0)
        assert np.array_equal(pipe.demosaic(d * 3.0, wb, M, _lib.QUALITY_BEST, True, 0).cpu().numpy(), want, equal_nan=True)
    finally:
        ctx.set_select_form("tile")
        ctx.set_lab_layout("auto")
