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


def test_batch_of_frames_in_one_grid(orc):
    """VERDICT r4 item 4 (BASELINE config 3): pysp_pipeline_batch_dev runs the Draft / EAG frames of a batch as ONE grid per 16 frames (blockIdx.z = frame,
    XCD-aware tile order per frame from the workgroup's number in the whole grid).  Ragged batches -- 1, 2, 3, 5, 17 and 33 frames, frame sizes whose tile
    count is and is not a multiple of 8, partial tiles -- every frame bit-identical to its own single-frame call and to the oracle
    (debayer/edge_assisted_gaussian.py:188-201, debayer/fast_resize.py:7-44), colour tails 0, 1, 2."""
    import torch
    from pysp_amd import _lib
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import rggb_frame
    wb, M = _wbM(orc)
    pipe = DevicePipeline(0)
    for (H, W) in ((64, 128), (70, 130), (200, 330), (36, 1000)):
        for n in (1, 2, 3, 5, 17, 33):
            if n > 5 and H * W > 70 * 130:
                continue
            frames = [rggb_frame(H, W, 300 + 7 * n + i) for i in range(n)]
            dev = [torch.from_numpy(f).cuda() for f in frames]
            for quality in (_lib.QUALITY_DRAFT, _lib.QUALITY_FAST):
                for tail in (0, 1, 2):
                    outs = pipe.batch(dev, wb, M, quality, False, 0, tail)
                    pipe.sync()
                    for i, (d, o) in enumerate(zip(dev, outs)):
                        single = pipe.batch([d], wb, M, quality, False, 0, tail)[0]
                        assert torch.equal(o, single), (H, W, n, quality, tail, i)
                    if tail == 0 and n in (3, 17):
                        ref = (orc.demosaic_draft if quality == _lib.QUALITY_DRAFT else orc.demosaic_eag)(frames[n - 1], wb)
                        assert np.array_equal(outs[n - 1].cpu().numpy(), ref), (H, W, n, quality)


def test_automatic_lab_layout_on_alternating_content(orc):
    """one context, the automatic Lab layout, streams that alternate scene and noise frames with periods 1, 4, 16, 64 and 300.
    (a) every frame is bit-identical to the fixed-layout result (a small frame, checked frame by frame: whatever the policy launches, the bits are the same);
    (b) at 24 MP the automatic policy costs at most 5 % more per frame than the better FIXED layout of that stream, for every period
    (tools/lab_layout_alternation.py prints the table that DESIGN.md section 7 quotes)."""
    import ctypes
    import sys, os
    import torch
    from pysp_amd import _lib
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import random_frame, rggb_frame
    wb, M = _wbM(orc)
    H, W = 600, 900
    pipe = DevicePipeline(0)
    pipe.ctx.set_lab_layout("planes")
    srcs = {"scene": torch.from_numpy(rggb_frame(H, W, 1000)).cuda(), "noise": torch.from_numpy(random_frame(H, W, 3)).cuda()}
    want = {k: pipe.demosaic_to_srgb(v, wb, M, _lib.QUALITY_BEST, False, 1).clone() for k, v in srcs.items()}
    assert np.array_equal(want["scene"].cpu().numpy(), orc.pipeline_srgb(rggb_frame(H, W, 1000), wb, M, 2, False, 1, False))
    pipe.ctx.set_lab_layout("auto")
    seen = set()
    for period in (1, 4, 16, 64, 300):
        for i in range(2 * period + 40 if period > 16 else 200):
            kind = "scene" if (i // period) % 2 == 0 else "noise"
            got = pipe.demosaic_to_srgb(srcs[kind], wb, M, _lib.QUALITY_BEST, False, 1)
            assert torch.equal(got, want[kind]), (period, i, kind)
            seen.add(pipe.ctx.lab_layout_in_use())
    assert seen == {0, 1}                               # the policy really moved between the two kernels while the bits stayed put
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from lab_layout_alternation import measure
    rows = measure((4000, 6000), 900)
    for r in rows:
        assert r["auto_over_best_fixed"] <= 1.05, rows
