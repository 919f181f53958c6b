"""GPU (-m gpu): checks added in round 2 -- stream ordering of the device-resident API, the drop-in behaviours the
reference has on views / odd plane counts, non-finite inputs, CFA patterns for every quality, the device-resident
drop-in objects.  Same bars as tests/test_gpu_parity.py (bit-exact vs the oracle unless a tolerance is stated)."""
import struct

import numpy as np
import pytest

from conftest import D65_XY, MULT, XYZ2CAM, load_golden, ulp_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wbobj():
    from pysp_amd.synth import default_wb
    return default_wb()


def _wbM(orc):
    return (1.0 / MULT).astype(np.float32), orc.final_matrix(XYZ2CAM, orc.xy_to_XYZ(D65_XY))


def _opcode_blob(coeffs, centre=(0.5, 0.5)):
    n = len(coeffs)
    payload = struct.pack(">I", n) + b"".join(struct.pack(">6d", *c) for c in coeffs) + struct.pack(">2d", *centre)
    return struct.pack(">I", 1) + struct.pack(">IIII", 1, 1, 0, len(payload)) + payload


# ---- library kernels are ordered with torch work (one stream), also on a non-default stream ------------------
def test_pipeline_is_ordered_on_torchs_current_stream(orc, wbobj):
    """DevicePipeline enqueues on torch's current stream: a torch op that consumes the result right after the call, and a
    block freed and reallocated right after it, see finished data without any host synchronisation.  The frame is large
    enough (AHD with 3 median stages, ~10 ms of kernels) that a missing dependency shows as garbage."""
    import torch
    from pysp_amd import _lib
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import rggb_frame
    wb, M = _wbM(orc)
    H, W = 1536, 2048
    bay = rggb_frame(H, W, 99)
    ref = orc.demosaic_ahd(bay, wb, M, False, 3)
    pipe = DevicePipeline(0)
    for stream in (None, torch.cuda.Stream()):
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream()):
            d = torch.from_numpy(bay).cuda(non_blocking=True)
            rgb = pipe.demosaic(d, wb, M, _lib.QUALITY_BEST, False, 3)
            doubled = rgb * 2.0                                   # torch op on the same stream, no sync in between
            del rgb, d
            junk = torch.full((H, W, 3), -7.0, device="cuda")     # may reuse the freed blocks: must be ordered after the kernels
            cur = torch.cuda.current_stream()
            assert pipe.ctx.get_stream() == int(cur.cuda_stream)
            got = doubled.cpu().numpy()
        assert np.array_equal(got, ref * np.float32(2.0))
        del junk


def test_entry_points_keep_the_callers_device(wbobj):
    import torch
    from pysp_amd.bayer_chan_mixer import bayer_to_rgbg
    if torch.cuda.device_count() < 2:
        # one GPU on the box: the guard's no-op path (same device) is what runs; the device must still read 0 afterwards
        bayer_to_rgbg(np.zeros((4, 4), np.float32))
        assert torch.cuda.current_device() == 0
        return
    torch.cuda.set_device(1)
    bayer_to_rgbg(np.zeros((4, 4), np.float32))                  # default context lives on the device current at first use
    torch.zeros(1, device="cuda")
    assert torch.cuda.current_device() == 1


# ---- apply_opcode_3_warp works in place on views and on any plane count, like the reference -------------------
def test_warp_in_place_on_views_and_plane_counts():
    from pysp_amd.dng_warp_corr import apply_opcode_3_warp
    rng = np.random.default_rng(12)
    coeffs = [(1.0, 0.02, 0.002, 0.0, 0.0, 0.0), (1.0, -0.01, 0.0, 0.0, 0.001, 0.0), (0.99, 0.0, 0.003, 0.0, 0.0, -0.001), (1.0, 0.03, 0.0, 0.0, 0.0, 0.0)]
    img = rng.random((120, 168, 3), dtype=np.float32)
    want = img.copy()
    apply_opcode_3_warp(want, _opcode_blob(coeffs[:3]))
    # the flipped / rot180 views RawBayerData.demosaic returns for Grbg / Gbrg / Bggr sensors (image.py:181)
    for view_of in (lambda a: a[::-1], lambda a: a[:, ::-1], lambda a: a[::-1, ::-1]):
        base = np.ascontiguousarray(view_of(img))                # contents such that the VIEW equals img
        v = view_of(base)
        assert not v.flags.c_contiguous and np.array_equal(v, img)
        apply_opcode_3_warp(v, _opcode_blob(coeffs[:3]))         # in place, through the view
        assert np.array_equal(view_of(base), want)
    # float64 image: warped through a float32 copy, written back
    v64 = img.astype(np.float64)
    apply_opcode_3_warp(v64, _opcode_blob(coeffs[:3]))
    assert v64.dtype == np.float64 and np.array_equal(v64.astype(np.float32), want)
    # plane counts 1, 2 and 4: each plane equals the same plane warped inside a 3-plane call with its coefficients
    for n in (1, 2, 4):
        im = rng.random((64, 96, n), dtype=np.float32)
        got = im.copy()
        apply_opcode_3_warp(got, _opcode_blob(coeffs[:n]))
        for p in range(n):
            tri = np.ascontiguousarray(np.repeat(im[:, :, p:p + 1], 3, axis=2))
            apply_opcode_3_warp(tri, _opcode_blob([coeffs[p]] * 3))
            assert np.array_equal(got[:, :, p], tri[:, :, 0])
    # a plane count that does not match the image is skipped, as in the reference (returns False, image untouched)
    keep = img.copy()
    apply_opcode_3_warp(keep, _opcode_blob(coeffs[:2]))
    assert np.array_equal(keep, img)
