"""GPU (-m gpu): checks added in round 4.  Same bars as tests/test_gpu_parity.py (bit-exact vs the oracle unless a tolerance is stated)."""
import numpy as np
import pytest

from conftest import D65_XY, MULT, XYZ2CAM

pytestmark = pytest.mark.gpu


def _wbM(orc):
    return (1.0 / MULT).astype(np.float32), orc.final_matrix(XYZ2CAM, orc.xy_to_XYZ(D65_XY))


def test_lab_grid_injection(orc):
    """the 33^3 Lab table of mode 1 is data.  A table with +-1 LSB on half of its entries (and one with large random entries, which
    drives the homogeneity vote's chroma distances past 2^24 where float32 rounding of the squares matters) is injected into the product context
    (pysp_ctx_set_lab_lut) and into the oracle; AHD with and without the HDR metric, stages 0 and 1, must stay bit-identical; restoring the built-in
    table restores the built-in results."""
    import torch
    from pysp_amd import _lib
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import random_frame, rggb_frame
    wb, M = _wbM(orc)
    pipe = DevicePipeline(0)
    ctx = pipe.ctx
    base = orc.cv410_lut()
    assert np.array_equal(ctx.get_lab_lut(), base)
    with pytest.raises(ValueError):
        g = base.copy(); g[3, 4, 5, 1] = -1
        ctx.set_lab_lut(g)
    assert np.array_equal(ctx.get_lab_lut(), base)                       # a refused table leaves the context untouched
    rng = np.random.default_rng(44)
    lsb = (base.astype(np.int32) + rng.integers(-1, 2, base.shape)).clip(0, 32767).astype(np.int16)
    wild = rng.integers(0, 32768, base.shape).astype(np.int16)
    frames = [rggb_frame(120, 176, 1000), random_frame(90, 134, 3)]
    ref_builtin = [orc.demosaic_ahd(f, wb, M, False, 1) for f in frames]
    try:
        for grid in (lsb, wild):
            ctx.set_lab_lut(grid)
            orc.set_cv410_lut(grid)
            assert np.array_equal(ctx.get_lab_lut(), grid)
            for f in frames:
                d = torch.from_numpy(f).cuda()
                for hdr in (False, True):
                    for st in (0, 1):
                        want = orc.demosaic_ahd(f * (np.float32(3.0) if hdr else np.float32(1.0)), wb, M, hdr, st)
                        dd = d * 3.0 if hdr else d
                        got = pipe.demosaic(dd, wb, M, _lib.QUALITY_BEST, hdr, st).cpu().numpy()
                        assert np.array_equal(got, want, equal_nan=True), (hdr, st)
        changed = [not np.array_equal(orc.demosaic_ahd(f, wb, M, False, 1), r) for f, r in zip(frames, ref_builtin)]
        assert any(changed)                                              # the injected table really reached the metric
    finally:
        ctx.set_lab_lut(None)
        orc.set_cv410_lut(None)
    assert np.array_equal(ctx.get_lab_lut(), base)
    for f, r in zip(frames, ref_builtin):
        assert np.array_equal(pipe.demosaic(torch.from_numpy(f).cuda(), wb, M, _lib.QUALITY_BEST, False, 1).cpu().numpy(), r)
