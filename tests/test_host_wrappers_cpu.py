"""CPU: host-side helpers of the drop-in wrappers that must not change a single bit -- the flat field's plane means taken over views of the
mosaic itself instead of over row copies (raw_correction.py), and the host team's block copies (pysp_amd/_hostpar.py)."""
import numpy as np


def test_flat_field_plane_means_over_direct_views_equal_the_reference_order():
    """reference raw_correction.py:33-40 takes np.mean over the views bayer_chan_mixer.py:4-21 returns: column slices of COPIES of the even / odd rows.
    pysp_amd takes the same means over views of the float32 mosaic itself (no 2 x 48 MB of copies at 24 MP): NumPy walks both the same way, so the
    float32 pairwise sums -- and the means -- are the same bits."""
    from pysp_amd.raw_correction import _plane_views
    rng = np.random.default_rng(5)
    for trial in range(40):
        H, W = 2 * int(rng.integers(1, 500)), 2 * int(rng.integers(1, 700))
        m = (rng.random((H, W), dtype=np.float32) * np.float32(rng.uniform(0.1, 8.0))).astype(np.float32)
        ev, od = m[0::2, :].astype(np.float32), m[1::2, :].astype(np.float32)
        ref = [np.mean(p) for p in (ev[:, 0::2], ev[:, 1::2], od[:, 1::2], od[:, 0::2])]
        got = [np.mean(p) for p in _plane_views(m)]
        assert [r.tobytes() for r in ref] == [g.tobytes() for g in got], (H, W)
    u = (rng.random((6, 8)) * 1000).astype(np.uint16)          # any other dtype goes through float32 first, as the reference's .astype does
    assert [float(np.mean(p)) for p in _plane_views(u)] == [float(np.mean(p)) for p in (u[0::2, 0::2].astype(np.float32), u[0::2, 1::2].astype(np.float32),
                                                                                        u[1::2, 1::2].astype(np.float32), u[1::2, 0::2].astype(np.float32))]


def test_host_team_copies(monkeypatch):
    from pysp_amd import _hostpar, _lib
    rng = np.random.default_rng(6)
    monkeypatch.setattr(_hostpar, "MIN_PARALLEL_ELEMS", 64)
    for team in (1, 3, 8):
        monkeypatch.setenv("PYSP_HOST_THREADS", str(team))
        for src in (rng.random((37, 11, 3)), rng.random((129, 7), dtype=np.float32), (rng.random((50, 50)) * 100).astype(np.int32),
                    rng.random((64, 64), dtype=np.float32)[::2, ::-1], rng.random(1000, dtype=np.float32)):
            dst = np.empty(src.shape, np.float32)
            assert _hostpar.copy_into(dst, src) is dst and np.array_equal(dst, src.astype(np.float32))
            p = _lib.f32_private(src)
            assert p.dtype == np.float32 and p.flags.c_contiguous and np.array_equal(p, np.array(src, dtype=np.float32, order="C", copy=True))
            assert not np.shares_memory(p, src)
    assert _hostpar.team() == 8
    monkeypatch.setenv("PYSP_HOST_THREADS", "0")
    assert _hostpar.team() == 1
    monkeypatch.delenv("PYSP_HOST_THREADS")
    assert 1 <= _hostpar.team() <= 16


def test_debayer_batch_argument_checks_need_no_gpu():
    """debayer_batch (round 5; not part of the reference's API) validates before it touches the library: mixed geometries / camera parameters raise ValueError,
    an unknown quality NotImplementedError (like image.py:176), an empty batch is an empty list."""
    import pytest
    from pysp_amd.const import QualityDemosaic
    from pysp_amd.debayer import debayer_batch
    from pysp_amd.image import RawRggbBayerData
    from pysp_amd.synth import default_wb, rggb_frame
    wb = default_wb()
    a, b = RawRggbBayerData(rggb_frame(8, 8, 1), wb, 10.0, 1.0), RawRggbBayerData(rggb_frame(8, 10, 1), wb, 10.0, 1.0)
    assert debayer_batch([], QualityDemosaic.Best) == []
    with pytest.raises(ValueError):
        debayer_batch([a, b], QualityDemosaic.Best)
    with pytest.raises(ValueError):
        debayer_batch([a], QualityDemosaic.Best, to="xyz")
    with pytest.raises(NotImplementedError):
        debayer_batch([a], 9)
    class OtherWb:                                            # cam_wb is duck-typed on this path: another white balance, same matrix
        def get_reciprocal_multipliers(self): return (wb.get_reciprocal_multipliers() * np.float32(1.25)).astype(np.float32)
        def get_matrix(self): return wb.get_matrix()
        def copy(self): return self
    c = RawRggbBayerData(rggb_frame(8, 8, 2), OtherWb(), 10.0, 1.0)
    with pytest.raises(ValueError):
        debayer_batch([a, c], QualityDemosaic.Fast)
