"""GPU (-m gpu): the HIP path, called through the C ABI (pysp_amd -> libpysp_hip.so), against
  (1) the golden fixtures produced by the reference's own code, and
  (2) the CPU oracle on the same seeded inputs.
Bars: bit-exact for Bayer indexing, votes, selection, medians and every float32 stage whose
operations are +,-,*,/ (the kernels are built with -ffp-contract=off and evaluate in the oracle's
order); <= 1 ULP for the two stages that go through a float64 pow (sRGB curves); stated tolerance
for WarpRectilinear, whose reference arithmetic calls libm powf.
"""
import json

import numpy as np
import pytest

from conftest import D65_XY, MULT, XYZ2CAM, load_golden, ulp_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wbobj():
    from pysp_amd.synth import default_wb
    return default_wb()


def _raw(bayer, wbobj, hdr=False, ev=10.0):
    from pysp_amd.image import RawRggbBayerData
    im = RawRggbBayerData(bayer, wbobj, ev, 1.0)
    im.set_hdr(hdr)
    return im


def _wbM(orc):
    return (1.0 / MULT).astype(np.float32), orc.final_matrix(XYZ2CAM, orc.xy_to_XYZ(D65_XY))


# ---- Bayer helpers ---------------------------------------------------------------------------------
def test_demux_remux_normalize_golden():
    from pysp_amd.bayer_chan_mixer import bayer_to_rgbg, rgbg_to_bayer
    from pysp_amd.normalization import bayer_normalize
    d, _ = load_golden("g1_demux")
    for src in ("f32", "u16"):
        for p, k in zip(bayer_to_rgbg(d[src]), ("r", "g1", "b", "g2")):
            assert p.dtype == np.float32 and np.array_equal(p, d[f"{src}_{k}"])
    assert np.array_equal(rgbg_to_bayer(*bayer_to_rgbg(d["f32"])), d["remux"])
    g, _ = load_golden("g6_normalize")
    assert np.array_equal(bayer_normalize(g["raw"], g["black"].tolist(), g["sat"].tolist()), g["out"])


def test_demux_remux_large_roundtrip():
    from pysp_amd.bayer_chan_mixer import bayer_to_rgbg, rgbg_to_bayer
    a = np.random.default_rng(0).random((1026, 2050), dtype=np.float32)
    planes = bayer_to_rgbg(a)
    assert np.array_equal(planes[0], a[0::2, 0::2]) and np.array_equal(planes[2], a[1::2, 1::2])
    assert np.array_equal(rgbg_to_bayer(*planes), a)
    with pytest.raises(ValueError):
        bayer_to_rgbg(a[:-1])


# ---- homogeneity vote --------------------------------------------------------------------------------
def test_build_map_golden_and_oracle(orc):
    from pysp_amd.debayer import build_map
    d, _ = load_golden("g3_build_map")
    for name in ("lab", "labq"):
        assert np.array_equal(build_map(d[name], 1, 3, False), d[name + "_h"])
        assert np.array_equal(build_map(d[name], 1, 3, True), d[name + "_v"])
    rng = np.random.default_rng(5)
    lab = (rng.random((203, 331, 3)) * 40).astype(np.float32)
    lab[50:90, 100:200] = np.round(lab[50:90, 100:200])       # many exact ties
    for v in (False, True):
        assert np.array_equal(build_map(lab, 1, 3, v), orc.build_map(lab, 1, v))
    assert np.array_equal(build_map(np.ones((9, 9, 3), np.float32), 1, 3, False), np.full((7, 7), 9, np.float32))
    with pytest.raises(ValueError):
        build_map(lab.astype(np.float64), 1, 3, False)


# ---- colour ---------------------------------------------------------------------------------------------
def test_cam_to_lin_srgb_golden():
    from pysp_amd.colorize.transform import cam_to_lin_srgb
    from pysp_amd.wb_cct.helpers_cam_mat import MatXyzToCamera
    d, _ = load_golden("g4_cam_to_rgb")
    for i in range(3):
        mat = MatXyzToCamera(d[f"m{i}"], d[f"white{i}"])
        assert np.array_equal(cam_to_lin_srgb(d["px"], mat, True), d[f"out{i}_clip"])
        assert np.array_equal(cam_to_lin_srgb(d["px"], mat, False), d[f"out{i}_noclip"])


def test_gamma(orc):
    from pysp_amd.colorize import lin_srgb_to_srgb, srgb_to_lin_srgb
    d, _ = load_golden("g5_gamma")
    enc = lin_srgb_to_srgb(d["x"])
    # vs the correctly rounded oracle: float64 pow on both sides -> identical but for double-rounding ties
    u = ulp_diff(enc, orc.lin_srgb_to_srgb(d["x"]))
    assert u.max() <= 1 and np.mean(u != 0) < 1e-4
    # vs NumPy's float32 power on the fixture box (platform-dependent last bit, amplified <= 4 ULP at the toe)
    assert ulp_diff(enc, d["enc"]).max() <= 4
    dec = srgb_to_lin_srgb(d["x"])
    u = ulp_diff(dec, orc.srgb_to_lin_srgb(d["x"]))
    assert u.max() <= 1 and np.mean(u != 0) < 1e-4
    assert ulp_diff(dec, d["dec"]).max() <= 1
    x = np.random.default_rng(1).random((257, 129, 3), dtype=np.float32) * 1.2 - 0.1
    assert ulp_diff(lin_srgb_to_srgb(x), orc.lin_srgb_to_srgb(x)).max() <= 1


# ---- demosaic: fixtures from the reference orchestration -----------------------------------------------------
@pytest.mark.parametrize("name", ["g8_demosaic_32x48", "g8_demosaic_34x50", "g8_demosaic_32x48_hdr", "g8_demosaic_34x50_hdr"])
def test_demosaic_golden(name, wbobj):
    from pysp_amd.colorize import lin_srgb_to_srgb
    from pysp_amd.const import QualityDemosaic
    d, meta = load_golden(name)
    hdr = meta["hdr"]
    if not hdr:
        assert np.array_equal(_raw(d["bayer"], wbobj).demosaic(QualityDemosaic.Draft).image, d["draft"])
        assert np.array_equal(_raw(d["bayer"], wbobj).demosaic(QualityDemosaic.Fast).image, d["eag"])
    for st in (0, 1, 3):
        dem = _raw(d["bayer"], wbobj, hdr).demosaic(QualityDemosaic.Best, st)
        assert dem.image.dtype == np.float32 and np.array_equal(dem.image, d[f"ahd{st}"]), st
        if st == 1:
            lin = dem.to_lin_srgb()
            assert np.array_equal(lin, d["ahd1_lin"])
            srgb = lin_srgb_to_srgb(lin / (1 + lin) if hdr else lin)
            assert ulp_diff(srgb, d["ahd1_srgb"]).max() <= 4


def test_cfa_patterns_golden(wbobj):
    from pysp_amd.base_types.image_base import BayerPattern
    from pysp_amd.const import QualityDemosaic
    from pysp_amd.image import RawBayerData
    d, _ = load_golden("g8_cfa_patterns")
    for pat in BayerPattern:
        rb = RawBayerData()
        rb.sensor_scaled = d["bayer"]; rb.cam_wb = wbobj; rb.current_ev = 9.0; rb.sensor_pattern = pat
        out = rb.demosaic(QualityDemosaic.Fast)
        assert np.array_equal(out.image, d[f"eag_{pat.name}"]), pat
        assert out.current_ev == 9.0 and out.mat_xyz is not None
        assert np.array_equal(rb.demosaic(QualityDemosaic.Draft).image, d[f"draft_{pat.name}"]), pat
        assert np.array_equal(rb.demosaic(QualityDemosaic.Best, 1).image, d[f"ahd1_{pat.name}"]), pat


def test_unknown_quality_raises(wbobj):
    with pytest.raises(NotImplementedError):
        _raw(np.zeros((4, 4), np.float32), wbobj).demosaic("best")
    from pysp_amd import _lib
    import ctypes
    rc = _lib.lib().pysp_demosaic_f32(_lib.default_context().handle, None, 4, 4, _lib.wb3([1, 1, 1]), None, 7, 0, 0, None)
    assert rc == _lib.PYSP_EBADARG
    a = np.zeros((4, 4), np.float32); o = np.zeros((4, 4, 3), np.float32)
    rc = _lib.lib().pysp_demosaic_f32(_lib.default_context().handle, _lib.ptr(a), 4, 4, _lib.wb3([1, 1, 1]), None, 7, 0, 0, _lib.ptr(o))
    assert rc == _lib.PYSP_ENOTIMPL and "not implemented" in _lib.last_error()
    rc = _lib.lib().pysp_demosaic_f32(_lib.default_context().handle, _lib.ptr(a), 3, 4, _lib.wb3([1, 1, 1]), None, 0, 0, 0, _lib.ptr(o))
    assert rc == _lib.PYSP_EBADARG


# ---- demosaic vs the oracle on seeded inputs: ragged sizes, tile seams, tiny frames, worst-case noise -------------
SIZES = [(2, 2), (2, 4), (4, 2), (4, 6), (6, 134), (70, 2), (62, 66), (64, 128), (66, 130), (130, 198), (256, 320)]


@pytest.mark.parametrize("H,W", SIZES)
def test_demosaic_vs_oracle_sizes(orc, wbobj, H, W):
    from pysp_amd.const import QualityDemosaic
    from pysp_amd.synth import random_frame, rggb_frame
    wb, M = _wbM(orc)
    for bay in (rggb_frame(H, W, 1000 + H + W), random_frame(H, W, H * W)):
        assert np.array_equal(_raw(bay, wbobj).demosaic(QualityDemosaic.Draft).image, orc.demosaic_draft(bay, wb))
        assert np.array_equal(_raw(bay, wbobj).demosaic(QualityDemosaic.Fast).image, orc.demosaic_eag(bay, wb))
        for st in (0, 1, 2):
            got = _raw(bay, wbobj).demosaic(QualityDemosaic.Best, st).image
            assert np.array_equal(got, orc.demosaic_ahd(bay, wb, M, False, st)), (H, W, st)


@pytest.mark.parametrize("H,W", [(66, 130), (200, 264)])
def test_demosaic_hdr_vs_oracle(orc, wbobj, H, W):
    from pysp_amd.const import QualityDemosaic
    from pysp_amd.synth import rggb_frame
    wb, M = _wbM(orc)
    bay = rggb_frame(H, W, 77, scale=4.0, clip_hi=False)
    for st in (0, 1, 3):
        got = _raw(bay, wbobj, hdr=True).demosaic(QualityDemosaic.Best, st).image
        assert np.array_equal(got, orc.demosaic_ahd(bay, wb, M, True, st))


def test_ahd_medium_frame_and_fused_pipeline(orc, wbobj):
    """1.5 MP frame: staged API (3 materialising calls) and the fused single-call pipeline both match."""
    from pysp_amd import _lib
    from pysp_amd.colorize import lin_srgb_to_srgb
    from pysp_amd.const import QualityDemosaic
    from pysp_amd.synth import rggb_frame
    H, W = 1000, 1500
    bay = rggb_frame(H, W, 1001)
    wb, M = _wbM(orc)
    ref_rgb = orc.demosaic_ahd(bay, wb, M, False, 1)
    dem = _raw(bay, wbobj).demosaic(QualityDemosaic.Best, 1)
    assert np.array_equal(dem.image, ref_rgb)
    lin = dem.to_lin_srgb()
    assert np.array_equal(lin, orc.cam_to_rgb(ref_rgb, M, True))
    ref_srgb = orc.pipeline_srgb(bay, wb, M, 2, False, 1, False)
    u = ulp_diff(lin_srgb_to_srgb(lin), ref_srgb)
    assert u.max() <= 1 and np.mean(u != 0) < 1e-4
    fused = np.empty((H, W, 3), np.float32)
    for q, orc_q in ((_lib.QUALITY_BEST, 2), (_lib.QUALITY_FAST, 1), (_lib.QUALITY_DRAFT, 0)):
        _lib.check(_lib.lib().pysp_pipeline_srgb_f32(_lib.default_context().handle, _lib.ptr(bay), H, W, _lib.wb3(wb), _lib.mat9(M), q, 0, 1, 0,
                                                    _lib.ptr(fused)))
        u = ulp_diff(fused, orc.pipeline_srgb(bay, wb, M, orc_q, False, 1, False))
        assert u.max() <= 1 and np.mean(u != 0) < 1e-4, q


# ---- size-independent properties at the benchmark's full size (24 MP) ---------------------------------------------
def test_full_size_24mp_properties(orc, wbobj):
    from pysp_amd import _lib
    from pysp_amd.synth import rggb_frame
    H, W = 4000, 6000
    bay = rggb_frame(H, W, 1000)
    wb, M = _wbM(orc)
    full = np.empty((H, W, 3), np.float32)
    L, ctx = _lib.lib(), _lib.default_context()
    _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(bay), H, W, _lib.wb3(wb), _lib.mat9(M), _lib.QUALITY_BEST, 0, 1, 0, _lib.ptr(full)))
    assert np.isfinite(full).all() and full.min() >= 0 and full.max() <= 1
    # (a) tile/translation invariance: a crop with >= 16 px margin reproduces the interior of the full frame,
    #     wherever it falls relative to the 64x32 tile grid (offsets are even so the CFA phase is kept)
    for (y0, x0, h, w) in ((0, 0, 300, 420), (1234, 2022, 310, 402), (H - 300, W - 420, 300, 420), (2000, 0, 260, 300)):
        crop = np.ascontiguousarray(bay[y0:y0 + h, x0:x0 + w])
        out = np.empty((h, w, 3), np.float32)
        _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(crop), h, w, _lib.wb3(wb), _lib.mat9(M), _lib.QUALITY_BEST, 0, 1, 0, _lib.ptr(out)))
        ys = slice(0 if y0 == 0 else 16, h if y0 + h == H else h - 16)
        xs = slice(0 if x0 == 0 else 16, w if x0 + w == W else w - 16)
        assert np.array_equal(out[ys, xs], full[y0:y0 + h, x0:x0 + w][ys, xs]), (y0, x0)
        # (b) the same crop against the oracle, bit for bit up to the pow stage
        u = ulp_diff(out, orc.pipeline_srgb(crop, wb, M, 2, False, 1, False))
        assert u.max() <= 1 and np.mean(u != 0) < 1e-4
    # (c) a constant mosaic in camera-neutral proportions comes out as one constant grey
    flat = np.empty((128, 192), np.float32)
    flat[0::2, 0::2] = 0.2 * 0.5; flat[0::2, 1::2] = 0.2; flat[1::2, 0::2] = 0.2; flat[1::2, 1::2] = 0.2 * 0.7
    out = np.empty((128, 192, 3), np.float32)
    _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(flat), 128, 192, _lib.wb3(wb), _lib.mat9(M), _lib.QUALITY_BEST, 0, 1, 0, _lib.ptr(out)))
    assert np.ptp(out) < 1e-6


@pytest.mark.gpu
def test_whole_frame_24mp_device_launch(orc, wbobj):
    """The benchmark's own launch (missing 4): pysp_pipeline_dev on the WHOLE 4000x6000 frame -- one select grid of
    215x143 tiles, one median grid -- equals (a) the banded host entry (256-row bands, other grids) bit for bit over the full frame and
    (b) the oracle on crops that cover tile rows / columns at the top-left, the interior, the bottom-right and the frame's last tiles."""
    import ctypes
    from pysp_amd import _lib
    from pysp_amd.synth import rggb_frame
    H, W = 4000, 6000
    bay = rggb_frame(H, W, 1000)
    wb, M = _wbM(orc)
    L, ctx = _lib.lib(), _lib.default_context()
    banded = np.empty((H, W, 3), np.float32)
    _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(bay), H, W, _lib.wb3(wb), _lib.mat9(M), _lib.QUALITY_BEST, 0, 1, 0, _lib.ptr(banded)))
    for tail, ref in ((2, banded), (0, None)):
        import torch
        dctx = _lib.Context(0)
        t_in = torch.from_numpy(bay).to("cuda:0")
        t_out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        _lib.check(L.pysp_pipeline_dev(dctx.handle, ctypes.c_void_p(t_in.data_ptr()), H, W, _lib.wb3(wb), _lib.mat9(M), _lib.QUALITY_BEST, 0, 1, tail, ctypes.c_void_p(t_out.data_ptr())))
        dctx.sync()
        whole = t_out.cpu().numpy()
        del t_in, t_out
        if ref is not None:
            assert np.array_equal(whole, ref)
        # oracle on crops (margin 16 px inside the crop unless the crop touches the frame edge)
        for (y0, x0, h, w) in ((0, 0, 200, 260), (1986, 2990, 180, 200), (H - 200, W - 260, 200, 260), (0, W - 200, 150, 200), (H - 150, 0, 150, 200)):
            crop = np.ascontiguousarray(bay[y0:y0 + h, x0:x0 + w])
            want = orc.pipeline_srgb(crop, wb, M, 2, False, 1, False) if tail == 2 else orc.demosaic_ahd(crop, wb, M, False, 1)
            ys = slice(0 if y0 == 0 else 16, h if y0 + h == H else h - 16)
            xs = slice(0 if x0 == 0 else 16, w if x0 + w == W else w - 16)
            got = whole[y0:y0 + h, x0:x0 + w][ys, xs]
            if tail == 0:
                assert np.array_equal(got, want[ys, xs]), (tail, y0, x0)
            else:       # the bar of test_full_size_24mp_properties for the stage behind the float64 pow
                u = ulp_diff(got, want[ys, xs])
                assert u.max() <= 1 and np.mean(u != 0) < 1e-4, (tail, y0, x0)


# ---- HDR raw fusion ---------------------------------------------------------------------------------------------------
def test_fuse_raw_golden_and_oracle(orc, wbobj):
    from pysp_amd.raw_hdr import fuse_exposures_to_raw
    from pysp_amd.synth import rggb_frame
    d, meta = load_golden("g9_fuse_raw")
    exps = [_raw(f, wbobj, ev=ev) for f, ev in zip(d["frames"], meta["evs"])]
    hdr, cnt = fuse_exposures_to_raw(exps)
    assert np.array_equal(hdr.sensor_scaled, d["fused"]) and np.array_equal(cnt, d["count"]) and cnt.dtype == np.int32
    assert hdr.get_hdr() and hdr.current_ev == meta["target_ev"] and hdr.lim_sat == meta["lim_sat"]
    assert fuse_exposures_to_raw([]) is None
    # 7 exposures, ragged width (W % 4 == 2), against the oracle
    H, W, K = 130, 202, 7
    base = rggb_frame(H, W, 5, scale=6.0, clip_hi=False)
    evs = [10.0 + k for k in range(K)]
    frames = [np.clip(base * np.float32(2.0 ** -k), 0, 1).astype(np.float32) for k in range(K)]
    frames[0][:8, :8] = 1.0
    for f in frames[1:]:
        f[:4, :4] = 1.0
    hdr, cnt = fuse_exposures_to_raw([_raw(f, wbobj, ev=e) for f, e in zip(frames, evs)])
    ref, refc, target, lim = orc.fuse_raw(frames, evs, 1.0 / MULT)
    assert np.array_equal(hdr.sensor_scaled, ref) and np.array_equal(cnt, refc)
    assert hdr.current_ev == target and hdr.lim_sat == lim


def test_fuse_from_debayer_golden_and_oracle(orc, wbobj):
    from pysp_amd.base_types.image_base import RawDemosaicData
    from pysp_amd.raw_hdr import fuse_exposures_from_debayer
    d, meta = load_golden("g9_fuse_debayer")

    def mk(img, ev):
        e = RawDemosaicData(img.copy(), (1.0 / d["mult"]).astype(np.float32))
        e.mat_xyz = wbobj.get_matrix(); e.current_ev = ev
        return e
    exps = [mk(a, ev) for a, ev in zip(d["rgb"], meta["evs"])]
    fused, cnt = fuse_exposures_from_debayer(exps)
    assert np.array_equal(fused, d["fused"]) and np.array_equal(cnt, d["count"]) and cnt.dtype == np.int32
    M = orc.final_matrix(XYZ2CAM, orc.xy_to_XYZ(D65_XY))
    _, _, left = orc.fuse_rgb(list(d["rgb"]), meta["evs"], 1.0 / d["mult"], M)
    for e, a in zip(exps, left):                      # side effect of the reference: undo/apply round trip
        assert e._wb_applied and np.array_equal(e.image, a)
    assert fuse_exposures_from_debayer([]) is None
    rng = np.random.default_rng(8)
    imgs = [np.clip(rng.random((77, 131, 3), dtype=np.float32) * (2.0 ** -k) * 1.5, 0, None).astype(np.float32) for k in range(5)]
    imgs[0][:5, :5] = 0.0
    for a in imgs[1:]:
        a[:5, :5] = 0.0
    evs = [8.0, 9.0, 10.0, 11.0, 11.0]                # two exposures share the maximum offset: the last one wins
    fused, cnt = fuse_exposures_from_debayer([mk(a, ev) for a, ev in zip(imgs, evs)])
    ref, refc, left = orc.fuse_rgb(imgs, evs, 1.0 / d["mult"], M)
    assert np.array_equal(fused, ref) and np.array_equal(cnt, refc)
    # round 3: the fusion is device resident.  Exposures whose image is still in HBM (as a demosaic leaves it) are fused there, host images are
    # uploaded straight from the caller's arrays (which stay untouched), results and round-tripped images stay lazy until read; PYSP_EAGER
    # semantics (set_lazy(False)) hand out plain ndarrays.  Same bits every way.
    import pysp_amd
    from pysp_amd.device_array import DeviceArray
    from pysp_amd import _lib as _plib
    ctx = _plib.default_context()
    originals = [a.copy() for a in imgs]
    was = pysp_amd.lazy_enabled()
    try:
        for lazy in (True, False):
            pysp_amd.set_lazy(lazy)
            exps = [mk(a, ev) for a, ev in zip(imgs, evs)]
            for k in (0, 2, 4):                           # some exposures resident, some on the host, one not white balanced
                exps[k].image = DeviceArray.from_host(ctx, imgs[k])
            exps[1]._wb_applied = False
            host_in = exps[1]._img
            fused, cnt = fuse_exposures_from_debayer(exps)
            assert isinstance(fused, DeviceArray if lazy else np.ndarray) and isinstance(cnt, np.ndarray) and cnt.dtype == np.int32
            ref1, refc1, left1 = orc.fuse_rgb(imgs, evs, 1.0 / d["mult"], M, applied=[True, False, True, True, True])
            assert np.array_equal(np.asarray(fused), ref1) and np.array_equal(cnt, refc1)
            assert np.array_equal(host_in, imgs[1]) and all(np.array_equal(a, o) for a, o in zip(imgs, originals))      # inputs untouched
            for e, a in zip(exps, left1):
                assert e._wb_applied and not e._wb_normalized
                assert (e._dev is not None) == lazy and np.array_equal(e.image, a) and isinstance(e.image, np.ndarray)
    finally:
        pysp_amd.set_lazy(was)


# ---- WarpRectilinear ------------------------------------------------------------------------------------------------------
def _opcode_coeffs(blob: bytes):
    """(planes x 6 coefficient rows, (cx, cy)) of the first WarpRectilinear opcode of an OpcodeList3 blob (chan_distortion_corr.py:74-83,102-121)."""
    import struct
    count = struct.unpack(">I", blob[:4])[0]
    off = 4
    for _ in range(count):
        oid, _v, _f, n = struct.unpack(">4I", blob[off:off + 16])
        off += 16
        if oid == 1:
            planes = struct.unpack(">I", blob[off:off + 4])[0]
            vals = struct.unpack(">%dd" % (planes * 6 + 2), blob[off + 4:off + 4 + 8 * (planes * 6 + 2)])
            cf = np.array(vals[:planes * 6]).reshape(planes, 6)
            if planes == 1:
                cf = np.repeat(cf, 3, axis=0)
            return cf, (vals[-2], vals[-1])
        off += n
    raise AssertionError("no WarpRectilinear opcode in the blob")


def test_warp_table(orc):
    """The reference evaluates r**4 and r**6 with libm powf (not always correctly rounded); the kernel
    uses exactly rounded products, so single coordinates may differ in the last bit."""
    from pysp_amd.dng_warp_corr import compute_offset_remapping_table, compute_remapping_table
    d, meta = load_golden("g7_warp_table")
    a = meta["args"]
    t = compute_remapping_table(a["kr0"], a["kr1"], a["kr2"], a["kr3"], a["kt0"], a["kt1"], a["width"], a["height"], a["cx"], a["cy"], a["scale"])
    assert t.shape == d["table"].shape and ulp_diff(t, d["table"]).max() <= 2
    s = meta["seeded_args"]
    t1 = compute_offset_remapping_table(d["table"], s[0], s[1], s[2], s[3], s[4], s[5], a["width"], a["height"], s[6], s[7], s[8])
    assert ulp_diff(t1, d["seeded"]).max() <= 2
    big = compute_remapping_table(1.0, 0.01, 0.002, 0.0, 0.0, 0.0, 1201, 801, 0.5, 0.5, 1.0)
    ref = orc.warp_table(1.0, 0.01, 0.002, 0.0, 0.0, 0.0, 1201, 801, 0.5, 0.5, 1.0)
    u = ulp_diff(big, ref)
    assert u.max() <= 2 and np.mean(u != 0) < 0.02


def test_warp_apply(orc):
    from pysp_amd.dng_warp_corr import apply_opcode_3_warp
    from oracle.checks import warp_phase_check
    d, _ = load_golden("g10_warp_apply")
    img = d["image"].copy()
    apply_opcode_3_warp(img, d["blob"].tobytes())
    # The fixture is the reference's own orchestration (chan_distortion_corr.py:43-121 with the compiled Cython table builder).  A table coordinate may
    # differ from the reference's in its last two bits (powf there, exact products here); the CLASSIFIED check (oracle/checks.py): every value that differs
    # from the fixture has its coordinate within 2 ULP of a 1/32-px quantisation boundary of cv2.remap AND is the Lanczos-4 interpolation at that neighbouring
    # phase, bit for bit; everything else is bit-identical.  (Until round 4: mean(diff > 0) < 2e-3 and max < 2e-2, which any small displacement would pass.)
    cf, centre = _opcode_coeffs(d["blob"].tobytes())
    st = warp_phase_check(img, d["image"], cf, centre, expected=d["warped"])
    assert st["differing_outside_boundary_set"] == 0 and st["differing_not_a_neighbouring_phase"] == 0
    rng = np.random.default_rng(3)
    big = rng.random((301, 402, 3), dtype=np.float32)
    coeffs = np.array([[1.0, 0.01, 0.002, 0.0, 0.0, 0.0], [1.0, -0.01, 0.002, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0, 0.001, -0.001]])
    import struct
    payload = struct.pack(">I", 3) + b"".join(struct.pack(">6d", *c) for c in coeffs) + struct.pack(">2d", 0.5, 0.5)
    blob = struct.pack(">I", 1) + struct.pack(">IIII", 1, 1, 0, len(payload)) + payload
    got = big.copy()
    apply_opcode_3_warp(got, blob)
    st = warp_phase_check(got, big, coeffs, (0.5, 0.5))
    assert st["differing_outside_boundary_set"] == 0 and st["differing_not_a_neighbouring_phase"] == 0 and st["frac_differing"] < 5e-3
    ident = big.copy()      # kr0 = 1, everything else 0 -> identity mapping -> exact copy
    payload = struct.pack(">I", 3) + struct.pack(">6d", 1, 0, 0, 0, 0, 0) * 3 + struct.pack(">2d", 0.5, 0.5)
    apply_opcode_3_warp(ident, struct.pack(">I", 1) + struct.pack(">IIII", 1, 1, 0, len(payload)) + payload)
    assert np.mean(ident != big) < 1e-3


# ---- BASELINE configs 3-5 end to end, device resident (pysp_amd.pipeline) ------------------------------------------------
def test_config3_batch_eag_device_resident(orc, wbobj):
    """Batch of frames, EAG + WB + CCM, frames sharded round-robin (here: the shard of rank 1 of 2)."""
    import torch
    from pysp_amd import _lib
    from pysp_amd.multi_gpu import frames_for_rank
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import rggb_frame
    pipe = DevicePipeline(0)
    wb, M = _wbM(orc)
    H, W = 500, 760
    mine = frames_for_rank(8, 1, 2)
    assert mine == [1, 3, 5, 7]
    outs = []
    for i in mine:
        bay = torch.from_numpy(rggb_frame(H, W, 1000 + i)).cuda()
        outs.append((i, pipe.demosaic_to_srgb(bay, wb, M, quality=_lib.QUALITY_FAST)))
    pipe.sync()
    for i, o in outs:
        u = ulp_diff(o.cpu().numpy(), orc.pipeline_srgb(rggb_frame(H, W, 1000 + i), wb, M, 1, False, 0, False))
        assert u.max() <= 1 and np.mean(u != 0) < 1e-4
    # the same shard through the batch entry point (one library call), EAG + WB + CCM = config 3's per-frame work
    frames = [torch.from_numpy(rggb_frame(H, W, 1000 + i)).cuda() for i in mine]
    lin = pipe.batch(frames, wb, M, quality=_lib.QUALITY_FAST, tail=1)
    pipe.sync()
    for i, o in zip(mine, lin):
        assert np.array_equal(o.cpu().numpy(), orc.cam_to_rgb(orc.demosaic_eag(rggb_frame(H, W, 1000 + i), wb), M, True))
    assert pipe.batch([], wb, M) == []
    with pytest.raises(ValueError):
        pipe.batch([frames[0], frames[1][:100].contiguous()], wb, M)


def test_config4_hdr_stack_device_resident(orc, wbobj):
    """7 exposures one stop apart -> raw fusion -> AHD(HDR) -> to_lin_srgb -> Reinhard -> sRGB."""
    import torch
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import rggb_frame
    pipe = DevicePipeline(0)
    wb, M = _wbM(orc)
    H, W, K = 412, 618, 7
    base = rggb_frame(H, W, 1000, scale=8.0, clip_hi=False)
    evs = [10.0 + k for k in range(K)]
    frames = [np.clip(base * np.float32(2.0 ** -k), 0, 1).astype(np.float32) for k in range(K)]
    out, cnt = pipe.hdr_stack_to_srgb([torch.from_numpy(f).cuda() for f in frames], evs, wbobj)
    pipe.sync()
    fused, refc, _, _ = orc.fuse_raw(frames, evs, wb)
    assert np.array_equal(cnt.cpu().numpy(), refc)
    ref = orc.pipeline_srgb(fused, wb, M, 2, True, 1, True)
    u = ulp_diff(out.cpu().numpy(), ref)
    assert u.max() <= 1 and np.mean(u != 0) < 1e-4


def test_config5_ahd3_warp_and_band_tiling(orc, wbobj):
    """AHD(postprocess_stages=3) + per-channel WarpRectilinear; and the intra-frame sharding rule: bands
    with a 20-row halo taken from the input reproduce the whole-frame demosaic exactly."""
    import torch
    from pysp_amd.multi_gpu import band_ranges
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import rggb_frame
    pipe = DevicePipeline(0)
    wb, M = _wbM(orc)
    H, W = 600, 900
    bay = rggb_frame(H, W, 1000)
    coeffs = np.array([[1.0, 0.01, 0.002, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0, 0.0, 0.0], [1.0, -0.01, 0.002, 0.0, 0.0, 0.0]])
    dbay = torch.from_numpy(bay).cuda()
    rgb = pipe.demosaic(dbay, wb, M, stages=3)
    warped = pipe.warp(rgb, coeffs, (0.5, 0.5))
    pipe.sync()
    ref_rgb = orc.demosaic_ahd(bay, wb, M, False, 3)
    assert np.array_equal(rgb.cpu().numpy(), ref_rgb)
    from oracle.checks import warp_phase_check
    st = warp_phase_check(warped.cpu().numpy(), ref_rgb, coeffs, (0.5, 0.5))      # every differing value: a neighbouring Lanczos phase at a 1/32-px boundary, nothing else
    assert st["differing_outside_boundary_set"] == 0 and st["differing_not_a_neighbouring_phase"] == 0
    # band tiling: 8 bands, halo 20 rows (7 + 4 per stage, rounded to the CFA), true borders keep their rules
    whole = rgb.cpu().numpy()
    for (y0, y1, r0, r1) in band_ranges(H, 8, halo=20):
        part = pipe.demosaic(dbay[r0:r1].contiguous(), wb, M, stages=3)
        pipe.sync()
        assert np.array_equal(part.cpu().numpy()[y0 - r0:y1 - r0], whole[y0:y1]), (y0, y1)


def test_uint16_fused_loader(orc, wbobj):
    """bayer_normalize fused into the tile loaders of all three demosaic kernels."""
    import torch
    from pysp_amd import _lib
    from pysp_amd.normalization import raw_to_rgb
    from pysp_amd.pipeline import DevicePipeline
    wb, M = _wbM(orc)
    rng = np.random.default_rng(21)
    black, sat = [64, 64, 66, 64], [4095, 4000, 4095, 4095]
    for (H, W) in ((2, 6), (66, 130), (300, 420)):
        raw = rng.integers(0, 4500, (H, W), dtype=np.uint16)
        norm = orc.bayer_normalize(raw, black, sat)
        assert np.array_equal(raw_to_rgb(raw, black, sat, wbobj, _lib.QUALITY_DRAFT), orc.demosaic_draft(norm, wb))
        assert np.array_equal(raw_to_rgb(raw, black, sat, wbobj, _lib.QUALITY_FAST), orc.demosaic_eag(norm, wb))
        for st in (0, 1):
            assert np.array_equal(raw_to_rgb(raw, black, sat, wbobj, _lib.QUALITY_BEST, st), orc.demosaic_ahd(norm, wb, M, False, st))
        u = ulp_diff(raw_to_rgb(raw, black, sat, wbobj, _lib.QUALITY_BEST, 1, tail=2), orc.pipeline_srgb(norm, wb, M, 2, False, 1, False))
        assert u.max() <= 1 and np.mean(u != 0) < 1e-3
    pipe = DevicePipeline(0)
    raw = rng.integers(0, 4500, (512, 768), dtype=np.uint16)
    out = pipe.raw_u16_to_rgb(torch.from_numpy(raw.view(np.int16)).cuda().view(torch.uint16), black, sat, wb, M, tail=1)
    pipe.sync()
    norm = orc.bayer_normalize(raw, black, sat)
    assert np.array_equal(out.cpu().numpy(), orc.cam_to_rgb(orc.demosaic_ahd(norm, wb, M, False, 1), M, True))


# ---- BASELINE configs 4 and 5 at their full sizes: size-independent properties -------------------------------------------
def test_full_size_45mp_hdr_stack(orc, wbobj):
    """7 x 45 MP exposures (8192x5464) fused on the device, AHD(HDR) + Reinhard + sRGB; crops of the fused
    mosaic and of the final image are checked against the oracle, the whole frame for sanity."""
    import torch
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import rggb_frame
    pipe = DevicePipeline(0)
    wb, M = _wbM(orc)
    H, W, K = 5464, 8192, 7
    base = rggb_frame(H, W, 1004, scale=8.0, clip_hi=False)
    evs = [10.0 + k for k in range(K)]
    frames = [torch.from_numpy(np.clip(base * np.float32(2.0 ** -k), 0, 1)).cuda() for k in range(K)]
    fused, cnt, target, lim = pipe.fuse_raw(frames, evs, wb)
    out, _ = pipe.hdr_stack_to_srgb(frames, evs, wbobj)
    pipe.sync()
    fused_h = fused.cpu().numpy()
    assert target == 13.0 and lim == 8.0 and int(cnt.max()) <= K and int(cnt.min()) >= 0
    for (y0, x0, h, w) in ((0, 0, 200, 320), (2700, 4000, 220, 300), (H - 200, W - 320, 200, 320)):
        crops = [np.ascontiguousarray(np.clip(base[y0:y0 + h, x0:x0 + w] * np.float32(2.0 ** -k), 0, 1)) for k in range(K)]
        ref, refc, _, _ = orc.fuse_raw(crops, evs, wb)
        assert np.array_equal(fused_h[y0:y0 + h, x0:x0 + w], ref)             # pointwise: crops are exact
        assert np.array_equal(cnt[y0:y0 + h, x0:x0 + w].cpu().numpy(), refc)
        # final image: interior of the crop (16 px margin unless the crop touches the true border)
        ref_srgb = orc.pipeline_srgb(np.ascontiguousarray(fused_h[y0:y0 + h, x0:x0 + w]), wb, M, 2, True, 1, True)
        ys = slice(0 if y0 == 0 else 16, h if y0 + h == H else h - 16)
        xs = slice(0 if x0 == 0 else 16, w if x0 + w == W else w - 16)
        u = ulp_diff(out[y0:y0 + h, x0:x0 + w].cpu().numpy()[ys, xs], ref_srgb[ys, xs])
        assert u.max() <= 1 and np.mean(u != 0) < 1e-4
    o = out.cpu().numpy()
    assert np.isfinite(o).all() and o.min() >= 0 and o.max() <= 1


def test_full_size_100mp_ahd3_warp(orc, wbobj):
    """100 MP medium-format frame (11648x8736): AHD(postprocess_stages=3), then WarpRectilinear; crops vs the oracle."""
    import torch
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import rggb_frame
    pipe = DevicePipeline(0)
    wb, M = _wbM(orc)
    H, W = 8736, 11648
    bay = rggb_frame(H, W, 1005)
    rgb = pipe.demosaic(torch.from_numpy(bay).cuda(), wb, M, stages=3)
    pipe.sync()
    for (y0, x0, h, w) in ((0, 0, 240, 300), (4000, 6002, 256, 300), (H - 240, W - 300, 240, 300)):
        ref = orc.demosaic_ahd(np.ascontiguousarray(bay[y0:y0 + h, x0:x0 + w]), wb, M, False, 3)
        ys = slice(0 if y0 == 0 else 24, h if y0 + h == H else h - 24)      # 7 + 4*3 rows of dependency, rounded up
        xs = slice(0 if x0 == 0 else 24, w if x0 + w == W else w - 24)
        assert np.array_equal(rgb[y0:y0 + h, x0:x0 + w].cpu().numpy()[ys, xs], ref[ys, xs]), (y0, x0)
    # identity warp reproduces the image, a real one keeps values finite and is almost the identity at the centre
    ident = np.array([[1.0, 0, 0, 0, 0, 0]] * 3)
    same = pipe.warp(rgb, ident, (0.5, 0.5))
    pipe.sync()
    assert float((same != rgb).float().mean()) < 1e-3
    coeffs = np.array([[1.0, 0.01, 0.002, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0, 0.0, 0.0], [1.0, -0.01, 0.002, 0.0, 0.0, 0.0]])
    warped = pipe.warp(rgb, coeffs, (0.5, 0.5))
    pipe.sync()
    assert bool(torch.isfinite(warped).all())
    cy, cx = H // 2, W // 2
    assert float((warped[cy - 8:cy + 8, cx - 8:cx + 8] - rgb[cy - 8:cy + 8, cx - 8:cx + 8]).abs().max()) < 0.05


# ---- the step before the path ---------------------------------------------------------------------------
def test_hot_pixel_threshold_and_flat_field(orc):
    from pysp_amd.image import RawBayerData
    from pysp_amd.raw_bad_pixel_corr import find_erroneous_pixels_threshold, find_shared_pixels
    from pysp_amd.raw_correction import flat_frame_correction
    from pysp_amd.synth import rggb_frame
    d, _ = load_golden("g11_cleanup")

    def raw(a):
        r = RawBayerData(); r.sensor_scaled = a
        return r
    masks = find_erroneous_pixels_threshold(raw(d["hot"]))
    assert all(m.dtype == np.bool_ and np.array_equal(m, d[f"mask{i}"]) for i, m in enumerate(masks))
    masks_b = find_erroneous_pixels_threshold(raw(d["hot"]), min_delta=0.01, min_neighbour_count=3)
    assert all(np.array_equal(m, d[f"maskb{i}"]) for i, m in enumerate(masks_b))
    shared = find_shared_pixels([masks, masks_b, masks], min_ratio=0.6)
    assert all(np.array_equal(s, a & b) for s, a, b in zip(shared, masks, masks_b))      # 2 of 3 needed
    for key, clamp in (("corrected", False), ("corrected_clamped", True)):
        img = raw(d["bayer"].copy())
        flat_frame_correction(img, raw(d["flat"]), clamp_high=clamp)
        assert np.array_equal(img.sensor_scaled, d[key], equal_nan=True)
    img = raw(d["bayer"].copy())
    flat_frame_correction(img, raw(np.zeros_like(d["flat"])))
    assert np.array_equal(img.sensor_scaled, d["corrected_zero_flat"], equal_nan=True)
    # larger, against the oracle
    bay = rggb_frame(302, 514, 12)
    bay[::37, ::41] = np.minimum(bay[::37, ::41] + 0.5, 1.0)
    got = find_erroneous_pixels_threshold(raw(bay))
    ref = orc.find_hot_threshold(bay)
    assert all(np.array_equal(g, r) for g, r in zip(got, ref)) and sum(int(g.sum()) for g in got) > 10
    yy, xx = np.mgrid[0:302, 0:514]
    flat = (1.0 - 0.6 * ((yy - 151) ** 2 + (xx - 257) ** 2) / (151 ** 2 + 257 ** 2)).astype(np.float32)
    flat[5, 5] = 0.0
    img = raw(bay.copy())
    flat_frame_correction(img, raw(flat))
    assert np.array_equal(img.sensor_scaled, orc.flat_field(bay, flat), equal_nan=True)


def test_demosaic_extreme_values_vs_oracle(orc, wbobj):
    """Out-of-range (negative, > 1), all-zero, all-one and denormal mosaics: same bits as the oracle (no NaNs involved)."""
    from pysp_amd.const import QualityDemosaic
    wb, M = _wbM(orc)
    rng = np.random.default_rng(99)
    H, W = 90, 118
    cases = {
        "signed": (rng.random((H, W), dtype=np.float32) * 1.7 - 0.2).astype(np.float32),
        "zeros": np.zeros((H, W), np.float32),
        "ones": np.ones((H, W), np.float32),
        "denormal": (rng.random((H, W), dtype=np.float32) * np.float32(1e-39)).astype(np.float32),
        "steps": np.kron(rng.integers(0, 2, (H // 6, W // 2)).astype(np.float32), np.ones((6, 2), np.float32))[:H, :W].copy(),
    }
    for name, bay in cases.items():
        bay = np.ascontiguousarray(bay[:H - H % 2, :W - W % 2])
        assert np.array_equal(_raw(bay, wbobj).demosaic(QualityDemosaic.Draft).image, orc.demosaic_draft(bay, wb)), name
        assert np.array_equal(_raw(bay, wbobj).demosaic(QualityDemosaic.Fast).image, orc.demosaic_eag(bay, wb)), name
        for hdr in (False, True):
            got = _raw(bay, wbobj, hdr=hdr).demosaic(QualityDemosaic.Best, 1).image
            ref = orc.demosaic_ahd(bay, wb, M, hdr, 1)
            assert np.array_equal(got, ref, equal_nan=True), (name, hdr)


def test_standalone_eag_helpers(orc):
    """Public helpers of edge_assisted_gaussian.py / gaussian.py outside the fused kernel."""
    from pysp_amd.debayer.edge_assisted_gaussian import resample_b, resample_channel, resample_g_to_full_resolution, resample_r, resample_rb
    from pysp_amd.debayer.gaussian import BayerPatternPosition, CV2_DEFAULT_UNNORM_GAUSSIAN_KERNEL, get_rgbg_kernel
    d, _ = load_golden("g8_resample")
    assert np.array_equal(resample_channel(d["sub"], d["g_sub"], d["g_hf"], BayerPatternPosition.TOP_LEFT), d["out_tl"])
    assert np.array_equal(resample_channel(d["sub"], d["g_sub"], d["g_hf"], BayerPatternPosition.BOTTOM_RIGHT), d["out_br"])
    assert np.array_equal(resample_g_to_full_resolution(d["sub"], d["g_sub"]), d["g_full"])
    k, _ = load_golden("g2_rgbg_kernel")
    for pos in BayerPatternPosition:
        for i, kern in enumerate(get_rgbg_kernel(CV2_DEFAULT_UNNORM_GAUSSIAN_KERNEL, pos)):
            assert np.array_equal(kern, k[f"pos{pos.value}_k{i}"])
    # resample_r / resample_b / resample_rb against the oracle's pieces, odd plane sizes, tiny planes
    rng = np.random.default_rng(17)
    for (h, w) in ((1, 1), (2, 3), (37, 53)):
        r = rng.random((h, w), dtype=np.float32); b = rng.random((h, w), dtype=np.float32)
        g1 = rng.random((h, w), dtype=np.float32); g2 = rng.random((h, w), dtype=np.float32)
        g_up = resample_g_to_full_resolution(g1, g2)
        assert np.array_equal(g_up, orc.resample_g_full(g1, g2))
        hf = g_up - orc.gaussian_blur3(g_up)
        planes = orc.bayer_to_rgbg(g_up)
        assert np.array_equal(resample_r(r, g_up), orc.resample_channel(r, planes[0], hf, 0))
        assert np.array_equal(resample_b(b, g_up), orc.resample_channel(b, planes[2], hf, 3))
        rr, bb = resample_rb(r, b, g_up)
        assert np.array_equal(rr, resample_r(r, g_up)) and np.array_equal(bb, resample_b(b, g_up))
    plain = resample_g_to_full_resolution(d["sub"], d["g_sub"], use_bilinear_weighting=False)
    assert np.array_equal(plain[0::2, 1::2], d["sub"]) and np.array_equal(plain[1::2, 0::2], d["g_sub"])
    with pytest.raises(NotImplementedError):
        resample_channel(d["sub"], d["g_sub"], d["g_hf"], BayerPatternPosition.TOP_RIGHT)


def test_fuse_raw_sixteen_exposures(orc, wbobj):
    """One pass worth of exposures (16) through the host entry point -- and, since round 4, more: the reference's loops take any number of exposures, the
    library runs them as passes of 16 with the partial sums carried between passes (same additions in the same order: same bits)."""
    from pysp_amd.raw_hdr import fuse_exposures_to_raw
    rng = np.random.default_rng(4)
    H, W = 20, 36
    for K in (16, 17, 37):
        frames = [rng.random((H, W), dtype=np.float32) for _ in range(K)]
        frames[K // 2][3:9, 5:11] = 0.0                       # some pixels carry no weight in one exposure
        for f in frames:
            f[12:14, 20:24] = 1.0                             # ... and some in none: the fallback to the largest-offset exposure (raw_hdr.py:144-148)
        evs = [8.0 + 0.25 * ((k * 7) % K) for k in range(K)]      # the largest offset sits in the middle of the list, not in the last pass
        hdr, cnt = fuse_exposures_to_raw([_raw(f, wbobj, ev=e) for f, e in zip(frames, evs)])
        ref, refc, _, _ = orc.fuse_raw(frames, evs, 1.0 / MULT)
        assert np.array_equal(hdr.sensor_scaled, ref) and np.array_equal(cnt, refc), K


def test_warp_with_prior_and_generic_remap(orc):
    """Seeded tables (prior mapping) through apply_opcode_3_warp, and the restated cv2.remap entry point itself."""
    import ctypes
    from pysp_amd import _lib
    from pysp_amd.dng_warp_corr import apply_opcode_3_warp, stack_warp_prior
    d, _ = load_golden("g10_warp_prior")
    img = d["image"].copy()
    apply_opcode_3_warp(img, d["blob"].tobytes(), prior=d["prior"])
    from oracle.checks import warp_phase_check
    cf, centre = _opcode_coeffs(d["blob"].tobytes())
    st = warp_phase_check(img, d["image"], cf, centre, prior=d["prior"], expected=d["warped"])      # classified like the unseeded path (test_warp_apply)
    assert st["differing_outside_boundary_set"] == 0 and st["differing_not_a_neighbouring_phase"] == 0
    ident = stack_warp_prior(d["image"], None, None, None)
    assert ident.shape == d["image"].shape + (2,) and np.array_equal(ident[..., 0, 0][0], np.arange(d["image"].shape[1], dtype=np.float32))
    rng = np.random.default_rng(6)
    src = rng.random((61, 83), dtype=np.float32)
    yy, xx = np.mgrid[0:61, 0:83].astype(np.float32)
    mx = (xx + rng.normal(0, 2.0, xx.shape)).astype(np.float32); my = (yy + rng.normal(0, 2.0, yy.shape)).astype(np.float32)
    mx = np.clip(mx, 0, 82).astype(np.float32); my = np.clip(my, 0, 60).astype(np.float32)
    out = np.empty_like(src)
    _lib.check(_lib.lib().pysp_remap_lanczos4_f32(_lib.default_context().handle, _lib.ptr(src), 61, 83, _lib.ptr(mx), _lib.ptr(my), _lib.ptr(out)))
    assert np.array_equal(out, orc.remap_lanczos4(src, mx, my))          # explicit maps: bit-exact


def test_warp_fused_kernel_equals_its_tables_through_the_oracle_remap(orc):
    """apply_opcode_3_warp evaluates the WarpRectilinear polynomial inside the remap kernel (k_warp_remap: per-channel LDS rectangle, 64 taps from
    LDS, or from global memory when the footprint leaves the rectangle).  Claim: bit-identical to the materialised form of
    chan_distortion_corr.py:86-97 -- the table of compute_remapping_table (G7: <= 2 ULP from the reference's), np.clip, cv2.remap LANCZOS4 as the
    oracle restates it -- for mild and extreme coefficients, sizes on both sides of the 64x16 block."""
    import struct
    from pysp_amd.dng_warp_corr import apply_opcode_3_warp
    from pysp_amd.dng_warp_corr.dng_warp_rectilinear_coords import compute_remapping_table
    for seed, (H, W), mag in ((0, (8, 9), 1e-3), (1, (61, 200), 3e-2), (2, (233, 130), 2e-1), (3, (16, 64), 1e-4), (4, (301, 402), 8e-3), (5, (97, 515), 0.6)):
        rng = np.random.default_rng(900 + seed)
        img = rng.random((H, W, 3), dtype=np.float32)
        coeffs = np.array([[1.0 + rng.normal(0, mag), rng.normal(0, mag), rng.normal(0, mag / 3), rng.normal(0, mag / 10), rng.normal(0, mag / 10), rng.normal(0, mag / 10)]
                           for _ in range(3)])
        cx, cy = float(rng.uniform(0.3, 0.7)), float(rng.uniform(0.3, 0.7))
        payload = struct.pack(">I", 3) + b"".join(struct.pack(">6d", *c) for c in coeffs) + struct.pack(">2d", cx, cy)
        got = img.copy()
        apply_opcode_3_warp(got, struct.pack(">I", 1) + struct.pack(">IIII", 1, 1, 0, len(payload)) + payload)
        for c in range(3):
            t = compute_remapping_table(*coeffs[c], W, H, cx, cy, 1.0)
            ref = orc.remap_lanczos4(np.ascontiguousarray(img[..., c]), np.clip(t[..., 0], 0, W - 1), np.clip(t[..., 1], 0, H - 1))
            assert np.array_equal(got[..., c], ref), (seed, c)


def test_wb_undo_apply_and_clean_xyz(orc, wbobj):
    """image_base.py:45-60 and transform.py:55-74 against the NumPy expressions they stand for."""
    from pysp_amd.base_types.image_base import RawDemosaicData
    from pysp_amd.colorize.rgb_space import LinRgbColorspace
    from pysp_amd.colorize.transform import cam_to_clean_xyz, clip_rgb
    rng = np.random.default_rng(13)
    img = (rng.random((45, 67, 3), dtype=np.float32) * 1.3 - 0.1).astype(np.float32)
    coeff = (1.0 / MULT).astype(np.float32)
    d = RawDemosaicData(img.copy(), coeff)
    d.mat_xyz = wbobj.get_matrix(); d.current_ev = 10.0
    d.wb_undo()
    undone = (img.astype(np.float64) / coeff[:3]).astype(np.float32)
    assert np.array_equal(d.image, undone) and not d._wb_applied
    d.wb_undo()                                           # second undo is a no-op
    assert np.array_equal(d.image, undone)
    d.wb_apply()
    assert np.array_equal(d.image, (undone * coeff[:3]).astype(np.float32)) and d._wb_applied
    assert np.array_equal(clip_rgb(img), np.clip(img, 0, 1))
    # cam_to_clean_xyz = detinted REC2020 working RGB, then that space's RGB->XYZ (both float64 dots)
    from pysp_amd.colorize.transform import final_matrix
    M2020 = final_matrix(wbobj.get_matrix(), LinRgbColorspace.REC2020)
    work = orc.cam_to_rgb(img, M2020, True)
    ref = orc.cam_to_rgb(work, LinRgbColorspace.REC2020.mat_to_xyz(), False)
    assert np.array_equal(cam_to_clean_xyz(img, wbobj.get_matrix()), ref)


def test_warp_band_rows_and_source_row_bound(orc):
    """Row-limited warp (one band of a frame sharded over GPUs): equals the whole-frame warp on its rows and reads
    only the rows pysp_warp_source_rows names -- everything else in the source buffer is NaN here."""
    import torch
    from pysp_amd.dng_warp_corr.dng_warp_rectilinear_coords import compute_remapping_table
    from pysp_amd.pipeline import DevicePipeline
    pipe = DevicePipeline(0)
    H, W = 300, 420
    rng = np.random.default_rng(77)
    rgb = torch.from_numpy(rng.random((H, W, 3), dtype=np.float32)).cuda()
    coeffs = np.array([[1.0, 0.06, 0.01, 0.0, 0.001, 0.0], [1.0, 0.0, 0.0, 0.0, 0.0, 0.0], [0.98, -0.05, 0.01, 0.0, 0.0, 0.002]])
    centre = (0.5, 0.48)
    whole = pipe.warp(rgb, coeffs, centre)
    pipe.sync()
    whole = whole.cpu().numpy()
    for (y0, y1) in [(0, 100), (100, 196), (196, 300), (37, 38)]:
        s0, s1 = pipe.warp_source_rows(H, W, coeffs, centre, y0, y1)
        # the same bound from the product's own coordinate table (compute_remapping_table runs the same arithmetic)
        first = []
        for c in range(3):
            tab = compute_remapping_table(*[float(v) for v in coeffs[c]], W, H, centre[0], centre[1], 1.0)
            my = np.clip(tab[y0:y1, :, 1], 0, H - 1)
            first.append((np.rint(my * np.float32(32)).astype(np.int64) >> 5) - 3)
        first = np.stack(first)
        assert (s0, s1) == (max(0, int(first.min())), min(H - 1, int(first.max()) + 7) + 1)
        assert s0 <= y1 and s1 >= y0
        poisoned = torch.full_like(rgb, float("nan"))
        poisoned[s0:s1] = rgb[s0:s1]
        out = torch.full_like(rgb, -1.0)
        pipe.warp_rows(poisoned, coeffs, centre, y0, y1, out)
        pipe.sync()
        o = out.cpu().numpy()
        assert np.array_equal(o[y0:y1], whole[y0:y1])
        assert (o[:y0] == -1).all() and (o[y1:] == -1).all()         # rows outside the band are not written
    with pytest.raises(ValueError):
        pipe.warp_rows(rgb, coeffs, centre, 10, 10, torch.empty_like(rgb))
    with pytest.raises(ValueError):
        pipe.warp_source_rows(H, W, coeffs, centre, -2, 10)


def _banded_gpu_worker(rank, world, port, q, exchange):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from pysp_amd import multi_gpu
        from pysp_amd.colorize.transform import final_matrix
        from pysp_amd.pipeline import DevicePipeline
        from pysp_amd.synth import default_wb, rggb_frame
        H, W, stages = 400, 512, 3
        bayer = rggb_frame(H, W, 4321)
        wbobj = default_wb()
        wb, M = wbobj.get_reciprocal_multipliers(), final_matrix(wbobj.get_matrix())
        coeffs = np.array([[1.0, 0.03, 0.002, 0, 0, 0], [1.0, 0, 0, 0, 0, 0], [1.0, -0.03, 0.002, 0, 0.001, 0]])
        pipe = DevicePipeline(0)                                           # both ranks rehearse on the one GPU of the box
        y0, y1, band = multi_gpu.demosaic_warp_banded(pipe, bayer, wb, M, coeffs, (0.5, 0.5), stages=stages, rank=rank, world=world,
                                                      exchange=exchange, via_host=True)
        ref = pipe.demosaic_warp(torch.from_numpy(bayer).cuda(), wb, M, coeffs, (0.5, 0.5), stages=stages)
        pipe.sync()
        q.put((rank, y0, y1, bool(torch.equal(band, ref[y0:y1]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["needed", "allgather"])
def test_config5_banded_two_ranks_one_gpu(exchange):
    """BASELINE config 5 sharded over ranks: AHD(3) per band from host rows with halo, exchange of the rows the warp
    needs (gloo + host staging here; RCCL send/recv on a multi-GPU node), row-limited warp == whole-frame result."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_banded_gpu_worker, args=(r, world, port, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=240) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    assert res == [(0, 0, 200, True), (1, 200, 400, True)]


def test_ca_removal_golden_and_oracle(orc, wbobj):
    """corr_ca/ca_removal.py:48-131: fixture from the reference's remove_ca_from_raw, then larger frames vs the oracle."""
    from pysp_amd.corr_ca import remove_ca_from_raw
    from pysp_amd.corr_ca.model.poly3 import Poly3CorrectionModel
    from pysp_amd.corr_ca.model.poly5 import Poly5CorrectionModel
    from pysp_amd.corr_ca.model.ptlens import PtLensCorrectionModel
    from pysp_amd.synth import rggb_frame
    d, meta = load_golden("g12_ca_removal")
    mk = {"poly5_pyfloat": lambda c: Poly5CorrectionModel(*c), "poly5_f64": lambda c: Poly5CorrectionModel(*[np.float64(v) for v in c]),
          "poly3_pyfloat": lambda c: Poly3CorrectionModel(*c), "ptlens_f64": lambda c: PtLensCorrectionModel(*[np.float64(v) for v in c])}
    for cname, (kr, kb) in meta["cases"].items():
        raw = _raw(d["bayer"].copy(), wbobj)
        remove_ca_from_raw(raw, mk[kr](meta["models"][kr]) if kr else None, mk[kb](meta["models"][kb]) if kb else None)
        assert raw.sensor_scaled.dtype == np.float32 and np.array_equal(raw.sensor_scaled, d["out_" + cname]), cname
    wb = wbobj.get_reciprocal_multipliers()
    for (H, W, seed) in [(2, 2, 1), (6, 4, 2), (130, 66, 3), (600, 900, 4)]:
        bay = rggb_frame(H, W, seed) if H > 8 else np.random.default_rng(seed).random((H, W), dtype=np.float32)
        mr, mb = Poly5CorrectionModel(0.05, -0.01), PtLensCorrectionModel(np.float64(-0.02), np.float64(0.03), np.float64(-0.04))
        raw = _raw(bay.copy(), wbobj)
        remove_ca_from_raw(raw, mr, mb)
        ref = orc.remove_ca(bay, mr.get_undistorted_quadrant(bay), mr.get_distorted_quadrant(bay), float(wb[0]),
                            mb.get_undistorted_quadrant(bay), mb.get_distorted_quadrant(bay), float(wb[2]))
        assert np.array_equal(raw.sensor_scaled, ref), (H, W)
        assert np.array_equal(raw.sensor_scaled[0::2, 1::2], bay[0::2, 1::2])            # green untouched
        if H > 100:
            assert np.mean(raw.sensor_scaled[0::2, 0::2] != bay[0::2, 0::2]) > 0.5
    # a channel without a model keeps its samples; an identity model still goes through the resampling filters
    bay = rggb_frame(64, 96, 9)
    raw = _raw(bay.copy(), wbobj)
    ident = Poly5CorrectionModel()
    remove_ca_from_raw(raw, ident, None)
    assert np.array_equal(raw.sensor_scaled[1::2, 1::2], bay[1::2, 1::2])
    assert np.array_equal(raw.sensor_scaled, orc.remove_ca(bay, ident.get_undistorted_quadrant(bay), ident.get_distorted_quadrant(bay), float(wb[0])))


def test_ca_removal_device_resident_batch(orc, wbobj):
    """pysp_remove_ca_dev: lens fields uploaded once, several frames corrected in place on the device, then demosaiced."""
    import torch
    from pysp_amd.corr_ca.model.poly5 import Poly5CorrectionModel
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import rggb_frame
    pipe = DevicePipeline(0)
    wb, M = _wbM(orc)
    H, W = 256, 384
    mr, mb = Poly5CorrectionModel(0.03, -0.008), Poly5CorrectionModel(np.float64(-0.025), np.float64(0.012))
    fr, fb = pipe.lens_fields(mr, (H, W)), pipe.lens_fields(mb, (H, W))
    for seed in (11, 12, 13):
        bay = rggb_frame(H, W, seed)
        d = torch.from_numpy(bay).cuda()
        pipe.remove_ca(d, wb, fr, fb)
        rgb = pipe.demosaic(d, wb, M, stages=1)
        pipe.sync()
        ref = orc.remove_ca(bay, fr[0].cpu().numpy(), fr[1].cpu().numpy(), float(wb[0]), fb[0].cpu().numpy(), fb[1].cpu().numpy(), float(wb[2]))
        assert np.array_equal(d.cpu().numpy(), ref), seed
        assert np.array_equal(rgb.cpu().numpy(), orc.demosaic_ahd(ref, wb, M, False, 1))
    only_b = torch.from_numpy(rggb_frame(H, W, 14)).cuda()
    keep = only_b.clone()
    pipe.remove_ca(only_b, wb, None, fb)
    pipe.sync()
    assert torch.equal(only_b[0::2, 0::2], keep[0::2, 0::2]) and not torch.equal(only_b[1::2, 1::2], keep[1::2, 1::2])
    with pytest.raises(ValueError):
        pipe.remove_ca(only_b, wb, (fr[0][:10], fr[1][:10]), None)


def test_contexts_are_reentrant_across_threads(orc, wbobj):
    """SURVEY 8b threading: entry points are re-entrant per pysp_ctx (own stream + workspace); ctypes releases the GIL."""
    import threading
    from pysp_amd import _lib
    from pysp_amd.synth import rggb_frame
    wb, M = _wbM(orc)
    H, W = 192, 256
    frames = [rggb_frame(H, W, 50 + i) for i in range(4)]
    refs = [orc.pipeline_srgb(f, wb, M, 2, False, 1, False) for f in frames]
    outs = [[None] * 6 for _ in frames]
    errors = []

    def work(i):
        try:
            ctx = _lib.Context(0)
            wb3, m9 = _lib.wb3(wb), _lib.mat9(M)
            for rep in range(6):
                out = np.empty((H, W, 3), np.float32)
                _lib.check(_lib.lib().pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(frames[i]), H, W, wb3, m9, 2, 0, 1, 0, _lib.ptr(out)))
                outs[i][rep] = out
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(frames))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for i in range(len(frames)):
        for rep in range(6):
            assert np.array_equal(outs[i][rep], refs[i]), (i, rep)
    # the error string is per thread: a failure elsewhere does not leak into this thread's message
    rc = _lib.lib().pysp_demosaic_f32(_lib.default_context().handle, _lib.ptr(frames[0]), 3, 5, _lib.wb3(wb), _lib.mat9(M), 2, 0, 1, _lib.ptr(outs[0][0]))
    assert rc != 0 and b"even" in _lib.lib().pysp_last_error()


def test_c_client_runs_on_gpu(tmp_path):
    """tests/c_abi_check.c (plain C99, no Python, no torch) against the library on the GPU: context, one demosaic, error codes."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_check")
    lib_dir = os.path.join(root, "pysp_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", os.path.join(root, "tests", "c_abi_check.c"), "-o", exe,
                           "-L" + lib_dir, "-lpysp_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("ok (GPU"), out.stdout + out.stderr
    assert "0.5 0.25 0.357143" in out.stdout          # Draft of a flat 0.25 mosaic times wb = (2, 1, 1/0.7)


def test_signed_zero_bits_match_oracle(orc, wbobj):
    """-0.0 samples: the filters accumulate from +0.0f in oracle and kernels alike, so even the sign bit of a zero result agrees."""
    from pysp_amd.const import QualityDemosaic
    wb, M = _wbM(orc)
    bay = np.full((8, 12), -0.0, np.float32)
    bay[2:4, 4:8] = 0.25
    for q, ref in ((QualityDemosaic.Fast, orc.demosaic_eag(bay, wb)), (QualityDemosaic.Best, orc.demosaic_ahd(bay, wb, M, False, 1))):
        got = _raw(bay.copy(), wbobj).demosaic(q).image
        assert got.tobytes() == ref.tobytes(), q


def test_fuzz_slice():
    """Ten seconds of tests/fuzz_parity.py (random sizes, value distributions, qualities, tails, uint16, CA, fusion) inside the suite."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "fuzz_parity.py"), "10", "20260101"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "bit-exact" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
