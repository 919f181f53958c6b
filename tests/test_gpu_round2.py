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


# ---- non-finite values travel through the path exactly as through the NumPy reference ----------------
def _poisoned(H, W, seed, scale=1.0):
    """Synthetic frame with quiet NaN, +Inf and -Inf sites: corners, edges, a 2x2 block, random interior sites."""
    from pysp_amd.synth import rggb_frame
    rng = np.random.default_rng(seed)
    bay = rggb_frame(H, W, seed, scale=scale, clip_hi=scale == 1.0).copy()
    vals = [np.nan, np.inf, -np.inf]
    sites = [(0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (0, W // 2), (H // 2, 0), (H - 1, W // 3), (H // 3, W - 1)]
    sites += [(int(rng.integers(2, H - 2)), int(rng.integers(2, W - 2))) for _ in range(18)]
    for i, (y, x) in enumerate(sites):
        bay[y, x] = vals[i % 3]
    y0, x0 = H // 2 + 1, W // 2 + 3
    bay[y0:y0 + 2, x0:x0 + 2] = np.nan
    return bay


def _same(got, ref):
    """Bit-equal where finite, same NaN positions, same signed infinities."""
    return np.array_equal(got, ref, equal_nan=True)


def test_non_finite_mosaic_matches_literal_oracle(orc, wbobj):
    """NaN (e.g. the 0/0 of raw_correction.py:45) and +-Inf in the mosaic: np.clip keeps NaN (transform.py:6-19), the float64
    np.dot spreads it over the pixel, build_map counts nothing through a NaN compare (pyx:56-57), the selection multiplies
    the unselected candidate by 0 (ahd.py:141-145: Inf * 0 = NaN).  GPU == literal oracle, NaN positions included, for
    Draft / EAG / AHD(0) with and without the HDR metric and every colour tail; with a median stage everywhere except
    within 4 px of a non-finite pre-median pixel (a median of a window holding NaN is unspecified in OpenCV, too)."""
    import torch
    from pysp_amd import _lib
    from pysp_amd.pipeline import DevicePipeline
    wb, M = _wbM(orc)
    pipe = DevicePipeline(0)
    for (H, W), seed in (((64, 96), 5), ((130, 158), 6)):
        for hdr in (False, True):
            bay = _poisoned(H, W, seed, scale=3.0 if hdr else 1.0)
            d = torch.from_numpy(bay).cuda()
            if not hdr:
                for q, fn in ((_lib.QUALITY_DRAFT, orc.demosaic_draft), (_lib.QUALITY_FAST, orc.demosaic_eag)):
                    raw = fn(bay, wb)
                    assert np.isnan(raw).any() and np.isinf(raw).any()
                    assert _same(pipe.demosaic(d, wb, M, q, False, 0).cpu().numpy(), raw), q
                    assert _same(pipe.batch([d], wb, M, q, False, 0, tail=1)[0].cpu().numpy(), orc.cam_to_rgb(raw, M, True)), q
                    assert _same(pipe.demosaic_to_srgb(d, wb, M, q, False, 0).cpu().numpy(), orc.pipeline_srgb(bay, wb, M, q, False, 0, False)), q
            raw = orc.demosaic_ahd(bay, wb, M, hdr, 0)
            assert np.isnan(raw).any()
            assert _same(pipe.demosaic(d, wb, M, _lib.QUALITY_BEST, hdr, 0).cpu().numpy(), raw), hdr
            assert _same(pipe.batch([d], wb, M, _lib.QUALITY_BEST, hdr, 0, tail=1)[0].cpu().numpy(), orc.cam_to_rgb(raw, M, True)), hdr
            for rh in (False, True):
                assert _same(pipe.demosaic_to_srgb(d, wb, M, _lib.QUALITY_BEST, hdr, 0, rh).cpu().numpy(),
                             orc.pipeline_srgb(bay, wb, M, 2, hdr, 0, rh)), (hdr, rh)
            # one median stage: exact outside the 9x9 neighbourhoods (two chained 5x5 medians) of non-finite pre-median pixels
            bad = ~np.isfinite(raw).all(axis=-1)
            near = np.zeros_like(bad)
            for dy in range(-4, 5):
                for dx in range(-4, 5):
                    near |= np.roll(np.roll(np.pad(bad, 4), dy, 0), dx, 1)[4:-4, 4:-4]
            got = pipe.demosaic(d, wb, M, _lib.QUALITY_BEST, hdr, 1).cpu().numpy()
            ref = orc.demosaic_ahd(bay, wb, M, hdr, 1)
            assert (~near).sum() > 1500 and np.array_equal(got[~near], ref[~near]), hdr


def test_hdr_votes_fast_and_literal_workgroups_in_one_launch(orc, wbobj):
    """Round 3: with the HDR metric a workgroup votes in the literal nine-cell form only if it wrote a non-finite luma into its Lab buffer (k_ahd.hip,
    s_nonfinite); every other workgroup takes the fast form.  A frame of 15 x 21 tiles whose few NaN / Inf sites sit in tile interiors, on tile seams
    and in tile corners (so that a neighbour sees them only in its halo) against the oracle's literal vote everywhere, bit for bit; and the same frame clean."""
    import torch
    from pysp_amd import _lib
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import rggb_frame
    wb, M = _wbM(orc)
    pipe = DevicePipeline(0)
    H, W = 420, 588
    clean = rggb_frame(H, W, 4242, scale=3.0, clip_hi=False)
    for sites in ((), ((200, 300, np.nan),), ((27, 27, np.inf), (28, 28, np.nan), (55, 83, -np.inf), (56, 84, np.inf), (139, 0, np.nan), (419, 587, np.inf), (251, 336, np.nan), (252, 335, np.inf))):
        bay = clean.copy()
        for y, x, v in sites:
            bay[y, x] = v
        d = torch.from_numpy(bay).cuda()
        ref = orc.demosaic_ahd(bay, wb, M, True, 0)
        assert np.isnan(ref).any() == bool(sites)
        assert _same(pipe.demosaic(d, wb, M, _lib.QUALITY_BEST, True, 0).cpu().numpy(), ref), len(sites)
        assert _same(pipe.demosaic_to_srgb(d, wb, M, _lib.QUALITY_BEST, True, 0, True).cpu().numpy(), orc.pipeline_srgb(bay, wb, M, 2, True, 0, True)), len(sites)


def test_non_finite_from_the_products_own_flat_field(orc, wbobj):
    """raw_correction.py:45 on a region where image and flat are both zero leaves NaN in the mosaic; that mosaic then goes
    through the README recipe (drop-in classes), equal to the oracle including the NaN positions."""
    from pysp_amd.base_types.image_base import RawBayerData_BaseType  # noqa: F401  (import path check)
    from pysp_amd.colorize import lin_srgb_to_srgb
    from pysp_amd.const import QualityDemosaic
    from pysp_amd.image import RawBayerData, RawRggbBayerData
    from pysp_amd.raw_correction import flat_frame_correction
    from pysp_amd.synth import rggb_frame
    wb, M = _wbM(orc)
    H, W = 72, 104
    bay = rggb_frame(H, W, 21).copy()
    flat = np.full((H, W), 0.8, np.float32)
    bay[20:26, 30:38] = 0.0
    flat[20:26, 30:38] = 0.0                      # 0 * mean / 0 = NaN
    flat[50, 60] = 0.0                            # x / 0 = +Inf -> replaced by the channel maximum (raw_correction.py:52)
    img = RawBayerData(); img.sensor_scaled = bay.copy()
    fl = RawBayerData(); fl.sensor_scaled = flat
    flat_frame_correction(img, fl)
    mos = img.sensor_scaled
    assert np.isnan(mos[20:26, 30:38]).all() and np.isfinite(mos[50, 60])
    for q, stages in ((QualityDemosaic.Draft, 0), (QualityDemosaic.Fast, 0), (QualityDemosaic.Best, 0)):
        dem = RawRggbBayerData(mos, wbobj, 10.0, 1.0).demosaic(q, stages)
        qi = {QualityDemosaic.Draft: 0, QualityDemosaic.Fast: 1, QualityDemosaic.Best: 2}[q]
        ref_raw = [orc.demosaic_draft, orc.demosaic_eag, lambda b, w: orc.demosaic_ahd(b, w, M, False, 0)][qi](mos, wb)
        assert _same(dem.image, ref_raw), q
        srgb = lin_srgb_to_srgb(dem.to_lin_srgb())
        assert _same(srgb, orc.pipeline_srgb(mos, wb, M, qi, False, 0, False)), q
        assert np.isnan(srgb).any()


def test_non_finite_pointwise_and_fusion(orc):
    """clip_rgb / cam_to_lin_srgb / the sRGB curves / wb on NaN and Inf (transform.py:6-19,52-53,89-111); raw HDR fusion."""
    import torch
    from pysp_amd.colorize.transform import clip_rgb, lin_srgb_to_srgb, srgb_to_lin_srgb
    from pysp_amd.pipeline import DevicePipeline
    wb, M = _wbM(orc)
    rng = np.random.default_rng(3)
    px = (rng.random((33, 47, 3), dtype=np.float32) * np.float32(1.6) - np.float32(0.3)).astype(np.float32)
    px[0, 0] = [np.nan, 0.5, 0.2]; px[0, 1] = [0.1, np.inf, 0.2]; px[0, 2] = [0.3, 0.4, -np.inf]; px[1, 1] = [np.nan, np.nan, np.nan]; px[2, 2] = [np.inf, -np.inf, np.nan]
    want_clip = np.clip(px, 0, 1)
    assert _same(clip_rgb(px), want_clip) and np.isnan(clip_rgb(px)[0, 0, 0])
    assert _same(lin_srgb_to_srgb(px), orc.lin_srgb_to_srgb(px)) and _same(srgb_to_lin_srgb(px), orc.srgb_to_lin_srgb(px))
    enc = lin_srgb_to_srgb(px)
    assert np.isnan(enc[0, 0, 0]) and enc[0, 1, 1] == orc.lin_srgb_to_srgb(np.ones(3, np.float32))[0] and enc[0, 2, 2] == 0.0   # NaN stays, +Inf clips to 1, -Inf to 0
    from pysp_amd import _lib
    L, ctx = _lib.lib(), _lib.default_context()
    for clip in (1, 0):
        out = np.empty_like(px)
        _lib.check(L.pysp_cam_to_rgb_f32(ctx.handle, _lib.ptr(px), px.size // 3, _lib.mat9(M), clip, _lib.ptr(out)))
        assert _same(out, orc.cam_to_rgb(px, M, bool(clip))), clip
    # raw fusion: a NaN sample poisons its pixel's weight sum (raw_hdr.py:135-148 literally)
    pipe = DevicePipeline(0)
    H, W, K = 48, 64, 3
    frames = [np.clip(rng.random((H, W), dtype=np.float32) * np.float32(2.0 ** -k), 0, 1) for k in range(K)]
    frames[1][5, 7] = np.nan; frames[0][9, 9] = np.inf; frames[2][11, 3] = -np.inf
    evs = [10.0, 11.0, 12.0]
    fused, count, _, _ = pipe.fuse_raw([torch.from_numpy(f).cuda() for f in frames], evs, wb)
    rf, rc = orc.fuse_raw(frames, evs, wb)[:2]
    assert _same(fused.cpu().numpy(), rf) and np.array_equal(count.cpu().numpy(), rc) and np.isnan(rf[5, 7])


# ---- bench.py starts its own ranks; configs 3 and 5 are bench workloads ---------------------
def _bench(*argv, timeout=600):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, cwd=root)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-1500:] + out.stderr[-3000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_cfg5_banded():
    """`python bench.py --gpus 2 --backend gloo --workload cfg5` as typed (no external launcher): two ranks share the box's
    one GPU, 100 MP AHD(3) + warp in two bands with the row exchange, ONE JSON line, n_gpus = the observed world size."""
    line = _bench("--gpus", "2", "--backend", "gloo", "--workload", "cfg5", "--steps", "2", "--warmup", "1", "--settle", "0")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["bands"] == 2 and line["config"]["backend"] == "gloo"
    assert set(line["phases_ms"]) == {"demosaic", "bound_allgather", "row_exchange", "warp"} and line["phases_ms"]["demosaic"] > 0
    assert line["value"] > 0 and line["roofline"]["kernel"] in ("k_ahd_median_stage", "k_ahd_select", "k_warp_remap")
    # round 3: every rank's own time and what its row exchange moved (a fraction of what the all-gather would deliver)
    rk = line["ranks_ms_per_step"]
    assert len(rk["per_rank"]) == 2 and rk["min"] <= rk["median"] <= rk["max"] and rk["max"] <= line["ms_per_step"] * 1.05
    got, would = rk["exchange_bytes_received_per_rank"], rk["allgather_bytes_received_per_rank_would_be"]
    assert all(0 < g < w // 20 for g, w in zip(got, would)) and sum(got) == sum(rk["exchange_bytes_sent_per_rank"])
    assert all(r <= 120 for r in rk["exchange_rows_received_per_rank"])


def test_bench_cfg3_batch_two_ranks_and_single():
    line = _bench("--gpus", "2", "--backend", "gloo", "--workload", "cfg3", "--frames", "2", "--steps", "3", "--warmup", "1", "--settle", "0")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["frames_per_step_total"] == 4
    assert line["roofline"]["kernel"] == "k_eag" and line["cpu_baseline"] is None
    one = _bench("--workload", "cfg3", "--frames", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert one["n_gpus"] == 1 and one["config"]["frames_per_step_total"] == 2 and one["value"] > 0


def test_bench_rccl_code_path_on_one_gpu():
    """The N > 1 code path with the real backend ("nccl" = RCCL): process group on the GPU, parameter broadcast inside the timed region,
    all_reduce of the elapsed time, barriers -- at world size 1 under the launcher the driver uses, because the box has one GPU."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for wl in ("ahd24", "cfg3", "cfg5"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29571",
               os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--workload", wl, "--steps", "4", "--warmup", "1", "--frames", "2", "--no-cpu-baseline", "--settle", "0"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert out.returncode == 0 and len(lines) == 1, out.stdout[-1500:] + out.stderr[-3000:]
        line = json.loads(lines[0])
        assert line["n_gpus"] == 1 and line["config"]["backend"] == "rccl" and line["value"] > 0, line


def test_bench_default_line_contract():
    """The driver's own command shape (N = 1, short run): metric, config, roofline with the VALU bound, cpu_baseline."""
    line = _bench("--gpus", "1", "--steps", "20", "--warmup", "5")
    assert line["metric"].startswith("megapixels/sec AHD debayer+cam->sRGB") and line["unit"] == "MP/s" and line["dtype"] == "f32"
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["vs_baseline"] is None
    r = line["roofline"]
    # round 5: `bound` names what binds -- the VALU issue rate of the two AHD kernels -- next to the contract's HBM figures (frac = hbm_frac); issue_bound carries
    # the class mix of the built library (profiles/isa_mix.json), the executed instruction counts (PMC) and the [lo, hi] bracket of the cost model
    assert r["bound"] == "valu_issue" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["traffic"] and r["hbm_frac"] == r["frac"]
    ib = r["issue_bound"]
    assert ib["kernel"] == r["kernel"] and set(ib["class_counts_static"]) == set("FABT") and ib["class_counts_static"]["B"] > 100
    lo, hi = ib["frac_of_issue_bound"]
    assert 0.5 < lo <= hi < 1.35, ib["frac_of_issue_bound"]             # the kernel runs where its instruction mix says it should (DESIGN.md 7.R4 a)
    slo, shi = ib["step_frac_of_issue_bound"]
    assert 0.5 < slo <= shi < 1.35 and slo > 5 * r["frac"]               # ... an order of magnitude closer to that bound than to the HBM roofline
    assert ib["stale"] in (False, True) and set(ib["all_kernels"]) == {"k_ahd_select", "k_ahd_median_stage"}
    assert line["config"]["streams_per_rank"] == 2 and line["config"]["per_kernel_sampling"].startswith("one stream")
    # round 4: frac is the STEP's fraction (16 B per output pixel over the step time); the dominant kernel's own figure sits under its own key
    assert abs(r["achieved"] - 16 * 24e6 / (line["ms_per_step"] * 1e-3) / 1e9) < 0.5 and r["frac"] == r["pipeline_frac"]
    dk = r["dominant_kernel"]
    assert dk["kernel"] == r["kernel"] and abs(dk["frac"] - dk["achieved"] / 8000.0) < 1e-4 and dk["frac"] > r["frac"] and dk["avg_launch_ms"] < line["ms_per_step"]
    assert r["traffic"] > r["alg_bytes_per_step"]
    assert r["valu"] and 0 < r["valu"]["frac_of_2cycle_issue"] <= 1.0 and r["valu"]["insts_per_px"] > 100
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    assert c["threads_used"] == c["cores"] and c["cores_available"] >= c["cores"] and c["thread_cap"]
    # round 3: the line verifies itself -- the GPU output of the timed whole-frame call against the oracle, bit for bit -- and says whether its PMC figures belong to this build
    v = line["verify"]
    assert v["frames"] >= 1 and v["max_ulp"] == 0 and v["bit_exact"] and v["bit_exact_demosaic"] and v["values_compared"] == v["frames"] * 4000 * 6000 * 3
    assert r["traffic_stale"] in (False, True) and r["valu_insts_per_px"] == r["valu"]["insts_per_px"] and len(r["lib_sha256"]) == 64


# ---- the sRGB curve, every float32 in [0, 1] --------------------------------------------------------
def test_srgb_curve_exhaustive(orc):
    """lin_srgb_to_srgb (transform.py:89-99) on ALL 1,065,353,217 float32 values of [0, 1] (every bit pattern from 0 to
    0x3F800000) plus a band above 1 and the negative side: the kernel's x ** 0.41666666f (devmath.h::srgb_pow_5_12, hardware
    log2/exp2 seed + one float64 Newton-type correction) against the oracle's float64 pow rounded once.  Claim: bit-identical
    on every input (north_star asks for 1 ULP)."""
    import ctypes
    import torch
    from pysp_amd import _lib
    L, ctx = _lib.lib(), _lib.Context(0)
    ctx.set_stream(int(torch.cuda.current_stream().cuda_stream))     # same stream as torch's arange below: ordered, no host sync needed
    chunk = 1 << 26
    hi = 0x3F800000 + 4096                      # a little beyond 1.0: clipped
    worst, differ, n = 0, 0, 0
    out = torch.empty(chunk, dtype=torch.float32, device="cuda")
    for lo in range(0, hi, chunk):
        m = min(chunk, hi - lo)
        x = torch.arange(lo, lo + m, dtype=torch.int32, device="cuda").view(torch.float32)
        _lib.check(L.pysp_lin_srgb_to_srgb_dev(ctx.handle, ctypes.c_void_p(x.data_ptr()), m, ctypes.c_void_p(out.data_ptr())))
        ctx.sync()
        got = out[:m].cpu().numpy()
        ref = orc.lin_srgb_to_srgb(x.cpu().numpy())
        bad = got != ref
        if bad.any():
            differ += int(bad.sum())
            worst = max(worst, int(ulp_diff(got[bad], ref[bad]).max()))
        n += m
    neg = np.array([-0.0, -1e-30, -1.0, -np.inf], np.float32)
    from pysp_amd.colorize.transform import lin_srgb_to_srgb
    assert np.array_equal(lin_srgb_to_srgb(neg.reshape(1, -1, 1).repeat(3, axis=2)), np.zeros((1, 4, 3), np.float32))
    print(f"sRGB curve: {n} inputs, {differ} differ from the oracle, worst {worst} ULP")
    assert n > 1_065_353_216 and differ == 0, (n, differ, worst)


# ---- device-resident drop-in objects; overlapped host pipeline ------------------------------------------
def test_readme_recipe_stays_on_the_gpu_until_read(orc, wbobj):
    """README.md:55-63: demosaic(...).to_lin_srgb() then lin_srgb_to_srgb(...).  The intermediates are lazy DeviceArrays:
    one upload, the kernels, one download; results identical to the eager (ndarray everywhere) mode and to the oracle."""
    import pysp_amd
    from pysp_amd import DeviceArray
    from pysp_amd.colorize import lin_srgb_to_srgb
    from pysp_amd.const import QualityDemosaic
    from pysp_amd.image import RawRggbBayerData
    from pysp_amd.synth import rggb_frame
    wb, M = _wbM(orc)
    bay = rggb_frame(300, 412, 77)
    ref_rgb = orc.demosaic_ahd(bay, wb, M, False, 1)
    ref_srgb = orc.pipeline_srgb(bay, wb, M, 2, False, 1, False)
    assert pysp_amd.lazy_enabled()
    dem = RawRggbBayerData(bay, wbobj, 10.0, 1.0).demosaic(QualityDemosaic.Best, 1)
    assert dem._dev is not None and dem._img is None and dem.is_valid()          # nothing downloaded yet
    lin = dem.to_lin_srgb()
    assert isinstance(lin, DeviceArray) and lin.on_device and lin._host is None and lin.shape == (300, 412, 3) and lin.dtype == np.float32
    assert dem._dev is not None                                                  # to_lin_srgb did not pull the image down either
    srgb = lin_srgb_to_srgb(lin)
    assert isinstance(srgb, np.ndarray) and srgb.dtype == np.float32 and ulp_diff(srgb, ref_srgb).max() == 0
    assert lin._host is None                                                     # still only on the GPU
    # reading .image downloads once; from then on it is an ordinary ndarray attribute the caller may edit in place
    img = dem.image
    assert isinstance(img, np.ndarray) and np.array_equal(img, ref_rgb) and dem._dev is None and dem.image is img
    img[0, 0, 0] = 0.5
    assert np.array_equal(dem.to_lin_srgb()[0, 0], orc.cam_to_rgb(img[:1, :1], M, True)[0, 0])     # the edit is what gets coloured
    # the lazy array quacks like the ndarray of the reference
    assert np.array_equal(np.asarray(lin), orc.cam_to_rgb(ref_rgb, M, True)) and lin._host is not None
    assert np.array_equal(lin / (1 + lin), np.asarray(lin) / (1 + np.asarray(lin))) and lin[3, 4, 1] == np.asarray(lin)[3, 4, 1]
    assert lin.mean() == np.asarray(lin).mean() and lin.astype(np.float64).dtype == np.float64 and len(lin) == 300
    # eager mode: plain ndarrays everywhere, same numbers
    pysp_amd.set_lazy(False)
    try:
        dem2 = RawRggbBayerData(bay, wbobj, 10.0, 1.0).demosaic(QualityDemosaic.Best, 1)
        lin2 = dem2.to_lin_srgb()
        assert isinstance(dem2._img, np.ndarray) and isinstance(lin2, np.ndarray)
        assert np.array_equal(dem2.image, ref_rgb) and np.array_equal(lin_srgb_to_srgb(lin2), srgb)
    finally:
        pysp_amd.set_lazy(True)


def test_lazy_array_edits_copies_and_threads(orc, wbobj):
    """once a host copy of a lazy array has been handed out it IS the array (an edit through it is what later calls see, as with the
    reference's ndarray); item assignment and in-place arithmetic work; copy / deepcopy / pickle give plain ndarrays (a device pointer is never
    duplicated); a Context cannot be copied; an array may be read and dropped on another thread."""
    import copy
    import pickle
    import threading
    from pysp_amd import DeviceArray, _lib
    from pysp_amd.colorize import lin_srgb_to_srgb
    from pysp_amd.const import QualityDemosaic
    from pysp_amd.image import RawRggbBayerData
    from pysp_amd.synth import rggb_frame
    wb, M = _wbM(orc)
    bay = rggb_frame(64, 96, 5)

    def fresh():
        return RawRggbBayerData(bay, wbobj, 10.0, 1.0).demosaic(QualityDemosaic.Fast).to_lin_srgb()
    ref = orc.cam_to_rgb(orc.demosaic_eag(bay, wb), M, True)
    # (1) edit through np.asarray, then a GPU call: the edit is honoured
    lin = fresh()
    assert isinstance(lin, DeviceArray) and lin.on_device
    a = np.asarray(lin)
    assert not lin.on_device                                   # the host copy is the array now
    a *= np.float32(0.5)
    assert np.array_equal(lin_srgb_to_srgb(lin), orc.lin_srgb_to_srgb(ref * np.float32(0.5)))
    # (2) item assignment and in-place arithmetic on the lazy object itself
    lin = fresh()
    lin[ref > 0.5] = 0.25
    lin += np.float32(0.125)
    want = ref.copy(); want[ref > 0.5] = 0.25; want += np.float32(0.125)
    assert isinstance(lin, DeviceArray) and np.array_equal(np.asarray(lin), want) and np.array_equal(lin_srgb_to_srgb(lin), orc.lin_srgb_to_srgb(want))
    # (3) copies and pickles are ndarrays with the same values; the original keeps working
    lin = fresh()
    c1, c2, c3 = copy.copy(lin), copy.deepcopy(lin), pickle.loads(pickle.dumps(lin))
    for c in (c1, c2, c3):
        assert type(c) is np.ndarray and np.array_equal(c, ref)
    c1[0, 0, 0] = 7.0
    assert np.asarray(lin)[0, 0, 0] == ref[0, 0, 0]
    dem = RawRggbBayerData(bay, wbobj, 10.0, 1.0).demosaic(QualityDemosaic.Fast)
    dem2 = copy.deepcopy(dem)                                  # the container of the reference is deep-copyable: so is this one
    assert np.array_equal(dem2.image, dem.image) and dem2.image is not dem.image
    # (3b) ADVICE r3: the ORIGINAL keeps working after a copy / pickle WITHOUT anyone reading .image first -- the copy released its device buffer
    #      (DeviceArray.numpy()), so the container has to have switched to the host copy itself
    from pysp_amd.raw_hdr import fuse_exposures_from_debayer
    for duplicate in (copy.copy, copy.deepcopy, lambda o: pickle.loads(pickle.dumps(o))):
        exps = []
        for k in range(2):
            e = RawRggbBayerData(bay * np.float32(0.5 ** k), wbobj, 10.0 + k, 1.0).demosaic(QualityDemosaic.Fast)
            e.mat_xyz = wbobj.get_matrix()
            exps.append(e)
        dup = duplicate(exps[1])
        assert exps[1]._dev is None and isinstance(exps[1]._img, np.ndarray)       # no dangling released buffer
        exps[1].wb_undo(); exps[1].wb_apply()                                    # used to raise ValueError('device copy ... released')
        fused, count = fuse_exposures_from_debayer(exps)                         # used to fail with AttributeError on exposure 2
        clean = [RawRggbBayerData(bay * np.float32(0.5 ** k), wbobj, 10.0 + k, 1.0).demosaic(QualityDemosaic.Fast) for k in range(2)]
        for c in clean:
            c.mat_xyz = wbobj.get_matrix()
        clean[1].wb_undo(); clean[1].wb_apply()
        fused2, count2 = fuse_exposures_from_debayer(clean)
        assert np.array_equal(np.asarray(fused), np.asarray(fused2), equal_nan=True) and np.array_equal(count, count2)
        assert np.array_equal(dup.image, np.asarray(RawRggbBayerData(bay * np.float32(0.5), wbobj, 11.0, 1.0).demosaic(QualityDemosaic.Fast).image))
    # (3c) two threads materialising ONE lazy array at the same moment get the same ndarray (download + release is one step under the context's lock)
    lin = fresh()
    got, go = [], threading.Barrier(2)

    def racer():
        go.wait()
        got.append(lin.numpy())
    ts = [threading.Thread(target=racer) for _ in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert len(got) == 2 and got[0] is got[1] and np.array_equal(got[0], ref)
    for f in (copy.copy, copy.deepcopy, pickle.dumps):
        with pytest.raises(TypeError):
            f(_lib.default_context())
    # (4) another thread reads and drops a lazy array made here
    lin = fresh()
    box = {}

    def reader(x):
        box["v"] = np.asarray(x).copy()
    t = threading.Thread(target=reader, args=(lin,)); t.start(); t.join()
    assert np.array_equal(box["v"], ref)
    lin = fresh()
    t = threading.Thread(target=lambda holder: holder.clear(), args=([lin],))
    del lin
    t.start(); t.join()
    assert np.array_equal(np.asarray(fresh()), ref)            # the context's buffer cache survived the foreign-thread free


def test_host_pipeline_in_overlapped_bands_equals_whole_frame(orc, wbobj):
    """The host-buffer entry points cut frames of more than 4 MP into 256-row bands (upload / kernels / download overlap):
    same bits as the device-resident whole-frame call, for every quality, the HDR metric, stages 0..3 and uint16 input."""
    import ctypes
    import torch
    from pysp_amd import _lib
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import rggb_frame
    wb, M = _wbM(orc)
    L, ctx = _lib.lib(), _lib.default_context()
    pipe = DevicePipeline(0)
    H, W = 2230, 2054                                  # 4.6 MP: 9 bands, the last one short; W % 4 == 2 exercises the unstaged store path
    bay = rggb_frame(H, W, 31)
    d = torch.from_numpy(bay).cuda()
    out = np.empty((H, W, 3), np.float32)
    for q, hdr, stages, rh in ((0, 0, 0, 0), (1, 0, 0, 0), (2, 0, 0, 0), (2, 0, 1, 0), (2, 1, 1, 1), (2, 0, 3, 0)):
        _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(bay), H, W, _lib.wb3(wb), _lib.mat9(M), q, hdr, stages, rh, _lib.ptr(out)))
        want = pipe.demosaic_to_srgb(d, wb, M, q, bool(hdr), stages, bool(rh)).cpu().numpy()
        assert np.array_equal(out, want), (q, hdr, stages, rh)
    _lib.check(L.pysp_demosaic_f32(ctx.handle, _lib.ptr(bay), H, W, _lib.wb3(wb), _lib.mat9(M), 2, 0, 1, _lib.ptr(out)))
    assert np.array_equal(out, pipe.demosaic(d, wb, M, _lib.QUALITY_BEST, False, 1).cpu().numpy())
    raw = (np.random.default_rng(4).random((H, W)) * 16383).astype(np.uint16)
    black, sat = (ctypes.c_float * 4)(500, 510, 520, 505), (ctypes.c_float * 4)(15000, 15100, 14900, 15050)
    _lib.check(L.pysp_pipeline_u16_f32(ctx.handle, _lib.ptr(raw), H, W, black, sat, _lib.wb3(wb), _lib.mat9(M), 2, 0, 1, 2, _lib.ptr(out)))
    want = pipe.raw_u16_to_rgb(torch.from_numpy(raw.view(np.int16)).cuda(), list(black), list(sat), wb, M, _lib.QUALITY_BEST, 1, 2).cpu().numpy()
    assert np.array_equal(out, want)


# ---- Lab modes: 1 (default) = the OpenCV-4.10 LUT + trilinear restatement, 0 = the closed form of round 1 ----------------
def test_lab_modes(orc, wbobj):
    """Every other test runs in the default mode 1 (fixtures g8_demosaic_* come from the reference's unchanged ahd.py with the
    NumPy cv410_lut restatement as cv2.cvtColor).  Here: pysp_ctx_set_lab_mode(ctx, 0) against the closed-form fixtures and the
    oracle in mode 0 -- larger frames, the HDR metric, non-finite sites -- and the size of the difference between the modes."""
    import torch
    from pysp_amd import _lib
    from pysp_amd.const import QualityDemosaic
    from pysp_amd.image import RawRggbBayerData
    from pysp_amd.pipeline import DevicePipeline
    from pysp_amd.synth import rggb_frame
    wb, M = _wbM(orc)
    ctx = _lib.default_context()
    pipe = DevicePipeline(0)
    assert ctx.get_lab_mode() == 1 and pipe.ctx.get_lab_mode() == 1 and orc.DEFAULT_LAB_MODE == 1
    big = rggb_frame(1000, 1504, 1000)
    a = pipe.demosaic(torch.from_numpy(big).cuda(), wb, M, _lib.QUALITY_BEST, False, 0).cpu().numpy()
    assert np.array_equal(a, orc.demosaic_ahd(big, wb, M, False, 0))
    try:
        ctx.set_lab_mode("closed_form"); pipe.ctx.set_lab_mode(0); orc.set_lab_mode(0)
        for name in ("g8_labmode_closed_form_32x48", "g8_labmode_closed_form_34x50_hdr"):
            d, meta = load_golden(name)
            for st in (0, 1):
                im = RawRggbBayerData(d["bayer"], wbobj, 10.0, 1.0)
                im.set_hdr(meta["hdr"])
                assert np.array_equal(im.demosaic(QualityDemosaic.Best, st).image, d[f"ahd{st}"]), (name, st)
        for (H, W, hdr, stages) in ((130, 158, False, 0), (256, 384, False, 1), (200, 264, True, 1), (66, 130, True, 0), (2, 2, False, 0), (6, 4, True, 1)):
            bay = rggb_frame(H, W, 500 + H, scale=3.0 if hdr else 1.0, clip_hi=not hdr)
            got = pipe.demosaic(torch.from_numpy(bay).cuda(), wb, M, _lib.QUALITY_BEST, hdr, stages).cpu().numpy()
            assert np.array_equal(got, orc.demosaic_ahd(bay, wb, M, hdr, stages)), (H, W, hdr, stages)
        bay = _poisoned(96, 128, 8)
        got = pipe.demosaic(torch.from_numpy(bay).cuda(), wb, M, _lib.QUALITY_BEST, False, 0).cpu().numpy()
        assert _same(got, orc.demosaic_ahd(bay, wb, M, False, 0))
        b = pipe.demosaic(torch.from_numpy(big).cuda(), wb, M, _lib.QUALITY_BEST, False, 0).cpu().numpy()
        assert np.array_equal(b, orc.demosaic_ahd(big, wb, M, False, 0))
    finally:
        ctx.set_lab_mode(1); pipe.ctx.set_lab_mode(1); orc.set_lab_mode(orc.DEFAULT_LAB_MODE)
    flips = np.mean((a != b).any(axis=-1))
    assert 0.005 < flips < 0.08, flips                 # the two restatements disagree on a few percent of the H/V decisions


# ---- device pointers that are only float-aligned, and frame sizes on both sides of every tile edge ------------------------
def test_unaligned_device_pointers_and_tile_edge_sizes(orc, wbobj):
    """The kernels take 16-byte loads / stores where the image allows (W % 4 == 0 and, for the median stage, 16-byte aligned images);
    a caller may hand over any float-aligned pointer (a slice of a larger buffer).  Results on views that start 4 bytes past a
    16-byte boundary equal the aligned results bit for bit, and both equal the oracle, for sizes around the 60 x 28 px median tile,
    the 28 x 28 px select tile and the 64 x 32 px EAG / Draft tile, W = 0 and 2 mod 4."""
    import torch
    from pysp_amd import _lib
    from pysp_amd.pipeline import DevicePipeline
    wb, M = _wbM(orc)
    pipe = DevicePipeline(0)
    rng = np.random.default_rng(11)
    sizes = [(28, 60), (30, 62), (56, 120), (58, 124), (26, 58), (84, 180), (62, 130), (34, 66)]
    for H, W in sizes:
        bay = rng.random((H, W), dtype=np.float32)
        refs = {"ahd1": orc.pipeline_srgb(bay, wb, M, 2, False, 1, False), "ahd2_lin": orc.demosaic_ahd(bay, wb, M, False, 2),
                "eag": orc.pipeline_srgb(bay, wb, M, 1, False, 0, False), "draft": orc.pipeline_srgb(bay, wb, M, 0, False, 0, False)}
        for off in (0, 1):
            src = torch.empty(H * W + 4, dtype=torch.float32, device="cuda")
            d = src[off:off + H * W].view(H, W)
            d.copy_(torch.from_numpy(bay))
            buf = torch.full((H * W * 3 + 8,), -7.0, dtype=torch.float32, device="cuda")
            out = buf[off:off + H * W * 3].view(H, W, 3)
            assert (d.data_ptr() % 16 == 4 * off) and (out.data_ptr() % 16 == 4 * off)
            got = {}
            pipe.demosaic_to_srgb(d, wb, M, _lib.QUALITY_BEST, False, 1, False, out=out); pipe.sync(); got["ahd1"] = out.cpu().numpy().copy()
            pipe.demosaic(d, wb, M, _lib.QUALITY_BEST, False, 2, out=out); pipe.sync(); got["ahd2_lin"] = out.cpu().numpy().copy()
            pipe.demosaic_to_srgb(d, wb, M, _lib.QUALITY_FAST, False, 0, False, out=out); pipe.sync(); got["eag"] = out.cpu().numpy().copy()
            pipe.demosaic_to_srgb(d, wb, M, _lib.QUALITY_DRAFT, False, 0, False, out=out); pipe.sync(); got["draft"] = out.cpu().numpy().copy()
            for k in refs:
                assert np.array_equal(got[k], refs[k]), (H, W, off, k)
            rest = buf.cpu().numpy()
            assert (rest[:off] == -7.0).all() and (rest[off + H * W * 3:] == -7.0).all(), (H, W, off, "wrote outside the image")
