"""Randomised GPU-vs-oracle parity run (bit-exact): python tests/fuzz_parity.py [seconds] [seed]
Test infrastructure (it calls the CPU oracle); a 10-second slice of it runs inside the GPU suite (test_gpu_parity.py::test_fuzz_slice).
Random even frame sizes, value distributions (uniform, heavy-tailed, saturated, zeros, negatives, tiny), qualities, HDR flag,
post-process stage counts, colour tails, Lab layout / Lab mode / select-kernel form of the context, Draft / EAG batches in one grid, uint16 input, CA removal, raw fusion, WarpRectilinear (fused kernel vs its own tables through the oracle remap).  Stops at the first mismatch with a reproducer line."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import oracle as orc
from pysp_amd import _lib
from pysp_amd.colorize.transform import final_matrix
from pysp_amd.pipeline import DevicePipeline
from pysp_amd.synth import default_wb

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
pipe = DevicePipeline(0)
wbobj = default_wb()
M0 = final_matrix(wbobj.get_matrix())


def sprinkle(rng, a):
    """Quiet NaN / +Inf / -Inf at a few random sites (anywhere, borders included)."""
    n = int(rng.integers(1, 6))
    for _ in range(n):
        a[int(rng.integers(0, a.shape[0])), int(rng.integers(0, a.shape[1]))] = (np.nan, np.inf, -np.inf)[int(rng.integers(0, 3))]
    return a


def frame(rng, H, W):
    kind = rng.integers(0, 7)
    if kind == 0: a = rng.random((H, W))
    elif kind == 1: a = rng.random((H, W)) ** 4
    elif kind == 2: a = np.clip(rng.normal(0.8, 0.4, (H, W)), 0, 1)
    elif kind == 3: a = rng.random((H, W)) * (rng.random((H, W)) > 0.5)
    elif kind == 4: a = rng.normal(0.3, 0.5, (H, W))                      # negatives and > 1
    elif kind == 5: a = rng.random((H, W)) * 1e-4
    else: a = np.round(rng.random((H, W)) * 4) / 4                          # many exact ties
    return a.astype(np.float32)


t_end = time.time() + budget
n = 0
counts = {}
cur_layout = None
while time.time() < t_end:
    seed = seed0 + n
    rng = np.random.default_rng(seed)
    if rng.random() < 0.02:                                   # now and then a frame of many tiles
        H, W = 2 * int(rng.integers(150, 400)), 2 * int(rng.integers(200, 520))
    else:
        H, W = 2 * int(rng.integers(1, 120)), 2 * int(rng.integers(1, 160))
    bay = frame(rng, H, W)
    nonfinite = rng.random() < 0.15          # non-finite sites: exact parity is claimed up to (not through) a median stage
    if nonfinite:
        bay = sprinkle(rng, bay)
    wb = (1.0 / rng.uniform(0.3, 1.0, 3)).astype(np.float32)
    M = M0 * rng.uniform(0.8, 1.2, (3, 3)) if rng.random() < 0.5 else M0
    case = int(rng.integers(0, 7))
    d = torch.from_numpy(bay).cuda()
    # context switches of the AHD kernels (round 4): Lab layout automatic / packed cells / float planes -- same bits in every one -- and now and then the
    # closed-form Lab metric (mode 0) on both sides.  From its own generator, so that the case streams of earlier seeds stay what they were.
    # The layout changes every 64 cases only (setting it restarts the automatic policy, which needs 16 launches per sample to move at all).
    layout = ("auto", "packed", "planes", "auto")[int(np.random.default_rng((seed // 64) ^ 0x5EED).integers(0, 4))]
    if layout != cur_layout:
        pipe.ctx.set_lab_layout(layout); cur_layout = layout
    lab_mode = 0 if np.random.default_rng(seed ^ 0x5EED).random() < 0.1 else 1
    pipe.ctx.set_lab_mode(lab_mode); orc.set_lab_mode(lab_mode)
    # round 5: the form of the select kernel (tiles / streaming down the columns: same bits), from its own generator as well
    form = ("tile", "stream")[int(np.random.default_rng(seed ^ 0xF0F0).integers(0, 2))]
    pipe.ctx.set_select_form(form)
    if case == 0:      # AHD with stages / hdr
        stages, hdr = int(rng.integers(0, 4)), bool(rng.integers(0, 2))
        if nonfinite: stages = 0
        got = pipe.demosaic(d, wb, M, _lib.QUALITY_BEST, hdr, stages); pipe.sync()
        ref = orc.demosaic_ahd(bay, wb, M, hdr, stages)
        tag = f"ahd stages={stages} hdr={hdr} layout={layout} lab_mode={lab_mode} form={form}"
    elif case == 1:    # fused pipeline to sRGB, any quality
        q, stages, hdr, rh = int(rng.integers(0, 3)), int(rng.integers(0, 3)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        if nonfinite: stages = 0
        got = pipe.demosaic_to_srgb(d, wb, M, q, hdr, stages, rh); pipe.sync()
        ref = orc.pipeline_srgb(bay, wb, M, q, hdr, stages, rh)
        tag = f"srgb q={q} stages={stages} hdr={hdr} reinhard={rh} layout={layout} lab_mode={lab_mode} form={form}"
    elif case == 2:    # EAG / Draft raw; a third of the time as a batch of 2-5 frames in one grid (round 5), the LAST frame of the batch is the one compared
        q = int(rng.integers(0, 2))
        nb = int(rng.integers(2, 6)) if rng.random() < 0.33 else 1
        if nb == 1:
            got = pipe.demosaic(d, wb, M, q, False, 0); pipe.sync()
        else:
            others = [torch.from_numpy(frame(rng, H, W)).cuda() for _ in range(nb - 1)]
            got = pipe.batch(others + [d], wb, M, q, False, 0, 0)[-1]; pipe.sync()
        ref = orc.demosaic_eag(bay, wb) if q == 1 else orc.demosaic_draft(bay, wb)
        tag = f"raw q={q} batch={nb}"
    elif case == 3:    # uint16 loader
        raw = (rng.random((H, W)) * 16383).astype(np.uint16)
        black, sat = rng.uniform(0, 600, 4).astype(np.float32), rng.uniform(8000, 16383, 4).astype(np.float32)
        q = int(rng.integers(0, 3))
        got = pipe.raw_u16_to_rgb(torch.from_numpy(raw.view(np.int16)).cuda(), black, sat, wb, M, q, 1, 2); pipe.sync()
        ref = orc.pipeline_srgb(orc.bayer_normalize(raw, black, sat), wb, M, q, False, 1, False)
        tag = f"u16 q={q} layout={layout} lab_mode={lab_mode} form={form}"
    elif case == 4:    # CA removal with random smooth quadrant fields
        if H < 4 or W < 4 or nonfinite:
            n += 1; continue
        h, w = H // 2, W // 2
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        def field(s):
            dy = (yy - (H - 1) / 2) * np.float32(1 + s * rng.uniform(-0.05, 0.05)); dx = (xx - (W - 1) / 2) * np.float32(1 + s * rng.uniform(-0.05, 0.05))
            return np.ascontiguousarray(np.stack([dy, dx], -1).astype(np.float32))
        f = [field(1), field(1), field(1), field(1)]
        t = [torch.from_numpy(a).cuda() for a in f]
        pipe.remove_ca(d, wb, (t[0], t[1]), (t[2], t[3])); pipe.sync()
        got = d
        ref = orc.remove_ca(bay, f[0], f[1], float(wb[0]), f[2], f[3], float(wb[2]))
        tag = "ca"
    elif case == 6:    # WarpRectilinear: the fused kernel against its own tables through the oracle's remap (bit-exact), random coefficients
        if nonfinite:
            n += 1; continue
        import struct
        from pysp_amd.dng_warp_corr import apply_opcode_3_warp
        from pysp_amd.dng_warp_corr.dng_warp_rectilinear_coords import compute_remapping_table
        img = rng.random((H, W, 3), dtype=np.float32)
        mag = 10.0 ** rng.uniform(-4, -0.3)
        cf = np.array([[1.0 + rng.normal(0, mag), rng.normal(0, mag), rng.normal(0, mag / 3), rng.normal(0, mag / 10), rng.normal(0, mag / 10), rng.normal(0, mag / 10)] for _ in range(3)])
        cx, cy = float(rng.uniform(0.2, 0.8)), float(rng.uniform(0.2, 0.8))
        payload = struct.pack(">I", 3) + b"".join(struct.pack(">6d", *c) for c in cf) + struct.pack(">2d", cx, cy)
        g = img.copy()
        apply_opcode_3_warp(g, struct.pack(">I", 1) + struct.pack(">IIII", 1, 1, 0, len(payload)) + payload)
        ref = np.empty_like(img)
        for c in range(3):
            t = compute_remapping_table(*cf[c], W, H, cx, cy, 1.0)
            ref[..., c] = orc.remap_lanczos4(np.ascontiguousarray(img[..., c]), np.clip(t[..., 0], 0, W - 1), np.clip(t[..., 1], 0, H - 1))
        got = torch.from_numpy(g)
        tag = "warp"
    else:              # raw HDR fusion
        K = int(rng.integers(2, 6))
        frames = [np.clip(frame(rng, H, W), 0, 1) for _ in range(K)]
        if nonfinite:
            frames[int(rng.integers(0, K))] = sprinkle(rng, frames[0].copy())
        evs = [float(10 + k + rng.uniform(-0.3, 0.3)) for k in range(K)]
        fused, count, _, _ = pipe.fuse_raw([torch.from_numpy(x).cuda() for x in frames], evs, wb); pipe.sync()
        rf, rc = orc.fuse_raw(frames, evs, wb)[:2]
        got, ref = fused, rf
        if not np.array_equal(count.cpu().numpy(), rc):
            print(f"MISMATCH (count) seed={seed} {H}x{W} fuse K={K}"); sys.exit(1)
        tag = f"fuse K={K}"
    g = got.cpu().numpy()
    same = np.array_equal(g, ref) or (np.array_equal(np.isnan(g), np.isnan(ref)) and np.array_equal(np.nan_to_num(g, nan=0.0), np.nan_to_num(ref, nan=0.0)))
    if not same:
        bad = np.argwhere(g != ref)
        print(f"MISMATCH seed={seed} {H}x{W} {tag}: {len(bad)} values, first at {bad[0]}: gpu {g[tuple(bad[0])]!r} oracle {ref[tuple(bad[0])]!r}")
        sys.exit(1)
    counts[tag.split()[0]] = counts.get(tag.split()[0], 0) + 1
    if tag.split()[0] in ("ahd", "srgb", "u16") and lab_mode == 1:
        counts["form_" + form] = counts.get("form_" + form, 0) + 1
        k = "planes_kernel_next" if pipe.ctx.lab_layout_in_use() == 1 else "packed_kernel_next"      # what the policy / the switch selects after this case
        counts[k] = counts.get(k, 0) + 1
    n += 1
    if n % 200 == 0:
        print(f"{n} cases ok, {counts}", flush=True)
print(f"done: {n} cases bit-exact, {counts}")
