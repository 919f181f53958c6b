#!/usr/bin/env python3
"""bench.py -- megapixels/s of the fused AHD demosaic + cam->sRGB path on synthetic 24 MP RGGB frames.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one frame of BASELINE.json configs[1]: 6000x4000 float32 RGGB mosaic, resident in HBM,
-> QualityDemosaic.Best (AHD, postprocess_steps=1) -> to_lin_srgb (clip + float64 CCM) ->
lin_srgb_to_srgb -> (H,W,3) float32 sRGB, resident in HBM; two kernels, one C-ABI call
(pysp_pipeline_srgb_dev).  Frames are independent, so ranks shard frames with no data-path
collective (weak scaling); the shared WB/CCM parameter block is broadcast from rank 0 over RCCL
once, before the timed region.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch  # first: keeps a single HIP runtime in the process (see pysp_amd/_lib.py)
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_PX = 16          # the fused path: 4 B mosaic read + 12 B RGB written
# per kernel: the select kernel reads the mosaic and writes RGB, a median stage reads and writes RGB
KERNEL_ALG_BYTES_PER_PX = {"k_ahd_select": 16, "k_ahd_median_stage": 24}

WORKLOADS = {
    # name: (H, W, quality, stages, description[, tail])   tail: 2 = to_lin_srgb + lin_srgb_to_srgb (default), 1 = to_lin_srgb only
    "ahd24": (4000, 6000, 2, 1, "24MP RGGB, QualityDemosaic.Best (AHD, postprocess_steps=1) + to_lin_srgb + lin_srgb_to_srgb"),
    "ahd24u16": (4000, 6000, 2, 1, "24MP RGGB uint16 sensor mosaic, bayer_normalize fused into the tile loader + AHD (postprocess_steps=1) + to_lin_srgb + lin_srgb_to_srgb (14 B/px)"),
    "eag24": (4000, 6000, 1, 0, "24MP RGGB, QualityDemosaic.Fast (EAG) + to_lin_srgb + lin_srgb_to_srgb"),
    "draft12": (3000, 4000, 0, 0, "12MP RGGB, QualityDemosaic.Draft + to_lin_srgb + lin_srgb_to_srgb"),
    "eag24ccm": (4000, 6000, 1, 0, "24MP RGGB, QualityDemosaic.Fast (EAG) + WB + 3x3 CCM (to_lin_srgb), BASELINE config 3 per frame", 1),
    "eag24raw": (4000, 6000, 1, 0, "24MP RGGB, QualityDemosaic.Fast (EAG) only (RawDemosaicData.image)", 0),
    # secondary kernels (BASELINE configs 4 and 5), reported with their own algorithmic bytes
    "fuse45": (5464, 8192, -1, 0, "raw_hdr fuse_exposures_to_raw, 7 x 45MP exposures -> HDR mosaic + count (36 B per output px)"),
    "warp100": (8736, 11648, -2, 0, "100MP RGB, DNG WarpRectilinear per-channel Lanczos-4 remap (24 B/px)"),
}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="ahd24", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=3, help="distinct resident input frames per rank, cycled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams (contexts) the frames of a rank are cycled over.  With 2, consecutive (independent) frames overlap: the next "
                         "frame's first kernel fills the drain of the previous frame's last one, +2 % throughput, but concurrent kernels "
                         "stretch each other, so per-kernel durations (rocprofv3's, too) stop meaning anything; the default keeps them clean")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the N > 1 path on one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
    dist = None
    n_dev = max(1, torch.cuda.device_count())
    dev_index = local_rank % n_dev                 # one rank per GPU; the modulo only matters for a gloo rehearsal on one GPU
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend="gloo")
    else:
        torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if (dist is None or args.backend == "nccl") else torch.device("cpu")

    from pysp_amd import _lib
    from pysp_amd.colorize.transform import final_matrix
    from pysp_amd.synth import default_wb, rggb_frame

    H, W, quality, stages, desc = WORKLOADS[args.workload][:5]
    tail = WORKLOADS[args.workload][5] if len(WORKLOADS[args.workload]) > 5 else 2
    mp_per_frame = H * W / 1e6

    # ---- shared parameters: rank 0 owns them, everyone receives them over RCCL/xGMI (once per batch)
    from pysp_amd.multi_gpu import broadcast_params
    if rank == 0:
        wbobj = default_wb()
        wb_np, M_np = broadcast_params(wbobj.get_reciprocal_multipliers(), final_matrix(wbobj.get_matrix()), src=0, device=coll_dev)
    else:
        wb_np, M_np = broadcast_params(np.zeros(3, np.float32), np.zeros((3, 3)), src=0, device=coll_dev)
    p = np.concatenate([wb_np.astype(np.float64), M_np.reshape(-1)])
    wb = _lib.wb3(wb_np)
    M = _lib.mat9(M_np)

    n_streams = max(1, args.streams)
    ctxs = [_lib.Context(dev_index) for _ in range(n_streams)]   # own HIP streams; kernels are timed with events on THOSE streams
    ctx = ctxs[0]
    L = _lib.lib()
    alg_bytes_per_px = ALG_BYTES_PER_PX

    if quality >= 0:
        # ---- inputs resident in HBM: frame i of rank r uses seed 1000 + r*frames + i
        frames = [torch.from_numpy(rggb_frame(H, W, 1000 + rank * args.frames + i)).to(dev) for i in range(max(1, args.frames))]
        outs = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(n_streams)]

        if args.workload.endswith("u16"):
            # 14-bit sensor counts, black 512, saturation 15871 per site (normalization.py:4-24 runs inside the tile loaders)
            frames = [(f * 15359.0 + 512.0).round().clamp(0, 16383).to(torch.int32).to(torch.int16).view(torch.uint16).contiguous() for f in frames]
            black = (ctypes.c_float * 4)(512.0, 512.0, 512.0, 512.0)
            sat = (ctypes.c_float * 4)(15871.0, 15871.0, 15871.0, 15871.0)
            alg_bytes_per_px = 14

            def step(i: int) -> None:
                f = frames[i % len(frames)]
                s = i % n_streams
                _lib.check(L.pysp_pipeline_u16_dev(ctxs[s].handle, ctypes.c_void_p(f.data_ptr()), H, W, black, sat, wb, M, quality, 0, stages, tail,
                                                   ctypes.c_void_p(outs[s].data_ptr())))
        else:
            def step(i: int) -> None:
                f = frames[i % len(frames)]
                s = i % n_streams                       # frame i runs on stream s, writing that stream's output buffer
                _lib.check(L.pysp_pipeline_dev(ctxs[s].handle, ctypes.c_void_p(f.data_ptr()), H, W, wb, M, quality, 0, stages, tail,
                                               ctypes.c_void_p(outs[s].data_ptr())))
    elif quality == -1:
        K = 7
        base = rggb_frame(H, W, 1000 + rank, scale=8.0, clip_hi=False)
        frames = [torch.from_numpy(np.clip(base * np.float32(2.0 ** -k), 0, 1)).to(dev) for k in range(K)]
        del base
        out = torch.empty((H, W), dtype=torch.float32, device=dev)
        cnt = torch.empty((H, W), dtype=torch.int32, device=dev)
        offs = [2.0 ** (10 + k - 13.0) for k in range(K)]
        site_w = np.array([wb_np[0], wb_np[1], wb_np[2], wb_np[1]], dtype=np.float32)
        bias = np.ascontiguousarray(np.stack([1.6 ** (-0.1 * np.abs(o * site_w)) for o in offs]).astype(np.float32))
        off32 = np.array(offs, dtype=np.float32)
        ptrs = (ctypes.c_void_p * K)(*[f.data_ptr() for f in frames])
        fp = ctypes.POINTER(ctypes.c_float)
        alg_bytes_per_px = 4 * K + 8

        def step(i: int) -> None:
            _lib.check(L.pysp_fuse_raw_dev(ctx.handle, ptrs, K, H, W, off32.ctypes.data_as(fp), bias.ctypes.data_as(fp), K - 1,
                                           ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cnt.data_ptr())))
    else:
        frames = [torch.rand((H, W, 3), dtype=torch.float32, device=dev, generator=torch.Generator(device=dev).manual_seed(1000 + rank))]
        out = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        coeffs = np.array([[1.0, 0.01, 0.002, 0.0, 0.0, 0.0], [1.0, 0.0, 0.002, 0.0, 0.0, 0.0], [1.0, -0.01, 0.002, 0.0, 0.0, 0.0]])
        cptr = coeffs.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        alg_bytes_per_px = 24

        def step(i: int) -> None:
            _lib.check(L.pysp_warp_rectilinear_dev(ctx.handle, ctypes.c_void_p(frames[0].data_ptr()), ctypes.c_void_p(out.data_ptr()), H, W,
                                                   cptr, 3, 0.5, 0.5, 1.0))
    torch.cuda.synchronize()

    def fence() -> None:
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    fence()

    # ---- timed region: exactly K steps, nothing but the kernels on the stream (no event records inside)
    for c in ctxs:
        c.set_kernel_timing(0)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations: the same steps again, each kernel bracketed by HIP events on the launch
    # stream (recording them inside the timed region would add ~50 us of event traffic to every 1.1 ms step)
    # (one stream only here, so that a kernel's duration is not stretched by a neighbour sharing the CUs)
    fence()
    ctx.set_kernel_timing(2)
    samples: dict = {}
    for i in range(min(16, max(4, args.steps))):
        step(i * n_streams)
        for name, ms in ctx.kernel_times():
            samples.setdefault(name, []).append(ms)
    ctx.set_kernel_timing(1)
    per_kernel = {k: float(np.mean(v)) for k, v in samples.items()}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps * mp_per_frame / elapsed
    dom = max(per_kernel, key=per_kernel.get) if per_kernel else None
    alg_bytes = KERNEL_ALG_BYTES_PER_PX.get(dom, alg_bytes_per_px) * H * W      # of the dominant kernel's own launch
    roofline = None
    traffic = None
    try:   # HBM bytes per launch from the PMC passes (rocprofv3 cannot run inside the benchmark itself)
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
        if args.workload == "ahd24" and dom in tj:
            traffic = tj[dom]
    except (OSError, ValueError):
        pass
    if dom:
        achieved = alg_bytes / (per_kernel[dom] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": round(per_kernel[dom], 4),
                    "all_kernels_ms": {k: round(v, 4) for k, v in per_kernel.items()},
                    "alg_bytes_per_px": KERNEL_ALG_BYTES_PER_PX.get(dom, alg_bytes_per_px),
                    "pipeline_frac": round(alg_bytes_per_px * H * W / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "note": "AHD is VALU-issue bound (725 + 543 wave-level instructions per pixel in the two kernels vs 16 B/px for the path, one every 3.0 / 3.3 cycles per SIMD, profiles/r1_v19_pmc_summary.csv); pipeline_frac is the whole two-kernel step against the path's 16 B/px; the HBM fraction is reported as required, not expected to approach 1"}

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline and quality >= 0:
        try:
            from oracle import oracle
            Mo = p[3:].reshape(3, 3)
            done, dt = 0, 0.0
            for f in frames:                         # bounded sample: whole resident frames until about 10 s of CPU work are spent
                sample = np.ascontiguousarray(f.cpu().numpy())
                t1 = time.perf_counter()
                oracle.pipeline_srgb(sample, p[:3].astype(np.float32), Mo, quality, False, stages, False)
                dt += time.perf_counter() - t1
                done += 1
                if dt > 10.0:
                    break
            cpu_baseline = {"value": round(done * H * W / 1e6 / dt, 3), "unit": "MP/s", "cores": oracle.threads(), "kind": "port",
                            "sample": f"{done} whole frame(s) of the benchmark ({H}x{W}), {dt:.1f} s, oracle/pysp_oracle.c, OpenMP, same path"}
        except Exception as exc:  # the oracle is a checker, never a dependency of the measured path
            cpu_baseline = {"value": None, "unit": "MP/s", "cores": os.cpu_count(), "kind": "port", "sample": f"unavailable: {exc}"}

    line = {
        "metric": "megapixels/sec AHD debayer+cam->sRGB, 24MP RGGB" if args.workload == "ahd24" else f"megapixels/sec {args.workload}",
        "value": round(value, 2), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "H": H, "W": W, "frames_per_rank_resident": len(frames), "streams_per_rank": n_streams, "sharding": "frame-parallel, no data-path collective; WB/CCM broadcast once over RCCL"},
        "roofline": roofline, "cpu_baseline": cpu_baseline,
    }
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
