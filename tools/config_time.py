"""Device-resident timing of BASELINE configs 4 and 5 on one GPU (kernels only, inputs in HBM): python tools/config_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pysp_amd.colorize.transform import final_matrix
from pysp_amd.pipeline import DevicePipeline
from pysp_amd.synth import default_wb, rggb_frame

pipe = DevicePipeline(0)
wbobj = default_wb()
wb, M = wbobj.get_reciprocal_multipliers(), final_matrix(wbobj.get_matrix())


def timed(fn, reps=5):
    fn(); pipe.sync(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); pipe.sync(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


# config 4: 7 x 45 MP exposures -> raw fusion -> AHD (HDR vote) -> to_lin_srgb -> x/(1+x) -> sRGB
H, W, K = 5464, 8192, 7
base = rggb_frame(H, W, 1000, scale=8.0, clip_hi=False)
frames = [torch.from_numpy(np.clip(base * np.float32(2.0 ** -k), 0, 1)).cuda() for k in range(K)]
evs = [10.0 + k for k in range(K)]
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
ms = timed(lambda: pipe.hdr_stack_to_srgb(frames, evs, wbobj, stages=1, out=out))
print("config 4 (7 x 45 MP fuse + AHD(HDR) + tone map + sRGB): %.2f ms = %.1f GMP/s of output" % (ms, H * W / 1e9 / (ms * 1e-3)))
del frames, out, base
torch.cuda.empty_cache()

# config 5 on one GPU: 100 MP AHD(postprocess_stages=3) + WarpRectilinear
H, W = 8736, 11648
bay = torch.from_numpy(rggb_frame(H, W, 1001)).cuda()
coeffs = np.array([[1.0, 0.01, 0.002, 0, 0, 0], [1.0, 0.0, 0.002, 0, 0, 0], [1.0, -0.01, 0.002, 0, 0, 0]])
rgb5, out5 = torch.empty((bay.shape[0], bay.shape[1], 3), dtype=torch.float32, device="cuda"), torch.empty((bay.shape[0], bay.shape[1], 3), dtype=torch.float32, device="cuda")
ms = timed(lambda: pipe.demosaic_warp(bay, wb, M, coeffs, (0.5, 0.5), stages=3, rgb=rgb5, out=out5), reps=3)
print("config 5 on 1 GPU (100 MP AHD(3) + warp): %.2f ms = %.1f GMP/s" % (ms, H * W / 1e9 / (ms * 1e-3)))
