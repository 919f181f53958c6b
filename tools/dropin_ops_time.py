"""Wall time (host work + PCIe + kernels) against kernel time of the drop-in NumPy entry points beside the demosaic call, 24 MP, on the GPU box:
    python tools/dropin_ops_time.py
Shows where a drop-in call's time goes when its inputs and outputs are host arrays, as with the reference."""
import os, struct, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pysp_amd
from pysp_amd import _lib
from pysp_amd.base_types.image_base import RawDemosaicData
from pysp_amd.dng_warp_corr import apply_opcode_3_warp
from pysp_amd.image import RawRggbBayerData
from pysp_amd.raw_bad_pixel_corr import find_erroneous_pixels_threshold
from pysp_amd.raw_correction import flat_frame_correction
from pysp_amd.raw_hdr import fuse_exposures_from_debayer, fuse_exposures_to_raw
from pysp_amd.synth import default_wb, rggb_frame

H, W = 4000, 6000
wbobj = default_wb()
ctx = _lib.default_context()


def timed(name, fn, bytes_moved, reps=4):
    best, k = 1e9, 0.0
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); dt = (time.perf_counter() - t0) * 1e3
        if dt < best:
            best, k = dt, ctx.last_kernel_ms()
    print("%-34s wall %7.1f ms   kernels %6.3f ms   PCIe at 57 GB/s %5.1f ms   rest (host) %6.1f ms" % (name, best, k, bytes_moved / 57e9 * 1e3, best - k - bytes_moved / 57e9 * 1e3))


bay = rggb_frame(H, W, 1000)
flat = RawRggbBayerData(np.clip(rggb_frame(H, W, 7) * 0.2 + 0.7, 0, 1), wbobj, 10.0, 1.0)
img = RawRggbBayerData(bay.copy(), wbobj, 10.0, 1.0)
timed("flat_frame_correction", lambda: flat_frame_correction(img, flat), 3 * H * W * 4)
timed("find_erroneous_pixels_threshold", lambda: find_erroneous_pixels_threshold(img), H * W * 4 + H * W)
K = 7
exps = [RawRggbBayerData(np.clip(bay * np.float32(2.0 ** -k), 0, 1), wbobj, 10.0 + k, 1.0) for k in range(K)]
timed("fuse_exposures_to_raw, K=7", lambda: fuse_exposures_to_raw(exps), (K + 2) * H * W * 4)
pysp_amd.set_lazy(False)
dem = []
for k in range(3):
    d = RawDemosaicData(np.repeat(exps[k].sensor_scaled[:, :, None], 3, axis=2), wbobj.get_reciprocal_multipliers())
    d.mat_xyz = wbobj.get_matrix(); d.current_ev = 10.0 + k
    dem.append(d)
timed("fuse_exposures_from_debayer, K=3", lambda: fuse_exposures_from_debayer(dem), (3 + 2) * H * W * 12, reps=3)
rgb = np.repeat(bay[:, :, None], 3, axis=2).copy()
coeffs = [[1.0, 0.01, 0.002, 0.0, 0.0, 0.0], [1.0, -0.01, 0.002, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0, 0.001, -0.001]]
payload = struct.pack(">I", 3) + b"".join(struct.pack(">6d", *c) for c in coeffs) + struct.pack(">2d", 0.5, 0.5)
blob = struct.pack(">I", 1) + struct.pack(">IIII", 1, 1, 0, len(payload)) + payload
timed("apply_opcode_3_warp", lambda: apply_opcode_3_warp(rgb, blob), 2 * H * W * 12)
