"""Timing of the chromatic-aberration removal at 24 MP: host-buffer drop-in call vs device-resident batch call.
Run on the GPU box:  python tools/ca_time.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pysp_amd import _lib
from pysp_amd.corr_ca import remove_ca_from_raw
from pysp_amd.corr_ca.model.poly5 import Poly5CorrectionModel
from pysp_amd.image import RawRggbBayerData
from pysp_amd.pipeline import DevicePipeline
from pysp_amd.synth import default_wb, rggb_frame
H, W = 4000, 6000
bay = rggb_frame(H, W, 1000)
wbobj = default_wb()
raw = RawRggbBayerData(bay.copy(), wbobj, 10.0, 1.0)
mr, mb = Poly5CorrectionModel(0.004, -0.001), Poly5CorrectionModel(-0.003, 0.001)
for it in range(2):
    raw.sensor_scaled = bay.copy()
    t0 = time.perf_counter(); remove_ca_from_raw(raw, mr, mb); t1 = time.perf_counter()
    print("drop-in call: total %.1f ms (host lens fields + PCIe), kernels %.3f ms" % ((t1 - t0) * 1e3, _lib.default_context().last_kernel_ms()))
pipe = DevicePipeline(0)
t0 = time.perf_counter(); fr, fb = pipe.lens_fields(mr, (H, W)), pipe.lens_fields(mb, (H, W)); t1 = time.perf_counter()
print("lens fields once per lens: %.1f ms" % ((t1 - t0) * 1e3))
d = torch.from_numpy(bay).cuda()
wb = wbobj.get_reciprocal_multipliers()
for it in range(5):
    pipe.remove_ca(d, wb, fr, fb); pipe.sync()
    print("device-resident: kernels %.3f ms = %.1f GMP/s" % (pipe.ctx.last_kernel_ms(), H * W / 1e9 / (pipe.ctx.last_kernel_ms() * 1e-3)))
pipe.ctx.set_kernel_timing(2)
pipe.remove_ca(d, wb, fr, fb); pipe.sync()
print(pipe.ctx.kernel_times())
