"""End-to-end (PCIe-inclusive) timing of the drop-in NumPy API at 24 MP on the GPU box: python tools/dropin_time.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pysp_amd import _lib
from pysp_amd.colorize import lin_srgb_to_srgb
from pysp_amd.colorize.transform import final_matrix
from pysp_amd.const import QualityDemosaic
from pysp_amd.image import RawRggbBayerData
from pysp_amd.synth import default_wb, rggb_frame

H, W = 4000, 6000
bay = rggb_frame(H, W, 1000)
wbobj = default_wb()
for it in range(8):
    t0 = time.perf_counter()
    raw = RawRggbBayerData(bay, wbobj, 10.0, 1.0)
    lin = raw.demosaic(QualityDemosaic.Best).to_lin_srgb()
    t1 = time.perf_counter()
    srgb = lin_srgb_to_srgb(lin)
    t2 = time.perf_counter()
    print("README recipe: demosaic+to_lin_srgb %.1f ms, lin_srgb_to_srgb %.1f ms, total %.1f ms = %.2f GMP/s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, H * W / 1e9 / (t2 - t0)))
# the same three calls in the opt-in deferred mode: nothing runs until lin_srgb_to_srgb asks for the result, then ONE banded host call (upload || kernels || download)
import pysp_amd
pysp_amd.set_lazy("deferred")
ts = []
for it in range(12):
    t0 = time.perf_counter()
    raw = RawRggbBayerData(bay, wbobj, 10.0, 1.0)
    srgb_d = lin_srgb_to_srgb(raw.demosaic(QualityDemosaic.Best).to_lin_srgb())
    ts.append((time.perf_counter() - t0) * 1e3)
pysp_amd.set_lazy(True)
assert np.array_equal(srgb_d, srgb)
print("README recipe, pysp_amd.set_lazy(\"deferred\") (opt-in: the mosaic must stay untouched until the result is read): first %.1f ms, then min %.2f / median %.2f / max %.2f ms = %.2f GMP/s at the median"
      % (ts[0], min(ts[1:]), sorted(ts[1:])[len(ts) // 2], max(ts[1:]), H * W / 1e6 / sorted(ts[1:])[len(ts) // 2]))
L = _lib.lib(); ctx = _lib.default_context()
wb = _lib.wb3(wbobj.get_reciprocal_multipliers()); M = _lib.mat9(final_matrix(wbobj.get_matrix()))
out = _lib.empty_f32((H, W, 3))          # page-locked result buffer, as the Python wrappers use
ts = []
for it in range(12):
    t0 = time.perf_counter()
    _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(bay), H, W, wb, M, 2, 0, 1, 0, _lib.ptr(out)))
    ts.append((time.perf_counter() - t0) * 1e3)
print("one fused call (host buffers, pinned result): first %.1f ms, then min %.2f / median %.2f / max %.2f ms = %.2f GMP/s at the median" % (ts[0], min(ts[1:]), sorted(ts[1:])[len(ts) // 2], max(ts[1:]), H * W / 1e6 / sorted(ts[1:])[len(ts) // 2]))
assert np.array_equal(out, srgb)
pag_out = np.empty((H, W, 3), np.float32); pag_out[:] = 0
ts = []
for it in range(12):
    t0 = time.perf_counter()
    _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(bay), H, W, wb, M, 2, 0, 1, 0, _lib.ptr(pag_out)))
    ts.append((time.perf_counter() - t0) * 1e3)
print("one fused call (host buffers, pageable result): min %.2f / median %.2f / max %.2f ms" % (min(ts[1:]), sorted(ts[1:])[len(ts) // 2], max(ts[1:])))
# a batch of frames through one band chain (pysp_pipeline_batch_f32): ms per frame against the same frames one call each
nfr = 8
frames = [rggb_frame(H, W, 2000 + k) for k in range(nfr)]
outs = [_lib.empty_f32((H, W, 3)) for _ in range(nfr)]
tab = lambda arrs: (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
for q, name in ((2, "AHD + sRGB"), (1, "EAG + sRGB")):
    tb, tsn = [], []
    for it in range(6):
        t0 = time.perf_counter()
        _lib.check(L.pysp_pipeline_batch_f32(ctx.handle, tab(frames), nfr, H, W, wb, M, q, 0, 1, 2, tab(outs)))
        tb.append((time.perf_counter() - t0) * 1e3 / nfr)
        t0 = time.perf_counter()
        for f, o in zip(frames, outs):
            _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(f), H, W, wb, M, q, 0, 1, 0, _lib.ptr(o)))
        tsn.append((time.perf_counter() - t0) * 1e3 / nfr)
    print("batch of %d host frames, %s: one chain min %.2f / median %.2f ms per frame = %.2f GMP/s; one call per frame min %.2f / median %.2f ms" %
          (nfr, name, min(tb[1:]), sorted(tb[1:])[len(tb) // 2], H * W / 1e6 / sorted(tb[1:])[len(tb) // 2], min(tsn[1:]), sorted(tsn[1:])[len(tsn) // 2]))
# raw copy rates
d = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
pin = torch.empty(H * W * 3, dtype=torch.float32).pin_memory()
pag = torch.from_numpy(out.reshape(-1))
for name, h in (("pageable", pag), ("pinned", pin)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    h.copy_(d, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s: H2D %.1f GB/s, D2H %.1f GB/s" % (name, h.numel() * 4 / 1e9 / (t1 - t0), h.numel() * 4 / 1e9 / (t2 - t1)))
