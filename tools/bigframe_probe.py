"""PCIe-inclusive host call on a 100 MP frame (35 bands): the single call (no limit on the host's run-ahead) against the batch chain with n = 1 (run-ahead 2 bands)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysp_amd import _lib
from pysp_amd.colorize.transform import final_matrix
from pysp_amd.synth import default_wb, rggb_frame
wbobj = default_wb()
L = _lib.lib(); ctx = _lib.default_context()
wb = _lib.wb3(wbobj.get_reciprocal_multipliers()); M = _lib.mat9(final_matrix(wbobj.get_matrix()))
tab = lambda arrs: (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
for (H, W) in ((8736, 11648), (4000, 6000), (5464, 8192)):
    f = rggb_frame(H, W, 77)
    o = _lib.empty_f32((H, W, 3))
    for q in (1, 2):
        ts, tb = [], []
        for it in range(6):
            t0 = time.perf_counter()
            _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(f), H, W, wb, M, q, 0, 1, 0, _lib.ptr(o)))
            ts.append((time.perf_counter() - t0) * 1e3)
            t0 = time.perf_counter()
            _lib.check(L.pysp_pipeline_batch_f32(ctx.handle, tab([f]), 1, H, W, wb, M, q, 0, 1, 2, tab([o])))
            tb.append((time.perf_counter() - t0) * 1e3)
        fl = H * W * 12 / 57e9 * 1e3
        print("%dx%d q=%d: single call %s | chain n=1 %s | download alone at 57 GB/s: %.2f ms" % (H, W, q, " ".join("%.2f" % t for t in ts[1:]), " ".join("%.2f" % t for t in tb[1:]), fl), flush=True)
