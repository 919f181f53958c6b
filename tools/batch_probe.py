"""The host batch chain (pysp_pipeline_batch_f32) at 24 MP, 8 frames, ms per frame per call: run under PYSP_BATCH_DEPTH=2|3|4|8|16 and PYSP_BAND_ROWS to see how far
the host may run ahead of the device (DESIGN.md 7.R5 b'): python tools/batch_probe.py   (on the GPU box)"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysp_amd import _lib
from pysp_amd.colorize.transform import final_matrix
from pysp_amd.synth import default_wb, rggb_frame
H, W = 4000, 6000
wbobj = default_wb()
L = _lib.lib(); ctx = _lib.default_context()
wb = _lib.wb3(wbobj.get_reciprocal_multipliers()); M = _lib.mat9(final_matrix(wbobj.get_matrix()))
nfr = 8
frames = [rggb_frame(H, W, 2000 + k) for k in range(nfr)]
outs = [_lib.empty_f32((H, W, 3)) for _ in range(nfr)]
hp = lambda a: ctypes.c_void_p(a.ctypes.data)
tab = lambda arrs: (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
for q in (2, 1):
    tb = []
    for it in range(5):
        t0 = time.perf_counter()
        _lib.check(L.pysp_pipeline_batch_f32(ctx.handle, tab(frames), nfr, H, W, wb, M, q, 0, 1, 2, tab(outs)))
        tb.append((time.perf_counter() - t0) * 1e3 / nfr)
    print("depth", os.environ.get("PYSP_BATCH_DEPTH"), "band", os.environ.get("PYSP_BAND_ROWS"), "q", q, " ".join("%.2f" % t for t in tb))
