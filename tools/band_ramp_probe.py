"""The band schedule of the host pipelines at 24 MP (and 100 MP): ms per fused host call (page-locked result) for uniform bands and for the ramp under its two knobs.
python tools/band_ramp_probe.py   (on the GPU box)"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysp_amd import _lib
from pysp_amd.colorize.transform import final_matrix
from pysp_amd.synth import default_wb, rggb_frame
wbobj = default_wb()
L = _lib.lib(); ctx = _lib.default_context()
wb = _lib.wb3(wbobj.get_reciprocal_multipliers()); M = _lib.mat9(final_matrix(wbobj.get_matrix()))


def run(f, o, H, W, q=2, n=12):
    ts = []
    for it in range(n):
        t0 = time.perf_counter()
        _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(f), H, W, wb, M, q, 0, 1, 0, _lib.ptr(o)))
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[2:])
    return ts[0], ts[len(ts) // 2]


for (H, W) in ((4000, 6000), (8736, 11648)):
    f = rggb_frame(H, W, 77); o = _lib.empty_f32((H, W, 3)); pag = np.empty((H, W, 3), np.float32); pag[:] = 0
    for k in ("PYSP_BAND_ROWS", "PYSP_BAND_FIRST_PX", "PYSP_BAND_CAP_PX"): os.environ.pop(k, None)
    for rows in (256, 512):
        os.environ["PYSP_BAND_ROWS"] = str(rows)
        print("%dx%d uniform %4d rows: min %.2f median %.2f ms" % ((H, W, rows) + run(f, o, H, W)), flush=True)
    os.environ.pop("PYSP_BAND_ROWS")
    for first in (256, 512, 1024):
        for cap in (3, 6, 12, 24):
            os.environ["PYSP_BAND_FIRST_PX"] = str(first * 1024); os.environ["PYSP_BAND_CAP_PX"] = str(cap * 1024 * 1024)
            print("%dx%d ramp first %4d Kpx cap %2d Mpx: min %.2f median %.2f ms | EAG min %.2f median %.2f | pageable result min %.2f median %.2f" % ((H, W, first, cap) + run(f, o, H, W) + run(f, o, H, W, 1) + run(f, pag, H, W)), flush=True)
