"""How the automatic Lab-layout policy of the AHD select kernel (pysp_ctx_set_lab_layout(-1), api.cpp) copes with streams that ALTERNATE content
(VERDICT r4 item 5): `period` scene frames, then `period` pure-noise frames, and so on, on one context, the host enqueueing freely ahead of the GPU.

    python tools/lab_layout_alternation.py [--frames 1200] [--size 4000x6000]      -> one table row per period: ms per frame for auto / packed / planes

The packed layout is 1-2 % faster on ordinary content and ~10 % slower on colour noise (every wave redoes its votes in float arithmetic); the planes layout
costs the same on both.  The bar: for every period the automatic policy stays within 5 % of the better FIXED layout.
"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(size=(4000, 6000), frames=1200, periods=(1, 4, 16, 64, 300)):
    import numpy as np
    import torch
    from pysp_amd import _lib
    from pysp_amd.colorize.transform import final_matrix
    from pysp_amd.synth import default_wb, random_frame, rggb_frame
    H, W = size
    wbobj = default_wb()
    wb = _lib.wb3(wbobj.get_reciprocal_multipliers()); M = _lib.mat9(final_matrix(wbobj.get_matrix()))
    L = _lib.lib()
    scene = [torch.from_numpy(rggb_frame(H, W, 1000 + i)).cuda() for i in range(2)]
    noise = [torch.from_numpy(random_frame(H, W, 10 + i)).cuda() for i in range(2)]
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    rows = []
    for period in periods:
        row = {"period": period}
        for layout in ("auto", "packed", "planes"):
            ctx = _lib.Context(0)
            ctx.set_kernel_timing(0)
            ctx.set_lab_layout(layout)

            def run(n):
                for i in range(n):
                    src = (scene if (i // period) % 2 == 0 else noise)[i % 2]
                    _lib.check(L.pysp_pipeline_dev(ctx.handle, ctypes.c_void_p(src.data_ptr()), H, W, wb, M, 2, 0, 1, 2, ctypes.c_void_p(out.data_ptr())))
                    if i % 256 == 255:
                        ctx.sync()                      # keep the launch queue bounded (the policy still sees the host hundreds of frames ahead)
            run(max(64, 2 * period))                    # warm up: clocks, and the policy's first decisions
            ctx.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(frames)
            ctx.sync()
            row[layout] = (time.perf_counter() - t0) / frames * 1e3
            del ctx
        row["auto_over_best_fixed"] = row["auto"] / min(row["packed"], row["planes"])
        rows.append(row)
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1200)
    ap.add_argument("--size", default="4000x6000")
    a = ap.parse_args()
    H, W = (int(v) for v in a.size.lower().split("x"))
    rows = measure((H, W), a.frames)
    print("# scene / noise alternation, %dx%d, %d frames per cell, ms per frame (AHD + colour tail, device resident)" % (H, W, a.frames))
    print("| period | auto | packed | planes | auto / better fixed layout |\n|---|---|---|---|---|")
    for r in rows:
        print("| %d | %.4f | %.4f | %.4f | %.3f |" % (r["period"], r["auto"], r["packed"], r["planes"], r["auto_over_best_fixed"]))
    print(json.dumps(rows))
