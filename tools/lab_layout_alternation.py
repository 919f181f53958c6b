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
        layouts = ("auto", "packed", "planes")
        ctxs, pos, spent = {}, {}, {}
        for layout in layouts:
            ctxs[layout] = _lib.Context(0)
            ctxs[layout].set_kernel_timing(0)
            ctxs[layout].set_lab_layout(layout)
            pos[layout], spent[layout] = 0, 0.0

        def run(layout, n):
            ctx, i0 = ctxs[layout], pos[layout]
            for i in range(i0, i0 + n):                 # the stream goes on where this layout's last slice ended
                src = (scene if (i // period) % 2 == 0 else noise)[i % 2]
                _lib.check(L.pysp_pipeline_dev(ctx.handle, ctypes.c_void_p(src.data_ptr()), H, W, wb, M, 2, 0, 1, 2, ctypes.c_void_p(out.data_ptr())))
                if i % 256 == 255:
                    ctx.sync()                          # keep the launch queue bounded (the policy still sees the host hundreds of frames ahead)
            pos[layout] = i0 + n
        for layout in layouts:
            run(layout, max(64, 2 * period))            # warm up: clocks, and the policy's first decisions
            ctxs[layout].sync()
        torch.cuda.synchronize()
        # the three layouts take turns in slices (whole alternation cycles), so that clock and temperature drift of the box hits all of them alike
        slices = 3
        per = max(2 * period, (frames // slices) // (2 * period) * (2 * period))
        for rep in range(slices):
            for layout in (layouts if rep % 2 == 0 else layouts[::-1]):
                t0 = time.perf_counter()
                run(layout, per)
                ctxs[layout].sync()
                spent[layout] += time.perf_counter() - t0
        for layout in layouts:
            row[layout] = spent[layout] / (slices * per) * 1e3
        row["frames_per_layout"] = slices * per
        row["auto_over_best_fixed"] = row["auto"] / min(row["packed"], row["planes"])
        rows.append(row)
        ctxs.clear()
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1200)
    ap.add_argument("--size", default="4000x6000")
    a = ap.parse_args()
    H, W = (int(v) for v in a.size.lower().split("x"))
    rows = measure((H, W), a.frames)
    print("# scene / noise alternation, %dx%d, about %d frames per cell in three interleaved slices, ms per frame (AHD + colour tail, device resident)" % (H, W, a.frames))
    print("| period | auto | packed | planes | auto / better fixed layout |\n|---|---|---|---|---|")
    for r in rows:
        print("| %d | %.4f | %.4f | %.4f | %.3f |" % (r["period"], r["auto"], r["packed"], r["planes"], r["auto_over_best_fixed"]))
    print(json.dumps(rows))
