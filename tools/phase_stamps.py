#!/usr/bin/env python3
"""Per-phase shader-clock cycles of k_ahd_select from in-kernel stamps (diagnostic build).

    bash tools/build_variant.sh stamps -DAHD_STAMPS
    PYSP_HIP_LIB=$PWD/tools/scratch/stamps.so python tools/phase_stamps.py > profiles/r3_phase_stamps_k_ahd_select.csv      (on the GPU box)

The -DAHD_STAMPS build writes s_memtime at every phase boundary of k_ahd_select (before and after each barrier) into a device array that no kernel
reads; this script runs the benchmark's call (24 MP AHD, one median stage, sRGB tail) a few times, clears the array, runs it once more and reduces
the stamps: per interval the mean / median / p90 cycles over all waves, its share of a wave's lifetime, and the lifetime itself.  The stamps cost a few
per cent (one s_memtime + one store per wave and boundary); the product build contains none of this.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: one HIP runtime in the process)
from pysp_amd import _lib  # noqa: E402
from pysp_amd.colorize.transform import final_matrix  # noqa: E402
from pysp_amd.synth import default_wb, rggb_frame  # noqa: E402

NAMES = ["P0 tile load + plane store", "barrier", "P1(H) green planes", "barrier", "P2(H) resample + CCM + Lab", "barrier", "P3(H) votes + P1(V) planes", "barrier",
         "P2(V) resample + CCM + Lab", "barrier", "P3(V) votes + vote map", "barrier", "P4 box + select + store"]


def main() -> None:
    H, W = 4000, 6000
    stream = "--stream" in sys.argv          # the streaming form of the select kernel (round 5): stamps accumulate per wave over its passes
    L = _lib.lib()
    if not hasattr(L, "pysp_debug_ahd_stamps"):
        raise SystemExit("this library has no stamps: build with -DAHD_STAMPS and point PYSP_HIP_LIB at it")
    L.pysp_debug_ahd_stamps.restype = ctypes.c_int
    L.pysp_debug_ahd_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    ctx = _lib.Context(0)
    ctx.set_lab_layout("packed")
    ctx.set_select_form("stream" if stream else "tile")
    wbobj = default_wb()
    wb, M = _lib.wb3(wbobj.get_reciprocal_multipliers()), _lib.mat9(final_matrix(wbobj.get_matrix()))
    frame = torch.from_numpy(rggb_frame(H, W, 1000)).cuda()
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")

    def step():
        _lib.check(L.pysp_pipeline_dev(ctx.handle, ctypes.c_void_p(frame.data_ptr()), H, W, wb, M, 2, 0, 1, 2, ctypes.c_void_p(out.data_ptr())))
    for _ in range(300):
        step()
    ctx.sync()
    nst = L.pysp_debug_ahd_stamps(None, 0, 1)           # clear
    step()
    ctx.sync()
    n_waves = (1 << 17) if stream else ((W // 2 + 13) // 14) * ((H // 2 + 13) // 14) * 4
    buf = np.zeros(n_waves * nst, dtype=np.uint64)
    assert L.pysp_debug_ahd_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size, 0) == nst
    if stream:
        # per wave: sums over its passes of (clock at boundary i - clock at the pass's first stamp), slot 15 = passes -> mean cycles per pass
        raw = buf.reshape(n_waves, nst).astype(np.float64)
        raw = raw[raw[:, 15] > 0]
        passes = raw[:, 15]
        st = (raw[:, :14] / passes[:, None])
        print(f'"streaming form: waves {len(st)}, passes per wave mean {passes.mean():.2f} / min {passes.min():.0f} / max {passes.max():.0f}; figures are mean cycles PER PASS; the first interval includes the wait at the pass-top barrier",,,,')
        ok = np.ones(len(st), bool)
    else:
        st = buf.reshape(n_waves, nst)[:, :14].astype(np.int64)
        ok = (st > 0).all(axis=1)
        st = st[ok]
    d = np.diff(st, axis=1)                                # 13 intervals
    life = st[:, 13] - st[:, 0]
    print("interval,mean_cycles,median_cycles,p90_cycles,share_of_wave_lifetime")
    for i, n in enumerate(NAMES):
        print(f'"{n}",{d[:, i].mean():.0f},{np.median(d[:, i]):.0f},{np.percentile(d[:, i], 90):.0f},{d[:, i].mean() / life.mean():.4f}')
    bar = d[:, 1::2].sum(axis=1)
    print(f'"all six barrier waits",{bar.mean():.0f},{np.median(bar):.0f},{np.percentile(bar, 90):.0f},{bar.mean() / life.mean():.4f}')
    print(f'"wave lifetime (first to last stamp)",{life.mean():.0f},{np.median(life):.0f},{np.percentile(life, 90):.0f},1.0')
    print(f'"waves with all stamps",{len(st)},of,{n_waves},')


if __name__ == "__main__":
    main()
