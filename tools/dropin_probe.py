"""Where the host-buffer (PCIe-inclusive) fused call spends its time at 24 MP, per transfer mechanism (round 5):

    python tools/dropin_probe.py            # runs every mode below in a child process (the switches are read once per process)

Modes: the event-chained band pipeline (default for page-locked results), the helper-thread pipeline (PYSP_HOST_PIPE=thread: round 4's
form), the device-to-host leg by a copy kernel (PYSP_D2H_KERNEL=1), and the runtime without its DMA engines (HSA_ENABLE_SDMA=0).
Per mode: input pageable / page-locked x result pageable / page-locked, 30 calls each, all times printed sorted (a bimodal
distribution shows), plus one traced call (PYSP_BAND_TRACE=1: host clock per band).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = [
    ("default (page-locked result: the mosaic is registered for the call, event-chained bands; pageable result: helper thread)", {}),
    ("no registration (round 4 + upload stream)", {"PYSP_H2D_REGISTER": "0"}),
    ("helper thread + blocking hipMemcpy D2H", {"PYSP_HOST_PIPE": "thread", "PYSP_D2H_SYNC": "1"}),
    ("helper thread + hipHostRegister of the input", {"PYSP_HOST_PIPE": "thread", "PYSP_H2D_REGISTER": "1"}),
    ("events for every source", {"PYSP_HOST_PIPE": "events"}),
    ("helper thread (round 4 + upload stream)", {"PYSP_HOST_PIPE": "thread"}),
    ("helper thread, 512-row bands", {"PYSP_HOST_PIPE": "thread", "PYSP_BAND_ROWS": "512"}),
    ("helper thread, 128-row bands", {"PYSP_HOST_PIPE": "thread", "PYSP_BAND_ROWS": "128"}),
]


def child() -> None:
    import time
    import numpy as np
    sys.path.insert(0, ROOT)
    from pysp_amd import _lib
    from pysp_amd.colorize.transform import final_matrix
    from pysp_amd.synth import default_wb, rggb_frame
    H, W = 4000, 6000
    bay = rggb_frame(H, W, 1000)
    wbobj = default_wb()
    L = _lib.lib(); ctx = _lib.default_context()
    wb = _lib.wb3(wbobj.get_reciprocal_multipliers()); M = _lib.mat9(final_matrix(wbobj.get_matrix()))
    pin_in = _lib.empty_f32((H, W)); pin_in[:] = bay
    pin_out = _lib.empty_f32((H, W, 3))
    pag_out = np.empty((H, W, 3), np.float32); pag_out[:] = 0
    u16 = np.clip(np.round(bay * 15359.0 + 512.0), 0, 16383).astype(np.uint16)
    black = (_lib.ctypes.c_float * 4)(512.0, 512.0, 512.0, 512.0); sat = (_lib.ctypes.c_float * 4)(15871.0, 15871.0, 15871.0, 15871.0)
    ref = None
    for name, src, dst in (("pageable in, pinned out ", bay, pin_out), ("pageable in, pageable out", bay, pag_out), ("pinned in,   pinned out ", pin_in, pin_out)):
        ts = []
        for it in range(31):
            t0 = time.perf_counter()
            _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(src), H, W, wb, M, 2, 0, 1, 0, _lib.ptr(dst)))
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[1:])
        if ref is None:
            ref = dst.copy()
        else:
            assert np.array_equal(ref, dst), "results differ between transfer modes"
        print("  %s: min %.2f  median %.2f  max %.2f | %s" % (name, ts[0], ts[len(ts) // 2], ts[-1], " ".join("%.2f" % t for t in ts)), flush=True)
    # a NEW pageable mosaic per call (what a stream of frames looks like: the registration never finds pages it has seen before)
    ts = []
    for it in range(13):
        fresh = np.empty_like(bay); fresh[:] = bay
        t0 = time.perf_counter()
        _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(fresh), H, W, wb, M, 2, 0, 1, 0, _lib.ptr(pin_out)))
        ts.append((time.perf_counter() - t0) * 1e3)
        del fresh
    ts = sorted(ts[1:])
    assert np.array_equal(ref, pin_out)
    print("  fresh pageable mosaic per call, pinned out: min %.2f  median %.2f  max %.2f | %s" % (ts[0], ts[len(ts) // 2], ts[-1], " ".join("%.2f" % t for t in ts)), flush=True)
    ts = []
    for it in range(31):
        t0 = time.perf_counter()
        _lib.check(L.pysp_pipeline_u16_f32(ctx.handle, _lib.ptr(u16), H, W, black, sat, wb, M, 2, 0, 1, 2, _lib.ptr(pin_out)))
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[1:])
    print("  uint16 pageable in, pinned out: min %.2f  median %.2f  max %.2f | %s" % (ts[0], ts[len(ts) // 2], ts[-1], " ".join("%.2f" % t for t in ts)), flush=True)


def main() -> None:
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        return
    for name, env in MODES:
        e = dict(os.environ); e.update(env)
        print("== %s %s" % (name, env), flush=True)
        subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=e)
    # one traced call per pipeline form
    for name, env in MODES[:2]:
        e = dict(os.environ); e.update(env); e["PYSP_BAND_TRACE"] = "1"
        print("== trace: %s" % name, flush=True)
        code = ("import sys; sys.path.insert(0, %r)\n"
                "import numpy as np\nfrom pysp_amd import _lib\nfrom pysp_amd.colorize.transform import final_matrix\nfrom pysp_amd.synth import default_wb, rggb_frame\n"
                "H, W = 4000, 6000\nbay = rggb_frame(H, W, 1000); wbobj = default_wb(); L = _lib.lib(); ctx = _lib.default_context()\n"
                "wb = _lib.wb3(wbobj.get_reciprocal_multipliers()); M = _lib.mat9(final_matrix(wbobj.get_matrix())); out = _lib.empty_f32((H, W, 3))\n"
                "for it in range(6): _lib.check(L.pysp_pipeline_srgb_f32(ctx.handle, _lib.ptr(bay), H, W, wb, M, 2, 0, 1, 0, _lib.ptr(out)))\n" % ROOT)
        subprocess.call([sys.executable, "-c", code], env=e)


if __name__ == "__main__":
    main()
