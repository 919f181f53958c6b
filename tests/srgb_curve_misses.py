"""Lists the float32 inputs of [0, 1] on which lin_srgb_to_srgb of the loaded library differs from the oracle (test infrastructure; used on the
variants of devmath.h::srgb_pow_5_12, PYSP_HIP_LIB=tools/scratch/<variant>.so python tests/srgb_curve_misses.py)."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from pysp_amd import _lib
from oracle import oracle as orc
L, ctx = _lib.lib(), _lib.Context(0)
ctx.set_stream(int(torch.cuda.current_stream().cuda_stream))
chunk = 1 << 26; hi = 0x3F800000 + 4096
out = torch.empty(chunk, dtype=torch.float32, device="cuda")
for lo in range(0, hi, chunk):
    m = min(chunk, hi - lo)
    x = torch.arange(lo, lo + m, dtype=torch.int32, device="cuda").view(torch.float32)
    _lib.check(L.pysp_lin_srgb_to_srgb_dev(ctx.handle, ctypes.c_void_p(x.data_ptr()), m, ctypes.c_void_p(out.data_ptr())))
    ctx.sync()
    got = out[:m].cpu().numpy(); xs = x.cpu().numpy(); ref = orc.lin_srgb_to_srgb(xs)
    bad = np.nonzero(got != ref)[0]
    for i in bad[:50]:
        print(hex(lo + int(i)), repr(float(xs[i])), got[i].view(np.int32) - ref[i].view(np.int32), repr(float(got[i])), repr(float(ref[i])), repr(float(xs[i]) ** float(np.float32(0.41666666))  * 1.055 - 0.055))
