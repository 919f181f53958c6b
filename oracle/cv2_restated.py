"""NumPy restatement of the eight OpenCV calls on pySP's hot path.  TEST INFRASTRUCTURE ONLY.

opencv_python==4.10.0.84 (reference requirements.txt:5) is not installed in this image and cannot be
installed, so the arithmetic inside these calls is PARITY UNPINNED.
This module states the semantics the whole build follows (border rule, tap values, evaluation
order), in plain NumPy float32, independently of oracle/pysp_oracle.c, so that
  * tests can cross-check the C oracle's primitives against a second implementation, and
  * tests/golden/gen_golden.py can run the reference's UNCHANGED orchestration (ahd.py,
    edge_assisted_gaussian.py, fast_resize.py) with this module standing in for `cv2`
    (fixtures produced that way carry "cv2_restated": true).
cvtColor(RGB2LAB) exists twice, both in NumPy with no call into the C oracle:
  * LAB_MODE = "closed_form": sRGB decode + D65 CIELab with table-driven pow / cbrt, the arithmetic of lab mode 0 of
    the product kernels and of oracle/pysp_oracle.c::rgb2lab_px, bit for bit (tables rebuilt here from libm);
  * LAB_MODE = "cv410_lut" (default since round 2, lab mode 1): OpenCV 4.10's default float32 path for sRGB input as published in
    modules/imgproc/src/color_lab.cpp (RGB2Lab_f with useInterpolation): clip, cvRound(v * 2^14), 33^3 int16
    LUT of closed-form Lab at the grid points, fixed-point trilinear interpolation, rescale.  Restated FROM
    MEMORY of that source (unpinnable here); tests/lab_flip_rate.py reports how many AHD decisions differ
    between the two, tools/gen_cv2_goldens.py records the real thing wherever cv2 is installed.

Call sites (relative to /root/reference): ahd.py:58,62,64,77-80,120-121,133-134,151;
edge_assisted_gaussian.py:86-87,141,143,156,170,184; fast_resize.py:28-29,39;
dng_warp_corr/chan_distortion_corr.py:94-97; corr_ca/ca_removal.py:96-127 (remap INTER_LINEAR).
"""
from __future__ import annotations

import numpy as np

BORDER_CONSTANT = 0
BORDER_REPLICATE = 1
BORDER_REFLECT = 2
BORDER_REFLECT_101 = 4
BORDER_DEFAULT = 4
COLOR_RGB2LAB = 45
INTER_LINEAR = 1
INTER_LANCZOS4 = 4

_F = np.float32
GK0 = _F(0.45186276)  # exp(0)/(1+2exp(-1/2))
GK1 = _F(0.27406862)  # exp(-1/2)/(1+2exp(-1/2))


def _pad(a: np.ndarray, t: int, b: int, l: int, r: int, mode: str) -> np.ndarray:
    pw = ((t, b), (l, r)) + ((0, 0),) * (a.ndim - 2)
    return np.pad(a, pw, mode=mode)


def copyMakeBorder(src, top, bottom, left, right, borderType):
    if borderType == BORDER_REFLECT:
        return _pad(src, top, bottom, left, right, "symmetric")   # fedcba|abcdefgh|hgfedcb
    if borderType == BORDER_REFLECT_101:
        return _pad(src, top, bottom, left, right, "reflect")     # gfedcb|abcdefgh|gfedcba
    if borderType == BORDER_REPLICATE:
        return _pad(src, top, bottom, left, right, "edge")
    raise NotImplementedError(borderType)


def GaussianBlur(src, ksize, sigmaX):
    assert tuple(ksize) == (3, 3) and float(sigmaX) == 1.0 and src.dtype == np.float32 and src.ndim == 2
    p = _pad(src, 0, 0, 1, 1, "reflect")
    rowp = p[:, 1:-1] * GK0 + (p[:, :-2] + p[:, 2:]) * GK1
    q = _pad(rowp, 1, 1, 0, 0, "reflect")
    return q[1:-1] * GK0 + (q[:-2] + q[2:]) * GK1


def filter2D(src, ddepth, kernel):
    assert ddepth == -1 and src.dtype == np.float32 and src.ndim == 2 and kernel.shape == (3, 3)
    kf = kernel.astype(np.float32)
    p = _pad(src, 1, 1, 1, 1, "reflect")
    h, w = src.shape
    acc = np.zeros_like(src)
    for a in range(3):
        for b in range(3):
            if kf[a, b] != 0:
                acc = acc + kf[a, b] * p[a:a + h, b:b + w]
    return acc


# ---- cvtColor(COLOR_RGB2LAB), float32 (ahd.py:58,62) ------------------------------------------------------------
LAB_MODE = "cv410_lut"            # or "closed_form"; tests/golden/gen_golden.py runs with the default


def _fmaf(a, b, c):
    """float32 fused multiply-add, exactly rounded, elementwise: the product of two float32 is exact in float64; the
    float64 sum is rounded once more on the way to float32, which is only wrong when it lands exactly on a float32
    midpoint while the discarded float64 rounding error says the true value lies beside it -- fixed up with TwoSum."""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32); c = np.asarray(c, np.float32)
    with np.errstate(all="ignore"):
        p = a.astype(np.float64) * b.astype(np.float64)
        c64 = c.astype(np.float64)
        s = p + c64
        bb = s - p
        err = (p - (s - bb)) + (c64 - bb)
        r = s.astype(np.float32)
        r64 = r.astype(np.float64)
        up = np.nextafter(r, np.float32(np.inf)).astype(np.float64)
        dn = np.nextafter(r, np.float32(-np.inf)).astype(np.float64)
        tie_up = (s > r64) & ((s - r64) == (up - s)) & (err > 0)      # rounded down to even, truth is above the midpoint
        tie_dn = (s < r64) & ((r64 - s) == (s - dn)) & (err < 0)      # rounded up to even, truth is below the midpoint
        ok = np.isfinite(s)
        r = np.where(ok & tie_up, up.astype(np.float32), r)
        r = np.where(ok & tie_dn, dn.astype(np.float32), r)
    return r.astype(np.float32)


# sRGB(D65) -> XYZ rows divided by the D65 white (0.950456, 1, 1.088754), float32
_LAB_C = [_F(v) for v in (0.43395275, 0.37621942, 0.18982783, 0.212671, 0.71516, 0.072169, 0.017757915, 0.109476522, 0.87276554)]
_LAB_DEC = (6, -5, 5 * 64 + 1)     # (mantissa bits per octave index, lowest exponent, segments): v in [2^-5, 1]
_LAB_CB = (5, -7, 8 * 32 + 1)      # t in [2^-7, 2)
_lab_tabs = {}


def _libm_cbrt():
    import ctypes
    import ctypes.util
    m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    m.cbrt.restype = ctypes.c_double
    m.cbrt.argtypes = [ctypes.c_double]
    return m.cbrt


def lab_tables():
    """(dec, cb): float32 arrays (n, 4) = value, slope, curvature of the parabola through f at a segment's ends and
    midpoint, and the segment start; f = ((v + 0.055) / 1.055) ** 2.4 and the cube root, evaluated by libm in float64."""
    if not _lab_tabs:
        import math
        cbrt = _libm_cbrt()
        for key, (nb, loexp, n), fn in (("dec", _LAB_DEC, lambda v: math.pow((v + 0.055) / 1.055, 2.4)), ("cb", _LAB_CB, cbrt)):
            b0 = int(np.float32(2.0 ** loexp).view(np.int32))
            tab = np.zeros((n, 4), np.float32)
            for i in range(n):
                x0 = float(np.int32(b0 + (i << (23 - nb))).view(np.float32))
                x1 = float(np.int32(b0 + ((i + 1) << (23 - nb))).view(np.float32))
                h = x1 - x0
                g0, gm, g1 = fn(x0), fn(x0 + 0.5 * h), fn(x1)
                tab[i] = (g0, (-3.0 * g0 + 4.0 * gm - g1) / h, (2.0 * g0 - 4.0 * gm + 2.0 * g1) / (h * h), x0)
            _lab_tabs[key] = tab
    return _lab_tabs["dec"], _lab_tabs["cb"]


def _lab_lut(tab, spec, x):
    nb, loexp, n = spec
    s = 23 - nb
    bits = np.ascontiguousarray(x, np.float32).view(np.int32)
    b0 = int(np.float32(2.0 ** loexp).view(np.int32))
    idx = np.clip((bits.astype(np.int64) - b0) >> s, 0, n - 1)        # out-of-range arguments are discarded by the caller's select
    x0 = (bits & np.int32(~((1 << s) - 1))).view(np.float32)
    fr = x - x0
    return _fmaf(_fmaf(tab[idx, 2], fr, tab[idx, 1]), fr, tab[idx, 0])


def _lab_closed_form(src):
    dec, cb = lab_tables()
    with np.errstate(invalid="ignore"):
        v = np.where(src < 0, _F(0), np.where(src > 1, _F(1), src)).astype(np.float32)   # OpenCV clips the float input to [0,1]
        v = np.where(np.isnan(v), _F(0), v)                                               # min/max clip: a NaN comes out as 0
    lin = np.where(v <= _F(0.04045), v * _F(0.07739938), _lab_lut(dec, _LAB_DEC, v)).astype(np.float32)
    R, G, B = lin[..., 0], lin[..., 1], lin[..., 2]
    C = _LAB_C
    X = _fmaf(B, C[2], _fmaf(G, C[1], R * C[0]))
    Y = _fmaf(B, C[5], _fmaf(G, C[4], R * C[3]))
    Z = _fmaf(B, C[8], _fmaf(G, C[7], R * C[6]))

    def f(t):
        return np.where(t > _F(0.008856), _lab_lut(cb, _LAB_CB, t), _fmaf(_F(7.787), t, _F(0.13793103))).astype(np.float32)
    fx, fy, fz = f(X), f(Y), f(Z)
    L = np.where(Y > _F(0.008856), _fmaf(_F(116.0), fy, _F(-16.0)), _F(903.3) * Y)
    return np.stack([L, _F(500.0) * (fx - fy), _F(200.0) * (fy - fz)], axis=-1).astype(np.float32)


# OpenCV 4.10 color_lab.cpp constants: lab_lut_shift 5 -> LAB_LUT_DIM 33; lab_base_shift 14 -> LAB_BASE 16384;
# trilinear_shift = 8 - 5 + 1 = 4 -> TRILINEAR_BASE 16
_CV_LUT = {}


def cv410_lab_lut():
    """(33,33,33,3) int16 RGB2LabLUT as initLabTabs builds it (indexed [r][q][p] = [B][G][R] grid point): closed-form
    Lab in float32 (softfloat == IEEE float32 arithmetic) of applyGamma(p/32), scaled to 14 bits and rounded."""
    if "lut" not in _CV_LUT:
        import math
        f32 = np.float32
        white = (0.950456, 1.0, 1.088754)
        xyz = (0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227)
        C = [f32((1.0 / white[i // 3] if i // 3 != 1 else 1.0) * xyz[i]) for i in range(9)]    # softdouble product -> softfloat
        g = np.arange(33, dtype=np.float32) / f32(32)
        gam = np.array([f32(x) / f32(12.92) if f32(x) <= f32(0.04045) else f32(math.pow(float((f32(x) + f32(0.055)) / f32(1.055)), 2.4)) for x in g], np.float32)
        B, G, R = np.meshgrid(gam, gam, gam, indexing="ij")
        X = R * C[0] + G * C[1] + B * C[2]
        Y = R * C[3] + G * C[4] + B * C[5]
        Z = R * C[6] + G * C[7] + B * C[8]
        lthresh, lscale, lbias = f32(216) / f32(24389), f32(841) / f32(108), f32(16) / f32(116)
        cbrt = np.vectorize(_libm_cbrt(), otypes=[np.float64])

        def fxyz(t):
            return np.where(t > lthresh, cbrt(t.astype(np.float64)).astype(np.float32), _fmaf(t, lscale, lbias)).astype(np.float32)
        FX, FY, FZ = fxyz(X), fxyz(Y), fxyz(Z)
        L = np.where(Y > lthresh, f32(116) * FY - f32(16), (f32(24389) / f32(27)) * Y).astype(np.float32)
        a = f32(500) * (FX - FY)
        b = f32(200) * (FY - FZ)
        base = f32(16384)
        lut = np.stack([np.rint(base * L / f32(100)), np.rint(base * (a + f32(128)) / f32(256)), np.rint(base * (b + f32(128)) / f32(256))], axis=-1)
        _CV_LUT["lut"] = lut.astype(np.int16)
    return _CV_LUT["lut"]


def set_cv410_lab_lut(grid=None):
    """Inject a (33,33,33,3) int16 grid for the "cv410_lut" restatement (None: back to the built-in closed-form grid) -- the same switch as
    oracle.set_cv410_lut and the product's pysp_ctx_set_lab_lut: a table recorded from real cv2 at the grid nodes becomes data."""
    if grid is None:
        _CV_LUT.pop("injected", None)
        return
    g = np.ascontiguousarray(grid, dtype=np.int16)
    if g.shape != (33, 33, 33, 3) or (g < 0).any():
        raise ValueError("Lab grid must be (33, 33, 33, 3) int16 with entries in [0, 32767]")
    _CV_LUT["injected"] = g.copy()


def active_cv410_lab_lut():
    return _CV_LUT["injected"] if "injected" in _CV_LUT else cv410_lab_lut()


def _lab_cv410_lut(src):
    lut = active_cv410_lab_lut().astype(np.int32)
    with np.errstate(invalid="ignore"):
        v = np.minimum(np.maximum(np.where(np.isnan(src), _F(0), src), _F(0)), _F(1)).astype(np.float32)
    iv = np.rint(v * _F(16384)).astype(np.int32)                      # cvRound: round half to even
    cx, cy, cz = iv[..., 0], iv[..., 1], iv[..., 2]
    tx, ty, tz = cx >> 9, cy >> 9, cz >> 9                             # LUT cell origin (0..32)
    x, y, z = (cx >> 5) & 15, (cy >> 5) & 15, (cz >> 5) & 15           # position inside the cell, 1/16 steps
    acc = np.zeros(src.shape[:-1] + (3,), np.int32)
    for k in range(8):
        dx, dy, dz = k & 1, (k >> 1) & 1, (k >> 2) & 1
        w = (x if dx else 16 - x) * (y if dy else 16 - y) * (z if dz else 16 - z)
        corner = lut[np.minimum(tz + dz, 32), np.minimum(ty + dy, 32), np.minimum(tx + dx, 32)]
        acc += corner * w[..., None]
    acc = (acc + (1 << 11)) >> 12                                      # CV_DESCALE(., trilinear_shift * 3)
    out = np.empty(src.shape, np.float32)
    out[..., 0] = acc[..., 0].astype(np.float32) * _F(100.0 / 16384.0)
    out[..., 1] = acc[..., 1].astype(np.float32) * _F(256.0 / 16384.0) - _F(128.0)
    out[..., 2] = acc[..., 2].astype(np.float32) * _F(256.0 / 16384.0) - _F(128.0)
    return out


def cvtColor(src, code, mode=None):
    assert code == COLOR_RGB2LAB and src.dtype == np.float32 and src.ndim == 3 and src.shape[2] == 3
    mode = LAB_MODE if mode is None else mode
    if mode == "closed_form":
        return _lab_closed_form(src)
    if mode == "cv410_lut":
        return _lab_cv410_lut(src)
    raise ValueError(mode)


def blur(src, ksize):
    assert tuple(ksize) == (3, 3) and src.dtype == np.float32 and src.ndim == 2
    p = _pad(src, 1, 1, 1, 1, "reflect")
    h, w = src.shape
    acc = np.zeros_like(src)
    for a in range(3):
        for b in range(3):
            acc = acc + p[a:a + h, b:b + w]
    return acc * _F(0.11111111)


def medianBlur(src, ksize):
    assert ksize == 5 and src.dtype == np.float32 and src.ndim == 2
    p = _pad(src, 2, 2, 2, 2, "edge")
    h, w = src.shape
    st = np.stack([p[a:a + h, b:b + w] for a in range(5) for b in range(5)], axis=0)
    return np.ascontiguousarray(np.sort(st, axis=0)[12])


def resize(src, dsize, interpolation=INTER_LINEAR):
    h, w = src.shape[:2]
    W, H = dsize
    assert (H, W) == (2 * h, 2 * w) and src.dtype == np.float32 and interpolation == INTER_LINEAR

    def taps(n_out, n_in):
        f = (np.arange(n_out, dtype=np.float32) + _F(0.5)) * _F(0.5) - _F(0.5)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= n_in - 1
        f[hi] = 0; s[hi] = n_in - 1
        s1 = np.minimum(s + 1, n_in - 1)
        return s, s1, (_F(1.0) - f).astype(np.float32), f

    sx, sx1, a0, a1 = taps(W, w)
    sy, sy1, b0, b1 = taps(H, h)
    ex = (slice(None), slice(None)) + (None,) * (src.ndim - 2)
    hor = src[:, sx] * a0[None, :][ex] + src[:, sx1] * a1[None, :][ex]
    ey = (slice(None), None) + (None,) * (src.ndim - 2)
    return hor[sy] * b0[ey] + hor[sy1] * b1[ey]


def _lanczos4_tab() -> np.ndarray:
    s45 = 0.70710678118654752440084436210485
    cs = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]])
    tab = np.zeros((32, 8), np.float32)
    for i in range(32):
        x = np.float32(i) * np.float32(1.0 / 32.0)
        if x < np.float32(1.1920929e-07):
            tab[i, 3] = 1
            continue
        y0 = -(float(x) + 3) * np.pi * 0.25
        s0, c0 = np.sin(y0), np.cos(y0)
        c = np.zeros(8, np.float32)
        for k in range(8):
            y = -(float(x) + 3 - k) * np.pi * 0.25
            c[k] = np.float32((cs[k, 0] * s0 + cs[k, 1] * c0) / (y * y))
        ssum = np.float32(0)
        for k in range(8):
            ssum = np.float32(ssum + c[k])
        tab[i] = c * (np.float32(1.0) / ssum)
    return tab


def remap(src, mapx, mapy, interpolation):
    assert interpolation in (INTER_LANCZOS4, INTER_LINEAR) and src.dtype == np.float32 and src.ndim == 2
    H, W = src.shape
    sx = np.rint(mapx.astype(np.float32) * _F(32)).astype(np.int64)
    sy = np.rint(mapy.astype(np.float32) * _F(32)).astype(np.int64)
    if interpolation == INTER_LINEAR:
        # corr_ca/ca_removal.py:96-127.  Coordinates quantised to 1/32 px like every cv2.remap mode; the 2x2 weights are
        # float products of the 1-D taps (1 - f, f); BORDER_CONSTANT 0; sum in source order, left to right.
        ix, iy = sx >> 5, sy >> 5
        fx = (sx & 31).astype(np.float32) * _F(1.0 / 32.0)
        fy = (sy & 31).astype(np.float32) * _F(1.0 / 32.0)
        wx = (_F(1) - fx, fx)
        wy = (_F(1) - fy, fy)
        p = np.pad(src, ((1, 1), (1, 1)), mode="constant")
        out = None
        for r in range(2):
            yy = np.clip(iy + r + 1, 0, H + 1)
            for c in range(2):
                xx = np.clip(ix + c + 1, 0, W + 1)
                t = p[yy, xx] * (wy[r] * wx[c])
                out = t if out is None else out + t
        return out
    tab = _lanczos4_tab()
    ix, iy = (sx >> 5) - 3, (sy >> 5) - 3
    wx, wy = tab[sx & 31], tab[sy & 31]           # (H,W,8)
    p = np.pad(src, ((8, 8), (8, 8)), mode="constant")
    out = np.zeros((H, W), np.float32)
    for r in range(8):
        yy = np.clip(iy + r + 8, 0, H + 15)
        row = None
        for c in range(8):
            xx = np.clip(ix + c + 8, 0, W + 15)
            t = p[yy, xx] * (wy[..., r] * wx[..., c])
            row = t if row is None else row + t
        out = out + row
    return out
