"""NumPy restatement of the eight OpenCV calls on pySP's hot path.  TEST INFRASTRUCTURE ONLY.

opencv_python==4.10.0.84 (reference requirements.txt:5) is not installed in this image and cannot be
installed, so the arithmetic inside these calls is PARITY UNPINNED.
This module states the semantics the whole build follows (border rule, tap values, evaluation
order), in plain NumPy float32, independently of oracle/pysp_oracle.c, so that
  * tests can cross-check the C oracle's primitives against a second implementation, and
  * tests/golden/gen_golden.py can run the reference's UNCHANGED orchestration (ahd.py,
    edge_assisted_gaussian.py, fast_resize.py) with this module standing in for `cv2`
    (fixtures produced that way carry "cv2_restated": true).
The one exception to "independent" is cvtColor(RGB2LAB): its pow/cbrt are bit-defined by the C
routine rgb2lab_px and are called through oracle.rgb2lab.

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


def cvtColor(src, code):
    assert code == COLOR_RGB2LAB and src.dtype == np.float32 and src.ndim == 3 and src.shape[2] == 3
    from . import oracle
    return oracle.rgb2lab(src)


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
