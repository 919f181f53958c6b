"""Checks built on the oracle (TEST INFRASTRUCTURE like the rest of oracle/: imported by tests/, by __graft_entry__.smoke() and by bench.py's verify leg only).

`warp_phase_check` -- the WarpRectilinear + Lanczos-4 path against the oracle with the ONE allowed cause of a difference spelled out and tested.

The reference's table builder (dng_warp_corr/dng_warp_rectilinear_coords.pyx:18-40) evaluates r**4 and r**6 through libm `powf`, the kernels through exactly
rounded products: a table coordinate may differ in its last two bits (tests/test_gpu_parity.py::test_warp_table: <= 2 ULP against the compiled reference, G7).
cv2.remap (chan_distortion_corr.py:94-97) quantises a coordinate to 1/32 px, `cvRound(x * 32)`, so such a difference is invisible unless the coordinate lies
within 2 ULP of a rounding boundary of that quantisation (or of the np.clip edge, :92-93), where the pixel takes the neighbouring Lanczos phase.  The check:

  * the set B of output pixels whose oracle coordinate, moved by -2 ULP and by +2 ULP (then clipped), quantises to two different 1/32-px values;
  * EVERY value that differs from the oracle lies in B (outside B the result is bit-identical);
  * every differing value IS the oracle's Lanczos-4 interpolation at the neighbouring phase: bit-identical to the oracle's remap of the same source at one of
    the four coordinates (x -/+ 2 ULP, y -/+ 2 ULP) -- B grows with the coordinates (a float32 ULP at x = 8192 is 1/32 of a quantisation step: on a 100 MP
    frame an eighth of the pixels lie in B), so membership alone would be a weak test there; this one is not;
  * the differing values are few (< 0.5 % of the frame: a table entry differs in ~2 % of the pixels and then crosses a boundary only now and then).

A bug that moved any pixel by anything but one Lanczos phase at a boundary fails, which the old `mean(diff > 0) < 2e-3 and max < 5e-2` bar did not guarantee.
"""
from __future__ import annotations

import numpy as np

from . import oracle


def _quant(v: np.ndarray, hi: float) -> np.ndarray:
    """cv2.remap's coordinate quantisation after chan_distortion_corr.py's np.clip: cvRound(clip(v, 0, hi) * 32), round half to even, in float32."""
    return np.rint(np.clip(v, np.float32(0), np.float32(hi)) * np.float32(32)).astype(np.int64)


def _near_boundary(v: np.ndarray, hi: float, ulps: int = 2) -> np.ndarray:
    lo_, hi_ = v, v
    for _ in range(ulps):
        lo_ = np.nextafter(lo_, np.float32(-np.inf))
        hi_ = np.nextafter(hi_, np.float32(np.inf))
    return _quant(lo_, hi) != _quant(hi_, hi)


def _bits_differ(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
    return (a.view(np.int32) != b.view(np.int32)) & ~(np.isnan(a) & np.isnan(b)) & ~((a == 0) & (b == 0))


def warp_phase_check(got, src, coeffs, centre, scale: float = 1.0, rows=None, prior=None, expected=None) -> dict:
    """Compare `got` (the warped image, (H, W, C), or just its rows [y0, y1) when `rows` is given and got has y1 - y0 rows) with the oracle's warp of `src`
    ((H, W, C), the image before the warp) under coefficient rows `coeffs` (C x 6), centre (cx, cy) and `scale`; `prior`: the (H, W, C, 2) seed of
    stack_warp_prior, or None.  `expected`: compare with this array (e.g. a fixture produced by the reference's own orchestration) instead of the oracle's remap.
    Returns the counts; raises AssertionError with the classification when a differing pixel lies outside the phase-boundary set."""
    src = np.asarray(src)
    H, W, C = src.shape
    y0, y1 = (0, H) if rows is None else (int(rows[0]), int(rows[1]))
    got = np.asarray(got)
    assert got.shape == (y1 - y0, W, C), (got.shape, (y1 - y0, W, C))
    coeffs = np.asarray(coeffs, dtype=np.float64).reshape(C, 6)
    n_diff = n_boundary = n_unexplained = n_not_neighbour = 0
    worst = 0.0
    for c in range(C):
        k = [float(np.float32(v)) for v in coeffs[c]]                 # (Cython takes the doubles as C floats, pyx:67-68)
        seed = None if prior is None else np.ascontiguousarray(prior[:, :, c, :], dtype=np.float32)
        t = oracle.warp_table(k[0], k[1], k[2], k[3], k[4], k[5], W, H, float(np.float32(centre[0])), float(np.float32(centre[1])), scale, seed=seed, rows=(y0, y1))
        x, y = np.ascontiguousarray(t[..., 0]), np.ascontiguousarray(t[..., 1])
        boundary = _near_boundary(x, W - 1) | _near_boundary(y, H - 1)
        if expected is None:
            ref = oracle.remap_lanczos4(np.ascontiguousarray(src[..., c], dtype=np.float32), np.clip(x, 0, W - 1), np.clip(y, 0, H - 1))
        else:
            ref = np.asarray(expected)[y0:y1, :, c] if np.asarray(expected).shape[0] == H else np.asarray(expected)[..., c]
        d = _bits_differ(got[..., c], ref)
        n_diff += int(d.sum()); n_boundary += int(boundary.sum()); n_unexplained += int((d & ~boundary).sum())
        if d.any():
            worst = max(worst, float(np.nanmax(np.abs(got[..., c][d].astype(np.float64) - np.asarray(ref)[d]))))
            # the differing values against the oracle's interpolation at the four coordinates 2 ULP around the oracle's own
            plane = np.ascontiguousarray(src[..., c], dtype=np.float32)
            xs, ys, gv = x[d], y[d], got[..., c][d]
            ok = np.zeros(gv.shape, bool)
            for sx in (-np.inf, np.inf):
                for sy in (-np.inf, np.inf):
                    xv, yv = xs, ys
                    for _ in range(2):
                        xv = np.nextafter(xv, np.float32(sx)); yv = np.nextafter(yv, np.float32(sy))
                    alt = oracle.remap_lanczos4(plane, np.clip(xv, 0, W - 1), np.clip(yv, 0, H - 1))
                    ok |= ~_bits_differ(gv, alt)
            n_not_neighbour += int((~ok).sum())
    n_px = (y1 - y0) * W * C
    out = {"values": n_px, "differing": n_diff, "at_phase_boundary": n_boundary, "differing_outside_boundary_set": n_unexplained,
           "differing_not_a_neighbouring_phase": n_not_neighbour, "frac_boundary": n_boundary / max(1, n_px), "frac_differing": n_diff / max(1, n_px),
           "max_abs_diff_in_boundary_set": worst, "bit_exact_outside_boundary_set": n_unexplained == 0}
    assert n_unexplained == 0, f"warp: {n_unexplained} values differ from the oracle although their coordinates lie more than 2 ULP from every 1/32-px boundary: {out}"
    assert n_not_neighbour == 0, f"warp: {n_not_neighbour} differing values are not the oracle's interpolation at a coordinate within 2 ULP either: {out}"
    assert out["frac_differing"] < 5e-3, f"warp: {out['frac_differing']:.3g} of the values took a neighbouring phase (expected < 0.5 %): {out}"
    return out
