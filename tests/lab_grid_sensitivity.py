#!/usr/bin/env python3
"""What does an LSB-level disagreement with cv2's Lab table cost?  TEST INFRASTRUCTURE (uses the CPU oracle).

Lab mode 1 restates OpenCV 4.10's RGB2Lab LUT path from memory.  Its 33^3 x 3 int16 table is built here with a float64
`pow` / `cbrt` rounded once to float32, where OpenCV builds it in softfloat float32 -- individual entries may differ by one unit.  Since round 4 the
table is DATA (pysp_ctx_set_lab_lut / oracle.set_cv410_lut), so this script can measure what such a difference does to AHD, on the benchmark's
synthetic 24 MP frame:

  * the built-in table with +-1 LSB on 1 % / 10 % / 50 % of its entries (seeded),
  * the table built with float32 powf / cbrtf (NumPy's float32 loops) instead of the float64 functions,
  * for scale: lab mode 0 (closed form) against mode 1 -- the 2.73 % of profiles/r2_lab_flip_rate_24mp.jsonl.

Reported per variant: table entries changed, homogeneity counts that differ, H/V decisions that flip, demosaiced pixels (no median stage) changed,
sRGB pixels changed after one median stage (any / by more than 1/255 / max).  Output: one JSON line per variant (profiles/r4_lab_grid_sensitivity.jsonl).

    python tests/lab_grid_sensitivity.py [--H 4000 --W 6000]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def grid_float32_functions():
    """The table of oracle/cv2_restated.py::cv410_lab_lut with the gamma power and the cube root evaluated by float32 library functions."""
    from oracle.cv2_restated import _fmaf
    f32 = np.float32
    white = (0.950456, 1.0, 1.088754)
    xyz = (0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227)
    C = [f32((1.0 / white[i // 3] if i // 3 != 1 else 1.0) * xyz[i]) for i in range(9)]
    g = np.arange(33, dtype=np.float32) / f32(32)
    gam = np.where(g <= f32(0.04045), g / f32(12.92), np.power((g + f32(0.055)) / f32(1.055), f32(2.4))).astype(np.float32)
    B, G, R = np.meshgrid(gam, gam, gam, indexing="ij")
    X = R * C[0] + G * C[1] + B * C[2]
    Y = R * C[3] + G * C[4] + B * C[5]
    Z = R * C[6] + G * C[7] + B * C[8]
    lthresh, lscale, lbias = f32(216) / f32(24389), f32(841) / f32(108), f32(16) / f32(116)

    def fxyz(t):
        return np.where(t > lthresh, np.cbrt(t.astype(np.float32)), _fmaf(t, lscale, lbias)).astype(np.float32)
    FX, FY, FZ = fxyz(X), fxyz(Y), fxyz(Z)
    L = np.where(Y > lthresh, f32(116) * FY - f32(16), (f32(24389) / f32(27)) * Y).astype(np.float32)
    a, b = f32(500) * (FX - FY), f32(200) * (FY - FZ)
    base = f32(16384)
    lut = np.stack([np.rint(base * L / f32(100)), np.rint(base * (a + f32(128)) / f32(256)), np.rint(base * (b + f32(128)) / f32(256))], axis=-1)
    return lut.astype(np.int16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=4000)
    ap.add_argument("--W", type=int, default=6000)
    args = ap.parse_args()
    from oracle import oracle
    from pysp_amd.synth import D65_XY, NEUTRAL_MULTIPLIERS, XYZ_TO_CAM, rggb_frame
    wb = (1.0 / NEUTRAL_MULTIPLIERS).astype(np.float32)
    M = oracle.final_matrix(XYZ_TO_CAM, oracle.xy_to_XYZ(D65_XY))
    H, W = args.H, args.W
    bay = rggb_frame(H, W, 1000)
    base_grid = oracle.cv410_lut()

    def run(mode, grid):
        oracle.set_lab_mode(mode)
        oracle.set_cv410_lut(grid)
        try:
            out0, t = oracle.demosaic_ahd(bay, wb, M, False, 0, taps=True)
            take_h = oracle.box3(t["map_h"]) < oracle.box3(t["map_v"])
            res = dict(map_h=t["map_h"], map_v=t["map_v"], take_h=take_h, out0=out0, srgb=oracle.pipeline_srgb(bay, wb, M, 2, False, 1, False))
        finally:
            oracle.set_lab_mode(oracle.DEFAULT_LAB_MODE)
            oracle.set_cv410_lut(None)
        return res

    ref = run(1, None)
    variants = []
    for frac in (0.01, 0.10, 0.50):
        rng = np.random.default_rng(int(frac * 1000))
        hit = rng.random(base_grid.shape) < frac
        step = np.where(rng.random(base_grid.shape) < 0.5, -1, 1)
        g = (base_grid.astype(np.int32) + hit * step).clip(0, 32767).astype(np.int16)
        variants.append((f"built-in table, +-1 LSB on {frac:.0%} of the entries", 1, g))
    variants.append(("table built with float32 powf / cbrtf", 1, grid_float32_functions()))
    variants.append(("lab mode 0 (closed form) for scale", 0, None))
    for name, mode, grid in variants:
        got = run(mode, grid)
        d0 = np.abs(got["out0"] - ref["out0"])
        ds = np.abs(got["srgb"] - ref["srgb"])
        rep = {"variant": name, "H": H, "W": W, "frame": "synthetic scene (SURVEY 8d, seed 1000)",
               "table_entries_changed": None if grid is None else int((grid != base_grid).sum()),
               "table_entries_changed_frac": None if grid is None else float((grid != base_grid).mean()),
               "table_max_abs_change": None if grid is None else int(np.abs(grid.astype(np.int32) - base_grid).max()),
               "homogeneity_counts_differ_h": float(np.mean(got["map_h"] != ref["map_h"])),
               "homogeneity_counts_differ_v": float(np.mean(got["map_v"] != ref["map_v"])),
               "decision_flip_rate": float(np.mean(got["take_h"] != ref["take_h"])),
               "demosaic_stages0_pixels_changed": float(np.mean(d0.max(axis=-1) > 0)), "demosaic_stages0_max_abs_delta": float(d0.max()),
               "srgb_stages1_pixels_changed": float(np.mean(ds.max(axis=-1) > 0)),
               "srgb_stages1_pixels_changed_by_more_than_1_255": float(np.mean(ds.max(axis=-1) > 1 / 255)),
               "srgb_stages1_max_abs_delta": float(ds.max()), "srgb_stages1_mean_abs_delta": float(ds.mean())}
        print(json.dumps(rep), flush=True)


if __name__ == "__main__":
    main()
