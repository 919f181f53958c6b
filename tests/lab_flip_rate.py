#!/usr/bin/env python3
"""How much does the choice of Lab restatement change AHD?  TEST INFRASTRUCTURE (uses the CPU oracle).

cv2.cvtColor(COLOR_RGB2LAB) (debayer/ahd.py:58,62 of the reference) is third-party arithmetic that cannot be pinned in this
image (no cv2).  Two restatements exist (oracle/cv2_restated.py, oracle/pysp_oracle.c):
    mode 0  closed form, table-driven pow / cbrt              -- round 1's metric, still selectable (pysp_ctx_set_lab_mode)
    mode 1  OpenCV 4.10's LUT (33^3, 14 bit) + trilinear path -- what the reference most likely runs; the default since round 2
AHD thresholds on these values with `<=` (debayer/ahd_homogeneity_cython.pyx:56-57), so this script measures, on the benchmark's
synthetic 24 MP frame and on pure noise: how many homogeneity counts differ, how many H/V decisions flip, and how far the
demosaiced / sRGB output moves.  Numbers quoted in DESIGN.md section 3.

    python tests/lab_flip_rate.py [--H 4000 --W 6000] [--hdr]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=4000)
    ap.add_argument("--W", type=int, default=6000)
    ap.add_argument("--hdr", action="store_true")
    args = ap.parse_args()
    from oracle import oracle
    from pysp_amd.synth import D65_XY, NEUTRAL_MULTIPLIERS, XYZ_TO_CAM, random_frame, rggb_frame
    wb = (1.0 / NEUTRAL_MULTIPLIERS).astype(np.float32)
    M = oracle.final_matrix(XYZ_TO_CAM, oracle.xy_to_XYZ(D65_XY))
    H, W = args.H, args.W
    frames = {"synthetic scene (SURVEY 8d, seed 1000)": rggb_frame(H, W, 1000, scale=3.0 if args.hdr else 1.0, clip_hi=not args.hdr),
              "pure noise (rng.random, seed 0)": random_frame(H, W, 0)}
    for name, bay in frames.items():
        res = {}
        for mode in (0, 1):
            oracle.set_lab_mode(mode)
            try:
                out0, t = oracle.demosaic_ahd(bay, wb, M, args.hdr, 0, taps=True)
                bh, bv = oracle.box3(t["map_h"]), oracle.box3(t["map_v"])
                res[mode] = dict(map_h=t["map_h"], map_v=t["map_v"], take_h=bh < bv, out0=out0,
                                 cand_differ=(t["r_h"] != t["r_v"]) | (t["g_h"] != t["g_v"]) | (t["b_h"] != t["b_v"]),
                                 srgb=oracle.pipeline_srgb(bay, wb, M, 2, args.hdr, 1, args.hdr))
            finally:
                oracle.set_lab_mode(oracle.DEFAULT_LAB_MODE)
        a, b = res[0], res[1]
        flips = a["take_h"] != b["take_h"]
        d0 = np.abs(a["out0"] - b["out0"])
        ds = np.abs(a["srgb"] - b["srgb"])
        report = {
            "frame": name, "H": H, "W": W, "hdr": args.hdr,
            "homogeneity_counts_differ_h": float(np.mean(a["map_h"] != b["map_h"])),
            "homogeneity_counts_differ_v": float(np.mean(a["map_v"] != b["map_v"])),
            "decision_flip_rate": float(np.mean(flips)),
            "decision_flip_rate_where_candidates_differ": float(np.mean(flips & a["cand_differ"])),
            "take_h_fraction_mode0": float(np.mean(a["take_h"])), "take_h_fraction_mode1": float(np.mean(b["take_h"])),
            "demosaic_stages0_pixels_changed": float(np.mean(d0.max(axis=-1) > 0)), "demosaic_stages0_max_abs_delta": float(d0.max()),
            "demosaic_stages0_mean_abs_delta": float(d0.mean()),
            "srgb_stages1_pixels_changed": float(np.mean(ds.max(axis=-1) > 0)), "srgb_stages1_max_abs_delta": float(ds.max()),
            "srgb_stages1_mean_abs_delta": float(ds.mean()),
            "srgb_stages1_pixels_changed_by_more_than_1_255": float(np.mean(ds.max(axis=-1) > 1 / 255)),
        }
        print(json.dumps(report), flush=True)


if __name__ == "__main__":
    main()
