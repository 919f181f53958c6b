#!/usr/bin/env python3
"""Record what the REAL OpenCV does for the nine cv2 calls on pySP's hot path.  Run this wherever
`opencv_python==4.10.0.84` (the reference's pin, requirements.txt:5) is installed -- the build image has no cv2:

    python tools/gen_cv2_goldens.py                      # writes tests/golden/cv2_calls.npz
    python tools/gen_cv2_goldens.py --reference /path/to/pySP   # additionally the whole reference pipeline with real cv2
                                                                 # on the G8 frames -> tests/golden/cv2_g8_*.npz

Needs only numpy, cv2 and the fixtures already committed under tests/golden/ (inputs are derived from them, so the
file is reproducible); `--reference` needs an importable checkout of the reference (+ its own dependencies).
tests/test_cv2_goldens.py consumes the files when they exist (skipped otherwise) and compares every restatement in
oracle/cv2_restated.py and oracle/pysp_oracle.c -- and, on the GPU box, the kernels -- with the real outputs.

Call sites (relative to the reference): debayer/ahd.py:58,62 (cvtColor), :64,77-80 (copyMakeBorder), :120-121
(GaussianBlur), :133-134 (blur), :151 (medianBlur); debayer/edge_assisted_gaussian.py:86-87, :141,143 (filter2D),
:156,170,184; debayer/fast_resize.py:28-29,39 (resize); dng_warp_corr/chan_distortion_corr.py:94-97 (remap LANCZOS4);
corr_ca/ca_removal.py:100-128 (remap LINEAR).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def call_inputs():
    """Inputs of every recorded call, derived from committed fixtures only (shared with the consuming test)."""
    rng = np.random.default_rng(410)
    g8 = np.load(os.path.join(GOLDEN, "g8_demosaic_34x50.npz"))
    g8h = np.load(os.path.join(GOLDEN, "g8_demosaic_34x50_hdr.npz"))
    g2 = np.load(os.path.join(GOLDEN, "g2_rgbg_kernel.npz"))
    g10 = np.load(os.path.join(GOLDEN, "g10_warp_apply.npz"))
    bay = g8["bayer"]
    r, g1, b, g2p = (np.ascontiguousarray(bay[0::2, 0::2]), np.ascontiguousarray(bay[0::2, 1::2]),
                     np.ascontiguousarray(bay[1::2, 1::2]), np.ascontiguousarray(bay[1::2, 0::2]))
    H, W = bay.shape
    inp = {"plane_r": r, "plane_b": b}
    inp["green_full"] = np.ascontiguousarray(g8["ahd0"][..., 1])
    inp["kernels"] = np.stack([g2[f"pos{p}_k{i}"] for p in range(4) for i in range(4)])          # (16,3,3) float64
    lin = g8["ahd1_lin"]
    linh = g8h["ahd1_lin"]
    cube = (rng.random((64, 64, 3), dtype=np.float32) * np.float32(1.5) - np.float32(0.25)).astype(np.float32)
    grey = np.repeat(np.geomspace(1e-6, 1.0, 4096).astype(np.float32)[None, :, None], 3, axis=2)
    fine = (rng.random((128, 128, 3), dtype=np.float32) * np.float32(0.02) + np.float32(0.2)).astype(np.float32)   # neighbours a few 1e-3 apart: the vote's regime
    inp["lab_lin"] = np.ascontiguousarray(lin); inp["lab_hdr_tonemapped"] = np.ascontiguousarray(linh / (1 + linh))
    inp["lab_cube"] = cube; inp["lab_grey"] = np.ascontiguousarray(grey); inp["lab_fine"] = fine
    # (round 4) the 33^3 grid NODES: input (p/32, q/32, r/32) for the node [r][q][p] ([B][G][R] order).  cvRound(p/32 * 2^14) = 512 p exactly, so the cell is p and
    # the position inside it 0: seven of the eight trilinear weights vanish and the output IS the table entry -- lab_grid_from_nodes() turns the recorded
    # float output back into OpenCV's int16 table, which pysp_ctx_set_lab_lut / oracle.set_cv410_lut / cv2_restated.set_cv410_lab_lut then take as data.
    g = np.arange(33, dtype=np.float32) / np.float32(32)
    Bn, Gn, Rn = np.meshgrid(g, g, g, indexing="ij")
    inp["lab_nodes"] = np.ascontiguousarray(np.stack([Rn, Gn, Bn], axis=-1).reshape(33 * 33, 33, 3))
    # cell boundaries and cvRound ties: v = (n + 1/2) / 2^14 is exact in float32 and lands on the tie of cvRound(v * 2^14) (round half to even);
    # n around multiples of 512 (a cell edge) and of 32 (a weight step), plus the float32 neighbours of every such value
    ns = np.unique(np.concatenate([(512 * np.arange(33)[:, None] + np.arange(-3, 4)[None, :]).ravel(), (32 * np.arange(0, 513, 37)[:, None] + np.arange(-2, 3)[None, :]).ravel()]))
    ns = ns[(ns >= 0) & (ns <= 16384)]
    ties = np.concatenate([(ns + d) / 16384.0 for d in (0.0, 0.5, 0.25, 0.75)]).astype(np.float32)
    ties = np.unique(np.concatenate([ties, np.nextafter(ties, np.float32(2)), np.nextafter(ties, np.float32(-1))]))
    ties = ties[(ties >= 0) & (ties <= 1)]
    sweep = np.full((3 * len(ties) + 4096, 3), np.float32(0.40625), np.float32)          # 13/32: a node, so the two fixed channels add no weight structure
    for c in range(3):
        sweep[c * len(ties):(c + 1) * len(ties), c] = ties
    sweep[3 * len(ties):] = ties[rng.integers(0, len(ties), (4096, 3))]                  # all three channels on ties / edges at once
    inp["lab_sweep"] = np.ascontiguousarray(sweep[None])
    inp["vote_map"] = rng.integers(0, 10, (H, W)).astype(np.float32)
    inp["chroma_diff"] = np.ascontiguousarray(g8["ahd0"][..., 0] - g8["ahd0"][..., 1])
    inp["quarter_rgb"] = np.ascontiguousarray(np.stack([r, (g1 + g2p) / 2, b], axis=-1))
    img = np.ascontiguousarray(g10["image"][..., 0])
    h, w = img.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    inp["remap_src"] = img
    inp["remap_x"] = np.clip(xx + np.float32(0.9) * np.sin(yy / np.float32(5)) - np.float32(0.3), 0, w - 1).astype(np.float32)
    inp["remap_y"] = np.clip(yy + np.float32(1.1) * np.cos(xx / np.float32(7)) + np.float32(0.2), 0, h - 1).astype(np.float32)
    return inp, (H, W)


LAB_INPUTS = ("lab_lin", "lab_hdr_tonemapped", "lab_cube", "lab_grey", "lab_fine", "lab_nodes", "lab_sweep")


def lab_grid_from_nodes(nodes_out):
    """OpenCV's int16 table from the recorded cvtColor output at the grid nodes: (33*33, 33, 3) float32 -> ((33,33,33,3) int16, exact?).
    In the LUT path L = l * (100 / 2^14) and a = a' * (256 / 2^14) - 128 with integers l, a': the inversion is exact, and `exact` says whether the
    recorded floats really are such values (False means real cv2 took another path -- IPP, or the non-interpolated one -- and the closed form
    restatement is the candidate instead)."""
    o = np.asarray(nodes_out, dtype=np.float64).reshape(33, 33, 33, 3)
    q = np.stack([o[..., 0] * (16384.0 / 100.0), (o[..., 1] + 128.0) * 64.0, (o[..., 2] + 128.0) * 64.0], axis=-1)
    grid = np.rint(q).astype(np.int64)
    back = np.stack([(grid[..., 0].astype(np.float32) * np.float32(100.0 / 16384.0)),
                     (grid[..., 1].astype(np.float32) * np.float32(256.0 / 16384.0) - np.float32(128.0)),
                     (grid[..., 2].astype(np.float32) * np.float32(256.0 / 16384.0) - np.float32(128.0))], axis=-1)
    exact = bool(np.array_equal(back, np.asarray(nodes_out, dtype=np.float32).reshape(33, 33, 33, 3))) and bool((grid >= 0).all() and (grid <= 32767).all())
    return grid.clip(0, 32767).astype(np.int16), exact


def record_calls(cv2):
    inp, (H, W) = call_inputs()
    out = {}
    r, b = inp["plane_r"], inp["plane_b"]
    out["cmb_1111"] = cv2.copyMakeBorder(r, 1, 1, 1, 1, cv2.BORDER_REFLECT)
    out["cmb_0101"] = cv2.copyMakeBorder(r, 0, 1, 0, 1, cv2.BORDER_REFLECT)
    out["cmb_1010"] = cv2.copyMakeBorder(b, 1, 0, 1, 0, cv2.BORDER_REFLECT)
    out["gauss"] = cv2.GaussianBlur(inp["green_full"], (3, 3), 1.0)
    out["filter2d"] = np.stack([cv2.filter2D(r, -1, k) for k in inp["kernels"]])
    for k in LAB_INPUTS:
        out[k + "_out"] = cv2.cvtColor(inp[k], cv2.COLOR_RGB2LAB)
    # OpenCV's own int16 table, ready for pysp_ctx_set_lab_lut / oracle.set_cv410_lut / cv2_restated.set_cv410_lab_lut (and whether the node outputs had the LUT path's form at all)
    grid, exact = lab_grid_from_nodes(out["lab_nodes_out"])
    out["lab_grid_s16"] = grid
    out["lab_grid_exact"] = np.array(exact)
    out["blur"] = cv2.blur(inp["vote_map"], (3, 3))
    out["median5"] = cv2.medianBlur(inp["chroma_diff"], 5)
    out["resize"] = cv2.resize(inp["quarter_rgb"], (W, H))
    out["remap_lanczos4"] = cv2.remap(inp["remap_src"], inp["remap_x"], inp["remap_y"], cv2.INTER_LANCZOS4)
    out["remap_linear"] = cv2.remap(inp["remap_src"], inp["remap_x"], inp["remap_y"], cv2.INTER_LINEAR)
    return out


def record_reference(cv2, ref_path):
    """The reference's own demosaic on the G8 frames with the real cv2 (needs the reference importable as package pySP)."""
    parent, name = os.path.split(os.path.abspath(ref_path.rstrip("/")))
    if name != "pySP":
        import tempfile
        farm = tempfile.mkdtemp(prefix="pysp_farm_")
        os.symlink(os.path.abspath(ref_path), os.path.join(farm, "pySP"))
        parent = farm
    sys.path.insert(0, parent)
    from pySP.const import QualityDemosaic
    from pySP.image import RawRggbBayerData
    from pySP.wb_cct.helpers_cam_mat import MatXyzToCamera

    class FakeWb:
        def __init__(self, mult, mat): self._m, self._mat = np.array(mult, dtype=np.float32), mat
        def get_reciprocal_multipliers(self): return np.copy(1.0 / self._m)
        def get_matrix(self): return self._mat
        def copy(self): return FakeWb(self._m, self._mat)
    for name in ("g8_demosaic_32x48", "g8_demosaic_34x50", "g8_demosaic_32x48_hdr", "g8_demosaic_34x50_hdr"):
        d = np.load(os.path.join(GOLDEN, name + ".npz"))
        hdr = bool(json.loads(str(d["meta"]))["hdr"])
        mat = MatXyzToCamera(d["xyz2cam"], d["white_xyz"])
        out = {"bayer": d["bayer"], "mult": d["mult"], "xyz2cam": d["xyz2cam"], "white_xyz": d["white_xyz"]}

        def mk():
            im = RawRggbBayerData(d["bayer"], FakeWb(d["mult"], mat), 10.0, 1.0)
            im.set_hdr(hdr)
            return im
        if not hdr:
            out["draft"] = mk().demosaic(QualityDemosaic.Draft).image
            out["eag"] = mk().demosaic(QualityDemosaic.Fast).image
        for st in (0, 1, 3):
            out[f"ahd{st}"] = mk().demosaic(QualityDemosaic.Best, st).image
        out["meta"] = np.array(json.dumps({"cv2": cv2.__version__, "hdr": hdr, "source": name}))
        np.savez_compressed(os.path.join(GOLDEN, "cv2_" + name + ".npz"), **out)
        print("wrote cv2_" + name)


def write_calls(cv2, out_dir):
    """Record the nine calls with the module `cv2` (the real one, or a stand-in with the same entry points: tests/test_cv2_goldens.py dry-runs the ingest with
    oracle/cv2_restated.py) into out_dir/cv2_calls.npz; returns the path."""
    out = record_calls(cv2)
    build = cv2.getBuildInformation() if hasattr(cv2, "getBuildInformation") else ""
    ipp = [l.strip() for l in build.splitlines() if "IPP" in l][:3]
    feats = cv2.getCPUFeaturesLine() if hasattr(cv2, "getCPUFeaturesLine") else ""
    out["meta"] = np.array(json.dumps({"cv2": getattr(cv2, "__version__", "?"), "numpy": np.__version__, "ipp": ipp, "cpu_features": feats}))
    path = os.path.join(out_dir, "cv2_calls.npz")
    np.savez_compressed(path, **out)
    return path, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=None, help="path of a pySP checkout: also record its full demosaic with the real cv2")
    ap.add_argument("--any-version", action="store_true", help="record even if cv2 is not 4.10.0 (the version is stored in the file)")
    args = ap.parse_args()
    try:
        import cv2
    except ImportError:
        sys.exit("gen_cv2_goldens.py needs the real OpenCV (pip install opencv-python==4.10.0.84); it is not available in this environment")
    if not cv2.__version__.startswith("4.10.0") and not args.any_version:
        sys.exit(f"cv2 {cv2.__version__} found, the reference pins 4.10.0.84 (pass --any-version to record anyway)")
    cv2.setNumThreads(1)
    cv2.ocl.setUseOpenCL(False)
    _path, out = write_calls(cv2, GOLDEN)
    print("wrote tests/golden/cv2_calls.npz:", {k: v.shape for k, v in out.items() if k != "meta"})
    if args.reference:
        record_reference(cv2, args.reference)


if __name__ == "__main__":
    main()
