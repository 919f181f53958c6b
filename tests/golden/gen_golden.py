#!/usr/bin/env python3
"""Generate the golden fixtures tests/golden/*.npz by running the REFERENCE's own code.

Runs only in the build container (needs /root/reference, read-only).  Nothing here is used at test
time; the fixtures are data (inputs + expected outputs), the reference itself never travels.

How the reference is imported:
  * a scratch directory holds one symlink  pySP -> /root/reference  (the package uses absolute
    `pySP.*` imports); it is deleted at exit;
  * the two Cython units are the reference's own .pyx compiled by oracle/Makefile `ref`
    (oracle/_ref/*.so) and registered under their package-qualified names;
  * third-party packages absent from the image are stood in for, each fixture saying so in `meta`:
      colour          -> one function xy_to_XYZ = [x/y, 1, (1-x-y)/y]            ("colour_shim")
      cv2             -> oracle/cv2_restated.py (restated semantics, UNPINNED)  ("cv2_restated")
      rawpy/exifread/tifftools -> empty modules, never called (file I/O only)  ("io_stubs")
The reference's orchestration (plane assembly, op order, WB-twice quirk, dispatch) runs unchanged.
"""
from __future__ import annotations

import importlib
import importlib.machinery
import importlib.util
import json
import os
import shutil
import struct
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PYSP_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import cv2_restated, oracle  # noqa: E402  (cv2 stand-in + xy_to_XYZ)

XYZ2CAM = [
    np.array([[0.9, -0.3, -0.1], [-0.4, 1.2, 0.2], [-0.1, 0.2, 0.6]], dtype=np.float32),
    np.array([[0.6722, -0.0635, -0.0963], [-0.4287, 1.2460, 0.2028], [-0.0908, 0.2162, 0.5668]], dtype=np.float32),
    np.array([[1.0498, -0.4114, -0.0825], [-0.3812, 1.1211, 0.2954], [-0.0415, 0.1146, 0.7024]], dtype=np.float32),
]
WHITES_XY = [(0.31272, 0.32903), (0.34567, 0.35850), (0.44758, 0.40745)]  # D65, D50, A
MULT = np.array([0.5, 1.0, 0.7], dtype=np.float32)


def setup_reference():
    subprocess = importlib.import_module("subprocess")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all", "ref"], stdout=subprocess.DEVNULL)
    farm = tempfile.mkdtemp(prefix="pysp_farm_")
    os.symlink(REF, os.path.join(farm, "pySP"))
    sys.path.insert(0, farm)

    colour = types.ModuleType("colour")
    colour.xy_to_XYZ = oracle.xy_to_XYZ
    sys.modules["colour"] = colour
    sys.modules["cv2"] = cv2_restated
    for name in ("rawpy", "exifread"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["rawpy"].LibRawError = type("LibRawError", (Exception,), {})
    tt = types.ModuleType("tifftools")
    tt.read_tiff = tt.Datatype = tt.Tag = None
    sys.modules["tifftools"] = tt

    import sysconfig
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    for qual, base in (("pySP.debayer.ahd_homogeneity_cython", "ahd_homogeneity_cython"),
                       ("pySP.dng_warp_corr.dng_warp_rectilinear_coords", "dng_warp_rectilinear_coords")):
        path = os.path.join(ROOT, "oracle", "_ref", base + ext)
        loader = importlib.machinery.ExtensionFileLoader(qual, path)
        spec = importlib.util.spec_from_file_location(qual, path, loader=loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        sys.modules[qual] = mod
    return farm


class FakeWb:
    """Duck-typed stand-in for CameraWhiteBalanceController (only the three methods the pixel path touches)."""

    def __init__(self, mult, mat):
        self._m = np.array(mult, dtype=np.float32)
        self._mat = mat

    def get_reciprocal_multipliers(self):
        return np.copy(1.0 / self._m)

    def get_matrix(self):
        return self._mat

    def copy(self):
        return FakeWb(self._m, self._mat)


def scene(H, W, seed, scale=1.0):
    """SURVEY.md section 8d synthetic scene sampled onto the RGGB lattice."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    s = 0.25 + 0.2 * np.sin(2 * np.pi * x / 257) * np.cos(2 * np.pi * y / 131)
    s += 0.15 * (((x // 37) + (y // 37)) % 2)
    s += 0.02 * rng.standard_normal((H, W))
    gains = np.array([[0.5, 1.0], [1.0, 0.7]])
    g = gains[(np.arange(H) % 2)[:, None], (np.arange(W) % 2)[None, :]]
    return np.clip(s * g * scale, 0, None).astype(np.float32)


def save(name, meta, **arrays):
    arrays["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items() if k != "meta"})


def main():
    farm = setup_reference()
    try:
        from pySP.bayer_chan_mixer import bayer_to_rgbg, rgbg_to_bayer
        from pySP.normalization import bayer_normalize
        from pySP.debayer.gaussian import BayerPatternPosition, CV2_DEFAULT_UNNORM_GAUSSIAN_KERNEL, get_rgbg_kernel
        from pySP.debayer import debayer_ahd, debayer_eag, debayer_fast
        from pySP.debayer.edge_assisted_gaussian import resample_channel, resample_g_to_full_resolution
        from pySP.debayer.ahd_homogeneity_cython import build_map
        from pySP.dng_warp_corr.dng_warp_rectilinear_coords import compute_remapping_table, compute_offset_remapping_table
        from pySP.dng_warp_corr.chan_distortion_corr import apply_opcode_3_warp
        from pySP.colorize.transform import cam_to_lin_srgb, lin_srgb_to_srgb, srgb_to_lin_srgb, clip_rgb
        from pySP.wb_cct.helpers_cam_mat import MatXyzToCamera
        from pySP.base_types.image_base import RawRggbBayerData_BaseType, RawDemosaicData, BayerPattern
        from pySP.const import QualityDemosaic
        from pySP.image import RawRggbBayerData, RawBayerData
        import pySP.raw_hdr as raw_hdr

        rng = np.random.default_rng(7)

        # ---- G1 demux / remux (bit exact)
        u16 = rng.integers(0, 65535, (6, 8), dtype=np.uint16)
        f32 = rng.random((10, 12), dtype=np.float32)
        du = bayer_to_rgbg(u16); df = bayer_to_rgbg(f32)
        save("g1_demux", {"ref": "bayer_chan_mixer.py:4-42"}, u16=u16, f32=f32,
             u16_r=du[0], u16_g1=du[1], u16_b=du[2], u16_g2=du[3],
             f32_r=df[0], f32_g1=df[1], f32_b=df[2], f32_g2=df[3],
             remux=rgbg_to_bayer(*df))

        # ---- G2 photosite kernels
        ks = {}
        for pos in BayerPatternPosition:
            for i, k in enumerate(get_rgbg_kernel(CV2_DEFAULT_UNNORM_GAUSSIAN_KERNEL, pos)):
                ks[f"pos{pos.value}_k{i}"] = k
        save("g2_rgbg_kernel", {"ref": "debayer/gaussian.py:19-54", "order": "TL,TR,BL,BR"}, **ks)

        # ---- G3 build_map (reference's own Cython unit, gcc strict IEEE)
        lab = (rng.random((14, 16, 3)) * np.array([100, 60, 60]) - np.array([0, 30, 30])).astype(np.float32)
        labq = np.round(lab / 8) * 8                       # engineered ties / equalities
        labq[4:9, 5:11] = labq[4, 5]                       # constant region
        labq = labq.astype(np.float32)
        save("g3_build_map", {"ref": "debayer/ahd_homogeneity_cython.pyx:22-68", "native": "oracle/_ref"},
             lab=lab, lab_h=build_map(lab, 1, 3, False), lab_v=build_map(lab, 1, 3, True),
             labq=labq, labq_h=build_map(labq, 1, 3, False), labq_v=build_map(labq, 1, 3, True))

        # ---- G4 cam_to_rgb_norm via cam_to_lin_srgb
        px = (rng.random((9, 11, 3), dtype=np.float32) * 1.6 - 0.3).astype(np.float32)
        g4 = {"px": px}
        for i, (m, w) in enumerate(zip(XYZ2CAM, WHITES_XY)):
            mat = MatXyzToCamera(m, oracle.xy_to_XYZ(w))
            g4[f"m{i}"] = m; g4[f"white{i}"] = oracle.xy_to_XYZ(w)
            g4[f"out{i}_clip"] = cam_to_lin_srgb(px, mat, clip_highlights=True)
            g4[f"out{i}_noclip"] = cam_to_lin_srgb(px, mat, clip_highlights=False)
        g4["clip_rgb"] = clip_rgb(px)
        save("g4_cam_to_rgb", {"ref": "colorize/transform.py:6-53,76-87", "colour_shim": True}, **g4)

        # ---- G5 gamma curves on a dense grid incl. thresholds +- few ULP, <0, >1
        def around(v, n=4):
            a = np.float32(v); out = [a]
            up = dn = a
            for _ in range(n):
                up = np.nextafter(up, np.float32(2)); dn = np.nextafter(dn, np.float32(-2)); out += [up, dn]
            return np.array(out, dtype=np.float32)
        grid = np.concatenate([np.linspace(-0.25, 1.25, 3001, dtype=np.float32), rng.random(2000, dtype=np.float32),
                               around(0.0031308), around(0.04045), around(0.0), around(1.0),
                               np.geomspace(1e-6, 1, 500).astype(np.float32)]).astype(np.float32)
        grid3 = np.resize(grid, (grid.size // 3) * 3).reshape(-1, 1, 3)
        save("g5_gamma", {"ref": "colorize/transform.py:89-111",
                          "note": "NumPy float32 power; last bit platform dependent (this box: AVX512 SVML)"},
             x=grid3, enc=lin_srgb_to_srgb(grid3), dec=srgb_to_lin_srgb(grid3))

        # ---- G6 bayer_normalize
        raw = rng.integers(0, 4096, (8, 10), dtype=np.uint16)
        black, sat = [64, 64, 66, 64], [4095, 4000, 4095, 4095]
        save("g6_normalize", {"ref": "normalization.py:4-24"}, raw=raw, black=np.array(black), sat=np.array(sat),
             out=bayer_normalize(raw, black, sat))

        # ---- G7 warp tables
        args = dict(kr0=1.0, kr1=0.01, kr2=0.002, kr3=-0.0005, kt0=0.0003, kt1=-0.0002, width=23, height=17, cx=0.48, cy=0.53, scale=0.9)
        t0 = compute_remapping_table(args["kr0"], args["kr1"], args["kr2"], args["kr3"], args["kt0"], args["kt1"], args["width"],
                                     args["height"], args["cx"], args["cy"], args["scale"])
        t1 = compute_offset_remapping_table(t0, 1.0, -0.01, 0.002, 0.0, 0.0, 0.0, args["width"], args["height"], 0.5, 0.5, 1.0)
        save("g7_warp_table", {"ref": "dng_warp_corr/dng_warp_rectilinear_coords.pyx:18-96", "native": "oracle/_ref", "args": args,
                               "seeded_args": [1.0, -0.01, 0.002, 0.0, 0.0, 0.0, 0.5, 0.5, 1.0]}, table=t0, seeded=t1)

        # ---- G8 full demosaic through the reference orchestration
        for (H, W) in ((32, 48), (34, 50)):
            for hdr in (False, True):
                bay = scene(H, W, 1000 + H, scale=3.0 if hdr else 1.0)
                if not hdr:
                    bay = np.clip(bay, 0, 1)
                mat = MatXyzToCamera(XYZ2CAM[0], oracle.xy_to_XYZ(WHITES_XY[0]))
                out = {"bayer": bay, "mult": MULT, "xyz2cam": XYZ2CAM[0], "white_xyz": oracle.xy_to_XYZ(WHITES_XY[0])}

                def mk():
                    im = RawRggbBayerData(bay, FakeWb(MULT, mat), 10.0, 1.0)
                    im.set_hdr(hdr)
                    return im
                if not hdr:
                    out["draft"] = mk().demosaic(QualityDemosaic.Draft).image
                    out["eag"] = mk().demosaic(QualityDemosaic.Fast).image
                for st in (0, 1, 3):
                    d = mk().demosaic(QualityDemosaic.Best, st)
                    out[f"ahd{st}"] = d.image
                    if st == 1:
                        lin = d.to_lin_srgb()
                        out["ahd1_lin"] = lin
                        out["ahd1_srgb"] = lin_srgb_to_srgb(lin / (1 + lin) if hdr else lin)
                save(f"g8_demosaic_{H}x{W}{'_hdr' if hdr else ''}",
                     {"ref": "image.py:156-183, debayer/*.py, image_base.py:62-64", "cv2_restated": True, "colour_shim": True,
                      "io_stubs": True, "hdr": hdr}, **out)

        # CFA canonicalisation (image.py:143-152,181,191-193): non-RGGB patterns through RawBayerData
        bay = np.clip(scene(16, 20, 5), 0, 1)
        mat = MatXyzToCamera(XYZ2CAM[0], oracle.xy_to_XYZ(WHITES_XY[0]))
        cfa = {"bayer": bay, "mult": MULT, "xyz2cam": XYZ2CAM[0], "white_xyz": oracle.xy_to_XYZ(WHITES_XY[0])}
        for pat in (BayerPattern.Rggb, BayerPattern.Bggr, BayerPattern.Grbg, BayerPattern.Gbrg):
            rb = RawBayerData()
            rb.sensor_scaled = bay; rb.cam_wb = FakeWb(MULT, mat); rb.current_ev = 9.0; rb.sensor_pattern = pat
            cfa[f"eag_{pat.name}"] = np.ascontiguousarray(rb.demosaic(QualityDemosaic.Fast).image)
            cfa[f"draft_{pat.name}"] = np.ascontiguousarray(rb.demosaic(QualityDemosaic.Draft).image)
            cfa[f"ahd1_{pat.name}"] = np.ascontiguousarray(rb.demosaic(QualityDemosaic.Best, 1).image)
        save("g8_cfa_patterns", {"ref": "image.py:143-152,181,185-197", "cv2_restated": True, "colour_shim": True, "io_stubs": True}, **cfa)

        # The same AHD frames with the OTHER Lab restatement standing in for cv2.cvtColor (the closed form of round 1,
        # oracle/cv2_restated.py LAB_MODE "closed_form"): pins lab mode 0 of the oracle and of the product.
        cv2_restated.LAB_MODE = "closed_form"
        try:
            for (H, W), hdr in (((32, 48), False), ((34, 50), True)):
                bay = scene(H, W, 1000 + H, scale=3.0 if hdr else 1.0)
                if not hdr:
                    bay = np.clip(bay, 0, 1)
                mat = MatXyzToCamera(XYZ2CAM[0], oracle.xy_to_XYZ(WHITES_XY[0]))
                out = {"bayer": bay, "mult": MULT, "xyz2cam": XYZ2CAM[0], "white_xyz": oracle.xy_to_XYZ(WHITES_XY[0])}
                for st in (0, 1):
                    im = RawRggbBayerData(bay, FakeWb(MULT, mat), 10.0, 1.0)
                    im.set_hdr(hdr)
                    out[f"ahd{st}"] = im.demosaic(QualityDemosaic.Best, st).image
                save(f"g8_labmode_closed_form_{H}x{W}{'_hdr' if hdr else ''}",
                     {"ref": "debayer/ahd.py:32-67", "cv2_restated": True, "lab_mode": "closed_form", "colour_shim": True, "io_stubs": True, "hdr": hdr}, **out)
        finally:
            cv2_restated.LAB_MODE = "cv410_lut"

        # resample_channel / resample_g standalone
        sub = rng.random((9, 7), dtype=np.float32); gs = rng.random((9, 7), dtype=np.float32); hf = (rng.random((18, 14), dtype=np.float32) - 0.5)
        save("g8_resample", {"ref": "debayer/edge_assisted_gaussian.py:51-143", "cv2_restated": True},
             sub=sub, g_sub=gs, g_hf=hf,
             out_tl=resample_channel(sub, gs, hf, BayerPatternPosition.TOP_LEFT),
             out_br=resample_channel(sub, gs, hf, BayerPatternPosition.BOTTOM_RIGHT),
             g_full=resample_g_to_full_resolution(sub, gs))

        # ---- G9 HDR raw fusion.  HEAD raises TypeError at raw_hdr.py:150 (constructor without arguments,
        # SURVEY App. C.1); the constructor NAME inside raw_hdr is rebound to a permissive class so that the
        # reference's arithmetic (lines 108-148) runs unchanged and its result can be read back.
        class _Permissive:
            def set_hdr(self, v):
                self.hdr = v
        raw_hdr.RawRggbBayerData = _Permissive
        base = scene(12, 16, 77, scale=4.0)
        evs = [9.0, 10.0, 11.0]
        frames = []
        for k, ev in enumerate(evs):
            f = np.clip(base * np.float32(2.0 ** -k), 0, 1).astype(np.float32)
            f[0:2, 0:2] = 1.0   # saturated in every exposure -> sum of weights == 0
            f[2:4, 0:2] = 0.0   # black in every exposure     -> sum of weights == 0
            frames.append(f)
        mat = MatXyzToCamera(XYZ2CAM[0], oracle.xy_to_XYZ(WHITES_XY[0]))
        exps = [RawRggbBayerData(f, FakeWb(MULT, mat), ev, 1.0) for f, ev in zip(frames, evs)]
        hdr_img, cnt = raw_hdr.fuse_exposures_to_raw(exps)
        save("g9_fuse_raw", {"ref": "raw_hdr.py:85-158", "io_stubs": True, "ctor_rebound": True, "evs": evs,
                             "target_ev": hdr_img.current_ev, "lim_sat": hdr_img.lim_sat},
             frames=np.stack(frames), mult=MULT, fused=hdr_img.sensor_scaled, count=cnt)

        # fuse_exposures_from_debayer
        dem = []
        for f, ev in zip(frames, evs):
            d = RawDemosaicData(np.repeat(f[:, :, None], 3, axis=2) * np.array([0.9, 1.0, 0.8], dtype=np.float32), 1.0 / MULT)
            d.mat_xyz = mat; d.current_ev = ev
            dem.append(d)
        rgb_in = np.stack([d.image for d in dem])
        fused_rgb, cnt_rgb = raw_hdr.fuse_exposures_from_debayer(dem)
        save("g9_fuse_debayer", {"ref": "raw_hdr.py:7-83", "io_stubs": True, "colour_shim": True, "evs": evs},
             rgb=rgb_in, mult=MULT, xyz2cam=XYZ2CAM[0], white_xyz=oracle.xy_to_XYZ(WHITES_XY[0]), fused=fused_rgb, count=cnt_rgb)

        # ---- G11 pre-demosaic cleanup: hot-pixel threshold detector and flat-field correction
        from pySP.raw_bad_pixel_corr import find_erroneous_pixels_threshold
        from pySP.raw_correction import flat_frame_correction
        bay = np.clip(scene(20, 28, 31), 0, 1)
        hot = bay.copy()
        for (yy, xx) in ((0, 0), (5, 9), (10, 3), (19, 27), (7, 7), (12, 20)):
            hot[yy, xx] = min(1.0, hot[yy, xx] + 0.4)
        rb = RawBayerData(); rb.sensor_scaled = hot
        masks = find_erroneous_pixels_threshold(rb)
        masks2 = find_erroneous_pixels_threshold(rb, min_delta=0.01, min_neighbour_count=3)
        yy, xx = np.mgrid[0:20, 0:28]
        flat = (0.9 - 0.5 * ((yy - 10) ** 2 + (xx - 14) ** 2) / 400.0).astype(np.float32)
        flat[3, 4] = 0.0            # division by zero -> +inf -> replaced by the channel maximum
        flat[8, 9] = -0.2           # negative result -> clamped to zero
        img = RawBayerData(); img.sensor_scaled = bay.copy()
        fl = RawBayerData(); fl.sensor_scaled = flat
        flat_frame_correction(img, fl)
        img2 = RawBayerData(); img2.sensor_scaled = bay.copy()
        flat_frame_correction(img2, fl, clamp_high=True)
        img3 = RawBayerData(); img3.sensor_scaled = bay.copy()
        zero = RawBayerData(); zero.sensor_scaled = np.zeros_like(flat)
        flat_frame_correction(img3, zero)
        save("g11_cleanup", {"ref": "raw_bad_pixel_corr.py:30-65, raw_correction.py:25-62", "io_stubs": True, "cv2_restated": True},
             hot=hot, **{f"mask{i}": m for i, m in enumerate(masks)}, **{f"maskb{i}": m for i, m in enumerate(masks2)},
             bayer=bay, flat=flat, corrected=img.sensor_scaled, corrected_clamped=img2.sensor_scaled, corrected_zero_flat=img3.sensor_scaled)

        # ---- G10 WarpRectilinear opcode list through apply_opcode_3_warp
        img = rng.random((20, 26, 3), dtype=np.float32)
        coeffs = [(1.0, 0.01, 0.002, 0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0, 0.0, 0.0), (1.0, -0.01, 0.002, 0.0, 0.0005, -0.0003)]
        payload = struct.pack(">I", 3) + b"".join(struct.pack(">6d", *c) for c in coeffs) + struct.pack(">2d", 0.5, 0.5)
        blob = struct.pack(">I", 2) + struct.pack(">IIII", 9, 1, 1, 4) + b"\0\0\0\0" + struct.pack(">IIII", 1, 1, 0, len(payload)) + payload
        warped = img.copy()
        apply_opcode_3_warp(warped, blob)
        from pySP.dng_warp_corr.chan_distortion_corr import stack_warp_prior
        pr_r = compute_remapping_table(1.0, 0.004, 0.0, 0.0, 0.0, 0.0, img.shape[1], img.shape[0], 0.5, 0.5, 1.0)
        prior = stack_warp_prior(img, pr_r, None, None)
        warped_prior = img.copy()
        apply_opcode_3_warp(warped_prior, blob, prior=prior)
        save("g10_warp_prior", {"ref": "dng_warp_corr/chan_distortion_corr.py:11-41,88-97", "cv2_restated": True, "native": "oracle/_ref"},
             image=img, blob=np.frombuffer(blob, dtype=np.uint8), prior=prior, warped=warped_prior)
        save("g10_warp_apply", {"ref": "dng_warp_corr/chan_distortion_corr.py:43-121", "cv2_restated": True, "native": "oracle/_ref"},
             image=img, blob=np.frombuffer(blob, dtype=np.uint8), coeffs=np.array(coeffs), centre=np.array([0.5, 0.5]), warped=warped)

        # ---- G12 chromatic-aberration removal: the apply half of corr_ca (model fitting is not on the path)
        # tile_roi_finder.py:5 imports `pipeline.border_control.linework.line`, a module that is not part of the
        # reference tree; it is only used by the fitting code, so an empty stand-in lets ca_removal import.
        for name in ("pipeline", "pipeline.border_control", "pipeline.border_control.linework", "pipeline.border_control.linework.line"):
            sys.modules[name] = types.ModuleType(name)
        sys.modules["pipeline.border_control.linework.line"].Line2DXeY = sys.modules["pipeline.border_control.linework.line"].Line2DYeX = object
        from pySP.corr_ca.ca_removal import remove_ca_from_raw
        from pySP.corr_ca.model.poly3 import Poly3CorrectionModel
        from pySP.corr_ca.model.poly5 import Poly5CorrectionModel
        from pySP.corr_ca.model.ptlens import PtLensCorrectionModel
        models = {
            "poly5_pyfloat": (lambda: Poly5CorrectionModel(0.03, -0.008), [0.03, -0.008]),
            "poly5_f64": (lambda: Poly5CorrectionModel(np.float64(-0.025), np.float64(0.012)), [-0.025, 0.012]),
            "poly3_pyfloat": (lambda: Poly3CorrectionModel(0.02), [0.02]),
            "ptlens_f64": (lambda: PtLensCorrectionModel(np.float64(0.01), np.float64(-0.02), np.float64(0.025)), [0.01, -0.02, 0.025]),
        }
        H, W = 40, 56
        probe = np.zeros((H, W), np.float32)
        arrays, meta_models = {}, {}
        for key, (make, coefs) in models.items():
            m = make()
            arrays[key + "_undist"] = m.get_undistorted_coordinates(probe)[:H // 2, :W // 2].astype(np.float32)
            arrays[key + "_dist"] = m.get_distorted_coordinates(probe)[:H // 2, :W // 2].astype(np.float32)
            full = m.get_distorted_coordinates(probe)
            arrays[key + "_dist_full"] = full
            meta_models[key] = coefs
        bay = scene(H, W, 77)
        cases = {"both": ("poly5_pyfloat", "poly5_f64"), "r_only": ("ptlens_f64", None), "b_only": (None, "poly3_pyfloat")}
        for cname, (kr, kb) in cases.items():
            raw = RawRggbBayerData(bay.copy(), FakeWb(MULT, XYZ2CAM[0]), 10.0, 1.0)
            remove_ca_from_raw(raw, models[kr][0]() if kr else None, models[kb][0]() if kb else None)
            arrays["out_" + cname] = raw.sensor_scaled
        save("g12_ca_removal", {"ref": "corr_ca/ca_removal.py:48-131, corr_ca/model/{generic,poly3,poly5,ptlens}.py", "cv2_restated": True, "io_stubs": True,
                                "solver_stub": "pipeline.border_control.linework.line (absent from the reference tree, fitting only)",
                                "models": meta_models, "cases": {k: list(v) for k, v in cases.items()}, "mult": [float(v) for v in MULT]},
             bayer=bay, **arrays)
    finally:
        sys.path.remove(farm)
        shutil.rmtree(farm, ignore_errors=True)


if __name__ == "__main__":
    main()
