"""ctypes front end of the CPU oracle (oracle/pysp_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (pysp_amd) never does.  Host-side float64 matrix construction is restated here in
NumPy from colorize/transform.py:40-49, colorize/rgb_space.py:19-52 and wb_cct/helpers_cam_mat.py:7-20
(all citations relative to /root/reference).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[ctypes.CDLL] = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split()
                    return "fma" in fl and "avx2" in fl
    except OSError:
        pass
    return False


def build(force: bool = False) -> None:
    """Compile the C restatement (and nothing else) with the recipe in oracle/Makefile."""
    libs = [os.path.join(_HERE, n) for n in ("liboracle.so", "liboracle_fma.so")]
    src = os.path.join(_HERE, "pysp_oracle.c")
    stale = not all(os.path.exists(p) for p in libs) or any(os.path.getmtime(p) < os.path.getmtime(src) for p in libs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


def _cgroup_cpu_quota() -> Optional[float]:
    """CPUs this process may use per the cgroup's CPU bandwidth limit (v2 cpu.max / v1 cfs quota), None when unlimited or unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


_DEFAULT_CAP = 16   # a 1-GPU box of the pool is a 16-CPU share of a 256-CPU host (the cores are visible, the share is what may be used)


def _team():
    """(OpenMP team size, why): OMP_NUM_THREADS if set; else the cgroup CPU quota if there is one; else min(cores in the affinity mask, 16)."""
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        return max(1, int(os.environ["OMP_NUM_THREADS"])), "OMP_NUM_THREADS"
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = _cgroup_cpu_quota()
    if q is not None:
        return max(1, min(n, int(q + 0.999))), f"cgroup CPU quota {q:g}"
    if n > _DEFAULT_CAP:
        return _DEFAULT_CAP, f"capped at the box's 16-CPU share per GPU ({n} cores visible, no cgroup quota readable; OMP_NUM_THREADS overrides)"
    return max(1, n), "all cores of the affinity mask"


def threads() -> int:
    """OpenMP team size used by the oracle (see _team)."""
    return _team()[0]


def thread_cap_reason() -> str:
    return _team()[1]


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        build()
        # A container may see every host core through os.cpu_count() while its CPU quota is far smaller;
        # an OpenMP team sized for the host then crawls.  Cap the team unless the caller chose a size.
        name = "liboracle_fma.so" if _cpu_has_fma() else "liboracle.so"
        _LIB = ctypes.CDLL(os.path.join(_HERE, name))
        _LIB.orc_set_threads(threads())
    return _LIB


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray, t=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _chk(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


# ----------------------------------------------------------------------------------------------
# host-side colour constants

D65_XY = (0.31272, 0.32903)  # wb_cct/standard_ill.py:33


def xy_to_XYZ(xy) -> np.ndarray:
    """colour.xy_to_XYZ for Y=1: [x/y, 1, (1-x-y)/y] (call sites rgb_space.py:14,40; helpers_exif.py:50)."""
    x, y = float(xy[0]), float(xy[1])
    return np.array([x / y, 1.0, (1.0 - x - y) / y], dtype=np.float64)


def bradford(cur_xyz, tgt_xyz) -> np.ndarray:
    """wb_cct/helpers_cam_mat.py:7-20."""
    m = np.array([[0.8951000, 0.2664000, -0.1614000],
                  [-0.7502000, 1.7135000, 0.0367000],
                  [0.0389000, -0.0685000, 1.0296000]])
    s = np.matmul(m, tgt_xyz) / np.matmul(m, cur_xyz)
    return np.matmul(np.linalg.inv(m), np.matmul(np.diag(s), m))


def rec709_to_xyz(dest_white_xyz) -> np.ndarray:
    """colorize/rgb_space.py:19-52 for REC709 (:54) adapted from D65 to dest_white_xyz."""
    prim = ((0.64, 0.33), (0.3, 0.6), (0.15, 0.06))
    mat = np.array([[p[0] / p[1] for p in prim], [1, 1, 1], [(1 - p[0] - p[1]) / p[1] for p in prim]], dtype=np.float64)
    white = xy_to_XYZ(D65_XY)
    s = np.linalg.inv(mat) @ white
    mat[:, 0] *= s[0]
    mat[:, 1] *= s[1]
    mat[:, 2] *= s[2]
    return bradford(white, np.array(dest_white_xyz, dtype=np.float64)) @ mat


def final_matrix(xyz_to_cam: np.ndarray, cam_white_xyz) -> np.ndarray:
    """colorize/transform.py:40-49: inv(rownorm(XYZ->cam @ sRGB->XYZ(cam white))) as float64 (3,3)."""
    c = np.matmul(xyz_to_cam, rec709_to_xyz(np.asarray(cam_white_xyz).tolist()))
    c = c / c.sum(axis=1)[:, np.newaxis]
    return np.linalg.inv(c)


def ahd_h() -> np.ndarray:
    """ahd.py:89-94 evaluated with NumPy float32 exactly as written."""
    h_optimal = np.array([-0.2569, 0.4339, 0.5138, 0.4339, -0.2569], dtype=np.float32)
    h_fast = np.array([-0.25, 0.5, 0.5, 0.5, -0.25], dtype=np.float32)
    ratio_optimal = 0.125
    h = (h_optimal * ratio_optimal) + (h_fast * (1 - ratio_optimal))
    return h / h.sum()


# ----------------------------------------------------------------------------------------------
# thin wrappers (NumPy in / NumPy out)

def bayer_to_rgbg(bayer: np.ndarray):
    H, W = bayer.shape
    outs = [np.empty((H // 2, W // 2), np.float32) for _ in range(4)]
    if bayer.dtype == np.uint16:
        b = np.ascontiguousarray(bayer)
        _chk(lib().orc_bayer_to_rgbg_u16(_p(b, ctypes.c_uint16), H, W, *[_p(o) for o in outs]), "bayer_to_rgbg")
    else:
        b = _f32(bayer)
        _chk(lib().orc_bayer_to_rgbg_f32(_p(b), H, W, *[_p(o) for o in outs]), "bayer_to_rgbg")
    return tuple(outs)


def rgbg_to_bayer(r, g1, b, g2) -> np.ndarray:
    r, g1, b, g2 = map(_f32, (r, g1, b, g2))
    h, w = r.shape
    out = np.empty((2 * h, 2 * w), np.float32)
    _chk(lib().orc_rgbg_to_bayer_f32(_p(r), _p(g1), _p(b), _p(g2), h, w, _p(out)), "rgbg_to_bayer")
    return out


def bayer_normalize(bayer_u16: np.ndarray, black: Sequence[float], sat: Sequence[float]) -> np.ndarray:
    b = np.ascontiguousarray(bayer_u16, dtype=np.uint16)
    H, W = b.shape
    bl, sa = _f32(black[:4]), _f32(sat[:4])
    out = np.empty((H, W), np.float32)
    _chk(lib().orc_bayer_normalize_u16(_p(b, ctypes.c_uint16), H, W, _p(bl), _p(sa), _p(out)), "bayer_normalize")
    return out


def rgb2lab(rgb: np.ndarray) -> np.ndarray:
    a = _f32(rgb)
    out = np.empty_like(a)
    _chk(lib().orc_rgb2lab(_p(a), ctypes.c_size_t(a.size // 3), _p(out)), "rgb2lab")
    return out


DEFAULT_LAB_MODE = 1     # like the product's contexts (pysp_ctx_set_lab_mode) and oracle/cv2_restated.py::LAB_MODE


def set_lab_mode(mode: int) -> None:
    """1 (default): the OpenCV 4.10 LUT + trilinear restatement; 0: closed-form Lab with table-driven pow / cbrt.
    Process-wide switch of the oracle; tests that change it restore DEFAULT_LAB_MODE."""
    _chk(lib().orc_set_lab_mode(int(mode)), "set_lab_mode")


def cv410_lut() -> np.ndarray:
    out = np.empty((33, 33, 33, 3), np.int16)
    _chk(lib().orc_cv410_lut(_p(out, ctypes.c_int16)), "cv410_lut")
    return out


def set_cv410_lut(grid=None) -> None:
    """Inject a (33,33,33,3) int16 Lab grid for lab mode 1 ([B][G][R] node, (L, a, b) scaled as OpenCV's RGB2LabLUT_s16), or None for the built-in one.
    Process-wide, like set_lab_mode; tests that inject restore with set_cv410_lut(None)."""
    if grid is None:
        _chk(lib().orc_set_cv410_lut(None), "set_cv410_lut")
        return
    g = np.ascontiguousarray(grid, dtype=np.int16)
    if g.shape != (33, 33, 33, 3):
        raise ValueError("Lab grid must have shape (33, 33, 33, 3)")
    _chk(lib().orc_set_cv410_lut(_p(g, ctypes.c_int16)), "set_cv410_lut")


def lab_tables():
    dec = np.empty((321, 4), np.float32); cb = np.empty((257, 4), np.float32)
    lib().orc_lab_tables(_p(dec), _p(cb)); return dec, cb


def lab_pow24(u):
    a = _f32(u); out = np.empty_like(a)
    lib().orc_lab_pow24(_p(a), ctypes.c_size_t(a.size), _p(out)); return out


def lab_cbrt(x):
    a = _f32(x); out = np.empty_like(a)
    lib().orc_lab_cbrt(_p(a), ctypes.c_size_t(a.size), _p(out)); return out


def build_map(lab_padded: np.ndarray, k_pad: int, is_vertical: bool) -> np.ndarray:
    a = _f32(lab_padded)
    Hp, Wp, _ = a.shape
    out = np.empty((Hp - 2 * k_pad, Wp - 2 * k_pad), np.float32)
    _chk(lib().orc_build_map(_p(a), Hp, Wp, int(k_pad), int(bool(is_vertical)), _p(out)), "build_map")
    return out


def _plane_op(name: str, src: np.ndarray) -> np.ndarray:
    a = _f32(src); h, w = a.shape
    out = np.empty_like(a)
    _chk(getattr(lib(), name)(_p(a), h, w, _p(out)), name)
    return out


def gaussian_blur3(src): return _plane_op("orc_gaussian_blur3", src)
def median5(src): return _plane_op("orc_median5", src)
def box3(src): return _plane_op("orc_box3", src)


def filter2d_3x3(src: np.ndarray, k: np.ndarray) -> np.ndarray:
    a = _f32(src); h, w = a.shape
    kk = np.ascontiguousarray(k, dtype=np.float64).reshape(9)
    out = np.empty_like(a)
    _chk(lib().orc_filter2d_3x3(_p(a), h, w, _p(kk, ctypes.c_double), _p(out)), "filter2d")
    return out


def resize2x_linear(src: np.ndarray) -> np.ndarray:
    a = _f32(src); h, w, c = a.shape
    out = np.empty((2 * h, 2 * w, c), np.float32)
    _chk(lib().orc_resize2x_linear(_p(a), h, w, c, _p(out)), "resize2x")
    return out


def get_rgbg_kernel(base_position: int):
    k = np.empty(36, np.float64)
    _chk(lib().orc_get_rgbg_kernel(int(base_position), _p(k, ctypes.c_double)), "get_rgbg_kernel")
    return tuple(k.reshape(4, 3, 3))


def resample_channel(sub, g_sub, g_hf, pos: int) -> np.ndarray:
    sub, g_sub, g_hf = map(_f32, (sub, g_sub, g_hf))
    h, w = sub.shape
    out = np.empty((2 * h, 2 * w), np.float32)
    _chk(lib().orc_resample_channel(_p(sub), _p(g_sub), _p(g_hf), h, w, int(pos), _p(out)), "resample_channel")
    return out


def resample_g_full(g1, g2) -> np.ndarray:
    g1, g2 = _f32(g1), _f32(g2)
    h, w = g1.shape
    out = np.empty((2 * h, 2 * w), np.float32)
    _chk(lib().orc_resample_g_full(_p(g1), _p(g2), h, w, _p(out)), "resample_g_full")
    return out


def demosaic_draft(bayer, wb) -> np.ndarray:
    b = _f32(bayer); H, W = b.shape; wbf = _f32(wb[:3])
    out = np.empty((H, W, 3), np.float32)
    _chk(lib().orc_demosaic_draft(_p(b), H, W, _p(wbf), _p(out)), "draft")
    return out


def demosaic_eag(bayer, wb) -> np.ndarray:
    b = _f32(bayer); H, W = b.shape; wbf = _f32(wb[:3])
    out = np.empty((H, W, 3), np.float32)
    _chk(lib().orc_demosaic_eag(_p(b), H, W, _p(wbf), _p(out)), "eag")
    return out


class _Taps(ctypes.Structure):
    _fields_ = [(n, c_f32p) for n in ("g_h", "g_v", "r_h", "r_v", "b_h", "b_v", "map_h", "map_v", "pre_post")]


def demosaic_ahd(bayer, wb, M, hdr: bool = False, stages: int = 1, taps: bool = False):
    b = _f32(bayer); H, W = b.shape; wbf = _f32(wb[:3])
    Mm = np.ascontiguousarray(M, dtype=np.float64).reshape(9)
    out = np.empty((H, W, 3), np.float32)
    if not taps:
        _chk(lib().orc_demosaic_ahd(_p(b), H, W, _p(wbf), _p(Mm, ctypes.c_double), int(bool(hdr)), int(stages), _p(out)), "ahd")
        return out
    t = {n: np.empty((H, W), np.float32) for n, _ in _Taps._fields_ if n != "pre_post"}
    t["pre_post"] = np.empty((H, W, 3), np.float32)
    st = _Taps(**{n: _p(a) for n, a in t.items()})
    _chk(lib().orc_demosaic_ahd_taps(_p(b), H, W, _p(wbf), _p(Mm, ctypes.c_double), int(bool(hdr)), int(stages), _p(out),
                                     ctypes.byref(st)), "ahd_taps")
    return out, t


def cam_to_rgb(rgb, M, clip: bool = True) -> np.ndarray:
    a = _f32(rgb)
    Mm = np.ascontiguousarray(M, dtype=np.float64).reshape(9)
    out = np.empty_like(a)
    _chk(lib().orc_cam_to_rgb(_p(a), ctypes.c_size_t(a.size // 3), _p(Mm, ctypes.c_double), int(bool(clip)), _p(out)), "cam_to_rgb")
    return out


def _ew(name, x):
    a = _f32(x); out = np.empty_like(a)
    _chk(getattr(lib(), name)(_p(a), ctypes.c_size_t(a.size), _p(out)), name)
    return out


def clip_rgb(x): return _ew("orc_clip_rgb", x)
def lin_srgb_to_srgb(x): return _ew("orc_lin_srgb_to_srgb", x)
def srgb_to_lin_srgb(x): return _ew("orc_srgb_to_lin_srgb", x)


def pipeline_srgb(bayer, wb, M, quality: int = 2, hdr: bool = False, stages: int = 1, reinhard: bool = False) -> np.ndarray:
    b = _f32(bayer); H, W = b.shape; wbf = _f32(wb[:3])
    Mm = np.ascontiguousarray(M, dtype=np.float64).reshape(9)
    out = np.empty((H, W, 3), np.float32)
    _chk(lib().orc_pipeline_srgb(_p(b), H, W, _p(wbf), _p(Mm, ctypes.c_double), int(quality), int(bool(hdr)), int(stages),
                                 int(bool(reinhard)), _p(out)), "pipeline_srgb")
    return out


def hdr_fuse_params(evs: Sequence[float], wb, target_ev: Optional[float] = None):
    """raw_hdr.py:111-136 host part: target EV, per-frame 2**(ev-target), per-site bias, argmax."""
    evs = [float(e) for e in evs]
    if target_ev is None:
        target_ev = 0
        for e in evs:
            target_ev += e
        target_ev /= len(evs)
    ev_offsets = [2 ** (e - target_ev) for e in evs]
    wb = np.asarray(wb, dtype=np.float32)
    site_w = np.array([wb[0], wb[1], wb[2], wb[1]], dtype=np.float32)  # r,g1,b,g2 (:130-133)
    bias = np.stack([1.6 ** (-0.1 * np.abs(off * site_w)) for off in ev_offsets]).astype(np.float32)  # :136
    return float(target_ev), ev_offsets, bias, int(np.argmax(ev_offsets))


def fuse_raw(frames: Sequence[np.ndarray], evs: Sequence[float], wb, target_ev: Optional[float] = None):
    fr = [_f32(f) for f in frames]
    K = len(fr); H, W = fr[0].shape
    target, offs, bias, kmax = hdr_fuse_params(evs, wb, target_ev)
    offs32 = np.array(offs, dtype=np.float32)
    ptrs = (c_f32p * K)(*[_p(f) for f in fr])
    out = np.empty((H, W), np.float32); cnt = np.empty((H, W), np.int32)
    _chk(lib().orc_fuse_raw(ptrs, K, H, W, _p(offs32), _p(np.ascontiguousarray(bias.reshape(-1))), kmax, _p(out),
                            _p(cnt, ctypes.c_int32)), "fuse_raw")
    return out, cnt, target, max(offs)


def fuse_rgb(images: Sequence[np.ndarray], evs: Sequence[float], coeffs, M=None, applied=None, target_ev: Optional[float] = None):
    """raw_hdr.py:7-83 on arrays: returns (fused [after CCM if M], count, images as left in the exposures)."""
    imgs = [_f32(a) for a in images]
    K = len(imgs); shape = imgs[0].shape
    evs = [float(e) for e in evs]
    if target_ev is None:
        target_ev = 0
        for e in evs:
            target_ev += e
        target_ev /= len(evs)
    offs = [2 ** (e - target_ev) for e in evs]
    bias = np.array([1.6 ** (-0.1 * off) for off in offs]).astype(np.float32)          # :60-61
    kmax = max(k for k, off in enumerate(offs) if off == np.max(offs))                # :67 (last match wins)
    cf = np.ascontiguousarray(np.broadcast_to(np.asarray(coeffs, dtype=np.float32).reshape(-1, 3), (K, 3)))
    ap = np.ascontiguousarray(np.ones(K, np.int32) if applied is None else np.asarray(applied, dtype=np.int32))
    offs32 = np.array(offs, dtype=np.float32)
    outs = [np.empty_like(a) for a in imgs]
    out = np.empty(shape, np.float32); cnt = np.empty(shape, np.int32)
    Mm = None if M is None else np.ascontiguousarray(M, dtype=np.float64).reshape(9)
    ptrs = (c_f32p * K)(*[_p(a) for a in imgs]); optrs = (c_f32p * K)(*[_p(a) for a in outs])
    _chk(lib().orc_fuse_rgb(ptrs, K, ctypes.c_size_t(imgs[0].size // 3), _p(cf), _p(ap, ctypes.c_int32), _p(offs32), _p(bias), kmax,
                            None if Mm is None else _p(Mm, ctypes.c_double), _p(out), _p(cnt, ctypes.c_int32), optrs), "fuse_rgb")
    return out, cnt, outs


def find_hot_threshold(bayer, min_delta: float = 0.025, min_neighbour_count: int = 5):
    b = _f32(bayer); H, W = b.shape
    masks = [np.empty((H // 2, W // 2), np.uint8) for _ in range(4)]
    _chk(lib().orc_find_hot_threshold(_p(b), H, W, ctypes.c_float(min_delta), int(min_neighbour_count),
                                      *[_p(m, ctypes.c_uint8) for m in masks]), "find_hot_threshold")
    return [m.astype(bool) for m in masks]


def plane_views(bayer: np.ndarray):
    """The four strided views bayer_chan_mixer.py:4-21 returns (r, g1, b, g2)."""
    evens = bayer[0::2, :].astype(np.float32); odds = bayer[1::2, :].astype(np.float32)
    return evens[:, 0::2], evens[:, 1::2], odds[:, 1::2], odds[:, 0::2]


def flat_field(bayer, flat, clamp_high: bool = False) -> np.ndarray:
    b, f = _f32(bayer), _f32(flat); H, W = b.shape
    mean = np.array([np.mean(p) for p in plane_views(f)], dtype=np.float32)      # raw_correction.py:44
    out = np.empty_like(b)
    _chk(lib().orc_flat_field(_p(b), _p(f), H, W, _p(mean), int(bool(clamp_high)), _p(out)), "flat_field")
    return out


def warp_table(kr0, kr1, kr2, kr3, kt0, kt1, width, height, cxn, cyn, scale, seed=None, rows=None) -> np.ndarray:
    """dng_warp_rectilinear_coords.pyx:18-96; rows=(y0, y1): only those rows of the table (a band of a frame too large to tabulate whole)."""
    y0, y1 = (0, int(height)) if rows is None else (int(rows[0]), int(rows[1]))
    out = np.empty((y1 - y0, width, 2), np.float32)
    sp = None
    if seed is not None:
        seed = _f32(seed); sp = _p(seed)
    f = ctypes.c_float
    _chk(lib().orc_warp_table_rows(f(kr0), f(kr1), f(kr2), f(kr3), f(kt0), f(kt1), int(width), int(height), f(cxn), f(cyn),
                                   f(scale), sp, y0, y1, _p(out)), "warp_table")
    return out


def lanczos4_table() -> np.ndarray:
    t = np.empty((32, 8), np.float32)
    lib().orc_lanczos4_table(_p(t)); return t


def remap_lanczos4(src, mapx, mapy) -> np.ndarray:
    """Restated cv2.remap(INTER_LANCZOS4): the result has the maps' shape (any shape: e.g. a band of rows of a larger warp); the source is the whole plane."""
    s, mx, my = map(_f32, (src, mapx, mapy))
    H, W = s.shape
    assert mx.shape == my.shape
    out = np.empty(mx.shape, np.float32)
    _chk(lib().orc_remap_lanczos4_n(_p(s), H, W, _p(mx), _p(my), ctypes.c_size_t(mx.size), _p(out)), "remap")
    return out


def remap_linear(src, mapx, mapy) -> np.ndarray:
    src, mapx, mapy = _f32(src), _f32(mapx), _f32(mapy)
    out = np.empty_like(src)
    _chk(lib().orc_remap_linear(_p(src), src.shape[0], src.shape[1], _p(mapx), _p(mapy), _p(out)), "remap_linear")
    return out


def remove_ca(bayer, quad_g_at_r=None, quad_r_at_g=None, wb_r: float = 1.0, quad_g_at_b=None, quad_b_at_g=None, wb_b: float = 1.0) -> np.ndarray:
    """corr_ca/ca_removal.py:48-131 with the lens models' quadrant coordinate fields (h,w,2) given; returns the new mosaic."""
    out = _f32(bayer).copy()
    H, W = out.shape
    qs = [None if q is None else _f32(q) for q in (quad_g_at_r, quad_r_at_g, quad_g_at_b, quad_b_at_g)]
    for q in qs:
        assert q is None or q.shape == (H // 2, W // 2, 2)
    ptr = [None if q is None else _p(q) for q in qs]
    _chk(lib().orc_remove_ca(_p(out), H, W, ptr[0], ptr[1], ctypes.c_float(wb_r), ptr[2], ptr[3], ctypes.c_float(wb_b)), "remove_ca")
    return out


def warp_rectilinear(image, coeffs, centre, scale: float = 1.0) -> np.ndarray:
    img = _f32(image).copy()
    H, W, _ = img.shape
    cf = np.ascontiguousarray(coeffs, dtype=np.float64).reshape(-1)
    _chk(lib().orc_warp_rectilinear(_p(img), H, W, _p(cf, ctypes.c_double), cf.size // 6, ctypes.c_double(centre[0]),
                                    ctypes.c_double(centre[1]), ctypes.c_float(scale)), "warp_rectilinear")
    return img
