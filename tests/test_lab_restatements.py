"""CPU: the two restatements of cv2.cvtColor(COLOR_RGB2LAB) float32 (call sites debayer/ahd.py:58,62 of the reference).

  closed_form : sRGB decode + D65 CIELab, pow / cbrt from quadratic-segment tables -- what the product kernels compute.
  cv410_lut   : OpenCV 4.10's default float path (33^3 int16 LUT, fixed-point trilinear), restated from memory.

Each exists twice, independently: plain NumPy in oracle/cv2_restated.py (what tests/golden/gen_golden.py hands the
reference's unchanged ahd.py as `cv2.cvtColor`) and C in oracle/pysp_oracle.c (what the GPU is compared with).  The G8
fixtures are therefore NOT oracle-vs-oracle: they come from the NumPy form, the C form must reproduce them."""
import numpy as np
import pytest

from conftest import load_golden


@pytest.fixture(scope="module")
def cv():
    from oracle import cv2_restated
    return cv2_restated


def _probe(seed=5, n=(257, 301)):
    rng = np.random.default_rng(seed)
    x = (rng.random(n + (3,), dtype=np.float32) * np.float32(1.5) - np.float32(0.25)).astype(np.float32)
    x[0, 0] = [0, 0, 0]; x[0, 1] = [1, 1, 1]; x[0, 2] = [0.04045] * 3
    x[0, 3] = np.nextafter(np.float32(0.04045), np.float32(1)); x[0, 4] = [np.nan, 0.5, np.inf]; x[0, 5] = [-np.inf, 2.0, -0.0]
    x[1, :, :] = np.geomspace(1e-7, 1.0, n[1]).astype(np.float32)[:, None]          # greys across the toe
    return x


def test_cv2_restated_module_never_calls_the_c_oracle():
    import inspect
    from oracle import cv2_restated
    src = inspect.getsource(cv2_restated)
    assert "import oracle" not in src and "from . import" not in src and "from oracle" not in src


def test_fmaf_emulation_is_exactly_rounded(cv):
    from fractions import Fraction
    rng = np.random.default_rng(1)
    a = rng.standard_normal(4000).astype(np.float32); b = rng.standard_normal(4000).astype(np.float32)
    c = (-(a.astype(np.float64) * b.astype(np.float64))).astype(np.float32)     # cancellation: the hard case for a double rounding
    c[::2] = rng.standard_normal(2000).astype(np.float32)
    # engineered float32 midpoint: a*b + c = 1 + 2^-24 + tiny
    a[:2] = np.float32(1 + 2.0 ** -12); b[:2] = np.float32(1 + 2.0 ** -12); c[0] = np.float32(-2.0 ** -11 + 2.0 ** -24); c[1] = -c[0]
    got = cv._fmaf(a, b, c)
    for i in range(0, 4000, 7):
        exact = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        lo = np.float32(float(exact))                # float(Fraction) is correctly rounded to double; refine to float32 exactly
        cands = [np.nextafter(lo, np.float32(-np.inf)), lo, np.nextafter(lo, np.float32(np.inf))]
        best = min(cands, key=lambda v: (abs(Fraction(float(v)) - exact), int(np.float32(v).view(np.int32)) & 1))
        assert got[i] == best, i


def test_closed_form_numpy_equals_c_oracle_bit_for_bit(orc, cv):
    x = _probe()
    try:
        orc.set_lab_mode(0)
        got = orc.rgb2lab(x)
    finally:
        orc.set_lab_mode(orc.DEFAULT_LAB_MODE)
    assert np.array_equal(cv.cvtColor(x, cv.COLOR_RGB2LAB, mode="closed_form"), got)
    dec, cb = cv.lab_tables()                        # NumPy builds the tables from libm itself
    od, oc = orc.lab_tables()
    assert np.array_equal(dec, od) and np.array_equal(cb, oc)


def test_cv410_lut_numpy_equals_c_oracle_bit_for_bit(orc, cv):
    x = _probe(6)
    assert orc.DEFAULT_LAB_MODE == 1 and cv.LAB_MODE == "cv410_lut"          # the default everywhere
    assert np.array_equal(cv.cvtColor(x, cv.COLOR_RGB2LAB, mode="cv410_lut"), orc.rgb2lab(x))
    assert np.array_equal(cv.cv410_lab_lut(), orc.cv410_lut())


def test_cv410_lut_structure(cv):
    """Known answers of the published algorithm: grid points reproduce the LUT entries exactly, output is quantised
    (L in steps of 100/2^14, a and b in steps of 1/64), grey stays near the neutral axis."""
    lut = cv.cv410_lab_lut()
    assert lut.shape == (33, 33, 33, 3) and lut.dtype == np.int16
    assert tuple(lut[0, 0, 0]) == (0, 8192, 8192) and lut[32, 32, 32, 0] == 16384 and abs(int(lut[32, 32, 32, 1]) - 8192) <= 1
    g = (np.arange(33, dtype=np.float32) / np.float32(32))
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(33, -1, 3)[..., ::-1].copy()   # [B][G][R] order -> RGB
    lab = cv.cvtColor(np.ascontiguousarray(pts, dtype=np.float32), cv.COLOR_RGB2LAB, mode="cv410_lut")
    want = lut.reshape(33, -1, 3).astype(np.float32)
    assert np.array_equal(lab[..., 0], want[..., 0] * np.float32(100 / 16384))
    assert np.array_equal(lab[..., 1], want[..., 1] * np.float32(1 / 64) - np.float32(128))
    x = _probe(7)
    lab = cv.cvtColor(x, cv.COLOR_RGB2LAB, mode="cv410_lut")
    assert np.array_equal(lab[..., 1] * 64, np.round(lab[..., 1] * 64)) and np.array_equal(lab[..., 0] * 163.84, np.round(lab[..., 0] * 163.84))
    cf = cv.cvtColor(x, cv.COLOR_RGB2LAB, mode="closed_form")
    fin = np.isfinite(x).all(axis=-1)
    d = np.abs(lab - cf)[fin]
    assert d[:, 0].max() < 0.5 and d[:, 1:].max() < 1.5          # interpolation error of the 33^3 grid, largest near black


@pytest.mark.parametrize("name", ["g8_labmode_closed_form_32x48", "g8_labmode_closed_form_34x50_hdr"])
def test_oracle_lab_mode_0_reproduces_reference_orchestration(orc, name):
    """The reference's unchanged ahd.py with the closed-form NumPy restatement as cv2.cvtColor == C oracle in lab mode 0
    (the default mode, 1, is what every g8_demosaic_* fixture pins)."""
    d, meta = load_golden(name)
    wb = (1.0 / d["mult"]).astype(np.float32)
    M = orc.final_matrix(d["xyz2cam"], d["white_xyz"])
    try:
        orc.set_lab_mode(0)
        for st in (0, 1):
            assert np.array_equal(orc.demosaic_ahd(d["bayer"], wb, M, meta["hdr"], st), d[f"ahd{st}"]), st
    finally:
        orc.set_lab_mode(orc.DEFAULT_LAB_MODE)


def test_g8_cfa_patterns_all_qualities(orc):
    """image.py:143-152,181: flip / rot90 into RGGB, demosaic, flip back -- for Draft, Fast and Best."""
    d, _ = load_golden("g8_cfa_patterns")
    wb = (1.0 / d["mult"]).astype(np.float32)
    M = orc.final_matrix(d["xyz2cam"], d["white_xyz"])
    tf = {"Rggb": lambda a: a, "Bggr": lambda a: np.rot90(a, 2), "Gbrg": lambda a: np.flip(a, axis=1), "Grbg": lambda a: np.flip(a, axis=0)}
    for pat, f in tf.items():
        bay = np.ascontiguousarray(f(d["bayer"]))
        assert np.array_equal(f(orc.demosaic_draft(bay, wb)), d[f"draft_{pat}"]), pat
        assert np.array_equal(f(orc.demosaic_eag(bay, wb)), d[f"eag_{pat}"]), pat
        assert np.array_equal(f(orc.demosaic_ahd(bay, wb, M, False, 1)), d[f"ahd1_{pat}"]), pat
