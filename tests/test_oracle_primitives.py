"""CPU: cross-checks inside the oracle -- the C primitives against the independent NumPy
restatement (oracle/cv2_restated.py), and the accuracy of the bit-defined pow/cbrt used by the
restated RGB->Lab."""
import numpy as np
import pytest

from oracle import cv2_restated as cv


@pytest.fixture(scope="module")
def img():
    rng = np.random.default_rng(11)
    return (rng.random((23, 31), dtype=np.float32) * 2 - 0.5).astype(np.float32)


def test_ahd_h_constants(orc):
    h = orc.ahd_h()
    assert h.dtype == np.float32
    assert [float(v).hex() for v in h] == ["-0x1.0533160000000p-2", "0x1.0000000000000p-1", "0x1.0533160000000p-1",
                                           "0x1.0000000000000p-1", "-0x1.0533160000000p-2"]


def test_gaussian_blur(orc, img):
    assert np.array_equal(orc.gaussian_blur3(img), cv.GaussianBlur(img, (3, 3), 1.0))


def test_filter2d(orc, img):
    for pos in range(4):
        for k in orc.get_rgbg_kernel(pos):
            assert np.array_equal(orc.filter2d_3x3(img, k), cv.filter2D(img, -1, k))


def test_median_box_resize(orc, img):
    assert np.array_equal(orc.median5(img), cv.medianBlur(img, 5))
    cnt = np.random.default_rng(1).integers(0, 10, img.shape).astype(np.float32)
    assert np.array_equal(orc.box3(cnt), cv.blur(cnt, (3, 3)))
    rgb = np.random.default_rng(2).random((7, 9, 3), dtype=np.float32)
    assert np.array_equal(orc.resize2x_linear(rgb), cv.resize(rgb, (18, 14)))


@pytest.mark.parametrize("shape", [(2, 2), (2, 6), (4, 2), (3, 5)])
def test_tiny_planes(orc, shape):
    a = np.random.default_rng(5).random(shape, dtype=np.float32)
    assert np.array_equal(orc.gaussian_blur3(a), cv.GaussianBlur(a, (3, 3), 1.0))
    assert np.array_equal(orc.median5(a), cv.medianBlur(a, 5))
    k = orc.get_rgbg_kernel(0)[1]
    assert np.array_equal(orc.filter2d_3x3(a, k), cv.filter2D(a, -1, k))


def test_remap_lanczos(orc):
    rng = np.random.default_rng(4)
    src = rng.random((19, 27), dtype=np.float32)
    yy, xx = np.mgrid[0:19, 0:27].astype(np.float32)
    mx = np.clip(xx + rng.normal(0, 1.5, xx.shape).astype(np.float32), 0, 26).astype(np.float32)
    my = np.clip(yy + rng.normal(0, 1.5, yy.shape).astype(np.float32), 0, 18).astype(np.float32)
    assert np.array_equal(orc.lanczos4_table(), cv._lanczos4_tab())
    assert np.array_equal(orc.remap_lanczos4(src, mx, my), cv.remap(src, mx, my, cv.INTER_LANCZOS4))
    ident = orc.remap_lanczos4(src, xx, yy)          # integer coordinates -> exact copy
    assert np.array_equal(ident, src)


def test_lab_pow_cbrt_accuracy(orc):
    v = np.geomspace(0.04045, 1.0, 200001).astype(np.float32)
    v[-1] = 1.0
    p = orc.lab_pow24(v)                     # ((v + 0.055) / 1.055) ** 2.4 from the quadratic-segment table
    assert np.max(np.abs(p / ((v.astype(np.float64) + 0.055) / 1.055) ** 2.4 - 1)) < 2.5e-7
    x = np.geomspace(0.008856, 1.99, 200001).astype(np.float32)
    c = orc.lab_cbrt(x)
    assert np.max(np.abs(c / np.cbrt(x.astype(np.float64)) - 1)) < 3.0e-7
    dec, cb = orc.lab_tables()
    assert dec.shape == (321, 4) and cb.shape == (257, 4) and np.isfinite(dec).all() and np.isfinite(cb).all()
    assert abs(dec[0, 0] - ((2.0 ** -5 + 0.055) / 1.055) ** 2.4) < 1e-9 and abs(dec[320, 0] - 1.0) < 1e-7 and abs(cb[224, 0] - 1.0) < 1e-7


def test_rgb2lab_against_closed_form(orc):
    """Lab mode 0 (closed form with table-driven pow / cbrt) against plain float64 CIELab; mode 1 (the default, OpenCV 4.10's
    33^3 LUT + trilinear) is by construction only an interpolation of it: within 0.5 in L and 1.5 in a, b."""
    rng = np.random.default_rng(9)
    rgb = (rng.random((64, 64, 3)) * 1.4 - 0.2).astype(np.float32)
    lut = orc.rgb2lab(rgb).astype(np.float64)
    try:
        orc.set_lab_mode(0)
        lab = orc.rgb2lab(rgb).astype(np.float64)
        white = orc.rgb2lab(np.ones((1, 1, 3), np.float32))
    finally:
        orc.set_lab_mode(orc.DEFAULT_LAB_MODE)
    c = np.clip(rgb.astype(np.float64), 0, 1)
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = lin @ m.T / np.array([0.950456, 1.0, 1.088754])
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16 / 116)
    L = np.where(xyz[..., 1] > 0.008856, 116 * f[..., 1] - 16, 903.3 * xyz[..., 1])
    ref = np.stack([L, 500 * (f[..., 0] - f[..., 1]), 200 * (f[..., 1] - f[..., 2])], axis=-1)
    assert np.max(np.abs(lab - ref)) < 5e-4
    assert np.max(np.abs(lut - ref)[..., 0]) < 0.5 and np.max(np.abs(lut - ref)[..., 1:]) < 1.5
    assert abs(white[0, 0, 0] - 100) < 1e-3 and np.max(np.abs(white[0, 0, 1:])) < 2e-2


def test_select_ties_go_vertical(orc):
    """ahd.py:139: H is taken only where map_h < map_v; a constant frame ties everywhere."""
    bay = np.full((8, 8), 0.25, np.float32)
    M = np.eye(3)
    out, taps = orc.demosaic_ahd(bay, np.ones(3, np.float32), M, False, 0, taps=True)
    assert np.array_equal(taps["map_h"], taps["map_v"]) and (taps["map_h"] == 9).all()
    assert np.array_equal(out[..., 1], taps["g_v"])


def test_warp_phase_check_on_the_oracle_itself(orc):
    """oracle/checks.py::warp_phase_check (the classified bar of the warp tests and bench lines), on CPU: the oracle's own warp passes with nothing differing, also on
    a band of rows and with a prior; a value moved by one float32 ULP far from any phase boundary fails; a value replaced by the interpolation at the NEIGHBOURING
    phase passes only where the coordinate really lies within 2 ULP of a 1/32-px boundary."""
    from oracle.checks import _near_boundary, warp_phase_check
    rng = np.random.default_rng(2)
    H, W = 70, 96
    img = rng.random((H, W, 3), dtype=np.float32)
    co = np.array([[1.0, 0.02, 0.004, 0.0, 0.001, 0.0], [1.0, 0.0, 0.0, 0.0, 0.0, 0.0], [0.99, -0.02, 0.004, 0.0, 0.0, 0.001]])
    ref = orc.warp_rectilinear(img, co, (0.5, 0.45))
    st = warp_phase_check(ref, img, co, (0.5, 0.45))
    assert st["differing"] == 0 and st["values"] == H * W * 3
    assert warp_phase_check(ref[20:40], img, co, (0.5, 0.45), rows=(20, 40))["differing"] == 0
    bad = ref.copy()
    t = orc.warp_table(*[float(np.float32(v)) for v in co[0]], W, H, 0.5, 0.45, 1.0)
    far = ~(_near_boundary(np.ascontiguousarray(t[..., 0]), W - 1) | _near_boundary(np.ascontiguousarray(t[..., 1]), H - 1))
    y, x = np.argwhere(far)[len(np.argwhere(far)) // 2]
    bad[y, x, 0] = np.nextafter(bad[y, x, 0], np.float32(2))
    with pytest.raises(AssertionError, match="more than 2 ULP"):
        warp_phase_check(bad, img, co, (0.5, 0.45))
    # row-band tables and the n-pixel remap agree with the whole-frame forms
    assert np.array_equal(orc.warp_table(1.0, 0.02, 0.004, 0.0, 0.001, 0.0, W, H, 0.5, 0.45, 1.0, rows=(10, 30)), t[10:30])
    mx, my = np.clip(t[..., 0], 0, W - 1), np.clip(t[..., 1], 0, H - 1)
    whole = orc.remap_lanczos4(np.ascontiguousarray(img[..., 0]), mx, my)
    assert np.array_equal(orc.remap_lanczos4(np.ascontiguousarray(img[..., 0]), mx[10:30], my[10:30]), whole[10:30])
    assert np.array_equal(orc.remap_lanczos4(np.ascontiguousarray(img[..., 0]), mx.ravel()[5:77], my.ravel()[5:77]), whole.ravel()[5:77])
