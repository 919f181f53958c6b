"""CPU: arithmetic identities the HIP kernels rely on for bit-exactness, checked with NumPy on the host (no GPU, no oracle).

* demosaic_common.h::div_by_sat -- bayer_normalize's float32 division v / sat is computed as float32(float64(v) * RN53(1 / sat));
* k_misc.hip::k_warp_remap -- cv2.remap's round-half-even of 32 * coordinate is read from the mantissa of 32 v + 1.5 * 2^23;
* devmath.h::clip01_cv -- np.clip(v, 0, 1) for finite v is the median of (v, 0, 1).
"""
import numpy as np


def _div_by_sat(v, sat):
    return (v.astype(np.float64) * (1.0 / sat.astype(np.float64))).astype(np.float32)


def test_division_by_saturation_level_through_float64_reciprocal():
    rng = np.random.default_rng(5)
    for it in range(12):
        sat = rng.uniform(1000, 70000, 1_000_000).astype(np.float32)
        if it % 3 == 0: sat = np.round(sat)
        if it % 4 == 0: sat = rng.uniform(1e-3, 1e6, sat.size).astype(np.float32)
        v = (rng.random(sat.size).astype(np.float32) * sat).astype(np.float32)
        if it % 2 == 0: v = np.round(v)
        assert np.array_equal(v / sat, _div_by_sat(v, sat))
    # every uint16 count against typical black / saturation levels (what the tile loaders see)
    counts = np.arange(65536, dtype=np.float32)
    for sat in (16383.0, 15871.0, 15871.5, 4095.0, 1023.0, 60000.25):
        for black in (0.0, 64.0, 511.75, 512.0, 600.0):
            s = np.float32(sat)
            v = np.clip(counts - np.float32(black), 0, s).astype(np.float32)
            assert np.array_equal(v / s, _div_by_sat(v, np.full_like(v, s)))
    # sat = 0 keeps the reference's inf / NaN
    with np.errstate(all="ignore"):
        v = np.array([0.0, 3.0], np.float32); z = np.zeros(2, np.float32)
        a, b = v / z, _div_by_sat(v, z)
    assert np.isnan(a[0]) and np.isnan(b[0]) and a[1] == b[1] == np.inf


def test_magic_number_rounding_of_remap_cells():
    rng = np.random.default_rng(6)
    v = np.concatenate([rng.uniform(0, 131071, 2_000_000), np.arange(0, 4096) / 64.0, np.arange(0, 4096) / 64.0 + 131000]).astype(np.float32)
    ref = np.rint(v * np.float32(32.0)).astype(np.int64)                       # lrintf(32 v), round half to even
    t = (v.astype(np.float64) * 32.0 + 12582912.0).astype(np.float32)          # one FMA: exact product, one rounding
    got = t.view(np.int32).astype(np.int64) - 0x4B400000
    assert np.array_equal(ref, got)
    assert (ref >> 5 == got >> 5).all() and (ref & 31 == got & 31).all()


def test_clip_is_the_median_for_finite_values():
    rng = np.random.default_rng(7)
    v = np.concatenate([rng.normal(0.5, 2.0, 1_000_000), [0.0, -0.0, 1.0, np.inf, -np.inf, 1e-45, -1e-45]]).astype(np.float32)
    med = np.sort(np.stack([v, np.zeros_like(v), np.ones_like(v)]), axis=0)[1]
    assert np.array_equal(np.clip(v, 0, 1), med)
