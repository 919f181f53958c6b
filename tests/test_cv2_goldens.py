"""Real-OpenCV pinning hook.  tests/golden/cv2_calls.npz (and cv2_g8_*.npz) are written by tools/gen_cv2_goldens.py on a
machine that has opencv_python==4.10.0.84; the build image has no cv2, so there these tests SKIP and the arithmetic inside
the cv2 calls stays "parity unpinned" (DESIGN.md section 3).  When the files exist, every restatement is held against the
real outputs: pure data movement and selection bit-exact, float filters within the stated ULP tolerance, and for
cvtColor(RGB2LAB) whichever restatement the real library matches is reported and must match within its quantum."""
import importlib.util
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, ulp_diff

CALLS = os.path.join(GOLDEN, "cv2_calls.npz")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_cv2_goldens", os.path.join(ROOT, "tools", "gen_cv2_goldens.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_generator_inputs_are_reproducible_without_cv2():
    """The recorded calls' inputs come from committed fixtures only, so the consuming test can rebuild them here."""
    inp, (H, W) = _gen().call_inputs()
    inp2, _ = _gen().call_inputs()
    assert (H, W) == (34, 50) and set(inp) == set(inp2)
    assert all(np.array_equal(inp[k], inp2[k]) for k in inp)
    assert inp["kernels"].shape == (16, 3, 3) and inp["lab_lin"].dtype == np.float32 and inp["remap_x"].shape == inp["remap_src"].shape


@pytest.fixture(scope="module")
def real():
    if not os.path.exists(CALLS):
        pytest.skip("tests/golden/cv2_calls.npz absent: run tools/gen_cv2_goldens.py where opencv_python==4.10.0.84 is installed")
    d = np.load(CALLS)
    return d, json.loads(str(d["meta"])), _gen().call_inputs()[0]


def test_real_cv2_data_movement_and_selection_bit_exact(real, orc):
    from oracle import cv2_restated as cv
    d, _, inp = real
    assert np.array_equal(cv.copyMakeBorder(inp["plane_r"], 1, 1, 1, 1, cv.BORDER_REFLECT), d["cmb_1111"])
    assert np.array_equal(cv.copyMakeBorder(inp["plane_r"], 0, 1, 0, 1, cv.BORDER_REFLECT), d["cmb_0101"])
    assert np.array_equal(cv.copyMakeBorder(inp["plane_b"], 1, 0, 1, 0, cv.BORDER_REFLECT), d["cmb_1010"])
    assert np.array_equal(cv.medianBlur(inp["chroma_diff"], 5), d["median5"]) and np.array_equal(orc.median5(inp["chroma_diff"]), d["median5"])
    # cv2.blur of the integer vote map: only the ORDER of two box sums is consumed (ahd.py:139); the sums are integers / 9
    bl = d["blur"]
    assert np.array_equal(np.rint(bl * 9), np.rint(cv.blur(inp["vote_map"], (3, 3)) * 9))


def test_real_cv2_float_filters_within_tolerance(real, orc):
    from oracle import cv2_restated as cv
    d, meta, inp = real
    tol = 2   # ULP; SURVEY.md section 7 hard part 4 (IPP / SIMD summation order inside the wheels is not observable from here)
    rep = {"gauss": int(ulp_diff(cv.GaussianBlur(inp["green_full"], (3, 3), 1.0), d["gauss"]).max()),
           "filter2d": int(max(ulp_diff(cv.filter2D(inp["plane_r"], -1, k), o).max() for k, o in zip(inp["kernels"], d["filter2d"]))),
           "resize": int(ulp_diff(cv.resize(inp["quarter_rgb"], (50, 34)), d["resize"]).max())}
    print("real cv2", meta["cv2"], "max ULP per call:", rep)
    assert all(v <= tol for v in rep.values()), rep
    for name, mode in (("remap_lanczos4", cv.INTER_LANCZOS4), ("remap_linear", cv.INTER_LINEAR)):
        got = cv.remap(inp["remap_src"], inp["remap_x"], inp["remap_y"], mode)
        assert np.max(np.abs(got - d[name])) <= 4 * np.finfo(np.float32).eps * max(1.0, float(np.abs(d[name]).max())), name


def test_real_cv2_lab_which_restatement(real):
    """Decides the Lab question of DESIGN.md section 3 the day a cv2 machine records the file."""
    from oracle import cv2_restated as cv
    d, meta, inp = real
    worst = {"closed_form": 0.0, "cv410_lut": 0.0}
    for k in _gen().LAB_INPUTS:
        fin = np.isfinite(inp[k]).all(axis=-1)
        for mode in worst:
            diff = np.abs(cv.cvtColor(inp[k], cv.COLOR_RGB2LAB, mode=mode) - d[k + "_out"])[fin]
            worst[mode] = max(worst[mode], float(diff.max()))
    print("real cv2", meta["cv2"], "max |Lab difference| per restatement:", worst)
    # the LUT path's outputs are multiples of 100/2^14 (L) and 1/64 (a, b): one quantum of slack for LUT entries whose
    # softfloat pow / cbrt rounded the other way; the closed form has to agree to float accuracy if it is the one
    assert worst["cv410_lut"] <= 1.0 / 64 + 1e-6 or worst["closed_form"] <= 2e-3, worst


def test_lab_grid_recovery_from_node_outputs_self_check(orc):
    """The generator's node trick, checked WITHOUT cv2: run the restated cvtColor on the node inputs the generator would record, recover the int16 grid
    from the float outputs, and get the built-in table back bit for bit -- also for a grid injected with +-1 LSB changes (so a real table that differs
    from the restated one in the last bit is recovered as it is, not as we believe it to be).  NumPy restatement and C oracle alike."""
    from oracle import cv2_restated as cv
    gen = _gen()
    inp, _ = gen.call_inputs()
    assert inp["lab_nodes"].shape == (33 * 33, 33, 3) and inp["lab_sweep"].shape[0] == 1
    grid, exact = gen.lab_grid_from_nodes(cv.cvtColor(inp["lab_nodes"], cv.COLOR_RGB2LAB, mode="cv410_lut"))
    assert exact and np.array_equal(grid, cv.cv410_lab_lut())
    grid_c, exact_c = gen.lab_grid_from_nodes(orc.rgb2lab(inp["lab_nodes"]))
    assert exact_c and np.array_equal(grid_c, orc.cv410_lut())
    rng = np.random.default_rng(7)
    other = (cv.cv410_lab_lut().astype(np.int32) + rng.integers(-1, 2, (33, 33, 33, 3))).clip(0, 32767).astype(np.int16)
    cv.set_cv410_lab_lut(other); orc.set_cv410_lut(other)
    try:
        g2, ex2 = gen.lab_grid_from_nodes(cv.cvtColor(inp["lab_nodes"], cv.COLOR_RGB2LAB, mode="cv410_lut"))
        g3, ex3 = gen.lab_grid_from_nodes(orc.rgb2lab(inp["lab_nodes"]))
        assert ex2 and ex3 and np.array_equal(g2, other) and np.array_equal(g3, other)
        for k in ("lab_sweep", "lab_cube", "lab_fine"):          # the two implementations agree under an injected grid too
            assert np.array_equal(cv.cvtColor(inp[k], cv.COLOR_RGB2LAB, mode="cv410_lut"), orc.rgb2lab(inp[k]))
    finally:
        cv.set_cv410_lab_lut(None); orc.set_cv410_lut(None)
    assert np.array_equal(orc.cv410_lut(), cv.cv410_lab_lut())
    # a closed-form output is NOT of the LUT path's form: the recovery says so instead of inventing a table
    _, ex_cf = gen.lab_grid_from_nodes(cv.cvtColor(inp["lab_nodes"], cv.COLOR_RGB2LAB, mode="closed_form"))
    assert not ex_cf


def test_real_cv2_lab_grid_as_data(real, orc):
    """The day cv2_calls.npz exists: OpenCV's own table, recovered from its outputs at the grid nodes, is injected (NumPy restatement, C oracle; the GPU twin
    is tests/test_gpu_round4.py::test_lab_grid_injection) and every other recorded cvtColor sample -- cell-edge and cvRound-tie sweep included -- must then be
    reproduced BIT FOR BIT.  If real cv2 did not take the LUT path at all the recovery is inexact and the closed form has to match instead."""
    from oracle import cv2_restated as cv
    d, meta, inp = real
    if "lab_nodes_out" not in d.files:
        pytest.skip("cv2_calls.npz predates the node dump: re-run tools/gen_cv2_goldens.py")
    gen = _gen()
    grid, exact = gen.lab_grid_from_nodes(d["lab_nodes_out"])
    n_diff = int((grid != cv.cv410_lab_lut()).sum())
    print("real cv2", meta["cv2"], ": LUT-path form at the nodes:", exact, "; entries differing from the restated table:", n_diff, "of", grid.size)
    if not exact:
        worst = max(float(np.abs(cv.cvtColor(inp[k], cv.COLOR_RGB2LAB, mode="closed_form") - d[k + "_out"])[np.isfinite(inp[k]).all(axis=-1)].max()) for k in gen.LAB_INPUTS)
        assert worst <= 2e-3, worst
        return
    cv.set_cv410_lab_lut(grid); orc.set_cv410_lut(grid)
    try:
        for k in gen.LAB_INPUTS:
            fin = np.isfinite(inp[k]).all(axis=-1)
            assert np.array_equal(cv.cvtColor(inp[k], cv.COLOR_RGB2LAB, mode="cv410_lut")[fin], d[k + "_out"][fin]), k
            assert np.array_equal(orc.rgb2lab(inp[k])[fin], d[k + "_out"][fin]), k
    finally:
        cv.set_cv410_lab_lut(None); orc.set_cv410_lut(None)


@pytest.mark.parametrize("name", ["g8_demosaic_32x48", "g8_demosaic_34x50", "g8_demosaic_32x48_hdr", "g8_demosaic_34x50_hdr"])
def test_real_cv2_full_reference_pipeline(orc, name):
    path = os.path.join(GOLDEN, "cv2_" + name + ".npz")
    if not os.path.exists(path):
        pytest.skip("cv2_" + name + ".npz absent: run tools/gen_cv2_goldens.py --reference <pySP checkout> where cv2 is installed")
    d = np.load(path)
    hdr = bool(json.loads(str(d["meta"]))["hdr"])
    wb = (1.0 / d["mult"]).astype(np.float32)
    M = orc.final_matrix(d["xyz2cam"], d["white_xyz"])
    if not hdr:
        assert ulp_diff(orc.demosaic_draft(d["bayer"], wb), d["draft"]).max() <= 2
        assert ulp_diff(orc.demosaic_eag(d["bayer"], wb), d["eag"]).max() <= 4
    rates = {}
    grid = None
    if os.path.exists(CALLS) and "lab_nodes_out" in np.load(CALLS).files:
        grid, exact = _gen().lab_grid_from_nodes(np.load(CALLS)["lab_nodes_out"])
        grid = grid if exact else None
    for mode in (0, 1, "1+real grid") if grid is not None else (0, 1):
        orc.set_lab_mode(0 if mode == 0 else 1)
        if mode == "1+real grid":
            orc.set_cv410_lut(grid)
        try:
            got = orc.demosaic_ahd(d["bayer"], wb, M, hdr, 0)
        finally:
            orc.set_lab_mode(orc.DEFAULT_LAB_MODE); orc.set_cv410_lut(None)
        rates[mode] = float(np.mean(ulp_diff(got, d["ahd0"]).max(axis=-1) > 4))      # pixels that took the other direction
    print(name, "fraction of pixels whose H/V decision differs from the real-cv2 reference, per Lab mode:", rates)
    assert min(rates.values()) <= 0.01, rates
