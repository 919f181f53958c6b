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


def check_data_movement_and_selection(d, meta, inp, orc):
    from oracle import cv2_restated as cv
    assert np.array_equal(cv.copyMakeBorder(inp["plane_r"], 1, 1, 1, 1, cv.BORDER_REFLECT), d["cmb_1111"])
    assert np.array_equal(cv.copyMakeBorder(inp["plane_r"], 0, 1, 0, 1, cv.BORDER_REFLECT), d["cmb_0101"])
    assert np.array_equal(cv.copyMakeBorder(inp["plane_b"], 1, 0, 1, 0, cv.BORDER_REFLECT), d["cmb_1010"])
    assert np.array_equal(cv.medianBlur(inp["chroma_diff"], 5), d["median5"]) and np.array_equal(orc.median5(inp["chroma_diff"]), d["median5"])
    # cv2.blur of the integer vote map: only the ORDER of two box sums is consumed (ahd.py:139); the sums are integers / 9
    bl = d["blur"]
    assert np.array_equal(np.rint(bl * 9), np.rint(cv.blur(inp["vote_map"], (3, 3)) * 9))


def check_float_filters(d, meta, inp, orc, tol=2):
    """tol: ULP.  2 for a real cv2 file (SURVEY.md section 7 hard part 4: IPP / SIMD summation order inside the wheels is not observable from here); the dry run
    of the ingest (a stand-in that IS the restatement) asks for 0, which is what makes a changed summation order visible at all."""
    from oracle import cv2_restated as cv
    rep = {"gauss": int(ulp_diff(cv.GaussianBlur(inp["green_full"], (3, 3), 1.0), d["gauss"]).max()),
           "filter2d": int(max(ulp_diff(cv.filter2D(inp["plane_r"], -1, k), o).max() for k, o in zip(inp["kernels"], d["filter2d"]))),
           "resize": int(ulp_diff(cv.resize(inp["quarter_rgb"], (50, 34)), d["resize"]).max())}
    print("real cv2", meta["cv2"], "max ULP per call:", rep)
    assert all(v <= tol for v in rep.values()), rep
    for name, mode in (("remap_lanczos4", cv.INTER_LANCZOS4), ("remap_linear", cv.INTER_LINEAR)):
        got = cv.remap(inp["remap_src"], inp["remap_x"], inp["remap_y"], mode)
        assert np.max(np.abs(got - d[name])) <= (4 if tol else 0) * np.finfo(np.float32).eps * max(1.0, float(np.abs(d[name]).max())), name


def check_lab_which_restatement(d, meta, inp):
    """Decides the Lab question of DESIGN.md section 3 the day a cv2 machine records the file."""
    from oracle import cv2_restated as cv
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


def check_lab_table_is_the_restated_one(d, meta, inp):
    """The strict pin of the table ENTRIES: the grid recovered from the recorded node outputs is the built-in restated table, entry for entry.  A real cv2 file
    that fails here (and passes check_lab_grid_as_data) says: right path, a few entries rounded the other way -- ship the recorded grid as data."""
    from oracle import cv2_restated as cv
    grid, exact = _gen().lab_grid_from_nodes(d["lab_nodes_out"])
    assert exact, "the node outputs are not of the LUT path's form"
    n_diff = int((grid != cv.cv410_lab_lut()).sum())
    assert n_diff == 0, f"{n_diff} of {grid.size} table entries differ from the restated table"


def check_lab_grid_as_data(d, meta, inp, orc):
    """OpenCV's own table, recovered from its outputs at the grid nodes, is injected (NumPy restatement, C oracle; the GPU twin
    is tests/test_gpu_round4.py::test_lab_grid_injection) and every other recorded cvtColor sample -- cell-edge and cvRound-tie sweep included -- must then be
    reproduced BIT FOR BIT.  If real cv2 did not take the LUT path at all the recovery is inexact and the closed form has to match instead."""
    from oracle import cv2_restated as cv
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


G8_NAMES = ["g8_demosaic_32x48", "g8_demosaic_34x50", "g8_demosaic_32x48_hdr", "g8_demosaic_34x50_hdr"]


def check_full_reference_pipeline(orc, name, golden_dir, calls_path):
    path = os.path.join(golden_dir, "cv2_" + name + ".npz")
    if not os.path.exists(path):
        pytest.skip("cv2_" + name + ".npz absent: run tools/gen_cv2_goldens.py --reference <pySP checkout> where cv2 is installed")
    CALLS = calls_path
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


# ---- the real files, the day they exist ---------------------------------------------------------------------------------------------------------------
def test_real_cv2_data_movement_and_selection_bit_exact(real, orc):
    check_data_movement_and_selection(real[0], real[1], real[2], orc)


def test_real_cv2_float_filters_within_tolerance(real, orc):
    check_float_filters(real[0], real[1], real[2], orc, tol=2)


def test_real_cv2_lab_which_restatement(real):
    check_lab_which_restatement(*real)


def test_real_cv2_lab_grid_as_data(real, orc):
    check_lab_grid_as_data(real[0], real[1], real[2], orc)


@pytest.mark.parametrize("name", G8_NAMES)
def test_real_cv2_full_reference_pipeline(orc, name):
    check_full_reference_pipeline(orc, name, GOLDEN, CALLS)


# ---- dry run of the ingest: the generator itself, fed a stand-in cv2, into a scratch directory; then every consumer on that file -------------
class _StandIn:
    """A module-like object with the entry points tools/gen_cv2_goldens.py calls, backed by oracle/cv2_restated.py; `overrides` replace single calls."""

    def __init__(self, **overrides):
        from oracle import cv2_restated as cv
        for k in dir(cv):
            if not k.startswith("_"):
                setattr(self, k, getattr(cv, k))
        self.__version__ = "4.10.0-dry-run(oracle/cv2_restated.py)"
        for k, v in overrides.items():
            setattr(self, k, v)


def _fabricate_reference_files(out_dir):
    """What `gen_cv2_goldens.py --reference` writes when cv2 is the stand-in: the reference's own orchestration with oracle/cv2_restated.py as cv2 is exactly how
    the committed G8 fixtures were produced (tests/golden/gen_golden.py), so the cv2_g8_* files of the dry run are those arrays under the consumer's keys."""
    for name in G8_NAMES:
        d = np.load(os.path.join(GOLDEN, name + ".npz"))
        hdr = bool(json.loads(str(d["meta"]))["hdr"])
        out = {k: d[k] for k in ("bayer", "mult", "xyz2cam", "white_xyz", "ahd0", "ahd1", "ahd3")}
        if not hdr:
            out["draft"], out["eag"] = d["draft"], d["eag"]
        out["meta"] = np.array(json.dumps({"cv2": "dry-run", "hdr": hdr, "source": name}))
        np.savez_compressed(os.path.join(out_dir, "cv2_" + name + ".npz"), **out)


def _consumers(d, meta, inp, orc, strict):
    """name -> passed?  (every consumer of the calls file, run to completion)"""
    res = {}
    for name, fn in (("data_movement_and_selection", lambda: check_data_movement_and_selection(d, meta, inp, orc)),
                     ("float_filters", lambda: check_float_filters(d, meta, inp, orc, tol=0 if strict else 2)),
                     ("lab_which_restatement", lambda: check_lab_which_restatement(d, meta, inp)),
                     ("lab_table_is_the_restated_one", lambda: check_lab_table_is_the_restated_one(d, meta, inp)),
                     ("lab_grid_as_data", lambda: check_lab_grid_as_data(d, meta, inp, orc))):
        try:
            fn()
            res[name] = True
        except AssertionError:
            res[name] = False
    return res


class TestIngestDryRun:
    """The first real cv2_calls.npz must be a one-shot: here the whole path -- generator, file, every consumer -- runs on a stand-in, and three deliberately
    wrong stand-ins show that each consumer catches the defect it owns and nothing else fails beside it."""

    def _record(self, tmp_path, stand_in):
        path, _ = _gen().write_calls(stand_in, str(tmp_path))
        d = np.load(path)
        return d, json.loads(str(d["meta"])), _gen().call_inputs()[0]

    def test_faithful_stand_in_passes_every_consumer(self, tmp_path, orc):
        d, meta, inp = self._record(tmp_path, _StandIn())
        assert meta["cv2"].startswith("4.10.0") and bool(d["lab_grid_exact"])
        assert _consumers(d, meta, inp, orc, strict=True) == {"data_movement_and_selection": True, "float_filters": True, "lab_which_restatement": True,
                                                             "lab_table_is_the_restated_one": True, "lab_grid_as_data": True}
        _fabricate_reference_files(str(tmp_path))
        for name in G8_NAMES:
            check_full_reference_pipeline(orc, name, str(tmp_path), os.path.join(str(tmp_path), "cv2_calls.npz"))

    def test_lab_table_off_by_one_lsb_is_caught_and_repaired_as_data(self, tmp_path, orc):
        """A `cv2` whose Lab table differs from the restated one by +-1 LSB on 1 % of the nodes (what softfloat-vs-double rounding in OpenCV's table builder would
        look like): only the strict table pin fails; the as-data consumer RECOVERS the perturbed table from the node outputs, injects it into both
        restatements and reproduces every other sample bit for bit."""
        from oracle import cv2_restated as cv
        rng = np.random.default_rng(5)
        base = cv.cv410_lab_lut()
        pert = base.astype(np.int32)
        hit = rng.random(base.shape[:3]) < 0.01
        pert[hit] += rng.choice([-1, 1], size=(int(hit.sum()), 3))
        pert = pert.clip(0, 32767).astype(np.int16)
        assert 0 < int((pert != base).sum()) < base.size // 20

        def cvt(src, code):
            cv.set_cv410_lab_lut(pert)
            try:
                return cv.cvtColor(src, code, mode="cv410_lut")
            finally:
                cv.set_cv410_lab_lut(None)
        d, meta, inp = self._record(tmp_path, _StandIn(cvtColor=cvt))
        grid, exact = _gen().lab_grid_from_nodes(d["lab_nodes_out"])
        assert exact and np.array_equal(grid, pert) and np.array_equal(d["lab_grid_s16"], pert)      # the file carries the perturbed table, ready for set_lab_lut
        res = _consumers(d, meta, inp, orc, strict=True)
        assert res == {"data_movement_and_selection": True, "float_filters": True, "lab_which_restatement": True, "lab_table_is_the_restated_one": False,
                       "lab_grid_as_data": True}, res
        assert np.array_equal(cv.active_cv410_lab_lut(), base) and np.array_equal(orc.cv410_lut(), base)      # the consumers left both restatements as they found them

    def test_filter2d_in_reverse_tap_order_is_caught(self, tmp_path, orc):
        """A `cv2` whose filter2D adds its nine products from the last tap to the first: the same real numbers, other float32 roundings (up to 3 ULP on these
        inputs).  The filter consumer fails -- at the dry run's 0-ULP bar and, on this input, at the 2-ULP bar kept for a real file too."""
        def f2d(src, ddepth, kernel):
            kf = kernel.astype(np.float32)
            p = np.pad(src, 1, mode="reflect")
            h, w = src.shape
            acc = np.zeros_like(src)
            for a in (2, 1, 0):
                for b in (2, 1, 0):
                    if kf[a, b] != 0:
                        acc = acc + kf[a, b] * p[a:a + h, b:b + w]
            return acc
        d, meta, inp = self._record(tmp_path, _StandIn(filter2D=f2d))
        res = _consumers(d, meta, inp, orc, strict=True)
        assert res == {"data_movement_and_selection": True, "float_filters": False, "lab_which_restatement": True, "lab_table_is_the_restated_one": True,
                       "lab_grid_as_data": True}, res
        print("the same stand-in under the real file's 2-ULP bar:", _consumers(d, meta, inp, orc, strict=False)["float_filters"])

    def test_median_with_reflect_border_is_caught(self, tmp_path, orc):
        """A `cv2` whose medianBlur pads with BORDER_REFLECT_101 instead of replicating the edge: only the bit-exact selection consumer fails."""
        def med(src, ksize):
            p = np.pad(src, 2, mode="reflect")
            h, w = src.shape
            st = np.stack([p[a:a + h, b:b + w] for a in range(5) for b in range(5)], axis=0)
            return np.ascontiguousarray(np.sort(st, axis=0)[12])
        d, meta, inp = self._record(tmp_path, _StandIn(medianBlur=med))
        res = _consumers(d, meta, inp, orc, strict=True)
        assert res == {"data_movement_and_selection": False, "float_filters": True, "lab_which_restatement": True, "lab_table_is_the_restated_one": True,
                       "lab_grid_as_data": True}, res
