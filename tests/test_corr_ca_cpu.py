"""CPU: the host half of corr_ca -- lens-model coordinate fields against the fixture the reference's own model code
produced (tests/golden/g12_ca_removal.npz), argument checks of remove_ca_from_raw."""
import numpy as np
import pytest

from conftest import load_golden


def _models():
    from pysp_amd.corr_ca.model.poly3 import Poly3CorrectionModel
    from pysp_amd.corr_ca.model.poly5 import Poly5CorrectionModel
    from pysp_amd.corr_ca.model.ptlens import PtLensCorrectionModel
    return {
        "poly5_pyfloat": lambda c: Poly5CorrectionModel(*c),
        "poly5_f64": lambda c: Poly5CorrectionModel(*[np.float64(v) for v in c]),      # what a fit leaves behind
        "poly3_pyfloat": lambda c: Poly3CorrectionModel(*c),
        "ptlens_f64": lambda c: PtLensCorrectionModel(*[np.float64(v) for v in c]),
    }


def test_model_fields_match_reference_fixture():
    d, meta = load_golden("g12_ca_removal")
    probe = np.zeros(d["bayer"].shape, np.float32)
    for key, coefs in meta["models"].items():
        m = _models()[key](coefs)
        assert np.array_equal(m.get_undistorted_quadrant(probe), d[key + "_undist"]), key
        assert np.array_equal(m.get_distorted_quadrant(probe), d[key + "_dist"]), key
        full = m.get_distorted_coordinates(probe)
        assert full.dtype == np.float32 and np.array_equal(full, d[key + "_dist_full"]), key
        inv = m.get_undistorted_coordinates(probe)
        assert np.array_equal(inv[:probe.shape[0] // 2, :probe.shape[1] // 2], d[key + "_undist"])
        # forward(inverse(r)) ~ r
        r = np.linspace(0.05, 1.0, 64).astype(np.float32)
        assert np.allclose(m.get_distorted(m.estimate_undistorted(r)), r, atol=2e-5)


def test_lens_fields_in_row_blocks_on_the_host_team_are_the_serial_bits(monkeypatch):
    """pysp_amd/_hostpar.py: the lens fields are evaluated in row blocks on a thread team.  Same bytes as the serial (= reference) expressions:
    every model, Python-float and float64 coefficients (the float64 detour widens every block at the same Newton step), ragged block sizes, a field
    whose inversion does not converge (the global stopping rule), and the fixture again with the team forced on at the fixture's small size."""
    from pysp_amd import _hostpar
    from pysp_amd.corr_ca.model import generic
    cases = [("poly3_pyfloat", (0.004,)), ("poly5_f64", (1.5e-3, -4e-4)), ("poly5_pyfloat", (2e-3, 1e-4)), ("ptlens_f64", (2e-4, -6e-4, 3e-4)),
             ("poly3_pyfloat", (0.9,))]
    for shape in ((262, 390), (98, 1030)):
        probe = np.zeros(shape, np.float32)
        for key, coefs in cases:
            monkeypatch.setenv("PYSP_HOST_THREADS", "1")
            m = _models()[key](coefs)
            u0, d0 = m.get_undistorted_quadrant(probe), m.get_distorted_quadrant(probe)
            r = generic.get_empty_radius_field(probe).reshape(-1)
            e0 = m.estimate_undistorted(r)
            for team in (2, 5, 7):
                monkeypatch.setenv("PYSP_HOST_THREADS", str(team))
                monkeypatch.setattr(_hostpar, "MIN_PARALLEL_ELEMS", 16)
                u1, d1, e1 = m.get_undistorted_quadrant(probe), m.get_distorted_quadrant(probe), m.estimate_undistorted(r)
                monkeypatch.setattr(_hostpar, "MIN_PARALLEL_ELEMS", 1 << 18)
                assert u1.dtype == u0.dtype and u1.tobytes() == u0.tobytes(), (key, shape, team)
                assert d1.dtype == d0.dtype and d1.tobytes() == d0.tobytes(), (key, shape, team)
                assert e1.dtype == e0.dtype and e1.shape == e0.shape and e1.tobytes() == e0.tobytes(), (key, shape, team)
    monkeypatch.setenv("PYSP_HOST_THREADS", "3")
    monkeypatch.setattr(_hostpar, "MIN_PARALLEL_ELEMS", 16)
    d, meta = load_golden("g12_ca_removal")
    probe = np.zeros(d["bayer"].shape, np.float32)
    for key, coefs in meta["models"].items():
        m = _models()[key](coefs)
        assert np.array_equal(m.get_undistorted_quadrant(probe), d[key + "_undist"]) and np.array_equal(m.get_distorted_quadrant(probe), d[key + "_dist"]), key
    assert [s.stop - s.start for s in _hostpar.blocks(10, 4)] == [2, 3, 2, 3] and _hostpar.blocks(3, 8) == [slice(0, 1), slice(1, 2), slice(2, 3)]
    assert _hostpar.pmap(lambda v: v * v, [1, 2, 3]) == [1, 4, 9]
    with pytest.raises(ZeroDivisionError):
        _hostpar.pmap(lambda v: 1 // v, [1, 0, 2])


def test_radius_and_coord_fields():
    from pysp_amd.corr_ca.model.generic import get_empty_coord_field, get_empty_radius_field, mirror_quadrant
    img = np.zeros((6, 8), np.float32)
    r = get_empty_radius_field(img)
    assert r.shape == (3, 4) and r.dtype == np.float32 and r[0, 0] == 1.0
    assert np.isclose(r[-1, -1], np.sqrt(0.5) / np.sqrt(3.5 ** 2 + 2.5 ** 2))
    c = get_empty_coord_field(img)
    assert c.dtype == np.int32 and c[2, 3].tolist() == [2, 3]
    with pytest.raises(ValueError):
        get_empty_radius_field(np.zeros((5, 8)))
    q = np.arange(24, dtype=np.float32).reshape(3, 4, 2) + 1
    f = mirror_quadrant(q, (6, 8))
    assert f[0, 7].tolist() == [q[0, 0, 0], -q[0, 0, 1]] and f[5, 0].tolist() == [-q[0, 0, 0], q[0, 0, 1]] and f[5, 7].tolist() == [-q[0, 0, 0], -q[0, 0, 1]]


def test_fit_roundtrip():
    from pysp_amd.corr_ca.model.poly3 import Poly3CorrectionModel
    from pysp_amd.corr_ca.model.poly5 import Poly5CorrectionModel
    from pysp_amd.corr_ca.model.ptlens import PtLensCorrectionModel
    ru = np.linspace(0.1, 0.95, 50)
    for truth, fresh in ((Poly5CorrectionModel(0.02, -0.005), Poly5CorrectionModel()), (PtLensCorrectionModel(0.01, -0.02, 0.015), PtLensCorrectionModel()),
                         (Poly3CorrectionModel(0.03), Poly3CorrectionModel())):
        pairs = np.stack([truth.get_distorted(ru), ru], axis=1)
        assert fresh.compute_coefficients(pairs)
        assert np.allclose(fresh.get_coefficients(), truth.get_coefficients(), atol=1e-9)
    assert Poly3CorrectionModel(7.0).get_coefficients() == 1.0            # clamped to [0, 1]


def test_remove_ca_argument_checks():
    from pysp_amd.corr_ca import remove_ca_from_raw
    from pysp_amd.corr_ca.model.generic import CaCorrectionModel

    class Raw:
        sensor_scaled = np.zeros((4, 4), np.float32)
    raw = Raw()
    remove_ca_from_raw(raw, None, None)                                  # nothing to do, no GPU touched
    assert raw.sensor_scaled is Raw.sensor_scaled

    class OneWay(CaCorrectionModel):
        pass
    with pytest.raises(ValueError, match="Red lens model is not reversible"):
        remove_ca_from_raw(raw, OneWay(), None)
    with pytest.raises(ValueError, match="Blue lens model is not reversible"):
        remove_ca_from_raw(raw, None, OneWay())
