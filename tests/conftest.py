import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import json
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(d["meta"]))
    return d, meta


def ulp_diff(a, b):
    """Distance in float32 ULPs (monotone integer mapping, handles sign)."""
    def key(x):
        i = np.ascontiguousarray(x, dtype=np.float32).view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)
    return np.abs(key(a) - key(b))


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


def synth_scene(H, W, seed, scale=1.0, clip=True):
    """SURVEY.md 8d synthetic RGGB frame (smooth field + 37 px checker + noise, per-channel gains)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    s = 0.25 + 0.2 * np.sin(2 * np.pi * x / 257) * np.cos(2 * np.pi * y / 131)
    s += 0.15 * (((x // 37) + (y // 37)) % 2)
    s += 0.02 * rng.standard_normal((H, W))
    gains = np.array([[0.5, 1.0], [1.0, 0.7]])
    g = gains[(np.arange(H) % 2)[:, None], (np.arange(W) % 2)[None, :]]
    out = s * g * scale
    return (np.clip(out, 0, 1) if clip else np.clip(out, 0, None)).astype(np.float32)


XYZ2CAM = np.array([[0.9, -0.3, -0.1], [-0.4, 1.2, 0.2], [-0.1, 0.2, 0.6]], dtype=np.float32)
MULT = np.array([0.5, 1.0, 0.7], dtype=np.float32)
D65_XY = (0.31272, 0.32903)
