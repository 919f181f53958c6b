"""CPU: the median-of-25 selection network the HIP kernel includes (pysp_amd/csrc/median25_run8.inc) is proved correct by the
0-1 principle over all 2^25 binary inputs of each of its eight windows (tools/gen_median_run.py, a few seconds)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("run", [8])
def test_median25_run_is_the_verified_generated_network(tmp_path, run):
    """The committed eight-pixel network is exactly what the generator emits after its exhaustive 0-1 verification, and
    the emitted text itself (not the generator's graph) returns the medians of random real-valued windows."""
    import re
    import numpy as np
    out = str(tmp_path / f"median25_run{run}.inc")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_median_run.py"), out], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, RUN=str(run)))
    assert res.returncode == 0 and f"all {run} windows verified" in res.stdout, res.stdout + res.stderr
    committed = open(os.path.join(ROOT, "pysp_amd", "csrc", f"median25_run{run}.inc")).read()
    assert open(out).read() == committed
    # interpret the emitted statements
    stmts = [l for l in committed.splitlines() if l and not l.startswith("//") and not l.startswith("MED_NEED")]
    rng = np.random.default_rng(3)
    for trial in range(20):
        w = rng.random((5, run + 4)).astype(np.float32) if trial else np.round(rng.random((5, run + 4)) * 3).astype(np.float32)   # ties too
        env = {"w": w, "MN2": min, "MX2": max, "MN3": min, "MX3": max, "MD3": lambda a, b, c: sorted((a, b, c))[1]}
        for l in stmts:
            l = l.rstrip(";").replace("const float ", "")
            name, expr = l.split(" = ", 1)
            env[name] = eval(re.sub(r"w\[(\d)\]\[(\d+)\]", r"w[\1, \2]", expr), {"__builtins__": {}}, env)
        for i in range(run):
            assert env[f"m{i}"] == np.median(w[:, i:i + 5]), (trial, i)
