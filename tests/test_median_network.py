"""CPU: the median-of-25 selection networks the HIP kernel includes (pysp_amd/csrc/median25_*.inc) are proved
correct by the 0-1 principle over all 2^25 binary inputs (tools/check_median25.c for the pairwise network,
tools/gen_median_run4.py for the four-pixel one; a few seconds each)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_median25_network_exhaustive(tmp_path):
    exe = str(tmp_path / "check_median25")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "check_median25.c")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failing binary inputs" in out.stdout


def test_median25_run4_is_the_verified_generated_network(tmp_path):
    """The committed four-pixel network is exactly what the generator emits after its exhaustive 0-1 verification, and the emitted
    text itself (not the generator's graph) returns the medians of random real-valued windows."""
    import re
    import numpy as np
    out = str(tmp_path / "median25_run4.inc")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_median_run4.py"), out], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "all four windows verified" in res.stdout, res.stdout + res.stderr
    committed = open(os.path.join(ROOT, "pysp_amd", "csrc", "median25_run4.inc")).read()
    assert open(out).read() == committed
    # interpret the emitted statements
    stmts = [l for l in committed.splitlines() if l and not l.startswith("//")]
    rng = np.random.default_rng(3)
    for trial in range(20):
        w = rng.random((5, 8)).astype(np.float32) if trial else np.round(rng.random((5, 8)) * 3).astype(np.float32)   # ties too
        env = {"w": w, "fminf": min, "fmaxf": max, "__builtin_amdgcn_fmed3f": lambda a, b, c: sorted((a, b, c))[1]}
        for l in stmts:
            l = l.rstrip(";").replace("const float ", "")
            name, expr = l.split(" = ", 1)
            env[name] = eval(re.sub(r"w\[(\d)\]\[(\d)\]", r"w[\1, \2]", expr), {"__builtins__": {}}, env)
        for i in range(4):
            assert env[f"m{i}"] == np.median(w[:, i:i + 5]), (trial, i)
