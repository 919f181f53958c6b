"""CPU: the median-of-25 selection network the HIP kernel includes (pysp_amd/csrc/median25_*.inc) is proved
correct by the 0-1 principle over all 2^25 binary inputs (tools/check_median25.c, about a second)."""
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
