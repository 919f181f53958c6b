#!/usr/bin/env python3
"""VGPRs / SGPRs / LDS / scratch of every gfx950 kernel in a built library (llvm-readelf --notes of the embedded code objects).
    python tools/kernel_resources.py [lib.so] [name filter]"""
import os
import shutil
import subprocess
import sys
import tempfile

TOOLS = "/opt/rocm/lib/llvm/bin"
KEYS = (".vgpr_count:", ".agpr_count:", ".sgpr_count:", ".group_segment_fixed_size:", ".private_segment_fixed_size:", ".vgpr_spill_count:")


def kernels(lib):
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(lib, tmp)
        subprocess.run([os.path.join(TOOLS, "llvm-objdump"), "--offloading", os.path.basename(lib)], cwd=tmp, check=True, capture_output=True)
        out = {}
        for f in os.listdir(tmp):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(TOOLS, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in notes.splitlines():
                line = line.strip()
                if line.startswith("- ."):
                    cur = {}
                    line = line[2:]
                if cur is None:
                    continue
                if line.startswith(".name:"):
                    out[line.split(":", 1)[1].strip()] = cur
                elif line.startswith(KEYS):
                    k, v = line.split(":")
                    cur[k.strip(".")] = int(v)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "pysp_amd", "csrc", "libpysp_hip.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, v in sorted(kernels(lib).items()):
        if flt in name:
            demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            print(f"{demangled[:90]:90s} vgpr {v.get('vgpr_count', 0):4d} sgpr {v.get('sgpr_count', 0):4d} lds {v.get('group_segment_fixed_size', 0):6d} scratch {v.get('private_segment_fixed_size', 0)}")
