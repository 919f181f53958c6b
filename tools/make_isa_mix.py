#!/usr/bin/env python3
"""profiles/isa_mix.json -- the VALU instruction-class mix of every hot kernel, taken from the BUILT library (not from a recompilation):

    python tools/make_isa_mix.py [path/to/libpysp_hip.so]

llvm-objdump extracts the gfx950 code objects from the shared object and disassembles them; every VALU instruction of a kernel is put into one of the four issue
classes that tools/ubench_valu4.hip measured on MI355X (profiles/r4_ubench_pairs.log; tools/isa_mix.py holds the classifier):

    F  full rate (f32 add / sub / mul / fma, v_mov, and / or / xor, right shifts, add / sub_u32)           2.38 cycles per wave64 instruction
    A  cheap half rate (min / max / med3, compares, cndmask, conversions, shifts, bfe, addc, perm ...)    1.7 - 2.65 inside a float32 stream
    B  multiplier family (v_dot2*, 24 / 32-bit integer multiplies and mads, every v_pk_*, all float64)      6.0 - 8.3 inside a float32 stream
    T  transcendental                                                                                       8.15

The counts are STATIC (one per instruction of the code object); the select and median kernels are straight-line code per phase, so the static mix is the mix a
wave executes up to the few loop trips of their loaders.  bench.py multiplies the mix with the executed instruction count of the PMC passes (profiles/traffic.json)
and the class costs into `roofline.issue_bound`: the time the kernel would take if its SIMDs issued without a gap.  The file carries the library's sha256, like
traffic.json: a line computed from another build says `stale`.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from isa_mix import ab_class  # noqa: E402

TOOLS = "/opt/rocm/lib/llvm/bin"
# the instances the benchmark workloads launch (demangled-name fragments -> the short name bench.py's Timeline uses)
WANT = {
    "k_ahd_select": "void k_ahd_select<false, false, false, 1, false>(AhdParams)",
    "k_ahd_select_planes": "void k_ahd_select<false, false, false, 2, false>(AhdParams)",
    "k_ahd_select_stream": "void k_ahd_select_stream<false, false>(AhdParams, AhdStreamQueues)",
    "k_ahd_median_stage": "k_ahd_median_stage(MedParams)",
    "k_warp_remap": "k_warp_remap(RemapParams)", "k_fuse_raw": "void k_fuse_raw<4>(FuseParams)",
}
for _k in ("k_eag", "k_draft"):                                  # <TINY, U16, TAIL>: key "k_eag/1" = float32 mosaic, colour tail 1; "k_eag/u16/1" = uint16 mosaic
    for _t in range(4):
        WANT[f"{_k}/{_t}"] = f"void {_k}<false, false, {_t}>(EagParams)"
        WANT[f"{_k}/u16/{_t}"] = f"void {_k}<false, true, {_t}>(EagParams)"
# Static counts that overstate what a wave executes: k_ahd_median_stage holds TWO copies of the colour tail (the staged 16-byte form that runs whenever W % 4 == 0,
# and the per-run fallback); one runs.  Its float64 / multiplier and transcendental instructions are the tail's: halve them.
DYNAMIC_SCALE = {"k_ahd_median_stage": {"B": 0.5, "T": 0.5}}
# class costs in SIMD cycles per wave64 instruction (profiles/r4_ubench_pairs.log): lo = next to an fma / in a 3:1 float32 mix, hi = the serial end of the bracket
COST = {"F": (2.38, 2.38), "A": (1.7, 2.65), "B": (6.0, 8.3), "T": (8.15, 8.15)}
A_UNPAIRED = 4.15       # a class-A instruction with no float32 instruction to hide behind (the median network: A outnumbers F) issues at the plain half rate


def model_cpi(k: dict):
    """[lo, hi] SIMD cycles per VALU instruction of a kernel with class counts k: class A instructions up to the number of F ones cost the paired figure,
    the excess the unpaired half rate."""
    f, a, b, t = k["F"], k["A"], k["B"], k["T"]
    paired, excess = min(a, f), max(0.0, a - f)
    n = max(1.0, f + a + b + t)
    return [(f * COST["F"][i] + paired * COST["A"][i] + excess * A_UNPAIRED + b * COST["B"][i] + t * COST["T"][i]) / n for i in (0, 1)]


def main() -> None:
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pysp_amd", "csrc", "libpysp_hip.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    out = {"_format": "kernel -> static VALU instruction counts per issue class (F, A, B, T: tools/make_isa_mix.py) of the built library; cost = SIMD cycles per wave64 "
                      "instruction [lo, hi] per class (profiles/r4_ubench_pairs.log)", "lib_sha256": sha, "cost_cycles": {k: list(v) for k, v in COST.items()}, "cost_A_unpaired": A_UNPAIRED, "kernels": {}}
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["cp", lib, os.path.join(tmp, "lib.so")], check=True)
        subprocess.run([os.path.join(TOOLS, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
        kernels = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            dis = subprocess.run([os.path.join(TOOLS, "llvm-objdump"), "-d", "--mcpu=gfx950", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                    kernels[cur] = {"F": 0, "A": 0, "B": 0, "T": 0, "other": 0}
                    continue
                if cur is None or not line.startswith("\t"):
                    continue
                op = line.split("//")[0].split()
                if not op:
                    continue
                c = ab_class(op[0])
                kernels[cur][c if c else "other"] += 1
    names = list(kernels)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    norm = lambda s: re.sub(r"\s+", "", s)
    for short, want in WANT.items():
        hit = None
        for n, d in zip(names, dem):
            if norm(want) == norm(d):
                hit = (n, d)
                break
        if hit is None:
            continue
        k = dict(kernels[hit[0]])
        scale = DYNAMIC_SCALE.get(short.split("/")[0], {})
        dyn = {c: k[c] * scale.get(c, 1.0) for c in "FABT"}
        lo, hi = model_cpi(dyn)
        out["kernels"][short] = {"symbol": hit[0], "demangled": hit[1], "F": k["F"], "A": k["A"], "B": k["B"], "T": k["T"], "valu_static": k["F"] + k["A"] + k["B"] + k["T"],
                                 "non_valu_static": k["other"], "executed_scale": scale or None, "cycles_per_inst_model": [round(lo, 3), round(hi, 3)]}
    path = os.path.join(ROOT, "profiles", "isa_mix.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: (v["F"], v["A"], v["B"], v["T"], v["cycles_per_inst_model"]) for k, v in out["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
