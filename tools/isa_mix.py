#!/usr/bin/env python3
"""isa_mix.py -- per-phase instruction-class histogram of the gfx950 code of one kernel.

    python tools/isa_mix.py k_ahd.hip 'k_ahd_select<false, false, false, 1>' [-D...] > profiles/r3_isa_mix_k_ahd_select.csv

Compiles the .hip file of pysp_amd/csrc to device assembly with the library's own flags (hipcc -S --offload-device-only: no
GPU needed), takes the body of the kernel whose demangled name contains the given string, cuts it into phases at every
`s_barrier` (the kernels' phases are barrier-separated) and counts instructions by class.  The counts are STATIC (one per
instruction in the code object); instructions inside a loop (between a label and a backward branch to it) are also counted
in the column `in_loop`.  The last rows weight the classes with issue costs in cycles per wave64 instruction -- by default the
ones measured by tools/ubench_valu3.hip on MI355X (profiles/r3_ubench_valu.log), or `--costs file.json`.

Classes:
  f32        v_add/sub/mul/fma/fmac/mad _f32 (full-rate float arithmetic)
  sel        v_min/max/med3/min3/max3 _f32, v_cmp*, v_cndmask (compare / select family)
  int_full   v_mov, v_and/or/xor/not, v_lshrrev, v_ashrrev, v_add/sub_u32: integer instructions measured at the full rate
  int_half   every other integer / bit / lane instruction (v_lshlrev, bfe, mul24, mad24, dot2, addc, add3, lshl_add, perm, packed, DPP, SDWA ...)
  cvt        v_cvt_*
  f64        any *_f64 VALU instruction (incl. cvt to/from f64)
  trans      v_exp/log/rcp/rsq/sqrt/sin/cos
  ds         LDS instructions
  vmem       global_/buffer_/flat_/scratch_ loads and stores
  salu       scalar ALU, scalar memory, branches
  sync       s_waitcnt, s_barrier, s_nop, s_sleep and friends (no work)
"""
from __future__ import annotations

import argparse
import csv
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pysp_amd", "csrc")
FLAGS = ["-O3", "-fno-slp-vectorize", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math"]
CLASSES = ["f32", "sel", "int_full", "int_half", "cvt", "f64", "trans", "ds", "vmem", "salu", "sync"]
# SIMD cycles per wave64 instruction with >= 2 waves per SIMD, measured by tools/ubench_valu3.hip on MI355X (profiles/r3_ubench_valu.log):
# full rate 2.3 (v_add/sub/mul/fma/fmac_f32, v_mov, v_and/or/xor, v_lshrrev/ashrrev, v_add/sub_u32), half rate 4.15 (everything else that is
# not transcendental: min/max/med3, compares, v_cndmask, conversions, v_lshlrev, bfe, 24-bit multiplies, dot2, three-operand integer ops,
# all float64, all packed and DPP forms), transcendentals 8.15.  A stream that mixes full- and half-rate instructions overlaps them partly:
# 1:1 fma:med3 runs at 4.9 cycles per pair (serial 6.45), 2:1 at 6.75 per triple (8.75), 3:1 at 8.6 (11.05).
RATE = {"F": 2.3, "S": 4.15, "T": 8.15}
RATE_OF = {"f32": "F", "sel": "S", "int_full": "F", "int_half": "S", "cvt": "S", "f64": "S", "trans": "T"}
DEFAULT_COSTS = {c: RATE[RATE_OF[c]] if c in RATE_OF else 0.0 for c in CLASSES}
FULL_RATE_INT = re.compile(r"v_(mov_b32|and_b32|or_b32|xor_b32|not_b32|lshrrev_b32|ashrrev_i32|add_u32|sub_u32|subrev_u32|add_co_u32|sub_co_u32|subrev_co_u32|add_i32|sub_i32)(_e32|_e64)?$")


def classify(op: str) -> str:
    if op.startswith("ds_"):
        return "ds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "sync" if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep", "s_endpgm", "s_setprio", "s_sethalt", "s_code_end")) else "salu"
    if not op.startswith("v_"):
        return "salu"
    if "_f64" in op:
        return "f64"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
        return "trans"
    if op.startswith("v_cvt_"):
        return "cvt"
    if re.match(r"v_(min|max|med3|min3|max3)_", op) or op.startswith(("v_cmp", "v_cndmask")):
        return "sel"
    if re.match(r"v_pk_", op) or op.endswith(("_dpp", "_sdwa")):
        return "int_half"           # packed, DPP and SDWA forms issue at half rate whatever they compute
    if re.match(r"v_(add|sub|subrev|mul|fma|fmac|mac|mad|fmaak|fmamk|madak|madmk)_f32", op):
        return "f32"
    return "int_full" if FULL_RATE_INT.match(op) else "int_half"


# Round 4 (tools/ubench_valu4.hip, profiles/r4_ubench_pairs.log): the half-rate instructions are TWO kinds once they sit in a float32 stream.
#   A  min / max / med3, compares, v_cndmask, conversions, shifts, bfe, addc, perm, three-operand adds: next to an fma they cost 2.2 - 2.7 cycles (pair 4.6 - 5.1),
#      and in a 3 : 1 mix they vanish (fma, mul, add, X = 8.7 - 8.9 cycles for four instructions);
#   B  the multiplier family -- v_dot2*, 24-bit and 32-bit integer multiplies / mads, every v_pk_*, all float64: next to an fma they cost 5.3 - 6.0 cycles
#      (pair 7.7 - 8.4, MORE than the two alone), 8.3 in a 3 : 1 mix (fma, mul, add, dot2 = 15.5), and they do not overlap with min / max / med3 either.
B_CLASS = re.compile(r"v_(dot\d\w*|mul_u32_u24|mul_i32_i24|mad_u32_u24|mad_i32_i24|mul_lo_u32|mul_hi_u32|mul_hi_i32|mul_lo_i32|mad_u64_u32|mad_i64_i32|pk_\w+)(_e32|_e64)?$")


def ab_class(op: str) -> str:
    """F (full rate), A (cheap half rate), B (multiplier family), T (transcendental) or '' for non-VALU."""
    c = classify(op)
    if c not in RATE_OF:
        return ""
    if c == "trans":
        return "T"
    if c == "f64" or B_CLASS.match(op):
        return "B"
    return "F" if RATE_OF[c] == "F" else "A"


def kernel_body(asm: str, want: str):
    """[(label or None, opcode)] of the kernel whose demangled symbol contains `want`."""
    syms = re.findall(r"^(_Z\w+):\s*; @", asm, flags=re.M)
    dem = subprocess.run(["c++filt"] + syms, capture_output=True, text=True).stdout.split("\n")
    norm = lambda s: re.sub(r"\s+", "", s)
    hits = [s for s, d in zip(syms, dem) if norm(want) in norm(d)]
    if len(hits) != 1:
        raise SystemExit(f"{want!r} matches {len(hits)} kernels: {[d for d in dem if norm(want) in norm(d)] or dem}")
    sym = hits[0]
    start = asm.index(f"\n{sym}:")
    end = asm.index(".Lfunc_end", start)
    out = []
    for line in asm[start:end].split("\n")[2:]:
        line = line.split(";")[0].rstrip()
        if not line:
            continue
        m = re.match(r"^(\.L\w+):", line)
        if m:
            out.append((m.group(1), None, None))
            continue
        t = line.split()
        if not t or t[0].startswith("."):
            continue
        out.append((None, t[0], " ".join(t[1:])))
    return sym, out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel")
    ap.add_argument("--costs")
    ap.add_argument("--phase-names", default="", help="comma-separated names for the barrier-separated phases, in code order")
    ap.add_argument("defs", nargs="*")
    args, extra = ap.parse_known_args()
    costs = dict(DEFAULT_COSTS)
    if args.costs:
        costs.update(json.load(open(args.costs)))
    src = args.source if os.path.exists(args.source) else os.path.join(CSRC, args.source)
    with tempfile.TemporaryDirectory() as tmp:
        s = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + args.defs + ["--offload-device-only", "-S", "-o", s, src, "-I" + os.path.join(ROOT, "include")],
                       check=True, stderr=subprocess.DEVNULL)
        asm = open(s).read()
    sym, body = kernel_body(asm, args.kernel)
    # loops: a branch to a label that was defined earlier
    pos = {}
    loop = [False] * len(body)
    for i, (lab, op, rest) in enumerate(body):
        if lab:
            pos[lab] = i
        elif op and op.startswith(("s_cbranch", "s_branch")):
            tgt = rest.split()[-1] if rest else ""
            if tgt in pos:
                for k in range(pos[tgt], i + 1):
                    loop[k] = True
    names = [n for n in args.phase_names.split(",") if n]
    AB = ["F", "A", "B", "T"]
    phases, cur, cur_loop = [], dict.fromkeys(CLASSES + AB, 0), 0
    for i, (lab, op, rest) in enumerate(body):
        if not op:
            continue
        c = classify(op)
        cur[c] += 1
        if ab_class(op):
            cur[ab_class(op)] += 1
        if loop[i] and c not in ("sync",):
            cur_loop += 1
        if op == "s_barrier":
            phases.append((cur, cur_loop)); cur, cur_loop = dict.fromkeys(CLASSES + AB, 0), 0
    phases.append((cur, cur_loop))
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", sym])
    w.writerow(["phase"] + CLASSES + ["valu_total", "in_loop", "full_rate", "half_rate", "serial_cycles", "cycles_per_valu_inst", "overlap_bound_cycles",
                                      "F", "A_cheap_half", "B_multiplier_family", "T", "mixed_model_lo_cycles", "mixed_model_hi_cycles", "mixed_lo_per_inst", "mixed_hi_per_inst"])
    tot = dict.fromkeys(CLASSES + AB, 0)
    valu = tuple(RATE_OF)

    def tail(ph):
        nv = sum(ph[c] for c in valu)
        nf = sum(ph[c] for c in valu if RATE_OF[c] == "F")
        cf = sum(ph[c] * costs[c] for c in valu if RATE_OF[c] == "F")
        cs = sum(ph[c] * costs[c] for c in valu if RATE_OF[c] != "F")
        # mixed-stream bracket from profiles/r4_ubench_pairs.log: F 2.38; A between 1.7 (3 : 1 mixes) and 2.65 (1 : 1) while there is full-rate work to pair it with,
        # 4.15 for the excess; B between 6.0 (1 : 1) and 8.3 (3 : 1); T 9.5
        nF, nA, nB, nT = ph["F"], ph["A"], ph["B"], ph["T"]
        pa = min(nA, nF)
        lo = nF * 2.38 + pa * 1.7 + (nA - pa) * 4.15 + nB * 6.0 + nT * 9.5
        hi = nF * 2.38 + pa * 2.65 + (nA - pa) * 4.15 + nB * 8.3 + nT * 9.5
        return [nv, None, nf, nv - nf, round(cf + cs, 1), round((cf + cs) / nv, 3) if nv else "", round(max(cf, cs), 1),
                nF, nA, nB, nT, round(lo, 1), round(hi, 1), round(lo / nv, 3) if nv else "", round(hi / nv, 3) if nv else ""]
    for k, (ph, nl) in enumerate(phases):
        t = tail(ph); t[1] = nl
        w.writerow([names[k] if k < len(names) else f"phase{k}"] + [ph[c] for c in CLASSES] + t)
        for c in CLASSES + AB:
            tot[c] += ph[c]
    t = tail(tot); t[1] = sum(n for _, n in phases)
    w.writerow(["total"] + [tot[c] for c in CLASSES] + t)
    w.writerow(["cost_cycles_per_inst"] + [costs[c] for c in CLASSES])


if __name__ == "__main__":
    main()
