#!/usr/bin/env python3
"""Generates pysp_amd/csrc/median25_run4.inc: the 5x5 medians of four horizontally adjacent pixels from one 5x8 window.

Scheme (all min/max, so the 0-1 principle applies; verified here on all 2^25 binary inputs of every window before the file is written):
  * the eight window columns are sorted by insertion: min3/med3/max3 of three, then two insertions, where rank i of
    (sorted s) + x is med3(s[i-1], s[i], x) -- 12 operations per column;
  * neighbouring sorted columns are merged pairwise (Batcher odd-even merge): A = c1|c2, B = c3|c4, C = c5|c6;
  * a pixel pair shares four columns = two merged lists; of their union only the ranks 8..13 can be the median of a window
    that adds five more samples, so only those six ranks of merge(A,B) / merge(B,C) are produced (pruned odd-even merge);
  * a window's median is the 13th smallest of (those 20) + (its own fifth column T, sorted):
        min(s13, max(s12,t1), max(s11,t2), max(s10,t3), max(s9,t4), max(s8,t5)).
288 three-input-fused operations for four medians (72 each) against 98 each for the pairwise 99-exchange network.
"""
import os, sys
sys.setrecursionlimit(10000)
INF = 'INF'

def oddeven_merge(lo, hi, r):
    step = r * 2
    if step < hi - lo:
        yield from oddeven_merge(lo, hi, step)
        yield from oddeven_merge(lo + r, hi, step)
        yield from [(i, i + r) for i in range(lo + r, hi - r, step)]
    else:
        yield (lo, lo + r)

class G:
    def __init__(self): self.n = []
    def inp(self, k): self.n.append(('in', k)); return len(self.n) - 1
    def op(self, o, *a): self.n.append((o,) + a); return len(self.n) - 1
    def mn(self, a, b): return b if a == INF else a if b == INF else self.op('min', a, b)
    def mx(self, a, b): return INF if INF in (a, b) else self.op('max', a, b)
    def live(self, outs):
        L, st = set(), list(outs)
        while st:
            x = st.pop()
            if x in L: continue
            L.add(x)
            if self.n[x][0] != 'in': st += list(self.n[x][1:])
        return L

def insert(g, s, x):
    """sorted s + one element: rank i of the result is clamp(x, s[i-1], s[i]) = med3 -- one operation per output"""
    return [g.mn(s[0], x)] + [g.op('med3', s[i - 1], s[i], x) for i in range(1, len(s))] + [g.mx(s[-1], x)]

def sort5(g, v):
    a, b, c = v[0], v[1], v[2]
    s = [g.op('min3', a, b, c), g.op('med3', a, b, c), g.op('max3', a, b, c)]
    return insert(g, insert(g, s, v[3]), v[4])          # 3 + 4 + 5 = 12 operations (9 exchanges would be 18)

def merge(g, A, B, W):
    w = [INF] * W
    w[:len(A)] = A
    w[W // 2:W // 2 + len(B)] = B
    for a, b in oddeven_merge(0, W - 1, 1):
        w[a], w[b] = g.mn(w[a], w[b]), g.mx(w[a], w[b])
    return w[:len(A) + len(B)]

def sel(g, S, T):
    terms = [S[5]] + [g.mx(S[4 - j], T[j]) for j in range(5)]
    m = g.op('min3', terms[0], terms[1], terms[2])
    return g.mn(m, g.op('min3', terms[3], terms[4], terms[5]))

NRUN = int(os.environ.get("RUN", "8"))      # medians per call: 4 (5x8 window) or 8 (5x12 window)
assert NRUN % 2 == 0
g = G()
# construction order = emission order: a sliding order keeps few values alive (two merged lists, one middle set, the odd columns still owed)
col = lambda c: sort5(g, [g.inp((c, r)) for r in range(5)])
sc, P, S, meds = {}, {}, {}, []
def pair(j):            # P[j] = columns 2j+1 | 2j+2 merged
    for c in (2 * j + 1, 2 * j + 2):
        if c not in sc: sc[c] = col(c)
    P[j] = merge(g, sc[2 * j + 1], sc[2 * j + 2], 16)
pair(0)
for j in range(NRUN // 2):          # windows 2j (= column 2j + P[j] + P[j+1]) and 2j+1 (= P[j] + P[j+1] + column 2j+5)
    pair(j + 1)
    S[j] = merge(g, P[j], P[j + 1], 32)[7:13]
    if j == 0: sc[0] = col(0)
    meds.append(sel(g, S[j], sc[2 * j]))
    if 2 * j + 5 not in sc: sc[2 * j + 5] = col(2 * j + 5)      # the last window's own column; otherwise sorted already as half of a pair
    meds.append(sel(g, S[j], sc[2 * j + 5]))

# ---- exhaustive 0-1 verification, bit-parallel over all 2^25 assignments of a window's inputs
N = 25
FULL = (1 << (1 << N)) - 1
def var(k):
    x, size = ((1 << (1 << k)) - 1) << (1 << k), 1 << (k + 1)
    while size < (1 << N):
        x |= x << size; size *= 2
    return x
for wi, m in enumerate(meds):
    live = sorted(g.live([m]))
    cone = [g.n[x][1] for x in live if g.n[x][0] == 'in']
    assert len(cone) == 25 and {c for c, _ in cone} == set(range(wi, wi + 5))
    env = {k: var(i) for i, k in enumerate(cone)}
    val = {}
    for x in live:
        t = g.n[x]
        if t[0] == 'in': val[x] = env[t[1]]
        elif t[0] == 'min': val[x] = val[t[1]] & val[t[2]]
        elif t[0] == 'max': val[x] = val[t[1]] | val[t[2]]
        elif t[0] == 'min3': val[x] = val[t[1]] & val[t[2]] & val[t[3]]
        elif t[0] == 'max3': val[x] = val[t[1]] | val[t[2]] | val[t[3]]
        else: a, b, c = (val[y] for y in t[1:]); val[x] = (a & b) | (a & c) | (b & c)
    cnt = [0] * 5
    for k in cone:
        carry = env[k]
        for b in range(5): cnt[b], carry = cnt[b] ^ carry, cnt[b] & carry
    ge13 = (cnt[4] | (cnt[3] & cnt[2] & (cnt[1] | cnt[0]))) & FULL      # at least 13 of the 25 inputs are 1
    assert val[m] == ge13, f"window {wi} is not a median network"
print(f"all {NRUN} windows verified on 2^25 binary inputs each")

# ---- fuse what is left: min(min(a, b), c) with a single-use inner min is one v_min3 (likewise max).  Semantics are unchanged
# (the verification above ran on the unfused graph and min3 = min o min), so this only rewrites the emission.
live = sorted(g.live(meds))
uses = {}
for x in live:
    if g.n[x][0] != 'in':
        for y in g.n[x][1:]: uses[y] = uses.get(y, 0) + 1
for m in meds: uses[m] = uses.get(m, 0) + 1
fused_away = set()
for x in live:
    t = g.n[x]
    if t[0] in ('min', 'max'):
        for i in (1, 2):
            y = t[i]
            if g.n[y][0] == t[0] and uses.get(y, 0) == 1 and y not in fused_away:
                g.n[x] = (t[0] + '3', g.n[y][1], g.n[y][2], t[3 - i])
                fused_away.add(y)
                break
live = sorted(g.live(meds))
nops = sum(1 for x in live if g.n[x][0] != 'in')
name = {}
lines = ["// GENERATED by tools/gen_median_run.py (verified exhaustively there) -- do not edit.",
         f"// in:  float w[5][{NRUN + 4}] (rows x columns of the window);  out: float m0 .. m{NRUN - 1} = medians of columns 0-4, 1-5, ...",
         f"// {nops} min/max/min3/med3/max3 operations."]
for x in live:
    t = g.n[x]
    if t[0] == 'in':
        name[x] = f"w[{t[1][1]}][{t[1][0]}]"
        continue
    name[x] = f"t{x}"
    a = [name[y] for y in t[1:]]
    # MN2 / MX2 / MN3 / MX3 / MD3 are defined by the including file (k_ahd.hip: raw v_min / v_max / v_min3 / v_max3 / v_med3)
    expr = {'min': "MN2({}, {})", 'max': "MX2({}, {})", 'min3': "MN3({}, {}, {})", 'max3': "MX3({}, {}, {})",
            'med3': "MD3({}, {}, {})"}[t[0]].format(*a)
    lines.append(f"const float t{x} = {expr};")
for i, m in enumerate(meds):
    lines.append(f"m{i} = {name[m]};")
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pysp_amd", "csrc", f"median25_run{NRUN}.inc")
open(out, "w").write("\n".join(lines) + "\n")
print("wrote", out, nops, "operations")
