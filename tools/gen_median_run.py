#!/usr/bin/env python3
"""Generates pysp_amd/csrc/median25_run8.inc: the 5x5 medians of eight horizontally adjacent pixels from one 5x12 window.

Scheme (all min/max/med3, so the 0-1 principle applies; every window of the FINAL network is verified here on all 2^25 binary
inputs before the file is written):
  * the window columns are sorted by insertion: min3/med3/max3 of three, then two insertions, where rank i of
    (sorted s) + x is med3(s[i-1], s[i], x) -- 12 operations per column;
  * neighbouring sorted columns are merged pairwise (Batcher odd-even merge): P0 = c1|c2, P1 = c3|c4, ... shared by up to four windows;
  * a pixel pair shares four columns = two merged lists; of their union only the ranks 8..13 can be the median of a window
    that adds five more samples, so only those six ranks of merge(Pj, Pj+1) are produced (pruned odd-even merge);
  * a window's median is the 13th smallest of (those 20) + (its own fifth column T, sorted) = the median of the six kept ranks s8..s13
    and t1..t5:  med3(s8, s13, med3(s9, t4, med3(s10, t3, med3(s11, t2, med3(s12, t1, t5))))) -- five operations;
  * resynthesis: every node of that graph is then matched, as a Boolean function of the window's 25 inputs, against
    min / max / min3 / max3 / med3 of all pairs and triples of nodes up to three levels below it (a clamp between two values whose
    order is implied by the network is one med3, ...), and a 0-1 programme picks the cheapest set of nodes that still produces the
    eight medians.  468 -> 399 operations here (merges 26 -> 19, pruned merges 36 -> 24).
Needs scipy (milp).  RUN=4 builds the four-pixel network instead.
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
    """median of 25 = 6th smallest of S[0..5] (ranks 8..13 of the shared 20, sorted) + T[0..4] (the fifth column, sorted): five nested
    clamps (found by tools/formula_dp.py 6 5 6; the 0-1 verification below covers it);  the min/max form min(s13, max(s12,t1), ..., max(s8,t5)) takes eight operations"""
    m = g.op('med3', S[4], T[0], T[4])
    m = g.op('med3', S[3], T[1], m)
    m = g.op('med3', S[2], T[2], m)
    m = g.op('med3', S[1], T[3], m)
    return g.op('med3', S[0], S[5], m)

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

# ---- bit-parallel evaluation over all 2^25 assignments of a window's inputs
N = 25
FULL = (1 << (1 << N)) - 1
def var(k):
    x, size = ((1 << (1 << k)) - 1) << (1 << k), 1 << (k + 1)
    while size < (1 << N):
        x |= x << size; size *= 2
    return x

# ---- resynthesis with three-input cells: functional matching + minimum cover
import itertools, random
def ev(op, a):
    if op == 'min': return a[0] & a[1]
    if op == 'max': return a[0] | a[1]
    if op == 'min3': return a[0] & a[1] & a[2]
    if op == 'max3': return a[0] | a[1] | a[2]
    return (a[0] & a[1]) | (a[0] & a[2]) | (a[1] & a[2])
def truth_tables(cone):
    """exact functions of a window's nodes over its 25 inputs (2^25-bit integers)"""
    tt, k = {}, 0
    for x in cone:
        t = g.n[x]
        if t[0] == 'in': tt[x] = var(k); k += 1
        else: tt[x] = ev(t[0], [tt[y] for y in t[1:]])
    assert k == N
    return tt
def below(x, depth):
    seen, fr = {x}, [x]
    for _ in range(depth):
        nf = []
        for y in fr:
            if g.n[y][0] == 'in': continue
            for z in g.n[y][1:]:
                if z not in seen: seen.add(z); nf.append(z)
        fr = nf
    seen.discard(x)
    return sorted(seen)
def resynthesise(depth=3, nsample=4096):
    import numpy as np
    from scipy.optimize import milp, LinearConstraint, Bounds
    from scipy.sparse import lil_matrix
    rnd = random.Random(1)
    sig = {}                                   # every node on nsample random inputs: the cheap filter before the exact comparison
    for x, t in enumerate(g.n):
        sig[x] = rnd.getrandbits(nsample) if t[0] == 'in' else ev(t[0], [sig[y] for y in t[1:]])
    alts, done = {}, set()
    for m in meds:
        cone = sorted(g.live([m]))
        todo = [x for x in cone if g.n[x][0] != 'in' and x not in done]
        if not todo: continue
        tt = truth_tables(cone)
        for x in todo:
            done.add(x)
            cand, sx, found = below(x, depth), sig[x], [g.n[x]]
            for u, v in itertools.combinations(cand, 2):
                for op in ('min', 'max'):
                    if ev(op, (sig[u], sig[v])) == sx and ev(op, (tt[u], tt[v])) == tt[x]: found.append((op, u, v))
            for u, v, w in itertools.combinations(cand, 3):
                a, o = sig[u] & sig[v], sig[u] | sig[v]
                if a & sig[w] == sx: op = 'min3'
                elif o | sig[w] == sx: op = 'max3'
                elif a | (o & sig[w]) == sx: op = 'med3'
                else: continue
                if ev(op, (tt[u], tt[v], tt[w])) == tt[x]: found.append((op, u, v, w))
            alts[x] = list(dict.fromkeys(found))
    # 0-1 programme: y_x = node x is computed, z_(x,a) = by implementation a;  y_x = sum_a z_(x,a);  z_(x,a) <= y_u for every operand u
    ops = sorted(alts)
    yi = {x: i for i, x in enumerate(ops)}
    zs = [(x, a) for x in ops for a in alts[x]]
    ny, nz = len(ops), len(zs)
    nrow = ny + sum(1 for _, a in zs for u in a[1:] if g.n[u][0] != 'in')
    A, lo, hi, r = lil_matrix((nrow, ny + nz)), [], [], 0
    for x in ops: A[yi[x], yi[x]] = 1
    for k, (x, a) in enumerate(zs): A[yi[x], ny + k] = -1
    lo += [0] * ny; hi += [0] * ny; r = ny
    for k, (x, a) in enumerate(zs):
        for u in a[1:]:
            if g.n[u][0] == 'in': continue
            A[r, ny + k] = 1; A[r, yi[u]] = -1; lo.append(-np.inf); hi.append(0); r += 1
    lb, ub = np.zeros(ny + nz), np.ones(ny + nz)
    for m in meds: lb[yi[m]] = 1
    res = milp(np.concatenate([np.ones(ny), np.zeros(nz)]), constraints=LinearConstraint(A.tocsr(), lo, hi),
               integrality=np.ones(ny + nz), bounds=Bounds(lb, ub), options={"time_limit": 900})
    assert res.status == 0, res.message
    for k, (x, a) in enumerate(zs):
        if res.x[ny + k] > 0.5: g.n[x] = a
    return int(round(res.fun))
before = sum(1 for x in g.live(meds) if g.n[x][0] != 'in')
after = resynthesise()
print(f"resynthesis: {before} -> {after} operations")

# ---- exhaustive 0-1 verification of the final network
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

live = sorted(g.live(meds))
nops = sum(1 for x in live if g.n[x][0] != 'in')
name = {}
lines = ["// GENERATED by tools/gen_median_run.py (verified exhaustively there) -- do not edit.",
         f"// in:  float w[5][{NRUN + 4}] (rows x columns of the window);  out: float m0 .. m{NRUN - 1} = medians of columns 0-4, 1-5, ...",
         f"// {nops} min/max/min3/med3/max3 operations."]
needed = []
for x in live:
    t = g.n[x]
    if t[0] == 'in':
        name[x] = f"w[{t[1][1]}][{t[1][0]}]"
        continue
    name[x] = f"t{x}"
    a = [name[y] for y in t[1:]]
    for y in t[1:]:          # MED_NEED(c): column c of the window is read for the first time below (the kernel loads its windows in pieces)
        if g.n[y][0] == 'in' and g.n[y][1][0] not in needed:
            needed.append(g.n[y][1][0]); lines.append(f"MED_NEED({needed[-1]})")
    # MN2 / MX2 / MN3 / MX3 / MD3 are defined by the including file (k_ahd.hip: raw v_min / v_max / v_min3 / v_max3 / v_med3)
    expr = {'min': "MN2({}, {})", 'max': "MX2({}, {})", 'min3': "MN3({}, {}, {})", 'max3': "MX3({}, {}, {})",
            'med3': "MD3({}, {}, {})"}[t[0]].format(*a)
    lines.append(f"const float t{x} = {expr};")
for i, m in enumerate(meds):
    lines.append(f"m{i} = {name[m]};")
lines.insert(3, "// MED_NEED order of the columns: " + " ".join(map(str, needed)))
assert needed == [1, 2, 3, 4, 0] + list(range(5, NRUN + 4)), "k_ahd.hip loads the window in pieces keyed to this order"
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pysp_amd", "csrc", f"median25_run{NRUN}.inc")
open(out, "w").write("\n".join(lines) + "\n")
print("wrote", out, nops, "operations")
