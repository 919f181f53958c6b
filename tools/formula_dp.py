"""Cheapest formula (tree, no sharing) for the rank-th smallest of two SORTED lists over {min, max, min3, max3, med3}, by dynamic programming over
the monotone Boolean functions of the abstract domain (number of ones in list a, number of ones in list b): python tools/formula_dp.py NA NB RANK.
`python tools/formula_dp.py 6 5 6` finds the five nested med3 that finish every window of the 5x5 median network
(tools/gen_median_run.py::sel): med3(a1, a6, med3(a2, b4, med3(a3, b3, med3(a4, b2, med3(a5, b1, b5))))), in a second."""
import sys, itertools
def solve(na, nb, rank, maxcost=9):
    # domain: (i, j) = number of ones in a (sorted ascending: ones at the top), in b
    dom = [(i, j) for i in range(na + 1) for j in range(nb + 1)]
    idx = {d: k for k, d in enumerate(dom)}
    def fn(pred):
        v = 0
        for d in dom:
            if pred(*d): v |= 1 << idx[d]
        return v
    # a_r (r = 1..na, r-th smallest) is 1 iff ones_a >= na - r + 1
    inputs = {}
    for r in range(1, na + 1): inputs[fn(lambda i, j, r=r: i >= na - r + 1)] = f"a{r}"
    for r in range(1, nb + 1): inputs[fn(lambda i, j, r=r: j >= nb - r + 1)] = f"b{r}"
    # target: rank-th smallest of union is 1 iff ones >= na + nb - rank + 1
    target = fn(lambda i, j: i + j >= na + nb - rank + 1)
    cost = {f: 0 for f in inputs}
    expr = dict(inputs)
    bylevel = {0: list(inputs)}
    for c in range(1, maxcost + 1):
        new = {}
        # two-input ops: costs c1 + c2 = c - 1
        for c1 in range(0, c):
            c2 = c - 1 - c1
            if c2 < c1: break
            for f in bylevel.get(c1, []):
                for g in bylevel.get(c2, []):
                    for op, h in (("min", f & g), ("max", f | g)):
                        if h not in cost and h not in new: new[h] = f"{op}({expr[f]},{expr[g]})"
        # three-input ops
        for c1 in range(0, c):
            for c2 in range(c1, c):
                c3 = c - 1 - c1 - c2
                if c3 < c2: break
                for f in bylevel.get(c1, []):
                    for g in bylevel.get(c2, []):
                        a, o = f & g, f | g
                        for h in bylevel.get(c3, []):
                            for op, v in (("min3", a & h), ("max3", o | h), ("med3", a | (o & h))):
                                if v not in cost and v not in new: new[v] = f"{op}({expr[f]},{expr[g]},{expr[h]})"
        for f, e in new.items(): cost[f] = c; expr[f] = e
        bylevel[c] = list(new)
        print("cost", c, "new functions", len(new), "total", len(cost), flush=True)
        if target in cost: break
    return cost.get(target), expr.get(target)
if __name__ == "__main__":
    na, nb, rank = map(int, sys.argv[1:4])
    print(solve(na, nb, rank))
