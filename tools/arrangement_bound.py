"""How far is the bank-aware entry order of the packed 1x16 format (pk_arrange_kernel, aqlm_amd/csrc/gemv_packed.hip) from
what any order could reach?  CPU-only study on one synthetic wave range (random codes):

  * `greedy`      restates the device algorithm (slot by slot, rotating priority inside a 16-lane LDS service group);
  * `degree_bound` a lower bound no order can beat: inside a service group a bank group that occurs d times among the
                  group's 64*T entries needs d service cycles, so the 4*T reads of the group cost >= max(4*T, max_b d_b)
                  (the edge-colouring bound of the lane x bank-group multigraph; x and codebook sides taken separately,
                  row pools ignored -> optimistic);
  * `anneal`      simulated annealing over swaps inside a row's pool: what a search that costs ~1000 proposals per entry
                  finds.

Cost unit = LDS cycles per service group and ds_read_b128 (1.0 = conflict-free), the figure tools/conflict_report.py
measures on the device layout.  Result (T = steps per wave; 512 / 1024 input groups):

    T = 6  (4096 -> 11008):  no order 2.99 + 2.91 | greedy 2.18 + 1.70 | annealed 1.57 + 1.43 | bound 1.28 + 1.29
    T = 29 (8192 -> 28672):  no order 3.07 + 2.96 | greedy 2.07 + 1.53 |                      | bound 1.13 + 1.12

i.e. the greedy deal collects 60 % (T = 6) of what an ideal order could and a search another 25 %.  That search is built
into the prepack since round 3 (pk_improve_kernel: best-swap local search with plateau moves, 6 sweeps); on the device
layout it reaches codebook 1.53 + x 1.47 (tools/conflict_report.py).

    python tools/arrangement_bound.py [T] [in_groups] [anneal iterations]
"""
import math
import random
import sys

import numpy as np

G0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
G1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
GROUPS = [G0, G1, [l + 32 for l in G0], [l + 32 for l in G1]]  # MI355X_MICROARCH.md: LDS service groups of a ds_read_b128
GRP_OF = [0] * 64
for _gi, _g in enumerate(GROUPS):
    for _l in _g:
        GRP_OF[_l] = _gi
NULL_C, NULL_X = 0, 15  # bank groups of the null entry's two addresses (all nulls read the same slots: broadcast)


def make_wave(T, in_groups, rng, slices=16):
    """One wave range of a stream: c[T][64][4] codebook slots, x[T][64][4] input groups (-1 = null), row[T][64]."""
    n_ls = 64 * T
    c = -np.ones((n_ls, 4), int)
    x = -np.ones((n_ls, 4), int)
    row = np.zeros(n_ls, int)
    q = r = 0
    while q < n_ls:
        n = rng.binomial(in_groups, 1.0 / slices)
        js = np.sort(rng.choice(in_groups, n, replace=False))
        cs = rng.integers(0, 65536 // slices, n)
        steps = max(1, (n + 3) // 4)
        for i in range(steps * 4):
            qq = q + i // 4
            if qq >= n_ls:
                break
            row[qq] = r
            if i < n:
                c[qq, i % 4], x[qq, i % 4] = cs[i], js[i]
        q += steps
        r += 1
    lanes = lambda a: a.reshape(64, T, *a.shape[1:]).swapaxes(0, 1).copy()  # lane l owns lane-steps [l*T, l*T+T)
    return lanes(c), lanes(x), lanes(row)


def cost(c, x):
    T = c.shape[0]
    tot = np.zeros(2)
    for t in range(T):
        for k in range(4):
            for g in GROUPS:
                for i, f in enumerate((c, x)):
                    tot[i] += np.bincount(np.unique(f[t, g, k]) % 16, minlength=16).max()
    return tot / (16 * T)


def _group_lane(grp, pos):
    half, g1 = grp >> 1, grp & 1
    h = (pos if pos < 4 else (pos + 8 if pos < 8 else pos + 12)) if not g1 else (pos + 4 if pos < 8 else (pos + 8 if pos < 12 else pos + 16))
    return half * 32 + h


def greedy(c, x, row):
    T = c.shape[0]
    pools = {}
    for l in range(64):
        for t in range(T):
            for k in range(4):
                pools.setdefault(int(row[t, l]), []).append((int(c[t, l, k]), int(x[t, l, k])))
    nc, nx = np.empty_like(c), np.empty_like(x)
    for s in range(4 * T):
        t, k = s >> 2, s & 3
        for grp in range(4):
            ux = uc = 0
            null_placed = False
            for r in range(16):
                tl = _group_lane(grp, (r - s) & 15)
                pool = pools[int(row[t, tl])]
                best = bi = -1
                for i, (cc, xx) in enumerate(pool):
                    if cc < 0:
                        sc = 3 if (null_placed or (not (ux >> NULL_X) & 1 and not (uc >> NULL_C) & 1)) else 0
                    else:
                        cf, xf = not (uc >> (cc % 16)) & 1, not (ux >> (xx % 16)) & 1
                        sc = 4 if (xf and cf) else (2 if xf else (1 if cf else 0))
                    if sc > best:
                        best, bi = sc, i
                cc, xx = pool[bi]
                if cc >= 0:
                    ux |= 1 << (xx % 16)
                    uc |= 1 << (cc % 16)
                else:
                    ux |= 1 << NULL_X
                    uc |= 1 << NULL_C
                    null_placed = True
                nc[t, tl, k], nx[t, tl, k] = cc, xx
                pool[bi] = pool[-1]
                pool.pop()
    return nc, nx


def degree_bound(c, x):
    T = c.shape[0]
    out = []
    for f in (c, x):
        tot = 0
        for g in GROUPS:
            v = f[:, g, :].reshape(-1)
            tot += max(4 * T, np.bincount(v[v >= 0] % 16, minlength=16).max())
        out.append(tot / (16 * T))
    return np.array(out)


def anneal(c, x, row, iters, t_hot=0.6, t_cold=0.03, w_sq=0.1, seed=0):
    T = c.shape[0]
    rnd = random.Random(seed)
    cl, xl = c.tolist(), x.tolist()
    hc = [[[[0] * 16 for _ in range(4)] for _ in range(4)] for _ in range(T)]
    hx = [[[[0] * 16 for _ in range(4)] for _ in range(4)] for _ in range(T)]
    nulls = [[[0] * 4 for _ in range(4)] for _ in range(T)]

    def add(t, l, k, sgn):
        g = GRP_OF[l]
        cv = cl[t][l][k]
        if cv < 0:
            nulls[t][k][g] += sgn
            if (sgn > 0 and nulls[t][k][g] == 1) or (sgn < 0 and nulls[t][k][g] == 0):
                hc[t][k][g][NULL_C] += sgn
                hx[t][k][g][NULL_X] += sgn
        else:
            hc[t][k][g][cv % 16] += sgn
            hx[t][k][g][xl[t][l][k] % 16] += sgn

    def score(t, k, g):
        a, b = hc[t][k][g], hx[t][k][g]
        return max(a) + max(b) + w_sq * (sum(v * v for v in a) + sum(v * v for v in b))

    def swap(p, q):
        add(*p, -1)
        add(*q, -1)
        (t, l, k), (t2, l2, k2) = p, q
        cl[t][l][k], cl[t2][l2][k2] = cl[t2][l2][k2], cl[t][l][k]
        xl[t][l][k], xl[t2][l2][k2] = xl[t2][l2][k2], xl[t][l][k]
        add(*p, 1)
        add(*q, 1)

    pos, everything = {}, []
    for l in range(64):
        for t in range(T):
            for k in range(4):
                add(t, l, k, 1)
                pos.setdefault(int(row[t, l]), []).append((t, l, k))
                everything.append((t, l, k))
    for it in range(iters):
        temp = t_hot * (t_cold / t_hot) ** (it / iters)
        p = everything[rnd.randrange(len(everything))]
        mates = pos[int(row[p[0], p[1]])]
        q = mates[rnd.randrange(len(mates))]
        a, b = (p[0], p[2], GRP_OF[p[1]]), (q[0], q[2], GRP_OF[q[1]])
        if a == b:
            continue
        old = score(*a) + score(*b)
        swap(p, q)
        d = score(*a) + score(*b) - old
        if d > 0 and rnd.random() >= math.exp(-d / temp):
            swap(p, q)
    return np.array(cl), np.array(xl)


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    in_groups = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    c, x, row = make_wave(T, in_groups, np.random.default_rng(0))
    fmt = lambda v: f"codebook {v[0]:.2f} + x {v[1]:.2f} = {v.sum():.2f}"
    print(f"T = {T}, {in_groups} input groups; LDS cycles per service group and read")
    print("  ascending j (no arrangement):", fmt(cost(c, x)))
    gc, gx = greedy(c, x, row)
    print("  greedy (device algorithm):   ", fmt(cost(gc, gx)))
    if iters:
        ac, ax = anneal(gc, gx, row, iters)
        print(f"  annealed ({iters} proposals):", fmt(cost(ac, ax)))
    print("  degree bound:                ", fmt(degree_bound(c, x)))
