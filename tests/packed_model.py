"""numpy model of the packed 1x16 format v4 (aqlm_amd/csrc/gemv_packed.hip): the bit-exact oracle for
aqlm_hip_prepack_1x16.  Test infrastructure only."""
import numpy as np

S, NG, PAD = 8, 32, 128
XBASE = 8192   # x[j] sits at LDS slot XBASE + j: the high half of an entry is that slot index


def align_up(v, a):
    return (v + a - 1) // a * a


def layout(out_features, in_features):
    in_groups = in_features // 8
    RG = ((out_features + NG - 1) // NG + 3) // 4 * 4
    n_rowoff = NG * S * (RG + 1)
    entries = out_features * in_groups + 3 * S * out_features  # capacity incl. null padding
    off_rowoff = 256
    n_perm = NG * S * RG
    off_perm = align_up(off_rowoff + n_rowoff * 4, 256)
    off_ent = align_up(off_perm + n_perm * 2, 256)
    total = align_up(off_ent + (entries + PAD) * 4, 256)
    return dict(in_groups=in_groups, RG=RG, n_rowoff=n_rowoff, n_perm=n_perm, entries=entries, off_rowoff=off_rowoff,
                off_perm=off_perm, off_ent=off_ent, total=total)


def pack(codes_unsigned):
    """codes_unsigned: [M, in_groups] ints in [0, 65536).  Returns (rowoff u32, rowperm u16, entries u32, layout).
    Inside a stream (g, s) the buckets are ordered by padded size, largest first, ties by row index."""
    M, in_groups = codes_unsigned.shape
    L = layout(M, in_groups * 8)
    RG = L["RG"]
    counts = np.zeros((NG, S, RG + 1), dtype=np.int64)
    sl = codes_unsigned >> 13
    for s in range(S):
        c = ((sl == s).sum(axis=1) + 3) // 4 * 4  # buckets padded to multiples of 4 entries
        for r in range(M):
            counts[r // RG, s, r % RG] = c[r]
    perm = np.zeros((NG, S, RG), dtype=np.uint16)   # position -> row
    rank = np.zeros((NG, S, RG), dtype=np.int64)    # row -> position
    for g in range(NG):
        for s in range(S):
            order = sorted(range(RG), key=lambda r: (-counts[g, s, r], r))
            perm[g, s] = order
            rank[g, s, order] = np.arange(RG)
            counts[g, s, :RG] = counts[g, s, order]
    flat = counts.reshape(-1)
    rowoff = np.concatenate([[0], np.cumsum(flat)[:-1]]).astype(np.uint32)
    ent = np.zeros(L["entries"], dtype=np.uint32)
    ro = rowoff.reshape(NG, S, RG + 1)
    for r in range(M):
        g, rl = divmod(r, RG)
        row = codes_unsigned[r]
        for s in range(S):
            js = np.nonzero((row >> 13) == s)[0]
            e = ((js.astype(np.uint32) + XBASE) << 16) | (row[js].astype(np.uint32) & 0x1FFF)
            b = int(ro[g, s, rank[g, s, rl]])
            pad = (-len(js)) % 4
            null = np.uint32((in_groups + XBASE) << 16)  # j = in_groups, code 0
            ent[b:b + len(e) + pad] = arrange(e, len(e) + pad, null)
    # K4b: per pair of neighbouring positions, order each lane's four entries over the levels (codebook bank conflicts)
    for g in range(NG):
        for s in range(S):
            for i in range(RG // 2):
                order_levels(ent, [int(ro[g, s, 2 * i]), int(ro[g, s, 2 * i + 1])],
                             [int(ro[g, s, 2 * i + 1] - ro[g, s, 2 * i]), int(ro[g, s, 2 * i + 2] - ro[g, s, 2 * i + 1])])
    return rowoff, perm.reshape(-1), ent, L


import itertools  # noqa: E402

PERMS = list(itertools.permutations(range(4)))  # lexicographic, like the table in prepack_level_kernel


def order_levels(ent, beg, length):
    """prepack_level_kernel for one pair of buckets: greedy choice, lane by lane (even bucket, then odd), of the first of
    the 24 orders of a lane's entries that adds the fewest codebook-residue collisions to its service group."""
    for region in (0, 64):
        used = [[0] * 4, [0] * 4]
        for lane in range(16):
            for b in range(2):
                off = region + 4 * lane
                if off + 4 > length[b]:
                    continue
                ev = [int(v) for v in ent[beg[b] + off: beg[b] + off + 4]]
                in_x = lane < 4 or lane >= 12
                grp = 0 if (in_x == (b == 0)) else 1
                best, best_cost = 0, 5
                for q, pm in enumerate(PERMS):
                    cost = sum((used[grp][k] >> (ev[pm[k]] & 15)) & 1 for k in range(4))
                    if cost < best_cost:
                        best, best_cost = q, cost
                o = [ev[PERMS[best][k]] for k in range(4)]
                ent[beg[b] + off: beg[b] + off + 4] = o
                for k in range(4):
                    used[grp][k] |= 1 << (o[k] & 15)


def home_lane(rho):
    return rho if rho < 4 else (rho + 8 if rho < 8 else rho - 4)


def arrange(entries, slots, null):
    """Bank-aware order inside a bucket (prepack_arrange_kernel): an entry whose x slot has residue rho = j mod 16 goes
    to its home lane (index 4*lane + level) while that lane exists and has a free level; the others fill the holes in
    index order; what is left is null padding."""
    EMPTY = 0xFFFFFFFF
    out = np.full(slots, EMPTY, dtype=np.uint32)
    m = min(16, slots // 4)
    cnt = [0] * 16
    rest = []
    for e in entries:
        L = home_lane((int(e) >> 16) & 15)
        if L < m and cnt[L] < 4:
            out[4 * L + cnt[L]] = e
            cnt[L] += 1
        else:
            rest.append(e)
    idx = 0
    for e in rest:
        while out[idx] != EMPTY:
            idx += 1
        out[idx] = e
        idx += 1
    out[out == EMPTY] = null
    return out
