"""numpy model of the packed 1x16 format v2 (aqlm_amd/csrc/gemv_packed.hip): the bit-exact oracle for
aqlm_hip_prepack_1x16.  Test infrastructure only."""
import numpy as np

S, NG, PAD = 8, 32, 128


def align_up(v, a):
    return (v + a - 1) // a * a


def layout(out_features, in_features):
    in_groups = in_features // 8
    RG = ((out_features + NG - 1) // NG + 3) // 4 * 4
    n_rowoff = NG * S * (RG + 1)
    entries = out_features * in_groups + 3 * S * out_features  # capacity incl. null padding
    off_rowoff = 256
    off_lo16 = align_up(off_rowoff + n_rowoff * 4, 256)
    off_hi8 = align_up(off_lo16 + (entries + PAD) * 2, 256)
    total = align_up(off_hi8 + entries + PAD, 256)
    return dict(in_groups=in_groups, RG=RG, n_rowoff=n_rowoff, entries=entries, off_rowoff=off_rowoff,
                off_lo16=off_lo16, off_hi8=off_hi8, total=total)


def pack(codes_unsigned):
    """codes_unsigned: [M, in_groups] ints in [0, 65536).  Returns (rowoff u32, lo16 u16, hi8 u8) of `entries` length."""
    M, in_groups = codes_unsigned.shape
    L = layout(M, in_groups * 8)
    RG = L["RG"]
    counts = np.zeros((NG, S, RG + 1), dtype=np.int64)
    sl = codes_unsigned >> 13
    for s in range(S):
        c = ((sl == s).sum(axis=1) + 3) // 4 * 4  # buckets padded to multiples of 4 entries
        for r in range(M):
            counts[r // RG, s, r % RG] = c[r]
    flat = counts.reshape(-1)
    rowoff = np.concatenate([[0], np.cumsum(flat)[:-1]]).astype(np.uint32)
    lo16 = np.zeros(L["entries"], dtype=np.uint16)
    hi8 = np.zeros(L["entries"], dtype=np.uint8)
    ro = rowoff.reshape(NG, S, RG + 1)
    for r in range(M):
        g, rl = divmod(r, RG)
        row = codes_unsigned[r]
        for s in range(S):
            js = np.nonzero((row >> 13) == s)[0]
            e = (js.astype(np.uint32) << 13) | (row[js].astype(np.uint32) & 0x1FFF)
            b = int(ro[g, s, rl])
            pad = (-len(js)) % 4
            e = np.concatenate([e, np.full(pad, in_groups << 13, dtype=np.uint32)])  # null entries: j = in_groups
            lo16[b:b + len(e)] = (e & 0xFFFF).astype(np.uint16)
            hi8[b:b + len(e)] = (e >> 16).astype(np.uint8)
    return rowoff, lo16, hi8, L
