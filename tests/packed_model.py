"""numpy model of the packed 1x16 g8 format v7 (aqlm_amd/csrc/gemv_packed.hip): the specification that
aqlm_hip_prepack_1x16 is held to, plus a straight-line simulation of the kernel's traversal (column walk, flag masks,
LDS slots, carries).  Test infrastructure only.

Format v5 in one paragraph: the 65536-entry codebook is cut into S = 16 slices (code >> 12); the rows into NG = 16
row-groups of RG rows; workgroup (g, s) of the kernel owns *stream* (g, s) = every code of the group's rows whose slice is
s.  A row's codes of the slice are rounded up to whole *lane-steps* of 4 entries (at least one; null entries pad), and the
lane-steps of rows 0, 1, 2, ... are laid end to end.  That sequence is cut into NW wave ranges of 64*T lane-steps, each
wave range into 64 lane *columns* of T lane-steps: lane l of wave w reads lane-steps [(w*64 + l)*T, +T).  In memory
entry (w, t, l, k) sits at (((st*NW + w)*T + t)*64 + l)*4 + k, so step t of a wave is one contiguous KiB.  An entry is
(XB + j) << 16 | (code & 0xfff) with XB = 4096: both halves, shifted left by 4, are LDS byte addresses (slice at 0, x at
64 KiB).  Per lane column a bit mask marks the steps that END a row (mask word t/32, bit t%32), `frow` is the row (within
the group) the column's first lane-step belongs to, `rowstart[st][r]` is the first lane-step of row r in stream st, and per
wave `winfo` holds (first row that STARTS in the wave, wave starts inside a row, steps = steps that carry content).
(In the device buffer the mask and frow bits ride in the spare nibbles of the entries.)

Format v7 adds two pack-time balancing steps on top (both lossless): *relabelling* -- the 65536 codebook entries are dealt to the
slices by how often the layer uses them (`plan_relabel`: longest-processing-time greedy, 4096 entries per slice), the packed
entries carry the new labels, `old_of_new` travels with the buffer -- and *variable geometry* (`Geometry`): slice s owns n[s]
of the 256 workgroups instead of 16, its rows are split evenly over them, streams are numbered slice by slice
(`plan_geometry`: greedy by work per workgroup)."""
import numpy as np

S_LOG = 4
S = 1 << S_LOG
NG = 256 // S
CODE_BITS = 16 - S_LOG
SLICE_ENTRIES = 1 << CODE_BITS
XB = SLICE_ENTRIES
MAX_T = 128
MAGIC = 0x37505141  # "AQP7"
VERSION = 7
MIN_GROUPS = NG // 2


class Geometry:
    """Which slice and rows stream `st` owns.  groups=None: NG row groups of RG rows for every slice, stream = g * S + s."""

    def __init__(self, M, groups=None):
        self.M = M
        groups = [NG] * S if groups is None else [int(v) for v in groups[:S]]
        assert sum(groups) == NG * S and min(groups) >= 1
        self.n = groups
        self.vg = any(v != NG for v in groups)
        self.first = [0] * (S + 1)
        for s in range(S):
            self.first[s + 1] = self.first[s] + groups[s]
        self.RG = (M + min(groups) - 1) // min(groups) if self.vg else (M + NG - 1) // NG

    def group_rows(self, s, k):
        if not self.vg:
            row0 = k * self.RG
            return row0, max(0, min(self.RG, self.M - row0))
        base, extra = divmod(self.M, self.n[s])
        return k * base + min(k, extra), base + (1 if k < extra else 0)

    def stream(self, st):
        """-> (slice, first row, rows)"""
        if not self.vg:
            s, k = st % S, st // S
        else:
            s = max(i for i in range(S) if self.first[i] <= st)
            k = st - self.first[s]
        return (s,) + self.group_rows(s, k)

    def stream_of(self, s, k):
        return self.first[s] + k if self.vg else k * S + s


def wave_steps(L):
    """wave-steps a workgroup runs when its longest stream has L lane-steps: NW * T."""
    nw = choose_waves(L)
    return nw * ((L + 64 * nw - 1) // (64 * nw))


def balanced_enough(codes_unsigned):
    """The repack's first question: does the longest stream (checkpoint labels, 16 x 16 geometry) already run as few wave-steps
    as perfectly even streams (+3.5 %) would?  Then labels and geometry stay as they are."""
    ls, a = lane_steps(codes_unsigned)
    total = int(ls.sum())
    even = (total * 1035 + NG * S * 1000 - 1) // (NG * S * 1000)
    return wave_steps(int(a[:, -1].max())) <= wave_steps(even)


def plan_relabel(usage, force=False):
    """usage [65536] -> new_of_old [65536] (LPT greedy: heaviest entry first, to the lightest slice with room; ties: lower label,
    lower slice), or None when the checkpoint's labels already load the slices within 2 % of even -- unless `force`: then the deal
    is made anyway (equal counts fall out round-robin over the slices), which is what spreads labels whose use is correlated with
    the ROW: the global masses are even, one stream per row group is not."""
    usage = np.asarray(usage, dtype=np.int64)
    mass0 = usage.reshape(S, SLICE_ENTRIES).sum(axis=1)
    total = int(mass0.sum())
    if total == 0 or (not force and float(mass0.max()) * S <= 1.02 * float(total)):
        return None
    order = np.argsort(-usage, kind="stable")
    mass, cnt = [0] * S, [0] * S
    new_of_old = np.zeros(65536, dtype=np.int64)
    for c in order:
        best = -1
        for s in range(S):
            if cnt[s] < SLICE_ENTRIES and (best < 0 or mass[s] < mass[best]):
                best = s
        new_of_old[c] = best * SLICE_ENTRIES + cnt[best]
        cnt[best] += 1
        mass[best] += int(usage[c])
    return new_of_old


def rowblock_codes(M, in_groups, p, seed):
    """Label use correlated with the ROW: the rows of block b (16 equal blocks of rows) draw a fraction `p` of
    their codes from the labels [4096 b, 4096 (b + 1)) and the rest uniformly -- every entry is used equally often over the layer,
    so no global histogram sees it, while one stream per row group carries most of that group's codes."""
    rng = np.random.default_rng(seed)
    blk = (np.arange(M) * S // M)[:, None]
    own = blk * SLICE_ENTRIES + rng.integers(0, SLICE_ENTRIES, size=(M, in_groups))
    uni = rng.integers(0, 65536, size=(M, in_groups))
    return np.where(rng.random((M, in_groups)) < p, own, uni).astype(np.int64)


def plan_labels(codes_unsigned):
    """The repack's decision about the labels, as aqlm_hip_prepack_1x16 takes it: balanced layout -> the checkpoint's labels (None);
    else the forced deal, kept only if the longest stream then runs fewer wave-steps."""
    if balanced_enough(codes_unsigned):
        return None
    new = plan_relabel(np.bincount(codes_unsigned.ravel(), minlength=65536), force=True)
    if new is None:
        return None
    _, a0 = lane_steps(codes_unsigned)
    _, a1 = lane_steps(new[codes_unsigned])
    if wave_steps(int(a1[:, -1].max())) >= wave_steps(int(a0[:, -1].max())):
        return None
    return new


def plan_geometry(slice_steps, M, min_groups=MIN_GROUPS):
    """lane-steps per slice -> workgroups per slice: `min_groups` each, the rest one by one to the slice whose groups carry the
    most work (lane-steps + a quarter step per row); uniform unless that shortens the longest stream by more than 6 %.
    (The library also keeps the row groups small enough for multi-row launches: it may start from a larger `min_groups`.)"""
    w = [float(v) + 0.25 * M for v in slice_steps]
    n = [min_groups] * S
    for _ in range(NG * S - min_groups * S):
        best = 0
        for s in range(1, S):
            if w[s] / n[s] > w[best] / n[best]:
                best = s
        n[best] += 1
    uni = max(v / NG for v in w)
    var = max(w[s] / n[s] for s in range(S))
    if all(v == NG for v in n) or var > 0.94 * uni or M < 512:
        return [NG] * S
    return n


def align_up(v, a):
    return (v + a - 1) // a * a


def choose_waves(max_lane_steps):
    q = (max_lane_steps + 63) // 64  # wave-steps of work per workgroup
    if q >= 48:  # long streams (entry-stream bound): 16 waves; mid-size layers: 14 (room for the pipelined kernel's DMA waves)
        return 16 if (q + 15) // 16 >= 13 else 14
    best, best_cost = 8, 1 << 30
    for nw in range(8, 3, -1):
        t = (q + nw - 1) // nw
        cost = (nw * t - q) * 8 + (8 - nw) + (12 if (t >= 3 and t % 3 != 0) else 0)
        if cost < best_cost:
            best, best_cost = nw, cost
    return best


def layout(M, in_groups, NW, T, entry_bytes=4, groups=None, relabel=False):
    RG = Geometry(M, groups).RG
    nst = NG * S
    off_winfo = 256
    off_rowstart = align_up(off_winfo + nst * 16 * 16, 256)
    off_acc = align_up(off_rowstart + nst * (RG + 1) * 4, 256)      # [8][M] u64 accumulator cells of the fused finalize, zero at rest
    off_ent = align_up(off_acc + 8 * M * 8, 1024)
    ent_bytes = nst * NW * T * (776 if entry_bytes == 3 else 1024)
    off_perm = align_up(off_ent + ent_bytes, 1024)     # relabelled: u16 old_of_new[65536], then the codebook image [65536][8] halfs
    off_cb = off_perm + 65536 * 2
    return dict(RG=RG, off_winfo=off_winfo, off_rowstart=off_rowstart, off_acc=off_acc, off_ent=off_ent, ent_bytes=ent_bytes,
                off_perm=off_perm, off_cb=off_cb, used=(off_cb + 65536 * 16) if relabel else off_ent + ent_bytes)


def slice_steps(codes_unsigned):
    """lane-steps per slice (what plan_geometry is fed with)."""
    sl = codes_unsigned >> CODE_BITS
    return [int(np.maximum(1, ((sl == s).sum(axis=1) + 3) // 4).sum()) for s in range(S)]


def lane_steps(codes_unsigned, geom=None):
    """[NG*S, RG] lane-steps per (stream, row) and their exclusive prefix sums [NG*S, RG+1]."""
    M, in_groups = codes_unsigned.shape
    geom = Geometry(M) if geom is None else geom
    RG = geom.RG
    sl = codes_unsigned >> CODE_BITS
    ls = np.zeros((NG * S, RG), dtype=np.int64)
    per_slice = [np.maximum(1, ((sl == s).sum(axis=1) + 3) // 4) for s in range(S)]
    for st in range(NG * S):
        s, row0, nrows = geom.stream(st)
        ls[st, :nrows] = per_slice[s][row0:row0 + nrows]
    a = np.zeros((NG * S, RG + 1), dtype=np.int64)
    a[:, 1:] = np.cumsum(ls, axis=1)
    return ls, a


def pack(codes_unsigned, NW=None, groups=None, new_of_old=None):
    """codes_unsigned [M, in_groups] ints in [0, 65536) -> dict with the arrays of the packed buffer.  `groups`: workgroups per
    slice (None: uniform); `new_of_old`: relabelling (None: the checkpoint's labels)."""
    M, in_groups = codes_unsigned.shape
    geom = Geometry(M, groups)
    RG = geom.RG
    if new_of_old is not None:
        codes_unsigned = np.asarray(new_of_old, dtype=np.int64)[codes_unsigned]
    ls, a = lane_steps(codes_unsigned, geom)
    maxL = int(a[:, RG].max())
    if NW is None:
        NW = choose_waves(maxL)
    T = (maxL + 64 * NW - 1) // (64 * NW)
    assert 1 <= T <= MAX_T
    MW = (T + 31) // 32
    nst = NG * S
    null = np.uint32((XB + in_groups) << 16)
    ent = np.full((nst, NW, T, 64, 4), null, dtype=np.uint32)
    mask = np.zeros((nst, NW, MW, 64), dtype=np.uint32)
    frow = np.zeros((nst, NW, 64), dtype=np.uint16)
    winfo = np.zeros((nst, NW, 4), dtype=np.uint32)
    for st in range(nst):
        s, row0, nrows = geom.stream(st)
        total = int(a[st, nrows])
        for r in range(nrows):
            row = codes_unsigned[row0 + r]
            js = np.nonzero((row >> CODE_BITS) == s)[0]
            e = ((js.astype(np.uint32) + XB) << 16) | (row[js].astype(np.uint32) & (SLICE_ENTRIES - 1))
            q0 = int(a[st, r])
            for i, v in enumerate(e):
                q, k = q0 + i // 4, i % 4
                w, rem = divmod(q, 64 * T)
                l, t = divmod(rem, T)
                ent[st, w, t, l, k] = v
            ql = q0 + int(ls[st, r]) - 1  # the row's last lane-step carries the flag
            w, rem = divmod(ql, 64 * T)
            l, t = divmod(rem, T)
            mask[st, w, t // 32, l] |= np.uint32(1 << (t % 32))
        starts = a[st, :nrows + 1]  # starts[r] for r < nrows, starts[nrows] = total
        for w in range(NW):
            w0 = w * 64 * T
            wfr = int(np.searchsorted(starts[:nrows], w0, side="left"))  # first row starting at or after w0
            cont = 1 if (w0 < total and wfr <= nrows and int(starts[wfr]) > w0) else 0
            if w0 >= total:
                steps = 0
            elif total - w0 >= T:
                steps = T
            else:
                steps = total - w0
            winfo[st, w] = (wfr, cont, steps, 0)
            for l in range(64):
                q = w0 + l * T
                r0 = nrows if q >= total else int(np.searchsorted(starts[:nrows], q, side="right")) - 1
                frow[st, w, l] = r0
    old_of_new = None
    if new_of_old is not None:
        old_of_new = np.zeros(65536, dtype=np.int64)
        old_of_new[np.asarray(new_of_old, dtype=np.int64)] = np.arange(65536)
    return dict(M=M, in_groups=in_groups, NW=NW, T=T, MW=MW, RG=RG, ent=ent, mask=mask, frow=frow, winfo=winfo,
                rowstart=a.astype(np.uint32), geom=geom, old_of_new=old_of_new)


def walk(P):
    """Yield (st, w, l, t, row_in_group_or_None, entries[4]) following the kernel's column walk.  `row` is the row the
    lane-step belongs to (None for trailing null steps)."""
    NW, T, RG, M = P["NW"], P["T"], P["RG"], P["M"]
    geom = P.get("geom") or Geometry(M)
    for st in range(NG * S):
        _, _, nrows = geom.stream(st)
        for w in range(NW):
            wfr, cont, steps, _ = (int(v) for v in P["winfo"][st, w])
            for l in range(64):
                row = int(P["frow"][st, w, l])
                for t in range(steps):
                    yield st, w, l, t, (row if row < nrows else None), P["ent"][st, w, t, l]
                    if (int(P["mask"][st, w, t // 32, l]) >> (t % 32)) & 1:
                        row += 1


def unpack(P):
    """Reconstruct the canonical codes [M, in_groups] from the packed arrays (lossless)."""
    M, in_groups, RG = P["M"], P["in_groups"], P["RG"]
    geom = P.get("geom") or Geometry(M)
    out = np.full((M, in_groups), -1, dtype=np.int64)
    for st, w, l, t, row, e in walk(P):
        s, row0, _ = geom.stream(st)
        for v in e:
            j = (int(v) >> 16) - XB
            if j == in_groups:
                continue
            assert row is not None and 0 <= j < in_groups
            assert out[row0 + row, j] == -1
            out[row0 + row, j] = (s << CODE_BITS) | (int(v) & 0xFFFF)
    assert (out >= 0).all()
    if P.get("old_of_new") is not None:   # relabelled: back to the checkpoint's labels
        out = np.asarray(P["old_of_new"], dtype=np.int64)[out]
    return out


def simulate(P, codebook, x):
    """The kernel's arithmetic in float64 with its exact bookkeeping: a lane adds up its column; at a row end it
    stores the sum to rowval[row] (one writer per row) and starts over; what is left at the end of the column goes to
    colend[column]; row r = rowval[r] + colend of the columns it crosses.  codebook [65536, 8], x [B, in_features] ->
    y [B, M] (unscaled).  A relabelled P reads the permuted codebook image, like the kernel."""
    M, in_groups, NW, T, RG = P["M"], P["in_groups"], P["NW"], P["T"], P["RG"]
    geom = P.get("geom") or Geometry(M)
    if P.get("old_of_new") is not None:
        codebook = codebook[np.asarray(P["old_of_new"], dtype=np.int64)]
    B = x.shape[0]
    xg = np.concatenate([x.reshape(B, in_groups, 8), np.zeros((B, 1, 8))], axis=1)
    y = np.zeros((B, M))
    for st in range(NG * S):
        s, row0, nrows = geom.stream(st)
        rowval = np.full((B, RG + 1), np.nan)
        colend = np.zeros((B, NW * 64))
        for w in range(NW):
            steps = int(P["winfo"][st, w, 2])
            for l in range(64):
                row = int(P["frow"][st, w, l])
                acc = np.zeros(B)
                for t in range(steps):
                    for v in P["ent"][st, w, t, l]:
                        c, j = int(v) & 0xFFFF, (int(v) >> 16) - XB
                        acc += xg[:, j] @ codebook[(s << CODE_BITS) | c]
                    if (int(P["mask"][st, w, t // 32, l]) >> (t % 32)) & 1:
                        assert np.isnan(rowval[0, row])  # exactly one writer per row
                        rowval[:, row] = acc
                        acc = np.zeros(B)
                        row += 1
                colend[:, w * 64 + l] = acc
        rs = P["rowstart"][st].astype(np.int64)
        for r in range(nrows):
            c0, c1 = rs[r] // T, (rs[r + 1] - 1) // T
            v = rowval[:, r].copy()
            for c in range(c0, c1):
                v += colend[:, c]
            y[:, row0 + r] += v
    assert not np.isnan(y).any()
    return y


SERVICE_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                  [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
SERVICE_GROUPS += [[l + 32 for l in g] for g in SERVICE_GROUPS]


def conflict_cycles(P, max_streams=6):
    """Mean LDS cycles per 16-lane service group and ds_read_b128 (1.0 = conflict-free) over the steps that carry
    content: max number of DISTINCT 16-B slots that fall into one bank group (slot % 16), for the x reads and the
    codebook reads (identical slots -- the null entries -- are broadcast)."""
    ent, in_groups = P["ent"], P["in_groups"]
    tot, n = 0.0, 0
    for st in range(0, ent.shape[0], max(1, ent.shape[0] // max_streams)):
        for w in range(P["NW"]):
            steps = int(P["winfo"][st, w, 2])
            for t in range(steps):
                for k in range(4):
                    v = ent[st, w, t, :, k].astype(np.int64)
                    for field in ((v >> 16) - XB, v & 0xFFFF):
                        for grp in SERVICE_GROUPS:
                            slots = np.unique(field[grp])
                            tot += np.bincount(slots % 16, minlength=16).max()
                            n += 1
    return tot / max(n, 1)


def x_stride(in_groups):
    return ((in_groups + 1 + 11) & ~15) + 4


def decode_device_buffer(raw, M, in_features, NW, T, entry_bytes, groups=None, relabel=False):
    """Bytes of a packed buffer (either entry width) -> the arrays of `pack` as the DEVICE holds them: dict with winfo,
    rowstart, mask [nst, NW, 1, 64] (T <= 32), frow [nst, NW, 64], slot / code [nst, NW, T, 64, 4] (slot = x slot incl.
    the copy offset), plus the derived j / copy; relabelled buffers also give old_of_new [65536] and the codebook image."""
    in_groups = in_features // 8
    lay = layout(M, in_groups, NW, T, entry_bytes, groups, relabel)
    assert raw.size == lay["used"], (raw.size, lay["used"])
    nst = NG * S
    winfo = raw[lay["off_winfo"]:lay["off_winfo"] + nst * NW * 16].view(np.uint32).reshape(nst, NW, 4)
    rowstart = raw[lay["off_rowstart"]:lay["off_rowstart"] + nst * (lay["RG"] + 1) * 4].view(np.uint32).reshape(nst, lay["RG"] + 1)
    region = raw[lay["off_ent"]:lay["off_ent"] + lay["ent_bytes"]]
    lanes = np.arange(64, dtype=np.uint64)
    if entry_bytes == 4:
        ent = region.view(np.uint32).reshape(nst, NW, T, 64, 4)
        slot, code = ent >> 20, (ent >> 4) & 0xFFF
        flags = (ent[..., 0] & 1).astype(np.uint32)                                    # [nst, NW, T, 64]
        e0, e1, e2, e3 = (ent[:, :, 0, :, k].astype(np.uint32) for k in range(4))
        frow = ((e0 >> 1) & 7) | ((e1 & 15) << 3) | ((e2 & 15) << 7) | ((e3 & 15) << 11)
        spare_ok = int(((ent >> 18) & 3).max()) == 0
        copy_bits = (ent >> 16) & 3
    else:
        region = region.reshape(nst, NW, T * 776)
        words = np.ascontiguousarray(region[:, :, :8 * T]).view(np.uint64).reshape(nst, NW, T)
        w = np.ascontiguousarray(region[:, :, 8 * T:]).view(np.uint32).reshape(nst, NW, T, 64, 3).astype(np.uint64)
        w0, w1, w2 = w[..., 0], w[..., 1], w[..., 2]
        e = np.stack([w0 & 0xFFFFFF, (w0 >> 24) | ((w1 & 0xFFFF) << 8), (w1 >> 16) | ((w2 & 0xFF) << 16), w2 >> 8], axis=-1)
        slot, code = (e >> 12).astype(np.uint32), (e & 0xFFF).astype(np.uint32)
        flags = ((words[..., None] >> lanes) & np.uint64(1)).astype(np.uint32)             # [nst, NW, T, 64]
        below = (np.uint64(1) << lanes) - np.uint64(1)
        pc = np.zeros((nst, NW, 64), dtype=np.int64)
        for t in range(T):
            m = words[:, :, t, None] & below
            pc += np.array([bin(int(v)).count("1") for v in m.reshape(-1)], dtype=np.int64).reshape(nst, NW, 64)
        frow = (winfo[:, :, 3, None].astype(np.int64) + pc).astype(np.uint32)
        spare_ok, copy_bits = True, None
    mask = np.zeros((nst, NW, 1, 64), dtype=np.uint32)
    for t in range(T):
        mask[:, :, 0, :] |= flags[:, :, t, :] << np.uint32(t)
    stride = x_stride(in_groups)
    copy = slot // stride
    j = slot - copy * stride
    if copy_bits is not None:
        spare_ok = spare_ok and bool((copy_bits == copy).all())
    out = dict(winfo=winfo, rowstart=rowstart, mask=mask, frow=frow, slot=slot, code=code, j=j, copy=copy, spare_ok=spare_ok)
    if relabel:
        out["old_of_new"] = raw[lay["off_perm"]:lay["off_perm"] + 65536 * 2].view(np.uint16).astype(np.int64)
        out["codebook_image"] = raw[lay["off_cb"]:lay["off_cb"] + 65536 * 16].view(np.uint16).reshape(65536, 8)
    return out
