// 1x16 g8 matvec (1..8 input rows) on slice-bucketed ("prepacked") codes, gfx950.  Packed format v7.
//
// Why a load-time repack: on MI355X a random 16-B codebook gather that hits L2 costs a whole 128-B line of the CU's
// L1-fill path (0.43 lane-gathers/clk/CU measured, profiles/r01_call1_mb_l2gather.log), which pins the direct kernel
// (gemv.hip) at ~5.5 % of the HBM roofline.  LDS gathers are >10x cheaper but only 160 KiB fit per CU, so the codebook is
// cut into S = 16 slices of 4096 entries (64 KiB), one slice per workgroup -- and every workgroup must find "its" codes.
// Bucketing the codes by slice ONCE, when the layer is loaded, removes that search.  (The reference also re-lays codes
// out at load time for its CPU kernel, inference.py:78-83.)
//
// Format v7 = format v6 + balancing at pack time (round 5; the reference's kernels are data-oblivious -- one warp per
// output row, cuda_kernel.cu:16-27 -- while a slice-bucketed kernel is only as fast as its fullest slice):
//   * relabelling: the 65536 codebook entries are dealt to the slices by how often the layer uses them (LPT greedy), so the
//     slices carry equal numbers of codes whatever the checkpoint's labelling; the permutation (unpack stays bit-exact) and a
//     permuted image of the codebook (what the kernels read) live behind the entries.  Evenly used codebooks keep their labels.
//   * variable geometry (16-B vectors): an entry used by more than 1/16 of the codes cannot be balanced by labels; the 256
//     workgroups are then dealt to the slices in proportion to their work: slice s gets n_s row groups of M / n_s rows
//     (stream index = slice-major; block -> stream keeps an XCD on 32 consecutive streams, i.e. on 2-3 slices).
// Format v6 (built by aqlm_hip_prepack_1x16; specification + simulation: tests/packed_model.py):
//   rows -> NG = 16 row-groups of RG rows; codes -> S = 16 slices by (code >> 12); workgroup (g, s) owns stream (g, s).
//   In a stream every row's codes of the slice are rounded up to whole LANE-STEPS of 4 entries (>= 1; null entries pad)
//   and the lane-steps of rows 0, 1, 2, ... are laid end to end.  The sequence is cut into NW wave ranges of 64*T
//   lane-steps, a wave range into 64 lane COLUMNS of T lane-steps: lane l of wave w walks lane-steps
//   [(w*64 + l)*T, +T).  Entry (w, t, l, k) is stored at (((st*NW + w)*T + t)*64 + l)*4 + k, so step t of a wave is ONE
//   contiguous KiB at an address that depends on nothing but (st, w, t): no bucket table, no dependent round trip
//   before the first code arrives (format v4 needed rowoff -> entries).
//   entry = (j << 4 | fhi) << 16 | (code & 0xfff) << 4 | flo: `half & 0xfff0` IS the LDS byte offset of x[j] resp. of
//   the codebook vector (one v_and_b32_sdwa each).  The spare nibbles carry the bookkeeping: bit 0 of a lane-step's
//   first entry = "a row ends with this lane-step"; the first lane-step of every column carries the column's starting
//   slot (15 bits over the remaining spare bits of entries 0 and 1).  null entry: j = in_groups (a zero vector in LDS).
//   winfo[st][w] = {first row that STARTS in wave w, wave starts inside a row, steps with content, start row of lane 0}
//   (read by the 3-byte-entry kernel and the unpacker; the 4-byte kernel runs all T steps of every wave range -- the
//   tail is null entries -- and finds the start rows in the entries).  rowstart[st][RG + 1] = first lane-step of every
//   row of the stream (epilogue).  Optional 3-byte entries (entry_bytes = 3, T <= 32): a wave range = T 8-byte row-end
//   flag words, then T steps of 64 x 12 B holding 4 x (slot:12 | code:12); 3.5 instead of 4.5 B per code, same speed.
//   4 bytes per code + 6 bytes of padding per (row, slice) on average + < 1 step per wave of tail: 4.5 B per code for
//   4096-wide layers, 4.2 for 8192-wide ones (canonical: 2 B per code).
//
// Kernel: grid = 256 workgroups = 16 groups x 16 slices (slice = block % 16 -> the two slices block % 8 and
// block % 8 + 8 live in one XCD's L2).  Slice, x and the stream's row-start table go to LDS by LDS-DMA
// (global_load_lds: no VGPR staging, no ds_write pass); the entry stream runs PD steps ahead in a register ring from
// fixed addresses, so it is in flight before the LDS fill completes.  A lane accumulates its column in fp32; at the
// lane-step that ends a row it STORES the sum to rowval[row] (exactly one lane-step per row does: unique writer, no
// atomics), at the end of its column it stores what it gathered after its last row end to colend[column].  Epilogue:
// row r = rowval[r] + the colend of the columns the row crosses before its last one, in column order (from the
// row-start table) -> the summation order inside a workgroup is fixed.  The 16 slice sums of a row then meet in one 64-bit
// cell of the packed buffer: fixed-point integers added with ONE returning atomic each (order-independent, hence
// deterministic); the workgroup that finds 15 earlier arrivals applies scale + bias, rounds once, writes y and zeroes
// the cell.  The fixed-point unit comes from max|x| (taken from the x image in LDS) and the layer's codebook range
// (descriptor): no overflow whatever the data.  Without a codebook range the kernel leaves fp32 partials
// [slice][batch][row] in a workspace and a second kernel finalizes.  Per entry: 2 v_and_sdwa + 2 ds_read_b128 + 4 v_dot2c (x B for B
// input rows: one codebook read, B x reads).  The kernels take their leading parameters as scalar arguments: the
// command processor preloads them into SGPRs (-amdgpu-kernarg-preload-count), so no kernel-argument fetch precedes the
// first load.
#include <algorithm>
#include <vector>

#include "aqlm_common.h"

// The file is compiled twice: as it is for 8-element codebook vectors (16 B; 16 slices of 4096 entries x 16 row groups), and
// through gemv_packed_g16.hip for 16-element vectors (AQLM_PK_G 16, AQLM_PK_S_LOG 5, AQLM_PK_NG_LOG 3: 32 B per entry, 32 slices
// of 2048 entries = 64 KiB, 8 row groups; two ds_read_b128 per side and entry).  The second build lives in its own namespace
// and exports its entry points under aqlm_hip_g16_*; the public entries of the first build forward to them (in_group_size 16 /
// a descriptor with slices_log2 == 5).  Matches the reference kernel's template on g in {8, 16} (cuda_kernel.cu:476-521).
#ifndef AQLM_PK_G
#define AQLM_PK_G 8
#endif
#if AQLM_PK_G == 16
#define PK_NS pk_g16
#define aqlm_hip_prepack_1x16_bytes aqlm_hip_g16_prepack_1x16_bytes
#define aqlm_hip_prepack_1x16 aqlm_hip_g16_prepack_1x16
#define aqlm_hip_prepack_1x16_ex aqlm_hip_g16_prepack_1x16_ex
#define aqlm_hip_packed_set_codebook aqlm_hip_g16_packed_set_codebook
#define aqlm_hip_packed_plan_relabel aqlm_hip_g16_packed_plan_relabel
#define aqlm_hip_packed_plan_relabel_ex aqlm_hip_g16_packed_plan_relabel_ex
#define aqlm_hip_packed_plan_geometry aqlm_hip_g16_packed_plan_geometry
#define aqlm_hip_packed_desc_read aqlm_hip_g16_packed_desc_read
#define aqlm_hip_unpack_1x16 aqlm_hip_g16_unpack_1x16
#define aqlm_hip_gemv_1x16_packed_cells aqlm_hip_g16_gemv_1x16_packed_cells
#define aqlm_hip_gemv_1x16_packed aqlm_hip_g16_gemv_1x16_packed
#define aqlm_hip_gemv_1x16_packed_chain aqlm_hip_g16_gemv_1x16_packed_chain
#define aqlm_hip_gemv_1x16_packed_partials aqlm_hip_g16_gemv_1x16_packed_partials
#define aqlm_hip_gemv_1x16_packed_publish aqlm_hip_g16_gemv_1x16_packed_publish
#define aqlm_hip_gemv_1x16_packed_multi aqlm_hip_g16_gemv_1x16_packed_multi
#define aqlm_hip_gemv_1x16_packed_multi_cells aqlm_hip_g16_gemv_1x16_packed_multi_cells
#define PK_API __attribute__((visibility("hidden")))  // internal to libaqlm_hip.so: reached through the public entries only
#else
#define PK_NS pk_g8
#define PK_API
#endif

namespace aqlm {
namespace PK_NS {

constexpr int PK_G = AQLM_PK_G;                              // elements of a codebook vector
constexpr int PK_VSH = PK_G == 8 ? 4 : 5;                    // log2(bytes of a vector)
constexpr uint32_t PK_VB = 1u << PK_VSH;                     // bytes of a codebook vector == bytes of x per input group
constexpr uint32_t PK_HMASK = 0xfff0u;                       // LDS byte offset inside a 16-bit half of an entry
// 32-byte vectors are read as two 16-byte halves.  All vectors start at even 16-B slots, so a plain "first halves, then second
// halves" would use only 8 of the 16 bank groups per read.  Odd lanes therefore read the halves in the opposite order: bit 4 of
// BOTH halves of an entry (spare: the offsets are multiples of 32) is the parity of the lane the entry is stored for, the
// first read uses the offsets as they are, the second flips bit 4.
constexpr uint32_t PK_PARITY_BITS = PK_G == 16 ? 0x00100010u : 0u;
static_assert(PK_G == 8 || PK_G == 16, "codebook vectors of 8 or 16 elements");

#ifndef AQLM_PK_S_LOG
#define AQLM_PK_S_LOG 4  // 16 slices of 64 KiB; 5 = 32 slices of 32 KiB (experiment builds: tools/microbench)
#endif
constexpr int PK_S_LOG = AQLM_PK_S_LOG;
constexpr int PK_S = 1 << PK_S_LOG;          // slices
#ifndef AQLM_PK_NG_LOG
#define AQLM_PK_NG_LOG (8 - AQLM_PK_S_LOG)  // row groups: by default PK_S * PK_NG == 256 workgroups == CUs
#endif
constexpr int PK_NG = 1 << AQLM_PK_NG_LOG;   // row groups
constexpr int PK_NST = PK_NG * PK_S;         // streams == workgroups of a layer
#ifndef AQLM_PK_XFIRST
#define AQLM_PK_XFIRST 1  // batch-1 LDS map: 1 = x first (a 64 KiB window, x copies possible), 0 = slice first (LDS = slice + x: several workgroups per CU)
#endif
constexpr int PK_CODE_BITS = 16 - PK_S_LOG;  // bits of a code inside its slice
constexpr int PK_SLICE_ENTRIES = 1 << PK_CODE_BITS;
constexpr uint32_t PK_SLICE_BYTES = PK_SLICE_ENTRIES * PK_VB;
constexpr int PK_MAX_NW = 16;
constexpr int PK_MAX_T = 1024;
constexpr int PK_MAX_GROUPS = (int)(65536u / PK_VB) - 2;  // the x offset of a group is a 16-bit byte offset; in_groups itself is the null slot
constexpr uint32_t PK_MAGIC = 0x37505141u;   // "AQP7"
constexpr int PK_VERSION = 7;
constexpr int PK_MIN_GROUPS = PK_NG / 2;     // variable geometry: workgroups of a slice (its rows per group stay <= 2 x the uniform count)
constexpr int PK_VG_MIN_ROWS = 512;          // ... and only layers of at least this many rows (every group owns >= 1 row)
constexpr uint32_t PK_XWIN_FULL = 65520;     // x window of the batch-1 kernel (x first, slice behind it)
// accumulator cell of the fused finalize: [arrivals : CNT bits][non-finite contributions : CNT bits][fixed-point sum]
constexpr int PK_CNT_BITS = PK_S_LOG + 1;                       // counts 0 .. PK_S
constexpr unsigned long long PK_CNT_MASK = (1ull << PK_CNT_BITS) - 1ull;
constexpr int PK_VAL_SHIFT = 2 * PK_CNT_BITS;                   // 10 for 16 slices
// |slice sum| < 2^e is stored in units of 2^(e - PK_FIX_BITS).  The finite test lets an addend reach 2 x the bound (rounding slack, a
// slightly stale codebook range), so PK_S addends stay below 2^(PK_FIX_BITS + 1 + PK_S_LOG), which must fit the signed sum field of
// 64 - PK_VAL_SHIFT bits: PK_FIX_BITS <= 60 - 3 PK_S_LOG.  16 slices: 47 (the 51 - PK_S_LOG of rounds 2-3); 32 slices (g16): 45 -- with
// 46 a stale range could wrap the sum instead of giving the NaN the header promises.
constexpr int PK_FIX_BITS = (51 - PK_S_LOG) < (60 - 3 * PK_S_LOG) ? (51 - PK_S_LOG) : (60 - 3 * PK_S_LOG);
static_assert(PK_FIX_BITS + 1 + PK_S_LOG <= 63 - 2 * (PK_S_LOG + 1), "PK_S addends of up to twice the bound must fit the sum field");

// x copies (batch-1 kernel): copy c of x starts at 16-B slot c * stride with stride = 4 (mod 16), i.e. its bank-group
// pattern is rotated by 4 c: an entry can read the copy whose bank group is still free in its service group.
__host__ __device__ static inline int pk_x_stride(int in_groups) { return ((in_groups + 1 + 11) & ~15) + 4; }  // >= in_groups + 1
static inline int pk_max_x_copies(int in_groups) { return std::max(1, std::min(4, 4095 / pk_x_stride(in_groups))); }
// start row of a column: 15 bits over the low nibbles of its first lane-step (entry 0: bits 1-3, entries 1-3: bits 0-3)
__host__ __device__ static inline uint32_t pk_get_start_row(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3) {
  return ((e0 >> 1) & 7u) | ((e1 & 15u) << 3) | ((e2 & 15u) << 7) | ((e3 & 15u) << 11);
}

constexpr int PK_STEP3 = 768;   // bytes of one wave step of 3-byte entries (64 lanes x 12 B)
constexpr int PK_WREG3 = 776;   // ... per step incl. its 8-B row-end flag word (flag words lead the wave range)

// Stream geometry: which slice and which rows workgroup / stream `st` owns.
//   uniform (vg == 0, formats <= v6): PK_NG row groups of RG rows for every slice, stream = group * PK_S + slice;
//   variable (vg == 1): slice s has n[s] row groups, the rows are split evenly over them (the first M % n[s] groups hold one
//   row more), streams are numbered slice by slice: stream = first[s] + group.
struct PkGeom {
  int M;
  int vg;
  int RG;                    // most rows of any stream: the row-start tables have RG + 1 entries per stream, the LDS row tables RG + 1
  uint16_t first[PK_S + 1];  // first stream of slice s (vg)
  uint8_t n[PK_S];           // row groups (= workgroups) of slice s
};

__host__ __device__ static inline void pk_group_rows(const PkGeom& G, int s, int k, int& row0, int& nrows) {
  if (!G.vg) {
    row0 = k * G.RG;
    const int n = G.M - row0;
    nrows = n < 0 ? 0 : (n < G.RG ? n : G.RG);
    return;
  }
  const int n = G.n[s], base = G.M / n, extra = G.M - base * n;
  row0 = k * base + (k < extra ? k : extra);
  nrows = base + (k < extra ? 1 : 0);
}
__host__ __device__ static inline void pk_stream_slice(const PkGeom& G, int st, int& s, int& k) {
  if (!G.vg) {
    s = st & (PK_S - 1);
    k = st >> PK_S_LOG;
    return;
  }
  s = 0;
  while (s + 1 < PK_S && (int)G.first[s + 1] <= st) ++s;
  k = st - (int)G.first[s];
}
__host__ __device__ static inline void pk_stream_rows(const PkGeom& G, int st, int& s, int& row0, int& nrows) {
  int k;
  pk_stream_slice(G, st, s, k);
  pk_group_rows(G, s, k, row0, nrows);
}
// stream and row-in-stream of (slice, row)
__host__ __device__ static inline void pk_row_stream(const PkGeom& G, int s, int row, int& st, int& r) {
  if (!G.vg) {
    const int k = row / G.RG;
    r = row - k * G.RG;
    st = k * PK_S + s;
    return;
  }
  const int n = G.n[s], base = G.M / n, extra = G.M - base * n, thr = extra * (base + 1);
  int k;
  if (row < thr) {
    k = row / (base + 1);
    r = row - k * (base + 1);
  } else {
    k = extra + (row - thr) / base;
    r = row - thr - (k - extra) * base;
  }
  st = (int)G.first[s] + k;
}

// geometry from the row groups per slice (nullptr: uniform); false if they do not describe PK_NST workgroups
static bool pk_make_geom(int M, const uint8_t* groups, PkGeom& G) {
  G.M = M;
  G.vg = 0;
  int sum = 0, mn = PK_NG;
  for (int s = 0; s < PK_S; ++s) {
    const int n = groups ? (int)groups[s] : PK_NG;
    if (n < 1) return false;
    G.n[s] = (uint8_t)n;
    G.first[s] = (uint16_t)sum;
    sum += n;
    mn = n < mn ? n : mn;
    if (n != PK_NG) G.vg = 1;
  }
  G.first[PK_S] = (uint16_t)sum;
  if (sum != PK_NST) return false;
  if (G.vg && (PK_G != 8 || M < PK_VG_MIN_ROWS)) return false;
  G.RG = G.vg ? (M + mn - 1) / mn : (M + PK_NG - 1) / PK_NG;
  return true;
}

struct PackedLayout {
  int M, in_groups, RG, NW, T, XC, EB;
  size_t nst, off_winfo, off_rowstart, off_acc, off_ent, ent_bytes, used;
  PkGeom G;
  bool relabel;            // permutation (u16 old_of_new[65536]) at off_perm, codebook image at off_cb
  size_t off_perm, off_cb;
};

__host__ __device__ static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int packed_max_batch(int in_groups, int RG);  // rows of x whose LDS image fits a CU (0: not even one)

static bool packed_shape_ok(int out_features, int in_features, int g) {
  return g == PK_G && out_features > 0 && in_features > 0 && in_features % PK_G == 0 && in_features / PK_G <= PK_MAX_GROUPS &&
         (out_features + PK_NG - 1) / PK_NG <= 32767 - PK_MAX_NW &&
         packed_max_batch(in_features / PK_G, (out_features + PK_NG - 1) / PK_NG) >= 1;  // slice + x + the row tables of a row group in 160 KiB
}

static bool packed_layout(int out_features, int in_features, int NW, int T, PackedLayout& L, int XC = 1, int EB = 4,
                          const uint8_t* groups = nullptr, bool relabel = false) {
  if (!packed_shape_ok(out_features, in_features, PK_G) || NW < 1 || NW > PK_MAX_NW || T < 1 || T > PK_MAX_T) return false;
  if (XC < 1 || XC > pk_max_x_copies(in_features / PK_G) || (PK_G != 8 && XC != 1)) return false;
  if (EB != 4 && !(EB == 3 && T <= 32 && PK_G == 8)) return false;  // 3-byte entries: the row-end flags of a column are one 32-bit mask
  if (!pk_make_geom(out_features, groups, L.G)) return false;
  if (L.G.vg && (EB != 4 || L.G.RG > 32767 - PK_MAX_NW || packed_max_batch(in_features / PK_G, L.G.RG) < 1)) return false;
  L.XC = XC;
  L.EB = EB;
  L.M = out_features;
  L.in_groups = in_features / PK_G;
  L.RG = L.G.RG;
  L.NW = NW;
  L.T = T;
  L.nst = (size_t)PK_NST;
  L.off_winfo = 256;
  L.off_rowstart = align_up(L.off_winfo + L.nst * PK_MAX_NW * 16, 256);         // [nst][RG + 1] u32 (offset independent of NW)
  // accumulator cells of the fused finalize: [AQLM_HIP_MAX_GEMV_BATCH][M] u64, zero at rest (offset independent of NW, T)
  L.off_acc = align_up(L.off_rowstart + L.nst * (size_t)(L.RG + 1) * 4, 256);
  L.off_ent = align_up(L.off_acc + (size_t)AQLM_HIP_MAX_GEMV_BATCH * out_features * 8, 1024);
  L.ent_bytes = L.nst * NW * T * (EB == 3 ? (size_t)PK_WREG3 : (size_t)1024);
  L.relabel = relabel;
  L.off_perm = align_up(L.off_ent + L.ent_bytes, 1024);
  L.off_cb = L.off_perm + (size_t)65536 * 2;
  L.used = relabel ? L.off_cb + (size_t)65536 * PK_VB : L.off_ent + L.ent_bytes;
  return L.ent_bytes < ((size_t)1 << 32);  // 32-bit buffer offsets
}

static bool desc_layout(const aqlm_hip_packed_desc* d, PackedLayout& L) {
  return d && d->magic == PK_MAGIC && d->version == PK_VERSION && d->slices_log2 == PK_S_LOG &&
         packed_layout(d->out_features, d->in_features, d->waves, d->steps, L, (int)d->x_copies, d->entry_bytes, d->slice_groups,
                       (d->flags & AQLM_HIP_PACKED_RELABELLED) != 0) &&
         L.used == d->used_bytes && d->rows_per_group == L.RG && ((d->flags & AQLM_HIP_PACKED_VARGEOM) != 0) == (L.G.vg != 0);
}

// Wave-steps of work in the longest stream -> waves per workgroup.
// Long streams (>= 13 steps per wave at 16 waves: the 28672-row / 28672-wide layers) are bound by the entry stream and want
// every wave the CU can hold: 16 (8192 -> 28672 with 14 / 12 waves: 30.9 / 30.7 us against 25.1; r03_mb_packed_variants.log).
// Mid-size layers are bound by the LDS and run as fast or faster on 14 waves (single-kernel regime, same box: 8192 -> 8192
// 10.6 vs 10.9 us, 14336 -> 4096 10.3 vs 10.5, 4096 -> 14336 10.45 vs 10.55, 4096 -> 11008 9.10 vs 9.09) -- and 14 waves
// leave room for the two DMA waves of the pipelined shared-input kernel, whose fill then hides completely (2 x 4096 ->
// 14336 in one launch: 18.85 us packed for 14 waves, 19.75 for 16).  (Round 2's "13-15 waves are 10-18 % slower" was
// measured with the two-kernel finalize and no longer holds.)  Small layers (< 48 wave-steps per workgroup): the wave
// ranges have T = ceil(q / NW) steps each, so the capacity NW * T overshoots the content by up to NW - 1 steps, a large
// share of such a layer: pick 4..8 waves with the least overshoot.
static int choose_waves(uint32_t max_lane_steps) {
  const int q = (int)((max_lane_steps + 63) / 64);
  if (tuning().packed_waves >= 1 && tuning().packed_waves <= PK_MAX_NW) return tuning().packed_waves;
  if (q >= 48) return (q + 15) / 16 >= 13 ? 16 : 14;
  // ... and a step count that is a multiple of the ring depth (3): a remainder of one or two steps runs through the
  // kernel's tail code and leaves late requests behind (measured, profiles/r02_mb_wave_counts.log: 4096x4096 with
  // 6 waves x 6 steps 6.09 us, 7 x 5 6.23 us, 5 x 7 6.22 us; 8192->1024 with 6 x 3 = 7 x 3 5.23 us, 5 x 4 5.55 us)
  int best = 8, best_cost = 1 << 30;
  for (int nw = 8; nw >= (PK_NST > 256 ? 2 : 4); --nw) {  // many small workgroups per CU: fewer waves each
    const int t = (q + nw - 1) / nw;
    const int cost = (nw * t - q) * 8 + (8 - nw) + ((t >= 3 && t % 3 != 0) ? 12 : 0);
    if (cost < best_cost) { best_cost = cost; best = nw; }
  }
  return best;
}

// ------------------------------------------------------------------------------------------------ prepack
// K0: how often every codebook entry is used (the relabelling plan is made from it on the host).
__global__ __launch_bounds__(256) void pk_hist_kernel(const uint16_t* codes, size_t n, uint32_t* hist) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) atomicAdd(&hist[codes[i]], 1u);
}

// K1: lane-steps per (slice, row) -> ls[s][row] (u16), and their totals per slice.  One wave per row.  `relabel` (nullable):
// new label of every checkpoint label.
__global__ __launch_bounds__(256) void pk_count_kernel(const uint16_t* codes, const uint16_t* relabel, uint16_t* ls, uint32_t* slice_steps,
                                                       int M, int in_groups) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  uint32_t cnt[PK_S];
#pragma unroll
  for (int s = 0; s < PK_S; ++s) cnt[s] = 0;
  for (int j = lane; j < in_groups; j += 64) {
    uint32_t code = codes[(size_t)row * in_groups + j];
    if (relabel) code = relabel[code];
    const uint32_t sl = code >> PK_CODE_BITS;
#pragma unroll
    for (int s = 0; s < PK_S; ++s) cnt[s] += (sl == (uint32_t)s);
  }
#pragma unroll
  for (int s = 0; s < PK_S; ++s) {
    uint32_t v = cnt[s];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    if (lane == 0) {
      const uint32_t steps = v == 0 ? 1u : (v + 3u) >> 2;
      ls[(size_t)s * M + row] = (uint16_t)steps;
      atomicAdd(&slice_steps[s], steps);
    }
  }
}

// K2: per stream, exclusive prefix sum of its rows' lane-steps -> a[st][0..RG] (entries past the stream's rows = its total);
// maxL = longest stream.
__global__ __launch_bounds__(256) void pk_scan_kernel(const uint16_t* ls, uint32_t* a, uint32_t* maxL, const PkGeom G) {
  __shared__ uint32_t sums[256];
  const int st = blockIdx.x;
  int sl, row0, nrows;
  pk_stream_rows(G, st, sl, row0, nrows);
  const uint16_t* src = ls + (size_t)sl * G.M + row0;
  uint32_t* row = a + (size_t)st * (G.RG + 1);
  const int t = threadIdx.x;
  const int n = G.RG + 1;
  const int chunk = (n + 255) / 256;
  const int lo = std::min(n, t * chunk), hi = std::min(n, lo + chunk);
  uint32_t s = 0;
  for (int i = lo; i < hi; ++i) s += i < nrows ? (uint32_t)src[i] : 0u;
  sums[t] = s;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 256; ++i) {
      const uint32_t v = sums[i];
      sums[i] = run;
      run += v;
    }
    atomicMax(maxL, run);
  }
  __syncthreads();
  uint32_t run = sums[t];
  for (int i = lo; i < hi; ++i) {
    row[i] = run;
    run += i < nrows ? (uint32_t)src[i] : 0u;
  }
}

__global__ __launch_bounds__(256) void pk_fill_kernel(uint32_t* p, size_t n, uint32_t v) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}

__device__ __forceinline__ size_t pk_entry_index(uint32_t q, int k, size_t st, int NW, int T) {
  const uint32_t w = q / (64u * T), rem = q - w * 64u * T;
  const uint32_t l = rem / T, t = rem - l * T;
  return ((((size_t)st * NW + w) * T + t) * 64 + l) * 4 + k;
}

// K3: scatter the entries, ascending j inside every (row, slice).  One wave per row.
__global__ __launch_bounds__(256) void pk_scatter_kernel(const uint16_t* codes, const uint16_t* relabel, const uint32_t* a, uint32_t* ent,
                                                         const PkGeom G, int in_groups, int NW, int T) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= G.M) return;
  uint32_t cnt[PK_S];  // entries of this row already placed, per slice
#pragma unroll
  for (int s = 0; s < PK_S; ++s) cnt[s] = 0;
  for (int j0 = 0; j0 < in_groups; j0 += 64) {
    const int j = j0 + lane;
    const bool ok = j < in_groups;
    uint32_t code = ok ? codes[(size_t)row * in_groups + j] : 0u;
    if (relabel) code = relabel[code];
    const uint32_t sl = ok ? (code >> PK_CODE_BITS) : 0xffffffffu;
#pragma unroll
    for (int s = 0; s < PK_S; ++s) {
      const bool mine = sl == (uint32_t)s;
      const unsigned long long m = __ballot(mine);
      if (mine) {
        int st, r;
        pk_row_stream(G, s, row, st, r);
        const uint32_t i = cnt[s] + __popcll(m & ((1ull << lane) - 1ull));
        const uint32_t q = a[(size_t)st * (G.RG + 1) + r] + (i >> 2);
        ent[pk_entry_index(q, i & 3, (size_t)st, NW, T)] = ((uint32_t)j << (16 + PK_VSH)) | ((code & (PK_SLICE_ENTRIES - 1)) << PK_VSH);
      }
      cnt[s] += __popcll(m);
    }
  }
}

// K3c (32-byte vectors): stamp the lane parity into every entry (see PK_PARITY_BITS); entry (.., t, lane, k) sits at
// word ((..) * 64 + lane) * 4 + k.
__global__ __launch_bounds__(256) void pk_parity_kernel(uint32_t* ent, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    ent[i] = (ent[i] & ~PK_PARITY_BITS) | (((i >> 2) & 1u) ? PK_PARITY_BITS : 0u);
}

// K3b: bank-aware order of the entries.  In the gemv kernel the 64 lanes of a wave execute entry slot (t, k) together:
// one ds_read_b128 for the codebook vectors, one for the x vectors.  The LDS services such a read in four groups of 16
// lanes (MI355X_MICROARCH.md: {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) and needs one extra cycle for every lane whose
// 16-B slot falls into a bank group (slot % 16) another lane of its group already uses: random slots cost ~3 cycles per
// group instead of 1 for both reads, and the loop is LDS-bound.  Two freedoms are used against that:
//   * a row's entries may be summed in any order: inside a wave range the entries of a row form a POOL that is re-dealt
//     over the row's slots (greedy, slot by slot; within a slot the lanes of a service group choose one after the other
//     with rotating priority; a lane takes from its row's pool the first entry whose codebook bank group AND x bank group
//     are still free in its service group, else a null entry -- all nulls read the same two addresses, which the LDS
//     broadcasts --, else one that is free in one of the two);
//   * x is small: the batch-1 kernel keeps XC copies of it in LDS, copy c rotated by 4 c bank groups, and an entry names
//     the copy it reads -- so the x side almost always finds a free bank group and the pool choice serves the codebook.
// The order of choices is fixed -> the layout is deterministic.  One wave per (stream, wave range): the 64 lanes scan
// the pool in parallel, the picks themselves are sequential.  Wave ranges of more than 32 steps keep ascending-j order.
constexpr int PK_ARR_MAX_T = 32;

__device__ __forceinline__ int pk_group_lane(int grp, int pos) {  // inverse of (service group, position) -> lane
  const int half = grp >> 1, g1 = grp & 1;
  int h;
  if (!g1) h = pos < 4 ? pos : (pos < 8 ? pos + 8 : pos + 12);          // G0: 0-3, 12-15, 20-27
  else h = pos < 8 ? pos + 4 : (pos < 12 ? pos + 8 : pos + 16);          // G1: 4-11, 16-19, 28-31
  return half * 32 + h;
}

__global__ __launch_bounds__(64) void pk_arrange_kernel(const uint32_t* a, uint32_t* ent, const PkGeom G, int in_groups, int NW,
                                                        int T, int XC) {
  extern __shared__ uint32_t arr_sm[];
  uint32_t* pool = arr_sm;                                                   // [64 * T * 4] entries in (lane, t, k) order
  uint16_t* rowa = reinterpret_cast<uint16_t*>(pool + (size_t)T * 256);    // [64 * T] first lane-step (in the wave range) of the slot's row
  uint16_t* rem = rowa + (size_t)T * 64;                                    // [64 * T] entries left in the pool of the row starting at that lane-step
  const size_t st = blockIdx.x;
  const int w = blockIdx.y, l = threadIdx.x;
  int sl_, row0_, nrows;
  pk_stream_rows(G, (int)st, sl_, row0_, nrows);
  const uint32_t* starts = a + st * (G.RG + 1);
  const uint32_t total = starts[nrows];
  uint32_t* wave_ent = ent + (((size_t)st * NW + w) * T) * 256;             // + (t * 64 + lane) * 4 + k
  const uint32_t w0 = (uint32_t)w * 64u * (uint32_t)T;
  const uint32_t q0 = w0 + (uint32_t)l * (uint32_t)T;
  for (int t = 0; t < T; ++t) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(wave_ent + ((size_t)t * 64 + l) * 4);
    *reinterpret_cast<u32x4*>(pool + ((size_t)l * T + t) * 4) = v;
  }
  {  // row range of every lane-step of this column, clipped to the wave range
    int r = nrows;
    if (q0 < total) {  // last r with starts[r] <= q0
      int lo = 0, hi = nrows;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (starts[mid] <= q0) lo = mid; else hi = mid;
      }
      r = lo;
    }
    for (int t = 0; t < T; ++t) {
      const uint32_t q = q0 + (uint32_t)t;
      uint32_t ra = q - w0, rb = ra + 1;  // trailing null lane-steps: pools of one step
      if (q < total) {
        while (starts[r + 1] <= q) ++r;
        const uint32_t rs = starts[r], re = starts[r + 1];
        ra = rs > w0 ? rs - w0 : 0u;
        rb = re < w0 + 64u * (uint32_t)T ? re - w0 : 64u * (uint32_t)T;
      }
      rowa[l * T + t] = (uint16_t)ra;
      if (q - w0 == ra) rem[ra] = (uint16_t)((rb - ra) * 4u);
    }
  }
  __syncthreads();
  const uint32_t null_j = (uint32_t)in_groups;
  const uint32_t xstride = (uint32_t)pk_x_stride(in_groups);
  for (int s = 0; s < 4 * T; ++s) {
    const int t = s >> 2, k = s & 3;
    for (int grp = 0; grp < 4; ++grp) {
      uint32_t ux = 0u, uc = 0u;  // bank groups taken in this (slot, service group); wave-uniform
      bool null_placed = false;   // the null entry's two addresses are already being read (further nulls are free)
      for (int r = 0; r < 16; ++r) {
        const int tl = pk_group_lane(grp, (r - s) & 15);  // the lane whose turn it is
        const uint32_t par = PK_G == 16 ? (uint32_t)(tl & 1) : 0u;  // 32-byte vectors: odd lanes read the upper half first
        const uint32_t null_x = 1u << (((null_j << (PK_VSH - 4)) & 15u) ^ par), null_c = 1u << par;
        const uint32_t ra = rowa[tl * T + t];
        const uint32_t n = rem[ra];
        const uint32_t base = ra * 4u;
        uint32_t key = 0u;  // (score + 1) << 20 | copy << 16 | (0xffff - index): the maximum is the best, lowest-index candidate
        for (uint32_t i = (uint32_t)l; i < n; i += 64u) {
          const uint32_t v = pool[base + i];
          const uint32_t j = v >> (16 + PK_VSH);
          uint32_t score, copy = 0u;
          if (j == null_j) score = (null_placed || (!(ux & null_x) && !(uc & null_c))) ? 3u : 0u;
          else {
            const bool cf = !((uc >> (((v >> 4) & 15u) ^ par)) & 1u);
            bool xf = false;
            for (uint32_t c = 0; c < (uint32_t)XC; ++c)
              if (!((ux >> ((((j << (PK_VSH - 4)) + 4u * c) & 15u) ^ par)) & 1u)) { xf = true; copy = c; break; }
            score = xf && cf ? 4u : (xf ? 2u : (cf ? 1u : 0u));
          }
          const uint32_t kk = ((score + 1u) << 20) | (copy << 16) | (0xffffu - i);
          key = kk > key ? kk : key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t other = (uint32_t)__shfl_xor((int)key, o, WAVE);
          key = other > key ? other : key;
        }
        if (key == 0u) __builtin_trap();  // every slot of a row has an entry left in the row's pool
        const uint32_t idx = 0xffffu - (key & 0xffffu), copy = (key >> 16) & 3u;
        const uint32_t v = pool[base + idx];           // same address in every lane: broadcast
        const uint32_t j = v >> (16 + PK_VSH);
        if (j != null_j) {
          ux |= 1u << ((((j << (PK_VSH - 4)) + 4u * copy) & 15u) ^ par);
          uc |= 1u << (((v >> 4) & 15u) ^ par);
        } else {
          ux |= null_x;
          uc |= null_c;
          null_placed = true;
        }
        __syncthreads();  // everybody has read pool[base + idx] and rem[ra]
        if (l == 0) {
          const uint32_t out = j == null_j ? v : ((v & ((1u << (16 + PK_VSH)) - 1u)) | ((j + copy * xstride) << (16 + PK_VSH)) | (copy << 16));
          wave_ent[((size_t)t * 64 + tl) * 4 + k] = out;
          pool[base + idx] = pool[base + n - 1u];      // swap-remove
          rem[ra] = (uint16_t)(n - 1u);
        }
        __syncthreads();
      }
    }
  }
}

// K3c: local search on the greedy deal.  The greedy order is good in a wave range's early slots and poor in its last ones
// (the pools run dry: whatever is left must be taken).  This pass walks the slots again; an entry that shares its codebook
// or x bank group with another lane of its service group looks at every other position of its row's pool and swaps with
// the one that lowers
//     cost(slot, service group) = 8 * (max_b count_c[b] + max_b count_x[b]) + sum_b count_c[b]^2 + sum_b count_x[b]^2
// (the max terms are the LDS cycles of the two reads, the squares break ties towards flatter histograms) the most, summed
// over the two (slot, service group) cells the swap touches; swaps that leave the cost unchanged are taken too (they walk
// plateaus), PK_IMPROVE_SWEEPS passes (3 for long wave ranges).  tools/arrangement_bound.py: LDS cycles per service group and read 2.18 + 1.70 ->
// 1.6 + 1.5 (lower bound of any order 1.27 + 1.26, simulated annealing 1.57 + 1.43).  Same launch shape and the same fixed
// order of decisions as K3b -> deterministic.  Entries are still plain here (parity, flags and row numbers are stamped
// later), null entries of a cell read one address (two with 32-byte vectors: one per lane parity) and count once.
constexpr int PK_IMPROVE_SWEEPS = 6;       // most of the gain comes in the first three (3.88 -> 3.34 / 3.23 / 3.16 ... 3.07 cycles)
constexpr long PK_IMPROVE_MAX_CODES = 8L << 20;  // packed_arrange = 1 (default): layers above this keep the greedy deal alone
constexpr int PK_IMPROVE_SWEEPS_LONG = 3;  // wave ranges of more than 12 steps: the 70B layers, bound by their entry stream anyway

struct PkHist {
  uint32_t w[4];  // 16 counts of 8 bits
  __device__ __forceinline__ void add(uint32_t bank, uint32_t delta) {  // delta = +1 or -1 (as uint32_t)
    const uint32_t v = delta << ((bank & 3u) * 8u);
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) w[i] += (bank >> 2) == i ? v : 0u;
  }
  __device__ __forceinline__ uint32_t count(uint32_t bank) const {
    uint32_t v = 0u;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) v = (bank >> 2) == i ? w[i] : v;
    return (v >> ((bank & 3u) * 8u)) & 255u;
  }
  __device__ __forceinline__ int cost() const {
    int mx = 0, sq = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = (int)((w[i >> 2] >> ((i & 3) * 8)) & 255u);
      mx = c > mx ? c : mx;
      sq += c * c;
    }
    return 8 * mx + sq;
  }
};

__device__ __forceinline__ int pk_lane_group(int lane) {  // LDS service group of a lane (inverse of pk_group_lane)
  const int h = lane & 31;
  const int g1 = (h >= 4 && h < 12) || (h >= 16 && h < 20) || h >= 28;
  return (lane >> 5) * 2 + g1;
}

__global__ __launch_bounds__(64) void pk_improve_kernel(const uint32_t* a, uint32_t* ent, const PkGeom G, int in_groups, int NW,
                                                        int T) {
  extern __shared__ uint32_t arr_sm[];
  uint32_t* e = arr_sm;                                                       // [64 * T * 4] entries in (lane, t, k) order: a row's pool is contiguous
  PkHist* hc = reinterpret_cast<PkHist*>(e + (size_t)T * 256);               // [4 T slots][4 service groups]
  PkHist* hx = hc + (size_t)T * 16;
  uint16_t* rowa = reinterpret_cast<uint16_t*>(hx + (size_t)T * 16);         // [64 * T] the pool of the lane-step's row: lane-steps [rowa, rowb)
  uint16_t* rowb = rowa + (size_t)T * 64;
  uint8_t* nulls = reinterpret_cast<uint8_t*>(rowb + (size_t)T * 64);        // [4 T][4][2] null entries per cell and lane parity
  const size_t st = blockIdx.x;
  const int w = blockIdx.y, l = threadIdx.x;
  int sl_, row0_, nrows;
  pk_stream_rows(G, (int)st, sl_, row0_, nrows);
  const uint32_t* starts = a + st * (G.RG + 1);
  const uint32_t total = starts[nrows];
  uint32_t* wave_ent = ent + (((size_t)st * NW + w) * T) * 256;
  const uint32_t w0 = (uint32_t)w * 64u * (uint32_t)T;
  if (w0 >= total) return;  // nothing but null entries
  const uint32_t q0 = w0 + (uint32_t)l * (uint32_t)T;
  for (int t = 0; t < T; ++t)
    *reinterpret_cast<u32x4*>(e + ((size_t)l * T + t) * 4) = *reinterpret_cast<const u32x4*>(wave_ent + ((size_t)t * 64 + l) * 4);
  for (int i = l; i < T * 16 * 2 * 4; i += 64) reinterpret_cast<uint32_t*>(hc)[i] = 0u;  // hc and hx
  for (int i = l; i < T * 32; i += 64) nulls[i] = 0;
  {
    int r = nrows;
    if (q0 < total) {
      int lo = 0, hi = nrows;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (starts[mid] <= q0) lo = mid; else hi = mid;
      }
      r = lo;
    }
    for (int t = 0; t < T; ++t) {
      const uint32_t q = q0 + (uint32_t)t;
      uint32_t ra = q - w0, rb = ra + 1;
      if (q < total) {
        while (starts[r + 1] <= q) ++r;
        const uint32_t rs = starts[r], re = starts[r + 1];
        ra = rs > w0 ? rs - w0 : 0u;
        rb = re < w0 + 64u * (uint32_t)T ? re - w0 : 64u * (uint32_t)T;
      }
      rowa[l * T + t] = (uint16_t)ra;
      rowb[l * T + t] = (uint16_t)rb;
    }
  }
  __syncthreads();
  const uint32_t null_j = (uint32_t)in_groups;
  // bank groups of an entry read by a lane of parity `par`; null entries: the zero vector's slot and x slot `in_groups`
  auto bank_c = [&](uint32_t v, uint32_t par) { return (v >> (16 + PK_VSH)) == null_j ? par : (((v >> 4) & 15u) ^ par); };
  auto bank_x = [&](uint32_t v, uint32_t par) { return (((v >> (16 + PK_VSH)) << (PK_VSH - 4)) & 15u) ^ par; };
  auto lane_par = [](int lane) { return PK_G == 16 ? (uint32_t)(lane & 1) : 0u; };
  // the cell's histograms after `v` joined (sign = +1) or left (-1) at lane parity `par`; `nz` = the cell's null count there
  auto apply = [&](PkHist& c, PkHist& x, uint32_t v, uint32_t par, uint32_t sign, uint32_t& nz) {
    if ((v >> (16 + PK_VSH)) == null_j) {
      const bool counts = sign == 1u ? nz == 0u : nz == 1u;  // the first null in, the last null out
      nz += sign;
      if (!counts) return;
    }
    c.add(bank_c(v, par), sign);
    x.add(bank_x(v, par), sign);
  };
  if (l == 0) {  // histograms of the greedy deal (sequential: pack time, 256 T entries)
    for (int lane = 0; lane < 64; ++lane)
      for (int t = 0; t < T; ++t)
        for (int k = 0; k < 4; ++k) {
          const int cell = (t * 4 + k) * 4 + pk_lane_group(lane);
          const uint32_t par = lane_par(lane);
          uint32_t nz = nulls[cell * 2 + par];
          apply(hc[cell], hx[cell], e[(lane * T + t) * 4 + k], par, 1u, nz);
          nulls[cell * 2 + par] = (uint8_t)nz;
        }
  }
  __syncthreads();
  const int sweeps = T <= 12 ? PK_IMPROVE_SWEEPS : PK_IMPROVE_SWEEPS_LONG;
  for (int sweep = 0; sweep < sweeps; ++sweep) {
    for (int s = 0; s < 4 * T; ++s) {
      const int t = s >> 2, k = s & 3;
      for (int lane = 0; lane < 64; ++lane) {  // wave-uniform: the entry under repair
        const int p = (lane * T + t) * 4 + k;
        const uint32_t v = e[p];
        if ((v >> (16 + PK_VSH)) == null_j) continue;
        const uint32_t par = lane_par(lane);
        const int cell = s * 4 + pk_lane_group(lane);
        const PkHist c0 = hc[cell], x0 = hx[cell];
        if (c0.count(bank_c(v, par)) <= 1u && x0.count(bank_x(v, par)) <= 1u) continue;  // alone in both bank groups
        const int cost0 = c0.cost() + x0.cost();
        const uint32_t nz0 = nulls[cell * 2 + par];
        const int base = (int)rowa[lane * T + t] * 4, n = ((int)rowb[lane * T + t] - (int)rowa[lane * T + t]) * 4;
        uint32_t key = ~0u;  // (cost change + 2^14) << 16 | pool index: the minimum is the best, lowest-index partner
        for (int i = l; i < n; i += 64) {
          const int p2 = base + i, q2 = p2 >> 2, k2 = p2 & 3, lane2 = q2 / T, t2 = q2 - lane2 * T;
          const int cell2 = (t2 * 4 + k2) * 4 + pk_lane_group(lane2);
          if (cell2 == cell) continue;
          const uint32_t v2 = e[p2], par2 = lane_par(lane2);
          PkHist c1 = c0, x1 = x0, c2 = hc[cell2], x2 = hx[cell2];
          const int cost2 = c2.cost() + x2.cost();
          uint32_t nz1 = nz0, nz2 = nulls[cell2 * 2 + par2];
          apply(c1, x1, v, par, ~0u, nz1);
          apply(c1, x1, v2, par, 1u, nz1);
          apply(c2, x2, v2, par2, ~0u, nz2);
          apply(c2, x2, v, par2, 1u, nz2);
          int cost_a = c1.cost() + x1.cost(), cost_b = c2.cost() + x2.cost();
          // hipcc 7.2 / gfx950 -O3 miscompiles the four-cost difference when it may fold it (every change came out positive: no
          // swap was ever taken; found with tools/microbench/arr_dbg.hip): keep the new costs as opaque values
          asm volatile("" : "+v"(cost_a), "+v"(cost_b));
          const int d = cost_a + cost_b - cost0 - cost2;
          const uint32_t kk = ((uint32_t)(d + (1 << 14)) << 16) | (uint32_t)i;
          key = kk < key ? kk : key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t other = (uint32_t)__shfl_xor((int)key, o, WAVE);
          key = other < key ? other : key;
        }
        if ((key >> 16) > (1u << 14)) continue;  // every partner makes it worse (or there is none)
        __syncthreads();  // everybody has read the cells
        if (l == 0) {
          const int p2 = base + (int)(key & 0xffffu), q2 = p2 >> 2, k2 = p2 & 3, lane2 = q2 / T, t2 = q2 - lane2 * T;
          const int cell2 = (t2 * 4 + k2) * 4 + pk_lane_group(lane2);
          const uint32_t v2 = e[p2], par2 = lane_par(lane2);
          PkHist c1 = hc[cell], x1 = hx[cell], c2 = hc[cell2], x2 = hx[cell2];
          uint32_t nz1 = nulls[cell * 2 + par], nz2 = nulls[cell2 * 2 + par2];
          apply(c1, x1, v, par, ~0u, nz1);
          apply(c1, x1, v2, par, 1u, nz1);
          apply(c2, x2, v2, par2, ~0u, nz2);
          apply(c2, x2, v, par2, 1u, nz2);
          hc[cell] = c1; hx[cell] = x1; hc[cell2] = c2; hx[cell2] = x2;
          nulls[cell * 2 + par] = (uint8_t)nz1;
          nulls[cell2 * 2 + par2] = (uint8_t)nz2;
          e[p] = v2;
          e[p2] = v;
        }
        __syncthreads();
      }
    }
  }
  for (int t = 0; t < T; ++t)
    *reinterpret_cast<u32x4*>(wave_ent + ((size_t)t * 64 + l) * 4) = *reinterpret_cast<const u32x4*>(e + ((size_t)l * T + t) * 4);
}

// K4: bookkeeping bits.  Thread (st, r): flag on the row's last lane-step.
__global__ __launch_bounds__(256) void pk_flag_kernel(const uint32_t* a, uint32_t* ent, const PkGeom G, int NW, int T) {
  const size_t st = blockIdx.y;
  const int r = blockIdx.x * 256 + threadIdx.x;
  int sl_, row0_, nrows;
  pk_stream_rows(G, (int)st, sl_, row0_, nrows);
  if (r >= nrows) return;
  const uint32_t ql = a[st * (G.RG + 1) + r + 1] - 1;
  atomicOr(&ent[pk_entry_index(ql, 0, st, NW, T)], 1u);
}

// K5: per lane column its starting row (in the spare bits of the column's first lane-step), per wave winfo.
__global__ __launch_bounds__(64) void pk_column_kernel(const uint32_t* a, uint32_t* ent, uint32_t* winfo, const PkGeom G,
                                                       int NW, int T) {
  const size_t st = blockIdx.x;
  const int w = blockIdx.y, l = threadIdx.x;
  int sl_, row0_, nrows;
  pk_stream_rows(G, (int)st, sl_, row0_, nrows);
  const uint32_t* starts = a + st * (G.RG + 1);  // starts[r], r < nrows; starts[nrows] == total (rows past the stream's have 0 steps)
  const uint32_t total = starts[nrows];
  const uint32_t w0 = (uint32_t)w * 64u * T;
  if (l == 0) {
    // wfr = first r in [0, nrows] with starts[r] >= w0  (informational: first row that starts in this wave range)
    int lo = 0, hi = nrows;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (starts[mid] >= w0) hi = mid; else lo = mid + 1;
    }
    uint32_t* wi = winfo + (st * NW + w) * 4;
    wi[0] = (uint32_t)lo;
    wi[1] = (w0 < total && starts[lo] > w0) ? 1u : 0u;
    wi[2] = w0 >= total ? 0u : (total - w0 >= (uint32_t)T ? (uint32_t)T : total - w0);
    wi[3] = 0u;
  }
  const uint32_t q = w0 + (uint32_t)l * T;
  int r0;
  if (q >= total) {
    r0 = nrows;
  } else {  // last r with starts[r] <= q
    int a0 = 0, b0 = nrows;  // invariant: starts[a0] <= q, answer in [a0, b0)
    while (b0 - a0 > 1) {
      const int mid = (a0 + b0) >> 1;
      if (starts[mid] <= q) a0 = mid; else b0 = mid;
    }
    r0 = a0;
  }
  const uint32_t f = (uint32_t)r0;
  uint32_t* e = ent + ((((size_t)st * NW + w) * T + 0) * 64 + l) * 4;
  e[0] |= (f & 7u) << 1;
  e[1] |= (f >> 3) & 15u;
  e[2] |= (f >> 7) & 15u;
  e[3] |= (f >> 11) & 15u;
}

// K6 (3-byte entries): the repack works on the 4-byte layout; this pass squeezes it.  A wave range becomes
// [T flag words of 8 B: bit l = lane l's lane-step t ends a row][T steps of 64 x 12 B]; an entry is slot << 12 | code
// (24 bits), four of them in three dwords.  The start row of a column is no longer stored: the kernel derives it from
// winfo[3] (start row of the wave range's first column) and the flag words.
__device__ __forceinline__ void pk_pack3(const u32x4& v, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
  auto e24 = [](uint32_t x) { return ((x >> 20) << 12) | ((x >> 4) & 0xfffu); };
  const uint32_t e0 = e24(v.x), e1 = e24(v.y), e2 = e24(v.z), e3 = e24(v.w);
  w0 = e0 | (e1 << 24);
  w1 = (e1 >> 8) | (e2 << 16);
  w2 = (e2 >> 16) | (e3 << 8);
}
__device__ __forceinline__ void pk_unpack3(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t (&e)[4]) {
  e[0] = w0 & 0xffffffu;
  e[1] = (w0 >> 24) | ((w1 & 0xffffu) << 8);
  e[2] = (w1 >> 16) | ((w2 & 0xffu) << 16);
  e[3] = w2 >> 8;
}

__global__ __launch_bounds__(64) void pk_compress_kernel(const uint32_t* ent4, uint8_t* ent3, uint32_t* winfo, const PkGeom G,
                                                         int NW, int T) {
  const size_t st = blockIdx.x;
  const int w = blockIdx.y, l = threadIdx.x;
  int sl_, row0_, nrows;
  pk_stream_rows(G, (int)st, sl_, row0_, nrows);
  const uint32_t* src = ent4 + (((size_t)st * NW + w) * T) * 256;
  uint8_t* dst = ent3 + ((size_t)st * NW + w) * T * PK_WREG3;
  uint32_t* wi = winfo + (st * NW + w) * 4;
  for (int t = 0; t < T; ++t) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + ((size_t)t * 64 + l) * 4);
    if (t == 0 && l == 0) wi[3] = wi[2] ? pk_get_start_row(v.x, v.y, v.z, v.w) : (uint32_t)nrows;
    uint32_t w0, w1, w2;
    pk_pack3(v, w0, w1, w2);
    uint32_t* o = reinterpret_cast<uint32_t*>(dst + (size_t)T * 8 + (size_t)t * PK_STEP3 + (size_t)l * 12);
    o[0] = w0; o[1] = w1; o[2] = w2;
    const unsigned long long fl = __ballot((v.x & 1u) != 0u);
    if (l == 0) *reinterpret_cast<unsigned long long*>(dst + (size_t)t * 8) = fl;
  }
}

__global__ __launch_bounds__(64) void pk_unpack3_kernel(const uint8_t* ent3, const uint32_t* winfo, const uint16_t* old_of_new,
                                                        uint16_t* codes, const PkGeom G, int in_groups, int NW, int T) {
  const int xstride = pk_x_stride(in_groups);
  const size_t st = blockIdx.x;
  const int w = blockIdx.y, l = threadIdx.x;
  int s, row0, nrows;
  pk_stream_rows(G, (int)st, s, row0, nrows);
  const uint32_t* wi = winfo + (st * NW + w) * 4;
  const int steps = (int)wi[2];
  const uint8_t* base = ent3 + ((size_t)st * NW + w) * T * PK_WREG3;
  // start row of this column: the wave range's start row + the row ends in the columns before it
  int row = (int)wi[3];
  for (int t = 0; t < steps; ++t) {
    const unsigned long long fl = *reinterpret_cast<const unsigned long long*>(base + (size_t)t * 8);
    row += __popcll(fl & ((1ull << l) - 1ull));
  }
  for (int t = 0; t < steps; ++t) {
    const uint32_t* p3 = reinterpret_cast<const uint32_t*>(base + (size_t)T * 8 + (size_t)t * PK_STEP3 + (size_t)l * 12);
    uint32_t e[4];
    pk_unpack3(p3[0], p3[1], p3[2], e);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int slot = (int)(e[k] >> 12);
      const int j = slot % xstride;
      if (j < in_groups && row < nrows) {
        const uint32_t c = ((uint32_t)s << PK_CODE_BITS) | (e[k] & 0xfffu);
        codes[(size_t)(row0 + row) * in_groups + j] = old_of_new ? old_of_new[c] : (uint16_t)c;
      }
    }
    const unsigned long long fl = *reinterpret_cast<const unsigned long long*>(base + (size_t)t * 8);
    row += (int)((fl >> l) & 1ull);
  }
}

// inverse of the repack: canonical codes [M][in_groups] from a packed buffer (lossless; used to drop / restore the
// canonical codes of inference-only models and by the tests).  One wave per (stream, wave range).
__global__ __launch_bounds__(64) void pk_unpack_kernel(const uint32_t* __restrict__ ent, const uint32_t* __restrict__ winfo,
                                                       const uint16_t* __restrict__ old_of_new, uint16_t* __restrict__ codes, const PkGeom G,
                                                       int in_groups, int NW, int T) {
  const int xstride = pk_x_stride(in_groups);
  const size_t st = blockIdx.x;
  const int w = blockIdx.y, l = threadIdx.x;
  int s, row0, nrows;
  pk_stream_rows(G, (int)st, s, row0, nrows);
  const uint32_t* wi = winfo + (st * NW + w) * 4;
  const int steps = (int)wi[2];
  const u32x4* col = reinterpret_cast<const u32x4*>(ent + (((size_t)st * NW + w) * T * 64 + l) * 4);
  int local = 0;
  // A column's lane-steps are independent loads; only the row counter is sequential.  UB of them are requested before the first is
  // decoded (round 6: the first cut loaded, decoded and stored one lane-step at a time -- with ~10 steps per column and 14 waves per
  // CU the kernel was a chain of HBM round trips: 32 us for the 10 MB of a 4096 x 4096 layer, bench.py detail.unpack_1x16_us).
  constexpr int UB = 8;
  for (int t0 = 0; t0 < steps; t0 += UB) {
    u32x4 ev[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int t = t0 + u < steps ? t0 + u : steps - 1;
      ev[u] = col[(size_t)t * 64];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (t0 + u >= steps) break;
      const uint32_t e[4] = {ev[u].x, ev[u].y, ev[u].z, ev[u].w};
      if (t0 + u == 0) local = (int)pk_get_start_row(e[0], e[1], e[2], e[3]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t v = e[k];
        const int j = (int)(v >> (16 + PK_VSH)) - (int)((v >> 16) & 3u) * xstride;
        if (j < in_groups && local < nrows) {
          const uint32_t c = ((uint32_t)s << PK_CODE_BITS) | ((v >> PK_VSH) & (uint32_t)(PK_SLICE_ENTRIES - 1));
          codes[(size_t)(row0 + local) * in_groups + j] = old_of_new ? old_of_new[c] : (uint16_t)c;
        }
      }
      local += (int)(e[0] & 1u);
    }
  }
}

// ------------------------------------------------------------------------------------------------ gemv
struct PackedGemvParams {
  const uint32_t* ent;
  const uint32_t* winfo;
  const uint32_t* rowstart;  // [nst][RG + 1]
  const uint8_t* codebook;
  const uint16_t* x;
  float* partial;  // [S][B][M]
  long x_row_stride;
  int M, in_groups, RG, NW, T, XC;
  uint32_t ent_bytes;
  // fused finalize (acc != nullptr): the 16 slice workgroups of a row meet in ONE 64-bit cell per (input row, output row)
  unsigned long long* acc;  // [B][M], zero at rest
  float cb_absmax;          // largest |codebook entry| of the layer (bounds the slice sums)
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
  long y_row_stride;
  // chain prefetch (optional): the layer that runs NEXT on this stream.  NPW extra waves of every workgroup pull the
  // next layer's stream of the same workgroup index (same XCD under the observed block % 8 placement) and a share of
  // its codebook slice towards this XCD's L2 while the other waves compute -- the next launch then starts L2-warm.
  // row-parallel shards (one-shot all-reduce over xGMI, xgmi_reduce.hip): instead of writing y, the workgroup that owns
  // a row's total PUBLISHES it -- fp32, system-scope store -- in this rank's pub buffer, and the last workgroup of the
  // launch raises the rank's flag.  nullptr: ordinary launch.
  float* pub;                    // this rank's pub[2][max_elems]
  uint32_t* pub_flag;            // this rank's flag[2]
  uint32_t* pub_epoch;           // [0] epoch, [4..11] arrival counters of the 8 workgroup shards, [12] top counter
  uint32_t pub_max_elems;
  const uint8_t* next_ent;       // entry area of the next layer's packed buffer (nullptr: no prefetch)
  const uint8_t* next_codebook;
  uint32_t next_block_bytes;     // bytes of one workgroup's stream in the next layer (NW' * T' KiB)
  int NPW;                       // prefetch waves in this launch (workgroup = NW + NPW waves)
  int fill_rotate;               // 1: workgroup g starts its slice fill at piece g * (pieces / row groups)
#ifdef AQLM_PACKED_TRACE
  unsigned long long* trace;  // [256 workgroups][8] wall-clock stamps (100 MHz), profiling builds only
  int dbg;                    // bit 0: skip the LDS reads + dot products, bit 1: no entry stream (out-of-range loads)
#endif
};

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(3))) const u32x4* lds_u32x4_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(1))) const void* gbl_void_ptr;
typedef const uint32_t __attribute__((address_space(4)))* const_u32_ptr;  // constant address space: uniform loads go through s_load

template <int WORD>
__device__ __forceinline__ uint32_t half_and(uint32_t w, uint32_t mask) {
  uint32_t d;  // d = ((w >> 16*WORD) & 0xffff) & mask in one instruction (sub-dword operand select)
  if constexpr (WORD == 0)
    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(d) : "v"(mask), "v"(w));
  else
    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d) : "v"(mask), "v"(w));
  return d;
}

__device__ __forceinline__ void lds_store_f32(uint32_t byte_addr, float v) {
  asm volatile("ds_write_b32 %0, %1" : : "v"(byte_addr), "v"(v) : "memory");
}
// LDS reads hipcc does not see (pipelined kernel's epilogue): while LDS-DMA requests are in flight the compiler guards every
// LDS read it knows of with vmcnt(0) -- it cannot tell which addresses the DMA writes.  The caller waits (lds_asm_wait)
// before it uses the values.
__device__ __forceinline__ uint32_t lds_asm_load_b32(uint32_t byte_addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(byte_addr));  // no "memory" clobber: with one hipcc treats the asm as a possible LDS reader and puts vmcnt(0) in front
  return v;
}
__device__ __forceinline__ u32x4 lds_asm_load_b128(uint32_t byte_addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
  return v;
}
// the wait names the registers it guards ("+v"): a bare `s_waitcnt` asm orders nothing for the compiler's scheduler, which
// is free to move a USE of an asm-loaded value in front of it (it happened: wrong row sums in one build, right ones in the next)
__device__ __forceinline__ void lds_asm_wait(uint32_t& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }
__device__ __forceinline__ void lds_asm_wait(uint32_t& a, uint32_t& b, uint32_t& c, u32x4& d0, u32x4& d1, u32x4& d2, u32x4& d3) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
}

// LDS map (byte offsets from the start of the workgroup's LDS, which is address 0: the kernel has no static LDS).
//   B == 1: x at 0 (XWIN bytes reserved), slice at XWIN, bookkeeping behind the slice   (x-first: both reads cost one op)
//   B  > 1: slice at 0, then B planes x[b][j] of XP = (in_groups + 1) * 16 bytes (a plane keeps the bank pattern of
//           the single-row case: slot j -> bank group j % 16; interleaving the rows would leave 16 / B groups), bookkeeping
//   bookkeeping: rowstart[RG + 1] u32 (LDS-DMA copy of the stream's row starts), rowval[B][RG + 1] f32 (sum of the
//   lane-steps of a row from the start of the column that holds its LAST lane-step), colend[B][16 * 64] f32 (what a
//   column accumulated after its last row end: the head of a row that continues in the next column)
template <int B, uint32_t XWIN>
struct PackedLds {
  static constexpr bool XFIRST = (B == 1) && AQLM_PK_XFIRST && XWIN != 0u;  // XWIN == 0: slice first also for one row (tall layers)
  static constexpr uint32_t SLICE = XFIRST ? XWIN : 0u;
  static constexpr uint32_t X = XFIRST ? 0u : PK_SLICE_BYTES;
  __host__ __device__ static uint32_t plane(int in_groups) { return (uint32_t)(in_groups + 1) * PK_VB; }
  __host__ __device__ static uint32_t rowstart(int in_groups) {
    return XFIRST ? XWIN + PK_SLICE_BYTES : PK_SLICE_BYTES + plane(in_groups) * B;
  }
  __host__ __device__ static uint32_t rs_bytes(int RG) { return ((uint32_t)(RG + 1) * 4u + 1023u) & ~1023u; }  // whole DMA pieces
  __host__ __device__ static uint32_t rowval(int in_groups, int RG) { return rowstart(in_groups) + rs_bytes(RG); }
  __host__ __device__ static uint32_t colend(int in_groups, int RG) { return rowval(in_groups, RG) + (uint32_t)B * (RG + 1) * 4u; }
  __host__ __device__ static uint32_t xmax(int in_groups, int RG) {  // 16-B aligned: read with ds_read_b128
    return (colend(in_groups, RG) + (uint32_t)B * PK_MAX_NW * 64 * 4 + 15u) & ~15u;
  }
  __host__ __device__ static uint32_t dump(int in_groups, int RG) {  // 1 KiB landing zone per prefetch wave (LDS-DMA needs a destination)
    return (xmax(in_groups, RG) + (uint32_t)B * PK_MAX_NW * 4u + 15u) & ~15u;
  }
  __host__ __device__ static size_t total(int in_groups, int RG, int npw = 0) {
    return npw ? (size_t)dump(in_groups, RG) + (size_t)npw * 1024
               : (size_t)xmax(in_groups, RG) + (size_t)B * PK_MAX_NW * 4;  // xmax[B][16 waves] u32: largest |x| seen by each wave (fused finalize)
  }
};

// `block` in [0, 256): the workgroup's index within its own layer (== blockIdx.x for a single-layer launch).
// PUB: the row-parallel shard's variant (the last arrival publishes the fp32 total for the peers instead of writing y).  A
// template parameter, not a run-time test of p.pub: carrying the publish branches in the ordinary kernel cost 0.15 us per
// launch (same box, profiles/r03_mb_ab_commits.log).
// VG: variable geometry (format v7): `ns` = the first stream of each of the 16 slices, one byte each; the workgroup finds its
// slice, row range and stream from them with a handful of instructions (no table in memory: a dependent load in front of the
// slice fill would cost more than the imbalance it repairs).
struct PackedVgArgs {
  uint32_t ns[4];
};

template <class T_, int B, int PD, uint32_t XWIN, int EB, bool PUB = false, bool VG = false>
__device__ __forceinline__ void gemv_1x16_packed_body(const PackedGemvParams& p, const int block, const int NWB,
                                                      const PackedVgArgs& vg = PackedVgArgs{}) {
  using LDS = PackedLds<B, XWIN>;
  using ring_t = typename std::conditional<EB == 3, u32x3, u32x4>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int NT = NWB << 6;  // NWB = waves in the workgroup (>= p.NW); passed in: blockDim lives in the hidden kernel arguments
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef AQLM_PACKED_TRACE
  unsigned long long tr[8];
  uint32_t tr_wait = 0, tr_work = 0;
  tr[0] = wall_clock64();
  const unsigned long long cyc0 = __builtin_readcyclecounter();  // s_memtime: shader clock
#define AQLM_TRACE(i) tr[i] = wall_clock64()
#else
#define AQLM_TRACE(i)
#endif
  // slice = block % 16: blocks are observed to land on XCD block % 8, so each XCD's L2 serves two 64 KiB slices
  // (speed only; any placement is correct).
  int slice, group, row_begin, nrows, stream_ix;
  if constexpr (VG) {
    static_assert(PK_S == 16 && PK_NST == 256, "variable geometry: 16 slices, 256 workgroups");
    // workgroups land on XCD block % 8: XCD x serves streams [32 x, 32 x + 32) -- consecutive streams share their slice
    stream_ix = ((block & 7) << 5) | (block >> 3);
    // ns = the first stream of slices 0..15, one byte each (slice 0 starts at 0, the last slice ends at 256).  Lane i < 16
    // looks at slice i: the slices that start at or before this stream answer the ballot, the last of them owns it.
    // (A scalar walk over the 16 bytes was ~280 dependent instructions in front of the first load: 0.5 us.)
    const uint32_t w = lane < 4 ? vg.ns[0] : (lane < 8 ? vg.ns[1] : (lane < 12 ? vg.ns[2] : vg.ns[3]));
    const uint32_t start_v = (w >> ((lane & 3) * 8)) & 255u;
    const unsigned long long m = __ballot((uint32_t)stream_ix >= start_v) & 0xffffull;
    const int s = __popcll(m) - 1;
    const int start = __builtin_amdgcn_readlane((int)start_v, s);
    const int end = s == PK_S - 1 ? PK_NST : __builtin_amdgcn_readlane((int)start_v, s < PK_S - 1 ? s + 1 : s);
    int n = end - start;
    n = n < 1 ? 1 : n;
    slice = s;
    group = stream_ix - start;
    const int base = p.M / n, extra = p.M - base * n;
    row_begin = group * base + (group < extra ? group : extra);
    nrows = base + (group < extra ? 1 : 0);
  } else {
    stream_ix = block;
    slice = block & (PK_S - 1);
    group = block >> PK_S_LOG;
    row_begin = group * p.RG;
    nrows = p.M - row_begin;
    nrows = nrows < 0 ? 0 : (nrows < p.RG ? nrows : p.RG);
  }
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw != 0u) __builtin_trap();  // LDS map above

  const int RG1 = p.RG + 1;
  const uint32_t rowstart_off = LDS::rowstart(p.in_groups);
  const uint32_t rowval_off = LDS::rowval(p.in_groups, p.RG);
  const uint32_t colend_off = LDS::colend(p.in_groups, p.RG);

  // ---- prologue: everything that needs no other data is issued first, in one burst -------------------------------
  const uint32_t XP = LDS::plane(p.in_groups);
  const uint32_t xstride16 = (uint32_t)pk_x_stride(p.in_groups) * PK_VB;
  __amdgpu_buffer_rsrc_t rs_ent = __builtin_amdgcn_make_buffer_rsrc((void*)p.ent, 0, p.ent_bytes, 0x00020000);
  const int wv = wave < p.NW ? wave : p.NW - 1;  // waves beyond the stream's wave count (shared-input launches) idle
  const uint32_t wbase = (uint32_t)(((size_t)stream_ix * p.NW + wv) * p.T) * (EB == 3 ? (uint32_t)PK_WREG3 : 1024u);
  const int Tm1 = p.T - 1;
  // (0) 3-byte entries: the row-end flag words of this wave range, word t in lane t -- the OLDEST load of the queue, so
  // it has landed whenever the wait of (5) returns
  u32x2 flagw = {0u, 0u};
  if constexpr (EB == 3)
    flagw = __builtin_amdgcn_raw_buffer_load_b64(rs_ent, (uint32_t)(lane < Tm1 ? lane : Tm1) * 8u, wbase, 0);
  // Chain prefetch: the last p.NPW waves of the workgroup take no part in the fill or the loop.  They ask for the NEXT
  // layer's bytes (LDS-DMA into a 1 KiB dump zone each: no registers, nothing to wait for before the fill barrier) and
  // meet the others at the barriers.
  const int NWD = NWB - p.NPW;  // waves that fill and compute
  const bool pfw = wave >= NWD;
  if (pfw) {
    const int pw = wave - NWD;
    const uint32_t dump = LDS::dump(p.in_groups, p.RG) + (uint32_t)pw * 1024u;
    const uint8_t* nsrc = p.next_ent + (size_t)block * p.next_block_bytes;
    for (uint32_t off = (uint32_t)pw * 1024u; off < p.next_block_bytes; off += (uint32_t)p.NPW * 1024u)
      __builtin_amdgcn_global_load_lds((gbl_void_ptr)(nsrc + off + lane * 16), (lds_void_ptr)(size_t)dump, 16, 0, AUX_NT);
    constexpr uint32_t SHARE = PK_SLICE_BYTES / PK_NG;  // the PK_NG workgroups of a slice split its next-layer image
    const uint8_t* csrc = p.next_codebook + (size_t)slice * PK_SLICE_BYTES + (size_t)group * SHARE;
    for (uint32_t off = (uint32_t)pw * 1024u; off < SHARE; off += (uint32_t)p.NPW * 1024u)
      __builtin_amdgcn_global_load_lds((gbl_void_ptr)(csrc + off + lane * 16), (lds_void_ptr)(size_t)dump, 16, 0, 0);
  }
  // (1) LDS-DMA: the 64 KiB slice (shared by the 16 workgroups of the XCD that hold it -> L2 hits) and x
  if (!pfw) {
    const uint8_t* src = p.codebook + (size_t)slice * PK_SLICE_BYTES;
    // rotated start: the PK_NG workgroups that fill the same slice from the same L2 walk it from different pieces, so at
    // any moment they ask different L2 channels (and, cold, each pulls a different part from HBM first)
    constexpr int PIECES = (int)(PK_SLICE_BYTES / 1024);
    const int rot = p.fill_rotate ? (group & (PK_NG - 1)) * (PIECES / PK_NG) : 0;
    for (int i0 = wave; i0 < PIECES; i0 += NWD) {
      const int i = (i0 + rot) & (PIECES - 1);
      __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + i * 1024 + lane * 16),
                                       (lds_void_ptr)(size_t)(LDS::SLICE + (uint32_t)i * 1024u), 16, 0, 0);
    }
    const int x16 = p.in_groups * (int)(PK_VB / 16);  // 16-byte units of a row of x
    const int nchunk = (x16 + 63) >> 6;              // KiB pieces per row of x
    // B == 1: XC rotated copies of the row (copy c at slot c * xstride); B > 1: one plane per row
    const int ncopy = B == 1 ? p.XC : B;
    for (int c = wave; c < nchunk * ncopy; c += NWD) {
      const int b = c / nchunk, i = c - b * nchunk;
      const int idx = i * 64 + lane;
      const uint32_t dst = LDS::X + (uint32_t)b * (B == 1 ? xstride16 : XP) + (uint32_t)i * 1024u;
      if (idx < x16)
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(p.x + (B == 1 ? (size_t)0 : (size_t)b * p.x_row_stride) + (size_t)idx * 8),
                                         (lds_void_ptr)(size_t)dst, 16, 0, 0);
    }
    // the stream's row starts (needed by the epilogue only; as an LDS-DMA they are older than the ring loads, see (5)).
    // Rows of the table are only 4-B aligned -> dword DMA, 256 B per wave-instruction.
    const uint32_t* rs_src = p.rowstart + (size_t)stream_ix * RG1;
    for (int i = wave; i * 64 < RG1; i += NWD) {
      const int idx = i * 64 + lane;
      if (idx < RG1)
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(rs_src + idx), (lds_void_ptr)(size_t)(rowstart_off + (uint32_t)i * 256u), 4, 0, 0);
    }
  }
  // (2) the entry stream of this wave: fixed addresses, PD steps ahead in a register ring with compile-time slots
  const uint32_t voff = (uint32_t)lane * (EB == 3 ? 12u : 16u);
  const uint32_t ebase = EB == 3 ? wbase + (uint32_t)p.T * 8u : wbase;
  auto fetch = [&](int t) -> ring_t {  // unconditional (a load under a branch derails hipcc's wait counts); steps past the
    // end of the range get an out-of-range offset: the buffer unit answers them with zeros and touches no memory
#ifdef AQLM_PACKED_TRACE
    const uint32_t vo = (t <= Tm1 && !(p.dbg & 2)) ? voff : 0xfffffff0u;
#else
    const uint32_t vo = t <= Tm1 ? voff : 0xfffffff0u;
#endif
    if constexpr (EB == 3) return __builtin_amdgcn_raw_buffer_load_b96(rs_ent, vo, ebase + (uint32_t)t * (uint32_t)PK_STEP3, AUX_NT);
    else return __builtin_amdgcn_raw_buffer_load_b128(rs_ent, vo, ebase + (uint32_t)t * 1024u, AUX_NT);
  };
  ring_t ring[PD];
#pragma unroll
  for (int k = 0; k < PD; ++k) ring[k] = fetch(pfw ? 0x7fffffff : k);  // prefetch waves: out-of-range requests (zeros, no memory traffic)
  // (3) steps of this wave through the scalar cache (not a VMEM op: it must not sit in the vmcnt queue, see (5))
  // 4-byte entries need neither: the start rows ride in the entries, and every wave range runs all T steps (the tail of a
  // stream is padded with null entries up to T steps -- the workgroup waits for its full ranges anyway)
  int steps = wave < p.NW ? p.T : 0;
  [[maybe_unused]] uint32_t wave_start_row = 0u;
  if constexpr (EB == 3) {
    const const_u32_ptr wi = (const_u32_ptr)(uintptr_t)(p.winfo + ((size_t)stream_ix * p.NW + wv) * 4);
    steps = wave < p.NW ? (int)wi[2] : 0;
    wave_start_row = wi[3];
  }
  // (4) LDS that needs no data: the zero vectors the null entries point at
  if (tid < B * (int)(PK_VB / 16))  // the null entries' x: PK_VB zero bytes per row of x
    *reinterpret_cast<u32x4*>(smem_raw + LDS::X + (uint32_t)(tid / (int)(PK_VB / 16)) * XP + (uint32_t)p.in_groups * PK_VB +
                              (uint32_t)(tid % (int)(PK_VB / 16)) * 16u) = u32x4{0u, 0u, 0u, 0u};
  const uint32_t xmax_off = LDS::xmax(p.in_groups, p.RG);
  if (tid < B * PK_MAX_NW) *reinterpret_cast<uint32_t*>(smem_raw + xmax_off + (uint32_t)tid * 4u) = 0u;  // slots of absent waves
  AQLM_TRACE(1);  // every load of the prologue has been issued
  // (5) the slice and x are older in the VMEM queue than the PD ring loads: wait for everything BUT the ring, so the
  // stream keeps flowing while the loop starts (a __syncthreads() here would emit vmcnt(0) and drain it)
  // (the builtin, not an asm string: hipcc's wait-count pass must learn that the LDS-DMA ops have retired, or it guards
  // the first use of the ring with vmcnt(0))
  // the parameters of the epilogue (the non-preloaded tail of the kernel arguments) are fetched NOW, under the LDS fill:
  // left to the compiler their s_load sits at the first use, behind the loop, with its whole latency exposed (0.3 us)
  asm volatile("" : : "s"(p.acc), "s"(p.partial), "s"(p.scales), "s"(p.bias), "s"(p.y), "s"(p.y_row_stride), "s"(p.cb_absmax));
  // ... and so are scale and bias of the row this thread finalizes (cold they are an HBM round trip: requested behind the
  // last-arrival test they sat at the very end of the kernel's critical path).  Unconditional, always-valid addresses
  // (a branch around a load ends in a vmcnt(0) at the join); younger than the ring, so the wait below lets them fly too.
  uint16_t scale_h, bias_h;
  {
    const int r = tid < nrows ? tid : (nrows > 0 ? nrows - 1 : 0);
    const uint16_t* sp = p.scales ? p.scales : reinterpret_cast<const uint16_t*>(p.rowstart);  // partials mode: unused
    const uint16_t* bp = p.bias ? p.bias : sp;
    scale_h = sp[row_begin + r];
    bias_h = bp[row_begin + r];
  }
  constexpr int PDW = PD + 2;
  if (!pfw) __builtin_amdgcn_s_waitcnt((PDW & 15) | (7 << 4) | (0 << 8) | ((PDW >> 4) << 14));  // vmcnt(PD + 2) lgkmcnt(0)
  else __builtin_amdgcn_s_waitcnt(63 | (7 << 4) | (0 << 8) | (3 << 14));                          // prefetch waves: lgkmcnt(0) only
  __builtin_amdgcn_s_barrier();
  AQLM_TRACE(2);

  uint32_t mask = PK_HMASK;
  asm volatile("" : "+v"(mask));  // the SDWA operand must sit in a VGPR
  // One accumulator chain per row of x for every batch size: a row's result must not depend on how many rows share the
  // launch (tested bit for bit).  Several independent chains per row were measured: no gain (the loop is not bound by
  // the dependent latency of v_dot2c).
  constexpr int NA = 1;
  float acc[B][NA];
#pragma unroll
  for (int b = 0; b < B; ++b)
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[b][a] = 0.f;
  uint32_t row_addr = 0;  // LDS byte address of rowval[0][current row of this column]

  // One lane-step = 4 entries.  Per entry: a_cb / a_x = LDS byte offsets of the codebook vector (inside the slice) and of
  // x[j] (inside the x area; for B > 1 `copy` is the x copy the entry names -- the planes hold one, so its offset is
  // taken out again).  ALL LDS reads of a group of entries are issued before the first dot product: with 2 waves per SIMD
  // the loop is bound by LDS latency, not bandwidth (traced: 0.3 us per step with two entries in flight per wave).
  constexpr int NV = (int)(PK_VB / 16);  // 16-byte reads per vector: 1 (8 elements) or 2 (16 elements)
  constexpr int EG0 = B <= 2 ? 4 : (B <= 4 ? 2 : 1);
  constexpr int EG = NV > 1 && EG0 > 1 && B > 1 ? EG0 / 2 : EG0;  // entries per read batch (registers: EG * NV * (1 + B) * 4)
  auto entries = [&](const uint32_t (&a_cb)[4], const uint32_t (&a_x)[4], const uint32_t (&copy)[4]) {
#ifdef AQLM_PACKED_TRACE
    if (p.dbg & 1) { acc[0][0] += __uint_as_float(a_cb[0] ^ a_x[1] ^ a_cb[2] ^ a_x[3]); return; }
#endif
#pragma unroll
    for (int g0 = 0; g0 < 4; g0 += EG) {
      u32x4 ev[EG][NV], xv[EG][B][NV];
#ifdef AQLM_PACKED_TRACE
      if (p.dbg & 8) {  // no LDS reads: the dot products run on register garbage (what does the VALU part cost alone?)
#pragma unroll
        for (int k = 0; k < EG; ++k)
#pragma unroll
          for (int h = 0; h < NV; ++h) {
            ev[k][h] = u32x4{a_cb[g0 + k], a_x[g0 + k], a_cb[g0 + k] ^ 0x3c00u, a_x[g0 + k] ^ 0x3c00u};
#pragma unroll
            for (int b = 0; b < B; ++b) xv[k][b][h] = u32x4{a_x[g0 + k], a_cb[g0 + k], a_x[g0 + k] ^ 0x3c00u, a_cb[g0 + k]};
          }
      } else
#endif
#pragma unroll
      for (int k = 0; k < EG; ++k) {
#pragma unroll
        for (int h = 0; h < NV; ++h) ev[k][h] = *(lds_u32x4_ptr)(size_t)((a_cb[g0 + k] ^ ((uint32_t)h * 16u)) + LDS::SLICE);
        if constexpr (B == 1) {
#pragma unroll
          for (int h = 0; h < NV; ++h) xv[k][0][h] = *(lds_u32x4_ptr)(size_t)((a_x[g0 + k] ^ ((uint32_t)h * 16u)) + LDS::X);
        } else {
#pragma unroll
          for (int h = 0; h < NV; ++h) {
            const uint32_t ax = (a_x[g0 + k] ^ ((uint32_t)h * 16u)) + LDS::X - copy[g0 + k] * xstride16;
#pragma unroll
            for (int b = 0; b < B; ++b) xv[k][b][h] = *(lds_u32x4_ptr)(size_t)(ax + (uint32_t)b * XP);
          }
        }
      }
#ifdef AQLM_PACKED_TRACE
      if (p.dbg & 4) {  // LDS reads but no dot products: one op per entry keeps the reads alive
#pragma unroll
        for (int k = 0; k < EG; ++k) acc[0][0] += __uint_as_float((ev[k][NV - 1].x ^ xv[k][0][NV - 1].w) & 0x007fffffu);
        continue;
      }
#endif
#pragma unroll
      for (int k = 0; k < EG; ++k)
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
          for (int h = 0; h < NV; ++h) acc[b][(g0 + k) % NA] = dot8<T_>(ev[k][h], xv[k][b][h], acc[b][(g0 + k) % NA]);
    }
  };
  auto total = [&](int b) -> float {  // fixed summation order of the chains
    float v = acc[b][0];
#pragma unroll
    for (int a = 1; a < NA; ++a) v += acc[b][a];
    return v;
  };
  auto flush = [&]() {  // a row ends here: exactly one lane-step per row does, so the store has a unique writer
#pragma unroll
    for (int b = 0; b < B; ++b) {
      lds_store_f32(row_addr + (uint32_t)(b * RG1) * 4u, total(b));
#pragma unroll
      for (int a = 0; a < NA; ++a) acc[b][a] = 0.f;
    }
    row_addr += 4u;
  };
  uint32_t cmask = 0u;  // 3-byte entries: bit t = this column's lane-step t ends a row
  [[maybe_unused]] uint32_t xrecip = 0u;
  if constexpr (EB == 3 && B > 1) xrecip = 0xffffffffu / (xstride16 >> 4) + 1u;  // slot / stride == mulhi(slot, xrecip) for slot < 2^16
  auto step = [&](const ring_t& e) {
    uint32_t a_cb[4], a_x[4], copy[4] = {0u, 0u, 0u, 0u};
    if constexpr (EB == 4) {
      const uint32_t w[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a_cb[k] = half_and<0>(w[k], mask);  // (plain v_and / v_lshrrev instead of the two SDWA ops: measured, no change)
        a_x[k] = half_and<1>(w[k], mask);
        if constexpr (B > 1) copy[k] = (w[k] >> 16) & 3u;
      }
      entries(a_cb, a_x, copy);
      if (e.x & 1u) flush();
    } else {
      // 96 bits = 4 x (slot:12 | code:12), entry k at bit 24 k
      const uint32_t w0 = e.x, w1 = e.y, w2 = e.z;
      const uint32_t t1 = __builtin_amdgcn_alignbit(w1, w0, 24);   // bits 24.. : code 1 in [11:0]
      const uint32_t t2 = __builtin_amdgcn_alignbit(w2, w1, 28);   // bits 60.. : slot 2 in [11:0]
      a_cb[0] = (w0 << 4) & 0xfff0u;   a_x[0] = (w0 >> 8) & 0xfff0u;
      a_cb[1] = (t1 << 4) & 0xfff0u;   a_x[1] = w1 & 0xfff0u;
      a_cb[2] = (w1 >> 12) & 0xfff0u;  a_x[2] = (t2 << 4) & 0xfff0u;
      a_cb[3] = (w2 >> 4) & 0xfff0u;   a_x[3] = half_and<1>(w2, mask);
      if constexpr (B > 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) copy[k] = __umulhi(a_x[k] >> 4, xrecip);
      }
      entries(a_cb, a_x, copy);
      if (cmask & 1u) flush();
      cmask >>= 1;
    }
  };

  if (steps > 0) {
    if constexpr (EB == 4) {  // the column's starting row rides in the spare bits of its first lane-step
      const uint32_t f = pk_get_start_row(ring[0].x, ring[0].y, ring[0].z, ring[0].w);
      row_addr = rowval_off + f * 4u;
    } else {  // flag word t sits in lane t: collect this lane's bit of every word, and count the row ends of the columns before it
      uint32_t pre = 0u;
      const uint32_t l31 = (uint32_t)lane & 31u;
      for (int t = 0; t < steps; ++t) {
        const uint32_t slo = (uint32_t)__builtin_amdgcn_readlane((int)flagw.x, t), shi = (uint32_t)__builtin_amdgcn_readlane((int)flagw.y, t);
        const uint32_t sel = lane >= 32 ? shi : slo;
        cmask |= ((sel >> l31) & 1u) << t;
        pre = __builtin_amdgcn_mbcnt_hi(shi, __builtin_amdgcn_mbcnt_lo(slo, pre));
      }
      row_addr = rowval_off + (wave_start_row + pre) * 4u;
    }
    int t = 0;
    for (; t + PD <= steps; t += PD) {
#pragma unroll
      for (int k = 0; k < PD; ++k) {  // single back-edge, static ring slots: no in-flight register is ever copied
#ifdef AQLM_PACKED_TRACE
        // profiling build: split a step into "waiting for its entries" and "LDS reads + dot products" (shader cycles)
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(((PD - 1) & 15) | (7 << 4) | (15 << 8));
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        step(ring[k]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long c2 = __builtin_amdgcn_s_memtime();
        tr_wait += (uint32_t)(c1 - c0);
        tr_work += (uint32_t)(c2 - c1);
#else
        step(ring[k]);                 // the slot's words are dead once their addresses are formed ...
#endif
        ring[k] = fetch(t + PD + k);   // ... so the refill lands in the same registers (no copy at the back-edge)
      }
    }
    const int rem = steps - t;
#pragma unroll
    for (int k = 0; k < PD - 1; ++k)
      if (k < rem) step(ring[k]);
  }
  // Fused finalize: the largest |x| of every input row, as the 15-bit magnitude pattern of the storage type (integer
  // order == magnitude order; a NaN compares above Inf, so it surfaces).  x sits in LDS and every workgroup of the layer
  // sees the same x, so all of them derive the same fixed-point scale from it in the epilogue.  Done AFTER the loop: the
  // waves finish it at different times (the SIMDs favour their older waves), so for most of them this is idle time, and
  // right behind the fill barrier it would stand between every wave and its first lane-step (measured: 0.35 us).
  if (p.acc != nullptr) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int b = 0; b < B; ++b) {
      us2 m = {0, 0};
      for (int idx = tid; idx < p.in_groups * (int)(PK_VB / 16); idx += NT) {
        const u32x4 v = *(lds_u32x4_ptr)(size_t)(LDS::X + (uint32_t)b * (B == 1 ? 0u : XP) + (uint32_t)idx * 16u);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) m = __builtin_elementwise_max(m, __builtin_bit_cast(us2, w[k] & 0x7fff7fffu));
      }
      // wave maximum on the VALU (DPP), one slot per wave: no LDS atomics, no shuffle round trips on the way to the loop
      const uint32_t mm = wave_max_u32(m.x > m.y ? (uint32_t)m.x : (uint32_t)m.y);
      if (lane == 0) *reinterpret_cast<uint32_t*>(smem_raw + xmax_off + (uint32_t)(b * PK_MAX_NW + wave) * 4u) = mm;
    }
  }

  // what the column gathered after its last row end belongs to a row that continues in the next column (0 otherwise)
  if (wave < p.NW) {
#pragma unroll
    for (int b = 0; b < B; ++b) lds_store_f32(colend_off + (uint32_t)((b * PK_MAX_NW + wave) * 64 + lane) * 4u, total(b));
  }
  AQLM_TRACE(4);
  asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");  // the asm LDS stores above are invisible to the compiler's counters
  __syncthreads();
  AQLM_TRACE(5);
  // ---- epilogue: row r = rowval[r] + the column remainders of the columns it crosses, in column order -------------
  float* pub_half = nullptr;
  uint32_t pub_e = 0u;
  if (PUB && p.pub != nullptr && p.acc != nullptr) {  // row-parallel shard: publish instead of writing y (epoch parity picks the half)
    pub_e = __hip_atomic_load(p.pub_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pub_half = p.pub + (size_t)(pub_e & 1u) * p.pub_max_elems;
  }
  {
    const uint32_t* rs = reinterpret_cast<const uint32_t*>(smem_raw + rowstart_off);
    const float* rowval = reinterpret_cast<const float*>(smem_raw + rowval_off);
    const float* colend = reinterpret_cast<const float*>(smem_raw + colend_off);
    const uint32_t T = (uint32_t)p.T;
    for (int r = tid; r < nrows; r += NT) {
      const uint32_t q0 = rs[r], q1 = rs[r + 1];
      const uint32_t c0 = q0 / T, c1 = (q1 - 1u) / T;  // first / last column the row touches (column = wave * 64 + lane)
      float v[B];
#pragma unroll
      for (int b = 0; b < B; ++b) {
        v[b] = rowval[b * RG1 + r];
        for (uint32_t c = c0; c < c1; ++c) v[b] += colend[(size_t)b * PK_MAX_NW * 64 + c];
      }
      if (p.acc == nullptr) {
#pragma unroll
        for (int b = 0; b < B; ++b) p.partial[((size_t)slice * B + b) * p.M + row_begin + r] = v[b];
      } else {
        // Fused finalize.  The slice sum goes into the row's cell as a fixed-point number in bits 63..10 (integer adds
        // commute: the total does not depend on the order the 16 workgroups arrive in), together with +1 in the arrival
        // counter (bits 4..0) and +1 in bits 9..5 if the value is not finite.  The unit 2^-sh comes from a bound every
        // workgroup of the layer computes identically: |slice sum| <= in_features * max|codebook| * max|x| < 2^e, so
        // with sh = 47 - e sixteen addends stay below 2^52 -- no overflow whatever the data, and ~2^-47 of the bound as
        // resolution (fp32 partials carry 2^-24 of their own magnitude).  ONE returning atomic per cell is the whole
        // hand-shake: whoever reads 15 earlier arrivals owns the total, applies scale and bias, rounds once, writes y and
        // puts the cell back to zero for the next launch.
        const int row = row_begin + r;
        unsigned long long old[B], mine[B];
        int sh[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
          uint32_t xm = 0u;  // every wave of the workgroup left the maximum of its share of x
          {                  // (16 slots = four 16-B reads in flight together)
            static_assert(PK_MAX_NW == 16, "four 16-byte reads cover the slots");
            u32x4 sl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sl[q] = *(lds_u32x4_ptr)(size_t)(xmax_off + (uint32_t)(b * PK_MAX_NW + q * 4) * 4u);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t a = sl[q].x > sl[q].y ? sl[q].x : sl[q].y, c = sl[q].z > sl[q].w ? sl[q].z : sl[q].w;
              const uint32_t d = a > c ? a : c;
              xm = d > xm ? d : xm;
            }
          }
          const float bound = (float)p.in_groups * (float)PK_G * p.cb_absmax * T_::to_float((uint16_t)xm);
          int e = 0;
          (void)frexpf(bound, &e);                               // bound < 2^e (e = 0 for bound == 0)
          const bool finite = bound < __builtin_inff() && fabsf(v[b]) <= 2.f * bound;  // false for NaN / Inf anywhere
          sh[b] = PK_FIX_BITS - e;
          const long long q = finite ? __float2ll_rn(ldexpf(v[b], sh[b])) : 0ll;
          mine[b] = ((unsigned long long)q << PK_VAL_SHIFT) + (finite ? 1ull : 1ull + (1ull << PK_CNT_BITS));
          old[b] = __hip_atomic_fetch_add(p.acc + (size_t)b * p.M + row, mine[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool first = r == tid;  // this thread's first (usually only) row: scale and bias were requested in the prologue
        const float scale = pub_half ? 1.f : T_::to_float(first ? scale_h : p.scales[row]);
        const float bias = (pub_half || !p.bias) ? 0.f : T_::to_float(first ? bias_h : p.bias[row]);
#pragma unroll
        for (int b = 0; b < B; ++b) {
          if ((old[b] & PK_CNT_MASK) == (unsigned long long)(PK_S - 1)) {
            const unsigned long long cell = old[b] + mine[b];
            const long long sum = (long long)cell >> PK_VAL_SHIFT;
            float sv = (float)ldexp((double)sum, -sh[b]);
            if ((cell >> PK_CNT_BITS) & PK_CNT_MASK) sv = __builtin_nanf("");
            if (pub_half)  // the shard's fp32 total, visible to the peers (write-through, system scope); scale / bias later
              __hip_atomic_store(pub_half + (size_t)b * p.M + row, sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else
              p.y[(size_t)b * p.y_row_stride + row] = T_::from_float(__builtin_fmaf(sv, scale, bias));
            __hip_atomic_store(p.acc + (size_t)b * p.M + row, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
  }
  if (pub_half) {
    // Every row total of this launch is published by exactly one workgroup before that workgroup arrives here: when all
    // PK_NST workgroups have arrived (8 sharded counters of PK_NST / 8 arrivals, then one of 8 -- a single counter would
    // serialise 256 device-scope atomics), everything is out and the rank's flag goes up.  Stores are drained first
    // (write-through + vmcnt(0) == published, cdna_hip_programming.md Guideline 16 R1).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      uint32_t* shard = p.pub_epoch + 4 + (block & 7);
      const uint32_t a = __hip_atomic_fetch_add(shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a + 1u == (uint32_t)(PK_NST / 8)) {
        __hip_atomic_store(shard, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t t = __hip_atomic_fetch_add(p.pub_epoch + 12, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1u == 8u) {
          __hip_atomic_store(p.pub_epoch + 12, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(p.pub_flag + (pub_e & 1u), pub_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
#ifdef AQLM_PACKED_TRACE
  AQLM_TRACE(6);
  if (p.trace && lane == 0) {
    unsigned long long* o = p.trace + ((size_t)block * PK_MAX_NW + wave) * 8;
    tr[3] = __builtin_readcyclecounter() - cyc0;  // shader cycles from entry to end (slot 3 is not a time stamp)
    tr[7] = ((unsigned long long)tr_wait << 32) | tr_work;  // steps of the main loop: cycles waiting for entries | cycles in LDS reads + dots
    for (int i = 0; i < 8; ++i) o[i] = tr[i];
  }
#endif
}

// What the prologue needs before it can issue its first load comes as individual leading arguments: with
// -amdgpu-kernarg-preload-count (Makefile) the command processor delivers those 14 dwords in SGPRs at wave launch, and
// the cold s_load round trip of the kernel-argument segment leaves the head of the critical path.  Struct arguments
// are not preloaded; the rest of the parameters (needed after the LDS fill) stay in one.
struct PackedGemvRest {
  const uint32_t* winfo;
  float* partial;
  long x_row_stride;
  unsigned long long* acc;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
  long y_row_stride;
  float cb_absmax;
  const uint8_t* next_ent;
  const uint8_t* next_codebook;
  uint32_t next_block_bytes;
  float* pub;
  uint32_t* pub_flag;
  uint32_t* pub_epoch;
  uint32_t pub_max_elems;
#ifdef AQLM_PACKED_TRACE
  unsigned long long* trace;
  int dbg;
#endif
};

template <class T_, int B, int PD, uint32_t XWIN, int EB, bool PUB = false>
__global__ __launch_bounds__(1024) void gemv_1x16_packed_kernel(const uint8_t* codebook, const uint16_t* x, const uint32_t* ent,
                                                                const uint32_t* rowstart, int in_groups, uint32_t geom, int RG,
                                                                uint32_t ent_bytes, int M, const PackedGemvRest rest) {
  // geom: waves 0..7 | x copies 8..11 | prefetch waves 12..14 | rotated fill 15 | steps 16..31
  const int NW = (int)(geom & 0xffu), XC = (int)((geom >> 8) & 0xfu), NPW = (int)((geom >> 12) & 7u), T = (int)(geom >> 16);
  PackedGemvParams p;
  p.NPW = NPW;
  p.fill_rotate = (int)((geom >> 15) & 1u);
  p.next_ent = rest.next_ent;
  p.next_codebook = rest.next_codebook;
  p.next_block_bytes = rest.next_block_bytes;
  p.pub = rest.pub;
  p.pub_flag = rest.pub_flag;
  p.pub_epoch = rest.pub_epoch;
  p.pub_max_elems = rest.pub_max_elems;
  p.ent = ent;
  p.winfo = rest.winfo;
  p.rowstart = rowstart;
  p.codebook = codebook;
  p.x = x;
  p.partial = rest.partial;
  p.acc = rest.acc;
  p.cb_absmax = rest.cb_absmax;
  p.scales = rest.scales;
  p.bias = rest.bias;
  p.y = rest.y;
  p.y_row_stride = rest.y_row_stride;
  p.x_row_stride = rest.x_row_stride;
  p.M = M;
  p.in_groups = in_groups;
  p.RG = RG;
  p.NW = NW;
  p.T = T;
  p.XC = XC;
  p.ent_bytes = ent_bytes;
#ifdef AQLM_PACKED_TRACE
  p.trace = rest.trace;
  p.dbg = rest.dbg;
#endif
  gemv_1x16_packed_body<T_, B, PD, XWIN, EB, PUB>(p, blockIdx.x, NW + NPW);
}

#if AQLM_PK_G == 8
// Variable-geometry twin (format v7, flag AQLM_HIP_PACKED_VARGEOM): same body; the 14 preloaded dwords now also carry the row
// groups of the 16 slices, so three of the ordinary kernel's arguments travel compressed: the row-start table as its distance
// in front of the entries, in_groups and the row-table size in one word, and the entry bytes are derived (4-byte entries only).
template <class T_, int B, int PD, uint32_t XWIN>
__global__ __launch_bounds__(1024) void gemv_1x16_packed_vg_kernel(const uint8_t* codebook, const uint16_t* x, const uint32_t* ent,
                                                                   uint32_t rowstart_back, uint32_t ig_rg, uint32_t geom, int M,
                                                                   uint32_t ns0, uint32_t ns1, uint32_t ns2, uint32_t ns3,
                                                                   const PackedGemvRest rest) {
  // geom: waves 0..7 | x copies 8..11 | rotated fill 15 | steps 16..31;  ig_rg: in_groups 0..11 | rows per group 12..27
  const int NW = (int)(geom & 0xffu), XC = (int)((geom >> 8) & 0xfu), T = (int)(geom >> 16);
  PackedGemvParams p;
  p.NPW = 0;
  p.fill_rotate = (int)((geom >> 15) & 1u);
  p.next_ent = nullptr;
  p.next_codebook = nullptr;
  p.next_block_bytes = 0;
  p.pub = nullptr;
  p.pub_flag = nullptr;
  p.pub_epoch = nullptr;
  p.pub_max_elems = 0;
  p.ent = ent;
  p.winfo = rest.winfo;
  p.rowstart = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(ent) - rowstart_back);
  p.codebook = codebook;
  p.x = x;
  p.partial = rest.partial;
  p.acc = rest.acc;
  p.cb_absmax = rest.cb_absmax;
  p.scales = rest.scales;
  p.bias = rest.bias;
  p.y = rest.y;
  p.y_row_stride = rest.y_row_stride;
  p.x_row_stride = rest.x_row_stride;
  p.M = M;
  p.in_groups = (int)(ig_rg & 0xfffu);
  p.RG = (int)(ig_rg >> 12);
  p.NW = NW;
  p.T = T;
  p.XC = XC;
  p.ent_bytes = (uint32_t)PK_NST * (uint32_t)NW * (uint32_t)T * 1024u;
#ifdef AQLM_PACKED_TRACE
  p.trace = rest.trace;
  p.dbg = rest.dbg;
#endif
  const PackedVgArgs vg{{ns0, ns1, ns2, ns3}};
  gemv_1x16_packed_body<T_, B, PD, XWIN, 4, false, true>(p, blockIdx.x, NW, vg);
}
#endif

// Several prepacked layers that multiply the same x (q/k/v, gate/up) in one launch of 256 workgroups per layer; the
// next layer's workgroups start as CUs free up, so one layer's tail and the next one's LDS fill overlap.
struct PackedSegment {
  const uint32_t* ent;
  const uint32_t* winfo;
  const uint32_t* rowstart;
  const uint8_t* codebook;
  float* partial;
  int M, RG, NW, T, XC;
  uint32_t ent_bytes;
  // fused finalize (acc != nullptr)
  unsigned long long* acc;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
  long y_row_stride;
  float cb_absmax;
};

struct PackedMultiParams {
  const uint16_t* x;
  long x_row_stride;
  int in_groups, nseg;
  PackedSegment seg[AQLM_HIP_MAX_SEGMENTS];
};

template <class T_, int B, int PD, uint32_t XWIN, int EB>
__global__ __launch_bounds__(1024) void gemv_1x16_packed_multi_kernel(const PackedMultiParams mp) {
  const int sidx = (int)blockIdx.x / PK_NST;
  PackedGemvParams p{};
  p.x = mp.x;
  p.x_row_stride = mp.x_row_stride;
  p.in_groups = mp.in_groups;
#pragma unroll
  for (int k = 0; k < AQLM_HIP_MAX_SEGMENTS; ++k) {
    if (k == 0 || sidx == k) {  // scalar select chain (no dynamic indexing of the kernel-argument struct)
      p.ent = mp.seg[k].ent;
      p.winfo = mp.seg[k].winfo;
      p.rowstart = mp.seg[k].rowstart;
      p.codebook = mp.seg[k].codebook;
      p.partial = mp.seg[k].partial;
      p.M = mp.seg[k].M;
      p.RG = mp.seg[k].RG;
      p.NW = mp.seg[k].NW;
      p.T = mp.seg[k].T;
      p.XC = mp.seg[k].XC;
      p.ent_bytes = mp.seg[k].ent_bytes;
      p.acc = mp.seg[k].acc;
      p.scales = mp.seg[k].scales;
      p.bias = mp.seg[k].bias;
      p.y = mp.seg[k].y;
      p.y_row_stride = mp.seg[k].y_row_stride;
      p.cb_absmax = mp.seg[k].cb_absmax;
    }
  }
  gemv_1x16_packed_body<T_, B, PD, XWIN, EB>(p, (int)blockIdx.x % PK_NST, (int)blockDim.x >> 6);
}

struct PackedFinalizeParams {
  const float* partial;  // [S][B][M]
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
  long y_row_stride;
  int M, B;
};

template <class T_>
__device__ __forceinline__ void packed_finalize_row(const PackedFinalizeParams& p, int row) {
  if (row >= p.M) return;
  const float scale = T_::to_float(p.scales[row]);
  const float bias = p.bias ? T_::to_float(p.bias[row]) : 0.f;
  for (int b = 0; b < p.B; ++b) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PK_S; ++k) s += p.partial[((size_t)k * p.B + b) * p.M + row];
    p.y[(size_t)b * p.y_row_stride + row] = T_::from_float(__builtin_fmaf(s, scale, bias));
  }
}

// scalar arguments: preloaded into SGPRs at wave launch (see gemv_1x16_packed_kernel); this kernel is one dependent load
// round trip long, the kernel-argument fetch would be a second one
template <class T_>
__global__ __launch_bounds__(256) void gemv_1x16_packed_finalize(const float* partial, const uint16_t* scales, const uint16_t* bias,
                                                                 uint16_t* y, long y_row_stride, int M, int B) {
  PackedFinalizeParams p;
  p.partial = partial;
  p.scales = scales;
  p.bias = bias;
  p.y = y;
  p.y_row_stride = y_row_stride;
  p.M = M;
  p.B = B;
  packed_finalize_row<T_>(p, blockIdx.x * 256 + threadIdx.x);
}

struct PackedFinalizeSegment {
  PackedFinalizeParams f;
  int block_begin;
};

struct PackedFinalizeMultiParams {
  int nseg;
  PackedFinalizeSegment seg[AQLM_HIP_MAX_SEGMENTS];
};

template <class T_>
__global__ __launch_bounds__(256) void gemv_1x16_packed_finalize_multi(const PackedFinalizeMultiParams mp) {
  PackedFinalizeParams p = mp.seg[0].f;
  int begin = 0;
#pragma unroll
  for (int k = 1; k < AQLM_HIP_MAX_SEGMENTS; ++k) {
    if (k < mp.nseg && (int)blockIdx.x >= mp.seg[k].block_begin) {
      p = mp.seg[k].f;
      begin = mp.seg[k].block_begin;
    }
  }
  packed_finalize_row<T_>(p, ((int)blockIdx.x - begin) * 256 + threadIdx.x);
}


// ---------------------------------------------------------------------------------------------- pipelined shared-input launch
// q/k/v (gate/up) of a decoder layer in ONE launch of 256 workgroups, each walking its (row group, slice) stream of every
// segment in turn with the codebook slices double-buffered: while the compute waves run the loop of segment k, two DMA
// waves pull the slice of segment k + 1 into the other buffer, so only the first fill of the launch is exposed (the
// plain multi kernel above starts a fresh workgroup per segment: fill and loop of a CU never overlap).  Batch 1, 4-byte
// entries, one x copy, single-kernel finalize; everything else takes the plain multi kernel.
//   LDS: x window | slice buffer 0 | slice buffer 1 | rowstart x 2 | rowval | colend | xmax.  The two slice buffers are
//   65536 bytes apart: an entry's codebook address is (word & 0xfff0) | (k & 1) << 16 -- one v_and_or_b32, as many
//   operations per entry as the single-layer kernel -- plus the constant offset of buffer 0 in the ds_read.
//   Waves: NWC compute waves (max over the segments' wave counts) + 2 DMA waves.  The compute waves' VMEM queue holds ring
//   fetches and the returning atomics only (hipcc counts those); the DMA waves hold LDS-DMA only (waited with vmcnt(0)).
//   Layers packed for 15 / 16 waves (SELF_DMA): no DMA waves; every compute wave requests its share of slice k + 1 when its
//   own steps of segment k are done, and the epilogue reads its row tables through asm (hipcc guards every LDS read it sees
//   with vmcnt(0) while LDS-DMA is in flight).
//   Barriers per segment: M_k (loop k done -> epilogue k may read rowval / colend) and F_k (epilogue k done AND slice
//   k + 1 landed).
//   Hand-shake: segment k's returning atomic is looked at in epilogue k + 1 (settle_pending), i.e. behind loop k + 1 -- its
//   round trip is off the critical path; the last segment settles at once.
constexpr uint32_t PP_XWIN = 16640;        // x window: (in_groups + 1) * vector bytes <= the window (inputs of <= 8192 features) ...
constexpr uint32_t PP_XWIN_SMALL = 8448;   // ... or <= 4096 features: 8 KiB more for the tables (template parameter XW of the kernel)
constexpr int PP_DMA_WAVES = 2;

// Bookkeeping behind the two slice buffers.  Two barriers per segment (M_k, F_k): row starts double-buffered, one set of row
// sums.  ONE barrier (ONEB): the compute waves go from epilogue k straight into loop k + 1, so row sums / column ends are
// double-buffered (loop k + 1 writes the other set) and the row starts triple-buffered (the DMA waves request segment
// k + 2's while epilogue k may still read segment k's).
struct PipeLds {
  uint32_t rs0, rs_bytes, rowval, rowval_bytes, colend, colend_bytes, xmax, total;  // row starts of buffer b at rs0 + b * rs_bytes, ...
};
__host__ __device__ static inline PipeLds pipe_lds(int max_rg, uint32_t xwin, bool oneb) {
  PipeLds l;
  l.rs_bytes = oneb ? (((uint32_t)(max_rg + 1) * 4u + 255u) & ~255u) : (((uint32_t)(max_rg + 1) * 4u + 1023u) & ~1023u);
  l.rs0 = xwin + 2u * PK_SLICE_BYTES;
  l.rowval = l.rs0 + (oneb ? 3u : 2u) * l.rs_bytes;
  l.rowval_bytes = ((uint32_t)(max_rg + 1) * 4u + 15u) & ~15u;
  l.colend = l.rowval + (oneb ? 2u : 1u) * l.rowval_bytes;
  l.colend_bytes = (uint32_t)PK_MAX_NW * 64u * 4u;
  l.xmax = l.colend + (oneb ? 2u : 1u) * l.colend_bytes;
  l.total = l.xmax + (uint32_t)PK_MAX_NW * 4u;
  return l;
}

struct PipeParams {
  const uint16_t* x;
  int in_groups, nseg, max_rg, nwc;  // nwc: compute waves of the workgroup
  int dma_waves;                     // PP_DMA_WAVES, or 0: every compute wave requests its share of the next slice itself
  int defer;                         // DMA-wave mode: look at a segment's atomics one segment later (packed_pipe == 4 switches it off)
  PackedSegment seg[AQLM_HIP_MAX_SEGMENTS];
};

__device__ __forceinline__ uint32_t and_or(uint32_t w, uint32_t mask_vgpr, uint32_t base) {
  uint32_t d;  // one scalar operand per VOP3 instruction (constant bus): the mask travels in a VGPR
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(w), "v"(mask_vgpr), "s"(base));
  return d;
}

template <class T_, bool SELF_DMA, uint32_t XW, bool ONEB>
__global__ __launch_bounds__(1024) void gemv_1x16_packed_pipe_kernel(const PipeParams mp) {
  static_assert(!(SELF_DMA && ONEB), "the single-barrier form needs the DMA waves");
  constexpr uint32_t PP_BUF0 = XW;
#ifndef AQLM_PP_PD
#define AQLM_PP_PD 3
#endif
  constexpr int PD = AQLM_PP_PD;  // entry steps in flight per wave (and requested ahead for the next segment)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw != 0u) __builtin_trap();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NWC = mp.nwc;
  // Layers packed for 15 or 16 waves leave no room for DMA waves: then every compute wave requests its share of segment
  // k + 1's slice when its own steps of segment k are done -- the requests land under the epilogue (barrier, row sums,
  // atomic round trip).  Not earlier: loads return in order, so a request in front of ring fetches would stall the loop
  // for the slice's latency, and hipcc guards VGPR loads with vmcnt(0) while LDS-DMA is in flight.
  constexpr bool self_dma = SELF_DMA;  // == (mp.dma_waves == 0)
  const bool dma_wave = wave >= NWC;
  const int dw = wave - NWC;  // 0 / 1 for the DMA waves
  const int block = (int)blockIdx.x;
  const int slice = block & (PK_S - 1), group = block >> PK_S_LOG;
  const PipeLds L = pipe_lds(mp.max_rg, XW, ONEB);
  const int NTC = NWC << 6;  // compute threads

  // scalar select of a segment's parameters (no dynamic indexing of the kernel-argument struct)
  auto segment = [&](int k) -> PackedSegment {
    PackedSegment s = mp.seg[0];
#pragma unroll
    for (int q = 1; q < AQLM_HIP_MAX_SEGMENTS; ++q)
      if (k == q) s = mp.seg[q];
    return s;
  };
  auto dma_slice = [&](const PackedSegment& s, int buf, int first, int stride) {  // pieces first, first + stride, ... of the slice
    const uint8_t* src = s.codebook + (size_t)slice * PK_SLICE_BYTES;
    constexpr int PIECES = (int)(PK_SLICE_BYTES / 1024);
    const int rot = group * (PIECES / PK_NG);
    for (int i0 = first; i0 < PIECES; i0 += stride) {
      const int i = (i0 + rot) & (PIECES - 1);
      __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + i * 1024 + lane * 16),
                                       (lds_void_ptr)(size_t)(PP_BUF0 + (uint32_t)buf * PK_SLICE_BYTES + (uint32_t)i * 1024u), 16, 0, 0);
    }
  };
  auto rs_buf = [](int k) -> int { return ONEB ? k % 3 : (k & 1); };  // row-start buffer of segment k
  auto dma_rowstart = [&](const PackedSegment& s, int buf, int first, int stride) {
    const int RG1 = s.RG + 1;
    const uint32_t* rs_src = s.rowstart + (size_t)block * RG1;
    for (int i = first; i * 64 < RG1; i += stride) {
      const int idx = i * 64 + lane;
      if (idx < RG1)
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(rs_src + idx), (lds_void_ptr)(size_t)(L.rs0 + (uint32_t)buf * L.rs_bytes + (uint32_t)i * 256u), 4, 0, 0);  // buf: rs_buf(segment)
    }
  };

  // ---- prologue: every wave helps with the first fill (slice 0, x, row starts 0), as in the single-layer kernel ---------
  PackedSegment s0 = segment(0);
  const int NWB = NWC + mp.dma_waves;
  dma_slice(s0, 0, wave, NWB);
  {
    const int x16 = mp.in_groups * (int)(PK_VB / 16);  // 16-byte units of x
    const int nchunk = (x16 + 63) >> 6;
    for (int c = wave; c < nchunk; c += NWB) {
      const int idx = c * 64 + lane;
      if (idx < x16)
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(mp.x + (size_t)idx * 8), (lds_void_ptr)(size_t)((uint32_t)c * 1024u), 16, 0, 0);
    }
  }
  dma_rowstart(s0, rs_buf(0), wave, NWB);
  if (tid < (int)(PK_VB / 16))  // the null entries' x
    *reinterpret_cast<u32x4*>(smem_raw + (uint32_t)mp.in_groups * PK_VB + (uint32_t)tid * 16u) = u32x4{0u, 0u, 0u, 0u};
  if (tid < PK_MAX_NW) *reinterpret_cast<uint32_t*>(smem_raw + L.xmax + (uint32_t)tid * 4u) = 0u;

  uint32_t mask = PK_HMASK;
  asm volatile("" : "+v"(mask));  // SDWA operand in a VGPR

  // Epilogue of segment k by the threads t0, t0 + nt, ... (single-barrier form): row sums -> fixed-point cells -> the last
  // arrival writes y.  Two phases -- the atomics of up to four rows of a thread go out before the first answer is looked at --
  // and asm LDS reads (the DMA waves run it with LDS-DMA in flight; see lds_asm_load_b32).
  auto epilogue_rows = [&](int k, int t0, int nt) {
    const PackedSegment s = segment(k);
    int nrows = s.M - group * s.RG;
    nrows = nrows < 0 ? 0 : (nrows < s.RG ? nrows : s.RG);
    const uint32_t T = (uint32_t)s.T;
    const int row_begin = group * s.RG;
    const uint32_t rs_off = L.rs0 + (uint32_t)rs_buf(k) * L.rs_bytes;
    const uint32_t rowval_k = L.rowval + (ONEB ? (uint32_t)(k & 1) * L.rowval_bytes : 0u);
    const uint32_t colend_k = L.colend + (ONEB ? (uint32_t)(k & 1) * L.colend_bytes : 0u);
    if (t0 >= nrows) return;
    uint32_t xm = 0u;
    {
      u32x4 sl0 = lds_asm_load_b128(L.xmax), sl1 = lds_asm_load_b128(L.xmax + 16u), sl2 = lds_asm_load_b128(L.xmax + 32u),
            sl3 = lds_asm_load_b128(L.xmax + 48u);
      uint32_t dummy0 = 0u, dummy1 = 0u, dummy2 = 0u;
      lds_asm_wait(dummy0, dummy1, dummy2, sl0, sl1, sl2, sl3);
      const u32x4 sl[4] = {sl0, sl1, sl2, sl3};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t a = sl[q].x > sl[q].y ? sl[q].x : sl[q].y, c = sl[q].z > sl[q].w ? sl[q].z : sl[q].w;
        const uint32_t d = a > c ? a : c;
        xm = d > xm ? d : xm;
      }
    }
    const float bound = (float)mp.in_groups * (float)PK_G * s.cb_absmax * T_::to_float((uint16_t)xm);
    int e = 0;
    (void)frexpf(bound, &e);
    const int sh = PK_FIX_BITS - e;
    const uint16_t* bias_src = s.bias ? s.bias : s.scales;
    for (int r0 = t0; r0 < nrows; r0 += 4 * nt) {
      unsigned long long old[4], mine[4];
      uint16_t sc[4], bi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = r0 + i * nt;
        old[i] = mine[i] = 0ull;
        sc[i] = bi[i] = 0;
        if (r < nrows) {
          uint32_t q0 = lds_asm_load_b32(rs_off + (uint32_t)r * 4u), q1 = lds_asm_load_b32(rs_off + (uint32_t)r * 4u + 4u);
          uint32_t vbits = lds_asm_load_b32(rowval_k + (uint32_t)r * 4u);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(vbits));
          float v = __uint_as_float(vbits);
          const uint32_t c0 = q0 / T;
          uint32_t c1 = (q1 - 1u) / T;
          c1 = c1 < (uint32_t)(PK_MAX_NW * 64) ? c1 : (uint32_t)(PK_MAX_NW * 64 - 1);
          for (uint32_t c = c0; c < c1; ++c) {
            uint32_t ce = lds_asm_load_b32(colend_k + c * 4u);
            lds_asm_wait(ce);
            v += __uint_as_float(ce);
          }
          const bool finite = bound < __builtin_inff() && fabsf(v) <= 2.f * bound;
          const long long qv = finite ? __float2ll_rn(ldexpf(v, sh)) : 0ll;
          mine[i] = ((unsigned long long)qv << PK_VAL_SHIFT) + (finite ? 1ull : 1ull + (1ull << PK_CNT_BITS));
          const int row = row_begin + r;
          old[i] = __hip_atomic_fetch_add(s.acc + row, mine[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sc[i] = s.scales[row];
          bi[i] = bias_src[row];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = r0 + i * nt;
        if (r < nrows && (old[i] & PK_CNT_MASK) == (unsigned long long)(PK_S - 1)) {
          const int row = row_begin + r;
          const unsigned long long cell = old[i] + mine[i];
          const long long sum = (long long)cell >> PK_VAL_SHIFT;
          float sv = (float)ldexp((double)sum, -sh);
          if ((cell >> PK_CNT_BITS) & PK_CNT_MASK) sv = __builtin_nanf("");
          s.y[row] = T_::from_float(__builtin_fmaf(sv, T_::to_float(sc[i]), s.bias ? T_::to_float(bi[i]) : 0.f));
          __hip_atomic_store(s.acc + row, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  };

  // Single-barrier form: who runs segment k's epilogue?  The two DMA waves, during loop k + 1, when that costs the compute
  // waves nothing -- at most two rows per DMA thread and a next loop at least as long as this one (measured: with the short
  // k / v loops behind q, or 688-row groups, the DMA waves become the critical path); else the compute waves, right after M_k.
  auto epi_on_dma = [&](int k) -> bool {
    if (!ONEB || k + 1 >= mp.nseg) return false;
    const PackedSegment a = segment(k), b = segment(k + 1);
    int nrows = a.M - group * a.RG;
    nrows = nrows < 0 ? 0 : (nrows < a.RG ? nrows : a.RG);
    return nrows <= 2 * PP_DMA_WAVES * 64 && b.NW * b.T >= a.NW * a.T;
  };

  if (dma_wave) {
    // ============================================ DMA waves ================================================================
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8));  // vmcnt(0) lgkmcnt(0): my share of the first fill has landed
    __builtin_amdgcn_s_barrier();                          // B0
    for (int k = 0; k < mp.nseg; ++k) {
      if (k + 1 < mp.nseg) {
        const PackedSegment sn = segment(k + 1);
        dma_slice(sn, (k + 1) & 1, dw, PP_DMA_WAVES);
        dma_rowstart(sn, rs_buf(k + 1), dw, PP_DMA_WAVES);
      }
      if constexpr (ONEB) {
        if (k >= 1 && epi_on_dma(k - 1)) epilogue_rows(k - 1, dw * 64 + lane, PP_DMA_WAVES * 64);  // while the compute waves run loop k
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8));  // slice k + 1 is in LDS when the compute waves leave M_k
        __builtin_amdgcn_s_barrier();                        // M_k
      } else {
        __builtin_amdgcn_s_barrier();                        // M_k
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8));
        __builtin_amdgcn_s_barrier();                        // F_k: slice k + 1 is in LDS
      }
    }
    return;
  }

  // ============================================== compute waves =============================================================
  __amdgpu_buffer_rsrc_t rs_ent = __builtin_amdgcn_make_buffer_rsrc((void*)s0.ent, 0, s0.ent_bytes, 0x00020000);
  auto fetch_from = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t wbase, int Tm1, int t) -> u32x4 {
    const uint32_t vo = t <= Tm1 ? (uint32_t)lane * 16u : 0xfffffff0u;  // past the range: zeros, no memory traffic
    return __builtin_amdgcn_raw_buffer_load_b128(rs, vo, wbase + (uint32_t)t * 1024u, AUX_NT);
  };
  u32x4 ring[PD];
  {
    const int wv = wave < s0.NW ? wave : s0.NW - 1;
    const uint32_t wbase = (uint32_t)(((size_t)block * s0.NW + wv) * s0.T) * 1024u;
#pragma unroll
    for (int j = 0; j < PD; ++j) ring[j] = fetch_from(rs_ent, wbase, wave < s0.NW ? s0.T - 1 : -1, j);
  }
  __builtin_amdgcn_s_waitcnt((PD & 15) | (7 << 4) | (0 << 8) | ((PD >> 4) << 14));  // vmcnt(PD): the fill, not the ring
  __builtin_amdgcn_s_barrier();                            // B0
  // the first ring steps: waited for here, by the builtin, so that the segment loop is ENTERED with nothing pending -- hipcc's
  // wait-count pass merges the entry state with the back edge's (a deferred atomic in flight) and would otherwise guard the
  // first use of the ring with vmcnt(0) on every segment (it emitted this wait itself before; now it knows)
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));

  // A row's hand-shake (returning atomic) is looked at one segment LATER, behind the next segment's loop, so its round trip
  // to the memory side is off the critical path (the compute waves wait for nothing else between segments; in the
  // self-service mode the wait for the LDS-DMA is counted so that it leaves the atomic in flight).  Only when every thread
  // has at most one row per segment.
  const bool tuning_defer = mp.defer != 0;
  bool defer_k = false;
  bool pend = false;
  unsigned long long pend_old = 0ull, pend_mine = 0ull;
  unsigned long long* pend_cell = nullptr;
  uint16_t* pend_y = nullptr;
  uint16_t pend_scale = 0, pend_bias = 0;
  int pend_sh = 0;
  bool pend_has_bias = false;
  auto settle_pending = [&]() {
    if (pend && (pend_old & PK_CNT_MASK) == (unsigned long long)(PK_S - 1)) {
      const unsigned long long cell = pend_old + pend_mine;
      const long long sum = (long long)cell >> PK_VAL_SHIFT;
      float sv = (float)ldexp((double)sum, -pend_sh);
      if ((cell >> PK_CNT_BITS) & PK_CNT_MASK) sv = __builtin_nanf("");
      *pend_y = T_::from_float(__builtin_fmaf(sv, T_::to_float(pend_scale), pend_has_bias ? T_::to_float(pend_bias) : 0.f));
      __hip_atomic_store(pend_cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    pend = false;
  };
  for (int k = 0; k < mp.nseg; ++k) {
    const PackedSegment s = segment(k);
    const int RG1 = s.RG + 1;
    int nrows = s.M - group * s.RG;
    nrows = nrows < 0 ? 0 : (nrows < s.RG ? nrows : s.RG);
    const int steps = wave < s.NW ? s.T : 0;
    const uint32_t bufsel = (uint32_t)(k & 1) << 16;
    const uint32_t rowval_k = L.rowval + (ONEB ? (uint32_t)(k & 1) * L.rowval_bytes : 0u);  // this segment's row sums / column ends
    const uint32_t colend_k = L.colend + (ONEB ? (uint32_t)(k & 1) * L.colend_bytes : 0u);
    const uint32_t wbase = (uint32_t)(((size_t)block * s.NW + (wave < s.NW ? wave : s.NW - 1)) * s.T) * 1024u;
    const int Tm1 = steps > 0 ? s.T - 1 : -1;
    // the first steps of the NEXT segment's entry stream are requested now: by the time the epilogue below waits for its
    // atomics they have long landed (VMEM returns in order), and loop k + 1 starts on data that is already here
    u32x4 ring_next[PD];
    __amdgpu_buffer_rsrc_t rs_next = rs_ent;
    if (k + 1 < mp.nseg) {
      const PackedSegment sn = segment(k + 1);
      rs_next = __builtin_amdgcn_make_buffer_rsrc((void*)sn.ent, 0, sn.ent_bytes, 0x00020000);
      const int wv = wave < sn.NW ? wave : sn.NW - 1;
      const uint32_t wb = (uint32_t)(((size_t)block * sn.NW + wv) * sn.T) * 1024u;
#pragma unroll
      for (int j = 0; j < PD; ++j) ring_next[j] = fetch_from(rs_next, wb, wave < sn.NW ? sn.T - 1 : -1, j);
    } else {
#pragma unroll
      for (int j = 0; j < PD; ++j) ring_next[j] = u32x4{0u, 0u, 0u, 0u};
    }
    float acc = 0.f;
    uint32_t row_addr = 0;
    auto step = [&](const u32x4& e) {
      const uint32_t w[4] = {e.x, e.y, e.z, e.w};
      uint32_t a_cb[4], a_x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a_cb[j] = and_or(w[j], mask, bufsel);
        a_x[j] = half_and<1>(w[j], mask);
      }
      constexpr int NV = (int)(PK_VB / 16);  // 16-byte reads per vector (2: second half at offset ^ 16, see PK_PARITY_BITS)
      u32x4 ev[4][NV], xv[4][NV];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < NV; ++h) {
          ev[j][h] = *(lds_u32x4_ptr)(size_t)((a_cb[j] ^ ((uint32_t)h * 16u)) + PP_BUF0);
          xv[j][h] = *(lds_u32x4_ptr)(size_t)(a_x[j] ^ ((uint32_t)h * 16u));
        }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < NV; ++h) acc = dot8<T_>(ev[j][h], xv[j][h], acc);
      if (e.x & 1u) {  // a row ends here: unique writer
        lds_store_f32(row_addr, acc);
        acc = 0.f;
        row_addr += 4u;
      }
    };
    if (steps > 0) {
      row_addr = rowval_k + pk_get_start_row(ring[0].x, ring[0].y, ring[0].z, ring[0].w) * 4u;
      int t = 0;
      for (; t + PD <= steps; t += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
          step(ring[j]);
          ring[j] = fetch_from(rs_ent, wbase, Tm1, t + PD + j);
        }
      }
      const int rem = steps - t;
#pragma unroll
      for (int j = 0; j < PD - 1; ++j)
        if (j < rem) step(ring[j]);
    }
    if (k == 0) {  // largest |x| (every segment multiplies the same x): per-wave maxima, once
      typedef unsigned short us2 __attribute__((ext_vector_type(2)));
      us2 m = {0, 0};
      for (int idx = tid; idx < mp.in_groups * (int)(PK_VB / 16); idx += NTC) {
        const u32x4 v = *(lds_u32x4_ptr)(size_t)((uint32_t)idx * 16u);
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) m = __builtin_elementwise_max(m, __builtin_bit_cast(us2, w4[j] & 0x7fff7fffu));
      }
      const uint32_t mm = wave_max_u32(m.x > m.y ? (uint32_t)m.x : (uint32_t)m.y);
      if (lane == 0) *reinterpret_cast<uint32_t*>(smem_raw + L.xmax + (uint32_t)wave * 4u) = mm;
    }
    if (wave < s.NW) lds_store_f32(colend_k + (uint32_t)(wave * 64 + lane) * 4u, acc);
    // every entry fetch of this wave has landed (the next segment's first steps were requested a loop ago): said with the
    // builtin on EVERY path, so that hipcc's wait-count pass knows no VGPR load is pending when the epilogue reuses the
    // ring's registers -- otherwise it guards that reuse with vmcnt(0), i.e. waits for the LDS-DMA issued just below
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));  // (both modes: the deferred atomic of the DMA-wave mode must be the only thing pending at the loop's back edge)
    if (self_dma && k + 1 < mp.nseg) {  // this wave's share of the next slice and row-start table (buffer (k + 1) & 1 is free: its
      const PackedSegment sn = segment(k + 1);  // last readers passed F_{k-1})
      dma_slice(sn, (k + 1) & 1, wave, NWC);
      dma_rowstart(sn, rs_buf(k + 1), wave, NWC);
    }
    rs_ent = rs_next;
#pragma unroll
    for (int j = 0; j < PD; ++j) ring[j] = ring_next[j];
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");  // the asm LDS stores are invisible to the compiler's counters
    __builtin_amdgcn_s_barrier();                          // M_k
    // ---- epilogue of segment k (compute waves): row sums -> fixed-point cell -> last arrival writes y ---------------------
    // (single-barrier form: where epi_on_dma(k) says so the DMA waves do it while this wave is already in loop k + 1)
    if (!epi_on_dma(k)) {
      const uint32_t rs_off = L.rs0 + (uint32_t)rs_buf(k) * L.rs_bytes;
      const uint32_t T = (uint32_t)s.T;
      const int row_begin = group * s.RG;
      settle_pending();  // segment k - 1's rows: their atomics went out one loop ago
      const bool defer = nrows <= NTC && k + 1 < mp.nseg && tuning_defer;
      defer_k = defer;
      for (int r = tid; r < nrows; r += NTC) {
        uint32_t q0, q1;
        float v;
        u32x4 sl[4];
        if constexpr (SELF_DMA) {  // asm reads: segment k + 1's slice is on its way into the other buffer (lds_asm_load_b32)
          q0 = lds_asm_load_b32(rs_off + (uint32_t)r * 4u);
          q1 = lds_asm_load_b32(rs_off + (uint32_t)r * 4u + 4u);
          uint32_t vbits = lds_asm_load_b32(rowval_k + (uint32_t)r * 4u);
#pragma unroll
          for (int q = 0; q < 4; ++q) sl[q] = lds_asm_load_b128(L.xmax + (uint32_t)(q * 4) * 4u);
          lds_asm_wait(q0, q1, vbits, sl[0], sl[1], sl[2], sl[3]);
          v = __uint_as_float(vbits);
          const uint32_t c0 = q0 / T;
          uint32_t c1 = (q1 - 1u) / T;
          c1 = c1 < (uint32_t)(PK_MAX_NW * 64) ? c1 : (uint32_t)(PK_MAX_NW * 64 - 1);  // a column index, whatever was read
          for (uint32_t c = c0; c < c1; ++c) {
            uint32_t ce = lds_asm_load_b32(colend_k + c * 4u);
            lds_asm_wait(ce);
            v += __uint_as_float(ce);
          }
        } else {
          const uint32_t* rs = reinterpret_cast<const uint32_t*>(smem_raw + rs_off);
          const float* rowval = reinterpret_cast<const float*>(smem_raw + rowval_k);
          const float* colend = reinterpret_cast<const float*>(smem_raw + colend_k);
          q0 = rs[r];
          q1 = rs[r + 1];
          const uint32_t c0 = q0 / T, c1 = (q1 - 1u) / T;
          v = rowval[r];
          for (uint32_t c = c0; c < c1; ++c) v += colend[c];
#pragma unroll
          for (int q = 0; q < 4; ++q) sl[q] = *(lds_u32x4_ptr)(size_t)(L.xmax + (uint32_t)(q * 4) * 4u);
        }
        uint32_t xm = 0u;
        {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t a = sl[q].x > sl[q].y ? sl[q].x : sl[q].y, c = sl[q].z > sl[q].w ? sl[q].z : sl[q].w;
            const uint32_t d = a > c ? a : c;
            xm = d > xm ? d : xm;
          }
        }
        const float bound = (float)mp.in_groups * (float)PK_G * s.cb_absmax * T_::to_float((uint16_t)xm);
        int e = 0;
        (void)frexpf(bound, &e);
        const bool finite = bound < __builtin_inff() && fabsf(v) <= 2.f * bound;
        const int sh = PK_FIX_BITS - e;
        const long long qv = finite ? __float2ll_rn(ldexpf(v, sh)) : 0ll;
        const unsigned long long mine = ((unsigned long long)qv << PK_VAL_SHIFT) + (finite ? 1ull : 1ull + (1ull << PK_CNT_BITS));
        const int row = row_begin + r;
        // the hand-shake goes out; it is looked at right away, or (defer, uniform) behind the next segment's loop.  The atomic
        // returns straight into the carried variable: a copy of its result would make hipcc wait for it here.
        pend_mine = mine;
        pend_sh = sh;
        pend_cell = s.acc + row;
        pend_y = s.y + row;
        pend_has_bias = s.bias != nullptr;
        pend_old = __hip_atomic_fetch_add(pend_cell, pend_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend_scale = s.scales[row];
        pend_bias = (s.bias ? s.bias : s.scales)[row];
        pend = true;
        if (!defer) settle_pending();
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
    if (self_dma) {  // my share of slice k + 1 has landed.  With a deferred hand-shake the wave's atomic and its scale / bias loads
      // are YOUNGER than the LDS-DMA (loads return in order): "at most 3 outstanding" then means the DMA is in, and the atomic
      // may stay in flight through the next loop.  A wave without rows issued none of the three and has to wait for everything.
      if (defer_k && (wave << 6) < nrows) __builtin_amdgcn_s_waitcnt(3 | (7 << 4) | (15 << 8));
      else __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
    }
    if constexpr (!ONEB) __builtin_amdgcn_s_barrier();    // F_k (single-barrier form: the other table set is written next)
  }
  settle_pending();
}

static bool pipe_eligible(const PackedLayout* Ls, int n, int in_groups, int& max_rg, int& nwc, int& dma_waves, uint32_t& xwin, bool& oneb) {
  max_rg = 0;
  nwc = 0;
  for (int k = 0; k < n; ++k) {
    if (Ls[k].EB != 4 || Ls[k].XC != 1) return false;
    max_rg = std::max(max_rg, Ls[k].RG);
    nwc = std::max(nwc, Ls[k].NW);
  }
  if (PK_SLICE_BYTES != 65536u || PK_NST != 256 || n < 2 || nwc > PK_MAX_NW || (uint32_t)(in_groups + 1) * PK_VB > PP_XWIN) return false;
  dma_waves = nwc + PP_DMA_WAVES <= PK_MAX_NW ? PP_DMA_WAVES : 0;
  if (tuning().packed_pipe == 2) dma_waves = 0;  // experiments: self-service DMA for every shape
  if (tuning().packed_pipe == 3 && dma_waves == 0) return false;  // experiments: round-3 first cut (DMA waves only)
  xwin = (uint32_t)(in_groups + 1) * PK_VB <= PP_XWIN_SMALL ? PP_XWIN_SMALL : PP_XWIN;
  oneb = dma_waves != 0 && tuning().packed_pipe != 5 && pipe_lds(max_rg, xwin, true).total <= 160u * 1024u;  // 5: two barriers always
  return pipe_lds(max_rg, xwin, oneb).total <= 160u * 1024u;
}

// ---------------------------------------------------------------------------------------------- host launch helpers
static int pick_pd(const PackedLayout& L) {
  const int t = tuning().packed_prefetch;
  if (t == 3 || t == 4 || t == 8) return t;
  // measured (profiles/r02_mb_packed_variants.log): 3 steps in flight per wave are best or within 1 % of best on every
  // shape; 8 are 5-10 % slower (the bigger burst of the prologue delays the codebook slice, which gates the loop)
  return 3;
}

// Instantiations: (dtype, B, PD, entry bytes).  Only the batch-1 kernels come with the deeper ring (PD = 8): with more
// rows the loop is LDS-bound and 4 steps in flight cover the stream.
// slice_first: the one-row image with the slice in front (no 64 KiB x window): for layers whose row tables do not fit
// behind the window (packed_b1_slice_first).
template <class KP, class Launch>
static int dispatch_packed(int dtype, int batch, int pd, int eb, bool slice_first, Launch&& launch) {
#define AQLM_PK_GO(TT, BB, PP, EE) launch(KP::template get<TT, BB, PP, EE, PK_XWIN_FULL>(), PackedLds<BB, PK_XWIN_FULL>{})
#define AQLM_PK_GO0(TT, PP, EE) launch(KP::template get<TT, 1, PP, EE, 0u>(), PackedLds<1, 0u>{})
#define AQLM_PK_CASE(BB)                                                                                      \
  case BB:                                                                                                    \
    if (dtype == AQLM_HIP_F16) return eb == 3 ? AQLM_PK_GO(F16, BB, 4, 3) : AQLM_PK_GO(F16, BB, 4, 4);          \
    return eb == 3 ? AQLM_PK_GO(BF16, BB, 4, 3) : AQLM_PK_GO(BF16, BB, 4, 4);
  switch (batch) {
    case 1:
#define AQLM_PK_B1(TT, EE) (pd == 8 ? AQLM_PK_GO(TT, 1, 8, EE) : (pd == 4 ? AQLM_PK_GO(TT, 1, 4, EE) : AQLM_PK_GO(TT, 1, 3, EE)))
      if (slice_first) {  // (ring depth 3, the default, only)
        if (dtype == AQLM_HIP_F16) return eb == 3 ? AQLM_PK_GO0(F16, 3, 3) : AQLM_PK_GO0(F16, 3, 4);
        return eb == 3 ? AQLM_PK_GO0(BF16, 3, 3) : AQLM_PK_GO0(BF16, 3, 4);
      }
      if (dtype == AQLM_HIP_F16) return eb == 3 ? AQLM_PK_B1(F16, 3) : AQLM_PK_B1(F16, 4);
      return eb == 3 ? AQLM_PK_B1(BF16, 3) : AQLM_PK_B1(BF16, 4);
#undef AQLM_PK_B1
    AQLM_PK_CASE(2)
    AQLM_PK_CASE(3)
    AQLM_PK_CASE(4)
    AQLM_PK_CASE(5)
    AQLM_PK_CASE(6)
    AQLM_PK_CASE(7)
    AQLM_PK_CASE(8)
  }
#undef AQLM_PK_CASE
#undef AQLM_PK_GO
#undef AQLM_PK_GO0
  return AQLM_HIP_E_INVALID;
}

struct SingleKernels {
  template <class T_, int B, int PD, int EB, uint32_t XW>
  static auto get() { return gemv_1x16_packed_kernel<T_, B, PD, XW, EB>; }
};
struct PublishKernels {  // row-parallel shards (aqlm_hip_gemv_1x16_packed_publish)
  template <class T_, int B, int PD, int EB, uint32_t XW>
  static auto get() { return gemv_1x16_packed_kernel<T_, B, PD, XW, EB, true>; }
};
struct MultiKernels {
  template <class T_, int B, int PD, int EB, uint32_t XW>
  static auto get() { return gemv_1x16_packed_multi_kernel<T_, B, PD, XW, EB>; }
};

// largest batch whose LDS image fits the CU
template <int BB>
static size_t packed_lds_total(int in_groups, int RG) { return PackedLds<BB, PK_XWIN_FULL>::total(in_groups, RG); }
// one row: x first (a 64 KiB window, both reads without an address add) where that fits, else slice first
static bool packed_b1_slice_first(int in_groups, int RG) { return packed_lds_total<1>(in_groups, RG) > 160 * 1024; }
static size_t packed_lds_need(int b, int in_groups, int RG) {
  switch (b) {
    case 1: return std::min(packed_lds_total<1>(in_groups, RG), PackedLds<1, 0u>::total(in_groups, RG));
    case 2: return packed_lds_total<2>(in_groups, RG);
    case 3: return packed_lds_total<3>(in_groups, RG);
    case 4: return packed_lds_total<4>(in_groups, RG);
    case 5: return packed_lds_total<5>(in_groups, RG);
    case 6: return packed_lds_total<6>(in_groups, RG);
    case 7: return packed_lds_total<7>(in_groups, RG);
    default: return packed_lds_total<8>(in_groups, RG);
  }
}
static int packed_max_batch(int in_groups, int RG) {
  // (the single-row image is not always the smallest: x first keeps a 64 KiB window in front of the slice)
  if (packed_lds_need(1, in_groups, RG) > 160 * 1024) return 0;
  int b = AQLM_HIP_MAX_GEMV_BATCH;
  while (b > 1 && packed_lds_need(b, in_groups, RG) > 160 * 1024) --b;
  return packed_lds_need(b, in_groups, RG) <= 160 * 1024 ? b : 0;
}

}  // namespace PK_NS
}  // namespace aqlm

using namespace aqlm;
using namespace aqlm::PK_NS;

#if AQLM_PK_G == 8
// the 16-element twin of this file (same signatures; hidden symbols of this library)
#pragma GCC visibility push(hidden)
extern "C" {
size_t aqlm_hip_g16_prepack_1x16_bytes(int, int, int);
int aqlm_hip_g16_prepack_1x16(const void*, int, int, int, void*, size_t, aqlm_hip_packed_desc*, void*);
int aqlm_hip_g16_prepack_1x16_ex(const void*, int, int, int, void*, size_t, aqlm_hip_packed_desc*, int, void*);
int aqlm_hip_g16_packed_set_codebook(aqlm_hip_packed_desc*, void*, const void*, void*);
int aqlm_hip_g16_packed_plan_relabel(const uint32_t*, int, uint16_t*);
int aqlm_hip_g16_packed_plan_relabel_ex(const uint32_t*, int, int, uint16_t*);
int aqlm_hip_g16_packed_plan_geometry(const uint64_t*, int, int, int, uint8_t*);
int aqlm_hip_g16_packed_desc_read(const void*, size_t, aqlm_hip_packed_desc*);
int aqlm_hip_g16_unpack_1x16(const aqlm_hip_packed_desc*, const void*, void*, void*);
int aqlm_hip_g16_gemv_1x16_packed_cells(const aqlm_hip_packed_desc*, const void*, const void*, const void*, const void*, const void*, void*, int,
                                        long, long, int, void*, size_t, void*);
int aqlm_hip_g16_gemv_1x16_packed(const aqlm_hip_packed_desc*, void*, const void*, const void*, const void*, const void*, void*, int, long, long,
                                  int, void*, size_t, void*);
int aqlm_hip_g16_gemv_1x16_packed_chain(const aqlm_hip_packed_desc*, void*, const void*, const void*, const void*, const void*, void*, int, long,
                                        long, int, void*, size_t, const aqlm_hip_packed_desc*, const void*, const void*, void*);
int aqlm_hip_g16_gemv_1x16_packed_partials(const aqlm_hip_packed_desc*, const void*, const void*, const void*, int, long, int, void*, size_t, void*);
int aqlm_hip_g16_gemv_1x16_packed_publish(const aqlm_hip_packed_desc*, void*, const void*, const void*, int, long, int, const aqlm_hip_xgmi*, void*,
                                          void*, void*);
int aqlm_hip_g16_gemv_1x16_packed_multi(const aqlm_hip_segment*, const aqlm_hip_packed_desc* const*, int, const void*, int, int, long, int, void*,
                                        size_t, void*);
int aqlm_hip_g16_gemv_1x16_packed_multi_cells(const aqlm_hip_segment*, const aqlm_hip_packed_desc* const*, int, const void*, int, int, long, int,
                                              void*, size_t, void*);
}
#pragma GCC visibility pop
// a descriptor of the twin's format (32 slices)
static inline bool pk_is_g16(const aqlm_hip_packed_desc* d) { return d && d->slices_log2 == 5; }
#define PK_G16_FORWARD(desc_expr, call) \
  if (pk_is_g16(desc_expr)) return call
#define PK_G16_FORWARD_IF(cond, call) \
  if (cond) return call
#else
#define PK_G16_FORWARD(desc_expr, call)
#define PK_G16_FORWARD_IF(cond, call)
#endif

// ---- planning steps of the repack (host, pure functions) -----------------------------------------------------------------
// Relabelling: deal the 65536 entries to the PK_S slices, heaviest first, each to the lightest slice that still has room
// (longest-processing-time greedy with PK_SLICE_ENTRIES entries per slice): the slices end up with equal code counts unless a
// single entry outweighs a slice's share.  Deterministic (ties: lower label, lower slice).  Within a slice the heaviest entries
// take consecutive slots, i.e. distinct LDS bank groups.  Returns 0 (and leaves new_of_old alone) when the checkpoint's labels
// already load the slices evenly: no permutation, no codebook image, the buffer of format v6.
// `force`: deal even when the slices' TOTAL masses are already even -- label use correlated with the ROW (rows of one block of the
// layer drawing their codes from one slice's labels) leaves every global histogram flat and one stream per row group 14 x the
// mean; the repack asks for it whenever the 16 x 16 layout is not balanced and keeps the result only if the
// longest stream got shorter.  Equal counts fall out round-robin over the slices (stable order, lightest slice first).
static int plan_relabel(const uint32_t* usage, uint16_t* new_of_old, bool force = false) {
  unsigned long long mass0[PK_S] = {}, total = 0;
  for (int c = 0; c < 65536; ++c) mass0[c >> PK_CODE_BITS] += usage[c];
  unsigned long long mx = 0;
  for (int s = 0; s < PK_S; ++s) {
    total += mass0[s];
    mx = std::max(mx, mass0[s]);
  }
  if (total == 0 || (!force && (double)mx * PK_S <= 1.02 * (double)total)) return 0;
  std::vector<uint32_t> order(65536);
  for (uint32_t c = 0; c < 65536; ++c) order[c] = c;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return usage[x] > usage[y]; });
  unsigned long long mass[PK_S] = {};
  int cnt[PK_S] = {};
  for (uint32_t i = 0; i < 65536; ++i) {
    const uint32_t c = order[i];
    int best = -1;
    for (int s = 0; s < PK_S; ++s)
      if (cnt[s] < PK_SLICE_ENTRIES && (best < 0 || mass[s] < mass[best])) best = s;
    new_of_old[c] = (uint16_t)(best * PK_SLICE_ENTRIES + cnt[best]);
    ++cnt[best];
    mass[best] += usage[c];
  }
  return 1;
}

// Geometry: deal the PK_NST workgroups to the slices in proportion to their work (lane-steps + a quarter step per row for the
// epilogue's hand-in): start from PK_MIN_GROUPS each, give the next workgroup to the slice whose groups are the longest.
// Uniform unless that shortens the longest stream by more than 6 % (the variable-geometry kernel pays ~0.1 us of scalar
// arithmetic in its prologue and serves single-layer launches only).  Returns 1 when groups[] is not uniform.
static int plan_geometry(const unsigned long long* slice_steps, int M, int in_features, uint8_t* groups) {
  for (int s = 0; s < PK_S; ++s) groups[s] = (uint8_t)PK_NG;
  if (PK_G != 8 || M < PK_VG_MIN_ROWS || !packed_shape_ok(M, in_features, PK_G)) return 0;
  const int in_groups = in_features / PK_G;
  double w[PK_S], uni = 0.;
  for (int s = 0; s < PK_S; ++s) {
    w[s] = (double)slice_steps[s] + 0.25 * M;
    uni = std::max(uni, w[s] / PK_NG);
  }
  const int mb_uniform = packed_max_batch(in_groups, (M + PK_NG - 1) / PK_NG);
  for (int nmin = PK_MIN_GROUPS; nmin < PK_NG; ++nmin) {
    int n[PK_S], left = PK_NST - nmin * PK_S, mn = PK_NG;
    for (int s = 0; s < PK_S; ++s) n[s] = nmin;
    while (left-- > 0) {
      int best = 0;
      for (int s = 1; s < PK_S; ++s)
        if (w[s] / n[s] > w[best] / n[best]) best = s;
      ++n[best];
    }
    double var = 0.;
    bool uniform = true;
    for (int s = 0; s < PK_S; ++s) {
      var = std::max(var, w[s] / n[s]);
      mn = std::min(mn, n[s]);
      uniform = uniform && n[s] == PK_NG;
      if (n[s] > 255) return 0;
    }
    if (uniform || var > 0.94 * uni) return 0;
    // the bigger row groups of the lightest slices must not cost the layer its multi-row launches (up to 4 rows in one launch; 5
    // and 6 rows may take two): else start from more groups per slice
    if (packed_max_batch(in_groups, (M + mn - 1) / mn) < std::min(mb_uniform, 4)) continue;
    for (int s = 0; s < PK_S; ++s) groups[s] = (uint8_t)n[s];
    return 1;
  }
  return 0;
}

extern "C" PK_API int aqlm_hip_packed_plan_relabel(const uint32_t* usage, int slices_log2, uint16_t* new_of_old) {
#if AQLM_PK_G == 8
  if (slices_log2 == 5) return aqlm_hip_g16_packed_plan_relabel(usage, slices_log2, new_of_old);
#endif
  if (!usage || !new_of_old || slices_log2 != PK_S_LOG) {
    set_last_error("aqlm_hip_packed_plan_relabel: null pointer or slices_log2 not 4 / 5");
    return AQLM_HIP_E_INVALID;
  }
  return plan_relabel(usage, new_of_old);
}

extern "C" PK_API int aqlm_hip_packed_plan_relabel_ex(const uint32_t* usage, int slices_log2, int force, uint16_t* new_of_old) {
#if AQLM_PK_G == 8
  if (slices_log2 == 5) return aqlm_hip_g16_packed_plan_relabel_ex(usage, slices_log2, force, new_of_old);
#endif
  if (!usage || !new_of_old || slices_log2 != PK_S_LOG) {
    set_last_error("aqlm_hip_packed_plan_relabel_ex: null pointer or slices_log2 not 4 / 5");
    return AQLM_HIP_E_INVALID;
  }
  return plan_relabel(usage, new_of_old, force != 0);
}

extern "C" PK_API int aqlm_hip_packed_plan_geometry(const uint64_t* slice_steps, int slices_log2, int out_features, int in_features,
                                                    uint8_t* slice_groups) {
#if AQLM_PK_G == 8
  if (slices_log2 == 5) return aqlm_hip_g16_packed_plan_geometry(slice_steps, slices_log2, out_features, in_features, slice_groups);
#endif
  if (!slice_steps || !slice_groups || slices_log2 != PK_S_LOG) {
    set_last_error("aqlm_hip_packed_plan_geometry: null pointer or slices_log2 not 4 / 5");
    return AQLM_HIP_E_INVALID;
  }
  unsigned long long st[PK_S];
  for (int s = 0; s < PK_S; ++s) st[s] = slice_steps[s];
  return plan_geometry(st, out_features, in_features, slice_groups);
}

// scratch of the repack, kept in the tail of the caller's capacity: usage counts, the relabelling table, lane-steps per
// (slice, row), lane-steps per slice
struct PkScratch {
  size_t off_hist, off_relabel, off_ls, off_steps, bytes;
};
static PkScratch pk_scratch(int M) {
  PkScratch sc;
  sc.off_hist = 0;
  sc.off_relabel = sc.off_hist + (size_t)65536 * 4;
  sc.off_ls = sc.off_relabel + (size_t)65536 * 2;
  sc.off_steps = align_up(sc.off_ls + (size_t)PK_S * M * 2, 256);
  sc.bytes = align_up(sc.off_steps + PK_S * 4, 1024);
  return sc;
}

extern "C" PK_API size_t aqlm_hip_prepack_1x16_bytes(int out_features, int in_features, int in_group_size) {
  PK_G16_FORWARD_IF(in_group_size == 16, aqlm_hip_g16_prepack_1x16_bytes(out_features, in_features, in_group_size));
  if (!packed_shape_ok(out_features, in_features, in_group_size)) return 0;
  // capacity for streams up to 1.5 x the balanced length (the repack balances the slices; what is left is the difference
  // between the rows of one slice), plus the permutation + codebook image of a relabelled buffer and the repack's scratch;
  // the bytes actually used come back in the descriptor and the buffer may be trimmed to them
  const size_t nst = (size_t)PK_NST;
  const size_t M = (size_t)out_features;
  const size_t RG2 = (M + PK_MIN_GROUPS - 1) / PK_MIN_GROUPS;                 // rows per group at most (variable geometry)
  const size_t in_groups = (size_t)in_features / PK_G;
  const size_t fair = (M * in_groups / 4 + (size_t)PK_S * M) / nst;            // lane-steps per stream when all are equal
  const size_t lane_steps = fair * 3 / 2 + RG2 + 64;                           // per stream
  const size_t ent = nst * (lane_steps * 16 + 16 * 1024);
  const size_t meta = 4096 + nst * PK_MAX_NW * 16 + nst * (RG2 + 1) * 4 + (size_t)AQLM_HIP_MAX_GEMV_BATCH * out_features * 8;
  const size_t image = (size_t)65536 * 2 + (size_t)65536 * PK_VB + 2048;
  // 3-byte entries: the repack builds the 4-byte layout in the tail of the buffer and squeezes it to the front
  return align_up(meta + ent + ent * PK_WREG3 / 1024 + image + pk_scratch(out_features).bytes + 4096, 1024);
}

extern "C" PK_API int aqlm_hip_prepack_1x16_ex(const void* codes, int out_features, int in_features, int in_group_size,
                                               void* packed, size_t packed_bytes, aqlm_hip_packed_desc* desc, int flags, void* stream_) {
  PK_G16_FORWARD_IF(in_group_size == 16, aqlm_hip_g16_prepack_1x16_ex(codes, out_features, in_features, in_group_size, packed, packed_bytes, desc, flags, stream_));
  hipStream_t stream = (hipStream_t)stream_;
  if (!codes || !packed || !desc) {
    set_last_error("aqlm_hip_prepack_1x16: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (!packed_shape_ok(out_features, in_features, in_group_size)) {
    set_last_error("aqlm_hip_prepack_1x16: unsupported shape (needs g=%d, in/g <= %d; got g=%d in=%d out=%d)", PK_G, PK_MAX_GROUPS,
                   in_group_size, in_features, out_features);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const size_t cap = aqlm_hip_prepack_1x16_bytes(out_features, in_features, in_group_size);
  if (packed_bytes < cap || !aligned16(packed)) {
    set_last_error("aqlm_hip_prepack_1x16: packed buffer needs %zu bytes (16-B aligned), got %zu", cap, packed_bytes);
    return AQLM_HIP_E_INVALID;
  }
  const int M = out_features, in_groups = in_features / PK_G;
  const size_t nst = (size_t)PK_NST;
  uint8_t* base = (uint8_t*)packed;
  const PkScratch sc = pk_scratch(M);
  const size_t work_end = (packed_bytes - sc.bytes) / 1024 * 1024;  // the layout (and the 4-byte working copy of 3-byte entries) stay below
  uint8_t* scratch = base + work_end;
  uint32_t* hist_d = (uint32_t*)(scratch + sc.off_hist);
  uint16_t* relabel_d = (uint16_t*)(scratch + sc.off_relabel);
  uint16_t* ls_d = (uint16_t*)(scratch + sc.off_ls);
  uint32_t* steps_d = (uint32_t*)(scratch + sc.off_steps);
  const int row_blocks = (M + 3) / 4;
  uint32_t* maxL_d = (uint32_t*)(base + 128);
  uint8_t groups[32] = {};
  for (int s = 0; s < PK_S; ++s) groups[s] = (uint8_t)PK_NG;
  std::vector<uint16_t> new_of_old, old_of_new;
  bool relabel = false;
  uint32_t steps_h[PK_S];
  uint32_t maxL = 0;
  // wave-steps a workgroup runs when its longest stream has L lane-steps (what the balancing can change at all: the kernel
  // runs whole steps of whole waves)
  auto capacity = [](uint32_t L) {
    const int nw = choose_waves(L);
    return (uint64_t)nw * ((L + 64u * nw - 1) / (64u * nw));
  };
  // counts lane-steps per (slice, row) with the given labels, builds the row starts of every stream for `groups` in place (their
  // offset does not depend on the wave count chosen later) and reads back the longest stream
  PackedLayout L0;
  auto count_and_scan = [&](const uint16_t* rl) -> int {
    if (!packed_layout(M, in_features, 1, 1, L0, 1, 4, groups, relabel)) {
      set_last_error("aqlm_hip_prepack_1x16: internal: geometry plan rejected");
      return AQLM_HIP_E_INVALID;
    }
    if (L0.off_ent > work_end) {
      set_last_error("aqlm_hip_prepack_1x16: capacity too small for the row tables");
      return AQLM_HIP_E_INVALID;
    }
    if (int e = check_hip(hipMemsetAsync(base, 0, L0.off_ent, stream), "prepack memset")) return e;
    if (int e = check_hip(hipMemsetAsync(steps_d, 0, PK_S * 4, stream), "prepack memset")) return e;
    hipLaunchKernelGGL(pk_count_kernel, dim3(row_blocks), dim3(256), 0, stream, (const uint16_t*)codes, rl, ls_d, steps_d, M, in_groups);
    hipLaunchKernelGGL(pk_scan_kernel, dim3((unsigned)nst), dim3(256), 0, stream, ls_d, (uint32_t*)(base + L0.off_rowstart), maxL_d, L0.G);
    if (int e = check_hip(hipMemcpyAsync(steps_h, steps_d, PK_S * 4, hipMemcpyDeviceToHost, stream), "prepack read-back")) return e;
    if (int e = check_hip(hipMemcpyAsync(&maxL, maxL_d, 4, hipMemcpyDeviceToHost, stream), "prepack read-back")) return e;
    return check_hip(hipStreamSynchronize(stream), "prepack sync");
  };
  // (1) the checkpoint's labels on the 16 x 16 geometry
  if (int e = count_and_scan(nullptr)) return e;
  // Balancing is for the longest stream: when it already runs as few wave-steps as perfectly even streams would (+3.5 %: the
  // noise between the row groups of uniform codes), labels and geometry stay -- no permutation, no codebook image.
  unsigned long long total_steps = 0;
  for (int s = 0; s < PK_S; ++s) total_steps += steps_h[s];
  const uint32_t even = (uint32_t)((total_steps * 1035ull + (unsigned long long)PK_NST * 1000ull - 1ull) / ((unsigned long long)PK_NST * 1000ull));
  const bool balanced = capacity(maxL) <= capacity(even);
  if (!balanced) {
    // (2) usage counts -> relabelling plan -> lane-steps with the new labels
    if (!(flags & AQLM_HIP_PREPACK_NO_RELABEL)) {
      std::vector<uint32_t> usage(65536);
      if (int e = check_hip(hipMemsetAsync(hist_d, 0, (size_t)65536 * 4, stream), "prepack memset")) return e;
      hipLaunchKernelGGL(pk_hist_kernel, dim3(2048), dim3(256), 0, stream, (const uint16_t*)codes, (size_t)M * in_groups, hist_d);
      if (int e = check_hip(hipMemcpyAsync(usage.data(), hist_d, (size_t)65536 * 4, hipMemcpyDeviceToHost, stream), "prepack read-back")) return e;
      if (int e = check_hip(hipStreamSynchronize(stream), "prepack sync")) return e;
      new_of_old.resize(65536);
      // forced: the layout is not balanced although the global masses may be (label use correlated with the row); kept only when
      // the longest stream runs fewer wave-steps with the new labels
      const uint32_t maxL_before = maxL;
      relabel = plan_relabel(usage.data(), new_of_old.data(), /*force=*/true) == 1;
      if (relabel) {
        old_of_new.resize(65536);
        for (uint32_t c = 0; c < 65536; ++c) old_of_new[new_of_old[c]] = (uint16_t)c;
        if (int e = check_hip(hipMemcpyAsync(relabel_d, new_of_old.data(), (size_t)65536 * 2, hipMemcpyHostToDevice, stream), "prepack relabel table")) return e;
        if (int e = count_and_scan(relabel_d)) return e;
        if (capacity(maxL) >= capacity(maxL_before)) {  // the deal did not help (e.g. one row that lives in one slice): labels as they are
          relabel = false;
          if (int e = count_and_scan(nullptr)) return e;
        }
      }
    }
    // (3) an entry that outweighs a slice: deal the workgroups to the slices by their work
    if (!(flags & AQLM_HIP_PREPACK_UNIFORM_ONLY)) {
      unsigned long long steps64[PK_S];
      for (int s = 0; s < PK_S; ++s) steps64[s] = steps_h[s];
      if (plan_geometry(steps64, M, in_features, groups))
        if (int e = count_and_scan(relabel ? relabel_d : nullptr)) return e;
    }
  }
  const uint16_t* rl = relabel ? relabel_d : nullptr;
  const PkGeom G = L0.G;
  uint32_t* a = (uint32_t*)(base + L0.off_rowstart);
  const int NW = choose_waves(maxL);
  const int T = (int)((maxL + 64u * NW - 1) / (64u * NW));
  const bool arrange = tuning().packed_arrange && T <= PK_ARR_MAX_T;
  // rotated copies of x: 1 by default -- with the row pools the x reads are already spread well, and up to 4 copies
  // measured within +-1 % (profiles/r02_mb_packed_variants.log); the knob keeps the mechanism testable
  int XC = 1;
  if (AQLM_PK_XFIRST && PK_G == 8 && arrange && !packed_b1_slice_first(in_groups, G.RG) && tuning().packed_xcopies >= 1 && tuning().packed_xcopies <= 4) XC = std::min(pk_max_x_copies(in_groups), tuning().packed_xcopies);
  // 32-bit entries by default (two operations instead of four to form an entry's addresses); 24-bit entries (-23 % bytes; wave
  // ranges of at most 32 steps) are the compact choice for inference-only deployments.  Re-measured per shape in round 5 (profiles/r05_entry_bytes_3_vs_4.md, A/B/A/B on one box): 24-bit
  // entries cost 5.8 % on 4096 x 4096, 3.3 % on 4096 -> 11008, 3.2 % on 4096 -> 1024, 2.3 % on 8192 x 8192 and 1.3-1.4 % on
  // 4096 <-> 14336; layers of more than 32 steps per wave (the 70B MLP) cannot take them at all.  They also keep a shared-input group
  // off the pipelined kernel (4-byte entries only).  So they stay the opt-in compact form (tuning key 3, `prepack_model(compact=True)`).
  const int EB = (PK_G == 8 && tuning().packed_entry_bytes == 3 && T <= 32 && !G.vg) ? 3 : 4;  // (the 3-byte form exists for 16-B vectors and uniform geometry only)
  PackedLayout L, L4;
  if (!packed_layout(M, in_features, NW, T, L, XC, EB, groups, relabel) || !packed_layout(M, in_features, NW, T, L4, XC, 4, groups, relabel) ||
      L.used + (EB == 3 ? L4.ent_bytes + 1024 : 0) > work_end) {
    set_last_error("aqlm_hip_prepack_1x16: the rows of this layer use the codebook too differently from one another for the packed "
                   "format (longest stream %u lane-steps after balancing, %d steps per wave)", maxL, T);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  uint32_t* winfo = (uint32_t*)(base + L.off_winfo);
  // the 4-byte working layout: in place for 4-byte entries, else in the tail of the capacity
  uint32_t* ent = EB == 4 ? (uint32_t*)(base + L.off_ent) : (uint32_t*)(base + (work_end - L4.ent_bytes) / 1024 * 1024);
  aqlm_hip_packed_desc d{};
  d.magic = PK_MAGIC;
  d.version = PK_VERSION;
  d.out_features = M;
  d.in_features = in_features;
  d.slices_log2 = PK_S_LOG;
  d.waves = NW;
  d.steps = T;
  d.entry_bytes = EB;
  d.used_bytes = L.used;
  d.x_copies = (uint32_t)XC;
  d.codebook_absmax = 0.f;  // unknown: the caller sets it (see include/aqlm_hip.h) to enable the fused finalize
  d.flags = (relabel ? AQLM_HIP_PACKED_RELABELLED : 0u) | (G.vg ? AQLM_HIP_PACKED_VARGEOM : 0u);
  d.rows_per_group = G.RG;
  for (int s = 0; s < PK_S; ++s) d.slice_groups[s] = groups[s];
  if (int e = check_hip(hipMemcpyAsync(base, &d, sizeof(d), hipMemcpyHostToDevice, stream), "prepack header")) return e;
  const uint32_t null_entry = (uint32_t)in_groups << (16 + PK_VSH);
  hipLaunchKernelGGL(pk_fill_kernel, dim3(2048), dim3(256), 0, stream, ent, L4.ent_bytes / 4, null_entry);
  hipLaunchKernelGGL(pk_scatter_kernel, dim3(row_blocks), dim3(256), 0, stream, (const uint16_t*)codes, rl, a, ent, G, in_groups, NW, T);
  if (arrange)
    hipLaunchKernelGGL(pk_arrange_kernel, dim3((unsigned)nst, NW), dim3(64), (size_t)T * 1024 + (size_t)T * 256, stream, a,
                       ent, G, in_groups, NW, T, XC);
  // the local search on the greedy deal (1-3 % faster matvec) takes ~10x the greedy deal's time (27 -> 120 ms for a 29 M-code
  // layer): by default (1) only layers of <= 8 Mi codes get it -- a 70B model then prepacks in ~13 s instead of ~1 min --,
  // 3 = always, 2 = never
  if (arrange && (tuning().packed_arrange == 3 || (tuning().packed_arrange == 1 && (long)M * in_groups <= PK_IMPROVE_MAX_CODES)))
    hipLaunchKernelGGL(pk_improve_kernel, dim3((unsigned)nst, NW), dim3(64), (size_t)T * (1024 + 512 + 256 + 32), stream, a, ent, G,
                       in_groups, NW, T);
  if (PK_PARITY_BITS) hipLaunchKernelGGL(pk_parity_kernel, dim3(2048), dim3(256), 0, stream, ent, L4.ent_bytes / 4);
  hipLaunchKernelGGL(pk_flag_kernel, dim3((G.RG + 255) / 256, (unsigned)nst), dim3(256), 0, stream, a, ent, G, NW, T);
  hipLaunchKernelGGL(pk_column_kernel, dim3((unsigned)nst, NW), dim3(64), 0, stream, a, ent, winfo, G, NW, T);
  if (EB == 3)
    hipLaunchKernelGGL(pk_compress_kernel, dim3((unsigned)nst, NW), dim3(64), 0, stream, ent, base + L.off_ent, winfo, G, NW, T);
  if (int e = check_hip(hipGetLastError(), "prepack launch")) return e;
  if (relabel) {  // the permutation travels with the buffer: unpack and the codebook image are made from it
    if (int e = check_hip(hipMemcpyAsync(base + L.off_perm, old_of_new.data(), (size_t)65536 * 2, hipMemcpyHostToDevice, stream), "prepack permutation")) return e;
  }
  if (int e = check_hip(hipStreamSynchronize(stream), "prepack sync")) return e;  // `d` and the tables live on the host stack / heap
  *desc = d;
  return 0;
}

extern "C" PK_API int aqlm_hip_prepack_1x16(const void* codes, int out_features, int in_features, int in_group_size,
                                     void* packed, size_t packed_bytes, aqlm_hip_packed_desc* desc, void* stream_) {
  return aqlm_hip_prepack_1x16_ex(codes, out_features, in_features, in_group_size, packed, packed_bytes, desc, 0, stream_);
}

// codebook image of a relabelled buffer: image[new] = codebook[old_of_new[new]], one 16-B piece per thread
namespace aqlm {
namespace PK_NS {
__global__ __launch_bounds__(256) void pk_codebook_image_kernel(const uint16_t* old_of_new, const u32x4* codebook, u32x4* image) {
  constexpr uint32_t PIECES = PK_VB / 16;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;  // < 65536 * PIECES
  const uint32_t c = i / PIECES, h = i - c * PIECES;
  image[i] = codebook[(uint32_t)old_of_new[c] * PIECES + h];
}
}  // namespace PK_NS
}  // namespace aqlm

extern "C" PK_API int aqlm_hip_packed_set_codebook(aqlm_hip_packed_desc* desc, void* packed, const void* codebook, void* stream_) {
  PK_G16_FORWARD(desc, aqlm_hip_g16_packed_set_codebook(desc, packed, codebook, stream_));
  PackedLayout L;
  if (!packed || !codebook || !desc_layout(desc, L) || !aligned16(packed) || !aligned16(codebook)) {
    set_last_error("aqlm_hip_packed_set_codebook: null / misaligned pointer or invalid descriptor");
    return AQLM_HIP_E_INVALID;
  }
  if (!L.relabel) return 0;  // the kernels read the caller's codebook
  uint8_t* base = (uint8_t*)packed;
  hipLaunchKernelGGL(pk_codebook_image_kernel, dim3(65536u * (PK_VB / 16) / 256u), dim3(256), 0, (hipStream_t)stream_,
                     (const uint16_t*)(base + L.off_perm), (const u32x4*)codebook, (u32x4*)(base + L.off_cb));
  if (int e = check_hip(hipGetLastError(), "codebook image launch")) return e;
  desc->flags |= AQLM_HIP_PACKED_HAS_CODEBOOK;
  return 0;
}

extern "C" PK_API int aqlm_hip_packed_desc_read(const void* header_host, size_t header_bytes, aqlm_hip_packed_desc* desc) {
#if AQLM_PK_G == 8
  if (header_host && desc && header_bytes >= sizeof(aqlm_hip_packed_desc)) {
    aqlm_hip_packed_desc h;
    memcpy(&h, header_host, sizeof(h));
    if (pk_is_g16(&h)) return aqlm_hip_g16_packed_desc_read(header_host, header_bytes, desc);
  }
#endif
  if (!header_host || !desc || header_bytes < sizeof(aqlm_hip_packed_desc)) {
    set_last_error("aqlm_hip_packed_desc_read: need the first %zu bytes of the packed buffer", sizeof(aqlm_hip_packed_desc));
    return AQLM_HIP_E_INVALID;
  }
  aqlm_hip_packed_desc d;
  memcpy(&d, header_host, sizeof(d));
  PackedLayout L;
  if (!desc_layout(&d, L)) {
    set_last_error("aqlm_hip_packed_desc_read: not a packed 1x16 buffer of format v6");
    return AQLM_HIP_E_INVALID;
  }
  *desc = d;
  return 0;
}

extern "C" PK_API int aqlm_hip_unpack_1x16(const aqlm_hip_packed_desc* desc, const void* packed, void* codes, void* stream_) {
  PK_G16_FORWARD(desc, aqlm_hip_g16_unpack_1x16(desc, packed, codes, stream_));
  hipStream_t stream = (hipStream_t)stream_;
  PackedLayout L;
  if (!packed || !codes || !desc_layout(desc, L)) {
    set_last_error("aqlm_hip_unpack_1x16: null pointer or invalid descriptor");
    return AQLM_HIP_E_INVALID;
  }
  const uint8_t* base = (const uint8_t*)packed;
  const uint16_t* perm = L.relabel ? (const uint16_t*)(base + L.off_perm) : nullptr;  // relabelled: back to the checkpoint's labels
  if (L.EB == 3)
    hipLaunchKernelGGL(pk_unpack3_kernel, dim3((unsigned)L.nst, L.NW), dim3(64), 0, stream, base + L.off_ent,
                       (const uint32_t*)(base + L.off_winfo), perm, (uint16_t*)codes, L.G, L.in_groups, L.NW, L.T);
  else
    hipLaunchKernelGGL(pk_unpack_kernel, dim3((unsigned)L.nst, L.NW), dim3(64), 0, stream, (const uint32_t*)(base + L.off_ent),
                       (const uint32_t*)(base + L.off_winfo), perm, (uint16_t*)codes, L.G, L.in_groups, L.NW, L.T);
  return check_hip(hipGetLastError(), "unpack launch");
}

// The fused finalize needs the layer's codebook range (descriptor field) and can be switched off for A/B runs.
static bool packed_fused(const aqlm_hip_packed_desc* desc) {
  return tuning().packed_fused_finalize != 0 && desc->codebook_absmax > 0.f && desc->codebook_absmax < __builtin_inff();
}

// main kernel of one launch (<= packed_max_batch rows).  With `fused.y` it also finalizes (accumulator cells inside the
// packed buffer, no workspace); without, it leaves fp32 slice partials [16][nb][M] in the workspace.
struct PackedFused {
  const void* scales = nullptr;
  const void* bias = nullptr;
  void* y = nullptr;
  long y_row_stride = 0;
  float cb_absmax = 0.f;
  void* cells = nullptr;  // caller-owned accumulator cells [rows][M] u64, zero at rest (one set per stream); null: the cells inside the packed buffer
  // row-parallel shard: publish the fp32 totals (aqlm_hip_gemv_1x16_packed_publish) instead of writing y
  float* pub = nullptr;
  uint32_t* pub_flag = nullptr;
  uint32_t* pub_epoch = nullptr;
  uint32_t pub_max_elems = 0;
};

// the layer that runs next on the stream (chain prefetch); all null = none
struct PackedNext {
  const uint8_t* ent = nullptr;
  const uint8_t* codebook = nullptr;
  uint32_t block_bytes = 0;
};

static int packed_launch_main(const PackedLayout& L, const void* packed, const void* codebook, const uint16_t* x, int nb,
                              long x_row_stride, int dtype, void* workspace, size_t workspace_bytes, hipStream_t stream,
                              const char* who, const PackedFused& fused = PackedFused{}, const PackedNext& next = PackedNext{}) {
  const size_t need = (fused.y || fused.pub) ? 0 : (size_t)PK_S * nb * L.M * sizeof(float);
  if (need && (!workspace || workspace_bytes < need)) {
    set_last_error("%s: workspace of %zu bytes required, got %zu", who, need, workspace_bytes);
    return AQLM_HIP_E_INVALID;
  }
  const uint8_t* base = (const uint8_t*)packed;
  PackedGemvParams p{};
  p.ent = (const uint32_t*)(base + L.off_ent);
  p.winfo = (const uint32_t*)(base + L.off_winfo);
  p.rowstart = (const uint32_t*)(base + L.off_rowstart);
  p.codebook = L.relabel ? base + L.off_cb : (const uint8_t*)codebook;  // relabelled: the permuted image (aqlm_hip_packed_set_codebook)
  p.x = x;
  p.partial = (float*)workspace;
  if (fused.y || fused.pub) {
    p.pub = fused.pub;
    p.pub_flag = fused.pub_flag;
    p.pub_epoch = fused.pub_epoch;
    p.pub_max_elems = fused.pub_max_elems;
    p.acc = fused.cells ? (unsigned long long*)fused.cells : (unsigned long long*)(const_cast<uint8_t*>(base) + L.off_acc);
    p.cb_absmax = fused.cb_absmax;
    p.scales = (const uint16_t*)fused.scales;
    p.bias = (const uint16_t*)fused.bias;
    p.y = (uint16_t*)fused.y;
    p.y_row_stride = fused.y_row_stride;
  }
  p.x_row_stride = x_row_stride;
  p.M = L.M;
  p.in_groups = L.in_groups;
  p.RG = L.RG;
  p.NW = L.NW;
  p.T = L.T;
  p.XC = L.XC;
  p.ent_bytes = (uint32_t)L.ent_bytes;
#ifdef AQLM_PACKED_TRACE
  p.trace = workspace && workspace_bytes >= need + (size_t)PK_NST * PK_MAX_NW * 8 * 8 ? (unsigned long long*)((uint8_t*)workspace + need) : nullptr;
  p.dbg = tuning().packed_debug;
#endif
  // chain prefetch: up to `packed_prefetch_waves` extra waves per workgroup (default 2) when a next layer is named
  int npw = 0;
  if (next.ent && next.codebook && next.block_bytes) {
    const int want = tuning().packed_prefetch_waves < 0 ? 0 : (tuning().packed_prefetch_waves == 0 ? 2 : tuning().packed_prefetch_waves);
    npw = std::min(std::min(want, 7), PK_MAX_NW - L.NW);
  }
  const uint32_t rotate = tuning().packed_fill_rotate ? 1u : 0u;
  auto launch = [&](auto kern, auto lds_map) -> int {
    const size_t lds = decltype(lds_map)::total(L.in_groups, L.RG, npw);
    if (lds > 160 * 1024) { npw = 0; }
    const size_t lds_final = decltype(lds_map)::total(L.in_groups, L.RG, npw);
    if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_final)) return e;
    PackedGemvRest rest{};
    rest.pub = p.pub;
    rest.pub_flag = p.pub_flag;
    rest.pub_epoch = p.pub_epoch;
    rest.pub_max_elems = p.pub_max_elems;
    if (npw) {
      rest.next_ent = next.ent;
      rest.next_codebook = next.codebook;
      rest.next_block_bytes = next.block_bytes;
    }
    rest.winfo = p.winfo;
    rest.partial = p.partial;
    rest.x_row_stride = p.x_row_stride;
    rest.acc = p.acc;
    rest.cb_absmax = p.cb_absmax;
    rest.scales = p.scales;
    rest.bias = p.bias;
    rest.y = p.y;
    rest.y_row_stride = p.y_row_stride;
#ifdef AQLM_PACKED_TRACE
    rest.trace = p.trace;
    rest.dbg = p.dbg;
#endif
    hipLaunchKernelGGL(kern, dim3(PK_NST), dim3((L.NW + npw) * 64), lds_final, stream, p.codebook, p.x, p.ent, p.rowstart, p.in_groups,
                       (uint32_t)p.NW | ((uint32_t)p.XC << 8) | ((uint32_t)npw << 12) | (rotate << 15) | ((uint32_t)p.T << 16), p.RG,
                       p.ent_bytes, p.M, rest);
    return check_hip(hipGetLastError(), "gemv_1x16_packed launch");
  };
  const bool sf = nb == 1 && packed_b1_slice_first(L.in_groups, L.RG);
#if AQLM_PK_G == 8
  if (L.G.vg) {  // variable geometry: its own kernels (4-byte entries, ring depth 3 / 4, no chain prefetch, no publish)
    if (p.pub != nullptr || L.EB != 4) {
      set_last_error("%s: a variable-geometry buffer runs on the single-layer matvec entries only (repack with "
                     "AQLM_HIP_PREPACK_UNIFORM_ONLY for the publish form)", who);
      return AQLM_HIP_E_UNSUPPORTED;
    }
    uint32_t ns[4] = {0u, 0u, 0u, 0u};
    for (int i = 0; i < PK_S; ++i) ns[i >> 2] |= (uint32_t)L.G.first[i] << ((i & 3) * 8);  // first stream of every slice (< 256)
    auto launch_vg = [&](auto kern, auto lds_map) -> int {
      const size_t lds = decltype(lds_map)::total(L.in_groups, L.RG, 0);
      if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
      PackedGemvRest rest{};
      rest.winfo = p.winfo;
      rest.partial = p.partial;
      rest.x_row_stride = p.x_row_stride;
      rest.acc = p.acc;
      rest.cb_absmax = p.cb_absmax;
      rest.scales = p.scales;
      rest.bias = p.bias;
      rest.y = p.y;
      rest.y_row_stride = p.y_row_stride;
#ifdef AQLM_PACKED_TRACE
      rest.trace = p.trace;
      rest.dbg = p.dbg;
#endif
      hipLaunchKernelGGL(kern, dim3(PK_NST), dim3(L.NW * 64), lds, stream, p.codebook, p.x, p.ent, (uint32_t)(L.off_ent - L.off_rowstart),
                         (uint32_t)p.in_groups | ((uint32_t)p.RG << 12), (uint32_t)p.NW | ((uint32_t)p.XC << 8) | (rotate << 15) | ((uint32_t)p.T << 16),
                         p.M, ns[0], ns[1], ns[2], ns[3], rest);
      return check_hip(hipGetLastError(), "gemv_1x16_packed (variable geometry) launch");
    };
#define AQLM_PK_VG(TT, BB, PP, XW) launch_vg(gemv_1x16_packed_vg_kernel<TT, BB, PP, XW>, PackedLds<BB, XW>{})
#define AQLM_PK_VG_CASE(BB) \
  case BB:                  \
    return dtype == AQLM_HIP_F16 ? AQLM_PK_VG(F16, BB, 4, PK_XWIN_FULL) : AQLM_PK_VG(BF16, BB, 4, PK_XWIN_FULL);
    switch (nb) {
      case 1:
        if (sf) return dtype == AQLM_HIP_F16 ? AQLM_PK_VG(F16, 1, 3, 0u) : AQLM_PK_VG(BF16, 1, 3, 0u);
        return dtype == AQLM_HIP_F16 ? AQLM_PK_VG(F16, 1, 3, PK_XWIN_FULL) : AQLM_PK_VG(BF16, 1, 3, PK_XWIN_FULL);
      AQLM_PK_VG_CASE(2)
      AQLM_PK_VG_CASE(3)
      AQLM_PK_VG_CASE(4)
      AQLM_PK_VG_CASE(5)
      AQLM_PK_VG_CASE(6)
      AQLM_PK_VG_CASE(7)
      AQLM_PK_VG_CASE(8)
    }
#undef AQLM_PK_VG_CASE
#undef AQLM_PK_VG
    return AQLM_HIP_E_INVALID;
  }
#endif
  if (p.pub != nullptr) return dispatch_packed<PublishKernels>(dtype, nb, pick_pd(L), L.EB, sf, launch);
  return dispatch_packed<SingleKernels>(dtype, nb, pick_pd(L), L.EB, sf, launch);
}

static int packed_check_args(const char* who, const aqlm_hip_packed_desc* desc, const void* packed, const void* codebook,
                             const void* x, int batch, long x_row_stride, int dtype, PackedLayout& L, int& max_b) {
  if (!packed || !codebook || !x || !desc) {
    set_last_error("%s: null pointer argument", who);
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("%s: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", who, dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (!desc_layout(desc, L)) {
    set_last_error("%s: invalid packed descriptor", who);
    return AQLM_HIP_E_INVALID;
  }
  if (batch < 1 || batch > AQLM_HIP_MAX_GEMV_BATCH || !aligned16(packed) || !aligned16(codebook) || !aligned16(x) ||
      (batch > 1 && x_row_stride % 8 != 0)) {
    set_last_error("%s: batch must be 1..%d and packed / codebook / x rows 16-B aligned (batch %d)", who,
                   AQLM_HIP_MAX_GEMV_BATCH, batch);
    return AQLM_HIP_E_INVALID;
  }
  if (L.relabel && !(desc->flags & AQLM_HIP_PACKED_HAS_CODEBOOK)) {
    set_last_error("%s: a relabelled buffer needs its codebook image: call aqlm_hip_packed_set_codebook first (and again whenever "
                   "the codebook changes)", who);
    return AQLM_HIP_E_INVALID;
  }
  max_b = packed_max_batch(L.in_groups, L.RG);
  if (max_b == 0) {
    set_last_error("%s: layer does not fit the LDS image (in_features %d)", who, desc->in_features);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  return 0;
}

static int gemv_1x16_packed_impl(const aqlm_hip_packed_desc* desc, void* packed, const void* codebook,
                                 const void* scales, const void* bias, const void* x, void* y, int batch,
                                 long x_row_stride, long y_row_stride, int dtype, void* workspace,
                                 size_t workspace_bytes, void* stream_, const PackedNext& next, void* cells = nullptr,
                                 size_t cells_bytes = 0);

extern "C" PK_API int aqlm_hip_gemv_1x16_packed_cells(const aqlm_hip_packed_desc* desc, const void* packed, const void* codebook,
                                               const void* scales, const void* bias, const void* x, void* y, int batch,
                                               long x_row_stride, long y_row_stride, int dtype, void* cells,
                                               size_t cells_bytes, void* stream_) {
  PK_G16_FORWARD(desc, aqlm_hip_g16_gemv_1x16_packed_cells(desc, packed, codebook, scales, bias, x, y, batch, x_row_stride, y_row_stride, dtype, cells, cells_bytes, stream_));
  if (!cells || !desc || !(desc->codebook_absmax > 0.f) || cells_bytes < (size_t)std::min(batch, AQLM_HIP_MAX_GEMV_BATCH) * desc->out_features * 8 ||
      (reinterpret_cast<uintptr_t>(cells) & 7u)) {
    set_last_error("aqlm_hip_gemv_1x16_packed_cells: needs a descriptor with the codebook range and %zu bytes of 8-B aligned, "
                   "zero-filled cells", desc ? (size_t)std::min(batch, AQLM_HIP_MAX_GEMV_BATCH) * desc->out_features * 8 : (size_t)0);
    return AQLM_HIP_E_INVALID;
  }
  if (!tuning().packed_fused_finalize) {
    set_last_error("aqlm_hip_gemv_1x16_packed_cells: the fused finalize is switched off (tuning knob packed_fused_finalize)");
    return AQLM_HIP_E_UNSUPPORTED;
  }
  return gemv_1x16_packed_impl(desc, const_cast<void*>(packed), codebook, scales, bias, x, y, batch, x_row_stride, y_row_stride, dtype,
                               nullptr, 0, stream_, PackedNext{}, cells, cells_bytes);
}

extern "C" PK_API int aqlm_hip_gemv_1x16_packed(const aqlm_hip_packed_desc* desc, void* packed, const void* codebook,
                                         const void* scales, const void* bias, const void* x, void* y, int batch,
                                         long x_row_stride, long y_row_stride, int dtype, void* workspace,
                                         size_t workspace_bytes, void* stream_) {
  PK_G16_FORWARD(desc, aqlm_hip_g16_gemv_1x16_packed(desc, packed, codebook, scales, bias, x, y, batch, x_row_stride, y_row_stride, dtype, workspace, workspace_bytes, stream_));
  return gemv_1x16_packed_impl(desc, packed, codebook, scales, bias, x, y, batch, x_row_stride, y_row_stride, dtype, workspace,
                               workspace_bytes, stream_, PackedNext{});
}

extern "C" PK_API int aqlm_hip_gemv_1x16_packed_chain(const aqlm_hip_packed_desc* desc, void* packed, const void* codebook,
                                               const void* scales, const void* bias, const void* x, void* y, int batch,
                                               long x_row_stride, long y_row_stride, int dtype, void* workspace,
                                               size_t workspace_bytes, const aqlm_hip_packed_desc* next_desc,
                                               const void* next_packed, const void* next_codebook, void* stream_) {
  PK_G16_FORWARD(desc, aqlm_hip_g16_gemv_1x16_packed_chain(desc, packed, codebook, scales, bias, x, y, batch, x_row_stride, y_row_stride, dtype, workspace, workspace_bytes, next_desc, next_packed, next_codebook, stream_));
  PackedNext next;
  PackedLayout LN;
  if (next_desc && next_packed && next_codebook) {
    if (!desc_layout(next_desc, LN) || !aligned16(next_packed) || !aligned16(next_codebook)) {
      set_last_error("aqlm_hip_gemv_1x16_packed_chain: invalid descriptor / misaligned buffer of the next layer");
      return AQLM_HIP_E_INVALID;
    }
    next.ent = (const uint8_t*)next_packed + LN.off_ent;
    next.codebook = LN.relabel ? (const uint8_t*)next_packed + LN.off_cb : (const uint8_t*)next_codebook;
    next.block_bytes = (uint32_t)(LN.ent_bytes / LN.nst);
    if (LN.G.vg) next = PackedNext{};  // the hint assumes workgroup b of both layers sits on one XCD with the same slice: uniform geometry only
  }
  return gemv_1x16_packed_impl(desc, packed, codebook, scales, bias, x, y, batch, x_row_stride, y_row_stride, dtype, workspace,
                               workspace_bytes, stream_, next);
}

static int gemv_1x16_packed_impl(const aqlm_hip_packed_desc* desc, void* packed, const void* codebook,
                                 const void* scales, const void* bias, const void* x, void* y, int batch,
                                 long x_row_stride, long y_row_stride, int dtype, void* workspace,
                                 size_t workspace_bytes, void* stream_, const PackedNext& next, void* cells, size_t cells_bytes) {
  hipStream_t stream = (hipStream_t)stream_;
  PackedLayout L;
  int max_b = 0;
  if (!scales || !y) {
    set_last_error("aqlm_hip_gemv_1x16_packed: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (int e = packed_check_args("aqlm_hip_gemv_1x16_packed", desc, packed, codebook, x, batch, x_row_stride, dtype, L, max_b)) return e;
  for (int b0 = 0; b0 < batch; b0 += max_b) {  // rows that do not fit one LDS image go in several launches
    const int nb = std::min(max_b, batch - b0);
    if (packed_fused(desc)) {
      PackedFused fz;
      fz.cb_absmax = desc->codebook_absmax;
      fz.scales = scales;
      fz.bias = bias;
      fz.y = (uint16_t*)y + (size_t)b0 * y_row_stride;
      fz.y_row_stride = y_row_stride;
      fz.cells = cells;  // launches of one call are stream-ordered: they share the cells
      if (int e = packed_launch_main(L, packed, codebook, (const uint16_t*)x + (size_t)b0 * x_row_stride, nb, x_row_stride, dtype,
                                     workspace, workspace_bytes, stream, "aqlm_hip_gemv_1x16_packed", fz,
                                     b0 + nb >= batch ? next : PackedNext{}))
        return e;
      continue;
    }
    if (int e = packed_launch_main(L, packed, codebook, (const uint16_t*)x + (size_t)b0 * x_row_stride, nb, x_row_stride, dtype,
                                   workspace, workspace_bytes, stream, "aqlm_hip_gemv_1x16_packed"))
      return e;
    PackedFinalizeParams f{};
    f.partial = (const float*)workspace;
    f.scales = (const uint16_t*)scales;
    f.bias = (const uint16_t*)bias;
    f.y = (uint16_t*)y + (size_t)b0 * y_row_stride;
    f.y_row_stride = y_row_stride;
    f.M = L.M;
    f.B = nb;
    if (dtype == AQLM_HIP_F16)
      hipLaunchKernelGGL(gemv_1x16_packed_finalize<F16>, dim3((L.M + 255) / 256), dim3(256), 0, stream, f.partial, f.scales, f.bias,
                         f.y, f.y_row_stride, f.M, f.B);
    else
      hipLaunchKernelGGL(gemv_1x16_packed_finalize<BF16>, dim3((L.M + 255) / 256), dim3(256), 0, stream, f.partial, f.scales, f.bias,
                         f.y, f.y_row_stride, f.M, f.B);
    if (int e = check_hip(hipGetLastError(), "gemv_1x16_packed_finalize launch")) return e;
  }
  return 0;
}

extern "C" PK_API int aqlm_hip_gemv_1x16_packed_partials(const aqlm_hip_packed_desc* desc, const void* packed, const void* codebook,
                                                  const void* x, int batch, long x_row_stride, int dtype, void* workspace,
                                                  size_t workspace_bytes, void* stream_) {
  PK_G16_FORWARD(desc, aqlm_hip_g16_gemv_1x16_packed_partials(desc, packed, codebook, x, batch, x_row_stride, dtype, workspace, workspace_bytes, stream_));
  PackedLayout L;
  int max_b = 0;
  if (int e = packed_check_args("aqlm_hip_gemv_1x16_packed_partials", desc, packed, codebook, x, batch, x_row_stride, dtype, L, max_b))
    return e;
  if (batch > max_b) {
    set_last_error("aqlm_hip_gemv_1x16_packed_partials: %d rows do not fit one LDS image (at most %d for in_features %d)", batch,
                   max_b, desc->in_features);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  return packed_launch_main(L, packed, codebook, (const uint16_t*)x, batch, x_row_stride, dtype, workspace, workspace_bytes,
                            (hipStream_t)stream_, "aqlm_hip_gemv_1x16_packed_partials");
}

static int gemv_1x16_packed_multi_impl(const aqlm_hip_segment* segments, const aqlm_hip_packed_desc* const* descs,
                                       int num_segments, const void* x, int in_features, int batch, long x_row_stride,
                                       int dtype, void* workspace, size_t workspace_bytes, void* cells, size_t cells_bytes,
                                       void* stream_);

extern "C" PK_API int aqlm_hip_gemv_1x16_packed_publish(const aqlm_hip_packed_desc* desc, void* packed, const void* codebook,
                                                 const void* x, int batch, long x_row_stride, int dtype,
                                                 const aqlm_hip_xgmi* xg, void* pub_own, void* flag_own, void* stream_) {
  PK_G16_FORWARD(desc, aqlm_hip_g16_gemv_1x16_packed_publish(desc, packed, codebook, x, batch, x_row_stride, dtype, xg, pub_own, flag_own, stream_));
  PackedLayout L;
  int max_b = 0;
  if (int e = packed_check_args("aqlm_hip_gemv_1x16_packed_publish", desc, packed, codebook, x, batch, x_row_stride, dtype, L, max_b))
    return e;
  if (!xg || !xg->epoch || !pub_own || !flag_own || !packed_fused(desc) || batch > max_b ||
      (size_t)batch * desc->out_features > (size_t)xg->max_elems) {
    set_last_error("aqlm_hip_gemv_1x16_packed_publish: needs the one-shot all-reduce state, a descriptor with the codebook range, "
                   "and batch (%d) rows that fit one launch (<= %d) and the state (%d elements)", batch, max_b, xg ? xg->max_elems : 0);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  PackedFused fz;
  fz.cb_absmax = desc->codebook_absmax;
  fz.pub = (float*)pub_own;
  fz.pub_flag = (uint32_t*)flag_own;
  fz.pub_epoch = (uint32_t*)xg->epoch;
  fz.pub_max_elems = (uint32_t)xg->max_elems;
  return packed_launch_main(L, packed, codebook, (const uint16_t*)x, batch, x_row_stride, dtype, nullptr, 0, (hipStream_t)stream_,
                            "aqlm_hip_gemv_1x16_packed_publish", fz);
}

extern "C" PK_API int aqlm_hip_gemv_1x16_packed_multi(const aqlm_hip_segment* segments, const aqlm_hip_packed_desc* const* descs,
                                               int num_segments, const void* x, int in_features, int batch,
                                               long x_row_stride, int dtype, void* workspace, size_t workspace_bytes,
                                               void* stream_) {
  PK_G16_FORWARD(descs && num_segments >= 1 ? descs[0] : nullptr, aqlm_hip_g16_gemv_1x16_packed_multi(segments, descs, num_segments, x, in_features, batch, x_row_stride, dtype, workspace, workspace_bytes, stream_));
  return gemv_1x16_packed_multi_impl(segments, descs, num_segments, x, in_features, batch, x_row_stride, dtype, workspace,
                                     workspace_bytes, nullptr, 0, stream_);
}

extern "C" PK_API int aqlm_hip_gemv_1x16_packed_multi_cells(const aqlm_hip_segment* segments, const aqlm_hip_packed_desc* const* descs,
                                                     int num_segments, const void* x, int in_features, int batch,
                                                     long x_row_stride, int dtype, void* cells, size_t cells_bytes,
                                                     void* stream_) {
  PK_G16_FORWARD(descs && num_segments >= 1 ? descs[0] : nullptr, aqlm_hip_g16_gemv_1x16_packed_multi_cells(segments, descs, num_segments, x, in_features, batch, x_row_stride, dtype, cells, cells_bytes, stream_));
  if (!cells || (reinterpret_cast<uintptr_t>(cells) & 7u) || !tuning().packed_fused_finalize) {
    set_last_error("aqlm_hip_gemv_1x16_packed_multi_cells: needs 8-B aligned, zero-filled cells and the fused finalize switched on");
    return AQLM_HIP_E_INVALID;
  }
  return gemv_1x16_packed_multi_impl(segments, descs, num_segments, x, in_features, batch, x_row_stride, dtype, nullptr, 0,
                                     cells, cells_bytes, stream_);
}

static int gemv_1x16_packed_multi_impl(const aqlm_hip_segment* segments, const aqlm_hip_packed_desc* const* descs,
                                       int num_segments, const void* x, int in_features, int batch, long x_row_stride,
                                       int dtype, void* workspace, size_t workspace_bytes, void* cells, size_t cells_bytes,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!segments || !descs || num_segments < 1 || num_segments > AQLM_HIP_MAX_SEGMENTS || !x) {
    set_last_error("aqlm_hip_gemv_1x16_packed_multi: 1..%d segments, their descriptors and a non-null x required (got %d)",
                   AQLM_HIP_MAX_SEGMENTS, num_segments);
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_gemv_1x16_packed_multi: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)",
                   dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (batch < 1 || batch > AQLM_HIP_MAX_GEMV_BATCH || !aligned16(x) || (batch > 1 && x_row_stride % 8 != 0)) {
    set_last_error("aqlm_hip_gemv_1x16_packed_multi: batch must be 1..%d with 16-B aligned rows (batch %d)",
                   AQLM_HIP_MAX_GEMV_BATCH, batch);
    return AQLM_HIP_E_INVALID;
  }
  {  // a variable-geometry segment (format v7) runs on the single-layer kernels only: one launch per segment, same bits
    bool any_vg = false;
    for (int k = 0; k < num_segments; ++k) {
      PackedLayout Lk;
      any_vg = any_vg || (descs[k] && desc_layout(descs[k], Lk) && Lk.G.vg);
    }
    if (any_vg) {
      size_t coff = 0;
      for (int k = 0; k < num_segments; ++k) {
        const aqlm_hip_segment& sg = segments[k];
        if (!descs[k] || descs[k]->in_features != in_features || descs[k]->out_features != sg.out_features) {
          set_last_error("aqlm_hip_gemv_1x16_packed_multi: segment %d: descriptor does not match in_features %d / out_features %d", k,
                         in_features, sg.out_features);
          return AQLM_HIP_E_INVALID;
        }
        const size_t cneed = (size_t)batch * sg.out_features * 8;
        if (cells && coff + cneed > cells_bytes) {
          set_last_error("aqlm_hip_gemv_1x16_packed_multi_cells: %zu bytes of cells required, got %zu", coff + cneed, cells_bytes);
          return AQLM_HIP_E_INVALID;
        }
        if (int e = gemv_1x16_packed_impl(descs[k], const_cast<void*>(sg.codes), sg.codebook, sg.scales, sg.bias, x, sg.y, batch, x_row_stride,
                                          sg.y_row_stride, dtype, workspace, workspace_bytes, stream_, PackedNext{},
                                          cells ? (uint8_t*)cells + coff : nullptr, cells ? cneed : 0))
          return e;
        coff += cneed;
      }
      return 0;
    }
  }
  PackedMultiParams mp{};
  PackedFinalizeMultiParams fm{};
  mp.x = (const uint16_t*)x;
  mp.x_row_stride = x_row_stride;
  mp.nseg = fm.nseg = num_segments;
  size_t need = 0;
  int fblocks = 0, max_rg = 0, nw = 0, pd = 4, eb = 0;
  bool fused = true;  // every segment must know its codebook range
  for (int k = 0; k < num_segments; ++k) fused = fused && descs[k] && packed_fused(descs[k]);
  size_t cells_need = 0;
  if (cells) {  // caller-owned accumulator cells: [segment][batch][M] u64
    if (!fused) {
      set_last_error("aqlm_hip_gemv_1x16_packed_multi_cells: every descriptor must carry the codebook range");
      return AQLM_HIP_E_INVALID;
    }
    for (int k = 0; k < num_segments; ++k) cells_need += (size_t)batch * segments[k].out_features * 8;
    if (cells_bytes < cells_need) {
      set_last_error("aqlm_hip_gemv_1x16_packed_multi_cells: %zu bytes of cells required, got %zu", cells_need, cells_bytes);
      return AQLM_HIP_E_INVALID;
    }
  }
  size_t cells_off = 0;
  for (int k = 0; k < num_segments; ++k) {
    const aqlm_hip_segment& sg = segments[k];
    if (!sg.codes || !sg.codebook || !sg.scales || !sg.y) {
      set_last_error("aqlm_hip_gemv_1x16_packed_multi: null pointer in segment %d", k);
      return AQLM_HIP_E_INVALID;
    }
    PackedLayout L;
    if (!desc_layout(descs[k], L) || descs[k]->in_features != in_features || descs[k]->out_features != sg.out_features ||
        !aligned16(sg.codes) || !aligned16(sg.codebook)) {
      set_last_error("aqlm_hip_gemv_1x16_packed_multi: segment %d: invalid descriptor, or it does not match in_features "
                     "%d / out_features %d", k, in_features, sg.out_features);
      return AQLM_HIP_E_INVALID;
    }
    const uint8_t* base = (const uint8_t*)sg.codes;
    PackedSegment& ps = mp.seg[k];
    ps.ent = (const uint32_t*)(base + L.off_ent);
    ps.winfo = (const uint32_t*)(base + L.off_winfo);
    ps.rowstart = (const uint32_t*)(base + L.off_rowstart);
    if (L.relabel && !(descs[k]->flags & AQLM_HIP_PACKED_HAS_CODEBOOK)) {
      set_last_error("aqlm_hip_gemv_1x16_packed_multi: segment %d: a relabelled buffer needs its codebook image "
                     "(aqlm_hip_packed_set_codebook)", k);
      return AQLM_HIP_E_INVALID;
    }
    ps.codebook = L.relabel ? base + L.off_cb : (const uint8_t*)sg.codebook;
    ps.partial = fused ? nullptr : (float*)((uint8_t*)workspace + need);
    if (fused) {
      ps.acc = cells ? (unsigned long long*)((uint8_t*)cells + cells_off) : (unsigned long long*)(const_cast<uint8_t*>(base) + L.off_acc);
      cells_off += (size_t)batch * sg.out_features * 8;
      ps.scales = (const uint16_t*)sg.scales;
      ps.bias = (const uint16_t*)sg.bias;
      ps.y = (uint16_t*)sg.y;
      ps.y_row_stride = sg.y_row_stride;
      ps.cb_absmax = descs[k]->codebook_absmax;
    }
    ps.M = L.M;
    ps.RG = L.RG;
    ps.NW = L.NW;
    ps.T = L.T;
    ps.XC = L.XC;
    ps.ent_bytes = (uint32_t)L.ent_bytes;
    mp.in_groups = L.in_groups;
    if (eb && eb != L.EB) {
      set_last_error("aqlm_hip_gemv_1x16_packed_multi: segments mix 3- and 4-byte entry formats");
      return AQLM_HIP_E_UNSUPPORTED;
    }
    eb = L.EB;
    max_rg = std::max(max_rg, L.RG);
    nw = std::max(nw, L.NW);
    pd = std::max(pd, pick_pd(L));
    PackedFinalizeSegment& fs = fm.seg[k];
    fs.f.partial = ps.partial;
    fs.f.scales = (const uint16_t*)sg.scales;
    fs.f.bias = (const uint16_t*)sg.bias;
    fs.f.y = (uint16_t*)sg.y;
    fs.f.y_row_stride = sg.y_row_stride;
    fs.f.M = sg.out_features;
    fs.f.B = batch;
    fs.block_begin = fblocks;
    fblocks += (sg.out_features + 255) / 256;
    need += (size_t)PK_S * batch * sg.out_features * sizeof(float);
  }
  if (!fused && (!workspace || workspace_bytes < need)) {
    set_last_error("aqlm_hip_gemv_1x16_packed_multi: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return AQLM_HIP_E_INVALID;
  }
  if (packed_max_batch(mp.in_groups, max_rg) < batch) {
    set_last_error("aqlm_hip_gemv_1x16_packed_multi: %d rows of %d features do not fit the LDS image", batch, in_features);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  // batch 1 with the single-kernel finalize: one workgroup per CU walks all segments, slices double-buffered (pipe kernel)
  if (fused && batch == 1 && tuning().packed_pipe) {
    PackedLayout Ls[AQLM_HIP_MAX_SEGMENTS];
    for (int k = 0; k < num_segments; ++k) desc_layout(descs[k], Ls[k]);
    int prg = 0, nwc = 0, dmaw = 0;
    uint32_t xwin = PP_XWIN;
    bool oneb = false;
    if (pipe_eligible(Ls, num_segments, mp.in_groups, prg, nwc, dmaw, xwin, oneb)) {
      PipeParams pp{};
      pp.x = mp.x;
      pp.in_groups = mp.in_groups;
      pp.nseg = num_segments;
      pp.max_rg = prg;
      pp.nwc = nwc;
      pp.dma_waves = dmaw;
      pp.defer = tuning().packed_pipe == 4 ? 0 : 1;
      for (int k = 0; k < num_segments; ++k) pp.seg[k] = mp.seg[k];
      const size_t lds = pipe_lds(prg, xwin, oneb).total;
      auto go = [&](auto kern) -> int {
        if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
        hipLaunchKernelGGL(kern, dim3(PK_NST), dim3((nwc + dmaw) * 64), lds, stream, pp);
        return check_hip(hipGetLastError(), "gemv_1x16_packed_pipe launch");
      };
#define AQLM_PP_GO(SELF, XWV, OB) \
  (dtype == AQLM_HIP_F16 ? go(gemv_1x16_packed_pipe_kernel<F16, SELF, XWV, OB>) : go(gemv_1x16_packed_pipe_kernel<BF16, SELF, XWV, OB>))
      if (xwin == PP_XWIN_SMALL) {
        if (dmaw == 0) return AQLM_PP_GO(true, PP_XWIN_SMALL, false);
        return oneb ? AQLM_PP_GO(false, PP_XWIN_SMALL, true) : AQLM_PP_GO(false, PP_XWIN_SMALL, false);
      }
      if (dmaw == 0) return AQLM_PP_GO(true, PP_XWIN, false);
      return oneb ? AQLM_PP_GO(false, PP_XWIN, true) : AQLM_PP_GO(false, PP_XWIN, false);
#undef AQLM_PP_GO
    }
  }
  auto launch = [&](auto kern, auto lds_map) -> int {
    const size_t lds = decltype(lds_map)::total(mp.in_groups, max_rg);
    if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
    hipLaunchKernelGGL(kern, dim3(PK_NST * num_segments), dim3(nw * 64), lds, stream, mp);
    return check_hip(hipGetLastError(), "gemv_1x16_packed_multi launch");
  };
  if (int e = dispatch_packed<MultiKernels>(dtype, batch, pd, eb, batch == 1 && packed_b1_slice_first(mp.in_groups, max_rg), launch)) return e;
  if (fused) return 0;
  if (dtype == AQLM_HIP_F16)
    hipLaunchKernelGGL(gemv_1x16_packed_finalize_multi<F16>, dim3(fblocks), dim3(256), 0, stream, fm);
  else
    hipLaunchKernelGGL(gemv_1x16_packed_finalize_multi<BF16>, dim3(fblocks), dim3(256), 0, stream, fm);
  return check_hip(hipGetLastError(), "gemv_1x16_packed_finalize_multi launch");
}
