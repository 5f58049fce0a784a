// 1x16 g8 matvec on slice-bucketed ("prepacked") codes, gfx950.
//
// Why a load-time repack: on MI355X a random 16-B codebook gather that hits L2 costs a whole 128-B line of the CU's
// L1-fill path (0.43 lane-gathers/clk/CU measured, profiles/r01_call1_mb_l2gather.log), which pins the direct kernel
// (gemv.hip) at ~5.5 % of the HBM roofline.  LDS gathers are >10x cheaper but only 160 KiB fit per CU, so the
// codebook has to be cut into 8 slices of 8192 entries with one slice per CU -- and then every CU must find "its"
// codes.  Scanning the canonical [out][in/8] code matrix for them (gemv_lds.hip) spends ~95 vector instructions per
// 512 codes to use 64.  Bucketing the codes by slice ONCE, when the layer is loaded, removes the scan.  (The
// reference also re-lays codes out at load time for its CPU kernel, inference.py:78-83.)
//
// Packed format v3 (built by aqlm_hip_prepack_1x16, checked bit-for-bit against a numpy model in tests/):
//   rows are split into NG = 32 row-groups of RG rows; codes into S = 8 slices by (code >> 13);
//   stream (g, s) = for each row of group g, in order: that row's codes of slice s (in the bank-aware order of
//   prepack_arrange_kernel below), each as ONE 32-bit entry  (8192 + j) << 16 | (code & 0x1fff): the two 16-bit halves, shifted left by 4, ARE the
//   LDS byte addresses of the codebook vector (slice at LDS 0..128 KiB) and of x[j] (x at LDS 128 KiB + 16 j), so an
//   entry costs two v_lshlrev_b32_sdwa instead of seven ALU ops of bit fiddling (the 24-bit two-plane format v2 did);
//   every (row, slice) bucket is padded to a multiple of 4 entries with null entries (j = in_groups, whose x is a
//   zero vector in LDS; code 0), so a lane fetches 4 consecutive entries with one aligned 16-B load;
//   inside a stream the rows are ordered by bucket size, largest first (stable): the four rows a wave processes in one
//   step then have similar sizes, so a wave whose rows all fit the first 64-entry pass skips the second pass entirely
//   (with the natural row order 94 % of the wave-steps of a 4096-wide layer execute a second, nearly empty pass);
//   rowperm[(g*S + s)*RG + p] = row (within the group) whose bucket sits at position p,
//   rowoff[(g*S + s)*(RG+1) + p] = global index of the first entry of the bucket at position p; slot RG closes the
//   stream.  ~4.1 bytes per code + 4 bytes per (row, slice): 2.1x the canonical 2 bytes per code.  The extra bytes are
//   free: the kernel runs at < 2 TB/s of HBM traffic, it is bound by LDS / ALU issue and latency, not by the stream.
//
// Kernel: grid = 256 workgroups = 32 groups x 8 slices (slice = bid % 8 = the XCD the block is observed to land on, so
// each XCD's L2 holds one slice; for speed only).  Workgroup (g, s): slice s of the codebook and x go to LDS; each quarter-wave (16 lanes) owns one
// row of the group at a time; a lane takes 4 consecutive entries of that row's bucket per step (the typical 64-entry
// bucket is one step): per entry 2 ds_read_b128 (codebook entry, x[j]) + 4 v_dot2c.  Entry loads run PD rows ahead.  fp32 partials [slice][row] -> workspace -> finalize kernel
// (adds the 8 slices, scale + bias, one rounding).  Every lane does useful work (no scan, no 8x re-read of codes).
#include <algorithm>

#include "aqlm_common.h"

namespace aqlm {

constexpr int PK_S = 8;        // slices
constexpr int PK_NG = 32;      // row groups  (PK_S * PK_NG == 256 workgroups == CUs)
constexpr int PK_SLICE_ENTRIES = 8192;
constexpr int PK_PAD = 128;    // entries of slack behind the stream (prefetch may run past the end)
constexpr int PK_MAX_RG = 4096;  // rows per row-group the load-time sort handles (out_features <= 131072)
constexpr uint32_t PK_XBASE = 8192;  // x[j] lives at LDS slot 8192 + j (16-B slots), right behind the codebook slice

struct PackedLayout {
  int M, in_groups, RG;
  size_t n_rowoff;   // NG * S * (RG + 1)
  size_t entries;    // capacity: M * in_groups real entries + up to 3 null entries per (row, slice)
  size_t n_perm;     // NG * S * RG
  size_t off_rowoff, off_perm, off_ent, total;
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static bool packed_layout(int out_features, int in_features, int g, PackedLayout& L) {
  if (g != 8 || out_features <= 0 || in_features <= 0 || in_features % 64 != 0 || in_features > 16320) return false;
  L.M = out_features;
  L.in_groups = in_features / 8;
  L.RG = ((out_features + PK_NG - 1) / PK_NG + 3) / 4 * 4;
  L.n_rowoff = (size_t)PK_NG * PK_S * (L.RG + 1);
  L.entries = (size_t)out_features * L.in_groups + (size_t)3 * PK_S * out_features;
  if ((L.entries + PK_PAD) * 4 >= ((size_t)1 << 32)) return false;  // 32-bit buffer offsets
  if (L.RG > PK_MAX_RG) return false;
  L.n_perm = (size_t)PK_NG * PK_S * L.RG;
  L.off_rowoff = 256;  // header
  L.off_perm = align_up(L.off_rowoff + L.n_rowoff * 4, 256);
  L.off_ent = align_up(L.off_perm + L.n_perm * 2, 256);
  L.total = align_up(L.off_ent + (L.entries + PK_PAD) * 4, 256);
  return true;
}

// ------------------------------------------------------------------------------------------------ prepack
// K1: per (row, slice) counts -> rowoff[] (as counts).  One wave per row.
__global__ __launch_bounds__(256) void prepack_count_kernel(const uint16_t* codes, uint32_t* rowoff, int M, int in_groups,
                                                            int RG) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  uint32_t cnt[PK_S];
#pragma unroll
  for (int s = 0; s < PK_S; ++s) cnt[s] = 0;
  for (int j = lane; j < in_groups; j += 64) {
    const uint32_t sl = codes[(size_t)row * in_groups + j] >> 13;
#pragma unroll
    for (int s = 0; s < PK_S; ++s) cnt[s] += (sl == (uint32_t)s);
  }
  const int g = row / RG, r = row - g * RG;
#pragma unroll
  for (int s = 0; s < PK_S; ++s) {
    uint32_t v = cnt[s];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    if (lane == 0) rowoff[((size_t)g * PK_S + s) * (RG + 1) + r] = (v + 3u) & ~3u;  // padded to 4 entries
  }
}

// K2: in-place exclusive prefix sum over the flat rowoff array (single block; load-time code, not a hot path).
__global__ __launch_bounds__(1024) void prepack_scan_kernel(uint32_t* rowoff, size_t n) {
  __shared__ uint32_t sums[1024];
  const int t = threadIdx.x;
  const size_t chunk = (n + 1023) / 1024;
  const size_t lo = std::min(n, (size_t)t * chunk), hi = std::min(n, lo + chunk);
  uint32_t s = 0;
  for (size_t i = lo; i < hi; ++i) s += rowoff[i];
  sums[t] = s;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 1024; ++i) {
      const uint32_t v = sums[i];
      sums[i] = run;
      run += v;
    }
  }
  __syncthreads();
  uint32_t run = sums[t];
  for (size_t i = lo; i < hi; ++i) {
    const uint32_t v = rowoff[i];
    rowoff[i] = run;
    run += v;
  }
}

// K1b: per stream (g, s): rank the rows by bucket size (descending, ties by row index) and put the counts in that
// order.  rank[] (row -> position) is kept in the rowperm array until the scatter / arrange passes are done; K5 inverts it.
__global__ __launch_bounds__(1024) void prepack_sort_kernel(uint32_t* rowoff, uint16_t* rank, int RG) {
  __shared__ uint32_t c[PK_MAX_RG];
  uint32_t* cnt = rowoff + (size_t)blockIdx.x * (RG + 1);
  uint16_t* rk = rank + (size_t)blockIdx.x * RG;
  for (int r = threadIdx.x; r < RG; r += 1024) c[r] = cnt[r];
  __syncthreads();
  uint32_t mine[PK_MAX_RG / 1024], pos[PK_MAX_RG / 1024];
#pragma unroll
  for (int i = 0; i < PK_MAX_RG / 1024; ++i) {
    const int r = threadIdx.x + i * 1024;
    mine[i] = r < RG ? c[r] : 0u;
    uint32_t k = 0;
    if (r < RG)
      for (int r2 = 0; r2 < RG; ++r2) k += (c[r2] > mine[i]) || (c[r2] == mine[i] && r2 < r);
    pos[i] = k;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PK_MAX_RG / 1024; ++i) {
    const int r = threadIdx.x + i * 1024;
    if (r < RG) {
      cnt[pos[i]] = mine[i];
      rk[r] = (uint16_t)pos[i];
    }
  }
}

// K4b: order of a lane's four entries over the four levels, chosen per PAIR of neighbouring buckets so that the
// codebook reads collide as little as possible.  The 16 lanes one ds_read_b128 pass services are l16 in {0-3,12-15} of
// the even position of a pair plus l16 in {4-11} of the odd one ("group A"), and the complementary halves ("group B");
// the x slots of those lanes are distinct by construction (K4), the codebook slots (code mod 16) are random: ~2.9
// passes per read, and the gather loop is LDS-bound (traced: 0.55 us per 64-row step = the LDS cycles of 8 reads per
// lane).  Greedy, deterministic: lanes are visited in order (even bucket, then odd bucket, lane by lane); each picks the
// first of the 24 orders of its entries that adds the fewest collisions to the levels of its group.  Moving an entry to
// another level of the SAME lane keeps K4's x property.  Only the two prefetched passes (entries 0-127) are treated.
__device__ __forceinline__ bool pk_lane_in_x_half(int l16) { return l16 < 4 || l16 >= 12; }

__global__ __launch_bounds__(64) void prepack_level_kernel(const uint32_t* rowoff, uint32_t* ent, int RG) {
  const int stream = blockIdx.y;
  const int pair = blockIdx.x * 64 + threadIdx.x;
  if (pair >= RG / 2) return;  // RG is a multiple of 4: every position has a partner
  const uint32_t* ro = rowoff + (size_t)stream * (RG + 1) + 2 * pair;
  const uint32_t beg[2] = {ro[0], ro[1]};
  const uint32_t len[2] = {ro[1] - ro[0], ro[2] - ro[1]};
  // the 24 orders of four items, lexicographic: entry perm[k] of the lane goes to level k
  const unsigned char P[24][4] = {{0,1,2,3},{0,1,3,2},{0,2,1,3},{0,2,3,1},{0,3,1,2},{0,3,2,1},{1,0,2,3},{1,0,3,2},
                                  {1,2,0,3},{1,2,3,0},{1,3,0,2},{1,3,2,0},{2,0,1,3},{2,0,3,1},{2,1,0,3},{2,1,3,0},
                                  {2,3,0,1},{2,3,1,0},{3,0,1,2},{3,0,2,1},{3,1,0,2},{3,1,2,0},{3,2,0,1},{3,2,1,0}};
  for (uint32_t region = 0; region < 128; region += 64) {
    uint32_t used[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};  // [group A / B][level]: bit r = codebook residue r taken
    for (int l = 0; l < 16; ++l) {
      for (int b = 0; b < 2; ++b) {
        const uint32_t off = region + 4u * (uint32_t)l;
        if (off + 4u > len[b]) continue;  // this lane has no chunk in this bucket
        u32x4* slot = reinterpret_cast<u32x4*>(ent + beg[b] + off);
        const u32x4 e = *slot;
        const uint32_t ev[4] = {e.x, e.y, e.z, e.w};
        const int grp = (pk_lane_in_x_half(l) == (b == 0)) ? 0 : 1;
        int best = 0, best_cost = 5;
        for (int q = 0; q < 24; ++q) {
          int cost = 0;
          for (int k = 0; k < 4; ++k) cost += (int)((used[grp][k] >> (ev[P[q][k]] & 15u)) & 1u);
          if (cost < best_cost) { best_cost = cost; best = q; }
        }
        u32x4 o;
        o.x = ev[P[best][0]]; o.y = ev[P[best][1]]; o.z = ev[P[best][2]]; o.w = ev[P[best][3]];
        *slot = o;
        used[grp][0] |= 1u << (o.x & 15u);
        used[grp][1] |= 1u << (o.y & 15u);
        used[grp][2] |= 1u << (o.z & 15u);
        used[grp][3] |= 1u << (o.w & 15u);
      }
    }
  }
}

// K5: rank (row -> position) -> rowperm (position -> row), in place.
__global__ __launch_bounds__(1024) void prepack_invert_kernel(uint16_t* rank, int RG) {
  __shared__ uint16_t t[PK_MAX_RG];
  uint16_t* rk = rank + (size_t)blockIdx.x * RG;
  for (int r = threadIdx.x; r < RG; r += 1024) t[r] = rk[r];
  __syncthreads();
  for (int r = threadIdx.x; r < RG; r += 1024) rk[t[r]] = (uint16_t)r;
}

// K3: scatter the entries.  One wave per row; ascending j within each (row, slice) bucket.
__global__ __launch_bounds__(256) void prepack_scatter_kernel(const uint16_t* codes, const uint32_t* rowoff,
                                                              const uint16_t* rank, uint32_t* ent, int M, int in_groups,
                                                              int RG) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int g = row / RG, r = row - g * RG;
  uint32_t base[PK_S];
#pragma unroll
  for (int s = 0; s < PK_S; ++s)
    base[s] = rowoff[((size_t)g * PK_S + s) * (RG + 1) + rank[((size_t)g * PK_S + s) * RG + r]];
  for (int j0 = 0; j0 < in_groups; j0 += 64) {
    const int j = j0 + lane;
    const bool ok = j < in_groups;
    const uint32_t code = ok ? codes[(size_t)row * in_groups + j] : 0u;
    const uint32_t sl = ok ? (code >> 13) : 0xffffffffu;
#pragma unroll
    for (int s = 0; s < PK_S; ++s) {
      const bool mine = sl == (uint32_t)s;
      const unsigned long long mask = __ballot(mine);
      const uint32_t before = __popcll(mask & ((1ull << lane) - 1ull));
      if (mine) ent[base[s] + before] = ((PK_XBASE + (uint32_t)j) << 16) | (code & 0x1fffu);
      base[s] += __popcll(mask);
    }
  }
  // pad every bucket to a multiple of 4 entries with null entries: j = in_groups (x[in_groups] is zero in LDS)
#pragma unroll
  for (int s = 0; s < PK_S; ++s) {
    const uint32_t pad = (0u - base[s]) & 3u;  // bucket starts are multiples of 4
    if ((uint32_t)lane < pad) ent[base[s] + lane] = (PK_XBASE + (uint32_t)in_groups) << 16;
  }
}

// K4: bank-aware order of the entries inside every (row, slice) bucket.  In the gemv kernel lane l16 of a quarter-wave
// reads entries 4*l16 + k (k = 0..3: "level" k) of its row's bucket, and the 16 lanes serviced together by one
// ds_read_b128 pass are l16 in {0-3, 12-15} of one row plus l16 in {4-11} of its neighbour row (service groups of a
// wave64 b128 read: lanes {0-3,12-15,20-27} / {4-11,16-19,28-31} / +32).  With ascending-j order the x[j] reads of a
// level hit random 16-B slots (~3-way bank conflicts; traced: the gather loop is LDS-bound).  Here every entry whose
// x slot has residue rho = j mod 16 is sent to its "home" lane -- residues 0-7 to l16 {0-3, 12-15}, residues 8-15 to
// l16 {4-11} -- at the next free level, so that the 16 lanes of a service group read 16 DIFFERENT slot residues
// whatever the neighbour row is.  Entries that find their home lane full (or absent in a short bucket) fill the
// remaining holes in index order.  The sum over a bucket is order-independent up to fp32 rounding; the order is fixed
// by this kernel (and mirrored by the numpy model in tests/), so results stay deterministic.
constexpr int PK_MAX_GROUPS = 2040;
constexpr uint32_t PK_EMPTY = 0xffffffffu;

__device__ __forceinline__ int pk_home_lane(uint32_t rho) { return rho < 4 ? (int)rho : (rho < 8 ? (int)rho + 8 : (int)rho - 4); }

__global__ __launch_bounds__(128) void prepack_arrange_kernel(const uint32_t* rowoff, const uint16_t* rank, uint32_t* ent, int M,
                                                              int in_groups, int RG) {
  __shared__ uint32_t in_s[2][PK_MAX_GROUPS + 32];
  __shared__ uint32_t out_s[2][PK_MAX_GROUPS + 32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 2 + w;
  if (row >= M) return;  // whole wave exits together; no block barrier is used below
  const int g = row / RG, r = row - g * RG;
  uint32_t start[PK_S], len[PK_S], off[PK_S];
  uint32_t run = 0;
#pragma unroll
  for (int s = 0; s < PK_S; ++s) {
    const size_t k = ((size_t)g * PK_S + s) * (RG + 1) + rank[((size_t)g * PK_S + s) * RG + r];
    start[s] = rowoff[k];
    len[s] = rowoff[k + 1] - start[s];
    off[s] = run;
    run += len[s];
  }
#pragma unroll
  for (int s = 0; s < PK_S; ++s)
    for (uint32_t i = lane; i < len[s]; i += 64) {
      in_s[w][off[s] + i] = ent[start[s] + i];
      out_s[w][off[s] + i] = PK_EMPTY;
    }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  if (lane < PK_S) {  // lane s arranges bucket s (serial: load-time code, a few microseconds per layer in total)
    uint32_t st = 0, ln = 0;
#pragma unroll
    for (int s = 0; s < PK_S; ++s)
      if (lane == s) { st = off[s]; ln = len[s]; }
    uint32_t* in = &in_s[w][st];
    uint32_t* out = &out_s[w][st];
    const uint32_t null_entry = (PK_XBASE + (uint32_t)in_groups) << 16;
    uint32_t n = ln;  // real entries come first, the (< 4) null entries of the scatter pass last
    while (n > 0 && in[n - 1] == null_entry) --n;
    const int m = ln / 4 < 16 ? (int)(ln / 4) : 16;
    unsigned long long cnt = 0;  // 16 x 3-bit level counters
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t e = in[i];
      const int L = pk_home_lane((e >> 16) & 15u);
      const uint32_t c = (uint32_t)(cnt >> (3 * L)) & 7u;
      if (L < m && c < 4) {
        out[4 * L + c] = e;
        cnt += 1ull << (3 * L);
        in[i] = PK_EMPTY;
      }
    }
    uint32_t idx = 0;
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t e = in[i];
      if (e == PK_EMPTY) continue;
      while (out[idx] != PK_EMPTY) ++idx;
      out[idx++] = e;
    }
    for (uint32_t i = 0; i < ln; ++i)
      if (out[i] == PK_EMPTY) out[i] = null_entry;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
  for (int s = 0; s < PK_S; ++s)
    for (uint32_t i = lane; i < len[s]; i += 64) ent[start[s] + i] = out_s[w][off[s] + i];
}

// ------------------------------------------------------------------------------------------------ gemv
struct PackedGemvParams {
  const uint32_t* rowoff;
  const uint16_t* rowperm;
  const uint32_t* ent;
  const uint8_t* codebook;
  const uint16_t* x;
  float* partial;  // [S][M]
  int M, in_groups, RG;
  uint32_t ent_bytes;
#ifdef AQLM_PACKED_TRACE
  unsigned long long* trace;  // [256 workgroups][8] wall-clock stamps (100 MHz), profiling builds only
#endif
};

// LDS access by absolute byte address.  The kernel has no static LDS, so its dynamic LDS starts at address 0 (checked
// at kernel entry): the codebook slice occupies [0, 128 KiB) and x slot j sits at (8192 + j) * 16.
typedef __attribute__((address_space(3))) const u32x4* lds_u32x4_ptr;

template <int WORD>
__device__ __forceinline__ uint32_t half_shl4(uint32_t w, uint32_t four) {
  uint32_t d;  // d = ((w >> 16*WORD) & 0xffff) << 4 in one instruction (sub-dword operand select)
  if constexpr (WORD == 0)
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(d) : "v"(four), "v"(w));
  else
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d) : "v"(four), "v"(w));
  return d;
}

// one 32-bit entry -> fp32 contribution
template <class T>
__device__ __forceinline__ float packed_entry(uint32_t w, uint32_t four, float acc) {
  const u32x4 e = *(lds_u32x4_ptr)(size_t)half_shl4<0>(w, four);   // codebook vector of (code & 0x1fff)
  const u32x4 xv = *(lds_u32x4_ptr)(size_t)half_shl4<1>(w, four);  // x[j]
  return dot8<T>(e, xv, acc);
}

// `block` in [0, 256): the workgroup's index within its own layer (== blockIdx.x for a single-layer launch).
// LPR = lanes per row: 16 (a quarter-wave per row: layers whose (row, slice) buckets hold ~64 entries and more), 8 or 4
// (16 rows per wave step: narrow layers,  [32 / 64 lanes per row for wide layers were measured slower: 14336->4096
// 16.3 vs 15.6 us, 8192->28672 39 vs 35 us -- more steps outweigh the avoided third-pass loop] e.g. the 1024-wide shards of a row-parallel 70B layer, whose buckets hold ~16
// entries -- with 16 lanes per row three quarters of the lanes would idle and the loop would need 4x the steps).
template <class T, int NWAVES, int PD, int LPR = 16>
__device__ __forceinline__ void gemv_1x16_packed_body(const PackedGemvParams& p, const int block) {
  constexpr int NT = NWAVES * 64;
  constexpr int RPW = 64 / LPR;          // rows per wave and step
  constexpr int STRIDE = NWAVES * RPW;   // rows between two consecutive rows of one lane group
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* const cbl = reinterpret_cast<u32x4*>(smem_raw);   // [8192] codebook slice
  u32x4* const xl = cbl + PK_SLICE_ENTRIES;                  // [in_groups + 1] x, 16 B per input group, then zeros

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & (LPR - 1), quarter = lane / LPR;  // lane within its row's group, group within the wave
  // slice = blockIdx % 8: blocks are observed to land on XCD blockIdx % 8, so each XCD's L2 serves ONE 128 KiB slice
  // (fetched once) instead of the whole codebook (speed only; any placement is correct).  Workgroups of one row-group
  // share nothing -- each walks its own bucket stream -- so they need not be co-located.
#ifdef AQLM_PACKED_TRACE
  unsigned long long tr[7];
  tr[0] = wall_clock64();
#define AQLM_TRACE(i) tr[i] = wall_clock64()
#else
#define AQLM_TRACE(i)
#endif
  const int slice = block & 7;
  const int group = block >> 3;
  const int row_begin = group * p.RG;
  int nrows = p.M - row_begin;
  nrows = nrows < 0 ? 0 : (nrows < p.RG ? nrows : p.RG);
  const uint32_t* const ro = p.rowoff + ((size_t)group * PK_S + slice) * (p.RG + 1);
  const uint16_t* const pm = p.rowperm + ((size_t)group * PK_S + slice) * p.RG;  // position -> row of the group

  __amdgpu_buffer_rsrc_t rs_ent = __builtin_amdgcn_make_buffer_rsrc((void*)p.ent, 0, p.ent_bytes, 0x00020000);
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw != 0u) __builtin_trap();  // see lds_u32x4_ptr

  // Software pipeline over this quarter-wave's rows r0, r0+STRIDE, ...: bucket bounds run 2*PD rows ahead (ring of
  // 2*PD slots), the first two 4-entry chunks of each lane PD rows ahead (ring of PD slots).  Rings are indexed with
  // compile-time slots only (main loop unrolled 2*PD times) and never copied, so no in-flight register is touched
  // before its row is consumed.  Everything below is issued before the LDS fill (HBM latency overlaps it).
  constexpr int NB2 = 2 * PD;
  const int r0 = wave * RPW + quarter;
  uint32_t bst[NB2], ben[NB2];
  uint16_t brow[NB2];
  u32x4 e_q[PD], e_q2[PD];              // chunk l16 and chunk l16 + 16 (buckets of 65..128 entries) of each row
  // Every load below is UNCONDITIONAL (rows past the end are clamped to the closing rowoff slot = an empty bucket;
  // chunks past a bucket's end read neighbouring entries or, past the planes, zeros from the bounds-checked buffer
  // descriptor, and are never consumed).  Loads inside divergent branches make hipcc's s_waitcnt bookkeeping fall
  // back to vmcnt(0) at every use, which serialises the whole prefetch pipeline (measured: 0.85 us per step).
  auto bounds = [&](int r, uint32_t& st, uint32_t& en, uint16_t& row) {  // r is a POSITION in the sorted stream
    const int a = r < p.RG ? r : p.RG, b = r + 1 < p.RG ? r + 1 : p.RG;
    st = ro[a];
    en = ro[b];
    row = pm[r < p.RG ? r : p.RG - 1];
  };
  auto fetch = [&](uint32_t st, int chunk, u32x4& e) {
    e = __builtin_amdgcn_raw_buffer_load_b128(rs_ent, (st + 4u * (uint32_t)chunk) * 4u, 0, 0);
  };
  // Prologue, in dependency order.  Loads return in issue order (one vmcnt counter), so the bucket bounds go FIRST:
  // the entry fetches that depend on them can then be issued while the 128 KiB codebook slice and x are still in
  // flight, instead of the slice being requested only after the rowoff round trip (traced: 1.9 us from kernel entry to
  // "all loads issued" before this ordering).
#pragma unroll
  for (int k = 0; k < NB2; ++k) bounds(r0 + k * STRIDE, bst[k], ben[k], brow[k]);
  constexpr int PER = PK_SLICE_ENTRIES / NT;
  static_assert(PK_SLICE_ENTRIES % NT == 0, "slice must split evenly over the workgroup");
  static_assert(2 * NT >= 2040, "x is staged in one pass of two pieces per thread (in_features <= 16320)");
  u32x4 stage[PER], xv[2];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.codebook) + (size_t)slice * PK_SLICE_ENTRIES;
    // the 32 workgroups of an XCD copy the same slice at the same time: each starts at a different 16 KiB piece so that
    // they do not all hit the same L2 lines (one channel) in lock step
#pragma unroll
    for (int k = 0; k < PER; ++k) stage[k] = src[tid + ((k + group) % PER) * NT];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + k * NT < p.in_groups ? tid + k * NT : p.in_groups - 1;
      xv[k] = *reinterpret_cast<const u32x4*>(p.x + (size_t)q * 8);
    }
  }
#pragma unroll
  for (int k = 0; k < PD; ++k) {  // needs the bounds only: vmcnt leaves the slice / x loads in flight
    fetch(bst[k], l16, e_q[k]);
    fetch(bst[k], l16 + LPR, e_q2[k]);
  }
  AQLM_TRACE(1);  // every load of the prologue has been issued
  // unconditional LDS writes (threads past the end write a dump slot): with the store under a branch hipcc sinks the
  // x load into the branch and guards it with vmcnt(0), i.e. the whole LDS fill then waits for the entry prefetch
  // (traced: workgroup barrier 2 us after the first wave had its data)
#pragma unroll
  for (int k = 0; k < 2; ++k) xl[tid + k * NT < p.in_groups ? tid + k * NT : p.in_groups + 1] = xv[k];
  if (tid == 0) xl[p.in_groups] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < PER; ++k) cbl[tid + ((k + group) % PER) * NT] = stage[k];
#ifdef AQLM_PACKED_TRACE
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes (hence its slice data) are done
  AQLM_TRACE(6);
#endif
  __syncthreads();
  AQLM_TRACE(2);  // LDS filled

  uint32_t four = 4u;
  asm volatile("" : "+v"(four));  // the SDWA shift count must sit in a VGPR
  auto consume = [&](const u32x4& e, float acc) -> float {
    acc = packed_entry<T>(e.x, four, acc);
    acc = packed_entry<T>(e.y, four, acc);
    acc = packed_entry<T>(e.z, four, acc);
    acc = packed_entry<T>(e.w, four, acc);
    return acc;
  };

  int r = r0;
  while (__any(r < nrows)) {
#pragma unroll
    for (int s6 = 0; s6 < NB2; ++s6) {  // no early exit: a single back-edge keeps every in-flight load in place
      const int es = s6 % PD;
      const uint32_t st = bst[s6], en = ben[s6];
      const uint32_t out_row = brow[s6];
      const u32x4 e1 = e_q[es], e2 = e_q2[es];
      // refill: entries of row r + PD*STRIDE (its bounds sit PD slots further in the ring), bounds of row r + 2*PD*STRIDE
      fetch(bst[(s6 + PD) % NB2], l16, e_q[es]);
      fetch(bst[(s6 + PD) % NB2], l16 + LPR, e_q2[es]);
      bounds(r + NB2 * STRIDE, bst[s6], ben[s6], brow[s6]);

      const int nchunks = (int)((en - st) >> 2);
      float acc = 0.f;
      if (l16 < nchunks) acc = consume(e1, acc);
      if (l16 + LPR < nchunks) acc = consume(e2, acc);
      for (int c = l16 + 2 * LPR; __any(c < nchunks); c += LPR) {  // buckets longer than 128 entries (rare): blocking loads
        u32x4 e3;
        fetch(st, c, e3);
        if (c < nchunks) acc = consume(e3, acc);
      }
      // reduction over the row's lanes on the VALU (DPP / readlane): no LDS traffic
      if constexpr (LPR == 16) acc = row16_sum(acc);
      else if constexpr (LPR == 8) acc = oct_sum(acc);
      else acc = quad_sum(acc);
      if (l16 == 0 && r < nrows) p.partial[(size_t)slice * p.M + row_begin + out_row] = acc;  // positions < nrows are the valid rows
#ifdef AQLM_PACKED_TRACE
      if (r == r0) AQLM_TRACE(3);  // first row done: the rowoff -> entries chain has arrived
      if (p.trace && tid == 0 && r >= r0 && s6 < 6 && r < r0 + 6 * STRIDE) p.trace[256 * 8 + (size_t)block * 8 + s6] = wall_clock64();
#endif
      r += STRIDE;
    }
  }
  AQLM_TRACE(4);
#ifdef AQLM_PACKED_TRACE
  __syncthreads();
  unsigned long long* wdone = reinterpret_cast<unsigned long long*>(smem_raw);  // the codebook slice is dead by now
  if (lane == 0) { wdone[wave] = tr[4]; wdone[NWAVES + wave] = tr[0]; wdone[2 * NWAVES + wave] = tr[6]; }
  __syncthreads();
  tr[5] = wall_clock64();
  if (tid == 0 && p.trace) {
    for (int i = 0; i < 6; ++i) p.trace[(size_t)block * 8 + i] = tr[i];
    unsigned long long mx = 0, me = 0, ms = 0;
    for (int w = 0; w < NWAVES; ++w) {
      mx = wdone[w] > mx ? wdone[w] : mx;
      me = wdone[NWAVES + w] > me ? wdone[NWAVES + w] : me;
      ms = wdone[2 * NWAVES + w] > ms ? wdone[2 * NWAVES + w] : ms;
    }
    p.trace[(size_t)block * 8 + 5] = me;  // last wave of the workgroup to START
    p.trace[(size_t)block * 8 + 6] = ms;  // last wave to have its share of the slice in LDS
    p.trace[(size_t)block * 8 + 7] = mx;  // last wave to leave the loop
  }
#endif
  // (An in-kernel finalize -- last-arriving slice workgroup of a row-group adds the eight partials -- was measured:
  // with __threadfence() it costs +80 us (the agent-scope buffer_inv throws away the L2 lines every other workgroup of
  // the XCD is streaming through); with write-through sc1 stores / sc1 loads and no fence it is correct but exactly as
  // slow as the separate finalize launch, +-0.3 us on every shape.  Hence the plain two-kernel form.)
}

template <class T, int NWAVES, int PD, int LPR = 16>
__global__ __launch_bounds__(NWAVES * 64) void gemv_1x16_packed_kernel(const PackedGemvParams p) {
  gemv_1x16_packed_body<T, NWAVES, PD, LPR>(p, blockIdx.x);
}

// Several prepacked layers that multiply the same x (gate/up) in one launch of 256 workgroups per layer; the second
// layer's workgroups start as CUs free up, so the first layer's tail and the second's LDS fill overlap.
struct PackedSegment {
  const uint32_t* rowoff;
  const uint16_t* rowperm;
  const uint32_t* ent;
  const uint8_t* codebook;
  float* partial;
  int M, RG;
  uint32_t ent_bytes;
};

struct PackedMultiParams {
  const uint16_t* x;
  int in_groups, nseg;
  PackedSegment seg[AQLM_HIP_MAX_SEGMENTS];
};

template <class T, int NWAVES, int PD, int LPR = 16>
__global__ __launch_bounds__(NWAVES * 64) void gemv_1x16_packed_multi_kernel(const PackedMultiParams mp) {
  const int sidx = (int)blockIdx.x >> 8;
  PackedGemvParams p{};
  p.x = mp.x;
  p.in_groups = mp.in_groups;
#pragma unroll
  for (int k = 0; k < AQLM_HIP_MAX_SEGMENTS; ++k) {
    if (k == 0 || sidx == k) {  // scalar select chain (no dynamic indexing of the kernel-argument struct)
      p.rowoff = mp.seg[k].rowoff;
      p.rowperm = mp.seg[k].rowperm;
      p.ent = mp.seg[k].ent;
      p.codebook = mp.seg[k].codebook;
      p.partial = mp.seg[k].partial;
      p.M = mp.seg[k].M;
      p.RG = mp.seg[k].RG;
      p.ent_bytes = mp.seg[k].ent_bytes;
    }
  }
  gemv_1x16_packed_body<T, NWAVES, PD, LPR>(p, (int)blockIdx.x & 255);
}

struct PackedFinalizeParams {
  const float* partial;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
  int M;
};

template <class T>
__global__ __launch_bounds__(256) void gemv_1x16_packed_finalize(const PackedFinalizeParams p) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= p.M) return;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < PK_S; ++k) s += p.partial[(size_t)k * p.M + row];
  const float scale = T::to_float(p.scales[row]);
  const float bias = p.bias ? T::to_float(p.bias[row]) : 0.f;
  p.y[row] = T::from_float(__builtin_fmaf(s, scale, bias));
}

struct PackedFinalizeSegment {
  PackedFinalizeParams f;
  int block_begin;
};

struct PackedFinalizeMultiParams {
  int nseg;
  PackedFinalizeSegment seg[AQLM_HIP_MAX_SEGMENTS];
};

template <class T>
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
  const int row = ((int)blockIdx.x - begin) * 256 + threadIdx.x;
  if (row >= p.M) return;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < PK_S; ++k) s += p.partial[(size_t)k * p.M + row];
  const float scale = T::to_float(p.scales[row]);
  const float bias = p.bias ? T::to_float(p.bias[row]) : 0.f;
  p.y[row] = T::from_float(__builtin_fmaf(s, scale, bias));
}

}  // namespace aqlm

using namespace aqlm;

extern "C" size_t aqlm_hip_prepack_1x16_bytes(int out_features, int in_features, int in_group_size) {
  PackedLayout L;
  return packed_layout(out_features, in_features, in_group_size, L) ? L.total : 0;
}

extern "C" int aqlm_hip_prepack_1x16(const void* codes, int out_features, int in_features, int in_group_size,
                                     void* packed, size_t packed_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PackedLayout L;
  if (!codes || !packed) {
    set_last_error("aqlm_hip_prepack_1x16: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (!packed_layout(out_features, in_features, in_group_size, L)) {
    set_last_error("aqlm_hip_prepack_1x16: unsupported shape (needs g=8, in %% 64 == 0, in <= 16320; got g=%d in=%d)",
                   in_group_size, in_features);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (packed_bytes < L.total || !aligned16(packed)) {
    set_last_error("aqlm_hip_prepack_1x16: packed buffer needs %zu bytes (16-B aligned), got %zu", L.total, packed_bytes);
    return AQLM_HIP_E_INVALID;
  }
  uint8_t* base = (uint8_t*)packed;
  // header (informational; the kernels take the layout from the shapes)
  const uint32_t hdr[16] = {0x31505141u, 4u, (uint32_t)L.M, (uint32_t)L.in_groups, 8u, (uint32_t)PK_S, (uint32_t)PK_NG,
                            (uint32_t)L.RG, (uint32_t)L.entries, (uint32_t)L.off_rowoff, (uint32_t)L.off_ent,
                            (uint32_t)L.off_perm, (uint32_t)(L.total & 0xffffffffu), (uint32_t)(L.total >> 32), 0u, 0u};
  if (int e = check_hip(hipMemsetAsync(base, 0, L.off_ent, stream), "prepack memset")) return e;
  if (int e = check_hip(hipMemcpyAsync(base, hdr, sizeof(hdr), hipMemcpyHostToDevice, stream), "prepack header")) return e;
  if (int e = check_hip(hipStreamSynchronize(stream), "prepack header sync")) return e;  // hdr is on the stack
  uint32_t* rowoff = (uint32_t*)(base + L.off_rowoff);
  uint16_t* perm = (uint16_t*)(base + L.off_perm);
  uint32_t* ent = (uint32_t*)(base + L.off_ent);
  if (int e = check_hip(hipMemsetAsync(ent, 0, L.total - L.off_ent, stream), "prepack memset entries")) return e;
  const int blocks = (L.M + 3) / 4;
  hipLaunchKernelGGL(prepack_count_kernel, dim3(blocks), dim3(256), 0, stream, (const uint16_t*)codes, rowoff, L.M,
                     L.in_groups, L.RG);
  hipLaunchKernelGGL(prepack_sort_kernel, dim3(PK_NG * PK_S), dim3(1024), 0, stream, rowoff, perm, L.RG);
  hipLaunchKernelGGL(prepack_scan_kernel, dim3(1), dim3(1024), 0, stream, rowoff, L.n_rowoff);
  hipLaunchKernelGGL(prepack_scatter_kernel, dim3(blocks), dim3(256), 0, stream, (const uint16_t*)codes, rowoff, perm, ent,
                     L.M, L.in_groups, L.RG);
  hipLaunchKernelGGL(prepack_arrange_kernel, dim3((L.M + 1) / 2), dim3(128), 0, stream, rowoff, perm, ent, L.M, L.in_groups,
                     L.RG);
  hipLaunchKernelGGL(prepack_level_kernel, dim3((L.RG / 2 + 63) / 64, PK_NG * PK_S), dim3(64), 0, stream, rowoff, ent, L.RG);
  hipLaunchKernelGGL(prepack_invert_kernel, dim3(PK_NG * PK_S), dim3(1024), 0, stream, perm, L.RG);
  return check_hip(hipGetLastError(), "prepack launch");
}

extern "C" int aqlm_hip_gemv_1x16_packed(const void* packed, const void* codebook, const void* scales, const void* bias,
                                         const void* x, void* y, int out_features, int in_features, int in_group_size,
                                         int dtype, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!packed || !codebook || !scales || !x || !y) {
    set_last_error("aqlm_hip_gemv_1x16_packed: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_gemv_1x16_packed: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  PackedLayout L;
  if (!packed_layout(out_features, in_features, in_group_size, L) || !aligned16(packed) || !aligned16(codebook) ||
      !aligned16(x)) {
    set_last_error("aqlm_hip_gemv_1x16_packed: unsupported shape or misaligned buffer (g=%d in=%d)", in_group_size,
                   in_features);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const size_t need = (size_t)PK_S * out_features * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    set_last_error("aqlm_hip_gemv_1x16_packed: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return AQLM_HIP_E_INVALID;
  }
  const uint8_t* base = (const uint8_t*)packed;
  PackedGemvParams p{};
  p.rowoff = (const uint32_t*)(base + L.off_rowoff);
  p.rowperm = (const uint16_t*)(base + L.off_perm);
  p.ent = (const uint32_t*)(base + L.off_ent);
  p.codebook = (const uint8_t*)codebook;
  p.x = (const uint16_t*)x;
  p.partial = (float*)workspace;
  p.M = L.M;
  p.in_groups = L.in_groups;
  p.RG = L.RG;
  p.ent_bytes = (uint32_t)((L.entries + PK_PAD) * 4);
#ifdef AQLM_PACKED_TRACE
  p.trace = workspace_bytes >= need + 2 * 256 * 8 * 8 ? (unsigned long long*)((uint8_t*)workspace + need) : nullptr;
#endif
  const size_t lds = (size_t)(PK_SLICE_ENTRIES + L.in_groups + 2) * 16;  // slice, x, zero slot, dump slot
  constexpr int NW = 16;
  auto launch = [&](auto kern) -> int {
    if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
    hipLaunchKernelGGL(kern, dim3(256), dim3(NW * 64), lds, stream, p);
    return check_hip(hipGetLastError(), "gemv_1x16_packed launch");
  };
  // rows per quarter-wave <= 2 (<= 4096-row layers): a shorter ring, so that the unrolled pipeline has fewer idle steps
  const bool short_rows = L.RG <= 2 * NW * 4;
  int e;
  if (L.in_groups <= 128)  // ~16 entries per bucket: 4 lanes per row
    e = dtype == AQLM_HIP_F16 ? launch(gemv_1x16_packed_kernel<F16, NW, 3, 4>) : launch(gemv_1x16_packed_kernel<BF16, NW, 3, 4>);
  else if (L.in_groups <= 256)  // ~32 entries per bucket: 8 lanes per row
    e = dtype == AQLM_HIP_F16 ? launch(gemv_1x16_packed_kernel<F16, NW, 3, 8>) : launch(gemv_1x16_packed_kernel<BF16, NW, 3, 8>);
  else if (short_rows)
    e = dtype == AQLM_HIP_F16 ? launch(gemv_1x16_packed_kernel<F16, NW, 2>) : launch(gemv_1x16_packed_kernel<BF16, NW, 2>);
  else
    e = dtype == AQLM_HIP_F16 ? launch(gemv_1x16_packed_kernel<F16, NW, 3>) : launch(gemv_1x16_packed_kernel<BF16, NW, 3>);
  if (e) return e;
  PackedFinalizeParams f{};
  f.partial = (const float*)workspace;
  f.scales = (const uint16_t*)scales;
  f.bias = (const uint16_t*)bias;
  f.y = (uint16_t*)y;
  f.M = out_features;
  if (dtype == AQLM_HIP_F16)
    hipLaunchKernelGGL(gemv_1x16_packed_finalize<F16>, dim3((out_features + 255) / 256), dim3(256), 0, stream, f);
  else
    hipLaunchKernelGGL(gemv_1x16_packed_finalize<BF16>, dim3((out_features + 255) / 256), dim3(256), 0, stream, f);
  return check_hip(hipGetLastError(), "gemv_1x16_packed_finalize launch");
}

extern "C" int aqlm_hip_gemv_1x16_packed_multi(const aqlm_hip_segment* segments, int num_segments, const void* x,
                                               int in_features, int in_group_size, int dtype, void* workspace,
                                               size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!segments || num_segments < 1 || num_segments > AQLM_HIP_MAX_SEGMENTS || !x) {
    set_last_error("aqlm_hip_gemv_1x16_packed_multi: 1..%d segments and a non-null x required (got %d)",
                   AQLM_HIP_MAX_SEGMENTS, num_segments);
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_gemv_1x16_packed_multi: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)",
                   dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  PackedMultiParams mp{};
  PackedFinalizeMultiParams fm{};
  mp.x = (const uint16_t*)x;
  mp.nseg = fm.nseg = num_segments;
  size_t need = 0;
  int fblocks = 0;
  for (int k = 0; k < num_segments; ++k) {
    const aqlm_hip_segment& sg = segments[k];
    if (!sg.codes || !sg.codebook || !sg.scales || !sg.y) {
      set_last_error("aqlm_hip_gemv_1x16_packed_multi: null pointer in segment %d", k);
      return AQLM_HIP_E_INVALID;
    }
    PackedLayout L;
    if (!packed_layout(sg.out_features, in_features, in_group_size, L) || !aligned16(sg.codes) ||
        !aligned16(sg.codebook) || !aligned16(x)) {
      set_last_error("aqlm_hip_gemv_1x16_packed_multi: unsupported shape or misaligned buffer (segment %d, g=%d in=%d)",
                     k, in_group_size, in_features);
      return AQLM_HIP_E_UNSUPPORTED;
    }
    const uint8_t* base = (const uint8_t*)sg.codes;
    PackedSegment& ps = mp.seg[k];
    ps.rowoff = (const uint32_t*)(base + L.off_rowoff);
    ps.rowperm = (const uint16_t*)(base + L.off_perm);
    ps.ent = (const uint32_t*)(base + L.off_ent);
    ps.codebook = (const uint8_t*)sg.codebook;
    ps.partial = (float*)((uint8_t*)workspace + need);
    ps.M = L.M;
    ps.RG = L.RG;
    ps.ent_bytes = (uint32_t)((L.entries + PK_PAD) * 4);
    mp.in_groups = L.in_groups;
    PackedFinalizeSegment& fs = fm.seg[k];
    fs.f.partial = ps.partial;
    fs.f.scales = (const uint16_t*)sg.scales;
    fs.f.bias = (const uint16_t*)sg.bias;
    fs.f.y = (uint16_t*)sg.y;
    fs.f.M = sg.out_features;
    fs.block_begin = fblocks;
    fblocks += (sg.out_features + 255) / 256;
    need += (size_t)PK_S * sg.out_features * sizeof(float);
  }
  if (!workspace || workspace_bytes < need) {
    set_last_error("aqlm_hip_gemv_1x16_packed_multi: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return AQLM_HIP_E_INVALID;
  }
  const size_t lds = (size_t)(PK_SLICE_ENTRIES + mp.in_groups + 2) * 16;  // slice, x, zero slot, dump slot
  constexpr int NW = 16;
  auto launch = [&](auto kern) -> int {
    if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
    hipLaunchKernelGGL(kern, dim3(256 * num_segments), dim3(NW * 64), lds, stream, mp);
    return check_hip(hipGetLastError(), "gemv_1x16_packed_multi launch");
  };
  int e;
  if (mp.in_groups <= 128)  // same lanes-per-row choice as the single-layer launch: results stay bit-identical
    e = dtype == AQLM_HIP_F16 ? launch(gemv_1x16_packed_multi_kernel<F16, NW, 3, 4>)
                              : launch(gemv_1x16_packed_multi_kernel<BF16, NW, 3, 4>);
  else if (mp.in_groups <= 256)
    e = dtype == AQLM_HIP_F16 ? launch(gemv_1x16_packed_multi_kernel<F16, NW, 3, 8>)
                              : launch(gemv_1x16_packed_multi_kernel<BF16, NW, 3, 8>);
  else
    e = dtype == AQLM_HIP_F16 ? launch(gemv_1x16_packed_multi_kernel<F16, NW, 3>)
                              : launch(gemv_1x16_packed_multi_kernel<BF16, NW, 3>);
  if (e) return e;
  if (dtype == AQLM_HIP_F16)
    hipLaunchKernelGGL(gemv_1x16_packed_finalize_multi<F16>, dim3(fblocks), dim3(256), 0, stream, fm);
  else
    hipLaunchKernelGGL(gemv_1x16_packed_finalize_multi<BF16>, dim3(fblocks), dim3(256), 0, stream, fm);
  return check_hip(hipGetLastError(), "gemv_1x16_packed_finalize_multi launch");
}
