// 1x16 g8 at 2 .. 128+ input rows: codebook SLICES in LDS, canonical codes scanned, one sparse MFMA per 64 codes (round 6).
//
// STATUS: parity-green, NOT on any default route (tuning key `gemm_variant` = 4 or `scan_max_rows` > 0 selects it).  Measured
// (profiles/r06_scan_kernel_vs_routes.log, us, this kernel / round 5's L2-gather MFMA op / dense fp16): 4096 x 4096 at 8 rows 17.8 /
// 12.8 / 12.8; 4096 -> 11008 30.2 / 31.8 / 20.6; 11008 -> 4096 32.9 / 30.3 / 29.6.  The knock-out runs
// (profiles/r06_scan_kernel_knockouts.log) say why: 11.5 us are fixed (launch 1.6, the 128 KiB slice 2.3, the x fragments 2.1, the
// finalize launch 2.8, first codes / drain 2.7) and every 16-row tile costs 0.6 us of VALU + LDS + MFMA work that is 8-fold
// redundant by construction (each code is looked at by the eight workgroups that own the eight slices) + 0.3 us for the K-range
// reduction.  Even with all three pipelines perfectly overlapped the 8-fold scan of 4096 -> 11008 is ~9.5 us on top of the fixed
// part, i.e. the dense GEMM's 20.6 us: the formulation trades the L2-gather floor (2.3 clocks per code and CU) for ~1 clock per
// code and CU plus 5 us more fixed cost, which does not pay at Llama layer sizes.  Kept as the data-oblivious, prepack-free
// multi-row kernel and as the record of that measurement.
//
// Replaces (behaviour, not code): the reference's handling of more than one row of the 1x16 scheme -- the per-row relaunch of its
// matvec (cuda_kernel.cpp:165-175) up to 6 rows and code1x16_matmat_dequant = dequantise W to HBM + cuBLAS (cuda_kernel.cpp:249-301)
// above.  Rounds 1-5 served these calls with (a) the slice-bucketed matvec, whose x operand is an LDS gather of 16 B per row and
// entry (8.9 / 12.3 / 18.3 / 36.3 us at 1 / 2 / 4 / 8 rows of 4096 -> 11008), and (b) MFMA kernels that gather the codebook from L2
// at 0.43 sixteen-byte gathers per clock and CU (13 us at 4096 x 4096, 31 us at 4096 -> 11008, whatever the row count): from 5 rows on
// the scheme lost to a dense fp16 GEMM.
//
// The formulation here keeps BOTH operands off those two floors:
//   * the codebook is cut into 8 slices of 8192 entries = 128 KiB; a workgroup (one per CU) holds ONE slice in LDS for its whole
//     life and walks 16-row tiles of the canonical codes [out][in / 8] (no prepacked copy, no dependence on the code histogram);
//   * a lane of the W fragment of v_mfma_f32_16x16x32 is exactly one code: lane (row r, k-quarter q).  The lane looks at ITS code:
//     in this workgroup's slice -> ds_read_b128 of the entry; not -> it reads a zero vector (address clamp: min(code ^ slice_bits,
//     8192) * 16, three VALU operations).  The MFMA then adds an exact 0 for the 7/8 of the lanes whose entries live in the other
//     slices; the eight workgroups that own the eight slices of a row group each produce one partial sum -- the matrix cores have
//     the room: at <= 16 rows they would otherwise idle;
//   * x never touches LDS: wave w of a workgroup owns the k range [w * 256 U, (w + 1) * 256 U) of the workgroup's K chunk and keeps
//     the x fragments of that range (8 U k-steps x 16 B per lane and 16 batch rows) in registers for ALL tiles;
//   * per tile the eight K-range partial tiles (16 x 16 fp32 each) meet in LDS (7 writers, the tile's reducer wave adds them in a
//     fixed order) and go to a workspace [8 slices x K chunks][B][M]; a small second kernel adds the planes in order, applies
//     scales + bias and rounds once.
// Cost per (tile, k-step, slice): one ds_read_b128 (4 LDS clocks per wave-instruction), one MFMA (~17 clocks of one SIMD), 3 VALU --
// 8 slices x M/16 x K/32 of them: 4096 -> 11008 = 2752 per CU = ~5 us of LDS / MFMA time for ANY row count up to 16 (32: two MFMAs
// per k-step, same LDS time); the codes are read 8 times, from L2 (the eight slices of a row group sit on one XCD).
// Arithmetic: exact fp16 / bf16 products, fp32 sums in a fixed order that depends on the layer shape only -> deterministic and
// batch-invariant (a row's bits do not depend on the other rows nor on their number); NaN / Inf in x poison their own row only.
#include <algorithm>

#include "aqlm_common.h"

namespace aqlm {
namespace scan {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4* lds_u32x4_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(1))) const void* gbl_void_ptr;

template <class T>
__device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, const f32x4& c);
template <>
__device__ __forceinline__ f32x4 mfma<F16>(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma<BF16>(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int NW = 8;                   // waves per workgroup = K ranges of the workgroup's chunk
constexpr int NSLICES = 8;              // codebook slices of 8192 entries (128 KiB of fp16 x 8)
constexpr uint32_t SLICE_ENTRIES = 8192;
constexpr uint32_t LDS_CB = 0;          // the slice: entry e at byte 16 e
constexpr uint32_t LDS_ZERO = 131072;   // entry 8192: sixteen zero bytes (1 KiB reserved: the red buffers stay KiB-aligned)
constexpr uint32_t LDS_RED = 132096;    // [2 buffers][NW - 1 writers][NBT][64 lanes][16 B] fp32 partial tiles
template <int NBT>
constexpr uint32_t lds_total() { return LDS_RED + 2u * (NW - 1) * NBT * 1024u; }  // 146432 (NBT 1) / 160768 (NBT 2) <= 163840

struct Params {
  const uint16_t* codes;    // [M][G] u16 (two's-complement containers of unsigned indices)
  const uint8_t* codebook;  // [65536][8] halfs
  const uint16_t* X;        // [B][xs]
  float* partial;           // [planes = NSLICES * kchunks][B][M]
  long xs;
  int M, B, G;              // G = in_features / 8 (a multiple of 32: whole units of 32 groups = 8 k-steps)
  int ntiles;               // ceil(M / 16)
  int RG;                   // row groups: tiles [rg * ntiles / RG, (rg + 1) * ntiles / RG)
  int kchunks;              // K chunks; chunk kc covers units [kc * upc, min((kc + 1) * upc, units))
  int units, upc;           // units = G / 32; units per chunk
  int per_xcd;              // work items (row group, chunk, slice) per XCD = RG * kchunks * NSLICES / 8
};

// One work item per workgroup.  U = units (8 k-steps = 256 features each) per wave, NBT = 16-row batch tiles per pass, D = tiles of
// code words in flight.
//
// What the loop is built around (measured on the first cuts, profiles/r06_scan_kernel_knockouts.log):
//   * NO branch around a VMEM instruction and no compiler-visible store in the loop: hipcc's wait-count insertion answers either
//     with vmcnt(0), which serialised the code prefetch on memory latency (1.5 us per tile).  Code loads are unconditional (clamped
//     addresses); the one store per tile is inline asm (loads return in order among themselves, so a store the compiler does not
//     count can only make one of its counted waits longer, never too short);
//   * two phases per tile: all 8 U addresses + ds_read_b128 first, then the 8 U (x NBT) MFMAs back to back -- the two waves of a
//     SIMD alternate, one gathering while the other multiplies (interleaved, every MFMA waited for a gather issued three
//     instructions earlier);
//   * the tile's reducer does not make the others wait: it requests the seven partial tiles right after the barrier and adds them
//     one tile later, inside its own gather phase's LDS latency.
template <class T, int U, int NBT, int D>
__global__ __launch_bounds__(NW * 64) void gemm_1x16_scan_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char scan_smem[];
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)scan_smem != 0u) __builtin_trap();  // LDS map above starts at 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int arow = lane & 15, kg = lane >> 4;
  // work item: contiguous eighths of the (row group, chunk, slice) list go to one XCD each (block b runs on XCD b % 8), so the
  // eight slice-workgroups that read the same codes share an L2 (speed only)
  const int w_item = ((int)blockIdx.x & 7) * p.per_xcd + ((int)blockIdx.x >> 3);
  const int per_rg = p.kchunks * NSLICES;
  const int rg = w_item / per_rg, q = w_item - rg * per_rg;
  const int slice = q & (NSLICES - 1), kc = q >> 3;
  if (rg >= p.RG) return;
  const int tile0 = (int)(((long)rg * p.ntiles) / p.RG), tile1 = (int)(((long)(rg + 1) * p.ntiles) / p.RG);

  // ---- prologue: the slice by LDS-DMA (128 pieces of 1 KiB, rotated by the row group so that the workgroups sharing a slice do not
  // all pull the same lines at the same moment), the zero entry, and this wave's first code words
  {
    const uint8_t* src = p.codebook + (size_t)slice * (SLICE_ENTRIES * 16u) + (size_t)lane * 16u;
    for (int i = wave; i < 128; i += NW) {
      const int piece = (i + rg * 8) & 127;
      __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + (size_t)piece * 1024u), (lds_void_ptr)(size_t)(LDS_CB + (uint32_t)piece * 1024u), 16, 0, 0);
    }
    if (wave == 0) *reinterpret_cast<u32x4*>(scan_smem + LDS_ZERO + (uint32_t)lane * 16u) = u32x4{0u, 0u, 0u, 0u};
  }
  // Units past the end of the chunk (the last waves of a ragged chunk: K = 11008 is 43 units) are loaded like the others, from a
  // clamped address, and skipped by wave-uniform branches that contain LDS reads and MFMAs only.
  const int ubase = kc * p.upc + wave * U;                            // first unit of this wave
  const int uend = min((kc + 1) * p.upc, p.units);                    // end of the chunk
  const uint32_t sb2 = ((uint32_t)slice << 13) * 0x10001u;            // both halves of a code word: (code ^ slice bits) < 8192 <=> in this slice
  int uoff[U];                                                        // unit offset actually loaded (wave-uniform)
  bool uvalid[U];
  const int ub = ubase < uend ? ubase : uend - 1;                     // (a wave with no unit at all reads the chunk's last one)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uvalid[u] = ubase + u < uend;
    uoff[u] = uvalid[u] ? u : 0;
  }
  auto load_codes = [&](int tile, u32x4 (&cw)[U]) {
    int r = tile * 16 + arow;
    r = r < p.M ? r : p.M - 1;
    const uint16_t* src = p.codes + (size_t)r * p.G + (size_t)(ub * 32 + kg * 8);
#pragma unroll
    for (int u = 0; u < U; ++u) cw[u] = *reinterpret_cast<const u32x4*>(src + uoff[u] * 32);
  };
  u32x4 ring[D][U];                                                   // code words of the current tile and of the D - 1 after it
  auto prime = [&]() {
#pragma unroll
    for (int d = 0; d < D; ++d) load_codes(tile0 + d < tile1 ? tile0 + d : tile1 - 1, ring[d]);
  };
  prime();

  const int plane = kc * NSLICES + slice;
  int it = 0;                                                         // tiles done by this workgroup (buffer parity)
  bool first_pass = true;
  for (int b0 = 0; b0 < p.B; b0 += 16 * NBT) {
    // ---- the x fragments of this wave's K range for this pass: lane (batch column arow, k-quarter kg), k-step t of unit u = groups
    // {8 kg + t} of the unit -- the same assignment as the code words (lane (row, kg) holds the codes of groups 8 kg .. 8 kg + 7), so
    // no cross-lane movement is needed anywhere.  Columns past the batch stay zero (their lanes do not load: at 8 rows that halves
    // what the prologue pulls through the L1).
    u32x4 xf[NBT][U][8];
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb) {
      const int b = b0 + nb * 16 + arow;
      const uint16_t* xr = p.X + (size_t)(b < p.B ? b : 0) * p.xs + (size_t)(ub * 32 + kg * 8) * 8;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          u32x4 v = u32x4{0u, 0u, 0u, 0u};
          if (b < p.B && uvalid[u]) v = *reinterpret_cast<const u32x4*>(xr + (uoff[u] * 32 + t) * 8);
          xf[nb][u][t] = v;
        }
    }
    if (!first_pass) prime();                                         // later passes walk the same tiles again: restart the code ring
    if (first_pass) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                   // the slice is in LDS, for good
      first_pass = false;
    }
    f32x4 rv[NBT][NW - 1];                                            // the reducer's requests of the previous tile, added one tile later
    f32x4 rmine[NBT];
    int rtile = -1;                                                   // tile this wave still has to finish (-1: none)
    auto finish = [&](int b0_) {                                      // wave-uniform: rtile >= 0
#pragma unroll
      for (int nb = 0; nb < NBT; ++nb) {
        f32x4 v = rmine[nb];
#pragma unroll
        for (int s = 0; s < NW - 1; ++s) v = v + rv[nb][s];           // order (red, red + 1, ..) mod NW: a function of the tile only
        const int b = b0_ + nb * 16 + arow, m = rtile * 16 + kg * 4;  // lane (batch column arow, rows 4 kg .. 4 kg + 3)
        if (b < p.B && m < p.M) {
          float* dst = p.partial + ((size_t)plane * p.B + b) * p.M + m;
          if ((p.M & 3) == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
          else
            for (int r = 0; r < 4; ++r)
              if (m + r < p.M) asm volatile("global_store_dword %0, %1, off\n\ts_nop 1" ::"v"(dst + r), "v"(v[r]) : "memory");
        }
      }
      rtile = -1;
    };
    for (int tile = tile0; tile < tile1; ++tile, ++it) {
      u32x4 cw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        cw[u] = ring[0][u];
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) ring[d][u] = ring[d + 1][u];
      }
      load_codes(tile + D < tile1 ? tile + D : tile1 - 1, ring[D - 1]);  // (past the end: the last tile again, never used)
      // ---- phase 1: addresses and gathers of all units.  Per code word (two codes): xor with the slice bits, packed min with 8192
      // (everything outside the slice -> the zero entry), then one shift per code
      u32x4 w[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!uvalid[u]) continue;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
          const u16x2 x2 = __builtin_bit_cast(u16x2, cw[u][h] ^ sb2);
          const u16x2 m2 = __builtin_elementwise_min(x2, u16x2{(unsigned short)SLICE_ENTRIES, (unsigned short)SLICE_ENTRIES});
          w[u][2 * h] = *(lds_u32x4_ptr)(size_t)(LDS_CB + ((uint32_t)m2[0] << 4));
          w[u][2 * h + 1] = *(lds_u32x4_ptr)(size_t)(LDS_CB + ((uint32_t)m2[1] << 4));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (rtile >= 0) finish(b0);                                     // the previous tile's sums: inside this tile's gather latency
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase 2: the MFMAs, two accumulator chains per batch tile
      f32x4 acc[NBT][2];
#pragma unroll
      for (int nb = 0; nb < NBT; ++nb) acc[nb][0] = acc[nb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!uvalid[u]) continue;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int nb = 0; nb < NBT; ++nb) acc[nb][t & 1] = mfma<T>(w[u][t], xf[nb][u][t], acc[nb][t & 1]);
      }
      // ---- the K ranges meet: everybody but the tile's reducer writes; the reducer requests the seven tiles and adds them later
      const int red = tile & (NW - 1), buf = it & 1;
      f32x4 mine[NBT];
#pragma unroll
      for (int nb = 0; nb < NBT; ++nb) mine[nb] = acc[nb][0] + acc[nb][1];
      if (wave != red) {
        const int s = (wave - red - 1) & (NW - 1);
#pragma unroll
        for (int nb = 0; nb < NBT; ++nb)
          *reinterpret_cast<f32x4*>(scan_smem + LDS_RED + (uint32_t)(((buf * (NW - 1) + s) * NBT + nb) * 1024) + (uint32_t)lane * 16u) = mine[nb];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (wave == red) {
#pragma unroll
        for (int nb = 0; nb < NBT; ++nb) {
          rmine[nb] = mine[nb];
#pragma unroll
          for (int s = 0; s < NW - 1; ++s)
            rv[nb][s] = *reinterpret_cast<const f32x4*>(scan_smem + LDS_RED + (uint32_t)(((buf * (NW - 1) + s) * NBT + nb) * 1024) + (uint32_t)lane * 16u);
        }
        rtile = tile;
      }
    }
    if (rtile >= 0) finish(b0);
  }
}

// Y[b][m] = round((sum over the planes, in plane order) * scales[m] + bias[m]); thread = 4 consecutive m of one batch row
struct FinalizeParams {
  const float* partial;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* Y;
  long ys;
  int M, B, planes;
};

template <class T>
__global__ __launch_bounds__(256) void gemm_1x16_scan_finalize_kernel(const FinalizeParams p) {
  const int m = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
  const int b = (int)blockIdx.y;
  if (m >= p.M) return;
  const size_t stride = (size_t)p.B * p.M;
  const float* src = p.partial + (size_t)b * p.M + m;
  float sc[4], bi[4];  // requested before the partials: one exposed round trip instead of two
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int mm = m + r < p.M ? m + r : p.M - 1;
    sc[r] = T::to_float(p.scales[mm]);
    bi[r] = p.bias ? T::to_float(p.bias[mm]) : 0.f;
  }
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if ((p.M & 3) == 0) {
    int k = 0;
    for (; k + 8 <= p.planes; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(k + j) * stride);
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += v[j][r];
    }
    for (; k < p.planes; ++k) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)k * stride);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] += v[r];
    }
  } else {
    for (int k = 0; k < p.planes; ++k)
      for (int r = 0; r < 4; ++r)
        if (m + r < p.M) s[r] += src[(size_t)k * stride + r];
  }
  uint16_t h[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) h[r] = T::from_float(__builtin_fmaf(s[r], sc[r], bi[r]));
  uint16_t* dst = p.Y + (size_t)b * p.ys + m;
  if ((p.M & 3) == 0 && (p.ys & 3) == 0 && ((uintptr_t)dst & 7u) == 0)
    *reinterpret_cast<u32x2*>(dst) = u32x2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
  else
    for (int r = 0; r < 4; ++r)
      if (m + r < p.M) dst[r] = h[r];
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
struct Plan {
  int kchunks, upc, U, RG, nbt;
};

static int device_cus() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  return cus;
}

// K chunks x units per wave x row groups: one round of the chip (a workgroup owns a CU: 128 KiB of LDS), as few planes as the x
// fragments' registers allow (U <= 3 at 16 rows per pass, <= 2 at 32), the fewest k-steps on the longest workgroup
bool plan(int B, int M, int K, Plan& pl) {
  if (B < 1 || M < 1 || K < 256 || K % 256 != 0) return false;
  const int units = K / 256, ntiles = (M + 15) / 16, cus = device_cus();
  // (the K chunking must not depend on the row count: a row's bits are the same in a 2-row and in a 100-row call)
  const int nbt = 1, umax = 2;  // (32-row passes -- two MFMAs per gather -- spill at U = 2: passes of 16 rows for now)
  long best = -1;
  for (int kc = 1; kc <= units && kc <= 32; ++kc) {
    const int upc = (units + kc - 1) / kc;
    if ((long)(kc - 1) * upc >= units) continue;                       // an empty chunk
    const int U = (upc + NW - 1) / NW;
    if (U > umax) continue;
    const int RG = std::max(1, std::min(ntiles, cus / (NSLICES * kc)));
    if ((RG * kc * NSLICES) % 8 != 0) continue;
    const long tiles = (ntiles + RG - 1) / RG;
    const long cost = tiles * (U * 8 + 3) + 2 * kc;                   // k-steps + the tile's reduction; a plane more costs the finalize a little
    if (best < 0 || cost < best) {
      best = cost;
      pl = Plan{kc, upc, U, RG, nbt};
    }
  }
  return best >= 0;
}

size_t workspace_bytes(int B, int M, int K) {
  Plan pl;
  if (!plan(B, M, K, pl)) return 0;
  return (size_t)NSLICES * pl.kchunks * (size_t)B * M * sizeof(float);
}

template <class T>
static int launch(const Params& p, const Plan& pl, hipStream_t stream) {
  const dim3 grid((unsigned)(pl.RG * pl.kchunks * NSLICES));
  auto go = [&](auto kern, size_t lds) -> int {
    if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, stream, p);
    return check_hip(hipGetLastError(), "gemm_1x16_scan launch");
  };
  const int depth = tuning().scan_prefetch;
  if (pl.U == 1) return depth == 2 ? go(gemm_1x16_scan_kernel<T, 1, 1, 2>, lds_total<1>()) : go(gemm_1x16_scan_kernel<T, 1, 1, 4>, lds_total<1>());
  return depth == 2 ? go(gemm_1x16_scan_kernel<T, 2, 1, 2>, lds_total<1>()) : go(gemm_1x16_scan_kernel<T, 2, 1, 4>, lds_total<1>());
}

// the whole op: scan kernel + finalize.  Callers have validated pointers, dtype and alignment.
int run(const void* codes, const void* codebook, const void* scales, const void* bias, const void* X, void* Y, int batch, int out_features,
        int in_features, long xs, long ys, int dtype, void* workspace, hipStream_t stream) {
  Plan pl;
  if (!plan(batch, out_features, in_features, pl)) return AQLM_HIP_E_UNSUPPORTED;
  Params p{};
  p.codes = (const uint16_t*)codes;
  p.codebook = (const uint8_t*)codebook;
  p.X = (const uint16_t*)X;
  p.partial = (float*)workspace;
  p.xs = xs;
  p.M = out_features;
  p.B = batch;
  p.G = in_features / 8;
  p.ntiles = (out_features + 15) / 16;
  p.RG = pl.RG;
  p.kchunks = pl.kchunks;
  p.units = in_features / 256;
  p.upc = pl.upc;
  p.per_xcd = pl.RG * pl.kchunks * NSLICES / 8;
  if (int e = dtype == AQLM_HIP_F16 ? launch<F16>(p, pl, stream) : launch<BF16>(p, pl, stream)) return e;
  FinalizeParams f{};
  f.partial = (const float*)workspace;
  f.scales = (const uint16_t*)scales;
  f.bias = (const uint16_t*)bias;
  f.Y = (uint16_t*)Y;
  f.ys = ys;
  f.M = out_features;
  f.B = batch;
  f.planes = NSLICES * pl.kchunks;
  const dim3 grid((unsigned)((out_features + 1023) / 1024), (unsigned)batch);
  if (dtype == AQLM_HIP_F16) hipLaunchKernelGGL(gemm_1x16_scan_finalize_kernel<F16>, grid, dim3(256), 0, stream, f);
  else hipLaunchKernelGGL(gemm_1x16_scan_finalize_kernel<BF16>, grid, dim3(256), 0, stream, f);
  return check_hip(hipGetLastError(), "gemm_1x16_scan_finalize launch");
}

}  // namespace scan
}  // namespace aqlm

using namespace aqlm;

extern "C" size_t aqlm_hip_gemm_1x16_scan_workspace_bytes(int batch, int out_features, int in_features) {
  return scan::workspace_bytes(batch, out_features, in_features);
}

extern "C" int aqlm_hip_gemm_1x16_scan(const void* codes, const void* codebook, const void* scales, const void* bias, const void* X, void* Y,
                                       int batch, int out_features, int in_features, long xs, long ys, int dtype, void* workspace,
                                       size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!codes || !codebook || !scales || !X || !Y) {
    set_last_error("aqlm_hip_gemm_1x16_scan: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (batch <= 0 || out_features <= 0 || in_features <= 0) {
    set_last_error("aqlm_hip_gemm_1x16_scan: sizes must be positive");
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_gemm_1x16_scan: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (in_features % 256 != 0 || !aligned16(codes) || !aligned16(codebook) || !aligned16(X) || xs % 8 != 0) {
    set_last_error("aqlm_hip_gemm_1x16_scan: needs in_features %% 256 == 0 (codebook vectors of 8) and 16-B aligned codes / codebook / X rows");
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const size_t need = scan::workspace_bytes(batch, out_features, in_features);
  if (need == 0) {
    set_last_error("aqlm_hip_gemm_1x16_scan: no plan for %d rows of %d -> %d", batch, in_features, out_features);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (!workspace || workspace_bytes < need || !aligned16(workspace)) {
    set_last_error("aqlm_hip_gemm_1x16_scan: 16-B aligned workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return AQLM_HIP_E_INVALID;
  }
  return scan::run(codes, codebook, scales, bias, X, Y, batch, out_features, in_features, xs, ys, dtype, workspace, stream);
}
