// 8 x 8-bit matvec (any g multiple of 8, e.g. the 2-bit 8x8 g32 scheme) through per-token look-up tables in LDS.  gfx950.
//
// Why: with 8 codebooks of 256 x g the direct kernel (gemv.hip) reads 8 x g/8 x 16 B of LDS and issues 8 x g/2
// v_dot2c per input group -- 16 B of LDS traffic per weight at g = 32, 4x the 2x8 scheme -- and ran at 3.5 % of the HBM
// roofline (15 us for 4096x4096).  The reference's CPU kernel avoids exactly this with
//     lut[j, c, v] = < codebooks[c, v], x_j >      ;      y[i] = sum_{j, c} lut[j, c, codes[i, j, c]]
// (numba_kernel.py:37-48).  On the GPU the table of one token is in_groups x 8 x 256 fp32 = 1 MiB (in = 4096): too big
// for one CU, so the input groups are cut into slabs of 16 (16 x 8 x 256 x 4 B = 128 KiB of LDS): workgroup
// (slab, row range) builds its slab of the table (a [2048 x g] x [g x 16] product, on the matrix cores), then every quarter-wave walks rows: lane = one input group = 8 code bytes = 8 ds_read_b32 + 8 adds.  fp32
// partials [slab][row] -> finalize (sum over slabs, scale, bias, one rounding).  Per code: one 4-B LDS read and one
// add instead of g/8 ds_read_b128 and g/2 v_dot2c.  Arithmetic is the same fp32 accumulation of exact fp16/bf16
// products, in a different association order.
#include <algorithm>
#include <type_traits>

#include "aqlm_common.h"

namespace aqlm {

constexpr int LUT_KC = 8;
constexpr int LUT_JS = 16;                 // canonical codes: input groups per slab (x all 8 codebooks)
constexpr int LUT_PJ = 128;                // planar codes: input groups per slab (x ONE codebook)
constexpr int LUT_ENTRIES = 32768;         // fp32 table entries of a slab = 128 KiB
constexpr int LUT_OPER_OFF = LUT_ENTRIES * 4;       // planar: the slab's codebook (<= 16 KiB) and its x (<= 8 KiB), staged once per workgroup
constexpr int LUT_XOPER_OFF = LUT_OPER_OFF + 16384;
constexpr int LUT_STAGE_OFF = LUT_OPER_OFF;         // row totals waiting for the hand-in (the operands are dead by then)
constexpr int LUT_STAGE_ROWS = 2048;
constexpr int LUT_SLOTS_OFF = LUT_OPER_OFF + 24576; // [16] per-wave maxima
constexpr int LUT_LDS_BYTES = LUT_SLOTS_OFF + 256;

typedef _Float16 lut_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 lut_bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class T>
__device__ __forceinline__ f32x4 lut_mfma16(const u32x4& a, const u32x4& b, const f32x4& c);
template <>
__device__ __forceinline__ f32x4 lut_mfma16<F16>(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lut_f16x8, a), __builtin_bit_cast(lut_f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 lut_mfma16<BF16>(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lut_bf16x8, a), __builtin_bit_cast(lut_bf16x8, b), c, 0, 0, 0);
}

struct LutParams {
  const uint8_t* codes;      // canonical [M][in_groups][8], or planar [8][M][jp] (aqlm_hip_8x8_planar_pack)
  const uint16_t* codebooks; // [8][256][G]
  const uint16_t* x;
  float* partial;            // [nslabs][M]
  int M, in_groups, nslabs, nranges, rows_per_range;
  int jp;                    // planar: bytes per row of a codebook plane (in_groups rounded up to 4)
  float cb_absmax;           // planar, fused finalize: max |codebook entry| (every workgroup sees one codebook only)
  // fused finalize (cells != nullptr): the slab sums of a row meet in one zero-at-rest 64-bit cell, see the body
  unsigned long long* cells;  // [M]
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
};

// magnitude pattern (bits & 0x7fff of every half) maximum of four dwords, folded into `m` (v_pk_max_u16)
typedef unsigned short lut_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lut_absmax(lut_us2& m, const u32x4& v) {
  m = __builtin_elementwise_max(m, __builtin_bit_cast(lut_us2, v.x & 0x7fff7fffu));
  m = __builtin_elementwise_max(m, __builtin_bit_cast(lut_us2, v.y & 0x7fff7fffu));
  m = __builtin_elementwise_max(m, __builtin_bit_cast(lut_us2, v.z & 0x7fff7fffu));
  m = __builtin_elementwise_max(m, __builtin_bit_cast(lut_us2, v.w & 0x7fff7fffu));
}
// packed 16-bit maximum over the 16 lanes of a DPP row; every lane of the row receives it
__device__ __forceinline__ lut_us2 lut_row16_pkmax(lut_us2 m) {
  m = __builtin_elementwise_max(m, __builtin_bit_cast(lut_us2, dpp_u32<0xB1>(__builtin_bit_cast(uint32_t, m))));
  m = __builtin_elementwise_max(m, __builtin_bit_cast(lut_us2, dpp_u32<0x4E>(__builtin_bit_cast(uint32_t, m))));
  m = __builtin_elementwise_max(m, __builtin_bit_cast(lut_us2, dpp_u32<0x141>(__builtin_bit_cast(uint32_t, m))));
  m = __builtin_elementwise_max(m, __builtin_bit_cast(lut_us2, dpp_u32<0x140>(__builtin_bit_cast(uint32_t, m))));
  return m;
}

// LDS image of a slab (round 4).  The table is 32 slots x 4 sub-tables x 256 values; entry (slot s, sub-table k, value v) at
//     v * 256 + (k & 1) * 128 + s * 4 + (k >> 1) * 65536   [bytes]:
// the bank of an entry is a function of s alone, and s is the LANE of the row walk (32 lanes = one row's 128 code bytes of the
// slab, one dword = 4 code bytes = k 0..3 each), so a wave-wide ds_read_b32 never has a bank conflict whatever the codes are.
//   canonical codes: slot s = (input group jl = s / 2, codebooks 4 ch .. 4 ch + 3 with ch = s & 1), k = codebook 4 ch + k
//   planar codes:    slot s, k = input group jb + 4 s + k of the slab's one codebook
// Round 3's image was lut[jl][c][v]: bank = v mod 32 -- random for the reads, and 16 equal banks for the 16 lanes of every
// table write (79 % of the LDS cycles were conflict cycles, profiles/r03_8x8_lut_kernel_pmc.json).
__device__ __forceinline__ uint32_t lut_perm(uint32_t cw, uint32_t base, uint32_t sel) { return __builtin_amdgcn_perm(cw, base, sel); }

// `block` = the workgroup's index within its own layer (== blockIdx.x for a single-layer launch).  NW waves per workgroup.
template <class T, int G, int NW, bool PLANAR>
__device__ __forceinline__ void gemv_8x8_lut_body(const LutParams& p, const int block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef __attribute__((address_space(3))) const float* lds_f32_ptr;
  constexpr int NT = NW * 64;
  constexpr int P = G / 8;  // pieces of 8 k per codebook vector: 1, 2 or 4

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef AQLM_LUT_TRACE  // profiling builds only (tools/microbench `make trace`): wall-clock stamps (100 MHz) per wave behind the cells
  unsigned long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define LUT_TRACE(i) do { __builtin_amdgcn_sched_barrier(0); tr[i] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LUT_TRACE(i)
#endif
  LUT_TRACE(0);
  const int slab = block % p.nslabs, range = block / p.nslabs;
  const int row_begin = range * p.rows_per_range;
  int nrows = p.M - row_begin;
  nrows = nrows < 0 ? 0 : (nrows < p.rows_per_range ? nrows : p.rows_per_range);
  // planar: slab = (codebook slab % 8, input groups jb .. jb + 127): with block -> XCD = block % 8 every XCD's L2 serves ONE
  // codebook (16 KiB at g = 32) and one plane of the codes.  canonical: slab = input groups j0 .. j0 + 15, all codebooks.
  const int pc = PLANAR ? (slab & 7) : 0;
  const int j0 = PLANAR ? (slab >> 3) * LUT_PJ : slab * LUT_JS;

  // ---- (1) everything the workgroup needs from memory is requested up front, smallest and most urgent first
  // x for the max |x| of the fused finalize: a few KiB, one or two 16-B pieces per thread (more only above 16 Ki features)
  const int chunks = p.in_groups * P;  // 16-B pieces of x
  const u32x4 zero = {0u, 0u, 0u, 0u};
  u32x4 xq0 = zero, xq1 = zero;
  if (p.cells != nullptr) {
    if (tid < chunks) xq0 = reinterpret_cast<const u32x4*>(p.x)[tid];
    if (tid + NT < chunks) xq1 = reinterpret_cast<const u32x4*>(p.x)[tid + NT];
  }
  const int col = lane & 15, kg = lane >> 4;
  // MFMA operands.  v_mfma_f32_16x16x32: A = 16 rows x 32 k (lane l: row l % 16, 8 k of piece l / 16), B = 16 columns
  // (lane l: column l % 16, same piece), D[row (l / 16) * 4 + r][column l % 16].  g < 32 pads k with zero pieces.
  //   canonical: A = 16 codebook rows, B = x of the slab's 16 groups.  A tile's rows are  q * 4 + vs * 2 + ch  =  codebook
  //     4 ch + k, value v0 + 2 q + vs  (k, v0 per tile: 4 x 32 tiles), so the 4 D registers of lane l are (v, ch = 0 / 1) and
  //     (v + 1, ch = 0 / 1) of group l % 16: two 8-byte LDS writes whose 16 lanes cover 128 contiguous bytes (conflict-free).
  //   planar: A = 16 consecutive values of the slab's codebook, B = x of 16 groups with ONE k and 16 consecutive slots
  //     (j = jb + 4 (s0 + column) + k): a D register is one ds_write_b32 over 16 slots x 4 values (2-way, free for writes).
  //   planar: the slab needs 256 codebook rows and 128 groups of x: 24 KiB at g = 32, fetched ONCE per workgroup with coalesced
  //     loads into LDS (every wave re-reading its fragments from memory moved 128-256 KiB through the L1 -- as much as the
  //     canonical layout's whole codebook); fragments come from there (ds_read_b128, 1 KiB contiguous per wave-read).
  constexpr int VT = 4;                            // planar: A fragments (16 table rows each) per wave
  constexpr int CT = 128 / (NW * VT);              // planar: B fragments per wave (NW 8: 4, NW 16: 2)
  constexpr int NA = PLANAR ? VT : 128 / NW;       // A fragments per wave
  constexpr int NB = PLANAR ? CT : 1;              // B fragments per wave
  u32x4 afrag[NA], bfrag[NB];
  constexpr int CBP = (256 * P + NT - 1) / NT;     // planar: 16-B pieces of the codebook / of x per thread
  constexpr int XP = (LUT_PJ * P + NT - 1) / NT;
  u32x4 stage_cb[PLANAR ? CBP : 1], stage_x[PLANAR ? XP : 1];
  if constexpr (PLANAR) {
#pragma unroll
    for (int q = 0; q < CBP; ++q) {
      const int i = tid + q * NT;
      stage_cb[q] = i < 256 * P ? reinterpret_cast<const u32x4*>(p.codebooks + (size_t)pc * 256 * G)[i] : zero;
    }
#pragma unroll
    for (int q = 0; q < XP; ++q) {
      const int i = tid + q * NT, j = j0 + i / P;
      stage_x[q] = (i < LUT_PJ * P && j < p.in_groups) ? reinterpret_cast<const u32x4*>(p.x + (size_t)j0 * G)[i] : zero;
    }
  } else {
    bfrag[0] = (kg < P && j0 + col < p.in_groups) ? reinterpret_cast<const u32x4*>(p.x + (size_t)(j0 + col) * G)[kg] : zero;
    const int arow_c = (col & 1) * 4, arow_v = (col >> 2) * 2 + ((col >> 1) & 1);  // this lane's A row within a tile
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      const int tile = wave * NA + t, k = tile >> 5, v0 = (tile & 31) * 8;
      const int cv = (arow_c + k) * 256 + v0 + arow_v;
      afrag[t] = kg < P ? reinterpret_cast<const u32x4*>(p.codebooks + (size_t)cv * G)[kg] : zero;
    }
  }
  // ---- row walk geometry: half-wave = one row, lane s of the half = dword s of the row's 128 code bytes of this slab: one
  // coalesced load per row.  Code words: raw buffer loads over the row range (rows past its end answer zeros and touch no
  // memory: all loads are unconditional); per step one 32-bit add forms the offset.  Groups past in_groups have ZERO table
  // entries (their x fragment is zero), so whatever code bytes a lane reads for them add nothing.
  const int s32 = lane & 31, half = lane >> 5;
  const int rfirst = wave * 2 + half;  // rows rfirst, rfirst + 2 NW, ...
  const int nsteps = (nrows + 2 * NW - 1) / (2 * NW);
  uint32_t cstride, coff;
  const uint8_t* cplane;
  if constexpr (PLANAR) {
    cstride = (uint32_t)p.jp;
    cplane = p.codes + ((size_t)pc * p.M + row_begin) * cstride;
    coff = (uint32_t)rfirst * cstride + (uint32_t)j0 + 4u * (uint32_t)s32;
  } else {
    cstride = (uint32_t)p.in_groups * 8u;
    cplane = p.codes + (size_t)row_begin * cstride;
    const int jl = s32 >> 1, ch = s32 & 1;
    const int jmine = j0 + jl < p.in_groups ? j0 + jl : p.in_groups - 1;
    coff = (uint32_t)rfirst * cstride + (uint32_t)jmine * 8u + (uint32_t)ch * 4u;
  }
  const __amdgpu_buffer_rsrc_t rs_codes = __builtin_amdgcn_make_buffer_rsrc((void*)cplane, 0, (uint32_t)nrows * cstride, 0x00020000);
  auto load_codes = [&](int step) -> uint32_t {
    return __builtin_amdgcn_raw_buffer_load_b32(rs_codes, coff + (uint32_t)step * ((uint32_t)(2 * NW) * cstride), 0, AUX_NT);
  };
  constexpr int RING = 8;  // steps of code words in flight per wave
  uint32_t cq[RING];
#pragma unroll
  for (int k = 0; k < RING; ++k) cq[k] = load_codes(k);
  LUT_TRACE(1);  // loads issued

  // ---- (2) max |x| while the codebook is on its way (15-bit magnitude patterns: integer order == magnitude order; NaN
  // sorts above Inf); canonical codes: max |codebook| from the A fragments -- the waves' fragments are the whole codebook
  lut_us2 mx = {0, 0}, mc = {0, 0};
  if (p.cells != nullptr) {
    lut_absmax(mx, xq0);
    lut_absmax(mx, xq1);
    for (int i = tid + 2 * NT; i < chunks; i += NT) lut_absmax(mx, reinterpret_cast<const u32x4*>(p.x)[i]);
  }

  // ---- (3) table on the matrix cores
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw != 0u) __builtin_trap();  // LDS map above starts at 0
  if constexpr (PLANAR) {
    // operands into LDS: the codebook as it lies in memory (row v at v * 2 G bytes); x with group jl = 4 s + k at unit k * 32 + s,
    // so that the 16 columns of a B fragment (one k, 16 consecutive slots) are 16 consecutive units
#pragma unroll
    for (int q = 0; q < CBP; ++q) {
      const int i = tid + q * NT;
      if (i < 256 * P) *reinterpret_cast<u32x4*>(smem_raw + LUT_OPER_OFF + i * 16) = stage_cb[q];
    }
#pragma unroll
    for (int q = 0; q < XP; ++q) {
      const int i = tid + q * NT, jl = i / P, piece = i % P;
      if (i < LUT_PJ * P) *reinterpret_cast<u32x4*>(smem_raw + LUT_XOPER_OFF + ((((jl & 3) * 32 + (jl >> 2)) * P + piece) * 16)) = stage_x[q];
    }
    __syncthreads();
    const int vgrp = wave & 3, cgrp = wave >> 2;
#pragma unroll
    for (int a = 0; a < NA; ++a)
      afrag[a] = kg < P ? *reinterpret_cast<const u32x4*>(smem_raw + LUT_OPER_OFF + (((vgrp * VT + a) * 16 + col) * P + kg) * 16) : zero;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int c = cgrp * CT + t, k = c & 3, s0 = (c >> 2) * 16;
      bfrag[t] = kg < P ? *reinterpret_cast<const u32x4*>(smem_raw + LUT_XOPER_OFF + (((k * 32 + s0 + col) * P + kg) * 16)) : zero;
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int v0 = (vgrp * VT + a) * 16 + kg * 4;  // this lane's first table row of the tile
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int c = cgrp * CT + t, k = c & 3, s0 = (c >> 2) * 16;
        const f32x4 d = lut_mfma16<T>(afrag[a], bfrag[t], f32x4{0.f, 0.f, 0.f, 0.f});
        float* const dst = reinterpret_cast<float*>(smem_raw + v0 * 256 + (k & 1) * 128 + (k >> 1) * 65536 + (s0 + col) * 4);
        dst[0] = d[0];
        dst[64] = d[1];
        dst[128] = d[2];
        dst[192] = d[3];
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      const f32x4 d = lut_mfma16<T>(afrag[t], bfrag[0], f32x4{0.f, 0.f, 0.f, 0.f});
      const int tile = wave * NA + t, k = tile >> 5, v0 = (tile & 31) * 8;
      unsigned char* const dst = smem_raw + (v0 + kg * 2) * 256 + (k & 1) * 128 + (k >> 1) * 65536 + col * 8;
      *reinterpret_cast<float2*>(dst) = float2{d[0], d[1]};
      *reinterpret_cast<float2*>(dst + 256) = float2{d[2], d[3]};
      if (p.cells != nullptr) lut_absmax(mc, afrag[t]);
    }
  }
  LUT_TRACE(2);  // table written
  if (p.cells != nullptr) {
    // one word per wave: max |codebook| pattern << 16 | max |x| pattern (v_pk_max_u16 keeps the halves apart)
    lut_us2 m = {(unsigned short)(mx.x > mx.y ? mx.x : mx.y), (unsigned short)(mc.x > mc.y ? mc.x : mc.y)};
    m = lut_row16_pkmax(m);
    const uint32_t mw = __builtin_bit_cast(uint32_t, m);
    const lut_us2 r01 = __builtin_elementwise_max(__builtin_bit_cast(lut_us2, (uint32_t)__builtin_amdgcn_readlane((int)mw, 0)),
                                                  __builtin_bit_cast(lut_us2, (uint32_t)__builtin_amdgcn_readlane((int)mw, 16)));
    const lut_us2 r23 = __builtin_elementwise_max(__builtin_bit_cast(lut_us2, (uint32_t)__builtin_amdgcn_readlane((int)mw, 32)),
                                                  __builtin_bit_cast(lut_us2, (uint32_t)__builtin_amdgcn_readlane((int)mw, 48)));
    if (lane == 0) reinterpret_cast<uint32_t*>(smem_raw + LUT_SLOTS_OFF)[wave] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(r01, r23));
  }
  LUT_TRACE(3);  // at the barrier
  __syncthreads();
  LUT_TRACE(4);  // table complete
  // fixed-point unit of the fused finalize: |slab sum| <= 128 (group, codebook) pairs x g x max|cb| x max|x| < 2^e; with
  // sh = 41 - e - ceil(log2(nslabs)) the nslabs addends of a row stay below 2^42 (the sum field is bits 63..20)
  int sh = 0;
  float bound = 0.f;
  if (p.cells != nullptr) {
    lut_us2 m = __builtin_bit_cast(lut_us2, reinterpret_cast<const uint32_t*>(smem_raw + LUT_SLOTS_OFF)[lane & (NW - 1)]);
    m = lut_row16_pkmax(m);  // NW <= 16 slots: every DPP row of 16 lanes has seen all of them
    const float cmax = PLANAR ? p.cb_absmax : T::to_float(m.y);
    bound = (float)(LUT_JS * LUT_KC * G) * cmax * T::to_float(m.x);
    int e = 0;
    (void)frexpf(bound, &e);
    sh = 41 - e - (32 - __builtin_clz((unsigned)(p.nslabs > 1 ? p.nslabs - 1 : 1)));
  }

  // ---- (4) rows.  Per step and half-wave: 4 v_perm_b32 (LDS address = {0, base byte 2, code byte, base byte 0}), 4
  // ds_read_b32, 3 adds.  The 32-lane sums of FOUR steps are then formed together: two butterfly stages (xor 1, xor 2) fold the
  // four partial vectors into one whose lane l carries step l & 3, two row rotations (4, 8) and one v_permlane16_swap finish
  // the sum over the half-wave -- 14 VALU ops per four rows instead of 4 x (5 dependent DPP adds + their wait states); lanes
  // 0..3 of each half put the four totals into the staging area in LDS.
  uint32_t base[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) base[k] = (uint32_t)((k & 1) * 128 + s32 * 4 + (k >> 1) * 65536);
  float* const stage = reinterpret_cast<float*>(smem_raw + LUT_STAGE_OFF);
  float* const out = p.partial + (size_t)slab * p.M + row_begin;
  const bool odd1 = (lane & 1) != 0, odd2 = (lane & 2) != 0;
  const bool writer = s32 < 4;
  const int wrow = rfirst + 2 * NW * (lane & 3);  // the row this lane would write for a quad starting at step 0
  // four steps, straight-line (ring slots S .. S + 3); steps past the end read zeros and are never handed in
  auto quad = [&](auto slot0, int sb, int row0) {
    constexpr int S = decltype(slot0)::value;
    float a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t cw = cq[S + t];
      cq[S + t] = load_codes(sb + t + RING);
      const float a0 = *(lds_f32_ptr)(uintptr_t)lut_perm(cw, base[0], 0x0c020400u);
      const float a1 = *(lds_f32_ptr)(uintptr_t)lut_perm(cw, base[1], 0x0c020500u);
      const float a2 = *(lds_f32_ptr)(uintptr_t)lut_perm(cw, base[2], 0x0c020600u);
      const float a3 = *(lds_f32_ptr)(uintptr_t)lut_perm(cw, base[3], 0x0c020700u);
      a[t] = (a0 + a1) + (a2 + a3);
    }
    const float m01 = (odd1 ? a[1] : a[0]) + dpp_f32<0xB1>(odd1 ? a[0] : a[1]);  // quad_perm [1,0,3,2]: lane parity p holds step p
    const float m23 = (odd1 ? a[3] : a[2]) + dpp_f32<0xB1>(odd1 ? a[2] : a[3]);  //                                  ... step 2 + p
    float u = (odd2 ? m23 : m01) + dpp_f32<0x4E>(odd2 ? m01 : m23);              // quad_perm [2,3,0,1]: lane l holds step l & 3
    u += dpp_f32<0x124>(u);  // row_ror:4
    u += dpp_f32<0x128>(u);  // row_ror:8 -- the sum over the lane's DPP row of 16
    const uint32_t ub = __builtin_bit_cast(uint32_t, u);
    const auto sw = __builtin_amdgcn_permlane16_swap(ub, ub, false, false);  // {rows 0 0 2 2, rows 1 1 3 3}
    u = __builtin_bit_cast(float, (uint32_t)sw[0]) + __builtin_bit_cast(float, (uint32_t)sw[1]);  // the half-wave's 32-lane sum
    const int r = wrow + 2 * NW * sb;
    if (writer && r < nrows) stage[r - row0] = u;
  };
  // Fused finalize: a row's slab sum is added to the row's cell as a fixed-point integer (bits 63..20; +1 in the arrival
  // counter, bits 9..0; +1 in bits 19..10 if the value is not finite) with ONE returning atomic -- integer adds commute, so the
  // total is independent of the arrival order -- and whoever finds nslabs - 1 earlier arrivals applies scale and bias, rounds
  // once, writes y and zeroes the cell.  The hand-in is a vector pass over the staged totals by the first rows / 64 waves (the
  // other waves skip its ~100 scalar-like instructions: with 4 waves per SIMD they were a third of the kernel's VALU time).
  unsigned long long pend_old = 0ull, pend_mine = 0ull;
  uint16_t pend_scale = 0, pend_bias = 0;  // requested with the atomic: behind the last-arrival test they would be a second round trip
  int pend_row = -1;
  const uint16_t* const bias_src = p.bias ? p.bias : p.scales;
  auto settle = [&]() {
    if (pend_row >= 0 && (pend_old & 1023ull) == (unsigned long long)(p.nslabs - 1)) {
      const unsigned long long cell = pend_old + pend_mine;
      float sv = (float)ldexp((double)((long long)cell >> 20), -sh);
      if ((cell >> 10) & 1023ull) sv = __builtin_nanf("");
      const float scale = T::to_float(pend_scale);
      const float bias = p.bias ? T::to_float(pend_bias) : 0.f;
      p.y[pend_row] = T::from_float(__builtin_fmaf(sv, scale, bias));
      __hip_atomic_store(p.cells + pend_row, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    pend_row = -1;
  };
  constexpr int CH_STEPS = LUT_STAGE_ROWS / (2 * NW);  // steps per staging pass (a multiple of 8)
  LUT_TRACE(5);  // walk starts
  for (int cb = 0; cb < nsteps; cb += CH_STEPS) {
    const int cend = cb + CH_STEPS < nsteps ? cb + CH_STEPS : nsteps;
    const int row0 = cb * 2 * NW;  // first row of this pass
    for (int sb = cb; sb < cend; sb += 8) {
      quad(std::integral_constant<int, 0>{}, sb, row0);
      if (sb + 4 >= cend) break;
      quad(std::integral_constant<int, 4>{}, sb + 4, row0);
    }
    LUT_TRACE(8);  // walked, totals staged
    __syncthreads();
    LUT_TRACE(9);  // every wave has walked
    const int cnt = (nrows - row0 < LUT_STAGE_ROWS ? nrows - row0 : LUT_STAGE_ROWS);
    for (int rr = tid; rr < cnt; rr += NT) {
      const float total = stage[rr];
      const int r = row0 + rr;
      if (p.cells == nullptr) {
        out[r] = total;
      } else {
        settle();  // this thread's previous row
        const bool finite = bound < __builtin_inff() && fabsf(total) <= 2.f * bound;  // false for NaN / Inf anywhere
        const long long q = finite ? __float2ll_rn(ldexpf(total, sh)) : 0ll;
        pend_mine = ((unsigned long long)q << 20) + (finite ? 1ull : 1025ull);
        pend_row = row_begin + r;
        pend_old = __hip_atomic_fetch_add(p.cells + pend_row, pend_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend_scale = p.scales[pend_row];
        pend_bias = bias_src[pend_row];
      }
    }
    if (cend < nsteps) __syncthreads();  // the staging area is refilled by the next pass
  }
  LUT_TRACE(6);  // rows handed in (atomics in flight)
  if (p.cells != nullptr) settle();
#ifdef AQLM_LUT_TRACE
  tr[7] = wall_clock64();
  if (lane == 0 && p.cells != nullptr) {
    unsigned long long* const out_tr = p.cells + ((p.M + 1023) & ~1023) + ((size_t)block * 16 + wave) * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) out_tr[i] = tr[i];
  }
#endif
}

// scalar arguments: preloaded into SGPRs at wave launch, no kernel-argument fetch at the head of the kernel
struct LutTail {  // what only the end of the kernel needs (not preloaded)
  unsigned long long* cells;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
  int jp;
  float cb_absmax;
};

template <class T, int G, int NW, bool PLANAR>
__global__ __launch_bounds__(NW * 64) void gemv_8x8_lut_kernel(const uint8_t* codes, const uint16_t* codebooks, const uint16_t* x,
                                                                float* partial, int M, int in_groups, int nslabs, int nranges,
                                                                int rows_per_range, const LutTail tail) {
  const LutParams p{codes, codebooks, x, partial, M, in_groups, nslabs, nranges, rows_per_range, tail.jp, tail.cb_absmax,
                    tail.cells, tail.scales, tail.bias, tail.y};
  gemv_8x8_lut_body<T, G, NW, PLANAR>(p, blockIdx.x);
}

// 2..AQLM_HIP_MAX_GEMV_BATCH input rows in one launch (round 5; single-kernel form only): blockIdx.y = the row of x.  Every row
// is its own set of workgroups -- the tables are functions of x, so nothing but the codes (L2 hits for the later rows) and the
// launch itself is shared -- with its own cells (cells + row * M) and its own output row: B rows cost one launch boundary
// instead of B (the reference serves any row count of this scheme through one kernel in a per-row loop, triton_kernel.py:161-182).
// A row's bits equal those of the same row launched alone.
struct LutTailRows {
  LutTail t;
  long y_row_stride;
};

template <class T, int G, int NW, bool PLANAR>
__global__ __launch_bounds__(NW * 64) void gemv_8x8_lut_rows_kernel(const uint8_t* codes, const uint16_t* codebooks, const uint16_t* x,
                                                                     float* partial, int M, int in_groups, int nslabs, int nranges,
                                                                     int rows_per_range, int x_row_stride, const LutTailRows tail) {
  const int row = (int)blockIdx.y;
  const LutParams p{codes, codebooks, x + (size_t)row * (size_t)x_row_stride, partial, M, in_groups, nslabs, nranges, rows_per_range, tail.t.jp,
                    tail.t.cb_absmax, tail.t.cells + (size_t)row * (size_t)M, tail.t.scales, tail.t.bias, tail.t.y + (size_t)row * tail.y_row_stride};
  gemv_8x8_lut_body<T, G, NW, PLANAR>(p, blockIdx.x);
}

// shared-input launch: up to AQLM_HIP_MAX_SEGMENTS layers (own codes / codebooks / partials) times one x
struct LutSegment {
  const uint8_t* codes;
  const uint16_t* codebooks;
  float* partial;
  int M, nranges, rows_per_range, block_begin;
  float cb_absmax;
  unsigned long long* cells;  // fused finalize (nullptr: partials + finalize kernel)
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
};

struct LutMultiParams {
  const uint16_t* x;
  int in_groups, nslabs, nseg, jp;
  LutSegment seg[AQLM_HIP_MAX_SEGMENTS];
};

template <class T, int G, int NW, bool PLANAR>
__global__ __launch_bounds__(NW * 64) void gemv_8x8_lut_multi_kernel(const LutMultiParams mp) {
  LutParams p{};
  p.x = mp.x;
  p.in_groups = mp.in_groups;
  p.nslabs = mp.nslabs;
  p.jp = mp.jp;
  int begin = 0;
#pragma unroll
  for (int k = 0; k < AQLM_HIP_MAX_SEGMENTS; ++k) {
    if (k == 0 || (k < mp.nseg && (int)blockIdx.x >= mp.seg[k].block_begin)) {  // scalar select chain
      p.codes = mp.seg[k].codes;
      p.codebooks = mp.seg[k].codebooks;
      p.partial = mp.seg[k].partial;
      p.M = mp.seg[k].M;
      p.nranges = mp.seg[k].nranges;
      p.rows_per_range = mp.seg[k].rows_per_range;
      p.cb_absmax = mp.seg[k].cb_absmax;
      p.cells = mp.seg[k].cells;
      p.scales = mp.seg[k].scales;
      p.bias = mp.seg[k].bias;
      p.y = mp.seg[k].y;
      begin = mp.seg[k].block_begin;
    }
  }
  gemv_8x8_lut_body<T, G, NW, PLANAR>(p, (int)blockIdx.x - begin);
}

// ---- planar code layout: [8 codebooks][M rows][jp = in_groups rounded up to 4] bytes -- the same bytes as the checkpoint's
// [M][in_groups][8], transposed so that a (codebook, 128-group) workgroup reads its codes as 128 contiguous bytes per row.
// (The reference re-lays its codes out at load time for its CPU look-up kernel too: inference.py:78-83.)
__global__ __launch_bounds__(256) void lut_planar_pack_kernel(const uint8_t* codes, uint8_t* planar, int M, int in_groups, int jp) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int q4 = jp >> 2;
  if (idx >= (long)M * q4) return;
  const int i = (int)(idx / q4), q = (int)(idx % q4);
  uint32_t w[4][2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = 4 * q + k;
    u32x2 v = {0u, 0u};
    if (j < in_groups) v = *reinterpret_cast<const u32x2*>(codes + ((size_t)i * in_groups + j) * 8);
    w[k][0] = v.x;
    w[k][1] = v.y;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) o |= ((w[k][c >> 2] >> ((c & 3) * 8)) & 0xffu) << (8 * k);
    *reinterpret_cast<uint32_t*>(planar + ((size_t)c * M + i) * jp + 4 * q) = o;
  }
}

__global__ __launch_bounds__(256) void lut_planar_unpack_kernel(const uint8_t* planar, uint8_t* codes, int M, int in_groups, int jp) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int q4 = jp >> 2;
  if (idx >= (long)M * q4) return;
  const int i = (int)(idx / q4), q = (int)(idx % q4);
  uint32_t o[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = *reinterpret_cast<const uint32_t*>(planar + ((size_t)c * M + i) * jp + 4 * q);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = 4 * q + k;
    if (j >= in_groups) continue;
    u32x2 v = {0u, 0u};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint32_t b = (o[c] >> (8 * k)) & 0xffu;
      if (c < 4) v.x |= b << (8 * c);
      else v.y |= b << (8 * (c - 4));
    }
    *reinterpret_cast<u32x2*>(codes + ((size_t)i * in_groups + j) * 8) = v;
  }
}

struct LutFinalizeParams {
  const float* partial;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
  int M, nslabs;
};

template <class T>
__global__ __launch_bounds__(256) void gemv_8x8_lut_finalize(const float* partial, const uint16_t* scales, const uint16_t* bias_ptr,
                                                             uint16_t* y, int M, int nslabs) {
  const LutFinalizeParams p{partial, scales, bias_ptr, y, M, nslabs};
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= p.M) return;
  float s = 0.f;
  for (int k = 0; k < p.nslabs; ++k) s += p.partial[(size_t)k * p.M + row];
  const float scale = T::to_float(p.scales[row]);
  const float bias = p.bias ? T::to_float(p.bias[row]) : 0.f;
  p.y[row] = T::from_float(__builtin_fmaf(s, scale, bias));
}

struct LutFinalizeSegment {
  LutFinalizeParams f;
  int block_begin;
};

struct LutFinalizeMultiParams {
  int nseg;
  LutFinalizeSegment seg[AQLM_HIP_MAX_SEGMENTS];
};

template <class T>
__global__ __launch_bounds__(256) void gemv_8x8_lut_finalize_multi(const LutFinalizeMultiParams mp) {
  LutFinalizeParams p = mp.seg[0].f;
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
  for (int k = 0; k < p.nslabs; ++k) s += p.partial[(size_t)k * p.M + row];
  const float scale = T::to_float(p.scales[row]);
  const float bias = p.bias ? T::to_float(p.bias[row]) : 0.f;
  p.y[row] = T::from_float(__builtin_fmaf(s, scale, bias));
}

static int lut_nslabs(int in_groups, bool planar) {
  return planar ? 8 * ((in_groups + LUT_PJ - 1) / LUT_PJ) : (in_groups + LUT_JS - 1) / LUT_JS;
}

size_t gemv_8x8_lut_workspace(int out_features, int in_features, int in_group_size) {
  const int in_groups = in_features / in_group_size;
  // the larger of the two code layouts' slab counts (the planar entries take the same workspace)
  const int nslabs = std::max(lut_nslabs(in_groups, false), lut_nslabs(in_groups, true));
  return (size_t)nslabs * out_features * sizeof(float);
}

// waves per workgroup: every wave runs the kernel's scalar-like parts (addresses, the fixed-point unit), so fewer, fatter waves
// do less of it; tuning key "lut_waves" (8 / 16) overrides
static int lut_waves() {
  const int w = tuning().lut_waves;
  return w == 8 || w == 16 ? w : 16;
}

template <class T, int G, int NW, bool PLANAR>
static int launch_lut(const LutParams& p, hipStream_t stream, int batch = 1, long x_row_stride = 0, long y_row_stride = 0) {
  if (batch > 1) {
    auto kern = gemv_8x8_lut_rows_kernel<T, G, NW, PLANAR>;
    if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LUT_LDS_BYTES)) return e;
    const LutTailRows tail{{p.cells, p.scales, p.bias, p.y, p.jp, p.cb_absmax}, y_row_stride};
    hipLaunchKernelGGL(kern, dim3(p.nslabs * p.nranges, batch), dim3(NW * 64), LUT_LDS_BYTES, stream, p.codes, p.codebooks, p.x, p.partial,
                       p.M, p.in_groups, p.nslabs, p.nranges, p.rows_per_range, (int)x_row_stride, tail);
    return check_hip(hipGetLastError(), "gemv_8x8_lut (rows) launch");
  }
  auto kern = gemv_8x8_lut_kernel<T, G, NW, PLANAR>;
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LUT_LDS_BYTES)) return e;
  const LutTail tail{p.cells, p.scales, p.bias, p.y, p.jp, p.cb_absmax};
  hipLaunchKernelGGL(kern, dim3(p.nslabs * p.nranges), dim3(NW * 64), LUT_LDS_BYTES, stream, p.codes, p.codebooks, p.x, p.partial, p.M,
                     p.in_groups, p.nslabs, p.nranges, p.rows_per_range, tail);
  return check_hip(hipGetLastError(), "gemv_8x8_lut launch");
}

template <class T, int G, bool PLANAR>
static int launch_lut_w(const LutParams& p, hipStream_t stream, int batch, long xs, long ys) {
  return lut_waves() == 8 ? launch_lut<T, G, 8, PLANAR>(p, stream, batch, xs, ys) : launch_lut<T, G, 16, PLANAR>(p, stream, batch, xs, ys);
}

template <bool PLANAR>
static int launch_lut_any(const LutParams& p, int G, int dtype, hipStream_t stream, int batch = 1, long xs = 0, long ys = 0) {
  if (dtype == AQLM_HIP_F16)
    return G == 8 ? launch_lut_w<F16, 8, PLANAR>(p, stream, batch, xs, ys) : G == 16 ? launch_lut_w<F16, 16, PLANAR>(p, stream, batch, xs, ys) : launch_lut_w<F16, 32, PLANAR>(p, stream, batch, xs, ys);
  return G == 8 ? launch_lut_w<BF16, 8, PLANAR>(p, stream, batch, xs, ys) : G == 16 ? launch_lut_w<BF16, 16, PLANAR>(p, stream, batch, xs, ys) : launch_lut_w<BF16, 32, PLANAR>(p, stream, batch, xs, ys);
}

// batch-1 8x8 matvec through LDS look-up tables; AQLM_HIP_E_UNSUPPORTED when the shape does not fit
// `fused`: workspace = out_features zero-at-rest 64-bit cells (one kernel); else fp32 slab partials + a finalize kernel
// `planar`: codes in the planar layout (aqlm_hip_8x8_planar_pack); fused then needs cb_absmax > 0
int gemv_8x8_lut(const void* codes, const void* codebooks, const void* scales, const void* bias, const void* x, void* y,
                 int out_features, int in_features, int in_group_size, int dtype, void* workspace, size_t workspace_bytes,
                 hipStream_t stream, bool fused, bool planar, float cb_absmax, int batch = 1, long x_row_stride = 0, long y_row_stride = 0) {
  const int G = in_group_size;
  if (G != 8 && G != 16 && G != 32) return AQLM_HIP_E_UNSUPPORTED;
  if (batch < 1 || batch > AQLM_HIP_MAX_GEMV_BATCH || (batch > 1 && (!fused || x_row_stride % 8 != 0 || x_row_stride > 0x7fffffffL))) return AQLM_HIP_E_INVALID;
  LutParams p{};
  p.codes = (const uint8_t*)codes;
  p.codebooks = (const uint16_t*)codebooks;
  p.x = (const uint16_t*)x;
  p.partial = (float*)workspace;
  p.M = out_features;
  p.in_groups = in_features / G;
  p.nslabs = lut_nslabs(p.in_groups, planar);
  p.jp = (p.in_groups + 3) & ~3;
  p.cb_absmax = cb_absmax;
  if (p.nslabs > 1023) return AQLM_HIP_E_UNSUPPORTED;  // (arrival counter of the fused finalize: 10 bits)
  if (fused) {
    if (!workspace || workspace_bytes < (size_t)batch * out_features * 8 || ((uintptr_t)workspace & 7)) return AQLM_HIP_E_INVALID;
    if (planar && !(cb_absmax > 0.f)) return AQLM_HIP_E_INVALID;
    p.cells = (unsigned long long*)workspace;
    p.scales = (const uint16_t*)scales;
    p.bias = (const uint16_t*)bias;
    p.y = (uint16_t*)y;
  } else if (workspace_bytes < (size_t)p.nslabs * out_features * sizeof(float) || !workspace) {
    return AQLM_HIP_E_INVALID;
  }
  p.nranges = std::max(1, 256 / p.nslabs);
  p.rows_per_range = (out_features + p.nranges - 1) / p.nranges;
  const int e = planar ? launch_lut_any<true>(p, G, dtype, stream, batch, x_row_stride, y_row_stride)
                       : launch_lut_any<false>(p, G, dtype, stream, batch, x_row_stride, y_row_stride);
  if (e || fused) return e;
  LutFinalizeParams f{};
  f.partial = (const float*)workspace;
  f.scales = (const uint16_t*)scales;
  f.bias = (const uint16_t*)bias;
  f.y = (uint16_t*)y;
  f.M = out_features;
  f.nslabs = p.nslabs;
  if (dtype == AQLM_HIP_F16)
    hipLaunchKernelGGL(gemv_8x8_lut_finalize<F16>, dim3((out_features + 255) / 256), dim3(256), 0, stream, f.partial, f.scales, f.bias,
                       f.y, f.M, f.nslabs);
  else
    hipLaunchKernelGGL(gemv_8x8_lut_finalize<BF16>, dim3((out_features + 255) / 256), dim3(256), 0, stream, f.partial, f.scales, f.bias,
                       f.y, f.M, f.nslabs);
  return check_hip(hipGetLastError(), "gemv_8x8_lut_finalize launch");
}

template <class T, int G, int NW, bool PLANAR>
static int launch_lut_multi(const LutMultiParams& mp, int blocks, hipStream_t stream) {
  auto kern = gemv_8x8_lut_multi_kernel<T, G, NW, PLANAR>;
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LUT_LDS_BYTES)) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), LUT_LDS_BYTES, stream, mp);
  return check_hip(hipGetLastError(), "gemv_8x8_lut_multi launch");
}

template <class T, int G, bool PLANAR>
static int launch_lut_multi_w(const LutMultiParams& mp, int blocks, hipStream_t stream) {
  return lut_waves() == 8 ? launch_lut_multi<T, G, 8, PLANAR>(mp, blocks, stream) : launch_lut_multi<T, G, 16, PLANAR>(mp, blocks, stream);
}

template <bool PLANAR>
static int launch_lut_multi_any(const LutMultiParams& mp, int blocks, int G, int dtype, hipStream_t stream) {
  if (dtype == AQLM_HIP_F16)
    return G == 8 ? launch_lut_multi_w<F16, 8, PLANAR>(mp, blocks, stream) : G == 16 ? launch_lut_multi_w<F16, 16, PLANAR>(mp, blocks, stream)
                                                                                      : launch_lut_multi_w<F16, 32, PLANAR>(mp, blocks, stream);
  return G == 8 ? launch_lut_multi_w<BF16, 8, PLANAR>(mp, blocks, stream) : G == 16 ? launch_lut_multi_w<BF16, 16, PLANAR>(mp, blocks, stream)
                                                                                     : launch_lut_multi_w<BF16, 32, PLANAR>(mp, blocks, stream);
}

// Shared-input variant: the ~256 workgroups are dealt to the segments in proportion to their rows.  A row's value does not
// depend on its range (same table, same per-row summation order), so results equal gemv_8x8_lut per segment bit for bit.
// workspace: sum over segments of gemv_8x8_lut_workspace(...).
int gemv_8x8_lut_multi(const aqlm_hip_segment* segments, int num_segments, const void* x, int in_features,
                       int in_group_size, int dtype, void* workspace, size_t workspace_bytes, hipStream_t stream, bool fused,
                       bool planar, const float* cb_absmax) {
  const int G = in_group_size;
  if (G != 8 && G != 16 && G != 32) return AQLM_HIP_E_UNSUPPORTED;
  LutMultiParams mp{};
  LutFinalizeMultiParams fm{};
  mp.x = (const uint16_t*)x;
  mp.in_groups = in_features / G;
  mp.nslabs = lut_nslabs(mp.in_groups, planar);
  mp.jp = (mp.in_groups + 3) & ~3;
  mp.nseg = fm.nseg = num_segments;
  long total = 0;
  for (int k = 0; k < num_segments; ++k) total += segments[k].out_features;
  const int total_ranges = std::max(num_segments, 256 / mp.nslabs);
  size_t need = 0;
  int blocks = 0, fblocks = 0;
  for (int k = 0; k < num_segments; ++k) {
    const aqlm_hip_segment& sg = segments[k];
    LutSegment& ls = mp.seg[k];
    ls.codes = (const uint8_t*)sg.codes;
    ls.codebooks = (const uint16_t*)sg.codebook;
    ls.partial = (float*)((uint8_t*)workspace + need);
    ls.cb_absmax = cb_absmax ? cb_absmax[k] : 0.f;
    if (fused) {  // the segment's cells, in segment order
      if (planar && !(ls.cb_absmax > 0.f)) return AQLM_HIP_E_INVALID;
      ls.cells = (unsigned long long*)((uint8_t*)workspace + need);
      ls.scales = (const uint16_t*)sg.scales;
      ls.bias = (const uint16_t*)sg.bias;
      ls.y = (uint16_t*)sg.y;
    }
    ls.M = sg.out_features;
    ls.nranges = std::max(1, (int)(((long)total_ranges * sg.out_features + total / 2) / total));
    ls.rows_per_range = (sg.out_features + ls.nranges - 1) / ls.nranges;
    ls.nranges = (sg.out_features + ls.rows_per_range - 1) / ls.rows_per_range;
    ls.block_begin = blocks;
    blocks += mp.nslabs * ls.nranges;
    LutFinalizeSegment& fs = fm.seg[k];
    fs.f.partial = ls.partial;
    fs.f.scales = (const uint16_t*)sg.scales;
    fs.f.bias = (const uint16_t*)sg.bias;
    fs.f.y = (uint16_t*)sg.y;
    fs.f.M = sg.out_features;
    fs.f.nslabs = mp.nslabs;
    fs.block_begin = fblocks;
    fblocks += (sg.out_features + 255) / 256;
    need += fused ? (size_t)sg.out_features * 8 : (size_t)mp.nslabs * sg.out_features * sizeof(float);
  }
  if (!workspace || workspace_bytes < need || mp.nslabs > 1023 || (fused && ((uintptr_t)workspace & 7))) return AQLM_HIP_E_INVALID;
  const int e = planar ? launch_lut_multi_any<true>(mp, blocks, G, dtype, stream) : launch_lut_multi_any<false>(mp, blocks, G, dtype, stream);
  if (e || fused) return e;
  if (dtype == AQLM_HIP_F16) hipLaunchKernelGGL(gemv_8x8_lut_finalize_multi<F16>, dim3(fblocks), dim3(256), 0, stream, fm);
  else hipLaunchKernelGGL(gemv_8x8_lut_finalize_multi<BF16>, dim3(fblocks), dim3(256), 0, stream, fm);
  return check_hip(hipGetLastError(), "gemv_8x8_lut_finalize_multi launch");
}

}  // namespace aqlm

using namespace aqlm;

static int lut_multi_entry(const char* name, const aqlm_hip_segment* segments, int num_segments, const void* x, int in_features,
                           int in_group_size, int dtype, void* workspace, size_t workspace_bytes, void* stream, bool fused, bool planar,
                           const float* cb_absmax) {
  if (!segments || num_segments < 1 || num_segments > AQLM_HIP_MAX_SEGMENTS || !x) {
    set_last_error("%s: 1..%d segments and a non-null x required (got %d)", name, AQLM_HIP_MAX_SEGMENTS, num_segments);
    return AQLM_HIP_E_INVALID;
  }
  if (in_features <= 0 || in_group_size <= 0 || in_features % in_group_size != 0) {
    set_last_error("%s: bad sizes in=%d g=%d", name, in_features, in_group_size);
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("%s: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", name, dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  for (int k = 0; k < num_segments; ++k) {
    const aqlm_hip_segment& sg = segments[k];
    if (!sg.codes || !sg.codebook || !sg.scales || !sg.y || sg.out_features <= 0) {
      set_last_error("%s: null pointer or non-positive size in segment %d", name, k);
      return AQLM_HIP_E_INVALID;
    }
    if (!aligned16(sg.codebook) || (reinterpret_cast<uintptr_t>(sg.codes) & 7u)) {
      set_last_error("%s: misaligned buffer in segment %d", name, k);
      return AQLM_HIP_E_UNSUPPORTED;
    }
  }
  if (!aligned16(x)) {
    set_last_error("%s: misaligned x", name);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const int e = gemv_8x8_lut_multi(segments, num_segments, x, in_features, in_group_size, dtype, workspace, workspace_bytes,
                                   (hipStream_t)stream, fused, planar, cb_absmax);
  if (e == AQLM_HIP_E_UNSUPPORTED) set_last_error("%s: in_group_size %d not in {8,16,32}", name, in_group_size);
  if (e == AQLM_HIP_E_INVALID)
    set_last_error("%s: workspace too small (sum of aqlm_hip_workspace_bytes(AQLM_HIP_OP_GEMV_8X8_LUT, ...) over the segments), cells "
                   "misaligned, or a planar fused call without a positive codebook_absmax", name);
  return e;
}

extern "C" int aqlm_hip_gemv_8x8_lut_multi(const aqlm_hip_segment* segments, int num_segments, const void* x,
                                           int in_features, int in_group_size, int dtype, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  return lut_multi_entry("aqlm_hip_gemv_8x8_lut_multi", segments, num_segments, x, in_features, in_group_size, dtype, workspace,
                         workspace_bytes, stream, false, false, nullptr);
}

extern "C" int aqlm_hip_gemv_8x8_lut_multi_fused(const aqlm_hip_segment* segments, int num_segments, const void* x,
                                                 int in_features, int in_group_size, int dtype, void* cells,
                                                 size_t cells_bytes, void* stream) {
  return lut_multi_entry("aqlm_hip_gemv_8x8_lut_multi_fused", segments, num_segments, x, in_features, in_group_size, dtype, cells,
                         cells_bytes, stream, true, false, nullptr);
}

extern "C" int aqlm_hip_gemv_8x8_lut_planar_multi(const aqlm_hip_segment* segments, const float* codebook_absmax, int num_segments,
                                                  const void* x, int in_features, int in_group_size, int dtype, void* workspace,
                                                  size_t workspace_bytes, int fused, void* stream) {
  if (fused && !codebook_absmax) {
    set_last_error("aqlm_hip_gemv_8x8_lut_planar_multi: the single-kernel form needs codebook_absmax per segment");
    return AQLM_HIP_E_INVALID;
  }
  return lut_multi_entry("aqlm_hip_gemv_8x8_lut_planar_multi", segments, num_segments, x, in_features, in_group_size, dtype, workspace,
                         workspace_bytes, stream, fused != 0, true, codebook_absmax);
}

static int lut_entry(const char* name, const void* codes, const void* codebooks, const void* scales, const void* bias, const void* x,
                     void* y, int out_features, int in_features, int in_group_size, int dtype, void* workspace, size_t workspace_bytes,
                     void* stream, bool fused, bool planar, float cb_absmax, int batch = 1, long x_row_stride = 0, long y_row_stride = 0) {
  if (!codes || !codebooks || !scales || !x || !y) {
    set_last_error("%s: null pointer argument", name);
    return AQLM_HIP_E_INVALID;
  }
  if (out_features <= 0 || in_features <= 0 || in_group_size <= 0 || in_features % in_group_size != 0) {
    set_last_error("%s: bad sizes out=%d in=%d g=%d", name, out_features, in_features, in_group_size);
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("%s: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", name, dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (!aligned16(codebooks) || !aligned16(x) || (reinterpret_cast<uintptr_t>(codes) & 7u)) {
    set_last_error("%s: misaligned buffer", name);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const int e = gemv_8x8_lut(codes, codebooks, scales, bias, x, y, out_features, in_features, in_group_size, dtype,
                             workspace, workspace_bytes, (hipStream_t)stream, fused, planar, cb_absmax, batch, x_row_stride, y_row_stride);
  if (e == AQLM_HIP_E_UNSUPPORTED) set_last_error("%s: in_group_size %d not in {8,16,32}", name, in_group_size);
  if (e == AQLM_HIP_E_INVALID)
    set_last_error("%s: workspace / cells too small (batch x out_features x 8 bytes), null or misaligned, a planar fused call without a "
                   "positive codebook_absmax, or batch outside 1..%d / x rows not 16-B aligned", name, AQLM_HIP_MAX_GEMV_BATCH);
  return e;
}

extern "C" int aqlm_hip_gemv_8x8_lut_batch(const void* codes, const void* codebooks, const void* scales, const void* bias, const void* x,
                                           void* y, int out_features, int in_features, int in_group_size, int batch, long x_row_stride,
                                           long y_row_stride, int dtype, int planar, float codebook_absmax, void* cells,
                                           size_t cells_bytes, void* stream) {
  return lut_entry("aqlm_hip_gemv_8x8_lut_batch", codes, codebooks, scales, bias, x, y, out_features, in_features, in_group_size, dtype,
                   cells, cells_bytes, stream, true, planar != 0, codebook_absmax, batch, x_row_stride, y_row_stride);
}

extern "C" int aqlm_hip_gemv_8x8_lut(const void* codes, const void* codebooks, const void* scales, const void* bias,
                                     const void* x, void* y, int out_features, int in_features, int in_group_size,
                                     int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  return lut_entry("aqlm_hip_gemv_8x8_lut", codes, codebooks, scales, bias, x, y, out_features, in_features, in_group_size, dtype,
                   workspace, workspace_bytes, stream, false, false, 0.f);
}

extern "C" int aqlm_hip_gemv_8x8_lut_fused(const void* codes, const void* codebooks, const void* scales, const void* bias,
                                           const void* x, void* y, int out_features, int in_features, int in_group_size,
                                           int dtype, void* cells, size_t cells_bytes, void* stream) {
  return lut_entry("aqlm_hip_gemv_8x8_lut_fused", codes, codebooks, scales, bias, x, y, out_features, in_features, in_group_size, dtype,
                   cells, cells_bytes, stream, true, false, 0.f);
}

extern "C" int aqlm_hip_gemv_8x8_lut_planar(const void* planar, const void* codebooks, const void* scales, const void* bias,
                                            const void* x, void* y, int out_features, int in_features, int in_group_size, int dtype,
                                            float codebook_absmax, void* workspace, size_t workspace_bytes, int fused, void* stream) {
  return lut_entry("aqlm_hip_gemv_8x8_lut_planar", planar, codebooks, scales, bias, x, y, out_features, in_features, in_group_size,
                   dtype, workspace, workspace_bytes, stream, fused != 0, true, codebook_absmax);
}

extern "C" size_t aqlm_hip_8x8_planar_bytes(int out_features, int in_features, int in_group_size) {
  if (out_features <= 0 || in_features <= 0 || in_group_size <= 0 || in_features % in_group_size) return 0;
  const size_t jp = (size_t)((in_features / in_group_size + 3) & ~3);
  return 8 * (size_t)out_features * jp;
}

static int planar_args(const char* name, const void* a, const void* b, int out_features, int in_features, int in_group_size) {
  if (!a || !b || out_features <= 0 || in_features <= 0 || in_group_size <= 0 || in_features % in_group_size) {
    set_last_error("%s: null pointer or bad sizes out=%d in=%d g=%d", name, out_features, in_features, in_group_size);
    return AQLM_HIP_E_INVALID;
  }
  if ((reinterpret_cast<uintptr_t>(a) & 7u) || (reinterpret_cast<uintptr_t>(b) & 7u)) {
    set_last_error("%s: buffers must be 8-byte aligned", name);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  return 0;
}

extern "C" int aqlm_hip_8x8_planar_pack(const void* codes_i8, int out_features, int in_features, int in_group_size, void* planar,
                                        size_t planar_bytes, void* stream) {
  if (int e = planar_args("aqlm_hip_8x8_planar_pack", codes_i8, planar, out_features, in_features, in_group_size)) return e;
  if (planar_bytes < aqlm_hip_8x8_planar_bytes(out_features, in_features, in_group_size)) {
    set_last_error("aqlm_hip_8x8_planar_pack: buffer of %zu bytes, aqlm_hip_8x8_planar_bytes() asks for %zu", planar_bytes,
                   aqlm_hip_8x8_planar_bytes(out_features, in_features, in_group_size));
    return AQLM_HIP_E_INVALID;
  }
  const int in_groups = in_features / in_group_size, jp = (in_groups + 3) & ~3;
  const long n = (long)out_features * (jp >> 2);
  hipLaunchKernelGGL(lut_planar_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)codes_i8,
                     (uint8_t*)planar, out_features, in_groups, jp);
  return check_hip(hipGetLastError(), "aqlm_hip_8x8_planar_pack launch");
}

extern "C" int aqlm_hip_8x8_planar_unpack(const void* planar, int out_features, int in_features, int in_group_size, void* codes_i8,
                                          void* stream) {
  if (int e = planar_args("aqlm_hip_8x8_planar_unpack", planar, codes_i8, out_features, in_features, in_group_size)) return e;
  const int in_groups = in_features / in_group_size, jp = (in_groups + 3) & ~3;
  const long n = (long)out_features * (jp >> 2);
  hipLaunchKernelGGL(lut_planar_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)planar,
                     (uint8_t*)codes_i8, out_features, in_groups, jp);
  return check_hip(hipGetLastError(), "aqlm_hip_8x8_planar_unpack launch");
}
