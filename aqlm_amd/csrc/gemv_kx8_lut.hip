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
constexpr int LUT_JS = 16;                       // input groups per slab
constexpr int LUT_ENTRIES = LUT_JS * LUT_KC * 256;  // 32768 fp32 = 128 KiB

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
  const uint8_t* codes;      // [M][in_groups][8]
  const uint16_t* codebooks; // [8][256][G]
  const uint16_t* x;
  float* partial;            // [nslabs][M]
  int M, in_groups, nslabs, nranges, rows_per_range;
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

// LDS image of a slab (round 4).  Entry (group jl, codebook c = 4 ch + k, value v) lives at byte
//     v * 256 + (k & 1) * 128 + (jl * 2 + ch) * 4 + (k >> 1) * 65536 :
// the bank of an entry is a function of (jl, ch) alone, and (jl, ch) is the LANE of the row walk (32 lanes = the 128 code
// bytes of one row's slab chunk, one dword = 4 codebooks each), so a wave-wide ds_read_b32 never has a bank conflict
// whatever the codes are.  Round 3's image was lut[jl][c][v]: bank = v mod 32 -- random for the reads, and 16 equal banks
// for the 16 lanes of every table write (79 % of the LDS cycles were conflict cycles, profiles/r03_8x8_lut_kernel_pmc.json).
__device__ __forceinline__ uint32_t lut_perm(uint32_t cw, uint32_t base, uint32_t sel) { return __builtin_amdgcn_perm(cw, base, sel); }

// row_shr:1 with the old value kept where the source lane does not exist: lane 0 of every DPP row takes `in`, the
// others take their left neighbour's `chain` -- a 16-deep shift register per row in ONE VALU op
__device__ __forceinline__ float lut_shift_in(float in, float chain) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, in), __builtin_bit_cast(int, chain),
                                                                0x111, 0xf, 0xf, false));
}

// `block` = the workgroup's index within its own layer (== blockIdx.x for a single-layer launch)
template <class T, int G>
__device__ __forceinline__ void gemv_8x8_lut_body(const LutParams& p, const int block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef AQLM_LUT_TRACE  // profiling builds only (tools/microbench `make trace`): wall-clock stamps (100 MHz) per wave behind the cells
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define LUT_TRACE(i) do { __builtin_amdgcn_sched_barrier(0); tr[i] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LUT_TRACE(i)
#endif
  LUT_TRACE(0);
  const int slab = block % p.nslabs, range = block / p.nslabs;
  const int j0 = slab * LUT_JS;
  const int row_begin = range * p.rows_per_range;
  int nrows = p.M - row_begin;
  nrows = nrows < 0 ? 0 : (nrows < p.rows_per_range ? nrows : p.rows_per_range);

  // ---- row walk geometry: half-wave = one row, lane s of the half = (group jl = s / 2, codebooks 4 ch .. 4 ch + 3) =
  // dword s of the row's 128-byte slab chunk: the 32 lanes read it with one coalesced load
  const int s32 = lane & 31, half = lane >> 5;
  const int jl = s32 >> 1, ch = s32 & 1;
  const bool group_ok = j0 + jl < p.in_groups;
  const int jmine = group_ok ? j0 + jl : p.in_groups - 1;
  const int rfirst = wave * 2 + half;  // rows rfirst, rfirst + 32, ...
  const int nsteps = (nrows + 31) >> 5;
  // code words: raw buffer loads over the row range (rows past its end answer zeros and touch no memory: all loads are
  // unconditional); per step one 32-bit add forms the offset
  const uint32_t cstride = (uint32_t)p.in_groups * 8u;
  const __amdgpu_buffer_rsrc_t rs_codes = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.codes + (size_t)row_begin * cstride), 0, (uint32_t)nrows * cstride, 0x00020000);
  const uint32_t coff = (uint32_t)rfirst * cstride + (uint32_t)jmine * 8u + (uint32_t)ch * 4u;
  auto load_codes = [&](int step) -> uint32_t {
    return __builtin_amdgcn_raw_buffer_load_b32(rs_codes, coff + (uint32_t)step * (32u * cstride), 0, AUX_NT);
  };
  constexpr int RING = 8;  // steps of code words in flight per wave
  uint32_t cq[RING];
#pragma unroll
  for (int k = 0; k < RING; ++k) cq[k] = load_codes(k);

  // ---- table on the matrix cores: lut[(c, v)][jl] = sum_k cb[c][v][k] * x[j0 + jl][k] is a [2048 x g] x [g x 16] product.
  // v_mfma_f32_16x16x32: A = 16 codebook rows x 32 k (lane l: row l % 16, 8 k of piece l / 16), B = x of the 16 groups
  // (lane l: group l % 16, same piece), D[row (l / 16) * 4 + r][group l % 16].  A tile's 16 rows are
  // row q * 4 + vs * 2 + ch  =  codebook 4 ch + k, value v0 + 2 q + vs   (k, v0 per tile: 4 x 32 tiles),
  // so the 4 D registers of lane l are (v, ch = 0 / 1) and (v + 1, ch = 0 / 1) of group l % 16: two 8-byte LDS writes
  // whose 16 lanes cover 128 contiguous bytes (conflict-free).  g < 32 pads k with zero pieces.  8 tiles per wave.
  {
    constexpr int P = G / 8;  // pieces of 8 k per vector: 1, 2 or 4
    const int col = lane & 15, kg = lane >> 4;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const int jb = j0 + col < p.in_groups ? j0 + col : p.in_groups - 1;
    const u32x4 bfrag = kg < P ? reinterpret_cast<const u32x4*>(p.x + (size_t)jb * G)[kg] : zero;
    const int arow_c = (col & 1) * 4, arow_v = (col >> 2) * 2 + ((col >> 1) & 1);  // this lane's A row within a tile
    u32x4 afrag[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int tile = wave * 8 + t, k = tile >> 5, v0 = (tile & 31) * 8;
      const int cv = (arow_c + k) * 256 + v0 + arow_v;
      afrag[t] = kg < P ? reinterpret_cast<const u32x4*>(p.codebooks + (size_t)cv * G)[kg] : zero;
    }
    LUT_TRACE(1);  // loads issued
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 d = lut_mfma16<T>(afrag[t], bfrag, f32x4{0.f, 0.f, 0.f, 0.f});
      const int tile = wave * 8 + t, k = tile >> 5, v0 = (tile & 31) * 8;
      unsigned char* const dst = smem_raw + (v0 + kg * 2) * 256 + (k & 1) * 128 + (k >> 1) * 65536 + col * 8;
      *reinterpret_cast<float2*>(dst) = float2{d[0], d[1]};
      *reinterpret_cast<float2*>(dst + 256) = float2{d[2], d[3]};
    }
    LUT_TRACE(2);  // table written
    if (p.cells != nullptr) {
      // Fused finalize needs a bound of the slab sums that every workgroup of the layer computes identically:
      // max|codebook| -- the 16 waves' A fragments are the whole codebook -- and max|x| over ALL input groups (an extra
      // read of x, a few KiB).  15-bit magnitude patterns (integer order == magnitude order; NaN sorts above Inf).
      lut_us2 mc = {0, 0}, mx = {0, 0};
#pragma unroll
      for (int t = 0; t < 8; ++t) lut_absmax(mc, afrag[t]);
      const int chunks = p.in_groups * (G / 8);  // 16-B pieces of x
      for (int i = tid; i < chunks; i += 1024) lut_absmax(mx, reinterpret_cast<const u32x4*>(p.x)[i]);
      const uint32_t wc = wave_max_u32(mc.x > mc.y ? (uint32_t)mc.x : (uint32_t)mc.y);
      const uint32_t wx = wave_max_u32(mx.x > mx.y ? (uint32_t)mx.x : (uint32_t)mx.y);
      if (lane == 0) {
        uint32_t* slots = reinterpret_cast<uint32_t*>(smem_raw + (size_t)LUT_ENTRIES * 4);  // [16 waves] codebook, [16 waves] x
        slots[wave] = wc;
        slots[16 + wave] = wx;
      }
    }
  }
  LUT_TRACE(3);  // at the barrier
  __syncthreads();
  LUT_TRACE(4);  // table complete
  // fixed-point unit of the fused finalize: |slab sum| <= 16 groups x 8 codebooks x g x max|cb| x max|x| < 2^e; with
  // sh = 41 - e - ceil(log2(nslabs)) the nslabs addends of a row stay below 2^42 (the sum field is bits 63..20)
  int sh = 0;
  float bound = 0.f;
  if (p.cells != nullptr) {
    const uint32_t* slots = reinterpret_cast<const uint32_t*>(smem_raw + (size_t)LUT_ENTRIES * 4);
    uint32_t cm = 0u, xm = 0u;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      cm = slots[w] > cm ? slots[w] : cm;
      xm = slots[16 + w] > xm ? slots[16 + w] : xm;
    }
    bound = (float)(LUT_JS * LUT_KC * G) * T::to_float((uint16_t)cm) * T::to_float((uint16_t)xm);
    int e = 0;
    (void)frexpf(bound, &e);
    sh = 41 - e - (32 - __builtin_clz((unsigned)(p.nslabs > 1 ? p.nslabs - 1 : 1)));
  }

  // ---- rows.  Per step and half-wave: 4 v_perm_b32 (LDS address = {0, base byte 2, code byte, base byte 0}), 4
  // ds_read_b32, 3 adds, a 32-lane DPP sum (the total lands in the half's upper DPP row), and one DPP shift that files
  // the total in a 16-deep per-row shift register: after a batch of <= 16 steps lane 16 + i of a half holds the total of
  // the batch's step (count - 1 - i), and ONE vector pass hands all of them in (store, or fixed-point atomic + settle).
  uint32_t base[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) base[k] = (uint32_t)((k & 1) * 128 + s32 * 4 + (k >> 1) * 65536);
  float* const out = p.partial + (size_t)slab * p.M + row_begin;
  // Fused finalize: the owner lane adds the row's slab sum to the row's cell as a fixed-point integer (bits 63..20; +1
  // in the arrival counter, bits 9..0; +1 in bits 19..10 if the value is not finite) with ONE returning atomic -- integer
  // adds commute, so the total is independent of the arrival order -- and whoever finds nslabs - 1 earlier arrivals
  // applies scale and bias, rounds once, writes y and zeroes the cell.  A batch's returned values are looked at behind
  // the next batch's table reads.
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
  const bool owner = (lane & 16) != 0;  // the upper DPP row of each half holds the totals
  typedef __attribute__((address_space(3))) const float* lds_f32_ptr;
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw != 0u) __builtin_trap();  // LDS map above starts at 0
  float chain = 0.f;
  int filed = 0;  // steps in the shift register
  // four steps, straight-line (ring slots S .. S + 3); steps past the end read zeros and are never handed in
  auto quad = [&](auto slot0, int sb) {
    constexpr int S = decltype(slot0)::value;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t cw = cq[S + t];
      cq[S + t] = load_codes(sb + t + RING);
      const float a0 = *(lds_f32_ptr)(uintptr_t)lut_perm(cw, base[0], 0x0c020400u);
      const float a1 = *(lds_f32_ptr)(uintptr_t)lut_perm(cw, base[1], 0x0c020500u);
      const float a2 = *(lds_f32_ptr)(uintptr_t)lut_perm(cw, base[2], 0x0c020600u);
      const float a3 = *(lds_f32_ptr)(uintptr_t)lut_perm(cw, base[3], 0x0c020700u);
      float acc = (a0 + a1) + (a2 + a3);
      acc = group_ok ? acc : 0.f;
      acc = row16_sum(acc);
      acc += dpp_f32<0x142>(acc);  // row_bcast:15 -- the upper row of each half now holds the 32-lane total
      chain = lut_shift_in(acc, chain);
    }
  };
  // lane 16 + i (48 + i) of the wave holds the row of step `done` - 1 - i, i < filed
  auto hand_in = [&](int done) {
    const int i = lane & 15;
    const int r = rfirst + 32 * (done - 1 - i);
    const bool mine = owner && i < filed && r < nrows;
    if (p.cells == nullptr) {
      if (mine) out[r] = chain;
    } else {
      settle();  // the previous batch's atomics
      if (mine) {
        const bool finite = bound < __builtin_inff() && fabsf(chain) <= 2.f * bound;  // false for NaN / Inf anywhere
        const long long q = finite ? __float2ll_rn(ldexpf(chain, sh)) : 0ll;
        pend_mine = ((unsigned long long)q << 20) + (finite ? 1ull : 1025ull);
        pend_row = row_begin + r;
        pend_old = __hip_atomic_fetch_add(p.cells + pend_row, pend_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend_scale = p.scales[pend_row];
        pend_bias = bias_src[pend_row];
      }
    }
    filed = 0;
  };
  LUT_TRACE(5);  // walk starts
  for (int sb = 0; sb < nsteps;) {
    quad(std::integral_constant<int, 0>{}, sb);
    sb += 4;
    filed += 4;
    if (filed == 16 || sb >= nsteps) hand_in(sb);
    if (sb >= nsteps) break;
    quad(std::integral_constant<int, 4>{}, sb);
    sb += 4;
    filed += 4;
    if (filed == 16 || sb >= nsteps) hand_in(sb);
  }
  LUT_TRACE(6);  // rows handed in (atomics in flight)
  if (p.cells != nullptr) settle();
#ifdef AQLM_LUT_TRACE
  tr[7] = wall_clock64();
  if (lane == 0 && p.cells != nullptr) {
    unsigned long long* const out_tr = p.cells + ((p.M + 1023) & ~1023) + ((size_t)block * 16 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) out_tr[i] = tr[i];
  }
#endif
}

// scalar arguments (13 dwords): preloaded into SGPRs at wave launch, no kernel-argument fetch at the head of the kernel
struct LutTail {  // what only the fused finalize needs (not preloaded: used at the end of the kernel)
  unsigned long long* cells;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
};

template <class T, int G>
__global__ __launch_bounds__(1024) void gemv_8x8_lut_kernel(const uint8_t* codes, const uint16_t* codebooks, const uint16_t* x,
                                                            float* partial, int M, int in_groups, int nslabs, int nranges,
                                                            int rows_per_range, const LutTail tail) {
  const LutParams p{codes, codebooks, x, partial, M, in_groups, nslabs, nranges, rows_per_range, tail.cells, tail.scales, tail.bias, tail.y};
  gemv_8x8_lut_body<T, G>(p, blockIdx.x);
}

// shared-input launch: up to AQLM_HIP_MAX_SEGMENTS layers (own codes / codebooks / partials) times one x
struct LutSegment {
  const uint8_t* codes;
  const uint16_t* codebooks;
  float* partial;
  int M, nranges, rows_per_range, block_begin;
  unsigned long long* cells;  // fused finalize (nullptr: partials + finalize kernel)
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
};

struct LutMultiParams {
  const uint16_t* x;
  int in_groups, nslabs, nseg;
  LutSegment seg[AQLM_HIP_MAX_SEGMENTS];
};

template <class T, int G>
__global__ __launch_bounds__(1024) void gemv_8x8_lut_multi_kernel(const LutMultiParams mp) {
  LutParams p{};
  p.x = mp.x;
  p.in_groups = mp.in_groups;
  p.nslabs = mp.nslabs;
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
      p.cells = mp.seg[k].cells;
      p.scales = mp.seg[k].scales;
      p.bias = mp.seg[k].bias;
      p.y = mp.seg[k].y;
      begin = mp.seg[k].block_begin;
    }
  }
  gemv_8x8_lut_body<T, G>(p, (int)blockIdx.x - begin);
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

size_t gemv_8x8_lut_workspace(int out_features, int in_features, int in_group_size) {
  const int in_groups = in_features / in_group_size;
  const int nslabs = (in_groups + LUT_JS - 1) / LUT_JS;
  return (size_t)nslabs * out_features * sizeof(float);
}

template <class T, int G>
static int launch_lut(const LutParams& p, hipStream_t stream) {
  auto kern = gemv_8x8_lut_kernel<T, G>;
  const size_t lds = (size_t)LUT_ENTRIES * 4 + 128;  // + the 32 maximum slots of the fused finalize
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  const LutTail tail{p.cells, p.scales, p.bias, p.y};
  hipLaunchKernelGGL(kern, dim3(p.nslabs * p.nranges), dim3(1024), lds, stream, p.codes, p.codebooks, p.x, p.partial, p.M,
                     p.in_groups, p.nslabs, p.nranges, p.rows_per_range, tail);
  return check_hip(hipGetLastError(), "gemv_8x8_lut launch");
}

// batch-1 8x8 matvec through LDS look-up tables; AQLM_HIP_E_UNSUPPORTED when the shape does not fit
// `fused`: workspace = out_features zero-at-rest 64-bit cells (one kernel); else fp32 slab partials + a finalize kernel
int gemv_8x8_lut(const void* codes, const void* codebooks, const void* scales, const void* bias, const void* x, void* y,
                 int out_features, int in_features, int in_group_size, int dtype, void* workspace, size_t workspace_bytes,
                 hipStream_t stream, bool fused) {
  const int G = in_group_size;
  if (G != 8 && G != 16 && G != 32) return AQLM_HIP_E_UNSUPPORTED;
  LutParams p{};
  p.codes = (const uint8_t*)codes;
  p.codebooks = (const uint16_t*)codebooks;
  p.x = (const uint16_t*)x;
  p.partial = (float*)workspace;
  p.M = out_features;
  p.in_groups = in_features / G;
  p.nslabs = (p.in_groups + LUT_JS - 1) / LUT_JS;
  if (p.nslabs > 1023) return AQLM_HIP_E_UNSUPPORTED;  // (arrival counter of the fused finalize: 10 bits)
  if (fused) {
    if (!workspace || workspace_bytes < (size_t)out_features * 8 || ((uintptr_t)workspace & 7)) return AQLM_HIP_E_INVALID;
    p.cells = (unsigned long long*)workspace;
    p.scales = (const uint16_t*)scales;
    p.bias = (const uint16_t*)bias;
    p.y = (uint16_t*)y;
  } else if (workspace_bytes < (size_t)p.nslabs * out_features * sizeof(float) || !workspace) {
    return AQLM_HIP_E_INVALID;
  }
  p.nranges = std::max(1, 256 / p.nslabs);
  p.rows_per_range = (out_features + p.nranges - 1) / p.nranges;
  int e;
  if (dtype == AQLM_HIP_F16)
    e = G == 8 ? launch_lut<F16, 8>(p, stream) : G == 16 ? launch_lut<F16, 16>(p, stream) : launch_lut<F16, 32>(p, stream);
  else
    e = G == 8 ? launch_lut<BF16, 8>(p, stream) : G == 16 ? launch_lut<BF16, 16>(p, stream) : launch_lut<BF16, 32>(p, stream);
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

template <class T, int G>
static int launch_lut_multi(const LutMultiParams& mp, int blocks, hipStream_t stream) {
  auto kern = gemv_8x8_lut_multi_kernel<T, G>;
  const size_t lds = (size_t)LUT_ENTRIES * 4 + 128;
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, stream, mp);
  return check_hip(hipGetLastError(), "gemv_8x8_lut_multi launch");
}

// Shared-input variant: the ~256 workgroups are dealt to the segments in proportion to their rows.  Bit-identical to
// gemv_8x8_lut per segment only when the row ranges coincide; in general equal to fp32 rounding (same table, same
// per-row summation order -- only the slab partials are the same, so in fact results ARE identical: a row's value does
// not depend on its range).  workspace: sum over segments of gemv_8x8_lut_workspace(...).
int gemv_8x8_lut_multi(const aqlm_hip_segment* segments, int num_segments, const void* x, int in_features,
                       int in_group_size, int dtype, void* workspace, size_t workspace_bytes, hipStream_t stream, bool fused) {
  const int G = in_group_size;
  if (G != 8 && G != 16 && G != 32) return AQLM_HIP_E_UNSUPPORTED;
  LutMultiParams mp{};
  LutFinalizeMultiParams fm{};
  mp.x = (const uint16_t*)x;
  mp.in_groups = in_features / G;
  mp.nslabs = (mp.in_groups + LUT_JS - 1) / LUT_JS;
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
    if (fused) {  // the segment's cells, in segment order
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
  int e;
  if (dtype == AQLM_HIP_F16)
    e = G == 8 ? launch_lut_multi<F16, 8>(mp, blocks, stream) : G == 16 ? launch_lut_multi<F16, 16>(mp, blocks, stream)
                                                                         : launch_lut_multi<F16, 32>(mp, blocks, stream);
  else
    e = G == 8 ? launch_lut_multi<BF16, 8>(mp, blocks, stream) : G == 16 ? launch_lut_multi<BF16, 16>(mp, blocks, stream)
                                                                          : launch_lut_multi<BF16, 32>(mp, blocks, stream);
  if (e || fused) return e;
  if (dtype == AQLM_HIP_F16) hipLaunchKernelGGL(gemv_8x8_lut_finalize_multi<F16>, dim3(fblocks), dim3(256), 0, stream, fm);
  else hipLaunchKernelGGL(gemv_8x8_lut_finalize_multi<BF16>, dim3(fblocks), dim3(256), 0, stream, fm);
  return check_hip(hipGetLastError(), "gemv_8x8_lut_finalize_multi launch");
}

}  // namespace aqlm

using namespace aqlm;

static int lut_multi_entry(const aqlm_hip_segment* segments, int num_segments, const void* x, int in_features, int in_group_size,
                           int dtype, void* workspace, size_t workspace_bytes, void* stream, bool fused) {
  if (!segments || num_segments < 1 || num_segments > AQLM_HIP_MAX_SEGMENTS || !x) {
    set_last_error("aqlm_hip_gemv_8x8_lut_multi: 1..%d segments and a non-null x required (got %d)", AQLM_HIP_MAX_SEGMENTS,
                   num_segments);
    return AQLM_HIP_E_INVALID;
  }
  if (in_features <= 0 || in_group_size <= 0 || in_features % in_group_size != 0) {
    set_last_error("aqlm_hip_gemv_8x8_lut_multi: bad sizes in=%d g=%d", in_features, in_group_size);
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_gemv_8x8_lut_multi: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  for (int k = 0; k < num_segments; ++k) {
    const aqlm_hip_segment& sg = segments[k];
    if (!sg.codes || !sg.codebook || !sg.scales || !sg.y || sg.out_features <= 0) {
      set_last_error("aqlm_hip_gemv_8x8_lut_multi: null pointer or non-positive size in segment %d", k);
      return AQLM_HIP_E_INVALID;
    }
    if (!aligned16(sg.codebook) || (reinterpret_cast<uintptr_t>(sg.codes) & 7u)) {
      set_last_error("aqlm_hip_gemv_8x8_lut_multi: misaligned buffer in segment %d", k);
      return AQLM_HIP_E_UNSUPPORTED;
    }
  }
  if (!aligned16(x)) {
    set_last_error("aqlm_hip_gemv_8x8_lut_multi: misaligned x");
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const int e = gemv_8x8_lut_multi(segments, num_segments, x, in_features, in_group_size, dtype, workspace, workspace_bytes,
                                   (hipStream_t)stream, fused);
  if (e == AQLM_HIP_E_UNSUPPORTED) set_last_error("aqlm_hip_gemv_8x8_lut_multi: in_group_size %d not in {8,16,32}", in_group_size);
  if (e == AQLM_HIP_E_INVALID) set_last_error("aqlm_hip_gemv_8x8_lut_multi: workspace too small (sum of aqlm_hip_workspace_bytes(AQLM_HIP_OP_GEMV_8X8_LUT, ...) over the segments)");
  return e;
}

extern "C" int aqlm_hip_gemv_8x8_lut_multi(const aqlm_hip_segment* segments, int num_segments, const void* x,
                                           int in_features, int in_group_size, int dtype, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  return lut_multi_entry(segments, num_segments, x, in_features, in_group_size, dtype, workspace, workspace_bytes, stream, false);
}

extern "C" int aqlm_hip_gemv_8x8_lut_multi_fused(const aqlm_hip_segment* segments, int num_segments, const void* x,
                                                 int in_features, int in_group_size, int dtype, void* cells,
                                                 size_t cells_bytes, void* stream) {
  return lut_multi_entry(segments, num_segments, x, in_features, in_group_size, dtype, cells, cells_bytes, stream, true);
}

static int lut_entry(const void* codes, const void* codebooks, const void* scales, const void* bias, const void* x, void* y,
                     int out_features, int in_features, int in_group_size, int dtype, void* workspace, size_t workspace_bytes,
                     void* stream, bool fused) {
  if (!codes || !codebooks || !scales || !x || !y) {
    set_last_error("aqlm_hip_gemv_8x8_lut: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (out_features <= 0 || in_features <= 0 || in_group_size <= 0 || in_features % in_group_size != 0) {
    set_last_error("aqlm_hip_gemv_8x8_lut: bad sizes out=%d in=%d g=%d", out_features, in_features, in_group_size);
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_gemv_8x8_lut: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (!aligned16(codebooks) || !aligned16(x) || (reinterpret_cast<uintptr_t>(codes) & 7u)) {
    set_last_error("aqlm_hip_gemv_8x8_lut: misaligned buffer");
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const int e = gemv_8x8_lut(codes, codebooks, scales, bias, x, y, out_features, in_features, in_group_size, dtype,
                             workspace, workspace_bytes, (hipStream_t)stream, fused);
  if (e == AQLM_HIP_E_UNSUPPORTED) set_last_error("aqlm_hip_gemv_8x8_lut: in_group_size %d not in {8,16,32}", in_group_size);
  if (e == AQLM_HIP_E_INVALID) set_last_error("aqlm_hip_gemv_8x8_lut: workspace / cells too small, null or misaligned");
  return e;
}

extern "C" int aqlm_hip_gemv_8x8_lut(const void* codes, const void* codebooks, const void* scales, const void* bias,
                                     const void* x, void* y, int out_features, int in_features, int in_group_size,
                                     int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  return lut_entry(codes, codebooks, scales, bias, x, y, out_features, in_features, in_group_size, dtype, workspace,
                   workspace_bytes, stream, false);
}

extern "C" int aqlm_hip_gemv_8x8_lut_fused(const void* codes, const void* codebooks, const void* scales, const void* bias,
                                           const void* x, void* y, int out_features, int in_features, int in_group_size,
                                           int dtype, void* cells, size_t cells_bytes, void* stream) {
  return lut_entry(codes, codebooks, scales, bias, x, y, out_features, in_features, in_group_size, dtype, cells, cells_bytes,
                   stream, true);
}
