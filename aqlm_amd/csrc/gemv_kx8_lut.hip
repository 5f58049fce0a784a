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

// `block` = the workgroup's index within its own layer (== blockIdx.x for a single-layer launch)
template <class T, int G>
__device__ __forceinline__ void gemv_8x8_lut_body(const LutParams& p, const int block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* const lut = reinterpret_cast<float*>(smem_raw);  // [16 groups][8 codebooks][256]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, quarter = lane >> 4;
  const int slab = block % p.nslabs, range = block / p.nslabs;
  const int j0 = slab * LUT_JS;
  const int row_begin = range * p.rows_per_range;
  int nrows = p.M - row_begin;
  nrows = nrows < 0 ? 0 : (nrows < p.rows_per_range ? nrows : p.rows_per_range);

  // code bytes of this quarter-wave's first rows go in flight before the table is built (clamped, unconditional)
  const int jmine = j0 + l16 < p.in_groups ? j0 + l16 : p.in_groups - 1;
  const bool group_ok = j0 + l16 < p.in_groups;
  const int r0 = wave * 4 + quarter;  // rows r0, r0 + 64, ...
  auto load_codes = [&](int r) -> u32x2 {
    const int rc = r < nrows ? r : (nrows > 0 ? nrows - 1 : 0);
    return __builtin_nontemporal_load(
        reinterpret_cast<const u32x2*>(p.codes + (((long)row_begin + rc) * p.in_groups + jmine) * 8));
  };
  u32x2 cq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) cq[k] = load_codes(r0 + 64 * k);

  // ---- table on the matrix cores: lut[jl][cv] = sum_k cb[cv][k] * x[j0+jl][k] is a [2048 x g] x [g x 16] product.
  // v_mfma_f32_16x16x32: A = 16 (c,v) rows x 32 k (lane l: row l%16, 8 k of piece l/16), B = x of the 16 groups
  // (lane l: group l%16, same piece), D[row (l/16)*4 + r][group l%16] -> 4 consecutive table entries = one 16-B LDS
  // write.  g < 32 pads k with zero pieces.  128 tiles per workgroup, 8 per wave (the VALU version of this build cost
  // 3.4 us per workgroup).
  {
    constexpr int P = G / 8;  // pieces of 8 k per vector: 1, 2 or 4
    const int col = lane & 15, kg = lane >> 4;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const int jb = j0 + col < p.in_groups ? j0 + col : p.in_groups - 1;
    const u32x4 bfrag = kg < P ? reinterpret_cast<const u32x4*>(p.x + (size_t)jb * G)[kg] : zero;
    u32x4 afrag[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int cv = (wave * 8 + t) * 16 + col;  // this lane's A row
      afrag[t] = kg < P ? reinterpret_cast<const u32x4*>(p.codebooks + (size_t)cv * G)[kg] : zero;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 d = lut_mfma16<T>(afrag[t], bfrag, f32x4{0.f, 0.f, 0.f, 0.f});
      const int cv0 = (wave * 8 + t) * 16 + kg * 4;
      *reinterpret_cast<f32x4*>(lut + col * (LUT_KC * 256) + cv0) = d;
    }
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
  __syncthreads();
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

  // ---- rows: lane = input group j0 + l16 (8 code bytes), quarter-wave = one row
  const float* const my = lut + l16 * (LUT_KC * 256);
  float* const out = p.partial + (size_t)slab * p.M + row_begin;
  // Fused finalize: lane 0 of a quarter-wave adds the row's slab sum to the row's cell as a fixed-point integer (bits
  // 63..20; +1 in the arrival counter, bits 9..0; +1 in bits 19..10 if the value is not finite) with ONE returning
  // atomic -- integer adds commute, so the total is independent of the arrival order -- and whoever finds
  // nslabs - 1 earlier arrivals applies scale and bias, rounds once, writes y and zeroes the cell.  The returned values
  // of a round are looked at one round later, so the atomics' round trip hides behind the next rows' table reads.
  unsigned long long pend_old[4], pend_mine[4];
  uint16_t pend_scale[4] = {0, 0, 0, 0}, pend_bias[4] = {0, 0, 0, 0};  // requested with the atomic: behind the last-arrival test they
  int pend_row[4] = {-1, -1, -1, -1};                                  // were a second round trip at the very end of the kernel
  const uint16_t* const bias_src = p.bias ? p.bias : p.scales;
  auto settle = [&](int k) {
    if (pend_row[k] >= 0 && (pend_old[k] & 1023ull) == (unsigned long long)(p.nslabs - 1)) {
      const int row = pend_row[k];
      const unsigned long long cell = pend_old[k] + pend_mine[k];
      float sv = (float)ldexp((double)((long long)cell >> 20), -sh);
      if ((cell >> 10) & 1023ull) sv = __builtin_nanf("");
      const float scale = T::to_float(pend_scale[k]);
      const float bias = p.bias ? T::to_float(pend_bias[k]) : 0.f;
      p.y[row] = T::from_float(__builtin_fmaf(sv, scale, bias));
      __hip_atomic_store(p.cells + row, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    pend_row[k] = -1;
  };
  for (int rbase = r0; __any(rbase < nrows); rbase += 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = rbase + 64 * k;
      const u32x2 cw = cq[k];
      cq[k] = load_codes(r + 256);  // next round's codes for this slot
      float acc = 0.f;
      if (group_ok) {
        acc += my[0 * 256 + (cw.x & 0xffu)];
        acc += my[1 * 256 + ((cw.x >> 8) & 0xffu)];
        acc += my[2 * 256 + ((cw.x >> 16) & 0xffu)];
        acc += my[3 * 256 + (cw.x >> 24)];
        acc += my[4 * 256 + (cw.y & 0xffu)];
        acc += my[5 * 256 + ((cw.y >> 8) & 0xffu)];
        acc += my[6 * 256 + ((cw.y >> 16) & 0xffu)];
        acc += my[7 * 256 + (cw.y >> 24)];
      }
      acc = row16_sum(acc);
      if (p.cells == nullptr) {
        if (l16 == 0 && r < nrows) out[r] = acc;
      } else {
        settle(k);  // the previous round's atomic of this slot
        if (l16 == 0 && r < nrows) {
          const bool finite = bound < __builtin_inff() && fabsf(acc) <= 2.f * bound;  // false for NaN / Inf anywhere
          const long long q = finite ? __float2ll_rn(ldexpf(acc, sh)) : 0ll;
          pend_mine[k] = ((unsigned long long)q << 20) + (finite ? 1ull : 1025ull);
          pend_row[k] = row_begin + r;
          pend_old[k] = __hip_atomic_fetch_add(p.cells + pend_row[k], pend_mine[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pend_scale[k] = p.scales[pend_row[k]];
          pend_bias[k] = bias_src[pend_row[k]];
        }
      }
    }
  }
  if (p.cells != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; ++k) settle(k);
  }
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
