// Fused dequant-tile -> MFMA GEMM for the 1x16 scheme at large batch (prefill / bs=128) on gfx950.
//
// Replaces (behaviour, not code): code1x16_matmat_dequant = Code1x16Dequant (W[out,in] materialised in HBM,
// reference cuda_kernel.cu:98-142) + F::linear / cuBLAS (cuda_kernel.cpp:249-301) + scale/bias epilogue launches.
//
// Key observation: with v_mfma_f32_32x32x16_{f16,bf16} the A operand of lane l is 8 consecutive k of row l%32
// (k-half l/32) -- which is exactly one 16-byte AQLM codebook vector (g=8), or one half of one (g=16).  So a
// gathered codebook entry IS an MFMA fragment: W is never written anywhere, not even to LDS.  Per 16-deep k step
// a wave issues ONE 16-B gather per lane (32 rows x 2 k-halves) and reuses it for every 32-column batch tile.
//
//   C[row = W row][col = batch]  +=  A = W[32 rows][16 k]  x  B = X^T[16 k][32 batch]
//
// Block = 4 waves = 128 output rows (one 32-row tile per wave, waves independent) x one K slice x all batch
// columns (<= 128, as NBT tiles of 32).  X is staged in 64-deep chunks through a double-buffered, XOR-swizzled
// LDS image shared by the 4 waves.  K is split over `ksplit` blocks so that the grid fills 256 CUs; fp32
// partials go to a workspace [ksplit][out][Bpad] and a second small kernel sums them, applies
// scales/bias, transposes to Y[B][out] and rounds once.
//
// Roofline: MFMA-bound in principle (2*B*out*in flop), but for out=in=4096, B=128 the 2.1 M random 16-B gathers
// (~1 lane/clk/CU) take ~2x the MFMA time, so this kernel is L2-gather bound like the gemv.
#include <algorithm>

#include "aqlm_common.h"

namespace aqlm {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <class T>
__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c);
template <>
__device__ __forceinline__ f32x16 mfma32<F16>(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma32<BF16>(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

struct GemmParams {
  const uint8_t* codes;
  const uint8_t* codebook;
  const uint16_t* X;
  float* partial;  // [ksplit][M][Bpad]
  int M, K, B, Bpad;
  int in_groups;
  int kslice;  // k elements per block (multiple of 64)
  int ksplit;
  long xs;
  int cb_bytes;
  uint32_t x_bytes;  // extent of the X slab (rows >= B read as zeros through the bounds-checked descriptor)
};

constexpr int BK = 64;  // k depth of one LDS chunk of X

// LDS image of one X chunk: row b (batch) holds 64 k = 8 pieces of 16 B; piece c is stored at slot c ^ ((b>>1)&7) so
// that the 16 lanes of a ds_read_b128 service group (16 distinct b mod 16) hit 16 distinct 16-B slots.
__device__ __forceinline__ int xswz(int b, int c) { return b * 8 + (c ^ ((b >> 1) & 7)); }

template <class T, int G, int NBT>
__global__ __launch_bounds__(256) void gemm_1x16_mfma_kernel(const GemmParams p) {
  constexpr int NROWS_X = NBT * 32;
  constexpr int PIECES = NROWS_X * 8;          // 16-B pieces per chunk
  constexpr int PER_THREAD = (PIECES + 255) / 256;
  constexpr int CODES_PER_CHUNK = BK / G;      // 8 (g8) or 4 (g16)
  constexpr int CWN = CODES_PER_CHUNK / 2;     // dwords of codes per chunk per row
  __shared__ __attribute__((aligned(16))) u32x4 xl[2][PIECES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5;
  const int row_blk = blockIdx.x / p.ksplit;
  const int ks = blockIdx.x - row_blk * p.ksplit;
  const int row_tile0 = row_blk * 128 + wave * 32;
  int my_row = row_tile0 + (lane & 31);
  const bool row_ok = my_row < p.M;
  if (!row_ok) my_row = p.M - 1;  // clamp: computed but never stored
  const int k_begin = ks * p.kslice;
  const int k_end = k_begin + p.kslice < p.K ? k_begin + p.kslice : p.K;
  const int nchunks = (k_end - k_begin + BK - 1) / BK;

  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.codebook, 0, p.cb_bytes, 0x00020000);
  const uint8_t* code_row = p.codes + (long)my_row * p.in_groups * 2;

  f32x16 acc[NBT];
#pragma unroll
  for (int t = 0; t < NBT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // global -> register staging of one X chunk.  The loads are UNCONDITIONAL buffer loads (rows >= B and k >= k_end
  // read as zeros through the bounds-checked descriptor / an out-of-range offset): with the load under a branch hipcc
  // emitted vmcnt(0) right behind it, i.e. every chunk step first waited ~1 us for X and only then issued the
  // codebook gathers of the next chunk (traced through the ISA: one L2 latency + one gather latency per 64-deep chunk).
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
  static_assert(PIECES % 256 == 0, "the X chunk splits evenly over the block");
  u32x4 xr[2][PER_THREAD];  // X chunks run two ahead in registers, one ahead in LDS
  uint32_t xoff[PER_THREAD];
#pragma unroll
  for (int s = 0; s < PER_THREAD; ++s) {
    const int q = tid + s * 256;
    xoff[s] = (uint32_t)(((long)(q >> 3) * p.xs + k_begin + (q & 7) * 8) * 2);
  }
  auto load_x = [&](int chunk, u32x4 (&dst)[PER_THREAD]) {
#pragma unroll
    for (int s = 0; s < PER_THREAD; ++s) {
      const bool in_k = k_begin + chunk * BK + (int)((tid + s * 256) & 7) * 8 < k_end;
      dst[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, in_k ? xoff[s] + (uint32_t)chunk * (BK * 2) : 0xfffffff0u, 0, 0);
    }
  };
  auto store_x = [&](int buf, const u32x4 (&src)[PER_THREAD]) {
#pragma unroll
    for (int s = 0; s < PER_THREAD; ++s) {
      const int q = tid + s * 256;
      xl[buf][xswz(q >> 3, q & 7)] = src[s];
    }
  };
  const int last_chunk = nchunks - 1;
  auto load_codes = [&](int chunk, uint32_t (&cw)[CWN]) {  // unconditional: chunks past the slice re-read the last one
    chunk = chunk < last_chunk ? chunk : last_chunk;
    const int k0 = k_begin + chunk * BK;
    const uint8_t* src = code_row + (long)(k0 / G) * 2;
    if constexpr (CWN == 4) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src);
      cw[0] = v.x; cw[1] = v.y; cw[2] = v.z; cw[3] = v.w;
    } else {
      const u32x2 v = *reinterpret_cast<const u32x2*>(src);
      cw[0] = v.x; cw[1] = v.y;
    }
  };

  // gathers: one 16-B entry per lane per 16-deep k step = 4 per 64-deep chunk; entry == MFMA A fragment
  auto gather = [&](const uint32_t (&cw)[CWN], u32x4 (&af)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint32_t code, piece;
      if constexpr (G == 8) {
        code = (cw[kk] >> (16 * half)) & 0xffffu;  // code index 2*kk + half
        piece = 0;
      } else {
        code = (cw[kk >> 1] >> (16 * (kk & 1))) & 0xffffu;  // code index kk, lane-half picks the 16-B half
        piece = half;
      }
      af[kk] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, code * (uint32_t)(G * 2) + piece * 16, 0, 0);
    }
  };

  // Software pipeline, in batches of NBATCH chunks.  Inside a batch the code is straight-line (fully unrolled, static
  // ring slots): all code words of the batch are requested up front, codebook gathers run GD-1 = 3 chunks ahead of
  // the MFMAs that consume them (one 64-deep chunk of MFMAs is ~0.3 us, an L2 gather ~1.5 us under load: a distance of
  // one chunk left every step waiting for its own gathers), X runs two chunks ahead in registers and one in LDS, and
  // NOTHING is in flight across the loop back-edge: hipcc's wait-count pass merges the prologue and back-edge states at
  // a loop header and then waits with vmcnt(0) for every register loaded in the previous iteration.  A batch pays one
  // pipeline refill instead (none at all for K slices of <= NBATCH chunks, e.g. 4096 x 4096 with the 8-way K split).
  constexpr int NBATCH = 8, GD = 4;
  uint32_t cw[NBATCH][CWN];
  u32x4 af[GD][4];
  auto mfmas = [&](int buf, const u32x4 (&af_cur)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int t = 0; t < NBT; ++t) {
        const u32x4 bfrag = xl[buf][xswz(t * 32 + (lane & 31), kk * 2 + half)];
        acc[t] = mfma32<T>(af_cur[kk], bfrag, acc[t]);
      }
    }
  };
  for (int base = 0; base < nchunks; base += NBATCH) {
#pragma unroll
    for (int c = 0; c < NBATCH; ++c) load_codes(base + c, cw[c]);
    load_x(base, xr[0]);
    load_x(base + 1, xr[1]);
#pragma unroll
    for (int c = 0; c < GD - 1; ++c) gather(cw[c], af[c]);
    store_x(0, xr[0]);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NBATCH; ++s) {
      const int buf = s & 1;  // NBATCH is even: the first chunk of every batch uses buffer 0
      if (s + GD - 1 < NBATCH) gather(cw[s + GD - 1], af[(s + GD - 1) % GD]);   // compile-time conditions only
      if (s + 2 < NBATCH) load_x(base + s + 2, xr[s % 2]);                       // slot of chunk s: already in LDS
      mfmas(buf, af[s % GD]);
      if (s + 1 < NBATCH) store_x(buf ^ 1, xr[(s + 1) % 2]);
      __syncthreads();
    }
  }

  // fp32 partials: C layout col = lane&31 (batch), row = (r&3) + 8*(r>>2) + 4*half
  float* out = p.partial + (long)ks * p.M * p.Bpad;
#pragma unroll
  for (int t = 0; t < NBT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_tile0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row < p.M) out[(long)row * p.Bpad + t * 32 + (lane & 31)] = acc[t][r];
    }
}

// Y[b][m] = (sum_s partial[s][m][b]) * scales[m] + bias[m]; 32x32 tile transpose through LDS.
struct FinalizeParams {
  const float* partial;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* Y;
  int M, B, Bpad, ksplit;
  long ys;
};

template <class T>
__global__ __launch_bounds__(256) void gemm_finalize_kernel(const FinalizeParams p) {
  __shared__ float tile[32][33];
  const int m0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + i * 8, b = b0 + tx;
    float s = 0.f;
    if (m < p.M && b < p.Bpad) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = k < p.ksplit ? p.partial[((long)k * p.M + m) * p.Bpad + b] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
      for (int k = 8; k < p.ksplit; ++k) s += p.partial[((long)k * p.M + m) * p.Bpad + b];
    }
    tile[ty + i * 8][tx] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = b0 + ty + i * 8, m = m0 + tx;
    if (m < p.M && b < p.B) {
      const float scale = T::to_float(p.scales[m]);
      const float bias = p.bias ? T::to_float(p.bias[m]) : 0.f;
      p.Y[(long)b * p.ys + m] = T::from_float(__builtin_fmaf(tile[tx][ty + i * 8], scale, bias));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Split-K-free variant (experimental, opt-in through the `gemm_splitk_free` tuning knob; measured SLOWER than the
// split-K kernel on MI355X at 4096x4096: 41 vs 30 us at B = 128 -- each block re-streams all of X through L2 and a
// 4-wave block cannot hide that latency): one block = 16 output rows x the whole K x all batch
// columns.  The 4 waves split K four ways, each streams ITS quarter of X through a wave-private LDS image (no block
// barrier in the main loop) and accumulates 16 x Bpad in registers with v_mfma_f32_16x16x32; at the end the 4
// accumulators are added through LDS and the block writes Y directly (scale, bias, one rounding): no fp32 partials
// in HBM, no second kernel.  A operand of lane l = 8 consecutive k of row l%16, k-group l/16 = one codebook entry.
// Cost model per CU (M = 4096: one block per CU): 8192 gathered 128-B lines + B*K*2 bytes of X through the 64 B/clk
// L1-fill path -> ~8 us + 6.8 us at B = 128.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class T>
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c);
template <>
__device__ __forceinline__ f32x4 mfma16<F16>(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<BF16>(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct Gemm16Params {
  const uint8_t* codes;
  const uint8_t* codebook;
  const uint16_t* X;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* Y;
  int M, K, B, in_groups;
  long xs, ys;
  int cb_bytes;
};

template <class T, int G, int NBT>  // NBT = 16-column batch tiles (Bpad = 16 * NBT <= 128)
__global__ __launch_bounds__(256) void gemm_1x16_mfma16_kernel(const Gemm16Params p) {
  constexpr int BPAD = NBT * 16;
  constexpr int PIECES = BPAD * 8;             // 16-B pieces of one 64-deep X chunk
  constexpr int PER_LANE = PIECES / 64;        // = 2 * NBT
  constexpr int CWN = (64 / G) / 2;            // dwords of codes per row per chunk: 4 (g8) / 2 (g16)
  __shared__ __attribute__((aligned(16))) u32x4 xl_all[4][PIECES];   // wave-private X chunk images (<= 64 KiB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4* const xl = xl_all[wave];
  const int arow = lane & 15, kg = lane >> 4;
  const int row0 = blockIdx.x * 16;
  int my_row = row0 + arow;
  if (my_row >= p.M) my_row = p.M - 1;  // clamp: computed, never stored
  const int kq = p.K >> 2;               // this wave's K range
  const int k_begin = wave * kq;
  const int nchunks = kq >> 6;

  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.codebook, 0, p.cb_bytes, 0x00020000);
  const uint8_t* const code_row = p.codes + (long)my_row * p.in_groups * 2;

  f32x4 acc[NBT];
#pragma unroll
  for (int t = 0; t < NBT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_codes = [&](int chunk, uint32_t (&cw)[CWN]) {
    const uint8_t* src = code_row + (long)((k_begin + chunk * 64) / G) * 2;
    if constexpr (CWN == 4) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src);
      cw[0] = v.x; cw[1] = v.y; cw[2] = v.z; cw[3] = v.w;
    } else {
      const u32x2 v = *reinterpret_cast<const u32x2*>(src);
      cw[0] = v.x; cw[1] = v.y;
    }
  };
  // two 32-deep steps per chunk; lane (arow, kg) needs code 4*step + kg (g8) or code 2*step + kg/2, half kg&1 (g16)
  auto gather = [&](const uint32_t (&cw)[CWN], u32x4 (&af)[2]) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      uint32_t off;
      if constexpr (G == 8) {
        const uint32_t dw = (kg & 2) ? cw[2 * st + 1] : cw[2 * st];
        off = ((dw >> (16 * (kg & 1))) & 0xffffu) * 16u;
      } else {
        const uint32_t dw = cw[st];
        off = ((dw >> (16 * (kg >> 1))) & 0xffffu) * 32u + (uint32_t)(kg & 1) * 16u;
      }
      af[st] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    }
  };
  // X chunk: lane takes pieces q = lane + 64*s -> row b = q >> 3, piece c = q & 7 (full 128-B rows: no over-fetch)
  auto load_x = [&](int chunk, u32x4 (&xr)[PER_LANE]) {
    const int k0 = k_begin + chunk * 64;
#pragma unroll
    for (int s = 0; s < PER_LANE; ++s) {
      const int q = lane + 64 * s, b = q >> 3, c = q & 7;
      xr[s] = b < p.B ? *reinterpret_cast<const u32x4*>(p.X + (long)b * p.xs + k0 + c * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto store_x = [&](const u32x4 (&xr)[PER_LANE]) {
#pragma unroll
    for (int s = 0; s < PER_LANE; ++s) {
      const int q = lane + 64 * s;
      xl[xswz(q >> 3, q & 7)] = xr[s];
    }
  };

  uint32_t cw_a[CWN], cw_b[CWN];
  u32x4 af_a[2], af_b[2];
  u32x4 xr[PER_LANE];
  load_x(0, xr);
  load_codes(0, cw_a);
  if (nchunks > 1) load_codes(1, cw_b);
  gather(cw_a, af_a);
  store_x(xr);

  auto step = [&](int ch, uint32_t (&cw_free)[CWN], const uint32_t (&cw_next)[CWN], const u32x4 (&af_cur)[2], u32x4 (&af_nxt)[2]) {
    const bool more = ch + 1 < nchunks;
    if (more) {
      load_x(ch + 1, xr);
      gather(cw_next, af_nxt);
      if (ch + 2 < nchunks) load_codes(ch + 2, cw_free);
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int t = 0; t < NBT; ++t) {
        const u32x4 bfrag = xl[xswz(t * 16 + arow, st * 4 + kg)];
        acc[t] = mfma16<T>(af_cur[st], bfrag, acc[t]);
      }
    }
    if (more) store_x(xr);  // same-wave LDS ops execute in order: these writes follow the reads above
  };
  for (int ch = 0; ch < nchunks; ch += 2) {
    step(ch, cw_a, cw_b, af_a, af_b);
    if (ch + 1 < nchunks) step(ch + 1, cw_b, cw_a, af_b, af_a);
  }

  // cross-wave K reduction through LDS (re-using the X images), then the fused epilogue
  __syncthreads();
  float* const red = reinterpret_cast<float*>(&xl_all[0][0]);  // [4 waves][16 rows][BPAD]
#pragma unroll
  for (int t = 0; t < NBT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 16 + kg * 4 + r) * BPAD + t * 16 + arow] = acc[t][r];
  __syncthreads();
  for (int q = tid; q < 16 * BPAD; q += 256) {
    const int b = q >> 4, r = q & 15;   // consecutive threads -> consecutive rows of one batch column (32-B runs of Y)
    const int row = row0 + r;
    if (b < p.B && row < p.M) {
      const float s = red[(0 * 16 + r) * BPAD + b] + red[(1 * 16 + r) * BPAD + b] + red[(2 * 16 + r) * BPAD + b] +
                      red[(3 * 16 + r) * BPAD + b];
      const float scale = T::to_float(p.scales[row]);
      const float bias = p.bias ? T::to_float(p.bias[row]) : 0.f;
      p.Y[(long)b * p.ys + row] = T::from_float(__builtin_fmaf(s, scale, bias));
    }
  }
}

template <class T, int G>
static int launch_gemm16(const Gemm16Params& p, int bpad, hipStream_t stream) {
  const int blocks = (p.M + 15) / 16;
  if (bpad <= 16) hipLaunchKernelGGL((gemm_1x16_mfma16_kernel<T, G, 1>), dim3(blocks), dim3(256), 0, stream, p);
  else if (bpad <= 32) hipLaunchKernelGGL((gemm_1x16_mfma16_kernel<T, G, 2>), dim3(blocks), dim3(256), 0, stream, p);
  else if (bpad <= 64) hipLaunchKernelGGL((gemm_1x16_mfma16_kernel<T, G, 4>), dim3(blocks), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((gemm_1x16_mfma16_kernel<T, G, 8>), dim3(blocks), dim3(256), 0, stream, p);
  return check_hip(hipGetLastError(), "gemm_1x16_mfma16 launch");
}

struct GemmPlan {
  int ksplit, kslice, Bpad, nbt;
};

static GemmPlan plan_gemm(int B, int M, int K) {
  GemmPlan g;
  g.nbt = (std::min(B, 128) + 31) / 32;
  g.Bpad = g.nbt * 32;
  const int row_blocks = (M + 127) / 128;
  const int kchunks = (K + BK - 1) / BK;
  int ksplit = std::max(1, 256 / row_blocks);
  ksplit = std::min(ksplit, kchunks);
  const int chunks_per = (kchunks + ksplit - 1) / ksplit;
  g.kslice = chunks_per * BK;
  g.ksplit = (kchunks + chunks_per - 1) / chunks_per;
  return g;
}

template <class T, int G>
static int launch_gemm(const GemmParams& p, int nbt, hipStream_t stream) {
  const int blocks = ((p.M + 127) / 128) * p.ksplit;
  switch (nbt) {
    case 1: hipLaunchKernelGGL((gemm_1x16_mfma_kernel<T, G, 1>), dim3(blocks), dim3(256), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((gemm_1x16_mfma_kernel<T, G, 2>), dim3(blocks), dim3(256), 0, stream, p); break;
    case 3: hipLaunchKernelGGL((gemm_1x16_mfma_kernel<T, G, 3>), dim3(blocks), dim3(256), 0, stream, p); break;
    default: hipLaunchKernelGGL((gemm_1x16_mfma_kernel<T, G, 4>), dim3(blocks), dim3(256), 0, stream, p); break;
  }
  return check_hip(hipGetLastError(), "gemm_1x16_mfma launch");
}

}  // namespace aqlm

using namespace aqlm;

namespace aqlm {
size_t gemv_8x8_lut_workspace(int out_features, int in_features, int in_group_size);
}

extern "C" size_t aqlm_hip_workspace_bytes(int op, int batch, int out_features, int in_features) {
  if (op == AQLM_HIP_OP_GEMV_8X8_LUT && batch > 0 && out_features > 0 && in_features > 0 && in_features % batch == 0)
    return aqlm::gemv_8x8_lut_workspace(out_features, in_features, /*in_group_size=*/batch);
  if (batch <= 0 || out_features <= 0 || in_features <= 0) return 0;
  if (op == AQLM_HIP_OP_GEMV_1X16_PACKED)  // fp32 partials [16 slices][rows of x][out]
    return (size_t)16 * std::min(batch, AQLM_HIP_MAX_GEMV_BATCH) * out_features * sizeof(float);
  if (op != AQLM_HIP_OP_GEMM_1X16_MFMA) return 0;
  const GemmPlan g = plan_gemm(batch, out_features, in_features);
  return (size_t)g.ksplit * out_features * g.Bpad * sizeof(float);
}

extern "C" int aqlm_hip_gemm_1x16_mfma(const void* codes, const void* codebook, const void* scales, const void* bias,
                                       const void* X, void* Y, int batch, int out_features, int in_features,
                                       int in_group_size, long xs, long ys, int dtype, void* workspace,
                                       size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!codes || !codebook || !scales || !X || !Y) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (batch <= 0 || out_features <= 0 || in_features <= 0) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: sizes must be positive");
    return AQLM_HIP_E_INVALID;
  }
  if (in_group_size != 8 && in_group_size != 16) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: only codebooks with 8 or 16 features are supported, got %d",
                   in_group_size);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (in_features % BK != 0 || !aligned16(codes) || !aligned16(codebook) || !aligned16(X) || xs % 8 != 0) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: needs in_features %% 64 == 0 and 16-B aligned codes/codebook/X rows");
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const size_t need = aqlm_hip_workspace_bytes(AQLM_HIP_OP_GEMM_1X16_MFMA, batch, out_features, in_features);
  if (!workspace || workspace_bytes < need) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return AQLM_HIP_E_INVALID;
  }
  // batch > 128 is processed in slabs of 128 columns (codes re-gathered per slab)
  const bool splitk_free = (in_features % 256 == 0) && tuning().gemm_splitk_free;  // measured slower: opt-in
  for (int b0 = 0; b0 < batch; b0 += 128) {
    const int nb = std::min(128, batch - b0);
    if (splitk_free) {
      Gemm16Params q{};
      q.codes = (const uint8_t*)codes;
      q.codebook = (const uint8_t*)codebook;
      q.X = (const uint16_t*)X + (long)b0 * xs;
      q.scales = (const uint16_t*)scales;
      q.bias = (const uint16_t*)bias;
      q.Y = (uint16_t*)Y + (long)b0 * ys;
      q.M = out_features;
      q.K = in_features;
      q.B = nb;
      q.in_groups = in_features / in_group_size;
      q.xs = xs;
      q.ys = ys;
      q.cb_bytes = 65536 * in_group_size * 2;
      const int bpad = (nb + 15) / 16 * 16;
      int e;
      if (dtype == AQLM_HIP_F16)
        e = in_group_size == 8 ? launch_gemm16<F16, 8>(q, bpad, stream) : launch_gemm16<F16, 16>(q, bpad, stream);
      else
        e = in_group_size == 8 ? launch_gemm16<BF16, 8>(q, bpad, stream) : launch_gemm16<BF16, 16>(q, bpad, stream);
      if (e) return e;
      continue;
    }
    const GemmPlan g = plan_gemm(nb, out_features, in_features);
    GemmParams p{};
    p.codes = (const uint8_t*)codes;
    p.codebook = (const uint8_t*)codebook;
    p.X = (const uint16_t*)X + (long)b0 * xs;
    p.partial = (float*)workspace;
    p.M = out_features;
    p.K = in_features;
    p.B = nb;
    p.Bpad = g.Bpad;
    p.in_groups = in_features / in_group_size;
    p.kslice = g.kslice;
    p.ksplit = g.ksplit;
    p.xs = xs;
    p.cb_bytes = 65536 * in_group_size * 2;
    p.x_bytes = (uint32_t)std::min<long>(((long)(nb - 1) * xs + in_features) * 2, 0xffffffffL);
    int e;
    if (dtype == AQLM_HIP_F16)
      e = in_group_size == 8 ? launch_gemm<F16, 8>(p, g.nbt, stream) : launch_gemm<F16, 16>(p, g.nbt, stream);
    else
      e = in_group_size == 8 ? launch_gemm<BF16, 8>(p, g.nbt, stream) : launch_gemm<BF16, 16>(p, g.nbt, stream);
    if (e) return e;
    FinalizeParams f{};
    f.partial = (const float*)workspace;
    f.scales = (const uint16_t*)scales;
    f.bias = (const uint16_t*)bias;
    f.Y = (uint16_t*)Y + (long)b0 * ys;
    f.M = out_features;
    f.B = nb;
    f.Bpad = g.Bpad;
    f.ksplit = g.ksplit;
    f.ys = ys;
    dim3 grid((out_features + 31) / 32, g.nbt);
    if (dtype == AQLM_HIP_F16) hipLaunchKernelGGL(gemm_finalize_kernel<F16>, grid, dim3(256), 0, stream, f);
    else hipLaunchKernelGGL(gemm_finalize_kernel<BF16>, grid, dim3(256), 0, stream, f);
    if (int e2 = check_hip(hipGetLastError(), "gemm_finalize launch")) return e2;
  }
  return 0;
}
