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

  // global -> register staging of one X chunk (zero beyond B or K)
  u32x4 xr[PER_THREAD];
  auto load_x = [&](int chunk) {
    const int k0 = k_begin + chunk * BK;
#pragma unroll
    for (int s = 0; s < PER_THREAD; ++s) {
      const int q = tid + s * 256;
      const int b = q >> 3, c = q & 7;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (q < PIECES && b < p.B && k0 + c * 8 < k_end) v = *reinterpret_cast<const u32x4*>(p.X + (long)b * p.xs + k0 + c * 8);
      xr[s] = v;
    }
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int s = 0; s < PER_THREAD; ++s) {
      const int q = tid + s * 256;
      if (q < PIECES) xl[buf][xswz(q >> 3, q & 7)] = xr[s];
    }
  };
  auto load_codes = [&](int chunk, uint32_t (&cw)[CWN]) {
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

  uint32_t cw_next[CWN];
  load_x(0);
  load_codes(0, cw_next);
  store_x(0);
  __syncthreads();

  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    uint32_t cw[CWN];
#pragma unroll
    for (int k = 0; k < CWN; ++k) cw[k] = cw_next[k];
    const bool more = ch + 1 < nchunks;
    if (more) {
      load_x(ch + 1);
      load_codes(ch + 1, cw_next);
    }
    // 4 k-steps of 16: one gather per lane per step; entry == MFMA A fragment
    u32x4 afrag[4];
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
      afrag[kk] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, code * (uint32_t)(G * 2) + piece * 16, 0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int t = 0; t < NBT; ++t) {
        const u32x4 bfrag = xl[buf][xswz(t * 32 + (lane & 31), kk * 2 + half)];
        acc[t] = mfma32<T>(afrag[kk], bfrag, acc[t]);
      }
    }
    if (more) store_x(buf ^ 1);
    __syncthreads();
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
    if (m < p.M && b < p.Bpad)
      for (int k = 0; k < p.ksplit; ++k) s += p.partial[((long)k * p.M + m) * p.Bpad + b];
    tile[ty + i * 8][tx] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = b0 + ty + i * 8, m = m0 + tx;
    if (m < p.M && b < p.B) {
      const float scale = T::to_float(p.scales[m]);
      const float bias = p.bias ? T::to_float(p.bias[m]) : 0.f;
      p.Y[(long)b * p.ys + m] = T::from_float(tile[tx][ty + i * 8] * scale + bias);
    }
  }
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

extern "C" size_t aqlm_hip_workspace_bytes(int op, int batch, int out_features, int in_features) {
  if (batch <= 0 || out_features <= 0 || in_features <= 0) return 0;
  if (op == AQLM_HIP_OP_GEMV_1X16_LDS || op == AQLM_HIP_OP_GEMV_1X16_PACKED) return (size_t)8 * out_features * sizeof(float);
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
  for (int b0 = 0; b0 < batch; b0 += 128) {
    const int nb = std::min(128, batch - b0);
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
