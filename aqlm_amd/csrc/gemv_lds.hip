// 1x16 matvec with the codebook resident in LDS as per-CU slices ("slice-scan" kernel), gfx950.
//
// Why: measured on MI355X (profiles/r01_call1_mb_l2gather.log) a random 16-B gather that hits L2 costs a whole
// 128-B line of the CU's 64 B/clk L1-fill path: 0.43 lane-gathers/clk/CU = 266 G gathers/s chip-wide, whatever the
// cache policy.  That caps the direct kernel (gemv.hip) at ~8 us for a 4096x4096 layer (5 MB of algorithmic bytes,
// 0.66 us at HBM speed).  LDS serves a 16-B gather at >10x that rate, but only 160 KiB fit per CU while the
// codebook is 1 MiB.  So:
//
//   * the 65536-entry codebook is cut into S = 8 slices of 8192 entries (128 KiB) by the top 3 bits of the code;
//   * the grid is exactly 256 workgroups (one per CU): 32 row-groups x 8 slices.  Workgroup (g, s) holds slice s
//     in LDS and SCANS every code of the rows of group g, but only acts on codes whose top bits equal s
//     (1 in 8 on average): ds_read_b128 of the entry + 4 v_dot2c against x, which the lane holds in registers
//     (lane <-> input position is fixed, so x is row-invariant);
//   * the 8 workgroups of a group sit on one XCD (bid % 8 is the XCD: speed only, never correctness), so a code row
//     is fetched from HBM once and the other 7 scans hit L2 (coalesced 1 KiB wave loads: 16 clk of L1-fill each,
//     versus 512 x 2.3 clk for the direct gathers of the same row);
//   * each workgroup writes fp32 partial sums [slice][row] to a workspace; a second tiny kernel adds the 8 slices,
//     applies scale + bias and rounds once.  (Kernel boundary = the cheapest CORRECT cross-XCD hand-off to start
//     with; an in-kernel last-arriver reduction is the next step, DESIGN.md.)
//
// The scan is VALU-bound (~64 vector instructions per 512 codes per wave); cost per row is ~12-24x lower than the
// direct gathers, paid 8x (every slice scans every row) plus the 128 KiB fill and the finalize launch.
#include <algorithm>

#include "aqlm_common.h"

namespace aqlm {

struct LdsGemvParams {
  const uint8_t* codes;
  const uint8_t* codebook;
  const uint16_t* x;
  float* partial;  // [S][M]
  int M, in_groups, nunits, pitch;
  int rows_per_group;
  long code_row_bytes;
};

// four per-lane values -> lane group (lane >> 4) = k holds the wave total of value k in all of its 16 lanes' lane 0..15
__device__ __forceinline__ float wave_reduce4(float a, float b, float c, float d, int lane) {
  const bool upper = lane >= 32;
  float p = upper ? a : c, q = upper ? b : d;  // what the other half keeps
  p = __shfl_xor(p, 32, WAVE);
  q = __shfl_xor(q, 32, WAVE);
  float u = (upper ? c : a) + p, v = (upper ? d : b) + q;
  const bool odd16 = (lane & 16) != 0;
  float w = odd16 ? u : v;
  w = __shfl_xor(w, 16, WAVE);
  float keep = (odd16 ? v : u) + w;
  keep = row16_sum(keep);
  return keep;  // lanes 0-15: a, 16-31: b, 32-47: c, 48-63: d
}

// LDS map (16-B units): [0, 8192) codebook slice | 8192: one all-zero entry | then x as [i][unit] rows of `pitch` |
// then rows_per_group floats of per-row accumulators.
template <class T, int ITERS, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void gemv_1x16_lds_kernel(const LdsGemvParams p) {
  constexpr int NT = NWAVES * 64;
  constexpr int SLICE_ENTRIES = 8192;              // 128 KiB of 16-B entries
  constexpr int NWS = NWAVES / ITERS;              // waves that share one `it`
  static_assert(NWAVES % ITERS == 0, "waves must split evenly over the iterations");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* const cbl = reinterpret_cast<u32x4*>(smem_raw);
  u32x4* const xl = cbl + SLICE_ENTRIES + 1;
  float* const rowacc = reinterpret_cast<float*>(xl + 8 * p.pitch);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> row arithmetic on the SALU
  // 32 groups x 8 slices; the 8 slices of a group share bid % 8 (observed: the XCD)
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int slice = local & 7;
  const int group = xcd * 4 + (local >> 3);
  const int row_begin = group * p.rows_per_group;
  int row_end = row_begin + p.rows_per_group;
  row_end = row_end < p.M ? row_end : p.M;
  const int nrows = row_end > row_begin ? row_end - row_begin : 0;
  const int nbatches = (nrows + 3) / 4;

  // this wave owns input units [it*64, it*64+64) of every row it touches -> its x slice is fixed
  const int it = wave % ITERS, wsub = wave / ITERS;
  const int u = it * 64 + lane;
  const bool lane_ok = u < p.nunits;
  const uint32_t pat = ((uint32_t)slice << 13) | ((uint32_t)slice << 29);
  const u32x4 never = {~pat, ~pat, ~pat, ~pat};  // code word whose halves match no slice-`slice` code

  // lane's code pointer for (batch, q) = base + (batch*4 + q) * row_bytes, advanced incrementally (no 64-bit multiplies)
  const uint8_t* const lane_base = p.codes + (long)row_begin * p.code_row_bytes + (long)u * 16;
  auto load_item = [&](int batch, int q) -> u32x4 {
    const int r = batch * 4 + q;
    if (r >= nrows || !lane_ok) return never;
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(lane_base + (uint32_t)r * (uint32_t)p.code_row_bytes));
  };
  // (1) code prefetch ring: the four rows of this wave's first batch go in flight before the LDS fill
  u32x4 ring[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ring[q] = load_item(wsub, q);

  // (2) x -> LDS in the [i][unit] layout, zero entry, row accumulators, (3) codebook slice -> LDS
  for (int q = tid; q < p.in_groups; q += NT) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(p.x + (long)q * 8);
    xl[(q & 7) * p.pitch + (q >> 3)] = v;
  }
  if (tid == 0) cbl[SLICE_ENTRIES] = u32x4{0u, 0u, 0u, 0u};
  for (int q = tid; q < p.rows_per_group; q += NT) rowacc[q] = 0.f;
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.codebook) + (long)slice * SLICE_ENTRIES;
    for (int q = tid; q < SLICE_ENTRIES; q += NT) cbl[q] = src[q];
  }
  __syncthreads();

  u32x4 xr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) xr[i] = lane_ok ? xl[i * p.pitch + u] : u32x4{0u, 0u, 0u, 0u};

  const unsigned char* const cb_bytes = reinterpret_cast<const unsigned char*>(cbl);
  for (int batch = wsub; batch < nbatches; batch += NWS) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32x4 cwv = ring[q];
      ring[q] = load_item(batch + NWS, q);
      const uint32_t cw[4] = {cwv.x, cwv.y, cwv.z, cwv.w};
      // Exec-masked LDS gathers: only lanes whose code belongs to this slice read (about 1 lane in 8) and only they
      // accumulate.  Phase 1 issues all eight masked reads, phase 2 consumes them under the same masks (one wait).
      // (Letting non-matching lanes read a shared zero entry instead costs ~70 clk per ds_read_b128: measured.)
      bool m[8];
      uint32_t off[8];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t t = cw[d] ^ pat;
        m[2 * d] = (t & 0xE000u) == 0u;
        off[2 * d] = (t & 0x1FFFu) << 4;
        m[2 * d + 1] = t < 0x20000000u;
        off[2 * d + 1] = (t >> 12) & 0x1FFF0u;
      }
      u32x4 e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (m[i]) e[i] = *reinterpret_cast<const u32x4*>(cb_bytes + off[i]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (m[i]) acc[q] = dot8<T>(e[i], xr[i], acc[q]);
      __builtin_amdgcn_sched_barrier(0);
    }
    const float tot = wave_reduce4(acc[0], acc[1], acc[2], acc[3], lane);
    const int r = batch * 4 + (lane >> 4);
    if ((lane & 15) == 0 && r < nrows) {
      if constexpr (ITERS == 1) rowacc[r] = tot;   // single writer per row
      else atomicAdd(&rowacc[r], tot);             // ds_add_f32: ITERS waves contribute to a row
    }
  }
  __syncthreads();
  for (int r = tid; r < nrows; r += NT) p.partial[(long)slice * p.M + row_begin + r] = rowacc[r];
}

struct LdsFinalizeParams {
  const float* partial;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
  int M;
};

template <class T>
__global__ __launch_bounds__(256) void gemv_1x16_lds_finalize(const LdsFinalizeParams p) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= p.M) return;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += p.partial[(long)k * p.M + row];
  const float scale = T::to_float(p.scales[row]);
  const float bias = p.bias ? T::to_float(p.bias[row]) : 0.f;
  p.y[row] = T::from_float(__builtin_fmaf(s, scale, bias));
}

template <class T, int ITERS, int NWAVES>
static int launch_lds(const LdsGemvParams& p, hipStream_t stream) {
  auto kern = gemv_1x16_lds_kernel<T, ITERS, NWAVES>;
  const size_t lds = (size_t)(8192 + 1 + 8 * p.pitch) * 16 + (size_t)p.rows_per_group * 4;
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  hipLaunchKernelGGL(kern, dim3(256), dim3(NWAVES * 64), lds, stream, p);
  return check_hip(hipGetLastError(), "gemv_1x16_lds launch");
}

}  // namespace aqlm

using namespace aqlm;

static size_t aqlm_gemv_1x16_lds_workspace(int out_features) { return (size_t)8 * out_features * sizeof(float); }

extern "C" int aqlm_hip_gemv_1x16_lds(const void* codes, const void* codebook, const void* scales, const void* bias,
                                      const void* x, void* y, int out_features, int in_features, int in_group_size,
                                      int dtype, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!codes || !codebook || !scales || !x || !y) {
    set_last_error("aqlm_hip_gemv_1x16_lds: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (out_features <= 0 || in_features <= 0) {
    set_last_error("aqlm_hip_gemv_1x16_lds: sizes must be positive");
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_gemv_1x16_lds: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (in_group_size != 8 || in_features % 64 != 0 || in_features > 14336 || !aligned16(codes) ||
      !aligned16(codebook) || !aligned16(x)) {
    set_last_error("aqlm_hip_gemv_1x16_lds: needs in_group_size 8, in_features %% 64 == 0 and <= 14336, 16-B aligned "
                   "buffers (got g=%d in=%d); use aqlm_hip_gemv_1x16", in_group_size, in_features);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const size_t need = aqlm_gemv_1x16_lds_workspace(out_features);
  if (!workspace || workspace_bytes < need) {
    set_last_error("aqlm_hip_gemv_1x16_lds: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return AQLM_HIP_E_INVALID;
  }
  LdsGemvParams p{};
  p.codes = (const uint8_t*)codes;
  p.codebook = (const uint8_t*)codebook;
  p.x = (const uint16_t*)x;
  p.partial = (float*)workspace;
  p.M = out_features;
  p.in_groups = in_features / 8;
  p.nunits = p.in_groups / 8;
  p.pitch = p.nunits | 1;
  p.rows_per_group = ((out_features + 31) / 32 + 3) / 4 * 4;  // multiple of 4: batches never straddle groups
  p.code_row_bytes = (long)p.in_groups * 2;
  const int iters = (p.nunits + 63) / 64;
  int e;
  if (dtype == AQLM_HIP_F16) {
    switch (iters) {
      case 1: e = launch_lds<F16, 1, 16>(p, stream); break;
      case 2: e = launch_lds<F16, 2, 16>(p, stream); break;
      case 3: e = launch_lds<F16, 3, 15>(p, stream); break;
      default: e = launch_lds<F16, 4, 16>(p, stream); break;
    }
  } else {
    switch (iters) {
      case 1: e = launch_lds<BF16, 1, 16>(p, stream); break;
      case 2: e = launch_lds<BF16, 2, 16>(p, stream); break;
      case 3: e = launch_lds<BF16, 3, 15>(p, stream); break;
      default: e = launch_lds<BF16, 4, 16>(p, stream); break;
    }
  }
  if (e) return e;
  LdsFinalizeParams f{};
  f.partial = (const float*)workspace;
  f.scales = (const uint16_t*)scales;
  f.bias = (const uint16_t*)bias;
  f.y = (uint16_t*)y;
  f.M = out_features;
  if (dtype == AQLM_HIP_F16)
    hipLaunchKernelGGL(gemv_1x16_lds_finalize<F16>, dim3((out_features + 255) / 256), dim3(256), 0, stream, f);
  else
    hipLaunchKernelGGL(gemv_1x16_lds_finalize<BF16>, dim3((out_features + 255) / 256), dim3(256), 0, stream, f);
  return check_hip(hipGetLastError(), "gemv_1x16_lds_finalize launch");
}
