// K x 8-bit (g = 8, K in {1, 2}) matvec with 16-fold replicated codebooks in LDS: conflict-free gathers.  gfx950.
//
// Why: in the plain K x 8 kernel (gemv.hip) every code is a random ds_read_b128 from a 4-8 KiB table.  A wave64
// ds_read_b128 is serviced in four fixed 16-lane groups; 16 random 16-B slots collide ~3-way, so the gather runs at
// ~5.5 lanes/clk/CU instead of 16 (measured through the packed 1x16 kernel's ablation).  With each entry
// stored 16 times -- copy r of entry v of codebook c at 16-B slot ((c*256 + v)*16 + r) -- and lane l reading copy
// r = l & 15, the 16 lanes of every service group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) touch 16 distinct
// slots of the 256-B bank row whatever the codes are.  This is the wave64 / 64-bank counterpart of the reference's
// 8-fold replication for 32-lane warps (cuda_kernel.cu:168-173), derived from MI355X's service groups.
//
// 2 x 256 x 16 x 16 B = 128 KiB (+ the x tile) fills the LDS, so the grid is one 16-wave workgroup per CU and every
// wave walks many rows (the 0.7 us replicated fill is paid once per CU, not once per 4 rows).  Rows are reduced four
// at a time (7 cross-lane steps per 4 rows).  Same contract and epilogue as gemv_kernel.
#include <algorithm>

#include "aqlm_common.h"

namespace aqlm {

struct RepParams {
  const uint8_t* codes;
  const uint8_t* codebooks;
  const uint16_t* scales;
  const uint16_t* bias;
  const uint16_t* x;
  uint16_t* y;
  int M, in_groups, nunits, iters, pitch;
  int rows_per_block;  // multiple of 4
  long code_row_bytes;
};

__device__ __forceinline__ float rep_reduce4(float a, float b, float c, float d, int lane) {
  const bool upper = lane >= 32;
  float p = upper ? a : c, q = upper ? b : d;
  p = __shfl_xor(p, 32, WAVE);
  q = __shfl_xor(q, 32, WAVE);
  const float u = (upper ? c : a) + p, v = (upper ? d : b) + q;
  const bool odd16 = (lane & 16) != 0;
  float w = odd16 ? u : v;
  w = __shfl_xor(w, 16, WAVE);
  float keep = (odd16 ? v : u) + w;
  keep = row16_sum(keep);
  return keep;  // lanes 0-15: a, 16-31: b, 32-47: c, 48-63: d
}

// `block` = the workgroup's index within its own code matrix (== blockIdx.x for a single-matrix launch)
// NQ = rows a wave owns per round: 4 (rows base + w + 16 q), or 1 when the workgroup has <= 16 rows (4096-row layers: no
// clamped duplicate requests, a quarter of the code-word registers).
template <class T, int KC, int ITERS, int NQ>
__device__ __forceinline__ void gemv_kx8_rep_body(const RepParams& p, const int block) {
  constexpr int NT = 1024;
  constexpr int UB = 8 * KC;       // code bytes per unit of 8 groups
  constexpr int CW = UB / 4;
  constexpr int ENTRIES = KC * 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* const cbl = reinterpret_cast<u32x4*>(smem_raw);   // [ENTRIES][16 replicas]
  u32x4* const xl = cbl + ENTRIES * 16;                      // [8][pitch]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row_begin = block * p.rows_per_block;
  int nrows = p.M - row_begin;
  nrows = nrows < 0 ? 0 : (nrows < p.rows_per_block ? nrows : p.rows_per_block);

  // Rows are dealt round-robin: in round `base`, wave w owns rows base + w + 16 q (q = 0..3).  All code words of the
  // wave's (up to 4 x ITERS) row pieces are requested up front with clamped, always-valid addresses, so a wave pays ONE
  // HBM latency per round, not one per row.
  auto load_round = [&](int base, uint32_t (&cwq)[NQ][ITERS][CW]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      int r = base + wave + 16 * q;
      r = r < nrows ? r : (nrows > 0 ? nrows - 1 : 0);
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        int u = it * 64 + lane;
        u = u < p.nunits ? u : p.nunits - 1;
        const uint8_t* ptr = p.codes + ((long)row_begin + r) * p.code_row_bytes + (long)u * UB;
        if constexpr (CW == 4) {
          const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr));
          cwq[q][it][0] = v.x; cwq[q][it][1] = v.y; cwq[q][it][2] = v.z; cwq[q][it][3] = v.w;
        } else {
          const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(ptr));
          cwq[q][it][0] = v.x; cwq[q][it][1] = v.y;
        }
      }
    }
  };

  // scale and bias of the row a lane will write in round `base` (lanes with lane % 16 == 0 do): requested with the round's
  // code words instead of behind its last cross-lane step
  uint16_t scale_h = 0, bias_h = 0;
  auto load_epilogue = [&](int base) {
    int r = base + wave + 16 * (lane >> 4);
    r = r < nrows ? r : (nrows > 0 ? nrows - 1 : 0);
    const uint16_t* bp = p.bias ? p.bias : p.scales;  // branch-free: a conditional load would end in a vmcnt(0) at the join
    scale_h = p.scales[row_begin + r];
    bias_h = bp[row_begin + r];
    bias_h = p.bias ? bias_h : (uint16_t)0;
  };

  // Every global load of the prologue is issued before the first LDS write: codebook entries, x, and the code words of
  // the first round (for <= 64 x 256 rows the only one).  Loads return in order, so the LDS fill proceeds as the codebook
  // arrives while the code words -- the one HBM latency of the kernel -- are already on their way; issued behind the
  // barrier (as in round 2) their latency was paid again on top of the fill's.
  // Replicated fill: thread t writes copy r = t & 15 of entries (t >> 4) + 64 k.  The 8 lanes of a ds_write_b128 service
  // group then hit 8 distinct 16-B slots (conflict free); the 16 threads of an entry read the same 16 B from L2 (one
  // request).  x (in_groups <= 1536 < 2 NT): at most two vectors per thread.
  // Measured on one box (profiles/r03_mb_kx8_hoisted_loads.log): with one row per wave (NQ = 1: 4096-row layers) the early
  // request gains 4 % (4096 -> 4096: 5.11 -> 4.90 us) to 10 % (11008 -> 4096: 8.6 -> 7.7 us); with four rows per wave
  // (4096 -> 11008) it LOSES 6 % (7.08 -> 7.49 us), so those keep requesting their code words behind the barrier.
  constexpr bool HOIST = NQ == 1;
  uint32_t cwq[NQ][ITERS][CW];
  {
    const int r = tid & 15;
    const u32x4* src = reinterpret_cast<const u32x4*>(p.codebooks);
    u32x4 v[ENTRIES / 64], xv[2];
#pragma unroll
    for (int k = 0; k < ENTRIES / 64; ++k) v[k] = src[(tid >> 4) + 64 * k];
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // guarded load HERE, guarded store below: two separate branches, so hipcc cannot sink the
      xv[k] = u32x4{0u, 0u, 0u, 0u};  // load into the store's block (behind the code words, where it would wait for them
      const int q = tid + k * NT;     // with vmcnt(0)); an unguarded store of a clamped address serialises hundreds of
      if (q < p.in_groups) xv[k] = *reinterpret_cast<const u32x4*>(p.x + (long)q * 8);  // threads on one LDS slot (+0.3 us)
    }
    __builtin_amdgcn_sched_barrier(0);  // keep this order: loads return in order, and the fill must not wait for the codes
    if constexpr (HOIST) {
      load_round(0, cwq);
      load_epilogue(0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < ENTRIES / 64; ++k) cbl[((tid >> 4) + 64 * k) * 16 + r] = v[k];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + k * NT;
      if (q < p.in_groups) xl[(q & 7) * p.pitch + (q >> 3)] = xv[k];
    }
  }
  __syncthreads();

  const uint32_t rep_off = (uint32_t)(lane & 15) << 4;  // this lane's replica: conflict-free in every service group
  const unsigned char* const cb_bytes = reinterpret_cast<const unsigned char*>(cbl);

  for (int base = 0; base < nrows; base += 16 * NQ) {
    if (!HOIST || base > 0) {
      load_round(base, cwq);
      load_epilogue(base);
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int r = base + wave + 16 * q;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int u = it * 64 + lane;
        if (r < nrows && u < p.nunits) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const u32x4 xv = xl[i * p.pitch + u];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
              const uint32_t code = code_at<1>(cwq[q][it], i * KC + c);
              const u32x4 e = *reinterpret_cast<const u32x4*>(cb_bytes + ((((uint32_t)c << 8) + code) << 8) + rep_off);
              acc[q] = dot8<T>(e, xv, acc[q]);
            }
          }
        }
      }
    }
    const float tot = rep_reduce4(acc[0], acc[1], acc[2], acc[3], lane);
    const int r = base + wave + 16 * (lane >> 4);
    if ((lane & 15) == 0 && (lane >> 4) < NQ && r < nrows) {
      const int row = row_begin + r;
      p.y[row] = T::from_float(__builtin_fmaf(tot, T::to_float(scale_h), T::to_float(bias_h)));
    }
  }
}

// leading scalar arguments: preloaded into SGPRs at wave launch (see gemv.hip); the epilogue's pointers stay in a struct
struct RepRest {
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
};

template <class T, int KC, int ITERS, int NQ>
__global__ __launch_bounds__(1024) void gemv_kx8_rep_kernel(const uint8_t* codes, const uint8_t* codebooks, const uint16_t* x, int M,
                                                            int in_groups, int nunits, int iters, int pitch, int rows_per_block,
                                                            long code_row_bytes, const RepRest rest) {
  RepParams p;
  p.codes = codes;
  p.codebooks = codebooks;
  p.scales = rest.scales;
  p.bias = rest.bias;
  p.x = x;
  p.y = rest.y;
  p.M = M;
  p.in_groups = in_groups;
  p.nunits = nunits;
  p.iters = iters;
  p.pitch = pitch;
  p.rows_per_block = rows_per_block;
  p.code_row_bytes = code_row_bytes;
  gemv_kx8_rep_body<T, KC, ITERS, NQ>(p, blockIdx.x);
}

// Shared-input launch: up to AQLM_HIP_MAX_SEGMENTS code matrices (own codebooks / scales / bias / y) times one x.
// Every workgroup belongs to one segment and fills LDS with that segment's replicated codebooks.
struct RepSegment {
  const uint8_t* codes;
  const uint8_t* codebooks;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
  int M, rows_per_block, block_begin;
};

struct RepMultiParams {
  RepParams common;
  int nseg;
  RepSegment seg[AQLM_HIP_MAX_SEGMENTS];
};

template <class T, int KC, int ITERS, int NQ>
__global__ __launch_bounds__(1024) void gemv_kx8_rep_multi_kernel(const RepMultiParams mp) {
  RepParams p = mp.common;
  int begin = 0;
#pragma unroll
  for (int k = 0; k < AQLM_HIP_MAX_SEGMENTS; ++k) {
    if (k == 0 || (k < mp.nseg && (int)blockIdx.x >= mp.seg[k].block_begin)) {  // scalar select chain
      p.codes = mp.seg[k].codes;
      p.codebooks = mp.seg[k].codebooks;
      p.scales = mp.seg[k].scales;
      p.bias = mp.seg[k].bias;
      p.y = mp.seg[k].y;
      p.M = mp.seg[k].M;
      p.rows_per_block = mp.seg[k].rows_per_block;
      begin = mp.seg[k].block_begin;
    }
  }
  gemv_kx8_rep_body<T, KC, ITERS, NQ>(p, (int)blockIdx.x - begin);
}

template <class T, int KC, int ITERS, int NQ>
static int launch_rep_multi_q(const RepMultiParams& mp, int blocks, hipStream_t stream) {
  auto kern = gemv_kx8_rep_multi_kernel<T, KC, ITERS, NQ>;
  const size_t lds = (size_t)KC * 256 * 16 * 16 + (size_t)8 * mp.common.pitch * 16;
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, stream, mp);
  return check_hip(hipGetLastError(), "gemv_kx8_rep_multi launch");
}

template <class T, int KC, int ITERS>
static int launch_rep_multi_i(const RepMultiParams& mp, int blocks, hipStream_t stream) {
  int max_rows = 0;
  for (int k = 0; k < mp.nseg; ++k) max_rows = std::max(max_rows, mp.seg[k].rows_per_block);
  return max_rows <= 16 ? launch_rep_multi_q<T, KC, ITERS, 1>(mp, blocks, stream) : launch_rep_multi_q<T, KC, ITERS, 4>(mp, blocks, stream);
}

template <class T, int KC>
static int launch_rep_multi(const RepMultiParams& mp, int blocks, hipStream_t stream) {
  switch (mp.common.iters) {
    case 1: return launch_rep_multi_i<T, KC, 1>(mp, blocks, stream);
    case 2: return launch_rep_multi_i<T, KC, 2>(mp, blocks, stream);
    default: return launch_rep_multi_i<T, KC, 3>(mp, blocks, stream);
  }
}

template <class T, int KC, int ITERS, int NQ>
static int launch_rep_q(const RepParams& p, int blocks, hipStream_t stream) {
  auto kern = gemv_kx8_rep_kernel<T, KC, ITERS, NQ>;
  const size_t lds = (size_t)KC * 256 * 16 * 16 + (size_t)8 * p.pitch * 16;
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  RepRest rest{p.scales, p.bias, p.y};
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, stream, p.codes, p.codebooks, p.x, p.M, p.in_groups, p.nunits, p.iters,
                     p.pitch, p.rows_per_block, p.code_row_bytes, rest);
  return check_hip(hipGetLastError(), "gemv_kx8_rep launch");
}

template <class T, int KC, int ITERS>
static int launch_rep_i(const RepParams& p, int blocks, hipStream_t stream) {
  return p.rows_per_block <= 16 ? launch_rep_q<T, KC, ITERS, 1>(p, blocks, stream) : launch_rep_q<T, KC, ITERS, 4>(p, blocks, stream);
}

template <class T, int KC>
static int launch_rep(const RepParams& p, int blocks, hipStream_t stream) {
  switch (p.iters) {
    case 1: return launch_rep_i<T, KC, 1>(p, blocks, stream);
    case 2: return launch_rep_i<T, KC, 2>(p, blocks, stream);
    default: return launch_rep_i<T, KC, 3>(p, blocks, stream);
  }
}

// Used by aqlm_hip_gemv_kx8 for batch 1, g = 8, K in {1,2} and enough rows to amortise the replicated fill.
// Returns AQLM_HIP_E_UNSUPPORTED when the shape does not fit (caller falls back to gemv_kernel).
int gemv_kx8_replicated(const void* codes, const void* codebooks, const void* scales, const void* bias, const void* x,
                        void* y, int out_features, int in_features, int num_codebooks, int dtype, hipStream_t stream) {
  const int in_groups = in_features / 8;
  if ((num_codebooks != 1 && num_codebooks != 2) || in_groups % 8 != 0) return AQLM_HIP_E_UNSUPPORTED;
  RepParams p{};
  p.codes = (const uint8_t*)codes;
  p.codebooks = (const uint8_t*)codebooks;
  p.scales = (const uint16_t*)scales;
  p.bias = (const uint16_t*)bias;
  p.x = (const uint16_t*)x;
  p.y = (uint16_t*)y;
  p.M = out_features;
  p.in_groups = in_groups;
  p.nunits = in_groups / 8;
  p.iters = (p.nunits + 63) / 64;
  p.pitch = p.nunits | 1;
  p.code_row_bytes = (long)in_groups * num_codebooks;
  const size_t lds = (size_t)num_codebooks * 256 * 16 * 16 + (size_t)8 * p.pitch * 16;
  if (lds > 160 * 1024 || p.iters > 3) return AQLM_HIP_E_UNSUPPORTED;  // in_features <= 12288
  p.rows_per_block = ((out_features + 255) / 256 + 3) / 4 * 4;
  const int blocks = (out_features + p.rows_per_block - 1) / p.rows_per_block;
  if (dtype == AQLM_HIP_F16)
    return num_codebooks == 1 ? launch_rep<F16, 1>(p, blocks, stream) : launch_rep<F16, 2>(p, blocks, stream);
  return num_codebooks == 1 ? launch_rep<BF16, 1>(p, blocks, stream) : launch_rep<BF16, 2>(p, blocks, stream);
}

// Shared-input variant (batch 1, g = 8, K in {1,2}): ~256 workgroups dealt to the segments in proportion to their rows.
// Returns AQLM_HIP_E_UNSUPPORTED when the shape does not fit (caller falls back to the plain LDS kernel).
int gemv_kx8_replicated_multi(const aqlm_hip_segment* segments, int num_segments, const void* x, int in_features,
                              int num_codebooks, int dtype, hipStream_t stream) {
  const int in_groups = in_features / 8;
  if ((num_codebooks != 1 && num_codebooks != 2) || in_groups % 8 != 0) return AQLM_HIP_E_UNSUPPORTED;
  RepMultiParams mp{};
  RepParams& p = mp.common;
  p.x = (const uint16_t*)x;
  p.in_groups = in_groups;
  p.nunits = in_groups / 8;
  p.iters = (p.nunits + 63) / 64;
  p.pitch = p.nunits | 1;
  p.code_row_bytes = (long)in_groups * num_codebooks;
  const size_t lds = (size_t)num_codebooks * 256 * 16 * 16 + (size_t)8 * p.pitch * 16;
  if (lds > 160 * 1024 || p.iters > 3) return AQLM_HIP_E_UNSUPPORTED;
  long total = 0;
  for (int k = 0; k < num_segments; ++k) total += segments[k].out_features;
  mp.nseg = num_segments;
  int blocks = 0;
  for (int k = 0; k < num_segments; ++k) {
    const aqlm_hip_segment& sg = segments[k];
    RepSegment& rs = mp.seg[k];
    rs.codes = (const uint8_t*)sg.codes;
    rs.codebooks = (const uint8_t*)sg.codebook;
    rs.scales = (const uint16_t*)sg.scales;
    rs.bias = (const uint16_t*)sg.bias;
    rs.y = (uint16_t*)sg.y;
    rs.M = sg.out_features;
    const int share = std::max(1, (int)((256L * sg.out_features + total / 2) / total));  // workgroups for this segment
    rs.rows_per_block = ((sg.out_features + share - 1) / share + 3) / 4 * 4;
    rs.block_begin = blocks;
    blocks += (sg.out_features + rs.rows_per_block - 1) / rs.rows_per_block;
  }
  if (dtype == AQLM_HIP_F16)
    return num_codebooks == 1 ? launch_rep_multi<F16, 1>(mp, blocks, stream) : launch_rep_multi<F16, 2>(mp, blocks, stream);
  return num_codebooks == 1 ? launch_rep_multi<BF16, 1>(mp, blocks, stream) : launch_rep_multi<BF16, 2>(mp, blocks, stream);
}

}  // namespace aqlm
