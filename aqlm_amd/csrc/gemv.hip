// AQLM dequant-fused matvec for gfx950 (MI355X), wave64, fp32 accumulation, fused scale+bias epilogue.
//
// Replaces (behaviour, not code): Code1x16MatVec / Code2x8MatVec / CodeKx8MatVec and their launchers
// (reference inference_lib/src/aqlm/inference_kernels/cuda_kernel.cu:7-95, 144-233, 296-390, 476-521, 555-620,
// 709-758), the per-row host loops + epilogue launches of code*_matmat (cuda_kernel.cpp:95-111, 148-182,
// 387-421, 552-586) and the Triton generic gemv (triton_kernel.py:30-205).
//
// Design (DESIGN.md section "gemv"):
//   * one wave64 per output row at a time; a lane owns one "unit" = U consecutive input groups whose codes
//     are 8 or 16 contiguous bytes, so a wave reads 512-1024 B of one code row per instruction (coalesced,
//     non-temporal: every code byte is used exactly once);
//   * x (all NB batch rows) is staged once per block in LDS in a [b][i][piece][unit] layout, so lane u's
//     ds_read_b128 of (b,i,piece) is at consecutive 16-B slots across lanes -> bank-conflict free;
//   * 1x16: the 1-2 MiB codebook cannot live in LDS; every code is a 16/32-B gather from the XCD's L2 through
//     a buffer descriptor (32-bit offsets, selectable cache policy).  This is the binding resource
//     (one 64-B L2 sector per 16 useful bytes, ~1 lane/clk/CU through the texture-addresser);
//   * Kx8: the 256-entry codebooks live in LDS and are gathered with ds_read_b128;
//   * all NB batch rows are produced by ONE launch: codes and gathered codebook vectors are reused across
//     the rows of x (the reference relaunches the matvec per row, cuda_kernel.cpp:165-175);
//   * y = acc * scales[row] + bias[row] in fp32, rounded once (reference: 3-4 extra launches).
#include <algorithm>

#include "aqlm_common.h"

namespace aqlm {

struct GemvParams {
  const uint8_t* codes;
  const uint8_t* codebooks;
  const uint16_t* scales;
  const uint16_t* bias;  // nullable
  const uint16_t* x;
  uint16_t* y;
  int M;              // out_features
  int in_groups;      // in_features / G
  int nunits;         // in_groups / U
  int iters;          // ceil(nunits / 64)
  int pitch;          // LDS pitch (in 16-B pieces) of one (b,i,piece) row of the x tile
  int rpw;            // rows per wave
  int prefetch;       // 1x16 only: warm the XCD's L2 with a slice of the codebook first
  int cb_bytes;       // total codebook bytes
  long xs, ys;        // row strides of x / y in elements
  long code_row_bytes;
};

template <int N>
__device__ __forceinline__ void load_code_word(const uint8_t* p, uint32_t (&cw)[N]) {
  if constexpr (N == 4) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    cw[0] = v.x; cw[1] = v.y; cw[2] = v.z; cw[3] = v.w;
  } else {
    static_assert(N == 2, "code word is 8 or 16 bytes");
    const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    cw[0] = v.x; cw[1] = v.y;
  }
}

// T: F16/BF16; CODE_BYTES: 1|2; KC: codebooks; G: in_group_size; U: groups per lane-unit; NB: batch rows;
// CB_LDS: codebooks in LDS (Kx8) or L2 gathers (1x16); NWAVES: waves per block; AUX: gather cache policy.
// `block` is the workgroup's index within its own code matrix (== blockIdx.x for a single-matrix launch).
template <class T, int CODE_BYTES, int KC, int G, int U, int NB, bool CB_LDS, int NWAVES, int AUX>
__device__ __forceinline__ void gemv_body(const GemvParams& p, const int block) {
  constexpr int P = G / 8;                  // 16-B pieces per codebook vector
  constexpr int UB = U * KC * CODE_BYTES;   // code bytes per unit
  constexpr int CW = UB / 4;
  static_assert(UB == 8 || UB == 16, "unit must be 8 or 16 code bytes");
  constexpr int NT = NWAVES * 64;
  constexpr int CB_SIZE = CB_LDS ? 256 : 65536;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* const xl = reinterpret_cast<u32x4*>(smem_raw);
  u32x4* const cbl = xl + NB * U * P * p.pitch;  // only touched when CB_LDS

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int row0 = (block * NWAVES + wave) * p.rpw;
  int nrows = p.M - row0;
  nrows = nrows < p.rpw ? nrows : p.rpw;

  // ---- prologue: every global load is issued before the first wait (no load-wait-store loops: hipcc puts a
  // vmcnt(0) in front of each ds_write of such a loop, i.e. one L2 round trip per iteration, ~1 us per kernel).
  // (a) x pieces, 8 per thread, staged in registers (clamped, always-valid addresses)
  const int pieces_per_row = p.in_groups * P;
  const int total_x = NB * pieces_per_row;
  auto x_piece = [&](int q) -> u32x4 {
    q = q < total_x ? q : total_x - 1;
    int b = 0, qq = q;
    if constexpr (NB > 1) { b = q / pieces_per_row; qq = q - b * pieces_per_row; }
    return *reinterpret_cast<const u32x4*>(p.x + (long)b * p.xs + (long)qq * 8);
  };
  auto x_store = [&](int q, const u32x4& v) {
    if (q < total_x) {
      int b = 0, qq = q;
      if constexpr (NB > 1) { b = q / pieces_per_row; qq = q - b * pieces_per_row; }
      const int j = qq / P, pp = qq % P;
      const int u = j / U, i = j % U;
      xl[((b * U + i) * P + pp) * p.pitch + u] = v;
    }
  };
  u32x4 xstage[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) xstage[k] = x_piece(tid + k * NT);

  // (b) scale / bias of the first row: unconditional loads (clamped row; bias pointer aliased to scales when absent)
  const uint16_t* const bias_src = p.bias ? p.bias : p.scales;
  const float bias_on = p.bias ? 1.f : 0.f;
  const int row_c0 = row0 < p.M ? row0 : p.M - 1;
  uint16_t scale_h = p.scales[row_c0], bias_h = bias_src[row_c0];

  // (c) first code word (HBM latency overlaps the LDS fill)
  uint32_t cw_next[CW];
#pragma unroll
  for (int k = 0; k < CW; ++k) cw_next[k] = 0;
  if (nrows > 0 && lane < p.nunits) load_code_word<CW>(p.codes + (long)row0 * p.code_row_bytes + (long)lane * UB, cw_next);

  // (d) codebooks (Kx8): staged the same way, compile-time trip count
  if constexpr (CB_LDS) {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.codebooks);
    constexpr int TOTAL = KC * 256 * P;
    constexpr int PER = (TOTAL + NT - 1) / NT;
    u32x4 cstage[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) cstage[k] = src[tid + k * NT < TOTAL ? tid + k * NT : TOTAL - 1];
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (tid + k * NT < TOTAL) cbl[tid + k * NT] = cstage[k];
  }
  // (e) x -> LDS: element (b, group j = u*U+i, piece pp) -> xl[((b*U+i)*P+pp)*pitch + u]
#pragma unroll
  for (int k = 0; k < 8; ++k) x_store(tid + k * NT, xstage[k]);
  for (int q0 = tid + 8 * NT; q0 < total_x; q0 += NT * 4) {  // rare: more than 8 pieces per thread (large batch x K)
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = x_piece(q0 + k * NT);
#pragma unroll
    for (int k = 0; k < 4; ++k) x_store(q0 + k * NT, v[k]);
  }

  // optional: touch a 16 KiB slice of the codebook so that this XCD's L2 is warm before the random gathers
  u32x4 pf[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pf[k] = u32x4{0u, 0u, 0u, 0u};
  if constexpr (!CB_LDS) {
    if (p.prefetch) {
      const int nchunks = p.cb_bytes >> 14;  // 16 KiB chunks
      const int chunk = (block >> 3) % (nchunks > 0 ? nchunks : 1);
      const u32x4* src = reinterpret_cast<const u32x4*>(p.codebooks + ((long)chunk << 14));
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (NT == 256 || tid < 256) pf[k] = src[k * 256 + (tid & 255)];
    }
  }
  __syncthreads();

  __amdgpu_buffer_rsrc_t rsrc;
  if constexpr (!CB_LDS) rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.codebooks, 0, p.cb_bytes, 0x00020000);

  for (int r = 0; r < nrows; ++r) {
    const int row = row0 + r;
    const float scale = T::to_float(scale_h);
    const float bias = T::to_float(bias_h) * bias_on;
    {
      const int rn = row + 1 < p.M ? row + 1 : p.M - 1;  // next row's epilogue operands (unused after the last row)
      scale_h = p.scales[rn];
      bias_h = bias_src[rn];
    }
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;

    for (int it = 0; it < p.iters; ++it) {
      const int u = it * 64 + lane;
      uint32_t cw[CW];
#pragma unroll
      for (int k = 0; k < CW; ++k) cw[k] = cw_next[k];
      // software prefetch of the next (row, iteration) code word
      {
        int nit = it + 1, nr = r;
        if (nit == p.iters) { nit = 0; nr = r + 1; }
        const int nu = nit * 64 + lane;
        if (nr < nrows && nu < p.nunits)
          load_code_word<CW>(p.codes + (long)(row0 + nr) * p.code_row_bytes + (long)nu * UB, cw_next);
      }
      if (u < p.nunits) {
        if constexpr (!CB_LDS) {
          // issue every gather of this unit before consuming any (8-16 x 16 B in flight per lane)
          u32x4 ent[U * KC * P];
#pragma unroll
          for (int i = 0; i < U; ++i)
#pragma unroll
            for (int c = 0; c < KC; ++c) {
              const uint32_t code = code_at<CODE_BYTES>(cw, i * KC + c);
              const uint32_t off = (uint32_t)(c * CB_SIZE + code) * (uint32_t)(G * 2);
#pragma unroll
              for (int pp = 0; pp < P; ++pp)
                ent[(i * KC + c) * P + pp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + pp * 16, 0, AUX);
            }
#pragma unroll
          for (int i = 0; i < U; ++i)
#pragma unroll
            for (int pp = 0; pp < P; ++pp)
#pragma unroll
              for (int b = 0; b < NB; ++b) {
                const u32x4 xv = xl[((b * U + i) * P + pp) * p.pitch + u];
#pragma unroll
                for (int c = 0; c < KC; ++c) acc[b] = dot8<T>(ent[(i * KC + c) * P + pp], xv, acc[b]);
              }
        } else {
#pragma unroll
          for (int i = 0; i < U; ++i)
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
              u32x4 xv[NB];
#pragma unroll
              for (int b = 0; b < NB; ++b) xv[b] = xl[((b * U + i) * P + pp) * p.pitch + u];
#pragma unroll
              for (int c = 0; c < KC; ++c) {
                const uint32_t code = code_at<CODE_BYTES>(cw, i * KC + c);
                const u32x4 e = cbl[(c * 256 + code) * P + pp];
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b] = dot8<T>(e, xv[b], acc[b]);
              }
            }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = wave_sum(acc[b]);
    if (lane == 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b) p.y[(long)b * p.ys + row] = T::from_float(__builtin_fmaf(acc[b], scale, bias));
    }
  }
  // keep the prefetch loads alive without ever waiting on them early
#pragma unroll
  for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(pf[k]));
}

// Leading scalar arguments = what the prologue needs to issue its first loads: with -amdgpu-kernarg-preload-count
// (Makefile) they arrive in SGPRs at wave launch instead of through an s_load round trip at the head of the kernel
// (struct arguments are not preloaded: the rest, used later, stays in one).
struct GemvRest {
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
  long ys;
  long code_row_bytes;
  int prefetch;
  int cb_bytes;
};

template <class T, int CODE_BYTES, int KC, int G, int U, int NB, bool CB_LDS, int NWAVES, int AUX>
__global__ __launch_bounds__(NWAVES * 64) void gemv_kernel(const uint8_t* codes, const uint8_t* codebooks, const uint16_t* x, int M,
                                                           int in_groups, int nunits, int iters, int pitch, int rpw, long xs,
                                                           const GemvRest rest) {
  GemvParams p;
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
  p.rpw = rpw;
  p.prefetch = rest.prefetch;
  p.cb_bytes = rest.cb_bytes;
  p.xs = xs;
  p.ys = rest.ys;
  p.code_row_bytes = rest.code_row_bytes;
  gemv_body<T, CODE_BYTES, KC, G, U, NB, CB_LDS, NWAVES, AUX>(p, blockIdx.x);
}

// Several code matrices that multiply the SAME x (q/k/v, gate/up of a decoder layer) in one launch: every workgroup
// belongs to one segment (own codes, codebook, scales, bias, y) and runs the unchanged body on it, so results are
// bit-identical to separate launches.  One launch instead of 2-3 removes the ~2 us dependent-launch gaps and lets the
// segments' ramp-up / tail overlap.
struct GemvSegment {
  const uint8_t* codes;
  const uint8_t* codebooks;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* y;
  long ys;
  int M;
  int block_begin;
};

struct GemvMultiParams {
  GemvParams common;  // x, geometry; per-segment fields are overwritten in the kernel
  int nseg;
  GemvSegment seg[AQLM_HIP_MAX_SEGMENTS];
};

template <class T, int CODE_BYTES, int KC, int G, int U, int NB, bool CB_LDS, int NWAVES, int AUX>
__global__ __launch_bounds__(NWAVES * 64) void gemv_multi_kernel(const GemvMultiParams mp) {
  GemvParams p = mp.common;
  int begin = 0;
  // scalar select chain (no dynamic indexing of the kernel-argument struct)
#pragma unroll
  for (int k = 0; k < AQLM_HIP_MAX_SEGMENTS; ++k) {
    if (k == 0 || (k < mp.nseg && (int)blockIdx.x >= mp.seg[k].block_begin)) {
      p.codes = mp.seg[k].codes;
      p.codebooks = mp.seg[k].codebooks;
      p.scales = mp.seg[k].scales;
      p.bias = mp.seg[k].bias;
      p.y = mp.seg[k].y;
      p.ys = mp.seg[k].ys;
      p.M = mp.seg[k].M;
      begin = mp.seg[k].block_begin;
    }
  }
  gemv_body<T, CODE_BYTES, KC, G, U, NB, CB_LDS, NWAVES, AUX>(p, (int)blockIdx.x - begin);
}

// ---------------------------------------------------------------------------------------------
// generic kernel: any scheme, any alignment.  One wave per row, lanes stride over input groups.
// The role triton_kernel.py plays in the reference (kernel_selector.py:91-94).
// ---------------------------------------------------------------------------------------------
struct GenericParams {
  const uint8_t* codes;
  const uint16_t* codebooks;
  const uint16_t* scales;
  const uint16_t* bias;
  const uint16_t* x;
  uint16_t* y;
  int M, in_groups, KC, nbits, G, code_bytes, batch;
  long xs, ys;
};

template <class T>
__global__ __launch_bounds__(256) void gemv_generic_kernel(const GenericParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const uint32_t mask = (1u << p.nbits) - 1u;
  const long cbsize = 1L << p.nbits;
  float acc[AQLM_HIP_MAX_GEMV_BATCH];
#pragma unroll
  for (int b = 0; b < AQLM_HIP_MAX_GEMV_BATCH; ++b) acc[b] = 0.f;
  for (int j = lane; j < p.in_groups; j += 64) {
    for (int c = 0; c < p.KC; ++c) {
      const long ci = ((long)row * p.in_groups + j) * p.KC + c;
      uint32_t code;
      if (p.code_bytes == 1) code = p.codes[ci];
      else code = reinterpret_cast<const uint16_t*>(p.codes)[ci];
      code &= mask;
      const uint16_t* e = p.codebooks + ((long)c * cbsize + code) * p.G;
      for (int t = 0; t < p.G; ++t) {
        const float w = T::to_float(e[t]);
#pragma unroll
        for (int b = 0; b < AQLM_HIP_MAX_GEMV_BATCH; ++b)
          if (b < p.batch) acc[b] += w * T::to_float(p.x[(long)b * p.xs + (long)j * p.G + t]);
      }
    }
  }
  const float scale = T::to_float(p.scales[row]);
  const float bias = p.bias ? T::to_float(p.bias[row]) : 0.f;
#pragma unroll
  for (int b = 0; b < AQLM_HIP_MAX_GEMV_BATCH; ++b) {
    if (b < p.batch) {
      const float s = wave_sum(acc[b]);
      if (lane == 0) p.y[(long)b * p.ys + row] = T::from_float(__builtin_fmaf(s, scale, bias));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static constexpr size_t kMaxXTileBytes = 64 * 1024;

template <class T, int CODE_BYTES, int KC, int G, int U, int NB, bool CB_LDS, int NWAVES, int AUX>
static int launch_gemv(const GemvParams& p, hipStream_t stream) {
  constexpr int P = G / 8;
  auto kern = gemv_kernel<T, CODE_BYTES, KC, G, U, NB, CB_LDS, NWAVES, AUX>;
  size_t lds = (size_t)NB * U * P * p.pitch * 16;
  if (CB_LDS) lds += (size_t)KC * 256 * G * 2;
  if (lds > 160 * 1024) {
    set_last_error("gemv: LDS request %zu B exceeds 160 KiB", lds);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  const int rows_per_block = NWAVES * p.rpw;
  const int blocks = (p.M + rows_per_block - 1) / rows_per_block;
  GemvRest rest{p.scales, p.bias, p.y, p.ys, p.code_row_bytes, p.prefetch, p.cb_bytes};
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(NWAVES * 64), lds, stream, p.codes, p.codebooks, p.x, p.M, p.in_groups, p.nunits,
                     p.iters, p.pitch, p.rpw, p.xs, rest);
  return check_hip(hipGetLastError(), "gemv launch");
}

template <class T, int CODE_BYTES, int KC, int G, int U, int NB, bool CB_LDS, int NWAVES, int AUX>
static int launch_gemv_multi(GemvMultiParams& mp, hipStream_t stream) {
  constexpr int P = G / 8;
  auto kern = gemv_multi_kernel<T, CODE_BYTES, KC, G, U, NB, CB_LDS, NWAVES, AUX>;
  size_t lds = (size_t)NB * U * P * mp.common.pitch * 16;
  if (CB_LDS) lds += (size_t)KC * 256 * G * 2;
  if (lds > 160 * 1024) {
    set_last_error("gemv_multi: LDS request %zu B exceeds 160 KiB", lds);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  const int rows_per_block = NWAVES * mp.common.rpw;
  int blocks = 0;
  for (int k = 0; k < mp.nseg; ++k) {
    mp.seg[k].block_begin = blocks;
    blocks += (mp.seg[k].M + rows_per_block - 1) / rows_per_block;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(NWAVES * 64), lds, stream, mp);
  return check_hip(hipGetLastError(), "gemv_multi launch");
}

template <int U>
static void finish_params(GemvParams& p, int in_groups, int out_features, int rows_hint_div) {
  p.in_groups = in_groups;
  p.nunits = in_groups / U;
  p.iters = (p.nunits + 63) / 64;
  p.pitch = p.nunits | 1;  // odd pitch: the x-tile fill (8 rows at one unit) does not hit one LDS slot
  int rpw = tuning().gemv_rows_per_wave;
  if (rpw <= 0) rpw = (out_features + rows_hint_div - 1) / rows_hint_div;
  p.rpw = std::max(1, std::min(rpw, 64));
}

// largest NB in {8,4,2,1} that is <= remaining and whose x tile fits
static int pick_nb(int remaining, size_t x_row_bytes) {
  for (int nb : {8, 4, 2, 1})
    if (nb <= remaining && (size_t)nb * x_row_bytes <= kMaxXTileBytes) return nb;
  return 0;
}

static int validate_common(const void* codes, const void* codebooks, const void* scales, const void* x, void* y,
                           int out_features, int in_features, int in_group_size, int batch, int dtype,
                           const char* who) {
  if (!codes || !codebooks || !scales || !x || !y) {
    set_last_error("%s: null pointer argument", who);
    return AQLM_HIP_E_INVALID;
  }
  if (out_features <= 0 || in_features <= 0 || in_group_size <= 0 || batch <= 0) {
    set_last_error("%s: sizes must be positive (out=%d in=%d g=%d batch=%d)", who, out_features, in_features,
                   in_group_size, batch);
    return AQLM_HIP_E_INVALID;
  }
  if (in_features % in_group_size != 0) {
    set_last_error("%s: in_features %d is not a multiple of in_group_size %d", who, in_features, in_group_size);
    return AQLM_HIP_E_INVALID;
  }
  if (batch > AQLM_HIP_MAX_GEMV_BATCH) {
    set_last_error("%s: batch %d > %d (use the gemm entry point)", who, batch, AQLM_HIP_MAX_GEMV_BATCH);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    // mirrors check_use_bfloat16, cuda_kernel.cpp:9-25
    set_last_error("%s: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", who, dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  return 0;
}

static int run_generic(const void* codes, const void* codebooks, const void* scales, const void* bias, const void* x,
                       void* y, int out_features, int in_features, int num_codebooks, int nbits, int in_group_size,
                       int batch, long xs, long ys, int dtype, hipStream_t stream) {
  GenericParams g;
  g.codes = (const uint8_t*)codes;
  g.codebooks = (const uint16_t*)codebooks;
  g.scales = (const uint16_t*)scales;
  g.bias = (const uint16_t*)bias;
  g.x = (const uint16_t*)x;
  g.y = (uint16_t*)y;
  g.M = out_features;
  g.in_groups = in_features / in_group_size;
  g.KC = num_codebooks;
  g.nbits = nbits;
  g.G = in_group_size;
  g.code_bytes = nbits <= 8 ? 1 : 2;
  g.batch = batch;
  g.xs = xs;
  g.ys = ys;
  const int blocks = (out_features + 3) / 4;
  if (dtype == AQLM_HIP_F16) hipLaunchKernelGGL(gemv_generic_kernel<F16>, dim3(blocks), dim3(256), 0, stream, g);
  else hipLaunchKernelGGL(gemv_generic_kernel<BF16>, dim3(blocks), dim3(256), 0, stream, g);
  return check_hip(hipGetLastError(), "gemv_generic launch");
}

// 1x16 --------------------------------------------------------------------------------------------
template <class T, int G, int NB>
static int launch_1x16(const GemvParams& p, hipStream_t s) {
  const int aux = tuning().gemv1x16_aux;
  if constexpr (NB == 1) {
    if (aux == AUX_NT) return launch_gemv<T, 2, 1, G, 8, NB, false, 4, AUX_NT>(p, s);
    if (aux == AUX_SC1) return launch_gemv<T, 2, 1, G, 8, NB, false, 4, AUX_SC1>(p, s);
  }
  return launch_gemv<T, 2, 1, G, 8, NB, false, 4, AUX_DEFAULT>(p, s);
}

template <class T, int G>
static int dispatch_1x16_multi_nb(int nb, GemvMultiParams& mp, hipStream_t s) {
  switch (nb) {
    case 1: return launch_gemv_multi<T, 2, 1, G, 8, 1, false, 4, AUX_DEFAULT>(mp, s);
    case 2: return launch_gemv_multi<T, 2, 1, G, 8, 2, false, 4, AUX_DEFAULT>(mp, s);
    case 4: return launch_gemv_multi<T, 2, 1, G, 8, 4, false, 4, AUX_DEFAULT>(mp, s);
    default: return launch_gemv_multi<T, 2, 1, G, 8, 8, false, 4, AUX_DEFAULT>(mp, s);
  }
}

template <class T, int G>
static int dispatch_1x16_nb(int nb, const GemvParams& p, hipStream_t s) {
  switch (nb) {
    case 1: return launch_1x16<T, G, 1>(p, s);
    case 2: return launch_1x16<T, G, 2>(p, s);
    case 4: return launch_1x16<T, G, 4>(p, s);
    default: return launch_1x16<T, G, 8>(p, s);
  }
}

}  // namespace aqlm

using namespace aqlm;

extern "C" int aqlm_hip_gemv_1x16(const void* codes, const void* codebook, const void* scales, const void* bias,
                                  const void* x, void* y, int out_features, int in_features, int in_group_size,
                                  int batch, long xs, long ys, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = validate_common(codes, codebook, scales, x, y, out_features, in_features, in_group_size, batch, dtype,
                              "aqlm_hip_gemv_1x16"))
    return e;
  if (in_group_size != 8 && in_group_size != 16) {
    // mirrors cuda_kernel.cpp:136-145
    set_last_error("aqlm_hip_gemv_1x16: only codebooks with 8 or 16 features are supported, got %d", in_group_size);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const int in_groups = in_features / in_group_size;
  const size_t x_row_bytes = (size_t)in_features * 2;
  const bool fast = !tuning().force_generic && (in_groups % 8 == 0) && aligned16(codes) && aligned16(codebook) &&
                    aligned16(x) && (xs % 8 == 0) && x_row_bytes <= kMaxXTileBytes;
  if (!fast)
    return run_generic(codes, codebook, scales, bias, x, y, out_features, in_features, 1, 16, in_group_size, batch, xs,
                       ys, dtype, stream);
  GemvParams p{};
  p.codes = (const uint8_t*)codes;
  p.codebooks = (const uint8_t*)codebook;
  p.scales = (const uint16_t*)scales;
  p.bias = (const uint16_t*)bias;
  p.M = out_features;
  p.xs = xs;
  p.ys = ys;
  p.code_row_bytes = (long)in_groups * 2;
  p.cb_bytes = 65536 * in_group_size * 2;
  p.prefetch = tuning().gemv1x16_prefetch_cb;
  finish_params<8>(p, in_groups, out_features, 4 * 4096);
  int done = 0;
  while (done < batch) {
    const int nb = pick_nb(batch - done, x_row_bytes);
    p.x = (const uint16_t*)x + (long)done * xs;
    p.y = (uint16_t*)y + (long)done * ys;
    int e;
    if (dtype == AQLM_HIP_F16)
      e = in_group_size == 8 ? dispatch_1x16_nb<F16, 8>(nb, p, stream) : dispatch_1x16_nb<F16, 16>(nb, p, stream);
    else
      e = in_group_size == 8 ? dispatch_1x16_nb<BF16, 8>(nb, p, stream) : dispatch_1x16_nb<BF16, 16>(nb, p, stream);
    if (e) return e;
    done += nb;
  }
  return 0;
}

extern "C" int aqlm_hip_gemv_1x16_multi(const aqlm_hip_segment* segments, int num_segments, const void* x,
                                        int in_features, int in_group_size, int batch, long xs, int dtype,
                                        void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!segments || num_segments < 1 || num_segments > AQLM_HIP_MAX_SEGMENTS) {
    set_last_error("aqlm_hip_gemv_1x16_multi: 1..%d segments required, got %d", AQLM_HIP_MAX_SEGMENTS, num_segments);
    return AQLM_HIP_E_INVALID;
  }
  long total_rows = 0;
  bool fast = true;
  for (int k = 0; k < num_segments; ++k) {
    const aqlm_hip_segment& sg = segments[k];
    if (int e = validate_common(sg.codes, sg.codebook, sg.scales, x, sg.y, sg.out_features, in_features, in_group_size,
                                batch, dtype, "aqlm_hip_gemv_1x16_multi"))
      return e;
    total_rows += sg.out_features;
    fast = fast && aligned16(sg.codes) && aligned16(sg.codebook);
  }
  if (in_group_size != 8 && in_group_size != 16) {
    set_last_error("aqlm_hip_gemv_1x16_multi: only codebooks with 8 or 16 features are supported, got %d", in_group_size);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const int in_groups = in_features / in_group_size;
  const size_t x_row_bytes = (size_t)in_features * 2;
  fast = fast && !tuning().force_generic && (in_groups % 8 == 0) && aligned16(x) && (xs % 8 == 0) &&
         x_row_bytes <= kMaxXTileBytes;
  if (!fast) {  // same results, one launch per segment
    for (int k = 0; k < num_segments; ++k) {
      const aqlm_hip_segment& sg = segments[k];
      if (int e = aqlm_hip_gemv_1x16(sg.codes, sg.codebook, sg.scales, sg.bias, x, sg.y, sg.out_features, in_features,
                                     in_group_size, batch, xs, sg.y_row_stride, dtype, stream_))
        return e;
    }
    return 0;
  }
  GemvMultiParams mp{};
  mp.nseg = num_segments;
  GemvParams& p = mp.common;
  p.xs = xs;
  p.code_row_bytes = (long)in_groups * 2;
  p.cb_bytes = 65536 * in_group_size * 2;
  p.prefetch = tuning().gemv1x16_prefetch_cb;
  finish_params<8>(p, in_groups, (int)std::min<long>(total_rows, 1 << 30), 4 * 4096);
  int done = 0;
  while (done < batch) {
    const int nb = pick_nb(batch - done, x_row_bytes);
    p.x = (const uint16_t*)x + (long)done * xs;
    for (int k = 0; k < num_segments; ++k) {
      const aqlm_hip_segment& sg = segments[k];
      mp.seg[k].codes = (const uint8_t*)sg.codes;
      mp.seg[k].codebooks = (const uint8_t*)sg.codebook;
      mp.seg[k].scales = (const uint16_t*)sg.scales;
      mp.seg[k].bias = (const uint16_t*)sg.bias;
      mp.seg[k].y = (uint16_t*)sg.y + (long)done * sg.y_row_stride;
      mp.seg[k].ys = sg.y_row_stride;
      mp.seg[k].M = sg.out_features;
    }
    int e;
    if (dtype == AQLM_HIP_F16)
      e = in_group_size == 8 ? dispatch_1x16_multi_nb<F16, 8>(nb, mp, stream) : dispatch_1x16_multi_nb<F16, 16>(nb, mp, stream);
    else
      e = in_group_size == 8 ? dispatch_1x16_multi_nb<BF16, 8>(nb, mp, stream) : dispatch_1x16_multi_nb<BF16, 16>(nb, mp, stream);
    if (e) return e;
    done += nb;
  }
  return 0;
}

namespace aqlm {
int gemv_kx8_replicated(const void* codes, const void* codebooks, const void* scales, const void* bias, const void* x,
                        void* y, int out_features, int in_features, int num_codebooks, int dtype, hipStream_t stream);

int gemv_kx8_replicated_multi(const aqlm_hip_segment* segments, int num_segments, const void* x, int in_features,
                              int num_codebooks, int dtype, hipStream_t stream);

template <class T, int KC>
static int dispatch_kx8_multi_nb(int nb, GemvMultiParams& mp, hipStream_t s) {
  switch (nb) {
    case 1: return launch_gemv_multi<T, 1, KC, 8, 8, 1, true, 4, 0>(mp, s);
    case 2: return launch_gemv_multi<T, 1, KC, 8, 8, 2, true, 4, 0>(mp, s);
    case 4: return launch_gemv_multi<T, 1, KC, 8, 8, 4, true, 4, 0>(mp, s);
    default: return launch_gemv_multi<T, 1, KC, 8, 8, 8, true, 4, 0>(mp, s);
  }
}

// Kx8 instances: (KC, G, U, NWAVES)
template <class T, int KC, int G, int U, int NWAVES>
static int dispatch_kx8_nb(int nb, const GemvParams& p, hipStream_t s) {
  switch (nb) {
    case 1: return launch_gemv<T, 1, KC, G, U, 1, true, NWAVES, 0>(p, s);
    case 2: return launch_gemv<T, 1, KC, G, U, 2, true, NWAVES, 0>(p, s);
    case 4: return launch_gemv<T, 1, KC, G, U, 4, true, NWAVES, 0>(p, s);
    default: return launch_gemv<T, 1, KC, G, U, 8, true, NWAVES, 0>(p, s);
  }
}
}  // namespace aqlm

// 2..8 rows of a 1x8 / 2x8 g8 layer: the matvec kernels pay one more x read + 4 dot products per code and row (2x8 g8 4096^2:
// 7.0 / 11.7 / 16.5 / 19.0 us at 2 / 3 / 6 / 8 rows), the fused dequant -> MFMA op with X resident in LDS (round 5, gemm_mfma.hip)
// costs 5.0 / 5.1 / 5.4 / 5.7 us (4096 -> 11008 at 6 rows: 34.5 vs 9.5 us; profiles/r05_gemm_kx8_xres.log).  From `kx8_mfma_min_rows`
// rows on (tuning key, default 2, 0 = never) the matvec entries hand the call to aqlm_hip_gemm_kx8_mfma: exact products, fp32 sums
// like the matvec kernels, and a row's bits there depend neither on the other rows nor on their number (2..16 rows) -- only the
// single-row kernels (replicated-LDS matvec, another summation order) differ from it in the last fp32 bit before the rounding.
namespace aqlm {
int gemm_kx8_xres_multi(const aqlm_hip_segment* segments, int num_segments, const void* X, int in_features, int K, int batch, long xs,
                        int dtype, hipStream_t stream);  // gemm_mfma.hip
}

static bool kx8_rows_take_mfma(int batch, int K, int G) {
  const int min_rows = aqlm::tuning().kx8_mfma_min_rows;
  return min_rows > 0 && batch >= min_rows && G == 8 && (K == 1 || K == 2) && !aqlm::tuning().force_generic;
}

extern "C" int aqlm_hip_gemv_kx8(const void* codes, const void* codebooks, const void* scales, const void* bias,
                                 const void* x, void* y, int out_features, int in_features, int num_codebooks,
                                 int in_group_size, int batch, long xs, long ys, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = validate_common(codes, codebooks, scales, x, y, out_features, in_features, in_group_size, batch, dtype,
                              "aqlm_hip_gemv_kx8"))
    return e;
  if (num_codebooks < 1 || num_codebooks > 16) {
    set_last_error("aqlm_hip_gemv_kx8: num_codebooks %d outside 1..16", num_codebooks);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (kx8_rows_take_mfma(batch, num_codebooks, in_group_size)) {
    const int e = aqlm_hip_gemm_kx8_mfma(codes, codebooks, scales, bias, x, y, batch, out_features, in_features, num_codebooks,
                                         in_group_size, xs, ys, dtype, stream_);
    if (e != AQLM_HIP_E_UNSUPPORTED) return e;  // (shapes outside the kernel: the matvec kernels below)
  }
  const int in_groups = in_features / in_group_size;
  const size_t x_row_bytes = (size_t)in_features * 2;
  const int K = num_codebooks, G = in_group_size;
  int U = 0, hint = 4 * 4096;
  if (K == 1 && G == 8) U = 8;
  else if (K == 2 && G == 8) U = 8;
  else if (K == 8 && G == 32) { U = 1; hint = 16 * 256; }
  // x tile budget: whatever the 160 KiB of LDS leaves next to the codebooks (8x8 g32: 128 KiB of codebooks)
  const size_t cb_lds = (size_t)K * 256 * G * 2;
  const size_t x_budget = std::min<size_t>(kMaxXTileBytes, 160 * 1024 - cb_lds - 2048);
  const bool fast = U != 0 && !tuning().force_generic && (in_groups % U == 0) && aligned16(codes) &&
                    aligned16(codebooks) && aligned16(x) && (xs % 8 == 0) && x_row_bytes + 256 <= x_budget;
  if (!fast)
    return run_generic(codes, codebooks, scales, bias, x, y, out_features, in_features, K, 8, G, batch, xs, ys, dtype,
                       stream);
  // batch 1, g = 8, 1 or 2 codebooks, enough rows (>= 16 per CU) to amortise the 64-128 KiB replicated fill:
  // conflict-free replicated-LDS kernel (measured cold, profiles/r03_mb_kx8_hoisted_loads.log: 4096->4096 4.9 vs 5.4 us,
  // 4096->11008 7.4 vs 9.0 us, 11008->4096 7.7 vs 8.3 us -- with the code words requested before the fill the wide-input
  // layers gain too; round 2 kept those on the plain kernel)
  const int rep = tuning().kx8_replicas;  // 1 = auto, 0 = never, 2 = whenever the shape fits
  if (batch == 1 && G == 8 && (K == 1 || K == 2) && rep != 0 && (rep == 2 || out_features >= 4096)) {
    const int e = gemv_kx8_replicated(codes, codebooks, scales, bias, x, y, out_features, in_features, K, dtype, stream);
    if (e != AQLM_HIP_E_UNSUPPORTED) return e;
  }
  GemvParams p{};
  p.codes = (const uint8_t*)codes;
  p.codebooks = (const uint8_t*)codebooks;
  p.scales = (const uint16_t*)scales;
  p.bias = (const uint16_t*)bias;
  p.M = out_features;
  p.xs = xs;
  p.ys = ys;
  p.code_row_bytes = (long)in_groups * K;
  p.cb_bytes = K * 256 * G * 2;
  p.prefetch = 0;
  if (U == 8) finish_params<8>(p, in_groups, out_features, hint);
  else finish_params<1>(p, in_groups, out_features, hint);
  int done = 0;
  while (done < batch) {
    int nb = 0;
    for (int cand : {8, 4, 2, 1})
      if (cand <= batch - done && (size_t)cand * (x_row_bytes + 256) <= x_budget) { nb = cand; break; }
    p.x = (const uint16_t*)x + (long)done * xs;
    p.y = (uint16_t*)y + (long)done * ys;
    int e;
    if (dtype == AQLM_HIP_F16) {
      if (K == 1) e = dispatch_kx8_nb<F16, 1, 8, 8, 4>(nb, p, stream);
      else if (K == 2) e = dispatch_kx8_nb<F16, 2, 8, 8, 4>(nb, p, stream);
      else e = dispatch_kx8_nb<F16, 8, 32, 1, 16>(nb, p, stream);
    } else {
      if (K == 1) e = dispatch_kx8_nb<BF16, 1, 8, 8, 4>(nb, p, stream);
      else if (K == 2) e = dispatch_kx8_nb<BF16, 2, 8, 8, 4>(nb, p, stream);
      else e = dispatch_kx8_nb<BF16, 8, 32, 1, 16>(nb, p, stream);
    }
    if (e) return e;
    done += nb;
  }
  return 0;
}

extern "C" int aqlm_hip_gemv_generic(const void* codes, const void* codebooks, const void* scales, const void* bias,
                                     const void* x, void* y, int out_features, int in_features, int num_codebooks,
                                     int nbits, int in_group_size, int batch, long xs, long ys, int dtype,
                                     void* stream) {
  if (int e = validate_common(codes, codebooks, scales, x, y, out_features, in_features, in_group_size, batch, dtype,
                              "aqlm_hip_gemv_generic"))
    return e;
  if (nbits < 1 || nbits > 16 || num_codebooks < 1) {
    set_last_error("aqlm_hip_gemv_generic: nbits %d / num_codebooks %d unsupported", nbits, num_codebooks);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  return run_generic(codes, codebooks, scales, bias, x, y, out_features, in_features, num_codebooks, nbits,
                     in_group_size, batch, xs, ys, dtype, (hipStream_t)stream);
}

extern "C" int aqlm_hip_gemv_kx8_multi(const aqlm_hip_segment* segments, int num_segments, const void* x,
                                       int in_features, int num_codebooks, int in_group_size, int batch, long xs,
                                       int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!segments || num_segments < 1 || num_segments > AQLM_HIP_MAX_SEGMENTS) {
    set_last_error("aqlm_hip_gemv_kx8_multi: 1..%d segments required, got %d", AQLM_HIP_MAX_SEGMENTS, num_segments);
    return AQLM_HIP_E_INVALID;
  }
  long total_rows = 0;
  bool aligned = aligned16(x);
  for (int k = 0; k < num_segments; ++k) {
    const aqlm_hip_segment& sg = segments[k];
    if (int e = validate_common(sg.codes, sg.codebook, sg.scales, x, sg.y, sg.out_features, in_features, in_group_size,
                                batch, dtype, "aqlm_hip_gemv_kx8_multi"))
      return e;
    total_rows += sg.out_features;
    aligned = aligned && aligned16(sg.codes) && aligned16(sg.codebook);
  }
  if (num_codebooks < 1 || num_codebooks > 16) {
    set_last_error("aqlm_hip_gemv_kx8_multi: num_codebooks %d outside 1..16", num_codebooks);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const int in_groups = in_features / in_group_size;
  const size_t x_row_bytes = (size_t)in_features * 2;
  const int K = num_codebooks;
  const size_t x_budget = std::min<size_t>(kMaxXTileBytes, 160 * 1024 - (size_t)K * 256 * 16 - 2048);
  const bool fast = in_group_size == 8 && (K == 1 || K == 2) && !tuning().force_generic && in_groups % 8 == 0 &&
                    aligned && xs % 8 == 0 && x_row_bytes + 256 <= x_budget;
  // 2+ rows of 1x8 / 2x8 g8 (round 5): ONE launch of the X-resident fused MFMA kernel over all layers -- X is loaded once, the layers'
  // codebooks sit side by side in LDS, the tile walk runs over the layers back to back (bit-identical to separate launches)
  {
    const int mr = tuning().kx8_multi_xres_min_rows;
    if (fast && mr > 0 && batch >= mr && batch <= 16 && (batch == 1 || kx8_rows_take_mfma(batch, K, in_group_size))) {
      const int e = aqlm::gemm_kx8_xres_multi(segments, num_segments, x, in_features, K, batch, xs, dtype, stream);
      if (e != AQLM_HIP_E_UNSUPPORTED) return e;
    }
  }
  // (layers whose codebooks + X do not fit one LDS together: one launch of the fused MFMA kernel per segment still beats one matvec launch over all of them)
  if (!fast || kx8_rows_take_mfma(batch, K, in_group_size)) {  // other schemes (8x8 ...) and odd shapes: one launch per segment, same results as the single-layer op
    for (int k = 0; k < num_segments; ++k) {
      const aqlm_hip_segment& sg = segments[k];
      if (int e = aqlm_hip_gemv_kx8(sg.codes, sg.codebook, sg.scales, sg.bias, x, sg.y, sg.out_features, in_features, K,
                                    in_group_size, batch, xs, sg.y_row_stride, dtype, stream_))
        return e;
    }
    return 0;
  }
  const int rep = tuning().kx8_replicas;
  if (batch == 1 && rep != 0 && (rep == 2 || total_rows >= 4096)) {
    const int e = gemv_kx8_replicated_multi(segments, num_segments, x, in_features, K, dtype, stream);
    if (e != AQLM_HIP_E_UNSUPPORTED) return e;
  }
  GemvMultiParams mp{};
  mp.nseg = num_segments;
  GemvParams& p = mp.common;
  p.xs = xs;
  p.code_row_bytes = (long)in_groups * K;
  p.cb_bytes = K * 256 * 8 * 2;
  p.prefetch = 0;
  finish_params<8>(p, in_groups, (int)std::min<long>(total_rows, 1 << 30), 4 * 4096);
  int done = 0;
  while (done < batch) {
    int nb = 0;
    for (int cand : {8, 4, 2, 1})
      if (cand <= batch - done && (size_t)cand * (x_row_bytes + 256) <= x_budget) { nb = cand; break; }
    p.x = (const uint16_t*)x + (long)done * xs;
    for (int k = 0; k < num_segments; ++k) {
      const aqlm_hip_segment& sg = segments[k];
      mp.seg[k].codes = (const uint8_t*)sg.codes;
      mp.seg[k].codebooks = (const uint8_t*)sg.codebook;
      mp.seg[k].scales = (const uint16_t*)sg.scales;
      mp.seg[k].bias = (const uint16_t*)sg.bias;
      mp.seg[k].y = (uint16_t*)sg.y + (long)done * sg.y_row_stride;
      mp.seg[k].ys = sg.y_row_stride;
      mp.seg[k].M = sg.out_features;
    }
    int e;
    if (dtype == AQLM_HIP_F16) e = K == 1 ? dispatch_kx8_multi_nb<F16, 1>(nb, mp, stream) : dispatch_kx8_multi_nb<F16, 2>(nb, mp, stream);
    else e = K == 1 ? dispatch_kx8_multi_nb<BF16, 1>(nb, mp, stream) : dispatch_kx8_multi_nb<BF16, 2>(nb, mp, stream);
    if (e) return e;
    done += nb;
  }
  return 0;
}
