// AQLM weight reconstruction W[out][in] = sum_c codebooks[c][code] (* scales[row]) for gfx950.
//
// Replaces (behaviour, not code): Code1x16Dequant / Code2x8Dequant / CodeKx8Dequant and launchers
// (reference cuda_kernel.cu:98-142, 235-294, 392-468, 523-553, 622-707, 760-817) and the scaled pybind variants
// code{1x16,2x8,1x8}_dequant (cuda_kernel.cpp:184-227, 423-448, 588-613).
//
// One thread reconstructs one input group (G halfs) at a time: consecutive lanes own consecutive groups, so the
// code loads are coalesced 64-128 B per wave and the 16-B stores of a wave cover a contiguous span of the row.
// All codebooks are summed in fp32 registers and rounded once (the reference's CodeKx8Dequant re-reads and
// re-writes W once per codebook, cuda_kernel.cu:434-453).  1x16 gathers come from L2 (buffer loads), Kx8 gathers from
// an LDS copy of the codebooks.  The kernel is HBM-write bound: 2*out*in bytes of W.
#include <algorithm>

#include "aqlm_common.h"

namespace aqlm {

struct DequantParams {
  const uint8_t* codes;
  const uint8_t* codebooks;
  const uint16_t* scales;  // nullable
  uint16_t* W;
  int M, in_groups;
  long total_groups;
  int cb_bytes;
};

template <class T, int CODE_BYTES, int KC, int G, bool CB_LDS, int NT>
__global__ __launch_bounds__(NT) void dequant_kernel(const DequantParams p) {
  constexpr int P = G / 8;
  constexpr int CB_SIZE = CB_LDS ? 256 : 65536;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* const cbl = reinterpret_cast<u32x4*>(smem_raw);
  if constexpr (CB_LDS) {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.codebooks);
    for (int q = threadIdx.x; q < KC * 256 * P; q += NT) cbl[q] = src[q];
    __syncthreads();
  }
  __amdgpu_buffer_rsrc_t rsrc;
  if constexpr (!CB_LDS) rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.codebooks, 0, p.cb_bytes, 0x00020000);

  const long stride = (long)gridDim.x * NT;
  for (long t = (long)blockIdx.x * NT + threadIdx.x; t < p.total_groups; t += stride) {
    const int row = (int)(t / p.in_groups);
    // codes of group t: KC containers, contiguous
    uint32_t code[KC];
    if constexpr (CODE_BYTES == 2) {
      const uint16_t* cp = reinterpret_cast<const uint16_t*>(p.codes) + t * KC;
#pragma unroll
      for (int c = 0; c < KC; ++c) code[c] = cp[c];
    } else if constexpr (KC == 8) {
      const u32x2 v = *reinterpret_cast<const u32x2*>(p.codes + t * 8);
#pragma unroll
      for (int c = 0; c < 8; ++c) code[c] = ((c < 4 ? v.x : v.y) >> ((c & 3) * 8)) & 0xffu;
    } else if constexpr (KC == 2) {
      const uint32_t v = *reinterpret_cast<const uint16_t*>(p.codes + t * 2);
      code[0] = v & 0xffu;
      code[1] = v >> 8;
    } else {
#pragma unroll
      for (int c = 0; c < KC; ++c) code[c] = p.codes[t * KC + c];
    }
    const float scale = p.scales ? T::to_float(p.scales[row]) : 1.f;
    uint16_t* out = p.W + t * G;
#pragma unroll
    for (int pp = 0; pp < P; ++pp) {
      float f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        u32x4 e;
        if constexpr (CB_LDS) e = cbl[(c * 256 + code[c]) * P + pp];
        else e = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(c * CB_SIZE + code[c]) * (G * 2) + pp * 16, 0, 0);
        f[0] += T::lo(e.x); f[1] += T::hi(e.x);
        f[2] += T::lo(e.y); f[3] += T::hi(e.y);
        f[4] += T::lo(e.z); f[5] += T::hi(e.z);
        f[6] += T::lo(e.w); f[7] += T::hi(e.w);
      }
      u32x4 o;
      o.x = (uint32_t)T::from_float(f[0] * scale) | ((uint32_t)T::from_float(f[1] * scale) << 16);
      o.y = (uint32_t)T::from_float(f[2] * scale) | ((uint32_t)T::from_float(f[3] * scale) << 16);
      o.z = (uint32_t)T::from_float(f[4] * scale) | ((uint32_t)T::from_float(f[5] * scale) << 16);
      o.w = (uint32_t)T::from_float(f[6] * scale) | ((uint32_t)T::from_float(f[7] * scale) << 16);
      *reinterpret_cast<u32x4*>(out + pp * 8) = o;
    }
  }
}

// generic: any scheme (runtime KC / nbits / G), one thread per output element group, scalar loads
struct DequantGenericParams {
  const uint8_t* codes;
  const uint16_t* codebooks;
  const uint16_t* scales;
  uint16_t* W;
  int M, in_groups, KC, nbits, G, code_bytes;
  long total_groups;
};

template <class T>
__global__ __launch_bounds__(256) void dequant_generic_kernel(const DequantGenericParams p) {
  const uint32_t mask = (1u << p.nbits) - 1u;
  const long cbsize = 1L << p.nbits;
  const long stride = (long)gridDim.x * 256;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < p.total_groups; t += stride) {
    const int row = (int)(t / p.in_groups);
    const float scale = p.scales ? T::to_float(p.scales[row]) : 1.f;
    for (int e = 0; e < p.G; ++e) {
      float f = 0.f;
      for (int c = 0; c < p.KC; ++c) {
        uint32_t code = p.code_bytes == 1 ? (uint32_t)p.codes[t * p.KC + c]
                                          : (uint32_t) reinterpret_cast<const uint16_t*>(p.codes)[t * p.KC + c];
        code &= mask;
        f += T::to_float(p.codebooks[((long)c * cbsize + code) * p.G + e]);
      }
      p.W[t * p.G + e] = T::from_float(f * scale);
    }
  }
}

template <class T, int CODE_BYTES, int KC, int G, bool CB_LDS, int NT>
static int launch_dequant(const DequantParams& p, hipStream_t stream) {
  auto kern = dequant_kernel<T, CODE_BYTES, KC, G, CB_LDS, NT>;
  const size_t lds = CB_LDS ? (size_t)KC * 256 * G * 2 : 0;
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  long need = (p.total_groups + NT - 1) / NT;
  // LDS-resident codebooks: persistent blocks amortise the fill; L2 gathers: plenty of small blocks
  const long cap = CB_LDS ? (lds > 48 * 1024 ? 256 : 256 * 8) : 256 * 32;
  const int blocks = (int)std::max<long>(1, std::min<long>(need, cap));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, stream, p);
  return check_hip(hipGetLastError(), "dequant launch");
}

static int run_dequant_generic(const void* codes, const void* codebooks, const void* scales, void* W, int out_features,
                               int in_features, int K, int nbits, int G, int dtype, hipStream_t stream) {
  DequantGenericParams g;
  g.codes = (const uint8_t*)codes;
  g.codebooks = (const uint16_t*)codebooks;
  g.scales = (const uint16_t*)scales;
  g.W = (uint16_t*)W;
  g.M = out_features;
  g.in_groups = in_features / G;
  g.KC = K;
  g.nbits = nbits;
  g.G = G;
  g.code_bytes = nbits <= 8 ? 1 : 2;
  g.total_groups = (long)out_features * g.in_groups;
  const int blocks = (int)std::max<long>(1, std::min<long>((g.total_groups + 255) / 256, 256 * 32));
  if (dtype == AQLM_HIP_F16) hipLaunchKernelGGL(dequant_generic_kernel<F16>, dim3(blocks), dim3(256), 0, stream, g);
  else hipLaunchKernelGGL(dequant_generic_kernel<BF16>, dim3(blocks), dim3(256), 0, stream, g);
  return check_hip(hipGetLastError(), "dequant_generic launch");
}

static int validate_dequant(const void* codes, const void* codebooks, void* W, int out_features, int in_features,
                            int G, int dtype, const char* who) {
  if (!codes || !codebooks || !W) {
    set_last_error("%s: null pointer argument", who);
    return AQLM_HIP_E_INVALID;
  }
  if (out_features <= 0 || in_features <= 0 || G <= 0 || in_features % G != 0) {
    set_last_error("%s: bad sizes out=%d in=%d g=%d", who, out_features, in_features, G);
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("%s: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", who, dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  return 0;
}

}  // namespace aqlm

using namespace aqlm;

extern "C" int aqlm_hip_dequant_1x16(const void* codes, const void* codebook, const void* scales, void* W,
                                     int out_features, int in_features, int in_group_size, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = validate_dequant(codes, codebook, W, out_features, in_features, in_group_size, dtype,
                               "aqlm_hip_dequant_1x16"))
    return e;
  if (in_group_size != 8 && in_group_size != 16) {
    set_last_error("aqlm_hip_dequant_1x16: only codebooks with 8 or 16 features are supported, got %d", in_group_size);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (tuning().force_generic || !aligned16(W) || !aligned16(codebook))
    return run_dequant_generic(codes, codebook, scales, W, out_features, in_features, 1, 16, in_group_size, dtype,
                               stream);
  DequantParams p{};
  p.codes = (const uint8_t*)codes;
  p.codebooks = (const uint8_t*)codebook;
  p.scales = (const uint16_t*)scales;
  p.W = (uint16_t*)W;
  p.M = out_features;
  p.in_groups = in_features / in_group_size;
  p.total_groups = (long)out_features * p.in_groups;
  p.cb_bytes = 65536 * in_group_size * 2;
  if (dtype == AQLM_HIP_F16)
    return in_group_size == 8 ? launch_dequant<F16, 2, 1, 8, false, 256>(p, stream)
                              : launch_dequant<F16, 2, 1, 16, false, 256>(p, stream);
  return in_group_size == 8 ? launch_dequant<BF16, 2, 1, 8, false, 256>(p, stream)
                            : launch_dequant<BF16, 2, 1, 16, false, 256>(p, stream);
}

extern "C" int aqlm_hip_dequant_kx8(const void* codes, const void* codebooks, const void* scales, void* W,
                                    int out_features, int in_features, int num_codebooks, int in_group_size, int dtype,
                                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = validate_dequant(codes, codebooks, W, out_features, in_features, in_group_size, dtype,
                               "aqlm_hip_dequant_kx8"))
    return e;
  const int K = num_codebooks, G = in_group_size;
  if (K < 1 || K > 16) {
    set_last_error("aqlm_hip_dequant_kx8: num_codebooks %d outside 1..16", K);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const bool tuned = !tuning().force_generic && aligned16(W) && aligned16(codebooks) && aligned16(codes) &&
                     ((K == 1 && G == 8) || (K == 2 && G == 8) || (K == 8 && G == 32));
  if (!tuned) return run_dequant_generic(codes, codebooks, scales, W, out_features, in_features, K, 8, G, dtype, stream);
  DequantParams p{};
  p.codes = (const uint8_t*)codes;
  p.codebooks = (const uint8_t*)codebooks;
  p.scales = (const uint16_t*)scales;
  p.W = (uint16_t*)W;
  p.M = out_features;
  p.in_groups = in_features / G;
  p.total_groups = (long)out_features * p.in_groups;
  p.cb_bytes = K * 256 * G * 2;
  if (dtype == AQLM_HIP_F16) {
    if (K == 1) return launch_dequant<F16, 1, 1, 8, true, 256>(p, stream);
    if (K == 2) return launch_dequant<F16, 1, 2, 8, true, 256>(p, stream);
    return launch_dequant<F16, 1, 8, 32, true, 1024>(p, stream);
  }
  if (K == 1) return launch_dequant<BF16, 1, 1, 8, true, 256>(p, stream);
  if (K == 2) return launch_dequant<BF16, 1, 2, 8, true, 256>(p, stream);
  return launch_dequant<BF16, 1, 8, 32, true, 1024>(p, stream);
}

extern "C" int aqlm_hip_dequant_generic(const void* codes, const void* codebooks, const void* scales, void* W,
                                        int out_features, int in_features, int num_codebooks, int nbits,
                                        int in_group_size, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int e = validate_dequant(codes, codebooks, W, out_features, in_features, in_group_size, dtype,
                               "aqlm_hip_dequant_generic"))
    return e;
  if (num_codebooks < 1 || nbits < 1 || nbits > 16) {
    set_last_error("aqlm_hip_dequant_generic: num_codebooks %d / nbits %d outside what the code containers hold",
                   num_codebooks, nbits);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  return run_dequant_generic(codes, codebooks, scales, W, out_features, in_features, num_codebooks, nbits,
                             in_group_size, dtype, stream);
}
