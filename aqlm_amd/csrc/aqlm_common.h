// Shared device/host helpers for the gfx950 AQLM kernels.  wave64 everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/aqlm_hip.h"

namespace aqlm {

constexpr int WAVE = 64;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// buffer cache-policy bits of gfx940+ raw buffer loads (cdna_hip_programming.md T8 / G16)
constexpr int AUX_DEFAULT = 0;
constexpr int AUX_SC0 = 1;
constexpr int AUX_NT = 2;
constexpr int AUX_SC1 = 16;

// ---------------------------------------------------------------------------------------------
// element-type traits: fp16 and bf16 share every kernel; arithmetic is v_dot2c_f32_{f16,bf16}
// with fp32 accumulation (the reference CUDA kernels accumulate each group in half precision,
// cuda_kernel.cu:69-75; we do not reproduce that rounding).
// ---------------------------------------------------------------------------------------------
struct F16 {
  static constexpr int id = AQLM_HIP_F16;
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
  }
  static __device__ __forceinline__ float to_float(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
  static __device__ __forceinline__ uint16_t from_float(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);  // v_cvt_f16_f32, round-to-nearest-even
  }
  // packed add of two f16 pairs (used by dequant to sum codebooks like embedding_bag does in storage dtype)
  static __device__ __forceinline__ float lo(uint32_t a) { return to_float((uint16_t)(a & 0xffffu)); }
  static __device__ __forceinline__ float hi(uint32_t a) { return to_float((uint16_t)(a >> 16)); }
};

struct BF16 {
  static constexpr int id = AQLM_HIP_BF16;
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
  }
  static __device__ __forceinline__ float to_float(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
  static __device__ __forceinline__ uint16_t from_float(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                               // round-to-nearest-even
    return (uint16_t)(u >> 16);
  }
  static __device__ __forceinline__ float lo(uint32_t a) { return __uint_as_float(a << 16); }
  static __device__ __forceinline__ float hi(uint32_t a) { return __uint_as_float(a & 0xffff0000u); }
};

// <8 halfs, 8 halfs> accumulated into fp32
template <class T>
__device__ __forceinline__ float dot8(const u32x4& w, const u32x4& x, float acc) {
  acc = T::dot2(w.x, x.x, acc);
  acc = T::dot2(w.y, x.y, acc);
  acc = T::dot2(w.z, x.z, acc);
  acc = T::dot2(w.w, x.w, acc);
  return acc;
}

// Reductions on the VALU (DPP): __shfl_xor compiles to ds_bpermute_b32, i.e. every step of a shuffle tree is an LDS
// instruction with LDS latency -- in the gather kernels that is 4-6 extra LDS round trips per output row on the unit
// that is already the bottleneck.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15); every lane of the row receives the total
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);  // row_half_mirror: the other quad of each group of 8
  v += dpp_f32<0x140>(v);  // row_mirror: the other half of the row
  return v;
}

// sum over aligned groups of 4 lanes; every lane of the group receives the total
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  return v;
}

// sum over aligned groups of 8 lanes; every lane of the group receives the total
__device__ __forceinline__ float oct_sum(float v) {
  v = quad_sum(v);
  v += dpp_f32<0x141>(v);  // row_half_mirror
  return v;
}

// sum over the wave; every lane receives the total
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}

// maximum over the wave of an unsigned value (DPP within the rows, scalar across them); wave-uniform result
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  uint32_t o;
  o = dpp_u32<0xB1>(v);  v = o > v ? o : v;
  o = dpp_u32<0x4E>(v);  v = o > v ? o : v;
  o = dpp_u32<0x141>(v); v = o > v ? o : v;
  o = dpp_u32<0x140>(v); v = o > v ? o : v;
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  const uint32_t a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
  return a > b ? a : b;
}

// compile-time extraction of code #idx from the dwords of one code word
template <int CODE_BYTES, int N>
__device__ __forceinline__ uint32_t code_at(const uint32_t (&cw)[N], int idx) {
  if constexpr (CODE_BYTES == 2) {
    return (cw[idx >> 1] >> ((idx & 1) * 16)) & 0xffffu;
  } else {
    return (cw[idx >> 2] >> ((idx & 3) * 8)) & 0xffu;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void set_last_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
// Raise a kernel's dynamic-LDS limit (needed above the 64 KiB default) once per (kernel, device): a process may drive
// several GPUs (device_map="auto"), and the attribute is per device.  Returns 0 or an error code.
int ensure_dynamic_lds(const void* kernel, size_t bytes);

struct Tuning {
  int gemv_rows_per_wave = 0;    // 0 = heuristic
  int gemv1x16_aux = AUX_DEFAULT;  // cache policy of the codebook gathers: 0 default, 1 sc0, 2 nt, 16 sc1
  int gemv1x16_prefetch_cb = 0;  // 1: each block touches a slice of the codebook first (warms its XCD's L2)
  int kx8_replicas = 1;          // K x 8 g8 batch-1: 1 = replicated-LDS kernel for >= 4096 rows, 0 = never, 2 = always
  int gemm_variant = 0;          // large-batch 1x16 op: 0 = slice-scan kernel (round 6) up to scan_max_rows (default 0: never) where it applies (g 8, in_features % 256 == 0), else as 5; 1 = register-staged split-K kernel (round 1), 2 = 16-row L2-gather kernel wherever it applies, 3 = K-split L2-gather pipeline only, 4 = slice-scan kernel at any row count, 5 = round 5's routing (16-row kernel <= 64 rows, pipeline above)
  int scan_max_rows = 0;         // rows up to which variant 0 takes the slice-scan kernel (0: never -- measured slower than round 5's routes, gemm_1x16_scan.hip)
  int scan_prefetch = 4;         // slice-scan kernel: tiles of code words in flight per wave (2 / 4)
  int kx8_xres = 1;              // fused K x 8 MFMA op at <= 16 rows: 1 = X resident in LDS (round 5), 0 = the streaming 16-row kernel (A/B runs)
  int kx8_xres_phased = 1;       // ... whose X image does not fit the LDS at once: 1 = the X-resident kernel in phases (round 5; also 17 .. 32 rows), 2 = up to 16 rows only, 0 = the streaming 16-row kernel
  int kx8_phase_tpb = 0;         // experiments: 1..3 = every <= 32-row call on the phased kernel with this many tiles per workgroup (0 = off)
  int kx8_phase_quads = 0;       // experiments: cap on the quads per phase (smaller image -> more workgroups per CU)
  int kx8_mfma_min_rows = 2;     // rows from which aqlm_hip_gemv_kx8[_multi] hands 1x8 / 2x8 g8 calls to the fused MFMA kernel (0 = never; 3 in round 4, 2 since the X-resident kernel)
  int kx8_multi_xres_min_rows = 2;  // shared-input launches of 1x8 / 2x8 g8: rows from which all layers run in ONE launch of the X-resident MFMA kernel (0 = never; 1 = batch 1 too)
  int kx8_ksplit = 0;            // fused K x 8 MFMA op with a workspace at >= 49 rows: 0 = K slices by the plan (<= 4), 1 = never split, 2 / 4 = force
  int kx8_rt = 0;                // ... 16-row tiles per block: 0 = by the plan, 1 / 2 = force (A/B runs)
  int gemm_debug = 0;            // LDS-DMA pipeline: knock-out switches for timing experiments (never set in production: results are wrong)
  int gemm_store_nt = 0;         // LDS-DMA pipeline: fp32 partials stored with the non-temporal hint
  int force_generic = 0;         // 1: route every gemv through the generic kernel (testing)
  int packed_fused_finalize = 1;  // 0 = two-kernel finalize even when the descriptor carries the codebook range (A/B runs)
  int packed_waves = 0;          // prepack: waves per workgroup of the packed 1x16 kernel (4 / 8 / 16); 0 = heuristic
  int packed_arrange = 1;        // prepack: bank-aware order of the entries: 1 = greedy deal + local search for layers of <= 8 Mi codes, greedy deal alone above; 3 = local search always; 2 = greedy deal only; 0 = ascending j
  int packed_xcopies = 0;        // prepack: rotated copies of x the batch-1 kernel keeps in LDS (1..4, capped by what fits); 0 = 1
  int packed_entry_bytes = 0;    // prepack: 0 / 4 = 32-bit entries; 3 = 24-bit entries (wave ranges of <= 32 steps)
  int packed_debug = 0;          // profiling builds (-DAQLM_PACKED_TRACE) only: 1 = no LDS reads / dots, 2 = no entry stream
  int packed_prefetch = 0;       // packed 1x16 kernel: steps of the entry stream in flight per wave (4 / 8); 0 = heuristic
  int packed_pipe = 1;           // shared-input launches of batch 1: 1 = one workgroup per CU walks the segments with double-buffered slices
  int packed_fill_rotate = 1;    // packed 1x16 kernel: 1 = the workgroups that share a codebook slice start their LDS fill at different pieces
  int lut_waves = 0;             // 8 x 8 look-up-table matvec: waves per workgroup (8 / 16); 0 = default
  int packed_prefetch_waves = 0; // chain prefetch: extra waves per workgroup that pull the next layer towards L2 (0 = 2, -1 = off)
};
Tuning& tuning();

// slice-scan MFMA kernel of the 1x16 g8 scheme at 2+ rows (gemm_1x16_scan.hip); the large-batch op of gemm_mfma.hip routes to it
namespace scan {
size_t workspace_bytes(int B, int M, int K);  // 0: no plan (in_features % 256 != 0)
int run(const void* codes, const void* codebook, const void* scales, const void* bias, const void* X, void* Y, int batch, int out_features,
        int in_features, long xs, long ys, int dtype, void* workspace, hipStream_t stream);
}  // namespace scan

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace aqlm
