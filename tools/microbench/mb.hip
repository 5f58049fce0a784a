// Micro-benchmarks that bound the AQLM kernels on MI355X (run on the GPU box; results quoted in DESIGN.md):
//   mb l2gather   random 16-B gathers from an L2-resident table (the 1x16 codebook access pattern)
//   mb ldsgather  random ds_read_b128 from an LDS-resident table (the Kx8 codebook access pattern)
//   mb stream     coalesced non-temporal streaming read (HBM ceiling for the code stream)
//   mb gemv       libaqlm_hip.so kernels through the C ABI: cold (rotating layers > 512 MiB, hipGraph replay)
//                 and warm, for tuning-knob variants
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <chrono>
#include <vector>

#include "../../include/aqlm_hip.h"

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// ---------------------------------------------------------------- fill kernels
__global__ void fill_u32(uint32_t* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = mix((uint32_t)i * 2654435761U + seed);
}
// fp16 values uniform in [-1, 1): two per dword
__global__ void fill_half(uint32_t* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t h = mix((uint32_t)i * 2654435761U + seed);
    const _Float16 a = (_Float16)(((int)(h & 0x7ff) - 1024) / 1024.0f);
    const _Float16 b = (_Float16)(((int)((h >> 11) & 0x7ff) - 1024) / 1024.0f);
    p[i] = (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
  }
}
__global__ void fill_one_half(uint16_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = 0x3c00;
}

// ---------------------------------------------------------------- L2 gather
template <int AUX, int PIECES, bool FLAT>
__global__ __launch_bounds__(256) void l2gather(const uint8_t* tab, uint32_t entry_mask, int iters, u32x4* out) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, (entry_mask + 1) * 16 * PIECES, 0x00020000);
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  uint32_t s = gid * 2654435761U;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it += 8) {
    u32x4 v[8 * PIECES];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t off = (mix(s + it + k) & entry_mask) * (16 * PIECES);
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        if constexpr (FLAT) v[k * PIECES + p] = *reinterpret_cast<const u32x4*>(tab + off + p * 16);
        else v[k * PIECES + p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + p * 16, 0, AUX);
      }
    }
#pragma unroll
    for (int k = 0; k < 8 * PIECES; ++k) acc ^= v[k];
  }
  out[gid] = acc;
}

// ---------------------------------------------------------------- LDS gather
template <int MODE>  // 0 random, 1 conflict-free (slot = lane & 15), 2 all lanes same address
__global__ __launch_bounds__(1024) void ldsgather(uint32_t entry_mask, int iters, u32x4* out) {
  extern __shared__ __attribute__((aligned(16))) u32x4 tab[];
  for (uint32_t q = threadIdx.x; q <= entry_mask; q += blockDim.x) tab[q] = u32x4{q, q * 3, q * 5, q * 7};
  __syncthreads();
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  uint32_t s = gid * 2654435761U;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it += 8) {
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      uint32_t idx = mix(s + it + k) & entry_mask;
      if (MODE == 1) idx = (idx & ~15u) | (lane & 15u);
      if (MODE == 2) idx = __builtin_amdgcn_readfirstlane(idx);
      v[k] = tab[idx];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k];
  }
  out[gid] = acc;
}

// ---------------------------------------------------------------- VALU / LDS issue rates of the packed kernel's inner loop
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int MODE>  // 0: v_dot2c_f32_f16, 1: v_fma_f32, 2..8: integer ops (see the chains below)
__global__ __launch_bounds__(1024) void valu_rate(int iters, float* out) {
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f;
  uint32_t w0 = threadIdx.x * 2654435761u, w1 = w0 ^ 0x3c003c00u, w2 = ~w0, w3 = ~w1;
  uint32_t m = 0xfffffff0u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 0) {
        a0 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w0), __builtin_bit_cast(h2, w1), a0, false);
        a1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w1), __builtin_bit_cast(h2, w0), a1, false);
        a2 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w0), __builtin_bit_cast(h2, w0), a2, false);
        a3 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w1), __builtin_bit_cast(h2, w1), a3, false);
      } else if (MODE == 1) {
        a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f);
        a2 = __builtin_fmaf(a2, 1.0001f, 0.5f); a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
      } else {
        // integer ops: four independent self-chains (the values collapse, the issue slots do not)
#define MB_CHAIN(INSN)                                   \
        asm volatile(INSN : "+v"(w0) : "v"(m));          \
        asm volatile(INSN : "+v"(w1) : "v"(m));          \
        asm volatile(INSN : "+v"(w2) : "v"(m));          \
        asm volatile(INSN : "+v"(w3) : "v"(m));
        if (MODE == 2) { MB_CHAIN("v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") }
        if (MODE == 3) { MB_CHAIN("v_and_b32 %0, %1, %0") }
        if (MODE == 4) { MB_CHAIN("v_lshrrev_b32 %0, 1, %0") }
        if (MODE == 5) { MB_CHAIN("v_pk_lshlrev_b16 %0, 1, %0") }
        if (MODE == 6) { MB_CHAIN("v_bfe_u32 %0, %0, 1, 31") }
        if (MODE == 7) { MB_CHAIN("v_and_b32 %0, 0xfffffff0, %0") }
        if (MODE == 8) { MB_CHAIN("v_xor_b32 %0, %1, %0") }
#undef MB_CHAIN
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(w0 ^ w1 ^ w2 ^ w3);
}

// ds_read_b128 with the addresses held in registers (no address arithmetic in the loop): MODE 0 random slots,
// 1 conflict-free (slot % 16 == position of the lane in its 16-lane service group), 2 two reads per "entry" like the kernel
__device__ __forceinline__ int mb_group_pos(int l) {
  const int h = l & 31;
  if (h < 4) return h;
  if (h < 12) return h - 4;
  if (h < 16) return h - 12 + 4;
  if (h < 20) return h - 16 + 8;
  if (h < 28) return h - 20 + 8;
  return h - 28 + 12;
}
template <int MODE>
__global__ __launch_bounds__(1024) void lds_rate(int iters, u32x4* out) {
  extern __shared__ __attribute__((aligned(16))) u32x4 tab[];  // 64 KiB = 4096 slots
  for (uint32_t q = threadIdx.x; q < 4096; q += blockDim.x) tab[q] = u32x4{q, q * 3, q * 5, q * 7};
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t idx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    uint32_t r = mix((blockIdx.x * 1024 + threadIdx.x) * 8 + k) & 4095u;
    if (MODE == 1) r = (r & ~15u) | (uint32_t)mb_group_pos(lane);
    idx[k] = r * 16u;
  }
  u32x4 acc = {0, 0, 0, 0};
  typedef __attribute__((address_space(3))) const u32x4* lp;
  const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) u32x4*)tab;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *(lp)(size_t)(base + idx[k]);
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : : "v"(v[k]));
    acc.x += v[0].x;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---------------------------------------------------------------- streaming read
__global__ __launch_bounds__(256) void stream_read(const u32x4* p, size_t n, u32x4* out) {
  u32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
    u32x4 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n; i += stride) acc ^= __builtin_nontemporal_load(p + i);
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void empty_kernel() {}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  void start(hipStream_t s = 0) { CK(hipEventRecord(a, s)); }
  float stop_ms(hipStream_t s = 0) {
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
  }
};

static double clock_ghz() {
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  return khz / 1e6;
}

template <int AUX, int PIECES, bool FLAT>
static void run_l2(const char* name, const uint8_t* tab, size_t table_bytes, int blocks_per_cu, u32x4* out) {
  const int blocks = 256 * blocks_per_cu, iters = 256;
  const uint32_t mask = (uint32_t)(table_bytes / (16 * PIECES)) - 1;
  Timer t;
  hipLaunchKernelGGL((l2gather<AUX, PIECES, FLAT>), dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out);
  CK(hipDeviceSynchronize());
  t.start();
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((l2gather<AUX, PIECES, FLAT>), dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out);
  const float ms = t.stop_ms() / reps;
  const double gathers = (double)blocks * 256 * iters;
  printf("l2gather %-22s table %7zu KiB  %2d blk/CU  %8.1f Ggather/s  %6.2f lane-gathers/clk/CU  useful %6.2f TB/s\n", name,
         table_bytes >> 10, blocks_per_cu, gathers / ms * 1e-6, gathers / (ms * 1e-3) / 256 / (clock_ghz() * 1e9),
         gathers * 16 * PIECES / ms * 1e-9);
}

static void bench_l2() {
  uint8_t* tab; u32x4* out;
  const size_t maxb = 64u << 20;
  CK(hipMalloc(&tab, maxb)); CK(hipMalloc(&out, (size_t)256 * 32 * 256 * 16));
  hipLaunchKernelGGL(fill_u32, dim3(2048), dim3(256), 0, 0, (uint32_t*)tab, maxb / 4, 1u);
  CK(hipDeviceSynchronize());
  printf("# clock %.2f GHz\n", clock_ghz());
  for (size_t kb : {16, 64, 256, 1024, 2048, 4096, 16384, 65536})
    run_l2<0, 1, false>("buffer/default", tab, kb << 10, 8, out);
  for (int bpc : {1, 2, 4, 8}) run_l2<0, 1, false>("buffer/default", tab, 1 << 20, bpc, out);
  run_l2<2, 1, false>("buffer/nt", tab, 1 << 20, 8, out);
  run_l2<16, 1, false>("buffer/sc1", tab, 1 << 20, 8, out);
  run_l2<1, 1, false>("buffer/sc0", tab, 1 << 20, 8, out);
  run_l2<17, 1, false>("buffer/sc0sc1", tab, 1 << 20, 8, out);
  run_l2<0, 1, true>("flat/default", tab, 1 << 20, 8, out);
  run_l2<0, 2, false>("buffer/default 32B", tab, 2 << 20, 8, out);
  run_l2<0, 4, false>("buffer/default 64B", tab, 4 << 20, 8, out);
  CK(hipFree(tab)); CK(hipFree(out));
}

template <int MODE>
static void run_lds(const char* name, size_t table_bytes, int threads, u32x4* out) {
  const int blocks = 256, iters = 1024;
  const uint32_t mask = (uint32_t)(table_bytes / 16) - 1;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ldsgather<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)table_bytes));
  Timer t;
  hipLaunchKernelGGL((ldsgather<MODE>), dim3(blocks), dim3(threads), table_bytes, 0, mask, iters, out);
  CK(hipDeviceSynchronize());
  t.start();
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((ldsgather<MODE>), dim3(blocks), dim3(threads), table_bytes, 0, mask, iters, out);
  const float ms = t.stop_ms() / reps;
  const double gathers = (double)blocks * threads * iters;
  printf("ldsgather %-14s table %4zu KiB  %4d thr/CU  %8.1f Ggather/s  %6.2f lane-gathers/clk/CU\n", name, table_bytes >> 10,
         threads, gathers / ms * 1e-6, gathers / (ms * 1e-3) / 256 / (clock_ghz() * 1e9));
}

static void bench_lds() {
  u32x4* out;
  CK(hipMalloc(&out, (size_t)256 * 1024 * 16));
  for (int thr : {256, 512, 1024}) {
    run_lds<0>("random", 8 << 10, thr, out);
    run_lds<1>("conflict-free", 8 << 10, thr, out);
  }
  run_lds<2>("broadcast", 8 << 10, 1024, out);
  run_lds<0>("random", 4 << 10, 1024, out);
  run_lds<0>("random", 128 << 10, 1024, out);
  run_lds<1>("conflict-free", 128 << 10, 1024, out);
  CK(hipFree(out));
}

static void bench_rates() {
  float* out; u32x4* out4;
  CK(hipMalloc(&out, (size_t)256 * 1024 * 4)); CK(hipMalloc(&out4, (size_t)256 * 1024 * 16));
  const int iters = 4096;
  auto run = [&](const char* name, auto kern, int threads, double per_iter) {
    Timer t;
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, iters, out);
    CK(hipDeviceSynchronize());
    t.start();
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, iters, out);
    const float ms = t.stop_ms() / 5;
    const double waves_per_simd = threads / 256.0;
    const double cyc = ms * 1e-3 * clock_ghz() * 1e9 / (iters * per_iter * waves_per_simd);
    printf("valu %-16s %4d thr/CU  %7.2f cycles per wave-instruction per SIMD\n", name, threads, cyc);
  };
  for (int thr : {256, 512, 1024}) {
    run("v_dot2c_f32_f16", valu_rate<0>, thr, 32);
    run("v_fma_f32", valu_rate<1>, thr, 32);
    run("v_and_b32_sdwa", valu_rate<2>, thr, 32);
    run("v_and_b32", valu_rate<3>, thr, 32);
    run("v_lshrrev_b32", valu_rate<4>, thr, 32);
    run("v_pk_lshlrev_b16", valu_rate<5>, thr, 32);
    run("v_bfe_u32", valu_rate<6>, thr, 32);
    run("v_and_b32 literal", valu_rate<7>, thr, 32);
    run("v_xor_b32", valu_rate<8>, thr, 32);
  }
  auto runl = [&](const char* name, auto kern, int threads) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    Timer t;
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 65536, 0, iters, out4);
    CK(hipDeviceSynchronize());
    t.start();
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 65536, 0, iters, out4);
    const float ms = t.stop_ms() / 5;
    const double reads = (double)iters * 8 * (threads / 64);  // wave-instructions per CU
    printf("lds  %-16s %4d thr/CU  %7.2f cycles per ds_read_b128 wave-instruction per CU  (%.1f B/clk/CU)\n", name, threads,
           ms * 1e-3 * clock_ghz() * 1e9 / reads, reads * 1024 / (ms * 1e-3 * clock_ghz() * 1e9));
  };
  for (int thr : {256, 512, 1024}) {
    runl("random", lds_rate<0>, thr);
    runl("conflict-free", lds_rate<1>, thr);
  }
  CK(hipFree(out)); CK(hipFree(out4));
}

static void bench_stream() {
  const size_t bytes = (size_t)2 << 30;
  u32x4 *buf, *out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, (size_t)256 * 64 * 256 * 16));
  hipLaunchKernelGGL(fill_u32, dim3(4096), dim3(256), 0, 0, (uint32_t*)buf, bytes / 4, 3u);
  CK(hipDeviceSynchronize());
  for (int bpc : {4, 8, 16, 32}) {
    Timer t;
    hipLaunchKernelGGL(stream_read, dim3(256 * bpc), dim3(256), 0, 0, buf, bytes / 16, out);
    CK(hipDeviceSynchronize());
    t.start();
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(stream_read, dim3(256 * bpc), dim3(256), 0, 0, buf, bytes / 16, out);
    const float ms = t.stop_ms() / 3;
    printf("stream nt-read 2 GiB  %2d blk/CU  %.2f TB/s\n", bpc, bytes / ms * 1e-9);
  }
  CK(hipFree(buf)); CK(hipFree(out));
}

// Zipf-distributed codes (what k-means + beam search leaves behind is not uniform: src/aq.py:286-356 of the reference):
// rank r is drawn with probability ~ (r + 1)^-alpha (inverse CDF, binary search), label = perm[rank] (identity: labels
// sorted by frequency; a random permutation otherwise).
__global__ void fill_zipf(uint16_t* codes, size_t n, const float* cdf, const uint16_t* perm, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed * 0x9e3779b9u;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    int lo = 0, hi = 65535;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    codes[i] = perm[lo];
  }
}
static double g_zipf = -1.0;       // < 0: uniform codes (fill_u32)
static bool g_zipf_sorted = false;  // labels sorted by frequency
static int g_prepack_flags = 0;     // AQLM_HIP_PREPACK_*
static float* g_zipf_cdf = nullptr;
static uint16_t* g_zipf_perm = nullptr;
static void zipf_setup(double alpha, bool sorted_labels) {
  g_zipf = alpha;
  g_zipf_sorted = sorted_labels;
  if (!g_zipf_cdf) { CK(hipMalloc(&g_zipf_cdf, 65536 * 4)); CK(hipMalloc(&g_zipf_perm, 65536 * 2)); }
  if (alpha < 0) return;
  std::vector<double> p(65536);
  double z = 0;
  for (int k = 0; k < 65536; ++k) { p[k] = pow((double)(k + 1), -alpha); z += p[k]; }
  std::vector<float> cdf(65536);
  double run = 0;
  for (int k = 0; k < 65536; ++k) { run += p[k] / z; cdf[k] = (float)run; }
  cdf[65535] = 2.0f;
  std::vector<uint16_t> perm(65536);
  for (int k = 0; k < 65536; ++k) perm[k] = (uint16_t)k;
  if (!sorted_labels) {
    uint64_t st = 0x9e3779b97f4a7c15ull;
    for (int k = 65535; k > 0; --k) {
      st = st * 6364136223846793005ull + 1442695040888963407ull;
      const int j = (int)((st >> 33) % (uint64_t)(k + 1));
      std::swap(perm[k], perm[j]);
    }
  }
  CK(hipMemcpy(g_zipf_cdf, cdf.data(), 65536 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(g_zipf_perm, perm.data(), 65536 * 2, hipMemcpyHostToDevice));
}

// ---------------------------------------------------------------- library kernels through the C ABI
struct Layer {
  void *codes, *cb, *scales, *x, *y;
  void* packed = nullptr;
  aqlm_hip_packed_desc desc;
};

struct Scheme {
  const char* name;
  int K, nbits, g;
  bool lds = false;  // (unused: the slice-scan experiment was removed)
  bool packed = false;  // route 1x16 through aqlm_hip_gemv_1x16_packed
  bool lut = false;     // route 8x8 through aqlm_hip_gemv_8x8_lut
  bool planar = false;  // ... on planar codes (aqlm_hip_gemv_8x8_lut_planar)
};
static void* g_ws = nullptr;
static size_t g_ws_bytes = 0;
static bool g_lut_fused = true;  // 8x8 look-up-table scheme: single-kernel form (cells in g_ws) or main + finalize

static size_t algo_bytes(int in, int out, const Scheme& s, int batch) {
  size_t n = (size_t)out * (in / s.g) * s.K * (s.nbits <= 8 ? 1 : 2);
  n += (size_t)s.K * ((size_t)1 << s.nbits) * s.g * 2;
  n += (size_t)batch * in * 2 + (size_t)batch * out * 2 + (size_t)out * 2;
  return n;
}

static bool g_chain = false;  // packed 1x16: name the next layer of the graph (chain prefetch)
static int launch_layer(const Scheme& s, const Layer& L, int in, int out, int batch, hipStream_t st, const Layer* next = nullptr) {
  if (s.nbits == 16 && s.packed && g_chain && next)
    return aqlm_hip_gemv_1x16_packed_chain(&L.desc, L.packed, L.cb, L.scales, nullptr, L.x, L.y, batch, in, out, AQLM_HIP_F16, g_ws, g_ws_bytes,
                                           &next->desc, next->packed, next->cb, st);
  if (s.lut && batch > 1)  // 2+ rows: one launch of rows x the single-row workgroups (single-kernel form; cells in g_ws)
    return aqlm_hip_gemv_8x8_lut_batch(s.planar ? L.packed : L.codes, L.cb, L.scales, nullptr, L.x, L.y, out, in, s.g, batch, in, out, AQLM_HIP_F16,
                                       s.planar ? 1 : 0, 1.0f, g_ws, g_ws_bytes, st);
  if (s.lut && s.planar)
    return aqlm_hip_gemv_8x8_lut_planar(L.packed, L.cb, L.scales, nullptr, L.x, L.y, out, in, s.g, AQLM_HIP_F16, 1.0f, g_ws, g_ws_bytes, g_lut_fused ? 1 : 0, st);
  if (s.lut && g_lut_fused)  // g_ws is zero-filled before every variant and left zero by every fused call
    return aqlm_hip_gemv_8x8_lut_fused(L.codes, L.cb, L.scales, nullptr, L.x, L.y, out, in, s.g, AQLM_HIP_F16, g_ws, g_ws_bytes, st);
  if (s.lut)
    return aqlm_hip_gemv_8x8_lut(L.codes, L.cb, L.scales, nullptr, L.x, L.y, out, in, s.g, AQLM_HIP_F16, g_ws, g_ws_bytes, st);
  if (s.nbits == 16 && s.packed)
    return aqlm_hip_gemv_1x16_packed(&L.desc, L.packed, L.cb, L.scales, nullptr, L.x, L.y, batch, in, out, AQLM_HIP_F16, g_ws, g_ws_bytes, st);
  if (s.nbits == 16)
    return aqlm_hip_gemv_1x16(L.codes, L.cb, L.scales, nullptr, L.x, L.y, out, in, s.g, batch, in, out, AQLM_HIP_F16, st);
  return aqlm_hip_gemv_kx8(L.codes, L.cb, L.scales, nullptr, L.x, L.y, out, in, s.K, s.g, batch, in, out, AQLM_HIP_F16, st);
}

// returns mean microseconds per launch over `reps` graph replays of `layers.size()` launches
static double time_graph(const Scheme& s, const std::vector<Layer>& layers, int in, int out, int batch, int reps) {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (size_t i = 0; i < layers.size(); ++i)
    if (int rc = launch_layer(s, layers[i], in, out, batch, st, &layers[(i + 1) % layers.size()])) { fprintf(stderr, "launch failed rc=%d: %s\n", rc, aqlm_hip_last_error()); exit(3); }
  CK(hipStreamSynchronize(st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (size_t i = 0; i < layers.size(); ++i) launch_layer(s, layers[i], in, out, batch, st, &layers[(i + 1) % layers.size()]);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  Timer t;
  t.start(st);
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
  const float ms = t.stop_ms(st);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
  return ms * 1e3 / ((double)reps * layers.size());
}

static double time_empty_graph(int n, int reps) {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  Timer t; t.start(st);
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
  const float ms = t.stop_ms(st);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
  return ms * 1e3 / ((double)reps * n);
}

static std::vector<Layer> make_layers(const Scheme& s, int in, int out, int batch, int n) {
  std::vector<Layer> v(n);
  const size_t code_bytes = (size_t)out * (in / s.g) * s.K * (s.nbits <= 8 ? 1 : 2);
  const size_t cb_bytes = (size_t)s.K * ((size_t)1 << s.nbits) * s.g * 2;
  for (int i = 0; i < n; ++i) {
    Layer& L = v[i];
    CK(hipMalloc(&L.codes, code_bytes)); CK(hipMalloc(&L.cb, cb_bytes)); CK(hipMalloc(&L.scales, (size_t)out * 2));
    CK(hipMalloc(&L.x, (size_t)batch * in * 2)); CK(hipMalloc(&L.y, (size_t)batch * out * 2));
    if (g_zipf >= 0 && s.nbits == 16)
      hipLaunchKernelGGL(fill_zipf, dim3(1024), dim3(256), 0, 0, (uint16_t*)L.codes, code_bytes / 2, g_zipf_cdf, g_zipf_perm, 17u * i + 1);
    else
      hipLaunchKernelGGL(fill_u32, dim3(1024), dim3(256), 0, 0, (uint32_t*)L.codes, code_bytes / 4, 17u * i + 1);
    hipLaunchKernelGGL(fill_half, dim3(256), dim3(256), 0, 0, (uint32_t*)L.cb, cb_bytes / 4, 31u * i + 2);
    hipLaunchKernelGGL(fill_half, dim3(64), dim3(256), 0, 0, (uint32_t*)L.x, (size_t)batch * in / 2, 7u * i + 3);
    hipLaunchKernelGGL(fill_one_half, dim3(64), dim3(256), 0, 0, (uint16_t*)L.scales, (size_t)out);
  }
  CK(hipDeviceSynchronize());
  if (s.planar) {
    const size_t pb = aqlm_hip_8x8_planar_bytes(out, in, s.g);
    for (auto& L : v) {
      CK(hipMalloc(&L.packed, pb));
      if (int rc = aqlm_hip_8x8_planar_pack(L.codes, out, in, s.g, L.packed, pb, nullptr)) { fprintf(stderr, "planar pack rc=%d %s\n", rc, aqlm_hip_last_error()); exit(4); }
    }
    CK(hipDeviceSynchronize());
  }
  if (s.packed) {
    const size_t pb = aqlm_hip_prepack_1x16_bytes(out, in, s.g);
    for (auto& L : v) {
      CK(hipMalloc(&L.packed, pb));
      if (int rc = aqlm_hip_prepack_1x16_ex(L.codes, out, in, s.g, L.packed, pb, &L.desc, g_prepack_flags, nullptr)) { fprintf(stderr, "prepack rc=%d %s\n", rc, aqlm_hip_last_error()); exit(4); }
      L.desc.codebook_absmax = 1.0f;  // fill_half draws from [-1, 1): the fused finalize may rely on it
      if (int rc = aqlm_hip_packed_set_codebook(&L.desc, L.packed, L.cb, nullptr)) { fprintf(stderr, "set_codebook rc=%d %s\n", rc, aqlm_hip_last_error()); exit(4); }
    }
    CK(hipDeviceSynchronize());
    {
      int mn = 255, mx = 0;
      for (int k = 0; k < (1 << v[0].desc.slices_log2); ++k) { mn = std::min(mn, (int)v[0].desc.slice_groups[k]); mx = std::max(mx, (int)v[0].desc.slice_groups[k]); }
      printf("# packed %d->%d: flags %u (1 relabelled, 2 variable geometry), workgroups per slice %d..%d, rows per group <= %d\n", in, out,
             v[0].desc.flags, mn, mx, v[0].desc.rows_per_group);
    }
    printf("# packed %d->%d: waves %d steps %d entry bytes %d x copies %d, %.3f B per code (capacity %.1f MB, used %.1f MB)\n", in, out,
           v[0].desc.waves, v[0].desc.steps, v[0].desc.entry_bytes, (int)v[0].desc.x_copies, (double)v[0].desc.used_bytes / ((double)out * (in / s.g)), pb * 1e-6, v[0].desc.used_bytes * 1e-6);
  }
  return v;
}

static void free_layers(std::vector<Layer>& v) {
  for (auto& L : v) { hipFree(L.codes); hipFree(L.cb); hipFree(L.scales); hipFree(L.x); hipFree(L.y); if (L.packed) hipFree(L.packed); }
}

struct Scheme; static void check_packed(const Scheme& s, const Layer& L, int in, int out, const Layer* next = nullptr);
static void bench_gemv(int argc, char** argv) {
  g_ws_bytes = (size_t)32 * 8 * 32768 * 4 + (1u << 22);
  CK(hipMalloc(&g_ws, g_ws_bytes));
  const Scheme S1x16P{"1x16g8P", 1, 16, 8, false, true};
  const Scheme S1x16g16P{"1x16g16P", 1, 16, 16, false, true};
  const Scheme S8x8L{"8x8g32LUT", 8, 8, 32, false, false, true};
  const Scheme S8x8LP{"8x8g32LUTP", 8, 8, 32, false, false, true, true};  // planar codes
  const Scheme S1x16{"1x16g8", 1, 16, 8}, S2x8{"2x8g8", 2, 8, 8}, S1x8{"1x8g8", 1, 8, 8}, S8x8{"8x8g32", 8, 8, 32}, S1x16g16{"1x16g16", 1, 16, 16};
  struct Case { Scheme s; int in, out; };
  std::vector<Case> cases = {{S1x16P, 4096, 4096}, {S1x16P, 4096, 11008}, {S1x16P, 4096, 14336}, {S1x16P, 14336, 4096}, {S1x16P, 4096, 1024}, {S1x16P, 8192, 28672}, {S1x16P, 1024, 28672}, {S1x16P, 2048, 28672}, {S1x16P, 8192, 8192}, {S1x16P, 28672, 8192}, {S1x16P, 8192, 1024},
                             {S1x16, 4096, 4096}, {S1x16, 4096, 11008}, {S1x16, 4096, 14336}, {S1x16, 14336, 4096}, {S1x16, 4096, 1024},
                             {S1x16, 8192, 28672}, {S1x16, 1024, 28672}, {S1x16g16, 4096, 4096}, {S1x16g16, 4096, 11008}, {S1x16g16, 8192, 28672},
                             {S1x16g16P, 4096, 4096}, {S1x16g16P, 4096, 11008}, {S1x16g16P, 11008, 4096}, {S1x16g16P, 4096, 14336}, {S1x16g16P, 8192, 8192}, {S1x16g16P, 8192, 28672}, {S1x16g16P, 28672, 8192}, {S2x8, 4096, 4096}, {S2x8, 4096, 11008}, {S2x8, 11008, 4096},
                             {S1x8, 4096, 4096}, {S8x8, 4096, 4096}, {S8x8, 4096, 11008}, {S8x8L, 4096, 4096}, {S8x8L, 4096, 11008}, {S8x8L, 11008, 4096},
                             {S8x8LP, 4096, 4096}, {S8x8LP, 4096, 11008}, {S8x8LP, 11008, 4096}};
  const bool quick = argc > 2 && !strcmp(argv[2], "quick");
  const char* only = argc > 3 ? argv[3] : nullptr;  // run only schemes whose name contains this
  const int only_out = argc > 4 ? atoi(argv[4]) : 0;
  const double gap = time_empty_graph(128, 20);
  printf("# empty-kernel graph: %.2f us per launch (launch gap floor)\n", gap);
  printf("%-9s %6s %6s %2s %-26s %9s %9s %8s %8s\n", "scheme", "in", "out", "B", "variant", "cold_us", "warm_us", "coldGB/s", "%8TB/s");
  for (const auto& c : cases) {
    if (only) {  // substring of the scheme name; a trailing '=' asks for the exact name ("8x8g32LUT=" does not match 8x8g32LUTP)
      const size_t n = strlen(only);
      if (n && only[n - 1] == '=' ? !(strlen(c.s.name) == n - 1 && !strncmp(c.s.name, only, n - 1)) : !strstr(c.s.name, only)) continue;
    }
    if (only_out && c.out != only_out) continue;
    const size_t ab1 = algo_bytes(c.in, c.out, c.s, 1);
    int n = (int)((600u << 20) / ab1) + 1;
    if (n > 160) n = 160;
    if (n < 8) n = 8;
    auto layers = make_layers(c.s, c.in, c.out, 8, n);
    if (c.s.packed) check_packed(c.s, layers[0], c.in, c.out);
    std::vector<Layer> one(layers.begin(), layers.begin() + 1);
    std::vector<Layer> warm(16, one[0]);
    struct Var { const char* name; const char* key; int val; };
    std::vector<std::vector<Var>> variants;
    variants.push_back({});
    if (c.s.nbits == 16 && !quick && !c.s.lds && !c.s.packed) {
      variants.push_back({{"aux=nt", "gemv1x16_aux", 2}});
      variants.push_back({{"aux=sc1", "gemv1x16_aux", 16}});
      variants.push_back({{"prefetch_cb", "gemv1x16_prefetch_cb", 1}});
      variants.push_back({{"rpw=2", "gemv_rows_per_wave", 2}});
      variants.push_back({{"rpw=4", "gemv_rows_per_wave", 4}});
      variants.push_back({{"rpw=8", "gemv_rows_per_wave", 8}});
      variants.push_back({{"rpw=4+prefetch", "gemv_rows_per_wave", 4}, {"", "gemv1x16_prefetch_cb", 1}});
    } else if (c.s.packed && !quick) {
      variants.push_back({{"two-kernel finalize", "packed_fused_finalize", 0}});
      if (getenv("MB_W67")) {  // focused A/B: small layers, 5 / 6 / 7 waves, three alternations
        for (int r = 0; r < 3; ++r) { variants.push_back({{"waves=6", "packed_waves", 6}}); variants.push_back({{"waves=7", "packed_waves", 7}}); variants.push_back({{"waves=5", "packed_waves", 5}}); }
      } else if (getenv("MB_ARR")) {  // focused A/B: greedy deal alone (2) vs greedy + local search (1), two alternations, B = 1 and 4
        for (int r = 0; r < 2; ++r) { variants.push_back({{"arrange=2 (greedy only)", "packed_arrange", 2}}); variants.push_back({{"arrange=1", "packed_arrange", 1}}); }
      } else if (getenv("MB_PD")) {  // focused A/B on the default packing: ring depth 3 vs 4, three alternations
        for (int r = 0; r < 3; ++r) { variants.push_back({{"prefetch=4", "packed_prefetch", 4}}); variants.push_back({{"prefetch=3", "packed_prefetch", 3}}); }
      } else if (getenv("MB_AB12")) {  // focused A/B: 12 vs 16 waves, three alternations
        for (int r = 0; r < 3; ++r) { variants.push_back({{"waves=12", "packed_waves", 12}}); variants.push_back({{"waves=16", "packed_waves", 16}}); }
      } else {
      variants.push_back({{"waves=9", "packed_waves", 9}});
      variants.push_back({{"waves=10", "packed_waves", 10}});
      variants.push_back({{"waves=12", "packed_waves", 12}});
      variants.push_back({{"waves=6", "packed_waves", 6}});
      variants.push_back({{"waves=4", "packed_waves", 4}});
      variants.push_back({{"waves=8", "packed_waves", 8}});
      variants.push_back({{"waves=13", "packed_waves", 13}});
      variants.push_back({{"waves=14", "packed_waves", 14}});
      variants.push_back({{"waves=16", "packed_waves", 16}});
      variants.push_back({{"prefetch=3", "packed_prefetch", 3}});
      variants.push_back({{"prefetch=4", "packed_prefetch", 4}});
      variants.push_back({{"entry=3B", "packed_entry_bytes", 3}});
      variants.push_back({{"xcopies=4", "packed_xcopies", 4}});
      variants.push_back({{"prefetch=8", "packed_prefetch", 8}});
      variants.push_back({{"arrange=0", "packed_arrange", 0}});
      }

    } else if (c.s.packed) {  // quick: the round-3 switches
      if (getenv("MB_CHAIN")) {
        variants.push_back({{"rotate=0", "packed_fill_rotate", 0}});
        variants.push_back({{"chain prefetch (2 waves)", "mb_chain", 2}});
        variants.push_back({{"chain prefetch (1 wave)", "mb_chain", 1}});
        variants.push_back({{"chain + rotate=0", "mb_chain", 2}, {"", "packed_fill_rotate", 0}});
      }
      variants.push_back({{"prefetch=8", "packed_prefetch", 8}});
      variants.push_back({{"prefetch=4", "packed_prefetch", 4}});
      variants.push_back({{"prefetch=8", "packed_prefetch", 8}});
      variants.push_back({{"default again", "packed_fill_rotate", 1}});
    } else if (c.s.lut) {
      variants.push_back({{"waves=8", "lut_waves", 8}});
      variants.push_back({{"two-kernel finalize", "mb_lut_two_kernel", 1}});
      variants.push_back({{"waves=16", "lut_waves", 16}});
      variants.push_back({{"waves=8", "lut_waves", 8}});
    } else if (c.s.nbits == 8 && c.s.g == 8) {
      variants.push_back({{"replicas=off", "kx8_replicas", 0}});
      variants.push_back({{"replicas=force", "kx8_replicas", 2}});
    } else if (!quick && !c.s.lds) {
      variants.push_back({{"rpw=2", "gemv_rows_per_wave", 2}});
      variants.push_back({{"rpw=4", "gemv_rows_per_wave", 4}});
      variants.push_back({{"rpw=8", "gemv_rows_per_wave", 8}});
    }
    for (const auto& var : variants) {
      std::string vn = var.empty() ? "default" : "";
      CK(hipMemset(g_ws, 0, g_ws_bytes));
      if (false) {  // (format v4 check, kept for reference)
        std::vector<uint16_t> y0(c.out), y1(c.out);
        CK(hipMemset(layers[0].y, 0xff, (size_t)c.out * 2));
        launch_layer(c.s, layers[0], c.in, c.out, 1, nullptr);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(y0.data(), layers[0].y, (size_t)c.out * 2, hipMemcpyDeviceToHost));
        for (const auto& kv : var) aqlm_hip_set_tuning(kv.key, kv.val);
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(layers[0].y, 0xff, (size_t)c.out * 2));
          const int rc = launch_layer(c.s, layers[0], c.in, c.out, 1, nullptr);
          CK(hipDeviceSynchronize());
          CK(hipMemcpy(y1.data(), layers[0].y, (size_t)c.out * 2, hipMemcpyDeviceToHost));
          size_t bad = 0;
          for (int i = 0; i < c.out; ++i) bad += y0[i] != y1[i];
          printf("# variant check rep %d: rc=%d mismatches=%zu of %d\n", rep, rc, bad, c.out);
        }
      }
      g_lut_fused = true;
      for (const auto& kv : var) {
        if (!strcmp(kv.key, "mb_lut_two_kernel")) g_lut_fused = false;
        else if (!strcmp(kv.key, "mb_chain")) { g_chain = true; aqlm_hip_set_tuning("packed_prefetch_waves", kv.val); }
        else aqlm_hip_set_tuning(kv.key, kv.val);
        vn += kv.name;
      }
      if (c.s.packed && !var.empty() && (!strcmp(var[0].key, "packed_fused_finalize") || !strcmp(var[0].key, "mb_chain") || !strcmp(var[0].key, "packed_fill_rotate"))) check_packed(c.s, layers[0], c.in, c.out, &layers[1]);
      if (c.s.packed && !var.empty() && (!strcmp(var[0].key, "packed_waves") || !strcmp(var[0].key, "packed_arrange") || !strcmp(var[0].key, "packed_entry_bytes") || !strcmp(var[0].key, "packed_xcopies"))) {  // a format parameter: repack
        const size_t pb = aqlm_hip_prepack_1x16_bytes(c.out, c.in, c.s.g);
        const auto pk_t0 = std::chrono::steady_clock::now();
        for (auto& L : layers) {
          if (int rc = aqlm_hip_prepack_1x16(L.codes, c.out, c.in, c.s.g, L.packed, pb, &L.desc, nullptr)) { fprintf(stderr, "prepack rc=%d %s\n", rc, aqlm_hip_last_error()); exit(4); }
          L.desc.codebook_absmax = 1.0f;
        }
        one[0] = layers[0];
        for (auto& w : warm) w = layers[0];
        printf("# repacked: waves %d steps %d entry bytes %d x copies %d, %.3f B per code, %.2f ms per layer\n", layers[0].desc.waves, layers[0].desc.steps,
               layers[0].desc.entry_bytes, (int)layers[0].desc.x_copies, (double)layers[0].desc.used_bytes / ((double)c.out * (c.in / c.s.g)),
               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - pk_t0).count() / (double)layers.size());
      }
      for (int batch : {1, 2, 4, 8}) {
        if (batch > 1 && (!var.empty() || quick) && !(getenv("MB_ARR") && batch == 4)) continue;
        const size_t ab = algo_bytes(c.in, c.out, c.s, batch);
        const double cold = time_graph(c.s, layers, c.in, c.out, batch, 4);
        const double w = time_graph(c.s, warm, c.in, c.out, batch, 20);
        printf("%-9s %6d %6d %2d %-26s %9.2f %9.2f %8.0f %8.1f\n", c.s.name, c.in, c.out, batch, vn.c_str(), cold, w, ab / cold * 1e-3,
               ab / cold * 1e-3 / 80.0);
        fflush(stdout);
      }
      for (const auto& kv : var) {
        if (!strcmp(kv.key, "mb_chain")) { g_chain = false; aqlm_hip_set_tuning("packed_prefetch_waves", 0); continue; }
        if (strcmp(kv.key, "mb_lut_two_kernel")) aqlm_hip_set_tuning(kv.key, (!strcmp(kv.key, "kx8_replicas") || !strcmp(kv.key, "packed_arrange") || !strcmp(kv.key, "packed_fused_finalize") || !strcmp(kv.key, "packed_fill_rotate")) ? 1 : 0);
      }
    }
    free_layers(layers);
  }
}

// ---------------------------------------------------------------- phase trace of the look-up-table kernel (trace build only)
// Needs the library built with -DAQLM_LUT_TRACE (make trace): every wave stamps wall_clock64 (100 MHz) behind the cells.
static void bench_lut_trace(int in, int out, int g, bool planar) {
  g_ws_bytes = (size_t)64 << 20;
  CK(hipMalloc(&g_ws, g_ws_bytes));
  CK(hipMemset(g_ws, 0, g_ws_bytes));
  const Scheme s{"8x8LUT", 8, 8, g, false, false, true, planar};
  const size_t ab1 = algo_bytes(in, out, s, 1);
  int n = (int)((600u << 20) / ab1) + 1;
  if (n > 160) n = 160;
  auto layers = make_layers(s, in, out, 1, n);
  if (const char* w = getenv("MB_LUT_WAVES")) aqlm_hip_set_tuning("lut_waves", atoi(w));
  unsigned long long* tr = (unsigned long long*)g_ws + ((out + 1023) & ~1023);
  const int in_groups = in / g, nslabs = planar ? 8 * ((in_groups + 127) / 128) : (in_groups + 15) / 16, nranges = std::max(1, 256 / nslabs), nblocks = nslabs * nranges;
  std::vector<unsigned long long> h((size_t)nblocks * 16 * 12);
  const char* names[12] = {"entry", "loads issued", "table written", "at the barrier", "table complete", "walk starts", "rows handed in", "end", "walked + staged", "all walked", "", ""};
  const int order[10] = {0, 1, 2, 3, 4, 5, 8, 9, 6, 7};
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(tr, 0, h.size() * 8));
    CK(hipDeviceSynchronize());
    for (int i = 1; i < n; ++i) launch_layer(s, layers[i], in, out, 1, nullptr);
    launch_layer(s, layers[0], in, out, 1, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < h.size(); i += 12) if (h[i]) t0 = std::min(t0, h[i]);
    printf("# 8x8g%d look-up-table kernel%s %d->%d cold, run %d (%d workgroups): us since the first wave's entry (min / mean / max)\n", g, planar ? " (planar codes)" : "", in, out, rep, nblocks);
    for (int kk = 0; kk < 10; ++kk) {
      const int k = order[kk];
      double mn = 1e9, mx = 0, sum = 0; size_t cnt = 0;
      for (size_t i = 0; i < h.size(); i += 12) {
        if (!h[i] || !h[i + k]) continue;
        const double v = (double)(h[i + k] - t0) * 0.01;
        mn = std::min(mn, v); mx = std::max(mx, v); sum += v; ++cnt;
      }
      printf("  %-16s %7.2f %7.2f %7.2f\n", names[k], mn, cnt ? sum / cnt : 0.0, mx);
    }
  }
  free_layers(layers);
}

// ---------------------------------------------------------------- phase trace of the packed kernel (trace build only)
// Needs the library built with -DAQLM_PACKED_TRACE (make trace): the kernel then stamps wall_clock64 (100 MHz) at
// entry / loads issued / LDS filled / first row done / loop done / end into the workspace tail.
static void bench_trace(int in, int out) {
  const Scheme s{"1x16g8P", 1, 16, 8, false, true};
  const size_t ab1 = algo_bytes(in, out, s, 1);
  int n = (int)((600u << 20) / ab1) + 1;
  auto layers = make_layers(s, in, out, 8, n);
  const size_t need = layers[0].desc.codebook_absmax > 0.f ? 0 : (size_t)16 * out * 4;  // fused finalize: no partials ahead of the stamps
  unsigned long long* tr = (unsigned long long*)((char*)g_ws + need);
  const int NWMAX = 16;
  std::vector<unsigned long long> h(256 * NWMAX * 8);
  const int NW = layers[0].desc.waves;
  const char* names[7] = {"entry", "loads issued", "LDS filled (barrier)", "-", "loop done", "2nd barrier", "end"};
  const int NRUN = 8;
  const int dbgs[NRUN] = {0, 0, 0, 0, 0, 1, 2, 2 | 4};  // runs: warm-up, full x 4 (rotated fill on / off alternating), no compute, no stream, no stream + no dots
  const int rots[NRUN] = {1, 1, 0, 1, 0, 1, 1, 1};
  if (const char* pf = getenv("MB_PREFETCH")) aqlm_hip_set_tuning("packed_prefetch", atoi(pf));
  for (int rep = 0; rep < NRUN; ++rep) {
    aqlm_hip_set_tuning("packed_debug", dbgs[rep]);
    aqlm_hip_set_tuning("packed_fill_rotate", rots[rep]);
    // steady state: the traced launch follows n - 1 launches of other layers back to back (they evict layer 0 from every
    // cache, keep the clocks up and the instruction cache warm; every launch stamps the same buffer, the last one stays)
    CK(hipMemset(tr, 0, h.size() * 8));
    CK(hipDeviceSynchronize());
    for (int i = 1; i < n; ++i) launch_layer(s, layers[i], in, out, 1, nullptr);
    launch_layer(s, layers[0], in, out, 1, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < NW; ++w) t0 = std::min(t0, h[(b * NWMAX + w) * 8]);
    printf("# packed %d->%d cold, run %d [debug %d: 0 full, 1 no LDS reads / dots, 2 no entry stream, +4 no dots, +8 no LDS reads; rotated fill %d] (waves %d, steps %d): time since the first wave's entry, us (min / mean / max over %d waves)\n",
           in, out, rep, dbgs[rep], rots[rep], NW, layers[0].desc.steps, 256 * NW);
    for (int i = 0; i < 7; ++i) {
      if (i == 3) continue;
      double mn = 1e9, mx = 0, sum = 0;
      for (int b = 0; b < 256; ++b) for (int w = 0; w < NW; ++w) {
        const double v = (double)(h[(b * NWMAX + w) * 8 + i] - t0) * 0.01;
        mn = std::min(mn, v); mx = std::max(mx, v); sum += v;
      }
      printf("  %-22s %7.2f %7.2f %7.2f\n", names[i], mn, sum / (256 * NW), mx);
    }
    {  // effective shader clock: s_memtime cycles / wall time between a wave's entry and its end
      double sum = 0; int cnt = 0;
      for (int b = 0; b < 256; ++b) for (int w = 0; w < NW; ++w) {
        const double us = (double)(h[(b * NWMAX + w) * 8 + 6] - h[(b * NWMAX + w) * 8 + 0]) * 0.01;
        if (us > 0) { sum += (double)h[(b * NWMAX + w) * 8 + 3] / us; ++cnt; }
      }
      printf("  shader clock (s_memtime cycles per microsecond of a wave's life, mean): %.0f MHz\n", cnt ? sum / cnt : 0.0);
      double wsum = 0, ksum = 0; int wc = 0;
      for (int b = 0; b < 256; ++b) for (int w = 0; w < NW; ++w) { const unsigned long long v = h[(b * NWMAX + w) * 8 + 7]; wsum += (double)(v >> 32); ksum += (double)(v & 0xffffffffull); ++wc; }
      printf("  main-loop steps, shader cycles per wave (mean): waiting for entries %.0f, LDS reads + dots %.0f  (%d steps)\n", wsum / wc, ksum / wc, layers[0].desc.steps / 3 * 3);
    }
    {  // who finishes late?  "loop done" by XCD (block % 8) and by wave index
      printf("  loop done by block %% 8:");
      for (int x = 0; x < 8; ++x) {
        double sum = 0, mx = 0; int cnt = 0;
        for (int b = x; b < 256; b += 8) for (int w = 0; w < NW; ++w) { const double v = (double)(h[(b * NWMAX + w) * 8 + 4] - t0) * 0.01; sum += v; mx = std::max(mx, v); ++cnt; }
        printf("  %.1f/%.1f", sum / cnt, mx);
      }
      printf("\n  loop done by wave:      ");
      for (int w = 0; w < NW; ++w) {
        double sum = 0;
        for (int b = 0; b < 256; ++b) sum += (double)(h[(b * NWMAX + w) * 8 + 4] - t0) * 0.01;
        printf(" %.1f", sum / 256);
      }
      printf("\n  loop done by block / 32 (mean of the workgroup's last wave):");
      for (int q = 0; q < 8; ++q) {
        double sum = 0;
        for (int b = q * 32; b < q * 32 + 32; ++b) { double mx = 0; for (int w = 0; w < NW; ++w) mx = std::max(mx, (double)(h[(b * NWMAX + w) * 8 + 4] - t0) * 0.01); sum += mx; }
        printf(" %.1f", sum / 32);
      }
      printf("\n");
    }
  }
  free_layers(layers);
}

// packed kernel vs the direct kernel on the same layer (quick on-device sanity check; the real parity tests are in tests/)
static void check_packed(const Scheme& s, const Layer& L, int in, int out, const Layer* next) {
  for (int batch : {1, 4, 8}) {
    std::vector<uint16_t> y0((size_t)batch * out), y1((size_t)batch * out);
    CK(hipMemset(L.y, 0xff, y0.size() * 2));
    int rc = aqlm_hip_gemv_1x16(L.codes, L.cb, L.scales, nullptr, L.x, L.y, out, in, s.g, batch, in, out, AQLM_HIP_F16, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y0.data(), L.y, y0.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemset(L.y, 0xff, y0.size() * 2));
    rc |= launch_layer(s, L, in, out, batch, nullptr, next);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y1.data(), L.y, y1.size() * 2, hipMemcpyDeviceToHost));
    auto h2f = [](uint16_t h) { _Float16 f; memcpy(&f, &h, 2); return (double)(float)f; };
    double num = 0, den = 0, worst = 0;
    for (size_t i = 0; i < y0.size(); ++i) { const double d = fabs(h2f(y0[i]) - h2f(y1[i])); num += d; den += fabs(h2f(y0[i])); worst = std::max(worst, d); }
    printf("# check packed vs direct %d->%d B=%d: rc=%d mean-rel %.3e worst-abs %.3e%s\n", in, out, batch, rc, num / den, worst,
           (num / den < 2e-3 && rc == 0) ? "" : "   <-- MISMATCH");
  }
}

// The prepacked 1x16 matvec on skewed code histograms (round 5): Zipf-distributed codes, random and frequency-sorted labels,
// with (a) the format v6 behaviour (no relabelling: packs only while the slices stay within the capacity), (b) relabelling
// alone, (c) relabelling + variable geometry (the default).  Cold protocol of `mb gemv` (hipGraph over distinct layers).
static void bench_skew(int only_out) {
  g_ws_bytes = (size_t)32 * 8 * 32768 * 4 + (1u << 22);
  CK(hipMalloc(&g_ws, g_ws_bytes));
  const Scheme S1x16P{"1x16g8P", 1, 16, 8, false, true};
  struct Shape { int in, out; };
  const Shape shapes[] = {{4096, 4096}, {4096, 11008}, {8192, 28672}};
  printf("%-6s %6s %6s %-8s %-7s %-26s %9s %8s %6s %s\n", "scheme", "in", "out", "alpha", "labels", "mode", "cold_us", "GB/s", "flags", "groups");
  for (const Shape& sh : shapes) {
    if (only_out && sh.out != only_out) continue;
    const size_t ab1 = algo_bytes(sh.in, sh.out, S1x16P, 1);
    int n = (int)((600u << 20) / ab1) + 1;
    n = std::min(std::max(n, 8), sh.out > 20000 ? 12 : 64);
    for (double alpha : {-1.0, 0.5, 0.8, 1.0, 1.2}) {
      for (int srt = 0; srt < (alpha < 0 ? 1 : 2); ++srt) {
        zipf_setup(alpha, srt != 0);
        struct Mode { const char* name; int flags; };
        const Mode modes[] = {{"relabel+vargeom (default)", 0}, {"relabel, uniform geometry", AQLM_HIP_PREPACK_UNIFORM_ONLY},
                              {"v6: labels as they are", AQLM_HIP_PREPACK_NO_RELABEL | AQLM_HIP_PREPACK_UNIFORM_ONLY}};
        for (const Mode& m : modes) {
          if (alpha < 0 && m.flags != 0) continue;
          g_prepack_flags = m.flags;
          // probe: does one layer pack at all?
          {
            void *codes, *packed;
            const size_t code_bytes = (size_t)sh.out * (sh.in / 8) * 2, pb = aqlm_hip_prepack_1x16_bytes(sh.out, sh.in, 8);
            CK(hipMalloc(&codes, code_bytes)); CK(hipMalloc(&packed, pb));
            hipLaunchKernelGGL(fill_zipf, dim3(1024), dim3(256), 0, 0, (uint16_t*)codes, code_bytes / 2, g_zipf_cdf, g_zipf_perm, 1u);
            aqlm_hip_packed_desc d;
            const int rc = alpha < 0 ? 0 : aqlm_hip_prepack_1x16_ex(codes, sh.out, sh.in, 8, packed, pb, &d, m.flags, nullptr);
            hipFree(codes); hipFree(packed);
            if (rc) {
              printf("%-6s %6d %6d %-8.1f %-7s %-26s %9s %8s  -> does not pack (rc %d): direct kernel\n", "1x16g8", sh.in, sh.out, alpha,
                     srt ? "sorted" : "random", m.name, "-", "-", rc);
              continue;
            }
          }
          auto layers = make_layers(S1x16P, sh.in, sh.out, 8, n);
          check_packed(S1x16P, layers[0], sh.in, sh.out);
          CK(hipMemset(g_ws, 0, g_ws_bytes));
          const double us = time_graph(S1x16P, layers, sh.in, sh.out, 1, 20);
          int mn = 255, mx = 0;
          for (int k = 0; k < 16; ++k) { mn = std::min(mn, (int)layers[0].desc.slice_groups[k]); mx = std::max(mx, (int)layers[0].desc.slice_groups[k]); }
          printf("%-6s %6d %6d %-8.1f %-7s %-26s %9.2f %8.0f %6u %d..%d  (waves %d steps %d, %.2f B/code)\n", "1x16g8", sh.in, sh.out, alpha,
                 alpha < 0 ? "-" : (srt ? "sorted" : "random"), m.name, us, ab1 / us * 1e-3, layers[0].desc.flags, mn, mx, layers[0].desc.waves,
                 layers[0].desc.steps, (double)layers[0].desc.used_bytes / ((double)sh.out * (sh.in / 8)));
          fflush(stdout);
          free_layers(layers);
        }
      }
    }
  }
  g_prepack_flags = 0;
  zipf_setup(-1.0, false);
}

// 8 x 8-bit schemes at 1..8 input rows: the look-up-table kernel as one launch of rows x the single-row workgroups (round 5)
// next to the plain LDS kernel (aqlm_hip_gemv_kx8) that served 2+ rows before.  Cold protocol of `mb gemv`.
static void bench_lutrows() {
  g_ws_bytes = (size_t)32 * 8 * 32768 * 4 + (1u << 22);
  CK(hipMalloc(&g_ws, g_ws_bytes));
  const Scheme S8x8LP{"8x8g32LUTP", 8, 8, 32, false, false, true, true}, S8x8L{"8x8g32LUT", 8, 8, 32, false, false, true}, S8x8{"8x8g32", 8, 8, 32};
  struct Shape { int in, out; };
  printf("%-11s %6s %6s %2s %9s %8s %8s\n", "scheme", "in", "out", "B", "cold_us", "vs_B1", "GB/s");
  for (const Shape& sh : {Shape{4096, 4096}, Shape{4096, 11008}, Shape{11008, 4096}}) {
    for (const Scheme* sp : {&S8x8LP, &S8x8L, &S8x8}) {
      const Scheme& s = *sp;
      const size_t ab1 = algo_bytes(sh.in, sh.out, s, 1);
      int n = std::min(std::max((int)((600u << 20) / ab1) + 1, 8), 96);
      auto layers = make_layers(s, sh.in, sh.out, 8, n);
      double us1 = 0;
      for (int B : {1, 2, 3, 4, 6, 8}) {
        CK(hipMemset(g_ws, 0, g_ws_bytes));
        const double us = time_graph(s, layers, sh.in, sh.out, B, 20);
        if (B == 1) us1 = us;
        printf("%-11s %6d %6d %2d %9.2f %8.2f %8.0f\n", s.name, sh.in, sh.out, B, us, us / us1, algo_bytes(sh.in, sh.out, s, B) / us * 1e-3);
        fflush(stdout);
      }
      free_layers(layers);
    }
  }
}

// ---------------------------------------------------------------- large-batch ops through the C ABI
static void bench_gemm(bool nosync) {
  const int in = 4096, out = 4096;
  const Scheme s{"1x16g8", 1, 16, 8};
  printf("%-28s %5s %10s %10s\n", "op", "B", "us", "TFLOP/s");
  for (int B : {16, 32, 64, 128, 256}) {
    auto layers = make_layers(s, in, out, 8, 24);
    void *X, *Y, *W, *ws;
    CK(hipMalloc(&X, (size_t)B * in * 2)); CK(hipMalloc(&Y, (size_t)B * out * 2)); CK(hipMalloc(&W, (size_t)in * out * 2));
    const size_t wsb = aqlm_hip_workspace_bytes(AQLM_HIP_OP_GEMM_1X16_MFMA, B, out, in);
    CK(hipMalloc(&ws, wsb));
    hipLaunchKernelGGL(fill_half, dim3(256), dim3(256), 0, 0, (uint32_t*)X, (size_t)B * in / 2, 99u);
    CK(hipDeviceSynchronize());
    auto time_it = [&](auto fn) {
      hipStream_t st; CK(hipStreamCreate(&st));
      fprintf(stderr, "  eager...\n");
      for (auto& L : layers) { fn(L, st); if (!nosync) CK(hipStreamSynchronize(st)); }
      CK(hipStreamSynchronize(st));
      fprintf(stderr, "  capture...\n");
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (auto& L : layers) fn(L, st);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      fprintf(stderr, "  timing...\n");
      Timer t; t.start(st);
      for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
      const float ms = t.stop_ms(st);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
      return ms * 1e3 / (5.0 * layers.size());
    };
    fprintf(stderr, "B=%d wsb=%zu fused\n", B, wsb);
    aqlm_hip_set_tuning("gemm_variant", 3);  // the K-split LDS-DMA pipeline by name (the default picks a kernel by batch and layer size)
    auto run_fused = [&]() {
      return time_it([&](const Layer& L, hipStream_t st) {
        int rc = aqlm_hip_gemm_1x16_mfma(L.codes, L.cb, L.scales, nullptr, X, Y, B, out, in, 8, in, out, AQLM_HIP_F16, ws, wsb, st);
        if (rc) { fprintf(stderr, "gemm rc=%d %s\n", rc, aqlm_hip_last_error()); exit(5); }
      });
    };
    const double fused = run_fused();
    if (getenv("MB_GEMM_SWEEP")) {  // knock-out runs of the LDS-DMA pipeline (timing only: the results are wrong)
      const struct { int dbg; const char* what; } ko[] = {{1, "no MFMA"}, {3, "no MFMA, no fragment reads"}, {4, "no stores"}, {8, "no X stream"}, {16, "gathers from one line"},
                                                         {7, "no MFMA / reads / stores"}, {15, "DMA of the gathers only"}, {31, "nothing but the skeleton"}, {32, "full, producers without priority"}, {0, "full again"}, {32, "full, producers without priority"}, {0, "full again"}};
      for (const auto& k : ko) {
        aqlm_hip_set_tuning("gemm_debug", k.dbg);
        const double t = run_fused();
        printf("%-28s %5d %10.2f %10s   knock-out %2d: %s\n", "gemm_1x16_mfma (LDS-DMA)", B, t, "-", k.dbg, k.what);
      }
      aqlm_hip_set_tuning("gemm_debug", 0);
    }
    // round-1 register-staged split-K kernel: same entry point behind the tuning knob; cross-check Y against the default first
    std::vector<uint16_t> y0((size_t)B * out), y1((size_t)B * out);
    aqlm_hip_gemm_1x16_mfma(layers[0].codes, layers[0].cb, layers[0].scales, nullptr, X, Y, B, out, in, 8, in, out, AQLM_HIP_F16, ws, wsb, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y0.data(), Y, y0.size() * 2, hipMemcpyDeviceToHost));
    aqlm_hip_set_tuning("gemm_variant", 1);
    CK(hipMemset(Y, 0xff, y1.size() * 2));
    aqlm_hip_gemm_1x16_mfma(layers[0].codes, layers[0].cb, layers[0].scales, nullptr, X, Y, B, out, in, 8, in, out, AQLM_HIP_F16, ws, wsb, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y1.data(), Y, y1.size() * 2, hipMemcpyDeviceToHost));
    {
      auto h2f = [](uint16_t h) { _Float16 f; memcpy(&f, &h, 2); return (float)f; };
      double num = 0, den = 0; size_t same = 0;
      for (size_t i = 0; i < y0.size(); ++i) { num += fabs(h2f(y0[i]) - h2f(y1[i])); den += fabs(h2f(y0[i])); same += y0[i] == y1[i]; }
      printf("# LDS-DMA pipeline vs register-staged kernel: mean-rel diff %.3e, %zu of %zu bit-identical%s\n", num / den, same, y0.size(), num / den < 1e-3 ? "" : "   <-- MISMATCH");
    }
    const double free_us = time_it([&](const Layer& L, hipStream_t st) {
      int rc = aqlm_hip_gemm_1x16_mfma(L.codes, L.cb, L.scales, nullptr, X, Y, B, out, in, 8, in, out, AQLM_HIP_F16, ws, wsb, st);
      if (rc) { fprintf(stderr, "gemm rc=%d %s\n", rc, aqlm_hip_last_error()); exit(5); }
    });
    // 16-row blocks without K split (round 4): same entry, knob 2; cross-check, then time
    aqlm_hip_set_tuning("gemm_variant", 2);
    CK(hipMemset(Y, 0xff, y1.size() * 2));
    aqlm_hip_gemm_1x16_mfma(layers[0].codes, layers[0].cb, layers[0].scales, nullptr, X, Y, B, out, in, 8, in, out, AQLM_HIP_F16, ws, wsb, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y1.data(), Y, y1.size() * 2, hipMemcpyDeviceToHost));
    {
      auto h2f = [](uint16_t h) { _Float16 f; memcpy(&f, &h, 2); return (float)f; };
      double num = 0, den = 0; size_t same = 0;
      for (size_t i = 0; i < y0.size(); ++i) { num += fabs(h2f(y0[i]) - h2f(y1[i])); den += fabs(h2f(y0[i])); same += y0[i] == y1[i]; }
      printf("# LDS-DMA pipeline vs 16-row no-split kernel: mean-rel diff %.3e, %zu of %zu bit-identical%s\n", num / den, same, y0.size(), num / den < 1e-3 ? "" : "   <-- MISMATCH");
    }
    const double r16_us = time_it([&](const Layer& L, hipStream_t st) {
      int rc = aqlm_hip_gemm_1x16_mfma(L.codes, L.cb, L.scales, nullptr, X, Y, B, out, in, 8, in, out, AQLM_HIP_F16, ws, wsb, st);
      if (rc) { fprintf(stderr, "gemm rc=%d %s\n", rc, aqlm_hip_last_error()); exit(5); }
    });
    printf("%-28s %5d %10.2f %10.1f\n", "gemm_1x16_mfma (16-row)", B, r16_us, 2.0 * B * in * out / r16_us * 1e-6);
    aqlm_hip_set_tuning("gemm_variant", 0);
    fprintf(stderr, "B=%d dequant\n", B);
    const double deq = time_it([&](const Layer& L, hipStream_t st) {
      aqlm_hip_dequant_1x16(L.codes, L.cb, L.scales, W, out, in, 8, AQLM_HIP_F16, st);
    });
    const double flop = 2.0 * B * in * out;
    printf("%-28s %5d %10.2f %10.1f\n", "gemm_1x16_mfma (LDS-DMA)", B, fused, flop / fused * 1e-6);
    printf("%-28s %5d %10.2f %10.1f\n", "gemm_1x16_mfma (round 1)", B, free_us, flop / free_us * 1e-6);
    printf("%-28s %5d %10.2f %10s   (reference pipeline = this + a %d x %d x %d library GEMM)\n", "dequant_1x16 alone", B, deq, "-", B, out, in);
    hipFree(X); hipFree(Y); hipFree(W); hipFree(ws);
    free_layers(layers);
  }
}

// ---------------------------------------------------------------- shared-input launches (q/k/v, gate/up) through the C ABI
static void bench_multi() {
  g_ws_bytes = (size_t)16 * 8 * 32768 * 4 + (1u << 22);
  CK(hipMalloc(&g_ws, g_ws_bytes));
  CK(hipMemset(g_ws, 0, g_ws_bytes));
  const Scheme s{"1x16g8P", 1, 16, 8, false, true};
  if (const char* w = getenv("MB_WAVES")) { aqlm_hip_set_tuning("packed_waves", atoi(w)); printf("# layers packed for %d waves\n", atoi(w)); }
  struct Group { const char* name; int in; std::vector<int> outs; };
  const std::vector<Group> groups = {{"3 x 4096->4096", 4096, {4096, 4096, 4096}}, {"Llama-3-8B q/k/v", 4096, {4096, 1024, 1024}},
                                     {"Llama-3-8B gate/up", 4096, {14336, 14336}}, {"Llama-2-7B gate/up", 4096, {11008, 11008}},
                                     {"Llama-3-70B q/k/v", 8192, {8192, 1024, 1024}}};
  printf("%-22s %-34s %9s %9s\n", "group", "variant", "cold_us", "coldGB/s");
  for (const auto& gr : groups) {
    const int nseg = (int)gr.outs.size();
    size_t ab = 0;
    for (int o : gr.outs) ab += algo_bytes(gr.in, o, s, 1);
    int n = (int)((600u << 20) / ab) + 1;
    if (n > 48) n = 48;
    std::vector<std::vector<Layer>> segs(nseg);
    for (int k = 0; k < nseg; ++k) segs[k] = make_layers(s, gr.in, gr.outs[k], 1, n);
    auto time_it = [&](bool one_launch) {
      hipStream_t st; CK(hipStreamCreate(&st));
      auto run = [&]() {
        for (int i = 0; i < n; ++i) {
          if (one_launch) {
            aqlm_hip_segment sg[4]; const aqlm_hip_packed_desc* ds[4];
            for (int k = 0; k < nseg; ++k) {
              const Layer& L = segs[k][i];
              sg[k] = aqlm_hip_segment{L.packed, L.cb, L.scales, nullptr, L.y, gr.outs[k], gr.outs[k], 0};
              ds[k] = &L.desc;
            }
            int rc = aqlm_hip_gemv_1x16_packed_multi(sg, ds, nseg, segs[0][i].x, gr.in, 1, gr.in, AQLM_HIP_F16, g_ws, g_ws_bytes, st);
            if (rc) { fprintf(stderr, "multi rc=%d %s\n", rc, aqlm_hip_last_error()); exit(6); }
          } else {
            for (int k = 0; k < nseg; ++k) {
              const Layer& L = segs[k][i];
              aqlm_hip_gemv_1x16_packed(&L.desc, L.packed, L.cb, L.scales, nullptr, segs[0][i].x, L.y, 1, gr.in, gr.outs[k], AQLM_HIP_F16, g_ws, g_ws_bytes, st);
            }
          }
        }
      };
      run();
      CK(hipStreamSynchronize(st));
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      run();
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      Timer t; t.start(st);
      for (int r = 0; r < 4; ++r) CK(hipGraphLaunch(ge, st));
      const float ms = t.stop_ms(st);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
      return ms * 1e3 / (4.0 * n);
    };
    const double sep = time_it(false);
    printf("%-22s %-34s %9.2f %9.0f\n", gr.name, "separate launches", sep, ab / sep * 1e-3);
    for (int rep = 0; rep < 2; ++rep) {
      aqlm_hip_set_tuning("packed_pipe", 0);
      const double plain = time_it(true);
      printf("%-22s %-34s %9.2f %9.0f\n", gr.name, "one launch, workgroup per segment", plain, ab / plain * 1e-3);
      aqlm_hip_set_tuning("packed_pipe", 1);
      const double pipe = time_it(true);
      printf("%-22s %-34s %9.2f %9.0f\n", gr.name, "one launch, pipelined segments", pipe, ab / pipe * 1e-3);
    }
    {  // every compute wave requests its share of the next slice itself (the mode of 15- / 16-wave layers) for every shape
      aqlm_hip_set_tuning("packed_pipe", 5);
      const double twob = time_it(true);
      printf("%-22s %-34s %9.2f %9.0f\n", gr.name, "pipelined, two barriers / segment", twob, ab / twob * 1e-3);
      aqlm_hip_set_tuning("packed_pipe", 2);
      const double selfdma = time_it(true);
      printf("%-22s %-34s %9.2f %9.0f\n", gr.name, "pipelined, no DMA waves", selfdma, ab / selfdma * 1e-3);
      aqlm_hip_set_tuning("packed_pipe", 1);
    }
    // outputs of the pipelined launch vs separate launches (layer set 0)
    {
      std::vector<std::vector<uint16_t>> ref(nseg);
      for (int k = 0; k < nseg; ++k) {
        const Layer& L = segs[k][0];
        CK(hipMemset(L.y, 0xff, (size_t)gr.outs[k] * 2));
        aqlm_hip_gemv_1x16_packed(&L.desc, L.packed, L.cb, L.scales, nullptr, segs[0][0].x, L.y, 1, gr.in, gr.outs[k], AQLM_HIP_F16, g_ws, g_ws_bytes, nullptr);
        CK(hipDeviceSynchronize());
        ref[k].resize(gr.outs[k]);
        CK(hipMemcpy(ref[k].data(), L.y, (size_t)gr.outs[k] * 2, hipMemcpyDeviceToHost));
        CK(hipMemset(L.y, 0xff, (size_t)gr.outs[k] * 2));
      }
      aqlm_hip_segment sg[4]; const aqlm_hip_packed_desc* ds[4];
      for (int k = 0; k < nseg; ++k) { const Layer& L = segs[k][0]; sg[k] = aqlm_hip_segment{L.packed, L.cb, L.scales, nullptr, L.y, gr.outs[k], gr.outs[k], 0}; ds[k] = &L.desc; }
      int rc = aqlm_hip_gemv_1x16_packed_multi(sg, ds, nseg, segs[0][0].x, gr.in, 1, gr.in, AQLM_HIP_F16, g_ws, g_ws_bytes, nullptr);
      CK(hipDeviceSynchronize());
      size_t bad = 0, tot = 0;
      for (int k = 0; k < nseg; ++k) {
        std::vector<uint16_t> y(gr.outs[k]);
        CK(hipMemcpy(y.data(), segs[k][0].y, (size_t)gr.outs[k] * 2, hipMemcpyDeviceToHost));
        for (int i = 0; i < gr.outs[k]; ++i) { bad += y[i] != ref[k][i]; ++tot; }
      }
      printf("# %s: pipelined vs separate launches rc=%d: %zu of %zu outputs differ%s\n", gr.name, rc, bad, tot, bad ? "   <-- MISMATCH" : "");
    }
    for (auto& v : segs) free_layers(v);
  }
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "all";
  if (const char* tune = getenv("MB_TUNE")) {  // MB_TUNE="packed_entry_bytes=3,packed_xcopies=4": library tuning knobs for the whole run
    std::string t(tune);
    size_t pos = 0;
    while (pos < t.size()) {
      size_t end = t.find(',', pos);
      if (end == std::string::npos) end = t.size();
      const std::string kv = t.substr(pos, end - pos);
      const size_t eq = kv.find('=');
      if (eq != std::string::npos) {
        const int rc = aqlm_hip_set_tuning(kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1));
        printf("# tuning %s (rc %d)\n", kv.c_str(), rc);
      }
      pos = end + 1;
    }
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s  CUs %d  clock %.0f MHz  L2 %d KiB\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1e3, prop.l2CacheSize >> 10);
  if (!strcmp(what, "l2gather") || !strcmp(what, "all")) bench_l2();
  if (!strcmp(what, "ldsgather") || !strcmp(what, "all")) bench_lds();
  if (!strcmp(what, "stream") || !strcmp(what, "all")) bench_stream();
  if (!strcmp(what, "rates") || !strcmp(what, "all")) bench_rates();
  if (!strcmp(what, "gemv") || !strcmp(what, "all")) bench_gemv(argc, argv);
  if (!strcmp(what, "lut_trace")) bench_lut_trace(argc > 2 ? atoi(argv[2]) : 4096, argc > 3 ? atoi(argv[3]) : 4096, argc > 4 ? atoi(argv[4]) : 32, argc > 5 && !strcmp(argv[5], "planar"));
  if (!strcmp(what, "gemm") || !strcmp(what, "all")) bench_gemm(argc > 2 && !strcmp(argv[2], "nosync"));
  if (!strcmp(what, "multi")) bench_multi();
  if (!strcmp(what, "skew")) bench_skew(argc > 2 ? atoi(argv[2]) : 0);
  if (!strcmp(what, "lutrows")) bench_lutrows();
  if (!strcmp(what, "trace")) {
    g_ws_bytes = (size_t)16 * 8 * 32768 * 4 + (1u << 22);
    CK(hipMalloc(&g_ws, g_ws_bytes));
    bench_trace(argc > 2 ? atoi(argv[2]) : 4096, argc > 3 ? atoi(argv[3]) : 4096);
  }
  return 0;
}
