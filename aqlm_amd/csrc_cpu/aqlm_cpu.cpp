// libaqlm_cpu.so -- native CPU kernels of the AQLM QuantizedLinear path (host-side fallback of the MI355X package).
//
// Replaces (behaviour, not code) the reference's numba kernel `numba_gemm_lut` / `aqlm_gemv_lut`
// (inference_lib/src/aqlm/inference_kernels/numba_kernel.py:10-65, benchmark/matmul_benchmark_cpu.py:100-111) for
// 8-bit codebooks, and gives 16-bit codebooks a direct kernel where the reference falls back to a full torch
// dequantisation (kernel_selector.py:99-102).  Plain C ABI (include/aqlm_cpu.h), fp32 in / fp32 out, caller-owned
// buffers, no allocation except the per-call look-up table the caller passes in as scratch.
//
// LUT kernel (K codebooks of 256 entries):   lut[j][c][v] = <codebooks[c][v], x_j>          (in_groups x K x 256 floats)
//                                            y[i] = scales[i] * sum_j sum_c lut[j][c][codes[j][i][c]] + bias[i]
// Codes are expected in the layout the reference permutes them to for this kernel, [in_groups][out][K] uint8
// (inference.py:78-83): consecutive output rows are consecutive bytes, so one input group's codes for a block of
// rows are one contiguous load and its 2-8 KiB table slab stays in L1 while the block is swept.  Parallel over
// blocks of output rows (every thread owns its rows: no write sharing -- the reference kernel races on
// `output_vec[i] +=` inside prange(j), numba_kernel.py:43-46).  The row sweep is vectorised with table gathers.
//
// Direct kernel (one codebook of 2^nbits <= 65536 entries, g = 8 | 16): y[i] = scales[i] * sum_j <cb[code[i][j]], x_j>,
// codebook held in fp32, rows in parallel, 8-wide FMA per code.
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#include <stdlib.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../include/aqlm_cpu.h"

namespace {

inline int resolve_threads(int nthreads) {
#ifdef _OPENMP
  return nthreads > 0 ? nthreads : omp_get_max_threads();
#else
  (void)nthreads;
  return 1;
#endif
}

// lut[j][c][v] for one input vector; in_groups * K * 256 floats
void build_lut(const float* x, const float* codebooks, float* lut, int in_groups, int K, int g, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static) collapse(2)
  for (int j = 0; j < in_groups; ++j) {
    for (int c = 0; c < K; ++c) {
      const float* xj = x + (size_t)j * g;
      const float* cb = codebooks + (size_t)c * 256 * g;
      float* out = lut + ((size_t)j * K + c) * 256;
      for (int v = 0; v < 256; ++v) {
        float s = 0.f;
        for (int k = 0; k < g; ++k) s += cb[(size_t)v * g + k] * xj[k];
        out[v] = s;
      }
    }
  }
}

#if defined(__x86_64__)
// The same table with 256-bit FMAs: eight v per vector, the codebooks read from a transposed copy cbt[c][k][v] (made per call:
// K * 256 * g floats, nothing against the in_groups * K * 256 * g multiply-adds of the build).  Sum over k in the same order
// as build_lut (one rounding per step instead of two).
__attribute__((target("avx2,fma"))) void build_lut_avx2(const float* x, const float* codebooks, float* cbt, float* lut, int in_groups,
                                                        int K, int g, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int c = 0; c < K; ++c)
    for (int v = 0; v < 256; ++v)
      for (int k = 0; k < g; ++k) cbt[((size_t)c * g + k) * 256 + v] = codebooks[((size_t)c * 256 + v) * g + k];
#pragma omp parallel for num_threads(nthreads) schedule(static) collapse(2)
  for (int j = 0; j < in_groups; ++j) {
    for (int c = 0; c < K; ++c) {
      const float* xj = x + (size_t)j * g;
      const float* t = cbt + (size_t)c * g * 256;
      float* out = lut + ((size_t)j * K + c) * 256;
      for (int v = 0; v < 256; v += 32) {  // four independent accumulators
        __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
        for (int k = 0; k < g; ++k) {
          const __m256 xk = _mm256_broadcast_ss(xj + k);
          const float* tk = t + (size_t)k * 256 + v;
          a0 = _mm256_fmadd_ps(_mm256_loadu_ps(tk), xk, a0);
          a1 = _mm256_fmadd_ps(_mm256_loadu_ps(tk + 8), xk, a1);
          a2 = _mm256_fmadd_ps(_mm256_loadu_ps(tk + 16), xk, a2);
          a3 = _mm256_fmadd_ps(_mm256_loadu_ps(tk + 24), xk, a3);
        }
        _mm256_storeu_ps(out + v, a0);
        _mm256_storeu_ps(out + v + 8, a1);
        _mm256_storeu_ps(out + v + 16, a2);
        _mm256_storeu_ps(out + v + 24, a3);
      }
    }
  }
}
#endif

// rows [r0, r1) of y: sum over input groups and codebooks of table look-ups
template <bool AVX2>
#if defined(__x86_64__)
__attribute__((target("avx2,fma")))
#endif
void sweep_rows_impl(const float* lut, const uint8_t* codes, float* acc, int r0, int r1, int in_groups, int out_features, int K) {
  for (int i = r0; i < r1; ++i) acc[i] = 0.f;
  for (int j = 0; j < in_groups; ++j) {
    const float* lj = lut + (size_t)j * K * 256;
    const uint8_t* cj = codes + ((size_t)j * out_features) * K;
    int i = r0;
#if defined(__x86_64__)
    for (; AVX2 && i + 8 <= r1; i += 8) {
      __m256 a = _mm256_loadu_ps(acc + i);
      const uint8_t* p = cj + (size_t)i * K;
      if (K == 1) {
        const __m256i idx = _mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i*)p));
        a = _mm256_add_ps(a, _mm256_i32gather_ps(lj, idx, 4));
      } else if (K == 2) {
        const __m256i w = _mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i*)p));  // 8 x (c0 | c1 << 8)
        const __m256i i0 = _mm256_and_si256(w, _mm256_set1_epi32(255));
        const __m256i i1 = _mm256_srli_epi32(w, 8);
        a = _mm256_add_ps(a, _mm256_i32gather_ps(lj, i0, 4));
        a = _mm256_add_ps(a, _mm256_i32gather_ps(lj + 256, i1, 4));
      } else {
        for (int c = 0; c < K; ++c) {
          alignas(32) int32_t ix[8];
          for (int q = 0; q < 8; ++q) ix[q] = p[(size_t)q * K + c];
          a = _mm256_add_ps(a, _mm256_i32gather_ps(lj + (size_t)c * 256, _mm256_load_si256((const __m256i*)ix), 4));
        }
      }
      _mm256_storeu_ps(acc + i, a);
    }
#endif
    for (; i < r1; ++i) {
      const uint8_t* p = cj + (size_t)i * K;
      float s = acc[i];
      for (int c = 0; c < K; ++c) s += lj[(size_t)c * 256 + p[c]];
      acc[i] = s;
    }
  }
}

#if defined(__x86_64__)
// 16 rows per step with 512-bit gathers (Zen 4 / 5, Sapphire Rapids: the GPU box's EPYC 9575F has a full-width datapath);
// the remainder of a row range and K > 2 go through the AVX2 / scalar code above.
__attribute__((target("avx512f,avx512bw,avx2,fma")))
void sweep_rows_avx512(const float* lut, const uint8_t* codes, float* acc, int r0, int r1, int in_groups, int out_features, int K) {
  for (int i = r0; i < r1; ++i) acc[i] = 0.f;
  const int r16 = r0 + (r1 - r0) / 16 * 16;
  for (int j = 0; j < in_groups; ++j) {
    const float* lj = lut + (size_t)j * K * 256;
    const uint8_t* cj = codes + ((size_t)j * out_features) * K;
    for (int i = r0; i < r16; i += 16) {
      __m512 a = _mm512_loadu_ps(acc + i);
      const uint8_t* p = cj + (size_t)i * K;
      if (K == 1) {
        const __m512i idx = _mm512_cvtepu8_epi32(_mm_loadu_si128((const __m128i*)p));
        a = _mm512_add_ps(a, _mm512_i32gather_ps(idx, lj, 4));
      } else {  // K == 2: 16 x (c0 | c1 << 8)
        const __m512i w = _mm512_cvtepu16_epi32(_mm256_loadu_si256((const __m256i*)p));
        a = _mm512_add_ps(a, _mm512_i32gather_ps(_mm512_and_si512(w, _mm512_set1_epi32(255)), lj, 4));
        a = _mm512_add_ps(a, _mm512_i32gather_ps(_mm512_srli_epi32(w, 8), lj + 256, 4));
      }
      _mm512_storeu_ps(acc + i, a);
    }
    for (int i = r16; i < r1; ++i) {
      const uint8_t* p = cj + (size_t)i * K;
      float s = acc[i];
      for (int c = 0; c < K; ++c) s += lj[(size_t)c * 256 + p[c]];
      acc[i] = s;
    }
  }
}
#endif

void sweep_rows_plain(const float* lut, const uint8_t* codes, float* acc, int r0, int r1, int in_groups, int out_features, int K) {
  for (int i = r0; i < r1; ++i) acc[i] = 0.f;
  for (int j = 0; j < in_groups; ++j) {
    const float* lj = lut + (size_t)j * K * 256;
    const uint8_t* cj = codes + ((size_t)j * out_features) * K;
    for (int i = r0; i < r1; ++i) {
      const uint8_t* p = cj + (size_t)i * K;
      float s = acc[i];
      for (int c = 0; c < K; ++c) s += lj[(size_t)c * 256 + p[c]];
      acc[i] = s;
    }
  }
}

// Scalar look-ups, two input groups per pass over the rows (halves the traffic of the accumulators): no gather instruction --
// on cores with three or four load ports (Zen 5) plain loads outrun the microcoded gathers.  K <= 2.
template <int K>
void sweep_rows_scalar(const float* lut, const uint8_t* codes, float* acc, int r0, int r1, int in_groups, int out_features) {
  for (int i = r0; i < r1; ++i) acc[i] = 0.f;
  int j = 0;
  for (; j + 2 <= in_groups; j += 2) {
    const float* l0 = lut + (size_t)j * K * 256;
    const float* l1 = l0 + (size_t)K * 256;
    const uint8_t* c0 = codes + ((size_t)j * out_features) * K;
    const uint8_t* c1 = c0 + (size_t)out_features * K;
#pragma GCC unroll 4
    for (int i = r0; i < r1; ++i) {
      const uint8_t* p0 = c0 + (size_t)i * K;
      const uint8_t* p1 = c1 + (size_t)i * K;
      float a = l0[p0[0]], b = l1[p1[0]];
      if (K == 2) {
        a += l0[256 + p0[1]];
        b += l1[256 + p1[1]];
      }
      acc[i] += a + b;
    }
  }
  for (; j < in_groups; ++j) {
    const float* l0 = lut + (size_t)j * K * 256;
    const uint8_t* c0 = codes + ((size_t)j * out_features) * K;
    for (int i = r0; i < r1; ++i) {
      const uint8_t* p0 = c0 + (size_t)i * K;
      float a = l0[p0[0]];
      if (K == 2) a += l0[256 + p0[1]];
      acc[i] += a;
    }
  }
}

// The same for 4 / 8 codebooks: one input group per pass, the K look-ups of a row summed as a tree (a chain of K dependent
// adds would be the critical path).
template <int K>
void sweep_rows_scalar_tree(const float* lut, const uint8_t* codes, float* acc, int r0, int r1, int in_groups, int out_features) {
  static_assert(K == 4 || K == 8, "tree written out for 4 and 8");
  for (int i = r0; i < r1; ++i) acc[i] = 0.f;
  for (int j = 0; j < in_groups; ++j) {
    const float* l = lut + (size_t)j * K * 256;
    const uint8_t* c = codes + ((size_t)j * out_features) * K;
#pragma GCC unroll 2
    for (int i = r0; i < r1; ++i) {
      const uint8_t* p = c + (size_t)i * K;
      float t = (l[p[0]] + l[256 + p[1]]) + (l[512 + p[2]] + l[768 + p[3]]);
      if (K == 8) t += (l[1024 + p[4]] + l[1280 + p[5]]) + (l[1536 + p[6]] + l[1792 + p[7]]);
      acc[i] += t;
    }
  }
}

void sweep_rows(const float* lut, const uint8_t* codes, float* acc, int r0, int r1, int in_groups, int out_features, int K) {
  // Which sweep: measured on the GPU box's EPYC 9575F (Zen 5, 1 thread, 2x8g8 4096 -> 11008): scalar look-ups 2.41 ms, AVX-512
  // gathers 3.05 ms; on a Sapphire Rapids Xeon the other way round (4.9 vs 3.8 ms) -- so AMD cores take the scalar sweep.
  // AQLM_CPU_SWEEP=scalar|gather overrides (experiments).
  static const char* const sweep_env = getenv("AQLM_CPU_SWEEP");
#if defined(__x86_64__)
  static const bool amd = __builtin_cpu_is("amd");
#else
  static const bool amd = false;
#endif
  const bool scalar = sweep_env ? !strcmp(sweep_env, "scalar") : (amd || K >= 4);  // 8 codebooks: the tree of scalar look-ups wins on both (Xeon: 5.8 vs 7.4 ms)
  if (scalar) {
    if (K == 1) return sweep_rows_scalar<1>(lut, codes, acc, r0, r1, in_groups, out_features);
    if (K == 2) return sweep_rows_scalar<2>(lut, codes, acc, r0, r1, in_groups, out_features);
    if (K == 4) return sweep_rows_scalar_tree<4>(lut, codes, acc, r0, r1, in_groups, out_features);
    if (K == 8) return sweep_rows_scalar_tree<8>(lut, codes, acc, r0, r1, in_groups, out_features);
    return sweep_rows_plain(lut, codes, acc, r0, r1, in_groups, out_features, K);
  }
#if defined(__x86_64__)
  static const bool has_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
  static const bool has_avx512 = has_avx2 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                                 !getenv("AQLM_CPU_NO_AVX512");
  if (has_avx512 && K <= 2) return sweep_rows_avx512(lut, codes, acc, r0, r1, in_groups, out_features, K);
  if (has_avx2) return sweep_rows_impl<true>(lut, codes, acc, r0, r1, in_groups, out_features, K);
#endif
  sweep_rows_plain(lut, codes, acc, r0, r1, in_groups, out_features, K);
}

// One output row of the direct 1 x n kernel: per code one gather of g floats and g multiply-adds.  Four independent
// accumulator sets -- with one, every code waits for the previous code's add (FMA latency per code: measured 2x the time) --
// and the codebook vectors of the codes 16 ahead are prefetched (the 2 MiB fp32 table of a 16-bit codebook lives in L2 / L3).
template <class CodeAt>
static inline void row_1xn_plain(const float* x, const float* codebook, CodeAt code_at, const float* scales, const float* bias, float* y,
                                 int batch, long xs, long ys, int in_groups, int g, int i) {
  constexpr int U = 4;
  for (int b = 0; b < batch; ++b) {
    const float* xb = x + (size_t)b * xs;
    float acc[U][16];
    for (int u = 0; u < U; ++u)
      for (int k = 0; k < g; ++k) acc[u][k] = 0.f;
    int j = 0;
    for (; j + U <= in_groups; j += U)
      for (int u = 0; u < U; ++u) {
        const float* v = codebook + (size_t)code_at(j + u) * g;
        const float* xj = xb + (size_t)(j + u) * g;
        for (int k = 0; k < g; ++k) acc[u][k] += v[k] * xj[k];
      }
    for (; j < in_groups; ++j) {
      const float* v = codebook + (size_t)code_at(j) * g;
      const float* xj = xb + (size_t)j * g;
      for (int k = 0; k < g; ++k) acc[0][k] += v[k] * xj[k];
    }
    float s = 0.f;
    for (int k = 0; k < g; ++k) s += (acc[0][k] + acc[1][k]) + (acc[2][k] + acc[3][k]);
    y[(size_t)b * ys + i] = s * scales[i] + (bias ? bias[i] : 0.f);
  }
}

#if defined(__x86_64__)
// eight table entries as floats: fp32 table, or fp16 table (half the footprint: 1 MiB for a 16-bit codebook with g = 8 -- it
// stays in L2 where the fp32 table spills to L3; measured 8.3 -> 5.0 ms for 4096 -> 11008 at 1 thread on the build host)
__attribute__((target("avx2,fma,f16c"))) static inline __m256 load8(const float* p) { return _mm256_loadu_ps(p); }
__attribute__((target("avx2,fma,f16c"))) static inline __m256 load8(const uint16_t* p) { return _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)p)); }

template <int G, class TableT, class CodeAt>
__attribute__((target("avx2,fma,f16c"))) static inline void row_1xn_avx2(const float* x, const TableT* codebook, CodeAt code_at, const float* scales,
                                                                        const float* bias, float* y, int batch, long xs, long ys,
                                                                        int in_groups, int i) {
  constexpr int U = 4, AHEAD = 16, V = G / 8;  // V 256-bit vectors per code
  for (int b = 0; b < batch; ++b) {
    const float* xb = x + (size_t)b * xs;
    __m256 acc[U][V];
    for (int u = 0; u < U; ++u)
      for (int h = 0; h < V; ++h) acc[u][h] = _mm256_setzero_ps();
    int j = 0;
    for (; j + U <= in_groups; j += U) {
      if (j + AHEAD + U <= in_groups)
        for (int u = 0; u < U; ++u) _mm_prefetch((const char*)(codebook + (size_t)code_at(j + AHEAD + u) * G), _MM_HINT_T0);
      for (int u = 0; u < U; ++u) {
        const TableT* v = codebook + (size_t)code_at(j + u) * G;
        const float* xj = xb + (size_t)(j + u) * G;
        for (int h = 0; h < V; ++h) acc[u][h] = _mm256_fmadd_ps(load8(v + 8 * h), _mm256_loadu_ps(xj + 8 * h), acc[u][h]);
      }
    }
    for (; j < in_groups; ++j) {
      const TableT* v = codebook + (size_t)code_at(j) * G;
      const float* xj = xb + (size_t)j * G;
      for (int h = 0; h < V; ++h) acc[0][h] = _mm256_fmadd_ps(load8(v + 8 * h), _mm256_loadu_ps(xj + 8 * h), acc[0][h]);
    }
    __m256 t = _mm256_add_ps(_mm256_add_ps(acc[0][0], acc[1][0]), _mm256_add_ps(acc[2][0], acc[3][0]));
    for (int h = 1; h < V; ++h) t = _mm256_add_ps(t, _mm256_add_ps(_mm256_add_ps(acc[0][h], acc[1][h]), _mm256_add_ps(acc[2][h], acc[3][h])));
    float lanes[8];
    _mm256_storeu_ps(lanes, t);
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += lanes[k];
    y[(size_t)b * ys + i] = s * scales[i] + (bias ? bias[i] : 0.f);
  }
}
#endif

#if defined(__x86_64__)
static bool cpu_has_f16c_avx2() {
  static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma") && __builtin_cpu_supports("f16c");
  return ok;
}

// the same row with an fp16 table (the caller checked cpu_has_f16c_avx2())
static void row_1xn_half(const float* x, const uint16_t* codebook, const void* codes, int code_bytes, uint32_t mask, const float* scales,
                         const float* bias, float* y, int batch, long xs, long ys, int in_groups, int g, int i) {
  const size_t row_at = (size_t)i * in_groups;
  const uint16_t* c16 = (const uint16_t*)codes + row_at;
  const uint8_t* c8 = (const uint8_t*)codes + row_at;
  auto at16 = [=](int j) -> uint32_t { return (uint32_t)c16[j] & mask; };
  auto at8 = [=](int j) -> uint32_t { return (uint32_t)c8[j] & mask; };
  if (code_bytes == 2) {
    if (g == 8) return row_1xn_avx2<8>(x, codebook, at16, scales, bias, y, batch, xs, ys, in_groups, i);
    return row_1xn_avx2<16>(x, codebook, at16, scales, bias, y, batch, xs, ys, in_groups, i);
  }
  if (g == 8) return row_1xn_avx2<8>(x, codebook, at8, scales, bias, y, batch, xs, ys, in_groups, i);
  row_1xn_avx2<16>(x, codebook, at8, scales, bias, y, batch, xs, ys, in_groups, i);
}
#endif

static void row_1xn(const float* x, const float* codebook, const void* codes, int code_bytes, uint32_t mask, const float* scales,
                    const float* bias, float* y, int batch, long xs, long ys, int in_groups, int g, int i) {
  const size_t row_at = (size_t)i * in_groups;
  const uint16_t* c16 = (const uint16_t*)codes + row_at;
  const uint8_t* c8 = (const uint8_t*)codes + row_at;
  auto at16 = [=](int j) -> uint32_t { return (uint32_t)c16[j] & mask; };
  auto at8 = [=](int j) -> uint32_t { return (uint32_t)c8[j] & mask; };
#if defined(__x86_64__)
  if (cpu_has_f16c_avx2()) {  // (every AVX2 + FMA core has F16C; the fp32 and the fp16 row share one target set)
    if (code_bytes == 2) {
      if (g == 8) return row_1xn_avx2<8>(x, codebook, at16, scales, bias, y, batch, xs, ys, in_groups, i);
      return row_1xn_avx2<16>(x, codebook, at16, scales, bias, y, batch, xs, ys, in_groups, i);
    }
    if (g == 8) return row_1xn_avx2<8>(x, codebook, at8, scales, bias, y, batch, xs, ys, in_groups, i);
    return row_1xn_avx2<16>(x, codebook, at8, scales, bias, y, batch, xs, ys, in_groups, i);
  }
#endif
  if (code_bytes == 2) return row_1xn_plain(x, codebook, at16, scales, bias, y, batch, xs, ys, in_groups, g, i);
  row_1xn_plain(x, codebook, at8, scales, bias, y, batch, xs, ys, in_groups, g, i);
}

}  // namespace

extern "C" int aqlm_cpu_abi_version(void) { return AQLM_CPU_ABI_VERSION; }

extern "C" int aqlm_cpu_max_threads(void) { return resolve_threads(0); }

extern "C" size_t aqlm_cpu_lut_scratch_floats(int in_features, int num_codebooks, int in_group_size) {
  if (in_features <= 0 || num_codebooks <= 0 || in_group_size <= 0 || in_features % in_group_size) return 0;
  // the table + the codebooks transposed to [c][k][256] (the vectorised table build reads them along v)
  return (size_t)(in_features / in_group_size) * num_codebooks * 256 + (size_t)num_codebooks * 256 * in_group_size;
}

extern "C" int aqlm_cpu_gemv_lut_kx8(const float* x, const float* codebooks, const uint8_t* codes_alt, const float* scales,
                                     const float* bias, float* y, int batch, long x_row_stride, long y_row_stride,
                                     int in_features, int out_features, int num_codebooks, int in_group_size, float* scratch,
                                     int nthreads) {
  if (!x || !codebooks || !codes_alt || !scales || !y || !scratch) return AQLM_CPU_E_INVALID;
  if (batch < 1 || in_features <= 0 || out_features <= 0 || num_codebooks < 1 || in_group_size < 1 ||
      in_features % in_group_size)
    return AQLM_CPU_E_INVALID;
  const int in_groups = in_features / in_group_size, K = num_codebooks;
  const int nt = resolve_threads(nthreads);
  for (int b = 0; b < batch; ++b) {  // one table per input row (the reference loops over rows too, numba_kernel.py:55-62)
    const float* xb = x + (size_t)b * x_row_stride;
    float* yb = y + (size_t)b * y_row_stride;
#if defined(__x86_64__)
    static const bool lut_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma") && !getenv("AQLM_CPU_NO_AVX2_LUT");
    if (lut_avx2) build_lut_avx2(xb, codebooks, scratch + (size_t)in_groups * K * 256, scratch, in_groups, K, in_group_size, nt);
    else
#endif
      build_lut(xb, codebooks, scratch, in_groups, K, in_group_size, nt);
    // rows per task: every task streams the whole table (in_groups * K KiB) once, so few big tasks -- 4096 rows = 16 KiB of
    // accumulators next to one group's table slab in L1 (measured at 1 thread, 2x8g8 8192 -> 28672: 26.3 ms with 256 rows, 18.7
    // with 2048, 17.6 with 4096) --, but at least ~4 per thread for the dynamic schedule
    int block = nt == 1 ? 4096 : (out_features / (4 * nt) + 63) / 64 * 64;
    block = block < 512 ? 512 : (block > 4096 ? 4096 : block);
    const int nblocks = (out_features + block - 1) / block;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int t = 0; t < nblocks; ++t) {
      const int r0 = t * block, r1 = r0 + block < out_features ? r0 + block : out_features;
      sweep_rows(scratch, codes_alt, yb, r0, r1, in_groups, out_features, K);
      for (int i = r0; i < r1; ++i) yb[i] = yb[i] * scales[i] + (bias ? bias[i] : 0.f);
    }
  }
  return 0;
}

extern "C" int aqlm_cpu_gemv_1xn(const float* x, const float* codebook, const void* codes, int code_bytes, const float* scales,
                                 const float* bias, float* y, int batch, long x_row_stride, long y_row_stride, int in_features,
                                 int out_features, int nbits, int in_group_size, int nthreads) {
  if (!x || !codebook || !codes || !scales || !y) return AQLM_CPU_E_INVALID;
  if (batch < 1 || in_features <= 0 || out_features <= 0 || nbits < 1 || nbits > 16 || (code_bytes != 1 && code_bytes != 2) ||
      (in_group_size != 8 && in_group_size != 16) || in_features % in_group_size)
    return AQLM_CPU_E_UNSUPPORTED;
  const int g = in_group_size, in_groups = in_features / g;
  const uint32_t mask = (1u << nbits) - 1u;
  const int nt = resolve_threads(nthreads);
#pragma omp parallel for num_threads(nt) schedule(static)
  for (int i = 0; i < out_features; ++i)
    row_1xn(x, codebook, codes, code_bytes, mask, scales, bias, y, batch, x_row_stride, y_row_stride, in_groups, g, i);
  return 0;
}

extern "C" int aqlm_cpu_gemv_1xn_f16(const float* x, const uint16_t* codebook_f16, const void* codes, int code_bytes, const float* scales,
                                     const float* bias, float* y, int batch, long x_row_stride, long y_row_stride, int in_features,
                                     int out_features, int nbits, int in_group_size, int nthreads) {
  if (!x || !codebook_f16 || !codes || !scales || !y) return AQLM_CPU_E_INVALID;
  if (batch < 1 || in_features <= 0 || out_features <= 0 || nbits < 1 || nbits > 16 || (code_bytes != 1 && code_bytes != 2) ||
      (in_group_size != 8 && in_group_size != 16) || in_features % in_group_size)
    return AQLM_CPU_E_UNSUPPORTED;
#if defined(__x86_64__)
  if (!cpu_has_f16c_avx2()) return AQLM_CPU_E_UNSUPPORTED;  // the caller falls back to the fp32 table
  const int g = in_group_size, in_groups = in_features / g;
  const uint32_t mask = (1u << nbits) - 1u;
  const int nt = resolve_threads(nthreads);
#pragma omp parallel for num_threads(nt) schedule(static)
  for (int i = 0; i < out_features; ++i)
    row_1xn_half(x, codebook_f16, codes, code_bytes, mask, scales, bias, y, batch, x_row_stride, y_row_stride, in_groups, g, i);
  return 0;
#else
  return AQLM_CPU_E_UNSUPPORTED;
#endif
}
