/*
 * CPU ORACLE (C restatement) for the AQLM matvec path  --  TEST / BASELINE INFRASTRUCTURE ONLY.
 *
 * Nothing under aqlm_amd/ or aqlm/ links or loads this file.  It is used by
 *   - tests/ (checked against oracle/aqlm_oracle.py, which is pinned to the reference's outputs), and
 *   - bench.py's `cpu_baseline` leg ("kind": "port"), timed on the host cores.
 *
 * Parity status: PINNED (transitively) -- see oracle/aqlm_oracle.py header.
 *
 * Two restatements (all file:line below are relative to /root/reference):
 *
 *  1. aqlm_oracle_lut_gemv_f32  -- the reference's CPU kernel, the numba LUT gemv
 *       inference_lib/src/aqlm/inference_kernels/numba_kernel.py:37-48
 *       benchmark/matmul_benchmark_cpu.py:100-111 (inline copy that the benchmark times)
 *     lut = x.reshape(-1,g) @ codebooks.reshape(-1,g).T ; y[i] += lut[j,c,codes_alt[j,i,c]] ; y *= scales
 *     Codes are in the CPU layout [in_groups, out, K] (inference.py:78-83), unsigned view
 *     (numba_kernel.py:59).  The reference only runs 8-bit codes here; code_bytes==2 extends the same
 *     loop to 16-bit codes.  The reference parallelises `prange(j)` with an
 *     unsynchronised `y[i] +=`; this restatement parallelises over
 *     disjoint output ranges instead, so it is race-free and deterministic.
 *
 *  2. aqlm_oracle_dequant_gemv_f32 -- what the reference executes on CPU for 1x16 (kernel_selector.py:99-102):
 *       inference_lib/src/aqlm/inference_kernels/dequantization.py:9-21  (dequantize + F.linear)
 *       inference_lib/src/aqlm/utils.py:43-70                          (_dequantize_weight)
 *     restated without materialising W:  y[i] = scale[i] * sum_j sum_c <cb[c][code[i,j,c]], x_j> + bias[i].
 *     Codes are in the checkpoint layout [out, in_groups, K].
 *
 * float32 throughout, like the reference's CPU paths (numba_kernel.py:30-32).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline uint32_t load_code(const void* codes, size_t idx, int code_bytes) {
  if (code_bytes == 1) return ((const uint8_t*)codes)[idx];
  return ((const uint16_t*)codes)[idx];
}

/* returns bytes of LUT scratch needed by aqlm_oracle_lut_gemv_f32 */
size_t aqlm_oracle_lut_bytes(int in_features, int num_codebooks, int nbits, int in_group_size) {
  return (size_t)(in_features / in_group_size) * (size_t)num_codebooks * ((size_t)1 << nbits) * sizeof(float);
}

int aqlm_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* numba_kernel.py:37-48.  codes_alt: [in_groups][out][K] unsigned, code_bytes in {1,2}. */
void aqlm_oracle_lut_gemv_f32(const float* x, const float* codebooks, const void* codes_alt, int code_bytes,
                              const float* scales, float* y, int in_features, int out_features,
                              int num_codebooks, int nbits, int in_group_size, float* lut, int nthreads) {
  const int g = in_group_size, K = num_codebooks;
  const int in_groups = in_features / g;
  const size_t cbsize = (size_t)1 << nbits;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  /* lut[j][c][v] = <x_j, codebooks[c][v]>           (numba_kernel.py:39-40) */
#pragma omp parallel for schedule(static)
  for (int j = 0; j < in_groups; ++j) {
    const float* xj = x + (size_t)j * g;
    float* lj = lut + (size_t)j * K * cbsize;
    for (size_t cv = 0; cv < (size_t)K * cbsize; ++cv) {
      const float* e = codebooks + cv * g;
      float s = 0.f;
      for (int t = 0; t < g; ++t) s += xj[t] * e[t];
      lj[cv] = s;
    }
  }
  /* y[i] += lut[j, c, codes_alt[j, i, c]]            (numba_kernel.py:42-46), disjoint i-ranges per thread */
#pragma omp parallel
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
    const int tid = 0, nt = 1;
#endif
    const int chunk = (out_features + nt - 1) / nt;
    const int i0 = tid * chunk, i1 = (i0 + chunk < out_features) ? i0 + chunk : out_features;
    for (int i = i0; i < i1; ++i) y[i] = 0.f;
    for (int j = 0; j < in_groups; ++j) {
      const float* lj = lut + (size_t)j * K * cbsize;
      const size_t rowbase = (size_t)j * out_features * K;
      for (int i = i0; i < i1; ++i) {
        float s = y[i];
        for (int c = 0; c < K; ++c) s += lj[(size_t)c * cbsize + load_code(codes_alt, rowbase + (size_t)i * K + c, code_bytes)];
        y[i] = s;
      }
    }
    for (int i = i0; i < i1; ++i) y[i] *= scales[i]; /* numba_kernel.py:47 */
  }
}

/* dequantization.py:9-21 + utils.py:43-70, fused.  codes: [out][in_groups][K] unsigned view. */
void aqlm_oracle_dequant_gemv_f32(const float* x, const float* codebooks, const void* codes, int code_bytes,
                                  const float* scales, const float* bias /* nullable */, float* y, int in_features,
                                  int out_features, int num_codebooks, int nbits, int in_group_size, int nthreads) {
  const int g = in_group_size, K = num_codebooks;
  const int in_groups = in_features / g;
  const size_t cbsize = (size_t)1 << nbits;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
  for (int i = 0; i < out_features; ++i) {
    float acc = 0.f;
    const size_t rowbase = (size_t)i * in_groups * K;
    for (int j = 0; j < in_groups; ++j) {
      const float* xj = x + (size_t)j * g;
      for (int c = 0; c < K; ++c) {
        const uint32_t v = load_code(codes, rowbase + (size_t)j * K + c, code_bytes);
        const float* e = codebooks + ((size_t)c * cbsize + v) * g;
        float s = 0.f;
        for (int t = 0; t < g; ++t) s += e[t] * xj[t];
        acc += s;
      }
    }
    acc *= scales[i];            /* utils.py:64-65 */
    if (bias) acc += bias[i];    /* F.linear bias, dequantization.py:21 */
    y[i] = acc;
  }
}

/* utils.py:43-70: materialise W[out][in] (float32), scales nullable. */
void aqlm_oracle_dequant_weight_f32(const float* codebooks, const void* codes, int code_bytes, const float* scales,
                                    float* W, int in_features, int out_features, int num_codebooks, int nbits,
                                    int in_group_size) {
  const int g = in_group_size, K = num_codebooks;
  const int in_groups = in_features / g;
  const size_t cbsize = (size_t)1 << nbits;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < out_features; ++i) {
    const size_t rowbase = (size_t)i * in_groups * K;
    for (int j = 0; j < in_groups; ++j) {
      float* w = W + (size_t)i * in_features + (size_t)j * g;
      for (int t = 0; t < g; ++t) w[t] = 0.f;
      for (int c = 0; c < K; ++c) {
        const uint32_t v = load_code(codes, rowbase + (size_t)j * K + c, code_bytes);
        const float* e = codebooks + ((size_t)c * cbsize + v) * g;
        for (int t = 0; t < g; ++t) w[t] += e[t];
      }
      if (scales)
        for (int t = 0; t < g; ++t) w[t] *= scales[i];
    }
  }
}
