/*
 * aqlm_cpu.h -- C ABI of libaqlm_cpu.so: native CPU kernels of the AQLM QuantizedLinear matvec (host-side companion
 * of libaqlm_hip.so).  fp32 activations / outputs, caller-owned buffers, no allocation.
 * Return 0 on success, AQLM_CPU_E_* otherwise.  Paths cited are relative to the reference tree (inference_lib/src/aqlm/).
 * Instruction sets are chosen at run time (SSE2 baseline; AVX2 + FMA, AVX-512 where the CPU has them).  Experiment switches,
 * read once per process: AQLM_CPU_SWEEP=scalar|gather (the look-up sweep of the LUT kernel; default: scalar on AMD cores and
 * for 4 / 8 codebooks, gathers otherwise), AQLM_CPU_NO_AVX512=1, AQLM_CPU_NO_AVX2_LUT=1 (scalar table build).
 */
#ifndef AQLM_CPU_H_
#define AQLM_CPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AQLM_CPU_ABI_VERSION 2
#define AQLM_CPU_E_INVALID (-1)
#define AQLM_CPU_E_UNSUPPORTED (-2)

int aqlm_cpu_abi_version(void);
int aqlm_cpu_max_threads(void); /* OpenMP's default team size on this host */

/*
 * y[b, :] = scales * (sum_j sum_c lut_b[j][c][codes_alt[j][:][c]]) + bias  for K codebooks of 256 entries.
 * Replaces: numba_gemm_lut / aqlm_gemv_lut (inference_kernels/numba_kernel.py:10-65; same kernel inline in
 *           benchmark/matmul_benchmark_cpu.py:100-111).
 * codebooks [K][256][in_group_size] fp32; codes_alt [in_features/in_group_size][out_features][K] uint8 -- the layout the
 * reference permutes `codes` to for this kernel (inference.py:78-83); scales [out], bias [out] or NULL; x / y row strides
 * in elements.  scratch: aqlm_cpu_lut_scratch_floats(...) floats (the per-token table and a transposed copy of the
 * codebooks; contents are undefined between calls).  nthreads <= 0: OpenMP default.
 */
size_t aqlm_cpu_lut_scratch_floats(int in_features, int num_codebooks, int in_group_size);
int aqlm_cpu_gemv_lut_kx8(const float* x, const float* codebooks, const uint8_t* codes_alt, const float* scales,
                          const float* bias, float* y, int batch, long x_row_stride, long y_row_stride, int in_features,
                          int out_features, int num_codebooks, int in_group_size, float* scratch, int nthreads);

/*
 * One codebook of 2^nbits entries (nbits <= 16; codes [out][in/g] in 8- or 16-bit containers, canonical layout),
 * in_group_size 8 or 16: y[b, i] = scales[i] * sum_j <codebook[code[i][j]], x_b[j*g:(j+1)*g]> + bias[i].
 * Replaces: the reference's CPU route for 1x16, dequantize_gemm = _dequantize_weight + F.linear
 *           (inference_kernels/kernel_selector.py:99-102, dequantization.py:9-21) -- without materialising W.
 */
int aqlm_cpu_gemv_1xn(const float* x, const float* codebook, const void* codes, int code_bytes, const float* scales,
                      const float* bias, float* y, int batch, long x_row_stride, long y_row_stride, int in_features,
                      int out_features, int nbits, int in_group_size, int nthreads);

/*
 * The same with the codebook stored as IEEE fp16 (converted with F16C on the fly): half the table, which keeps a 16-bit
 * codebook in L2 -- exact whenever the fp32 codebook values are fp16-representable (AQLM checkpoints store fp16 codebooks), and
 * that is the caller's check.  AQLM_CPU_E_UNSUPPORTED on a CPU without AVX2 + FMA + F16C: use aqlm_cpu_gemv_1xn.  (ABI 2.)
 */
int aqlm_cpu_gemv_1xn_f16(const float* x, const uint16_t* codebook_f16, const void* codes, int code_bytes, const float* scales,
                          const float* bias, float* y, int batch, long x_row_stride, long y_row_stride, int in_features,
                          int out_features, int nbits, int in_group_size, int nthreads);

#ifdef __cplusplus
}
#endif
#endif /* AQLM_CPU_H_ */
