// Micro-test: do ds_read_b128 returns and VALU issue of the SAME wave / SIMD overlap, or add up?  Per iteration a wave issues R reads
// (conflict-free, lane-linear + rotating base) and then N v_dot2c per read on the returned data; 14 waves per CU like the packed kernel.
//   hipcc --offload-arch=gfx950 -O3 -o lds_valu lds_valu.hip && ./lds_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x4* lds_u32x4_ptr;

template <int R, int N>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (uint32_t i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((uint32_t*)smem)[i] = 0x3c003c00u;
  __syncthreads();
  uint32_t a = (threadIdx.x & 63) * 16u + (threadIdx.x >> 6) * 1024u;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const uint32_t c = 0x3c003c00u;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = *(lds_u32x4_ptr)(size_t)((a + r * 4096u) & 0xfff0u);
    a += 1024u;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int n = 0; n < N; ++n)
        acc[n & 3] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, v[r][n & 3]), __builtin_bit_cast(f16x2, c), acc[n & 3], false);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[threadIdx.x] = acc[0];
}

template <int R, int N>
void run(float* out, int waves) {
  hipFuncSetAttribute((const void*)k<R, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<R, N>), dim3(256), dim3(waves * 64), 81920, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<R, N>), dim3(256), dim3(waves * 64), 81920, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_it_clk = ms * 1e-3 / iters * 2.4e9;  // per iteration of one wave slot (all waves run concurrently)
  const int wps = waves / 4 > 0 ? waves / 4 : 1;                   // waves per SIMD
  const double valu = (double)wps * R * N * 4.0;                     // clk of VALU issue per SIMD and iteration (4 clk per wave64 instruction)
  printf("waves/SIMD %d  reads/iter %d  dots/read %2d : %7.1f clk per iteration;  VALU issue alone %6.0f;  the rest = %5.1f clk per ds_read_b128 and SIMD\n", wps, R, N,
         per_it_clk, valu, (per_it_clk - valu) / (wps * R));
}

int main() {
  float* out; hipMalloc(&out, 65536);
  for (int waves : {4, 8, 16}) {
    run<4, 1>(out, waves); run<4, 4>(out, waves); run<4, 8>(out, waves); run<4, 16>(out, waves); run<1, 16>(out, waves); run<8, 4>(out, waves);
  }
  return 0;
}
