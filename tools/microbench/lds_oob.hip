// Micro-test: what does a ds_read_b128 beyond the workgroup's LDS allocation return on gfx950, and what does it cost?
//   hipcc --offload-arch=gfx950 -O3 -o lds_oob lds_oob.hip && ./lds_oob
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4* lds_u32x4_ptr;

__global__ void probe(uint32_t* out, uint32_t lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (uint32_t i = threadIdx.x; i < lds_bytes / 4; i += blockDim.x) ((uint32_t*)smem)[i] = 0xA5000000u + i;
  __syncthreads();
  const uint32_t addrs[6] = {0u, lds_bytes - 16u, lds_bytes, lds_bytes + 4096u, 163840u, 1048560u};
  for (int k = 0; k < 6; ++k) {
    const u32x4 v = *(lds_u32x4_ptr)(size_t)(addrs[k]);
    if (threadIdx.x == 0) { out[k * 4 + 0] = v.x; out[k * 4 + 1] = v.y; out[k * 4 + 2] = v.z; out[k * 4 + 3] = v.w; }
  }
}

// mode 0: every lane reads a random in-range entry; 1: 1/8 of the lanes random in range, the rest one in-range zero entry (same address);
// 2: 1/8 in range, the rest out of range (distinct addresses); 3: 1/8 in range, the rest exec-masked off
__global__ __launch_bounds__(512) void timing(uint32_t* out, const uint32_t* idx, int iters, int mode, uint32_t lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (uint32_t i = threadIdx.x; i < lds_bytes / 4; i += blockDim.x) ((uint32_t*)smem)[i] = i;
  __syncthreads();
  uint32_t a[16];
  for (int j = 0; j < 16; ++j) {
    uint32_t r = idx[(threadIdx.x * 16 + j) & 65535];
    const bool in = mode == 0 || (r & 7u) == 0u;
    uint32_t e = (r >> 3) & 8191u;
    if (!in) e = mode == 1 ? 8192u : (mode == 2 ? 16384u + (r >> 3) % 40000u : e);
    a[j] = e * 16u | (in ? 0u : 0x80000000u);
  }
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t ad = a[j] & 0x7fffffffu;
      u32x4 v = {0, 0, 0, 0};
      if (mode == 3) { if (!(a[j] >> 31)) v = *(lds_u32x4_ptr)(size_t)ad; }
      else v = *(lds_u32x4_ptr)(size_t)ad;
      acc += v;
    }
    asm volatile("" ::: "memory");
  }
  if (acc.x == 0x12345u) out[threadIdx.x] = acc.y + acc.z + acc.w;
}

int main() {
  uint32_t* out; hipMalloc(&out, 4096 * 4);
  uint32_t* idx; hipMalloc(&idx, 65536 * 4);
  uint32_t* h = (uint32_t*)malloc(65536 * 4);
  for (int i = 0; i < 65536; ++i) h[i] = (uint32_t)rand() * 2654435761u >> 4;
  hipMemcpy(idx, h, 65536 * 4, hipMemcpyHostToDevice);
  for (uint32_t lds : {16384u, 148480u}) {
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipMemset(out, 0xff, 4096);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), lds, 0, out, lds);
    uint32_t r[24]; hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
    hipError_t e = hipDeviceSynchronize();
    printf("LDS allocation %u B (%s): reads at 0 / last / end / end+4096 / 163840 / 1048560:\n", lds, hipGetErrorString(e));
    for (int k = 0; k < 6; ++k) printf("   %08x %08x %08x %08x\n", r[k * 4], r[k * 4 + 1], r[k * 4 + 2], r[k * 4 + 3]);
  }
  hipFuncSetAttribute((const void*)timing, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  const uint32_t lds = 131072 + 1024;
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(timing, dim3(256), dim3(512), lds, 0, out, idx, 10, mode, lds);
    hipEventRecord(e0);
    hipLaunchKernelGGL(timing, dim3(256), dim3(512), lds, 0, out, idx, iters, mode, lds);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double reads_per_cu = (double)iters * 16 * 8;  // wave-instructions per CU
    printf("mode %d: %.3f ms, %.2f ns per wave-instruction per CU (= %.1f clk at 2.4 GHz); err %s\n", mode, ms, ms * 1e6 / reads_per_cu, ms * 1e6 / reads_per_cu * 2.4,
           hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
