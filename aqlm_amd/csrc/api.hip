// Host-side glue shared by every entry point of libaqlm_hip.so: error reporting, ABI version, tuning knobs.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#include "aqlm_common.h"

namespace aqlm {

static thread_local char g_err[512] = "";

void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_last_error("%s: %s (%d)", what, hipGetErrorString(e), (int)e);
  return (int)e;
}

int ensure_dynamic_lds(const void* kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return 0;
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> granted;
  int dev = 0;
  if (int e = check_hip(hipGetDevice(&dev), "hipGetDevice")) return e;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = granted[{kernel, dev}];
  if (have >= bytes) return 0;
  if (int e = check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                        "hipFuncSetAttribute(MaxDynamicSharedMemorySize)"))
    return e;
  have = bytes;
  return 0;
}

Tuning& tuning() {
  static Tuning t;
  return t;
}

static int* tuning_slot(const char* key) {
  Tuning& t = tuning();
  if (!key) return nullptr;
  if (!strcmp(key, "gemv_rows_per_wave")) return &t.gemv_rows_per_wave;
  if (!strcmp(key, "gemv1x16_aux")) return &t.gemv1x16_aux;
  if (!strcmp(key, "gemv1x16_prefetch_cb")) return &t.gemv1x16_prefetch_cb;
  if (!strcmp(key, "kx8_replicas")) return &t.kx8_replicas;
  if (!strcmp(key, "gemm_variant")) return &t.gemm_variant;
  if (!strcmp(key, "kx8_mfma_min_rows")) return &t.kx8_mfma_min_rows;
  if (!strcmp(key, "kx8_xres")) return &t.kx8_xres;
  if (!strcmp(key, "kx8_xres_phased")) return &t.kx8_xres_phased;
  if (!strcmp(key, "kx8_phase_tpb")) return &t.kx8_phase_tpb;
  if (!strcmp(key, "kx8_phase_quads")) return &t.kx8_phase_quads;
  if (!strcmp(key, "kx8_multi_xres_min_rows")) return &t.kx8_multi_xres_min_rows;
  if (!strcmp(key, "kx8_ksplit")) return &t.kx8_ksplit;
  if (!strcmp(key, "kx8_rt")) return &t.kx8_rt;
  if (!strcmp(key, "scan_max_rows")) return &t.scan_max_rows;
  if (!strcmp(key, "scan_prefetch")) return &t.scan_prefetch;
  if (!strcmp(key, "gemm_debug")) return &t.gemm_debug;
  if (!strcmp(key, "gemm_store_nt")) return &t.gemm_store_nt;
  if (!strcmp(key, "force_generic")) return &t.force_generic;
  if (!strcmp(key, "packed_waves")) return &t.packed_waves;
  if (!strcmp(key, "packed_fused_finalize")) return &t.packed_fused_finalize;
  if (!strcmp(key, "packed_prefetch")) return &t.packed_prefetch;
  if (!strcmp(key, "packed_arrange")) return &t.packed_arrange;
  if (!strcmp(key, "packed_xcopies")) return &t.packed_xcopies;
  if (!strcmp(key, "packed_entry_bytes")) return &t.packed_entry_bytes;
  if (!strcmp(key, "packed_debug")) return &t.packed_debug;
  if (!strcmp(key, "packed_fill_rotate")) return &t.packed_fill_rotate;
  if (!strcmp(key, "packed_pipe")) return &t.packed_pipe;
  if (!strcmp(key, "packed_prefetch_waves")) return &t.packed_prefetch_waves;
  if (!strcmp(key, "lut_waves")) return &t.lut_waves;
  return nullptr;
}

// position-sensitive checksum (include/aqlm_hip.h): per-thread partial sums, wave reduction through the atomics' return path is
// not needed -- 64-bit atomic adds commute, so the result does not depend on the order the waves arrive in
__global__ __launch_bounds__(256) void checksum_kernel(const uint8_t* data, size_t bytes, unsigned long long* out) {
  const size_t nwords = (bytes + 3) / 4;
  const bool aligned = (reinterpret_cast<uintptr_t>(data) & 3u) == 0;
  unsigned long long a = 0, b = 0;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) {
    uint32_t w = 0;
    if (aligned && i * 4 + 4 <= bytes) w = reinterpret_cast<const uint32_t*>(data)[i];
    else
      for (size_t k = 0; k < 4 && i * 4 + k < bytes; ++k) w |= (uint32_t)data[i * 4 + k] << (8 * k);
    a += w;
    b += (unsigned long long)w * (((unsigned long long)i * 0x9E3779B97F4A7C15ull) | 1ull);
  }
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, WAVE);
    b += __shfl_xor(b, o, WAVE);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], a);
    atomicAdd(&out[1], b);
  }
}

}  // namespace aqlm

extern "C" int aqlm_hip_checksum(const void* data, size_t bytes, void* out_u64x2, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!out_u64x2 || (!data && bytes) || (reinterpret_cast<uintptr_t>(out_u64x2) & 7u)) {
    aqlm::set_last_error("aqlm_hip_checksum: null pointer argument or misaligned output");
    return AQLM_HIP_E_INVALID;
  }
  if (int e = aqlm::check_hip(hipMemsetAsync(out_u64x2, 0, 16, stream), "checksum memset")) return e;
  if (bytes == 0) return 0;
  const size_t nwords = (bytes + 3) / 4;
  const unsigned blocks = (unsigned)std::min<size_t>(2048, (nwords + 255) / 256);
  hipLaunchKernelGGL(aqlm::checksum_kernel, dim3(blocks), dim3(256), 0, stream, (const uint8_t*)data, bytes, (unsigned long long*)out_u64x2);
  return aqlm::check_hip(hipGetLastError(), "checksum launch");
}

extern "C" int aqlm_hip_abi_version(void) { return AQLM_HIP_ABI_VERSION; }

extern "C" const char* aqlm_hip_last_error(void) { return aqlm::g_err; }

extern "C" int aqlm_hip_set_tuning(const char* key, int value) {
  int* s = aqlm::tuning_slot(key);
  if (!s) {
    aqlm::set_last_error("aqlm_hip_set_tuning: unknown key '%s'", key ? key : "(null)");
    return AQLM_HIP_E_INVALID;
  }
  *s = value;
  return 0;
}

extern "C" int aqlm_hip_get_tuning(const char* key, int* value) {
  int* s = aqlm::tuning_slot(key);
  if (!s || !value) {
    aqlm::set_last_error("aqlm_hip_get_tuning: unknown key '%s'", key ? key : "(null)");
    return AQLM_HIP_E_INVALID;
  }
  *value = *s;
  return 0;
}
