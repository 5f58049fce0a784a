// One-shot all-reduce over xGMI fused into the finalize of the prepacked 1x16 matvec (row-parallel / "in"-split layers).
//
// No reference counterpart (the reference has no tensor parallelism); north star: the 70B layer
// 8192 -> 28672 split along `in` over 8 MI355X.  Each rank's packed kernel leaves fp32 slice partials [16][B][M]; the
// single-GPU finalize adds the 16 slices, scales, adds the bias and rounds.  Here the slice sums of the R ranks have to
// be added as well.  A library all-reduce of 56-112 KiB is pure latency (one more launch + a ring of 2 (R - 1) hops);
// instead every rank PUBLISHES its slice-summed fp32 vector in a buffer the peers have mapped (IPC), raises a flag,
// and the finalize of every rank reads all R vectors straight over xGMI (point-to-point links: 7 concurrent reads of
// 112 KiB) and adds them IN RANK ORDER -- so every rank rounds the very same fp32 sum: the replicas of y are bit-identical.
//
// Protocol (cdna_hip_programming.md Guideline 16, recipe R1, with SYSTEM scope instead of agent scope because the
// consumer is another GPU):
//   state per rank (device memory, mapped by every peer): pub[2][B * M] fp32, flag[2] u32; private: epoch u32.
//   publish kernel  e = epoch; every thread stores its sums to pub[e & 1] with system-scope (write-through) stores; every
//                   wave drains its stores (vmcnt(0)); the LAST block to finish (device-scope ticket) stores flag[e & 1] = e.
//   reduce kernel   every block: one lane polls the R flags (relaxed system-scope loads, bounded, s_sleep) until all == e;
//                   one system-scope acquire; then system-scope loads of the R vectors, sum in rank order, epilogue.
//                   The last block to finish bumps `epoch` (and resets the ticket) for the next call.
//   pub / flag are double-buffered by the parity of the epoch: a rank that races ahead into call e + 1 writes the other
//   half; it cannot reach call e + 2 before every peer has finished READING call e (its own call e + 1 waits for their
//   flags e + 1, which they raise only after their reduce of call e).  The epoch lives in device memory and is advanced by
//   the kernels themselves, so a captured hipGraph replays correctly (a kernel argument would be frozen).
//   Every spin is bounded: on a time-out the status word is set and y is filled with NaN -- a lost peer shows up as NaN
//   and an error code, never as a hang.
#include "aqlm_common.h"

namespace aqlm {

constexpr int XG_SLICES = 16;  // slice partials of the packed kernel (PK_S)

typedef __attribute__((address_space(1))) uint32_t* gu32_ptr;
typedef __attribute__((address_space(1))) float* gf32_ptr;

struct XgmiParams {
  const float* partial;          // [16][B][M] slice partials of this rank
  float* const* peer_pub;        // [world] device pointers: peer r's pub base (2 x max_elems floats)
  uint32_t* const* peer_flag;    // [world] device pointers: peer r's flag[2]
  uint32_t* epoch;               // this rank: [0] epoch (starts at 1), [1] ticket of the publish kernel, [2] ticket of the reduce kernel
  uint32_t* status;              // this rank: set to 1 on a time-out
  const uint16_t* scales;
  const uint16_t* bias;          // applied by every rank alike (each holds the full y)
  uint16_t* y;
  long y_row_stride;
  int M, B, rank, world;
  uint32_t max_elems;
  uint32_t spin_limit;
};

__device__ __forceinline__ void sys_store_f32(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float sys_load_f32(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t sys_load_u32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void xgmi_publish_kernel(const XgmiParams p) {
  const uint32_t e = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int n = p.M * p.B;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float* pub = p.peer_pub[p.rank] + (size_t)(e & 1u) * p.max_elems;
  if (i < n) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < XG_SLICES; ++k) s += p.partial[(size_t)k * n + i];
    sys_store_f32(pub + i, s);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t done = __hip_atomic_fetch_add(p.epoch + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done + 1 == gridDim.x) {  // all blocks of this rank have drained: raise the flag (system scope)
      __hip_atomic_store(p.epoch + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(p.peer_flag[p.rank] + (e & 1u), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <class T_>
__global__ __launch_bounds__(256) void xgmi_reduce_kernel(const XgmiParams p) {
  __shared__ int ok_s;
  const uint32_t e = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) {
    int ok = 1;
    for (int r = 0; r < p.world && ok; ++r) {
      const uint32_t* f = p.peer_flag[r] + (e & 1u);
      uint32_t spins = 0;
      while (sys_load_u32(f) != e) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > p.spin_limit) { ok = 0; break; }
      }
    }
    if (!ok) __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // ONE system-scope acquire after the polls
    ok_s = ok;
  }
  __syncthreads();
  const int n = p.M * p.B;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int b = i / p.M, m = i - b * p.M;
    float s = 0.f;
    if (ok_s) {
      for (int r = 0; r < p.world; ++r) s += sys_load_f32(p.peer_pub[r] + (size_t)(e & 1u) * p.max_elems + i);  // rank order
      const float scale = T_::to_float(p.scales[m]);
      const float bias = p.bias ? T_::to_float(p.bias[m]) : 0.f;
      p.y[(size_t)b * p.y_row_stride + m] = T_::from_float(__builtin_fmaf(s, scale, bias));
    } else {
      p.y[(size_t)b * p.y_row_stride + m] = (uint16_t)0x7fffu;  // NaN in fp16 and bf16: a lost peer must not pass silently
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t done = __hip_atomic_fetch_add(p.epoch + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done + 1 == gridDim.x) {  // every block of this rank has read what it needs: next call, next epoch
      __hip_atomic_store(p.epoch + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.epoch, e + 1u == 0u ? 1u : e + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace aqlm

using namespace aqlm;

extern "C" size_t aqlm_hip_xgmi_state_bytes(int max_elems) {
  // [pub 2 x max_elems floats][flag 2 u32, padded to 64 B][epoch / tickets 4 u32][status u32 ...] -- see aqlm_hip.h
  return max_elems > 0 ? (size_t)2 * max_elems * 4 + 256 : 0;
}

extern "C" int aqlm_hip_xgmi_finalize(const aqlm_hip_xgmi* xg, const void* partial, const void* scales, const void* bias,
                                      void* y, int out_features, int batch, long y_row_stride, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!xg || !scales || !y || !xg->peer_pub || !xg->peer_flag || !xg->epoch || !xg->status) {
    set_last_error("aqlm_hip_xgmi_finalize: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_xgmi_finalize: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (xg->world < 1 || xg->rank < 0 || xg->rank >= xg->world || out_features < 1 || batch < 1 ||
      (size_t)out_features * batch > (size_t)xg->max_elems) {
    set_last_error("aqlm_hip_xgmi_finalize: rank %d of %d, %d x %d elements, state sized for %d", xg->rank, xg->world, batch,
                   out_features, xg->max_elems);
    return AQLM_HIP_E_INVALID;
  }
  XgmiParams p{};
  p.partial = (const float*)partial;
  p.peer_pub = (float* const*)xg->peer_pub;
  p.peer_flag = (uint32_t* const*)xg->peer_flag;
  p.epoch = (uint32_t*)xg->epoch;
  p.status = (uint32_t*)xg->status;
  p.scales = (const uint16_t*)scales;
  p.bias = (const uint16_t*)bias;
  p.y = (uint16_t*)y;
  p.y_row_stride = y_row_stride;
  p.M = out_features;
  p.B = batch;
  p.rank = xg->rank;
  p.world = xg->world;
  p.max_elems = (uint32_t)xg->max_elems;
  p.spin_limit = xg->spin_limit ? xg->spin_limit : (1u << 22);  // x ~0.5 us per poll: seconds, not forever
  const int blocks = (out_features * batch + 255) / 256;
  if (partial) hipLaunchKernelGGL(xgmi_publish_kernel, dim3(blocks), dim3(256), 0, stream, p);  // null: the matvec kernel has published
  if (dtype == AQLM_HIP_F16) hipLaunchKernelGGL(xgmi_reduce_kernel<F16>, dim3(blocks), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(xgmi_reduce_kernel<BF16>, dim3(blocks), dim3(256), 0, stream, p);
  return check_hip(hipGetLastError(), "xgmi finalize launch");
}
