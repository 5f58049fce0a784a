// 8 codebooks x 8 bits, 32-element vectors (8x8 g32, 2 bits per weight) at 2 .. 64 batch rows on gfx950 (round 5): Y = X W^T with
// W never materialised.
//
// Replaces (behaviour, not code): what the reference runs for this scheme beyond one row -- its Triton kernel in a per-row loop
// (triton_kernel.py:161-182) up to 6 rows and dequantize_gemm = _dequantize_weight + F.linear (dequantization.py:9-21,
// utils.py:43-70) above.  Here, until round 5: 2..8 rows = the look-up-table matvec once per row (4.5 us per extra row of a
// 4096 x 4096 layer), more rows = dequant kernel + hipBLASLt (36-41 us).
//
// The eight codebooks (8 x 256 x 64 B = 128 KiB) live in LDS for the life of a workgroup, re-laid out in 16-byte piece planes
// [codebook][piece of 8 elements][entry] so that the 16 lanes of an LDS service group that gather the same piece of 16 random
// entries spread over the 16 bank groups like 16 random draws (the checkpoint layout [entry][4 pieces] would put them on 4).
// A group of 32 input features is exactly one k-step of v_mfma_f32_16x16x32: lane (row r, piece p) of the A fragment gathers
// piece p of codebook c's entry codes[r][j][c] -- one ds_read_b128 per codebook -- and the eight codebooks' terms of a weight
// meet in the fp32 accumulator (8 MFMAs per k-step and batch tile: the matrix cores have room, and W is never rounded: exact
// fp16 / bf16 products, fp32 sums, like every other kernel here).
//
// A workgroup owns 16-row output tiles (tile = blockIdx.x, += gridDim.x) over ALL of K; its 8 waves split K in chunks of 8 groups
// (chunk = wave, += 8), so nothing is synchronised inside a tile.  Codes and X reach the lanes in ROW-MAJOR order (lane = 4 x row
// + 16-byte piece: 64 contiguous bytes per row, 16 requests per load instead of 64) and are turned into fragment order by
// ds_bpermute_b32 (no LDS memory, 2 LDS cycles each: 6 per k-step next to 8 gathers of 4-12 cycles); both are requested ahead
// in registers (codes: one chunk = 8 k-steps ahead, across tile boundaries; X: 4 k-steps).  The eight K shares of a tile meet in
// LDS in wave order (deterministic, independent of the batch), then scale + bias + one rounding.
//
// Cost model (4096 x 4096, <= 16 rows): 1 MiB of LDS gathers per tile and workgroup = 4096 LDS cycles conflict-free, ~3x that
// with 16 random entries per service group; fill of the codebooks 128 KiB per workgroup through its L1 (~2 us); X re-read per
// tile from L2 (B x 8 KiB).  Independent of the number of rows up to 16; 8 MFMAs more per k-step for every further 16 rows.
#include <algorithm>
#include <type_traits>

#include "aqlm_common.h"

namespace aqlm {

namespace {

typedef _Float16 e8_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 e8_bf16x8 __attribute__((ext_vector_type(8)));
typedef float e8_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4* e8_lds_u32x4_ptr;

template <class T>
__device__ __forceinline__ e8_f32x4 e8_mfma(const u32x4& a, const u32x4& b, const e8_f32x4& c);
template <>
__device__ __forceinline__ e8_f32x4 e8_mfma<F16>(const u32x4& a, const u32x4& b, const e8_f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(e8_f16x8, a), __builtin_bit_cast(e8_f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ e8_f32x4 e8_mfma<BF16>(const u32x4& a, const u32x4& b, const e8_f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(e8_bf16x8, a), __builtin_bit_cast(e8_bf16x8, b), c, 0, 0, 0);
}

constexpr int E8_NW = 8;    // waves per workgroup (K shares of a tile)
constexpr int E8_XD = 4;    // k-steps X is requested ahead (divides 8)
constexpr uint32_t E8_CB_BYTES = 8u * 4u * 256u * 16u;  // [codebook][piece][entry][16 B]

struct E8Params {
  const uint8_t* codes;      // [M][in_groups][8] u8
  const uint8_t* codebooks;  // [8][256][32] halfs
  const uint16_t* X;         // [B][xs]
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* Y;
  long xs, ys;
  int M, B, in_groups, ntiles, nchunks;  // nchunks = in_groups / 8 >= E8_NW
};

__device__ __forceinline__ u32x4 e8_bperm(int addr, const u32x4& v) {
  u32x4 r;
  r.x = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.x);
  r.y = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.y);
  r.z = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.z);
  r.w = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.w);
  return r;
}

template <class T, int NBT>  // NBT = 16-column batch tiles (B <= 16 NBT)
__global__ __launch_bounds__(E8_NW * 64) void gemm_8x8g32_rows16_kernel(const E8Params p) {
  constexpr int NW = E8_NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char e8_smem[];
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)e8_smem != 0u) __builtin_trap();  // the LDS map starts at 0
  constexpr uint32_t RED = E8_CB_BYTES;  // [NW][NBT][64 lanes][16 B] fp32 partial tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int arow = lane & 15, kg = lane >> 4;  // fragment order: A row / batch column, 8-element piece of the k-step
  const int rrow = lane >> 2, rq = lane & 3;   // row-major order of the loads: row, 16-byte piece
  const int grid = (int)gridDim.x;

  // ---- per-lane load bases (row-major) ------------------------------------------------------------------------------------------
  const uint8_t* xbase[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) {
    int b = bt * 16 + rrow;
    b = b < p.B ? b : p.B - 1;
    xbase[bt] = (const uint8_t*)(p.X + (size_t)b * p.xs) + rq * 16;
  }
  auto load_x = [&](int chunk, int jj, u32x4 (&dst)[NBT]) {
    const size_t off = ((size_t)chunk * 8 + (size_t)jj) * 64;  // 32 halfs per group
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) dst[bt] = *reinterpret_cast<const u32x4*>(xbase[bt] + off);
  };
  auto load_codes = [&](int tile, int chunk) -> u32x4 {
    int r = tile * 16 + rrow;
    r = r < p.M ? r : p.M - 1;
    return *reinterpret_cast<const u32x4*>(p.codes + ((size_t)r * p.in_groups + (size_t)chunk * 8) * 8 + rq * 16);
  };
  const int xsrc = 4 * (4 * arow + kg);  // bpermute address: fragment lane (arow, kg) <- row-major lane 4 arow + kg
  const int csrc = 16 * arow;            // + 4 (jj >> 1): the row-major lane that holds groups 2 (jj >> 1), + 1 of row arow

  int tile = (int)blockIdx.x, chunk = wave;
  u32x4 cnext{};
  u32x4 xr[E8_XD][NBT];
  if (tile < p.ntiles) cnext = load_codes(tile, chunk);
#pragma unroll
  for (int d = 0; d < E8_XD; ++d) load_x(chunk, d, xr[d]);

  // ---- the codebooks -> LDS in piece planes (registers: a DMA of 16-byte pieces at a 64-byte stride would cost the addresser 32
  // lines per instruction for a quarter of their bytes)
#pragma unroll
  for (int it = 0; it < 2048 / (NW * 64); ++it) {
    const int e = it * (NW * 64) + tid;  // c * 256 + v
    const u32x4* src = reinterpret_cast<const u32x4*>(p.codebooks + (size_t)e * 64);
    const u32x4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
    const uint32_t base = (uint32_t)(e >> 8) * 16384u + (uint32_t)(e & 255) * 16u;
    *reinterpret_cast<u32x4*>(e8_smem + base) = v0;
    *reinterpret_cast<u32x4*>(e8_smem + base + 4096u) = v1;
    *reinterpret_cast<u32x4*>(e8_smem + base + 8192u) = v2;
    *reinterpret_cast<u32x4*>(e8_smem + base + 12288u) = v3;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // the codebooks are in LDS, for good

  const uint32_t gbase = (uint32_t)kg * 4096u;  // this lane's piece plane
  e8_f32x4 acc[NBT][2];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) acc[bt][0] = acc[bt][1] = e8_f32x4{0.f, 0.f, 0.f, 0.f};

  while (tile < p.ntiles) {
    const u32x4 ccur = cnext;
    int nchunk = chunk + NW, ntile = tile;
    bool last = false;
    if (nchunk >= p.nchunks) {
      nchunk = wave;
      ntile = tile + grid;
      last = true;
    }
    if (ntile < p.ntiles) cnext = load_codes(ntile, nchunk);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      // ---- this k-step's X: out of the ring, into fragment order; the slot is refilled E8_XD k-steps ahead
      u32x4 xf[NBT];
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) xf[bt] = e8_bperm(xsrc, xr[jj % E8_XD][bt]);
      {
        const int pj = jj + E8_XD;
        if (pj < 8) load_x(chunk, pj, xr[jj % E8_XD]);
        else load_x(nchunk, pj - 8, xr[jj % E8_XD]);
      }
      // ---- this k-step's code word (8 bytes: one per codebook) of the lane's row
      const int ca = csrc + 4 * (jj >> 1);
      const uint32_t c0 = (uint32_t)__builtin_amdgcn_ds_bpermute(ca, (int)((jj & 1) ? ccur.z : ccur.x));
      const uint32_t c1 = (uint32_t)__builtin_amdgcn_ds_bpermute(ca, (int)((jj & 1) ? ccur.w : ccur.y));
      // ---- 8 gathers = 8 A fragments; 8 NBT MFMAs on two accumulators per batch tile
      u32x4 w[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t cw = c < 4 ? c0 : c1;
        const uint32_t t1 = ((c & 3) == 0 ? cw << 4 : cw >> (8 * (c & 3) - 4)) & 0xff0u;  // code byte x 16
        w[c] = *(e8_lds_u32x4_ptr)(size_t)((uint32_t)c * 16384u + gbase + t1);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) acc[bt][c & 1] = e8_mfma<T>(w[c], xf[bt], acc[bt][c & 1]);
    }
    if (last) {
      // ---- the eight K shares of the tile meet in LDS, in wave order; waves 0 .. NBT - 1 finish one batch tile each
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) {
        *reinterpret_cast<e8_f32x4*>(e8_smem + RED + (uint32_t)((wave * NBT + bt) * 1024) + (uint32_t)lane * 16u) = acc[bt][0] + acc[bt][1];
        acc[bt][0] = acc[bt][1] = e8_f32x4{0.f, 0.f, 0.f, 0.f};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (wave < NBT) {
        const int bt = wave;
        e8_f32x4 v = *reinterpret_cast<const e8_f32x4*>(e8_smem + RED + (uint32_t)(bt * 1024) + (uint32_t)lane * 16u);
#pragma unroll
        for (int w8 = 1; w8 < NW; ++w8)
          v = v + *reinterpret_cast<const e8_f32x4*>(e8_smem + RED + (uint32_t)((w8 * NBT + bt) * 1024) + (uint32_t)lane * 16u);
        const int m = tile * 16 + kg * 4;  // lane (arow, kg) holds output rows 4 kg .. 4 kg + 3 of batch column arow
        const int b = bt * 16 + arow;
        if (b < p.B && m < p.M) {
          uint16_t* dst = p.Y + (size_t)b * p.ys + m;
          uint16_t h[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int mm = m + r < p.M ? m + r : p.M - 1;
            const float sc = T::to_float(p.scales[mm]), bi = p.bias ? T::to_float(p.bias[mm]) : 0.f;
            h[r] = T::from_float(__builtin_fmaf(v[r], sc, bi));
          }
          if ((p.M & 3) == 0 && (p.ys & 3) == 0 && ((uintptr_t)dst & 7u) == 0) *reinterpret_cast<u32x2*>(dst) = u32x2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
          else
            for (int r = 0; r < 4; ++r)
              if (m + r < p.M) dst[r] = h[r];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // the partial tiles are free again
    }
    tile = ntile;
    chunk = nchunk;
  }
}

template <class T, int NBT>
int launch_e8(const E8Params& p, hipStream_t stream) {
  auto kern = gemm_8x8g32_rows16_kernel<T, NBT>;
  const size_t lds = (size_t)E8_CB_BYTES + (size_t)E8_NW * NBT * 1024;
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  static const int cus = [] {  // (initialised once, thread-safe; every GPU of a node is the same part)
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  // one workgroup per CU (the codebooks fill its LDS); tiles are dealt round-robin, so the last round is as even as it gets
  const int rounds = (p.ntiles + cus - 1) / cus;
  const int grid = (p.ntiles + rounds - 1) / rounds;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(E8_NW * 64), lds, stream, p);
  return check_hip(hipGetLastError(), "gemm_8x8g32_rows16 launch");
}

}  // namespace

}  // namespace aqlm

using namespace aqlm;

extern "C" int aqlm_hip_gemm_8x8_mfma(const void* codes, const void* codebooks, const void* scales, const void* bias, const void* X,
                                      void* Y, int batch, int out_features, int in_features, int in_group_size, long xs, long ys,
                                      int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!codes || !codebooks || !scales || !X || !Y) {
    set_last_error("aqlm_hip_gemm_8x8_mfma: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (batch <= 0 || out_features <= 0 || in_features <= 0) {
    set_last_error("aqlm_hip_gemm_8x8_mfma: sizes must be positive");
    return AQLM_HIP_E_INVALID;
  }
  if (in_group_size != 32) {
    set_last_error("aqlm_hip_gemm_8x8_mfma: 8 codebooks of 256 x 32 only, got group %d", in_group_size);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_gemm_8x8_mfma: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (in_features % 256 != 0 || in_features < 256 * E8_NW || !aligned16(codebooks) || !aligned16(X) || !aligned16(codes) || xs % 8 != 0) {
    set_last_error("aqlm_hip_gemm_8x8_mfma: needs in_features %% 256 == 0, >= %d, and 16-B aligned codes / codebooks / X rows", 256 * E8_NW);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  for (int b0 = 0; b0 < batch; b0 += 64) {  // slabs of 64 rows (the codes are re-read per slab: they are 2 bits per weight)
    const int nb = std::min(64, batch - b0);
    E8Params p{};
    p.codes = (const uint8_t*)codes;
    p.codebooks = (const uint8_t*)codebooks;
    p.X = (const uint16_t*)X + (long)b0 * xs;
    p.scales = (const uint16_t*)scales;
    p.bias = (const uint16_t*)bias;
    p.Y = (uint16_t*)Y + (long)b0 * ys;
    p.xs = xs;
    p.ys = ys;
    p.M = out_features;
    p.B = nb;
    p.in_groups = in_features / 32;
    p.ntiles = (out_features + 15) / 16;
    p.nchunks = p.in_groups / 8;
    const int nbt = nb <= 16 ? 1 : (nb <= 32 ? 2 : 4);
    int e;
    if (dtype == AQLM_HIP_F16) e = nbt == 1 ? launch_e8<F16, 1>(p, stream) : (nbt == 2 ? launch_e8<F16, 2>(p, stream) : launch_e8<F16, 4>(p, stream));
    else e = nbt == 1 ? launch_e8<BF16, 1>(p, stream) : (nbt == 2 ? launch_e8<BF16, 2>(p, stream) : launch_e8<BF16, 4>(p, stream));
    if (e) return e;
  }
  return 0;
}
