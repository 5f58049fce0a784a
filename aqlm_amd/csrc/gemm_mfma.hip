// Fused dequant-tile -> MFMA GEMM for the 1x16 scheme at large batch (prefill / bs=128) on gfx950.
//
// Replaces (behaviour, not code): code1x16_matmat_dequant = Code1x16Dequant (W[out,in] materialised in HBM,
// reference cuda_kernel.cu:98-142) + F::linear / cuBLAS (cuda_kernel.cpp:249-301) + scale/bias epilogue launches.
//
// Key observation: with v_mfma_f32_32x32x16_{f16,bf16} the A operand of lane l is 8 consecutive k of row l%32
// (k-half l/32) -- which is exactly one 16-byte AQLM codebook vector (g=8), or one half of one (g=16).  So a
// gathered codebook entry IS an MFMA fragment: W is never written anywhere, not even to LDS.  Per 16-deep k step
// a wave issues ONE 16-B gather per lane (32 rows x 2 k-halves) and reuses it for every 32-column batch tile.
//
//   C[row = W row][col = batch]  +=  A = W[32 rows][16 k]  x  B = X^T[16 k][32 batch]
//
// Block = 4 waves = 128 output rows (one 32-row tile per wave, waves independent) x one K slice x all batch
// columns (<= 128, as NBT tiles of 32).  X is staged in 64-deep chunks through a double-buffered, XOR-swizzled
// LDS image shared by the 4 waves.  K is split over `ksplit` blocks so that the grid fills 256 CUs; fp32
// partials go to a workspace [ksplit][out][Bpad] and a second small kernel sums them, applies
// scales/bias, transposes to Y[B][out] and rounds once.
//
// Roofline: MFMA-bound in principle (2*B*out*in flop), but for out=in=4096, B=128 the 2.1 M random 16-B gathers
// (~1 lane/clk/CU) take ~2x the MFMA time, so this kernel is L2-gather bound like the gemv.
#include <algorithm>

#include <type_traits>

#include "aqlm_common.h"

namespace aqlm {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <class T>
__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c);
template <>
__device__ __forceinline__ f32x16 mfma32<F16>(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma32<BF16>(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

struct GemmParams {
  const uint8_t* codes;
  const uint8_t* codebook;
  const uint16_t* X;
  float* partial;  // [ksplit][M][Bpad]
  int M, K, B, Bpad;
  int in_groups;
  int kslice;  // k elements per block (multiple of 64)
  int ksplit;
  long xs;
  int cb_bytes;
  uint32_t x_bytes;  // extent of the X slab (rows >= B read as zeros through the bounds-checked descriptor)
};

constexpr int BK = 64;  // k depth of one LDS chunk of X

// LDS image of one X chunk: row b (batch) holds 64 k = 8 pieces of 16 B; piece c is stored at slot c ^ ((b>>1)&7) so
// that the 16 lanes of a ds_read_b128 service group (16 distinct b mod 16) hit 16 distinct 16-B slots.
__device__ __forceinline__ int xswz(int b, int c) { return b * 8 + (c ^ ((b >> 1) & 7)); }

template <class T, int G, int NBT>
__global__ __launch_bounds__(256) void gemm_1x16_mfma_kernel(const GemmParams p) {
  constexpr int NROWS_X = NBT * 32;
  constexpr int PIECES = NROWS_X * 8;          // 16-B pieces per chunk
  constexpr int PER_THREAD = (PIECES + 255) / 256;
  constexpr int CODES_PER_CHUNK = BK / G;      // 8 (g8) or 4 (g16)
  constexpr int CWN = CODES_PER_CHUNK / 2;     // dwords of codes per chunk per row
  __shared__ __attribute__((aligned(16))) u32x4 xl[2][PIECES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5;
  const int row_blk = blockIdx.x / p.ksplit;
  const int ks = blockIdx.x - row_blk * p.ksplit;
  const int row_tile0 = row_blk * 128 + wave * 32;
  int my_row = row_tile0 + (lane & 31);
  const bool row_ok = my_row < p.M;
  if (!row_ok) my_row = p.M - 1;  // clamp: computed but never stored
  const int k_begin = ks * p.kslice;
  const int k_end = k_begin + p.kslice < p.K ? k_begin + p.kslice : p.K;
  const int nchunks = (k_end - k_begin + BK - 1) / BK;

  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.codebook, 0, p.cb_bytes, 0x00020000);
  const uint8_t* code_row = p.codes + (long)my_row * p.in_groups * 2;

  f32x16 acc[NBT];
#pragma unroll
  for (int t = 0; t < NBT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // global -> register staging of one X chunk.  The loads are UNCONDITIONAL buffer loads (rows >= B and k >= k_end
  // read as zeros through the bounds-checked descriptor / an out-of-range offset): with the load under a branch hipcc
  // emitted vmcnt(0) right behind it, i.e. every chunk step first waited ~1 us for X and only then issued the
  // codebook gathers of the next chunk (traced through the ISA: one L2 latency + one gather latency per 64-deep chunk).
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
  static_assert(PIECES % 256 == 0, "the X chunk splits evenly over the block");
  u32x4 xr[2][PER_THREAD];  // X chunks run two ahead in registers, one ahead in LDS
  uint32_t xoff[PER_THREAD];
#pragma unroll
  for (int s = 0; s < PER_THREAD; ++s) {
    const int q = tid + s * 256;
    xoff[s] = (uint32_t)(((long)(q >> 3) * p.xs + k_begin + (q & 7) * 8) * 2);
  }
  auto load_x = [&](int chunk, u32x4 (&dst)[PER_THREAD]) {
#pragma unroll
    for (int s = 0; s < PER_THREAD; ++s) {
      const bool in_k = k_begin + chunk * BK + (int)((tid + s * 256) & 7) * 8 < k_end;
      dst[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, in_k ? xoff[s] + (uint32_t)chunk * (BK * 2) : 0xfffffff0u, 0, 0);
    }
  };
  auto store_x = [&](int buf, const u32x4 (&src)[PER_THREAD]) {
#pragma unroll
    for (int s = 0; s < PER_THREAD; ++s) {
      const int q = tid + s * 256;
      xl[buf][xswz(q >> 3, q & 7)] = src[s];
    }
  };
  const int last_chunk = nchunks - 1;
  auto load_codes = [&](int chunk, uint32_t (&cw)[CWN]) {  // unconditional: chunks past the slice re-read the last one
    chunk = chunk < last_chunk ? chunk : last_chunk;
    const int k0 = k_begin + chunk * BK;
    const uint8_t* src = code_row + (long)(k0 / G) * 2;
    if constexpr (CWN == 4) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src);
      cw[0] = v.x; cw[1] = v.y; cw[2] = v.z; cw[3] = v.w;
    } else {
      const u32x2 v = *reinterpret_cast<const u32x2*>(src);
      cw[0] = v.x; cw[1] = v.y;
    }
  };

  // gathers: one 16-B entry per lane per 16-deep k step = 4 per 64-deep chunk; entry == MFMA A fragment
  auto gather = [&](const uint32_t (&cw)[CWN], u32x4 (&af)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint32_t code, piece;
      if constexpr (G == 8) {
        code = (cw[kk] >> (16 * half)) & 0xffffu;  // code index 2*kk + half
        piece = 0;
      } else {
        code = (cw[kk >> 1] >> (16 * (kk & 1))) & 0xffffu;  // code index kk, lane-half picks the 16-B half
        piece = half;
      }
      af[kk] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, code * (uint32_t)(G * 2) + piece * 16, 0, 0);
    }
  };

  // Software pipeline, in batches of NBATCH chunks.  Inside a batch the code is straight-line (fully unrolled, static
  // ring slots): all code words of the batch are requested up front, codebook gathers run GD-1 = 3 chunks ahead of
  // the MFMAs that consume them (one 64-deep chunk of MFMAs is ~0.3 us, an L2 gather ~1.5 us under load: a distance of
  // one chunk left every step waiting for its own gathers), X runs two chunks ahead in registers and one in LDS, and
  // NOTHING is in flight across the loop back-edge: hipcc's wait-count pass merges the prologue and back-edge states at
  // a loop header and then waits with vmcnt(0) for every register loaded in the previous iteration.  A batch pays one
  // pipeline refill instead (none at all for K slices of <= NBATCH chunks, e.g. 4096 x 4096 with the 8-way K split).
  constexpr int NBATCH = 8, GD = 4;
  uint32_t cw[NBATCH][CWN];
  u32x4 af[GD][4];
  auto mfmas = [&](int buf, const u32x4 (&af_cur)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int t = 0; t < NBT; ++t) {
        const u32x4 bfrag = xl[buf][xswz(t * 32 + (lane & 31), kk * 2 + half)];
        acc[t] = mfma32<T>(af_cur[kk], bfrag, acc[t]);
      }
    }
  };
  for (int base = 0; base < nchunks; base += NBATCH) {
#pragma unroll
    for (int c = 0; c < NBATCH; ++c) load_codes(base + c, cw[c]);
    load_x(base, xr[0]);
    load_x(base + 1, xr[1]);
#pragma unroll
    for (int c = 0; c < GD - 1; ++c) gather(cw[c], af[c]);
    store_x(0, xr[0]);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NBATCH; ++s) {
      const int buf = s & 1;  // NBATCH is even: the first chunk of every batch uses buffer 0
      if (s + GD - 1 < NBATCH) gather(cw[s + GD - 1], af[(s + GD - 1) % GD]);   // compile-time conditions only
      if (s + 2 < NBATCH) load_x(base + s + 2, xr[s % 2]);                       // slot of chunk s: already in LDS
      mfmas(buf, af[s % GD]);
      if (s + 1 < NBATCH) store_x(buf ^ 1, xr[(s + 1) % 2]);
      __syncthreads();
    }
  }

  // fp32 partials: C layout col = lane&31 (batch), row = (r&3) + 8*(r>>2) + 4*half
  float* out = p.partial + (long)ks * p.M * p.Bpad;
#pragma unroll
  for (int t = 0; t < NBT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_tile0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row < p.M) out[(long)row * p.Bpad + t * 32 + (lane & 31)] = acc[t][r];
    }
}

// Y[b][m] = (sum_s partial[s][m][b]) * scales[m] + bias[m]; 32x32 tile transpose through LDS.
struct FinalizeParams {
  const float* partial;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* Y;
  int M, B, Bpad, ksplit;
  long ys;
};

template <class T>
__global__ __launch_bounds__(256) void gemm_finalize_kernel(const FinalizeParams p) {
  __shared__ float tile[32][33];
  const int m0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + i * 8, b = b0 + tx;
    float s = 0.f;
    if (m < p.M && b < p.Bpad) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = k < p.ksplit ? p.partial[((long)k * p.M + m) * p.Bpad + b] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
      for (int k = 8; k < p.ksplit; ++k) s += p.partial[((long)k * p.M + m) * p.Bpad + b];
    }
    tile[ty + i * 8][tx] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = b0 + ty + i * 8, m = m0 + tx;
    if (m < p.M && b < p.B) {
      const float scale = T::to_float(p.scales[m]);
      const float bias = p.bias ? T::to_float(p.bias[m]) : 0.f;
      p.Y[(long)b * p.ys + m] = T::from_float(__builtin_fmaf(tile[tx][ty + i * 8], scale, bias));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA pipeline (default since round 3): every global read of the main loop is a global_load_lds -- the X tiles,
// the codes AND the codebook gathers -- so nothing of the stream occupies VGPRs, the wave's VMEM queue holds one kind
// of operation (hipcc waits with vmcnt(0) for any register load that sits beside an LDS-DMA, cdna_hip_programming.md
// section 5 trap (b)) and all waits are counted by hand (static counts, raw s_barrier).
//
//   block  = 8 waves = 128 output rows (16 per wave) x one K slice x <= 128 batch columns; 1 block per CU
//   MFMA   = v_mfma_f32_16x16x32: A = W (lane (r, kg): row r, 8 k of k-group kg = ONE codebook vector for g = 8, one
//            half of one for g = 16), B = X^T (lane (r, kg): batch column 16 t + r, the same 8 k), C = 4 consecutive
//            output rows of one batch column per lane -> partial / Y stores of 16 B
//   gather = one global_load_lds_dwordx4 per 32-deep k step and wave: lane (r, kg) names the address of ITS fragment,
//            the DMA drops the 64 fragments into LDS in lane order, and the A operand is read back with one
//            conflict-free ds_read_b128.  The gathered entry is still the MFMA fragment; LDS is only the landing zone
//            that lets NS - 1 = 3 chunks (48 wave-gathers = 3072 lane-gathers per CU) be in flight without registers.
//   codes  = 16 B per row and 64-deep chunk, DMA'd 2 NS - 1 chunks ahead into a wave-private ring, read back one
//            iteration before the gather that needs them
//   X      = Bpad x 64 k chunks into the XOR-swizzled image of xswz() (swizzle applied to the SOURCE address: the DMA
//            writes lane-linear), shared by the 8 waves
//   K split over the blocks of one XCD (block -> XCD affinity b % 8, speed only): fp32 partials [ks][b][m] stay in that
//   XCD's L2 for the finalize kernel, which is mapped the same way.  One barrier per chunk.
//
// Per-CU traffic through the L1 fill path for 4096 x 4096, B = 128: 8192 gathered lines (1 MiB) + 128 KiB of X.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class T>
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c);
template <>
__device__ __forceinline__ f32x4 mfma16<F16>(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<BF16>(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

typedef __attribute__((address_space(3))) const u32x4* glds_u32x4_ptr;
typedef __attribute__((address_space(3))) const uint16_t* glds_u16_ptr;
typedef __attribute__((address_space(3))) void* glds_void_ptr;
typedef __attribute__((address_space(1))) const void* ggbl_void_ptr;

struct GldsParams {
  const uint8_t* codes;     // [M][in_groups] u16
  const uint8_t* codebook;  // [65536][G] halfs
  const uint16_t* X;        // [B][xs]
  float* partial;           // [ksplit][B][M] fp32 (ksplit > 1)
  const uint16_t* scales;   // ksplit == 1: the block writes Y itself
  const uint16_t* bias;
  uint16_t* Y;
  long xs, ys;
  int M, B, in_groups;
  int row_blocks, ksplit;
  int chunks_base, chunks_rem;  // K slice ks covers chunks_base + (ks < chunks_rem) chunks of 64 k
  int store_nt;                 // partial stores with the non-temporal hint (tuning knob, A/B runs)
  int dbg;                      // knock-out switches for timing experiments (results are wrong): 1 no MFMA, 2 no fragment reads, 4 no stores, 8 no X DMA, 16 gathers from one line
};

constexpr int GL_NS = 4;       // stages of the W / X ring
constexpr int GL_NSLOT = 8;    // slots of a wave's code ring (>= GL_NS + 1, power of two)
constexpr int GL_WAVES = 8;      // consumer waves (16 output rows each)
constexpr int GL_PRODUCERS = 4;  // DMA waves
constexpr uint32_t GL_W_BYTES = GL_WAVES * 2048u;  // per stage: 2 fragments of 1 KiB per wave

template <int XW>
struct GldsLds {
  static constexpr uint32_t X_BYTES = XW * 64u * 128u;        // XW * 64 batch rows x 64 k
  static constexpr uint32_t W_BYTES = GL_W_BYTES;             // per stage: 2 fragments of 1 KiB per consumer wave
  static constexpr uint32_t STAGE = W_BYTES + X_BYTES;
  static constexpr uint32_t CODES = GL_NS * STAGE;            // [producer][slot][512 B]
  static constexpr uint32_t TOTAL = CODES + GL_PRODUCERS * GL_NSLOT * 512u;
};

constexpr int gl_vmcnt(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }  // vmcnt(n) only

// Roles.  Measured with every wave doing both jobs (round 3, profiles/r03_mb_gemm_symmetric.log): a chunk took
// TA time + compute time -- a wave that is issuing LDS-DMA sits in the issue stage while the texture addresser works
// through the 64 scattered lines of each gather (its queue is a few instructions deep), and the addresser idles while the
// waves compute.  So the block is split: 8 consumer waves (16 rows each) only read fragments and run MFMAs, 4 producer
// waves (one per SIMD) only issue DMA -- codes, gathers, X -- and are parked in the issue stage almost all the time,
// which keeps the addresser fed.  One s_barrier per chunk joins them: the producers arrive when the chunk's DMA has
// landed (counted vmcnt), the consumers when they are done with the previous chunk (whose stage is refilled next).
template <class T, int G, int NBT, int XW>  // NBT = 16-column batch tiles computed; XW * 64 = batch rows staged
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_1x16_glds_kernel(const GldsParams p) {
  using LDS = GldsLds<XW>;
  constexpr int P = 1 + 4 + 2 * XW;  // LDS-DMA operations per producer wave and chunk: codes, 4 fragments, 2 XW pieces of X
  static_assert((GL_NS - 2) * P < 64, "vmcnt is 6 bits");
  extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)glds_smem != 0u) __builtin_trap();  // LDS map below starts at 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int arow = lane & 15, kg = lane >> 4;

  // block -> (row block, K slice): the K slices of a row block run on one XCD (observed placement: block b on XCD b % 8)
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int rb_hi = slot / p.ksplit;
  const int row_blk = rb_hi * 8 + xcd, ks = slot - rb_hi * p.ksplit;
  if (row_blk >= p.row_blocks) return;
  const int n = p.chunks_base + (ks < p.chunks_rem ? 1 : 0);                                // chunks of this block, >= GL_NS - 1
  const int chunk0 = ks * p.chunks_base + (ks < p.chunks_rem ? ks : p.chunks_rem);

  if (wave >= GL_WAVES) {
    // ============================================ producer ============================================================
    const int pw = wave - GL_WAVES;                 // serves consumer waves 2 pw and 2 pw + 1: 32 rows
    if (!(p.dbg & 32)) __builtin_amdgcn_s_setprio(3);  // the address arithmetic of the DMA wave goes ahead of the consumers' issue slots
    const int prow0 = row_blk * 128 + pw * 32;
    // codes: g = 8: lanes 0..31 move 16 B (8 codes) of row prow0 + lane; g = 16: all lanes move 4 B (2 codes), row lane / 2
    constexpr int CODE_LANES = G == 8 ? 32 : 64;
    constexpr int CODE_BYTES_PER_CHUNK = G == 8 ? 16 : 8;  // per row
    const uint8_t* code_src;
    {
      int r = prow0 + (G == 8 ? lane : (lane >> 1));
      r = r < p.M ? r : p.M - 1;
      code_src = p.codes + ((size_t)r * p.in_groups) * 2 + (size_t)chunk0 * CODE_BYTES_PER_CHUNK + (G == 8 ? 0 : (lane & 1) * 4);
    }
    // X: piece (pw * 2 XW + x) of a chunk image = 64 slots of 16 B; slot s holds batch row s >> 3, k piece (s & 7) ^ swizzle
    const uint8_t* x_src[2 * XW];
#pragma unroll
    for (int x = 0; x < 2 * XW; ++x) {
      const int s = (pw * 2 * XW + x) * 64 + lane;
      int b = s >> 3;
      const int c = (s & 7) ^ ((b >> 1) & 7);
      b = b < p.B ? b : p.B - 1;  // rows past the batch: a valid row, computed and never stored
      x_src[x] = (const uint8_t*)(p.X + (size_t)b * p.xs + (size_t)chunk0 * 64 + c * 8);
    }
    const uint32_t code_ring = LDS::CODES + (uint32_t)pw * (GL_NSLOT * 512u);
    // this lane's codes of a chunk inside a ring slot: consumer wave cw, k step s -> row 16 cw + arow; g = 8: code 4 s + kg,
    // g = 16: code 2 s + kg / 2 (and the lane takes the 16-B half kg & 1 of the 32-B entry)
    constexpr uint32_t ROW_BYTES = G == 8 ? 16u : 8u;
    constexpr uint32_t CODE_STEP = G == 8 ? 8u : 4u;  // bytes between the codes of k step 0 and k step 1
    const uint32_t code_off0 = (uint32_t)arow * ROW_BYTES + (G == 8 ? (uint32_t)kg * 2u : (uint32_t)(kg >> 1) * 2u);
    const uint32_t half_off = G == 8 ? 0u : (uint32_t)(kg & 1) * 16u;

    auto dma_codes = [&](int chunk) {  // chunk may run past the slice: clamped (the slot is written, never used)
      const int cc = chunk < n ? chunk : n - 1;
      if (lane < CODE_LANES) {
        ggbl_void_ptr src = (ggbl_void_ptr)(code_src + (size_t)cc * CODE_BYTES_PER_CHUNK);
        glds_void_ptr dst = (glds_void_ptr)(size_t)(code_ring + (uint32_t)(chunk & (GL_NSLOT - 1)) * 512u);
        if constexpr (G == 8) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
        else __builtin_amdgcn_global_load_lds(src, dst, 4, 0, 0);
      }
    };
    auto read_codes = [&](int chunk, uint32_t (&c)[4]) {
      const uint32_t a = code_ring + (uint32_t)(chunk & (GL_NSLOT - 1)) * 512u + code_off0;
#pragma unroll
      for (int cw = 0; cw < 2; ++cw)
#pragma unroll
        for (int s = 0; s < 2; ++s) c[cw * 2 + s] = *(glds_u16_ptr)(size_t)(a + (uint32_t)cw * (16u * ROW_BYTES) + (uint32_t)s * CODE_STEP);
    };
    auto dma_stage = [&](int chunk, int stage, const uint32_t (&c)[4]) {
      const uint32_t base = (uint32_t)stage * LDS::STAGE;
#pragma unroll
      for (int f = 0; f < 4; ++f)  // fragment f = (consumer wave 2 pw + f / 2, k step f % 2)
        __builtin_amdgcn_global_load_lds((ggbl_void_ptr)(p.codebook + (size_t)((p.dbg & 16) ? (c[f] & 7u) : c[f]) * (G * 2) + half_off),
                                         (glds_void_ptr)(size_t)(base + (uint32_t)(2 * pw) * 2048u + (uint32_t)f * 1024u), 16, 0, 0);
#pragma unroll
      for (int x = 0; x < 2 * XW; ++x)
        __builtin_amdgcn_global_load_lds((ggbl_void_ptr)(x_src[x] + (size_t)((p.dbg & 8) ? 0 : chunk) * 128),
                                         (glds_void_ptr)(size_t)(base + LDS::W_BYTES + (uint32_t)(pw * 2 * XW + x) * 1024u), 16, 0, 0);
    };
    // prologue: codes of the first NS chunks (one exposed round trip: the gathers depend on them), then NS - 1 issue
    // iterations.  Issue iteration i: codes of chunk i + 2 NS - 1, stage of chunk q = i + NS - 1, read back the codes of
    // chunk q + 1 (their DMA is NS iterations old: covered by the wait that opens iteration i).
    uint32_t creg[4];
#pragma unroll
    for (int j = 0; j < GL_NS; ++j) dma_codes(j);
    __builtin_amdgcn_s_waitcnt(gl_vmcnt(0));
    read_codes(0, creg);
#pragma unroll
    for (int i = -(GL_NS - 1); i < 0; ++i) {
      const int q = i + GL_NS - 1;
      dma_codes(q + GL_NS);
      dma_stage(q, q, creg);
      read_codes(q + 1, creg);
    }
    int stage = 0;  // = i % NS
    int i = 0;
    for (; i + GL_NS - 1 < n; ++i) {
      __builtin_amdgcn_s_waitcnt(gl_vmcnt((GL_NS - 2) * P));  // what this wave issued NS - 1 iterations ago has landed
      __builtin_amdgcn_s_barrier();                           // chunk i is complete; the consumers are done with chunk i - 1
      const int q = i + GL_NS - 1;
      const int qstage = stage == 0 ? GL_NS - 1 : stage - 1;  // = q % NS
      dma_codes(q + GL_NS);
      dma_stage(q, qstage, creg);
      read_codes(q + 1, creg);
      stage = stage == GL_NS - 1 ? 0 : stage + 1;
    }
    static_assert(GL_NS == 4, "the tail below is written out for three iterations");
    __builtin_amdgcn_s_waitcnt(gl_vmcnt(2 * P));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_waitcnt(gl_vmcnt(1 * P));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_waitcnt(gl_vmcnt(0));
    __builtin_amdgcn_s_barrier();
    return;
  }

  // ============================================== consumer ==============================================================
  // register blocking: a wave owns RT row tiles x CT batch tiles (2 x NBT / 2 from two batch tiles on): per k step
  // RT + CT fragment reads feed RT * CT MFMAs -- the consumers' ds_reads share the LDS port with the landing DMA
  constexpr int RT = NBT >= 2 ? 2 : 1, CT = NBT / RT;
  static_assert(RT * CT == NBT, "batch tiles split evenly");
  const int rp = wave % (GL_WAVES / RT), bh = wave / (GL_WAVES / RT);  // row-tile group, batch-tile group
  f32x4 acc[RT][CT];
#pragma unroll
  for (int j = 0; j < RT; ++j)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int stage = 0;
  for (int c = 0; c < n; ++c) {
    __builtin_amdgcn_s_barrier();
    const uint32_t base = (uint32_t)stage * LDS::STAGE;
    // all fragment reads of the chunk are issued before the first MFMA (left alone, hipcc pairs every read with an
    // lgkmcnt(0) and the wave pays one LDS latency per MFMA)
    if (p.dbg & 2) continue;
    u32x4 a[2][RT], b[2][CT];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < RT; ++j)
        a[s][j] = *(glds_u32x4_ptr)(size_t)(base + (uint32_t)(rp * RT + j) * 2048u + (uint32_t)s * 1024u + (uint32_t)lane * 16u);
#pragma unroll
      for (int t = 0; t < CT; ++t)
        b[s][t] = *(glds_u32x4_ptr)(size_t)(base + LDS::W_BYTES + (uint32_t)xswz((bh * CT + t) * 16 + arow, s * 4 + kg) * 16u);
    }
    if (p.dbg & 1) {  // reads kept alive, no matrix work
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < RT; ++j)
#pragma unroll
          for (int t = 0; t < CT; ++t) acc[j][t][0] += __uint_as_float((a[s][j].x ^ b[s][t].y) & 0x007fffffu);
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < RT; ++j)
#pragma unroll
          for (int t = 0; t < CT; ++t) acc[j][t] = mfma16<T>(a[s][j], b[s][t], acc[j][t]);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * (RT + CT), 0);  // DS reads first ...
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NBT, 0);        // ... then the MFMAs
    }
    stage = stage == GL_NS - 1 ? 0 : stage + 1;
  }

  // ---- epilogue: lane (arow, kg) holds rows 4 kg .. 4 kg + 3 of row tile rp RT + j, batch column 16 (bh CT + t) + arow ------
  if (p.dbg & 4) {  // no stores (one lane keeps the accumulators alive)
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < RT; ++j)
#pragma unroll
      for (int t = 0; t < CT; ++t) v += acc[j][t][0] + acc[j][t][1] + acc[j][t][2] + acc[j][t][3];
    if (v == 1.2345e-30f) p.partial[0] = v;
    return;
  }
  const bool vec = (p.M & 3) == 0;  // 16-B aligned partial rows / 8-B aligned Y rows
#pragma unroll
  for (int j = 0; j < RT; ++j) {
    const int m = row_blk * 128 + (rp * RT + j) * 16 + kg * 4;
    if (p.ksplit > 1) {
      float* out = p.partial + (size_t)ks * p.B * p.M;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int b = (bh * CT + t) * 16 + arow;
        if (b < p.B && m < p.M) {
          float* dst = out + (size_t)b * p.M + m;
          if (vec) {
            if (p.store_nt) __builtin_nontemporal_store(acc[j][t], reinterpret_cast<f32x4*>(dst));
            else *reinterpret_cast<f32x4*>(dst) = acc[j][t];
          } else
            for (int r = 0; r < 4; ++r)
              if (m + r < p.M) dst[r] = acc[j][t][r];
        }
      }
    } else {
      float sc[4], bi[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = m + r < p.M ? m + r : p.M - 1;
        sc[r] = T::to_float(p.scales[mm]);
        bi[r] = p.bias ? T::to_float(p.bias[mm]) : 0.f;
      }
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int b = (bh * CT + t) * 16 + arow;
        if (b < p.B && m < p.M) {
          uint16_t* dst = p.Y + (size_t)b * p.ys + m;
          uint16_t h[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = T::from_float(__builtin_fmaf(acc[j][t][r], sc[r], bi[r]));
          if (vec && (p.ys & 3) == 0 && ((uintptr_t)dst & 7u) == 0) *reinterpret_cast<u32x2*>(dst) = u32x2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
          else
            for (int r = 0; r < 4; ++r)
              if (m + r < p.M) dst[r] = h[r];
        }
      }
    }
  }
}

// Y[b][m] = (sum_ks partial[ks][b][m]) * scales[m] + bias[m]: thread = 4 consecutive m of one batch row, all K slices
// requested before the first add.  Blocks are mapped like the main kernel's (row block -> XCD row_blk % 8), so the
// partials are read from the L2 they were written to (speed only).
struct GldsFinalizeParams {
  const float* partial;
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* Y;
  long ys;
  int M, B, ksplit, row_blocks, bchunks, rb_rows;  // rb_rows: rows of a row block (128 or 64), 32 threads x 4 rows cover 128
};

template <class T>
__global__ __launch_bounds__(256) void gemm_glds_finalize_kernel(const GldsFinalizeParams p) {
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int rb_hi = slot / p.bchunks;
  const int row_blk = rb_hi * 8 + xcd, bc = slot - rb_hi * p.bchunks;
  if (row_blk >= p.row_blocks) return;
  // a block covers 128 consecutive rows: one row block of the main kernel, or two of its 64-row blocks (same XCD pairing
  // is not attempted for those: speed only)
  const int m = row_blk * 128 + ((int)threadIdx.x & 31) * 4;
  const int b = bc * 8 + ((int)threadIdx.x >> 5);
  if (m >= p.M || b >= p.B) return;
  const size_t plane = (size_t)p.B * p.M;
  const float* src = p.partial + (size_t)b * p.M + m;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  float sc[4], bi[4];  // requested before the partials: one exposed round trip instead of two
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int mm = m + r < p.M ? m + r : p.M - 1;
    sc[r] = T::to_float(p.scales[mm]);
    bi[r] = p.bias ? T::to_float(p.bias[mm]) : 0.f;
  }
  if ((p.M & 3) == 0) {
    int k = 0;
    for (; k + 8 <= p.ksplit; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(k + j) * plane);
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += v[j][r];
    }
    for (; k < p.ksplit; ++k) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)k * plane);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] += v[r];
    }
  } else {
    for (int k = 0; k < p.ksplit; ++k)
      for (int r = 0; r < 4; ++r)
        if (m + r < p.M) s[r] += src[(size_t)k * plane + r];
  }
  uint16_t h[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) h[r] = T::from_float(__builtin_fmaf(s[r], sc[r], bi[r]));
  uint16_t* dst = p.Y + (size_t)b * p.ys + m;
  if ((p.M & 3) == 0 && (p.ys & 3) == 0 && ((uintptr_t)dst & 7u) == 0) *reinterpret_cast<u32x2*>(dst) = u32x2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
  else
    for (int r = 0; r < 4; ++r)
      if (m + r < p.M) dst[r] = h[r];
}

struct GldsPlan {
  int row_blocks, ksplit, chunks_base, chunks_rem, nbt, xw;
};

// K split: enough blocks for one round of the chip, K slices of >= 8 chunks (512 k) where K allows, never fewer than
// NS - 1 chunks (the pipeline's prologue), chunks dealt evenly.
static bool plan_glds(int B, int M, int K, GldsPlan& g) {
  const int kchunks = K / BK;
  if (K % BK != 0 || kchunks < GL_NS - 1 || B < 1 || B > 128) return false;
  g.row_blocks = (M + 127) / 128;
  int ksplit = std::max(1, 256 / g.row_blocks);
  ksplit = std::min(ksplit, std::max(1, kchunks / 8));
  ksplit = std::min(ksplit, kchunks / (GL_NS - 1));
  g.ksplit = std::max(1, ksplit);
  g.chunks_base = kchunks / g.ksplit;
  g.chunks_rem = kchunks % g.ksplit;
  const int nbt = (B + 15) / 16;
  g.nbt = nbt <= 2 ? nbt : (nbt <= 4 ? 4 : (nbt <= 6 ? 6 : 8));
  g.xw = g.nbt <= 4 ? 1 : 2;
  return true;
}

template <class T, int G>
static int launch_glds(const GldsParams& p, const GldsPlan& g, hipStream_t stream) {
  const int rb8 = ((p.M + 127) / 128 + 7) / 8;  // 128-row groups per XCD
  const dim3 grid((unsigned)(8 * rb8 * p.ksplit));
  auto go = [&](auto kern, size_t lds) -> int {
    if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
    hipLaunchKernelGGL(kern, grid, dim3((GL_WAVES + GL_PRODUCERS) * 64), lds, stream, p);
    return check_hip(hipGetLastError(), "gemm_1x16_glds launch");
  };
  switch (g.nbt) {
    case 1: return go(gemm_1x16_glds_kernel<T, G, 1, 1>, GldsLds<1>::TOTAL);
    case 2: return go(gemm_1x16_glds_kernel<T, G, 2, 1>, GldsLds<1>::TOTAL);
    case 4: return go(gemm_1x16_glds_kernel<T, G, 4, 1>, GldsLds<1>::TOTAL);
    case 6: return go(gemm_1x16_glds_kernel<T, G, 6, 2>, GldsLds<2>::TOTAL);
    default: return go(gemm_1x16_glds_kernel<T, G, 8, 2>, GldsLds<2>::TOTAL);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// 16-row blocks, no K split (round 4).  The pipeline above minimises what goes through a CU's L1 (X is read once per K slice)
// and pays for it with two launches, 16.8 MB of fp32 partials and a finalize: 8.3 + 3.9 us of a 23 us op at 128 rows, and the
// same 8.3 us at 16 rows.  Here a block owns 16 output rows over ALL of K: no partials, no workspace, no second kernel -- and
// every block streams all of X (B x K x 2 bytes) through its L1 next to its 8192 gathers.  That trade wins when X is small
// next to the gathers' 128-B lines (16 x 8 KiB rows of X = 128 KiB against 1 MiB of lines) and is about even at 128 rows.
//   wave 0            codes -> LDS -> 2 gathers per 64-k chunk (one 1-KiB MFMA A fragment per 32 k), NSW - 1 steps ahead: the
//                     gathers are latency-bound (a few hundred must be in flight), the fragments are small (2 KiB per chunk)
//   waves 1..NXP      the X chunks (B rows x 128 B each, XOR-swizzled slots as above) by LDS-DMA, NSX - 1 steps ahead
//   R16_NC consumers  wave c multiplies the block's ONE A fragment with its batch tiles; fp32 accumulators over all of K,
//                     scale + bias + one rounding in the epilogue (16 x 16 tile: 4 rows x 1 batch column per lane)
// One s_barrier per STEP of CPB chunks joins the three roles (counted vmcnt on the DMA waves, as above).  A step lasts only
// 0.15-0.35 us per chunk, and between "my DMA of this step is accepted" and "everybody has passed the barrier" the texture
// addresser has nothing queued: with one chunk per step that gap was a quarter of the time (measured: 0.30 gathers / clk / CU
// against the 0.43 of the flat-out microbenchmark), so small batches take 4 chunks per step.
constexpr int R16_NC = 4;     // consumer waves
// where the op takes this kernel (measured, profiles/r04_gemm_rows16_shapes.json): every block streams all of X, so the trade
// turns with batch x layer size -- up to 16 rows everywhere, up to 64 rows for layers of <= 4096 x 4096
constexpr int R16_ROWS_ANY = 16, R16_ROWS_SMALL = 64;
constexpr long R16_SMALL_LAYER = 1L << 24;

template <int NBT, int CPB>
struct R16Lds {
  static constexpr int NXP = 2 * NBT >= 4 ? 4 : 2 * NBT;            // X producer waves
  static constexpr int PXW = CPB * 2 * NBT / NXP;                   // 1-KiB pieces (8 batch rows x 64 k) per X wave and step
  static constexpr uint32_t X_CHUNK = (uint32_t)NBT * 2048u;        // 16 NBT batch rows x 128 B
  static constexpr uint32_t X_STAGE = CPB * X_CHUNK;
  static constexpr uint32_t W_STAGE = CPB * 2048u;                  // 2 fragments of 1 KiB per chunk
  // Ring depths.  Steps of several chunks carry 256-512 lane-gathers each, so 3 steps in flight cover the gather latency; short
  // rings also mean a short prologue and, at <= 16 rows, 72 KiB of LDS: two blocks per CU, whose prologues and tails overlap
  // (measured against 8-stage rings: 4096 -> 11008 at 16 rows 33.1 -> 30.5 us, 4096^2 13.7 -> 13.2).
#ifndef AQLM_R16_DEEP
#define AQLM_R16_DEEP 0  // 1: the first cut's 8-stage rings (A/B builds)
#endif
  static constexpr int NSW = CPB == 1 ? 16 : (AQLM_R16_DEEP ? 8 : 4);  // stages of the fragment ring
  static constexpr int NSX = CPB == 1 ? (NBT <= 2 ? 16 : (NBT <= 4 ? 12 : 7))   // stages of the X ring: as deep in TIME as a load takes,
                             : AQLM_R16_DEEP ? (CPB == 4 ? (NBT <= 1 ? 8 : 4) : (NBT <= 4 ? 6 : 3))
                                             : (CPB == 4 ? (NBT <= 1 ? 4 : 3) : (NBT <= 4 ? 4 : 3));  // within 160 KiB
  static constexpr int NSLOT = 2 * NSW;                             // slots of the code ring (power of two)
  static constexpr uint32_t CODE_SLOT = 256u * CPB;                 // 16 rows x CPB x 16 B (g = 8) or x 8 B (g = 16, half used)
  static constexpr uint32_t W = 0;
  static constexpr uint32_t X = NSW * W_STAGE;
  static constexpr uint32_t CODES = X + NSX * X_STAGE;
  static constexpr uint32_t TOTAL = CODES + NSLOT * CODE_SLOT;
  static constexpr int WAVES = 1 + NXP + R16_NC;
  static_assert(TOTAL <= 160u * 1024u, "LDS");
  static_assert((NSLOT & (NSLOT - 1)) == 0, "code ring");
};

template <int R, int P>  // the last R + 1 steps land: one counted wait + barrier each (the wait count must be an immediate)
__device__ __forceinline__ void r16_drain() {
  __builtin_amdgcn_s_waitcnt(gl_vmcnt(R * P));
  __builtin_amdgcn_s_barrier();
  if constexpr (R > 0) r16_drain<R - 1, P>();
}

struct R16Params {
  const uint8_t* codes;     // [M][in_groups] u16
  const uint8_t* codebook;  // [65536][G] halfs
  const uint16_t* X;        // [B][xs]
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* Y;
  long xs, ys;
  int M, B, in_groups, nsteps;  // nsteps = K / (64 CPB)
};

template <class T, int G, int NBT, int CPB>
__global__ __launch_bounds__((R16Lds<NBT, CPB>::WAVES * 64)) void gemm_1x16_rows16_kernel(const R16Params p) {
  using LDS = R16Lds<NBT, CPB>;
  constexpr int NSW = LDS::NSW, NSX = LDS::NSX;
  extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)glds_smem != 0u) __builtin_trap();  // LDS map above starts at 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int arow = lane & 15, kg = lane >> 4;
  const int row0 = (int)blockIdx.x * 16;
  const int n = p.nsteps;  // >= max(NSW, NSX) - 1 (host)

  if (wave == 0) {
    // ============================================ gather producer ====================================================
    __builtin_amdgcn_s_setprio(3);
    constexpr int P0 = 1 + 2 * CPB;  // LDS-DMA operations per step: codes, 2 fragments per chunk
    static_assert((NSW - 2) * P0 < 64, "vmcnt is 6 bits");
    // codes of a step: row r holds ROW_BYTES = CPB x 16 B (g = 8: 8 codes per chunk) or CPB x 8 B (g = 16: 4 codes); one DMA
    // instruction moves the 16 rows in pieces of 16 B (4 B when a row has only 8), lanes in (row, piece) order, so a slot of the
    // ring reads [row][chunk][codes]
    constexpr uint32_t CHUNK_BYTES = G == 8 ? 16u : 8u;  // code bytes of one row and chunk
    constexpr uint32_t RB = CHUNK_BYTES * CPB;
    constexpr uint32_t DMA_BYTES = RB >= 16u ? 16u : 4u;
    constexpr int PARTS = (int)(RB / DMA_BYTES);          // lanes per row
    constexpr int CODE_LANES = 16 * PARTS;
    static_assert(CODE_LANES <= 64, "one DMA instruction moves the codes of a step");
    const uint8_t* code_src;
    {
      int r = row0 + lane / PARTS;
      r = r < p.M ? r : p.M - 1;
      code_src = p.codes + ((size_t)r * p.in_groups) * 2 + (size_t)(lane % PARTS) * DMA_BYTES;
    }
    constexpr uint32_t ROW_BYTES = CHUNK_BYTES * CPB;
    constexpr uint32_t CODE_STEP = G == 8 ? 8u : 4u;  // bytes between the codes of k step 0 and k step 1 of a chunk
    const uint32_t code_off0 = (uint32_t)arow * ROW_BYTES + (G == 8 ? (uint32_t)kg * 2u : (uint32_t)(kg >> 1) * 2u);
    const uint32_t half_off = G == 8 ? 0u : (uint32_t)(kg & 1) * 16u;
    auto dma_codes = [&](int step) {  // step may run past K: clamped (the slot is written, never used)
      const int cc = step < n ? step : n - 1;
      if (lane < CODE_LANES) {
        ggbl_void_ptr src = (ggbl_void_ptr)(code_src + (size_t)cc * RB);
        glds_void_ptr dst = (glds_void_ptr)(size_t)(LDS::CODES + (uint32_t)(step & (LDS::NSLOT - 1)) * LDS::CODE_SLOT);
        if constexpr (DMA_BYTES == 16u) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
        else __builtin_amdgcn_global_load_lds(src, dst, 4, 0, 0);
      }
    };
    auto read_codes = [&](int step, uint32_t (&c)[2 * CPB]) {
      const uint32_t a = LDS::CODES + (uint32_t)(step & (LDS::NSLOT - 1)) * LDS::CODE_SLOT + code_off0;
#pragma unroll
      for (int u = 0; u < CPB; ++u)
#pragma unroll
        for (int s = 0; s < 2; ++s) c[u * 2 + s] = *(glds_u16_ptr)(size_t)(a + (uint32_t)u * CHUNK_BYTES + (uint32_t)s * CODE_STEP);
    };
    auto dma_stage = [&](int stage, const uint32_t (&c)[2 * CPB]) {
#pragma unroll
      for (int f = 0; f < 2 * CPB; ++f)
        __builtin_amdgcn_global_load_lds((ggbl_void_ptr)(p.codebook + (size_t)c[f] * (G * 2) + half_off),
                                         (glds_void_ptr)(size_t)(LDS::W + (uint32_t)stage * LDS::W_STAGE + (uint32_t)f * 1024u), 16, 0, 0);
    };
    uint32_t creg[2 * CPB];
#pragma unroll
    for (int j = 0; j < NSW; ++j) dma_codes(j);
    __builtin_amdgcn_s_waitcnt(gl_vmcnt(0));
    read_codes(0, creg);
    for (int q = 0; q < NSW - 1; ++q) {  // steps 0 .. NSW - 2 before the first barrier
      dma_codes(q + NSW);
      dma_stage(q, creg);
      read_codes(q + 1, creg);  // requested before the wait above (q + 1 < NSW)
    }
    int stage = NSW - 1;  // stage of step i + NSW - 1
    int i = 0;
    for (; i + NSW - 1 < n; ++i) {
      __builtin_amdgcn_s_waitcnt(gl_vmcnt((NSW - 2) * P0));  // what this wave issued NSW - 1 iterations ago has landed
      __builtin_amdgcn_s_barrier();                           // step i is complete; the consumers are done with step i - 1
      const int q = i + NSW - 1;
      dma_codes(q + NSW);
      dma_stage(stage, creg);
      read_codes(q + 1, creg);  // requested NSW - 1 iterations ago: covered by the wait that opened this iteration
      stage = stage == NSW - 1 ? 0 : stage + 1;
    }
    r16_drain<NSW - 2, P0>();  // steps n - NSW + 1 .. n - 1 are in flight, one more lands per barrier
    return;
  }

  if (wave <= LDS::NXP) {
    // ============================================== X producer =======================================================
    constexpr int PX = LDS::PXW;
    constexpr int PPC = 2 * NBT;  // pieces per chunk
    static_assert((NSX - 2) * PX < 64, "vmcnt is 6 bits");
    const int xw = wave - 1;
    const uint8_t* x_src[PX];
    uint32_t x_dst[PX];
#pragma unroll
    for (int x = 0; x < PX; ++x) {
      const int piece = xw * PX + x;             // of the step: chunk piece / PPC, chunk piece piece % PPC
      const int u = piece / PPC, pc = piece % PPC;
      const int s = pc * 64 + lane;              // slot of the chunk image: batch row s >> 3, k piece (s & 7) ^ swizzle
      int b = s >> 3;
      const int c = (s & 7) ^ ((b >> 1) & 7);
      b = b < p.B ? b : p.B - 1;  // rows past the batch: a valid row, computed and never stored
      x_src[x] = (const uint8_t*)(p.X + (size_t)b * p.xs + c * 8) + (size_t)u * 128;
      x_dst[x] = LDS::X + (uint32_t)u * LDS::X_CHUNK + (uint32_t)pc * 1024u;
    }
    auto dma_x = [&](int step, int stage) {
      const int cc = step < n ? step : n - 1;
#pragma unroll
      for (int x = 0; x < PX; ++x)
        __builtin_amdgcn_global_load_lds((ggbl_void_ptr)(x_src[x] + (size_t)cc * (128 * CPB)),
                                         (glds_void_ptr)(size_t)(x_dst[x] + (uint32_t)stage * LDS::X_STAGE), 16, 0, 0);
    };
    for (int q = 0; q < NSX - 1; ++q) dma_x(q, q);
    int stage = NSX - 1;
    int i = 0;
    for (; i + NSX - 1 < n; ++i) {
      __builtin_amdgcn_s_waitcnt(gl_vmcnt((NSX - 2) * PX));
      __builtin_amdgcn_s_barrier();
      dma_x(i + NSX - 1, stage);
      stage = stage == NSX - 1 ? 0 : stage + 1;
    }
    r16_drain<NSX - 2, PX>();
    return;
  }

  // ================================================ consumer ============================================================
  constexpr int CT = NBT >= R16_NC ? NBT / R16_NC : 1;  // batch tiles per consumer wave
  const int cw = wave - 1 - LDS::NXP;
  const bool active = cw * CT < NBT;
  f32x4 acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int sw = 0, sx = 0;
  for (int c = 0; c < n; ++c) {
    __builtin_amdgcn_s_barrier();
    if (active) {
      const uint32_t wbase = LDS::W + (uint32_t)sw * LDS::W_STAGE, xbase = LDS::X + (uint32_t)sx * LDS::X_STAGE;
#pragma unroll
      for (int u = 0; u < CPB; ++u) {
        u32x4 a[2], b[2][CT];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          a[s] = *(glds_u32x4_ptr)(size_t)(wbase + (uint32_t)(u * 2 + s) * 1024u + (uint32_t)lane * 16u);
#pragma unroll
          for (int t = 0; t < CT; ++t)
            b[s][t] = *(glds_u32x4_ptr)(size_t)(xbase + (uint32_t)u * LDS::X_CHUNK + (uint32_t)xswz((cw * CT + t) * 16 + arow, s * 4 + kg) * 16u);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int t = 0; t < CT; ++t) acc[t] = mfma16<T>(a[s], b[s][t], acc[t]);
      }
    }
    sw = sw == NSW - 1 ? 0 : sw + 1;
    sx = sx == NSX - 1 ? 0 : sx + 1;
  }
  if (!active) return;
  // ---- epilogue: lane (arow, kg) holds rows 4 kg .. 4 kg + 3 of the block, batch column 16 (cw CT + t) + arow --------------
  const int m = row0 + kg * 4;
  if (m >= p.M) return;
  float sc[4], bi[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int mm = m + r < p.M ? m + r : p.M - 1;
    sc[r] = T::to_float(p.scales[mm]);
    bi[r] = p.bias ? T::to_float(p.bias[mm]) : 0.f;
  }
  const bool vec = (p.M & 3) == 0 && (p.ys & 3) == 0 && ((uintptr_t)p.Y & 7u) == 0;  // (Y only 2- / 4-byte aligned: scalar stores)
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int b = (cw * CT + t) * 16 + arow;
    if (b < p.B) {
      uint16_t* dst = p.Y + (size_t)b * p.ys + m;
      uint16_t h[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = T::from_float(__builtin_fmaf(acc[t][r], sc[r], bi[r]));
      if (vec) *reinterpret_cast<u32x2*>(dst) = u32x2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
      else
        for (int r = 0; r < 4; ++r)
          if (m + r < p.M) dst[r] = h[r];
    }
  }
}

struct R16Plan {
  int nbt, cpb, nsteps;
};

// chunks per step: 4 for <= 32 rows, 2 above, where K is a multiple of the step and long enough for the rings; else 1
static bool plan_rows16(int B, int K, int G, R16Plan& r) {
  if (K % BK != 0 || B < 1 || B > 128) return false;
  const int t = (B + 15) / 16;
  r.nbt = t <= 1 ? 1 : (t <= 2 ? 2 : (t <= 4 ? 4 : 8));
  const int want = r.nbt <= 2 ? 4 : 2;
  const int chunks = K / BK;
  // (steps of several chunks move the codes in 16-B pieces: the rows of the code matrix must then be 16-B aligned)
  if (chunks % want == 0 && chunks / want >= 7 && ((K / G) * 2) % 16 == 0) r.cpb = want;  // (7 steps: any ring depth of the several-chunk configurations)
  else if (chunks >= 15) r.cpb = 1;                            // rings of 16 / <= 16 stages
  else return false;
  r.nsteps = chunks / r.cpb;
  return true;
}

template <class T, int G>
static int launch_rows16(const R16Params& p, const R16Plan& r, hipStream_t stream) {
  const dim3 grid((unsigned)((p.M + 15) / 16));
  auto go = [&](auto kern, size_t lds, int waves) -> int {
    if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
    hipLaunchKernelGGL(kern, grid, dim3(waves * 64), lds, stream, p);
    return check_hip(hipGetLastError(), "gemm_1x16_rows16 launch");
  };
#define AQLM_R16_CASE(NBT_, CPB_) return go(gemm_1x16_rows16_kernel<T, G, NBT_, CPB_>, R16Lds<NBT_, CPB_>::TOTAL, R16Lds<NBT_, CPB_>::WAVES)
  if (r.cpb == 1) {
    switch (r.nbt) {
      case 1: AQLM_R16_CASE(1, 1);
      case 2: AQLM_R16_CASE(2, 1);
      case 4: AQLM_R16_CASE(4, 1);
      default: AQLM_R16_CASE(8, 1);
    }
  }
  switch (r.nbt) {
    case 1: AQLM_R16_CASE(1, 4);
    case 2: AQLM_R16_CASE(2, 4);
    case 4: AQLM_R16_CASE(4, 2);
    default: AQLM_R16_CASE(8, 2);
  }
#undef AQLM_R16_CASE
}



// ---------------------------------------------------------------------------------------------------------------------------
// K x 8-bit schemes, large batch (round 4): Y = X W^T with W never materialised.  Replaces (behaviour) Code2x8Dequant /
// CodeKx8Dequant + F::linear + the epilogue launches of code2x8_matmat_dequant / code1x8_matmat_dequant
// (cuda_kernel.cpp:450-484, 615-649).  Unlike 1x16 there is no gather floor: the codebooks (K x 4 KiB) live in LDS, so a weight
// vector costs an LDS read, and 2-bit codes + the streamed X are all that moves: the op is FASTER than a dense fp16 GEMM at every
// batch size it takes.  Same frame as the 16-row kernel above (a block owns 16 output rows over all of K; X by LDS-DMA into a
// ring of steps; one barrier per step), but the four compute waves split K, not the batch: within a step of 2 CPB k-steps (32 k
// each) wave c takes k-steps c, c + 4, ...; for each it reads its lane's code (row = lane % 16, group = 4 t + lane / 16: K bytes,
// prefetched 3 steps ahead in registers), gathers the K codebook vectors from LDS -- each IS a lane of a 16 x 32 MFMA A fragment
// -- and multiplies each with every batch tile of the step's X: the K terms of a weight meet in the fp32 accumulator (K MFMAs per
// tile; the matrix cores have room), so W is never rounded -- exact fp16 / bf16 products summed in fp32, like the matvec kernels.  The four partial accumulators meet in LDS at the end in wave order (deterministic), then
// scale + bias + one rounding.
constexpr int KX_NC = 4;   // compute waves
constexpr int KX_PD = 3;   // steps the codes are requested ahead

template <int K, int NBT, int CPB, int RT>
struct KxLds {
  static constexpr int NXP = 2 * NBT >= 4 ? 4 : 2 * NBT;            // X producer waves
  static constexpr int PXW = CPB * 2 * NBT / NXP;                   // 1-KiB pieces (8 batch rows x 64 k) per X wave and step
  static constexpr uint32_t X_CHUNK = (uint32_t)NBT * 2048u;
  static constexpr uint32_t X_STAGE = CPB * X_CHUNK;
  static constexpr int NSX = CPB == 4 ? (NBT <= 1 ? 4 : 3) : (NBT <= 4 ? 4 : 3);
  static constexpr uint32_t CB = 0;                                 // [K][256][16 B]
  static constexpr uint32_t X = (uint32_t)K * 4096u;
  static constexpr uint32_t TOTAL = X + NSX * X_STAGE;
  static constexpr uint32_t RED = X;                                // [KX_NC][RT][NBT][1 KiB] fp32 partial tiles, after the last step
  static constexpr int WAVES = NXP + KX_NC;
  static_assert(KX_NC * RT * NBT * 1024u <= NSX * X_STAGE, "the reduction reuses the X ring");
  static_assert(TOTAL <= 160u * 1024u, "LDS");
};

struct KxParams {
  const uint8_t* codes;      // [M][in_groups][K] u8
  const uint8_t* codebooks;  // [K][256][8] halfs
  const uint16_t* X;         // [B][xs]
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* Y;
  long xs, ys;
  int M, B, in_groups, nsteps;  // nsteps = steps of 64 CPB features of ONE K slice
  // K split (round 5): block = (row block blockIdx.x % row_blocks, K slice blockIdx.x / row_blocks); with ksplit > 1 the block leaves
  // its fp32 tile in partial[ks][b][m] and gemm_glds_finalize_kernel sums the slices, scales, adds the bias and rounds once
  float* partial;
  int ksplit, row_blocks;
};

// RT = 16-row tiles per block: every block streams all of X through its L1, so tall layers (>= 8192 rows: still >= 256 blocks)
// take two tiles per block -- half the X traffic, every X fragment feeds two MFMAs.
template <class T, int K, int NBT, int CPB, int RT>
__global__ __launch_bounds__((KxLds<K, NBT, CPB, RT>::WAVES * 64)) void gemm_kx8_rows16_kernel(const KxParams p) {
  using LDS = KxLds<K, NBT, CPB, RT>;
  constexpr int NSX = LDS::NSX;
  extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)glds_smem != 0u) __builtin_trap();  // LDS map above starts at 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int arow = lane & 15, kg = lane >> 4;
  const int ks = p.ksplit > 1 ? (int)blockIdx.x / p.row_blocks : 0;
  const int row0 = (p.ksplit > 1 ? (int)blockIdx.x - ks * p.row_blocks : (int)blockIdx.x) * (16 * RT);
  const int n = p.nsteps;  // >= NSX - 1 (host)
  const size_t s0 = (size_t)ks * (size_t)n;  // first step of this block's K slice

  if (wave < LDS::NXP) {
    // ============================================== X producer =======================================================
    constexpr int PX = LDS::PXW;
    constexpr int PPC = 2 * NBT;  // pieces per chunk
    static_assert((NSX - 2) * PX < 64, "vmcnt is 6 bits");
    const uint8_t* x_src[PX];
    uint32_t x_dst[PX];
#pragma unroll
    for (int x = 0; x < PX; ++x) {
      const int piece = wave * PX + x;
      const int u = piece / PPC, pc = piece % PPC;
      const int s = pc * 64 + lane;  // slot of the chunk image: batch row s >> 3, k piece (s & 7) ^ swizzle
      int b = s >> 3;
      const int c = (s & 7) ^ ((b >> 1) & 7);
      b = b < p.B ? b : p.B - 1;
      x_src[x] = (const uint8_t*)(p.X + (size_t)b * p.xs + c * 8) + (size_t)u * 128;
      x_dst[x] = LDS::X + (uint32_t)u * LDS::X_CHUNK + (uint32_t)pc * 1024u;
    }
    auto dma_x = [&](int step, int stage) {
      const int cc = step < n ? step : n - 1;
#pragma unroll
      for (int x = 0; x < PX; ++x)
        __builtin_amdgcn_global_load_lds((ggbl_void_ptr)(x_src[x] + (s0 + (size_t)cc) * (128 * CPB)),
                                         (glds_void_ptr)(size_t)(x_dst[x] + (uint32_t)stage * LDS::X_STAGE), 16, 0, 0);
    };
    for (int q = 0; q < NSX - 1; ++q) dma_x(q, q);
    __builtin_amdgcn_s_barrier();  // the codebooks are in LDS (compute waves)
    int stage = NSX - 1;
    int i = 0;
    for (; i + NSX - 1 < n; ++i) {
      __builtin_amdgcn_s_waitcnt(gl_vmcnt((NSX - 2) * PX));
      __builtin_amdgcn_s_barrier();
      dma_x(i + NSX - 1, stage);
      stage = stage == NSX - 1 ? 0 : stage + 1;
    }
    r16_drain<NSX - 2, PX>();
    return;
  }

  // ================================================ compute =============================================================
  const int cw = wave - LDS::NXP;
  constexpr int KS = 2 * CPB / KX_NC;  // k-steps of a step per wave: k-step t = cw + 4 j
  static_assert(KS >= 1 && KS * KX_NC == 2 * CPB, "k-steps deal evenly");
  // codebooks -> LDS (the compute waves only: the X waves are already streaming)
  for (int i = tid - LDS::NXP * 64; i < K * 256; i += KX_NC * 64)
    *reinterpret_cast<u32x4*>(glds_smem + LDS::CB + (uint32_t)i * 16u) = reinterpret_cast<const u32x4*>(p.codebooks)[i];
  // this lane's codes: row (tile rt) arow, group (step * 8 CPB) + 4 t + kg -> K bytes; requested KX_PD steps ahead, static ring slots
  const uint8_t* code_base[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int r = row0 + rt * 16 + arow;
    r = r < p.M ? r : p.M - 1;
    code_base[rt] = p.codes + ((size_t)r * p.in_groups + kg) * K;
  }
  auto load_codes = [&](int step, uint32_t (&c)[RT][KS]) {  // unconditional; steps past the end re-read the last one
    const int cc = step < n ? step : n - 1;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const uint8_t* src = code_base[rt] + ((s0 + (size_t)cc) * (8 * CPB) + 4 * (cw + KX_NC * j)) * K;
        if constexpr (K == 2) c[rt][j] = *reinterpret_cast<const uint16_t*>(src);
        else c[rt][j] = *src;
      }
  };
  uint32_t cring[KX_PD][RT][KS];
#pragma unroll
  for (int q = 0; q < KX_PD; ++q) load_codes(q, cring[q]);
  f32x4 acc[RT][NBT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < NBT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // codebooks visible to every compute wave

  int sx = 0;
  auto do_step = [&](const uint32_t (&c)[RT][KS]) {
    __builtin_amdgcn_s_barrier();  // the step's X has landed; everybody is done with the stage that is refilled next
    const uint32_t xbase = LDS::X + (uint32_t)sx * LDS::X_STAGE;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int t = cw + KX_NC * j;  // k-step of the step: chunk t / 2, half t % 2
      u32x4 w[RT][K];  // one A fragment lane per codebook: the K terms are accumulated by K MFMAs, never summed in the storage type
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int k = 0; k < K; ++k)
          w[rt][k] = *(glds_u32x4_ptr)(size_t)(LDS::CB + (uint32_t)k * 4096u + ((c[rt][j] >> (8 * k)) & 0xffu) * 16u);
      u32x4 b[NBT];
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt)
        b[bt] = *(glds_u32x4_ptr)(size_t)(xbase + (uint32_t)(t >> 1) * LDS::X_CHUNK + (uint32_t)xswz(bt * 16 + arow, (t & 1) * 4 + kg) * 16u);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int bt = 0; bt < NBT; ++bt) acc[rt][bt] = mfma16<T>(w[rt][k], b[bt], acc[rt][bt]);
    }
    sx = sx == NSX - 1 ? 0 : sx + 1;
  };
  int i = 0;
  for (; i + KX_PD <= n; i += KX_PD) {
#pragma unroll
    for (int q = 0; q < KX_PD; ++q) {  // static ring slots: the refill lands in the registers just consumed
      do_step(cring[q]);
      load_codes(i + KX_PD + q, cring[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < KX_PD - 1; ++q)
    if (i + q < n) do_step(cring[q]);

  // ---- the four K shares meet in LDS (the X ring is free once every compute wave is past its last step) ----------------
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt)
      *reinterpret_cast<f32x4*>(glds_smem + LDS::RED + (uint32_t)((cw * RT + rt) * NBT + bt) * 1024u + (uint32_t)lane * 16u) = acc[rt][bt];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // wave cw finishes tiles cw, cw + 4, ... of the RT x NBT output tiles: lane (arow, kg) holds rows 4 kg .. 4 kg + 3, batch column arow
  const bool vec = (p.M & 3) == 0 && (p.ys & 3) == 0 && ((uintptr_t)p.Y & 7u) == 0;  // (Y only 2- / 4-byte aligned: scalar stores)
  for (int tile = cw; tile < RT * NBT; tile += KX_NC) {
    const int rt = tile / NBT, bt = tile - rt * NBT;
    f32x4 v = *reinterpret_cast<const f32x4*>(glds_smem + LDS::RED + (uint32_t)tile * 1024u + (uint32_t)lane * 16u);
#pragma unroll
    for (int w = 1; w < KX_NC; ++w) {  // wave order: the result does not depend on who finishes
      const f32x4 o = *reinterpret_cast<const f32x4*>(glds_smem + LDS::RED + (uint32_t)(w * RT * NBT + tile) * 1024u + (uint32_t)lane * 16u);
      v = v + o;
    }
    const int m = row0 + rt * 16 + kg * 4;
    const int b = bt * 16 + arow;
    if (b < p.B && m < p.M) {
      if (p.ksplit > 1) {  // this K slice's share of the tile: fp32, summed by the finalize kernel in slice order
        float* dst = p.partial + ((size_t)ks * p.B + b) * p.M + m;
        if ((p.M & 3) == 0) *reinterpret_cast<f32x4*>(dst) = v;
        else
          for (int r = 0; r < 4; ++r)
            if (m + r < p.M) dst[r] = v[r];
        continue;
      }
      uint16_t* dst = p.Y + (size_t)b * p.ys + m;
      uint16_t h[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = m + r < p.M ? m + r : p.M - 1;
        const float sc = T::to_float(p.scales[mm]), bi = p.bias ? T::to_float(p.bias[mm]) : 0.f;
        h[r] = T::from_float(__builtin_fmaf(v[r], sc, bi));
      }
      if (vec) *reinterpret_cast<u32x2*>(dst) = u32x2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
      else
        for (int r = 0; r < 4; ++r)
          if (m + r < p.M) dst[r] = h[r];
    }
  }
}

struct KxPlan {
  int nbt, cpb, rt, nsteps, ksplit;  // nsteps: per K slice
};

constexpr int KX_MAX_KSPLIT = 4;

// `can_split`: the caller gave a workspace for fp32 partials (aqlm_hip_gemm_kx8_mfma_ws).
static bool plan_kx8(int B, int M, int Kf, KxPlan& r, bool can_split = false) {
  if (Kf % (2 * BK) != 0 || B < 1 || B > 128) return false;
  const int t = (B + 15) / 16;
  r.nbt = t <= 1 ? 1 : (t <= 2 ? 2 : (t <= 4 ? 4 : 8));
  const int chunks = Kf / BK;
  // (steps of 4 chunks at 64 / 128 rows -- X ring of 3 x 32 / 2 x 64 KiB -- measured in round 5: 4096^2 at 64 rows 13.4 -> 12.0 us, every other
  // shape and 128 rows equal or up to 16 % slower, profiles/r05_gemm_kx8_steps_of_4_chunks.log: not kept)
  r.cpb = (r.nbt <= 2 && chunks % 4 == 0) ? 4 : 2;
  r.nsteps = chunks / r.cpb;
  r.ksplit = 1;
  // two row tiles per block where that still fills the chip AND X is what the block mostly moves (measured, 4096 -> 11008: 128 rows
  // 46.5 -> 39.6 us, 64 rows 28.5 -> 26.9, but 16 rows 18.2 -> 20.8: 344 blocks are 1.3 rounds of the chip)
  r.rt = ((M + 31) / 32 >= 256 && r.nbt >= 4) ? 2 : 1;
  if (r.nsteps < 3) return false;  // the X ring's prologue (NSX <= 4)
  // K split (round 5, given a workspace): the K range dealt to 2 / 4 blocks whose fp32 tiles a finalize launch sums (+ ~3 us).  Measured
  // (profiles/r05_gemm_kx8_ksplit.log, 2x8 g8, 128 rows, us; no split -> split): it pays (a) where the layer has too few rows to fill the
  // chip -- 4096 -> 1024: 64 blocks, 15.2 -> 11.0 with 4 slices of one tile -- and (b) on long K with two tiles per block, where a step's
  // X fragments feed twice the MFMAs -- 11008 -> 4096: 45.7 -> 35.0 (64 rows: 37.5 -> 24.7); it does NOT pay at 4096 x 4096 (16.6 -> 18.0:
  // a 32-step block is not X-bound, the finalize is pure cost) nor where two tiles per block already fill the chip (8192 x 8192).
  const int force_ks = tuning().kx8_ksplit, force_rt = tuning().kx8_rt;
  if (can_split && r.nbt >= 4 && force_ks != 1) {
    const int rb16 = (M + 15) / 16, rb32 = (M + 31) / 32;
    int rt = 0, ks = 1;
    if (force_ks > 1) {
      rt = force_rt == 1 ? 1 : 2;
      ks = force_ks;
    } else if (rb16 <= 128) {
      rt = 1;
      ks = std::min(KX_MAX_KSPLIT, 256 / rb16);
    } else if (Kf >= 8192 && rb32 < 256) {
      rt = 2;
      ks = std::min(KX_MAX_KSPLIT, std::max(1, 256 / rb32));
    }
    while (ks > 1 && (r.nsteps % ks != 0 || r.nsteps / ks < 3)) --ks;
    if (ks > 1) {
      r.rt = rt;
      r.ksplit = ks;
      r.nsteps /= ks;
    }
  } else if (force_rt == 1 || force_rt == 2) {
    r.rt = force_rt;
  }
  return true;
}

template <class T, int K>
static int launch_kx8(const KxParams& p, const KxPlan& r, hipStream_t stream) {
  const dim3 grid((unsigned)(((p.M + 16 * r.rt - 1) / (16 * r.rt)) * r.ksplit));
  auto go = [&](auto kern, size_t lds, int waves) -> int {
    if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
    hipLaunchKernelGGL(kern, grid, dim3(waves * 64), lds, stream, p);
    return check_hip(hipGetLastError(), "gemm_kx8_rows16 launch");
  };
#define AQLM_KX_CASE(NBT_, CPB_, RT_) return go(gemm_kx8_rows16_kernel<T, K, NBT_, CPB_, RT_>, KxLds<K, NBT_, CPB_, RT_>::TOTAL, KxLds<K, NBT_, CPB_, RT_>::WAVES)
#define AQLM_KX_RT(NBT_, CPB_)        \
  do {                                \
    if (r.rt == 2) AQLM_KX_CASE(NBT_, CPB_, 2); \
    AQLM_KX_CASE(NBT_, CPB_, 1);      \
  } while (0)
  if (r.cpb == 4) {
    if (r.nbt == 1) AQLM_KX_RT(1, 4);
    AQLM_KX_RT(2, 4);
  }
  switch (r.nbt) {
    case 1: AQLM_KX_RT(1, 2);
    case 2: AQLM_KX_RT(2, 2);
    case 4: AQLM_KX_RT(4, 2);
    default: AQLM_KX_RT(8, 2);
  }
#undef AQLM_KX_RT
#undef AQLM_KX_CASE
}

struct GemmPlan {
  int ksplit, kslice, Bpad, nbt;
};

static GemmPlan plan_gemm(int B, int M, int K) {
  GemmPlan g;
  g.nbt = (std::min(B, 128) + 31) / 32;
  g.Bpad = g.nbt * 32;
  const int row_blocks = (M + 127) / 128;
  const int kchunks = (K + BK - 1) / BK;
  int ksplit = std::max(1, 256 / row_blocks);
  ksplit = std::min(ksplit, kchunks);
  const int chunks_per = (kchunks + ksplit - 1) / ksplit;
  g.kslice = chunks_per * BK;
  g.ksplit = (kchunks + chunks_per - 1) / chunks_per;
  return g;
}

template <class T, int G>
static int launch_gemm(const GemmParams& p, int nbt, hipStream_t stream) {
  const int blocks = ((p.M + 127) / 128) * p.ksplit;
  switch (nbt) {
    case 1: hipLaunchKernelGGL((gemm_1x16_mfma_kernel<T, G, 1>), dim3(blocks), dim3(256), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((gemm_1x16_mfma_kernel<T, G, 2>), dim3(blocks), dim3(256), 0, stream, p); break;
    case 3: hipLaunchKernelGGL((gemm_1x16_mfma_kernel<T, G, 3>), dim3(blocks), dim3(256), 0, stream, p); break;
    default: hipLaunchKernelGGL((gemm_1x16_mfma_kernel<T, G, 4>), dim3(blocks), dim3(256), 0, stream, p); break;
  }
  return check_hip(hipGetLastError(), "gemm_1x16_mfma launch");
}

}  // namespace aqlm

using namespace aqlm;

namespace aqlm {
size_t gemv_8x8_lut_workspace(int out_features, int in_features, int in_group_size);
}

extern "C" size_t aqlm_hip_workspace_bytes(int op, int batch, int out_features, int in_features) {
  if (op == AQLM_HIP_OP_GEMV_8X8_LUT && batch > 0 && out_features > 0 && in_features > 0 && in_features % batch == 0)
    return aqlm::gemv_8x8_lut_workspace(out_features, in_features, /*in_group_size=*/batch);
  if (batch <= 0 || out_features <= 0 || in_features <= 0) return 0;
  if (op == AQLM_HIP_OP_GEMV_1X16_PACKED || op == AQLM_HIP_OP_GEMV_1X16_G16_PACKED)  // fp32 partials [slices][rows of x][out]
    return (size_t)(op == AQLM_HIP_OP_GEMV_1X16_PACKED ? 16 : 32) * std::min(batch, AQLM_HIP_MAX_GEMV_BATCH) * out_features * sizeof(float);
  if (op == AQLM_HIP_OP_GEMM_KX8_MFMA)  // fp32 partials of the K-split form (64+ rows); 0 below: the op then needs no workspace
    return batch >= 49 ? (size_t)KX_MAX_KSPLIT * std::min(batch, 128) * out_features * sizeof(float) : 0;
  if (op != AQLM_HIP_OP_GEMM_1X16_MFMA) return 0;
  // covers both kernels of the op (the LDS-DMA pipeline and the register-staged one behind the `gemm_variant` knob)
  const GemmPlan g = plan_gemm(batch, out_features, in_features);
  size_t need = (size_t)g.ksplit * out_features * g.Bpad * sizeof(float);
  GldsPlan q;
  if (plan_glds(std::min(batch, 128), out_features, in_features, q) && q.ksplit > 1)
    need = std::max(need, (size_t)q.ksplit * std::min(batch, 128) * out_features * sizeof(float));
  need = std::max(need, scan::workspace_bytes(batch, out_features, in_features));  // the slice-scan kernel's planes (gemm_variant 0 / 4)
  return need;
}

// ---------------------------------------------------------------------------------------------------------------------------
// K x 8-bit schemes at <= 16 batch rows (round 5): X RESIDENT in LDS.  The 16-row kernel above streams all of X through every
// block's L1 and LDS (128 KiB per 16 output rows at 4096 features) behind one barrier per step; traced at 16 rows it spent its
// time in that chain (8.1 us for ~1 us of bytes on 4096 x 4096, 18 us on 4096 -> 11008: 688 blocks x 128 KiB of X).  For the
// batches that matter most here -- the 3..8 rows of a decode call, a 16-row speculative verify -- X is SMALL: B rows x K x 2 bytes
// (<= 128 KiB) fit the LDS next to the codebooks.  So a workgroup loads X ONCE (LDS-DMA, all 8 waves), then walks its 16-row
// output tiles (tile = blockIdx.x, += gridDim.x) with NOTHING left to synchronise inside a tile: the 8 waves split K, each runs
// code (8 B per lane: its row's 4 consecutive groups = 4 k-steps) -> 2 K codebook gathers + 1 X fragment per k-step -> MFMA chains
// on two accumulators, the partial tiles meet in LDS (double-buffered: one barrier per tile) and wave 0 scales, adds the bias,
// rounds once and stores while the others are already in the next tile.  The k-step -> group mapping is the lane's own
// (k-step j of quad q = groups 16 q + 4 kg + j for lane piece kg): A and B fragments only have to agree on it.
// X image: [64-k chunk][row b < B][8 pieces of 16 B], piece index XOR-ed with s(b) = ((b >> 1) & 7) ^ ((b & 8) >> 1): the 16 lanes
// of every LDS service group read 16 distinct bank groups (rows >= B read row B - 1: same address, broadcast).
// Numerics: exact products, fp32 sums (per wave over its quads in order, then the 8 waves in wave order): a row's bits do not
// depend on the other rows of the call, nor on B.
constexpr int KR_NW = 8;
constexpr int KR_MAX_SEG = AQLM_HIP_MAX_SEGMENTS;

// One layer, or (round 5) up to KR_MAX_SEG layers that multiply the SAME X (q / k / v, gate / up: aqlm_hip_gemv_kx8_multi): the X
// image is loaded once, every layer's codebooks (K x 4 KiB each) sit side by side in LDS, and the tile walk runs over the layers'
// tiles back to back (tile -> layer by three comparisons).  A tile's arithmetic does not know about the other layers: outputs are
// bit-identical to separate launches.
struct KrSeg {
  const uint8_t* codes;      // [M][in_groups][K] u8
  const uint8_t* codebooks;  // [K][256][8] halfs
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* Y;
  long ys;
  int M;
  int tile0;                 // first tile of the layer in the launch's tile sequence
};

template <int NSEG>
struct KrParams {
  KrSeg seg[NSEG];
  const uint16_t* X;         // [B][xs]
  long xs;
  int B, in_groups, ntiles, nseg;
};

template <int K>
struct KrLds {  // sized by the number of layers of the launch: two layers leave room for a second workgroup per CU where four would not
  static constexpr uint32_t CB = 0;                                                        // [layer][K][256][16 B]
  static constexpr uint32_t red(int nseg) { return (uint32_t)nseg * K * 4096u; }          // [2][KR_NW][64 lanes][16 B] fp32 partial tiles
  static constexpr uint32_t x(int nseg) { return red(nseg) + 2u * KR_NW * 1024u; }        // the X image, rounded up to whole KiB (DMA granularity)
  static size_t total(int B, int in_features, int nseg) { return (size_t)x(nseg) + (((size_t)B * in_features * 2 + 1023) & ~(size_t)1023); }
};

__device__ __forceinline__ uint32_t kr_swz(uint32_t b) { return ((b >> 1) & 7u) ^ ((b & 8u) >> 1); }

template <class T, int K, int NSEG>
__global__ __launch_bounds__(KR_NW * 64) void gemm_kx8_xres_kernel(const KrParams<NSEG> p) {
  using LDS = KrLds<K>;
  extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)glds_smem != 0u) __builtin_trap();  // LDS map above starts at 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int arow = lane & 15, kg = lane >> 4;
  const uint32_t B = (uint32_t)p.B;
  const int nquads = p.in_groups >> 4;  // 128 k each (host: in_features % 128 == 0)
  const uint32_t lds_red = LDS::red(NSEG == 1 ? 1 : p.nseg), lds_x = LDS::x(NSEG == 1 ? 1 : p.nseg);
  auto seg_of = [&](int tile) -> int {  // wave-uniform
    if constexpr (NSEG == 1) return 0;
    else return (int)(tile >= p.seg[1].tile0) + (int)(tile >= p.seg[2].tile0) + (int)(tile >= p.seg[3].tile0);  // (absent layers: tile0 = ntiles)
  };

  // ---- prologue: codebooks and the whole of X by LDS-DMA; the first tile's codes are requested before anything is waited for
  for (int i = wave; i < K * 4 * (NSEG == 1 ? 1 : p.nseg); i += KR_NW) {
    const int sg = i / (K * 4), off = i - sg * (K * 4);
    __builtin_amdgcn_global_load_lds((ggbl_void_ptr)(p.seg[sg].codebooks + (size_t)off * 1024 + lane * 16),
                                     (glds_void_ptr)(size_t)(LDS::CB + (uint32_t)i * 1024u), 16, 0, 0);
  }
  {
    // piece q = (chunk c, row b, 16-byte piece sl) = (q / ppc, (q % ppc) / 8, q % 8).  ONE division per lane; after that (c, r) are
    // stepped by the quotient and remainder of the 512 pieces all waves advance together, and the source is a 32-bit byte offset
    // (host: 16 rows x the row stride fit): the first cut recomputed q / ppc and a 64-bit b * xs per KiB -- ~33 VALU instructions
    // with quarter-rate multiplies, 16 times per wave at 16 rows, which the fill had to wait for (counters: VALU 2.2 us per SIMD)
    const uint32_t ppc = B * 8u;                                  // pieces per chunk
    const uint32_t npieces = (uint32_t)(p.in_groups >> 3) * ppc;  // chunks x rows x 8
    const uint32_t dq = (KR_NW * 64u) / ppc, dr = (KR_NW * 64u) - dq * ppc;
    const uint32_t xs2 = (uint32_t)p.xs * 2u;                     // bytes per row of X (< 2^24: host)
    uint32_t q = (uint32_t)wave * 64u + (uint32_t)lane;
    uint32_t c = q / ppc, r = q - c * ppc;
    const uint8_t* xb = (const uint8_t*)p.X;
    for (uint32_t q0 = (uint32_t)wave * 64u; q0 < npieces; q0 += KR_NW * 64u) {
      const bool in = q < npieces;  // (the last KiB may be partly padding: any valid source will do)
      const uint32_t cc = in ? c : 0u, rr = in ? r : 0u;
      const uint32_t b = rr >> 3, sl = rr & 7u;
      const uint32_t off = (b & 0xffu) * (xs2 & 0xffffffu) + cc * 128u + ((sl ^ kr_swz(b)) << 4);
      __builtin_amdgcn_global_load_lds((ggbl_void_ptr)(xb + off), (glds_void_ptr)(size_t)(lds_x + q0 * 16u), 16, 0, 0);
      q += KR_NW * 64u;
      c += dq;
      r += dr;
      if (r >= ppc) {
        r -= ppc;
        c += 1u;
      }
    }
  }
  const uint32_t brow = (uint32_t)arow < B ? (uint32_t)arow : B - 1u;  // batch column of this lane's B fragments
  const uint32_t bsw = kr_swz(brow);
  typedef typename std::conditional<K == 2, u32x2, uint32_t>::type code_t;
  auto code_ptr = [&](int tile, int quad) -> const code_t* {
    const int sg = seg_of(tile);
    int r = (tile - p.seg[sg].tile0) * 16 + arow;
    r = r < p.seg[sg].M ? r : p.seg[sg].M - 1;
    return reinterpret_cast<const code_t*>(p.seg[sg].codes + ((size_t)r * p.in_groups + (size_t)quad * 16 + (size_t)kg * 4) * K);
  };
  int tile = (int)blockIdx.x;
  code_t cnext{};
  if (tile < p.ntiles && wave < nquads) cnext = *code_ptr(tile, wave);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // X and the codebooks are in LDS, for good

  int buf = 0;
  for (; tile < p.ntiles; tile += (int)gridDim.x) {
    const int sg = seg_of(tile);
    const uint32_t cb = LDS::CB + (uint32_t)sg * (uint32_t)(K * 4096);  // this layer's codebooks
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int quad = wave; quad < nquads; quad += KR_NW) {
      const code_t cw = cnext;
      // the next code word of this wave: the next quad of the tile, else the first quad of its next tile
      {
        int nq = quad + KR_NW, nt = tile;
        if (nq >= nquads) { nq = wave; nt = tile + (int)gridDim.x; }
        if (nt < p.ntiles && nq < nquads) cnext = *code_ptr(nt, nq);
      }
      uint32_t cwords[2];
      if constexpr (K == 2) { cwords[0] = cw.x; cwords[1] = cw.y; } else { cwords[0] = cw; cwords[1] = 0u; }
      u32x4 w[4][K], xb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const uint32_t byte = K == 2 ? (cwords[j >> 1] >> (16 * (j & 1) + 8 * k)) & 0xffu : (cwords[0] >> (8 * j)) & 0xffu;
          w[j][k] = *(glds_u32x4_ptr)(size_t)(cb + (uint32_t)k * 4096u + byte * 16u);
        }
        const uint32_t c = (uint32_t)quad * 2u + (uint32_t)(kg >> 1), pc = (uint32_t)(kg & 1) * 4u + (uint32_t)j;
        xb[j] = *(glds_u32x4_ptr)(size_t)(lds_x + ((c * B + brow) * 8u + (pc ^ bsw)) * 16u);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[(j * K + k) & 1] = mfma16<T>(w[j][k], xb[j], acc[(j * K + k) & 1]);
    }
    // ---- the eight K shares meet in LDS; wave 0 finishes the tile while the others start the next one
    const f32x4 mine = acc[0] + acc[1];
    *reinterpret_cast<f32x4*>(glds_smem + lds_red + (uint32_t)(buf * KR_NW + wave) * 1024u + (uint32_t)lane * 16u) = mine;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave == 0) {
      f32x4 v = mine;
#pragma unroll
      for (int w8 = 1; w8 < KR_NW; ++w8)  // wave order: the result does not depend on who finishes first
        v = v + *reinterpret_cast<const f32x4*>(glds_smem + lds_red + (uint32_t)(buf * KR_NW + w8) * 1024u + (uint32_t)lane * 16u);
      const KrSeg& S = p.seg[sg];
      const int m = (tile - S.tile0) * 16 + kg * 4;
      const int b = arow;
      if (b < p.B && m < S.M) {
        uint16_t* dst = S.Y + (size_t)b * S.ys + m;
        uint16_t h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mm = m + r < S.M ? m + r : S.M - 1;
          const float sc = T::to_float(S.scales[mm]), bi = S.bias ? T::to_float(S.bias[mm]) : 0.f;
          h[r] = T::from_float(__builtin_fmaf(v[r], sc, bi));
        }
        if ((S.M & 3) == 0 && (S.ys & 3) == 0 && ((uintptr_t)dst & 7u) == 0) *reinterpret_cast<u32x2*>(dst) = u32x2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
        else
          for (int r = 0; r < 4; ++r)
            if (m + r < S.M) dst[r] = h[r];
      }
    }
    buf ^= 1;
  }
}

// does the X-resident kernel take the call?  <= 16 rows whose image fits the LDS next to the codebooks and the partial tiles
template <int K>
static bool xres_fits(int B, int in_features, long xs, int nseg = 1) {
  return B >= 1 && B <= 16 && in_features % 128 == 0 && KrLds<K>::total(B, in_features, nseg) <= 160u * 1024u && xs > 0 && xs < (1l << 22);
}

template <class T, int K, int NSEG>
static int launch_kx8_xres(const KrParams<NSEG>& p, int in_features, hipStream_t stream) {
  auto kern = gemm_kx8_xres_kernel<T, K, NSEG>;
  const size_t lds = KrLds<K>::total(p.B, in_features, p.nseg);
  if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  // one workgroup per CU and as many as fit its LDS (small X: two or more share a CU and overlap their latencies)
  static const int cus = [] {  // (initialised once, thread-safe; every GPU of a node is the same part)
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int per_cu = std::max(1, std::min(4, (int)((160u * 1024u) / lds)));
  const int grid = std::min(p.ntiles, cus * per_cu);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(KR_NW * 64), lds, stream, p);
  return check_hip(hipGetLastError(), "gemm_kx8_xres launch");
}

// ---------------------------------------------------------------------------------------------------------------------------
// X-resident kernel in PHASES (round 5): down-projection-shaped layers (11008 / 14336 / 28672 input features) at 7..16 rows have an X
// image of 150-900 KiB -- it does not fit the LDS, and the call fell to the streaming 16-row kernel (16.7 us for 8 rows of
// 11008 -> 4096 where 6 rows cost 10.6).  Here the image is loaded in nphases pieces of `qpp` quads (128 features each): all waves
// walk the block's one to three tiles over the quads of the piece that is resident, the fp32 accumulators of the tiles stay in
// registers across the phases (TPB x 2 x f32x4 per wave), and the K shares meet in LDS once, at the end.  The fill of a phase is
// not overlapped with the arithmetic of the previous one (one image buffer) -- it costs what the bytes cost (X once per
// workgroup) instead of the streaming kernel's step chain.  Same arithmetic per (row, quad) as gemm_kx8_xres_kernel and the
// same summation order (per wave over its quads in ascending order, then the waves in wave order): the results are bit-identical
// to the single-phase kernel's, so a row's bits still depend neither on the other rows nor on their number.
struct KpParams {
  const uint8_t* codes;      // [M][in_groups][K] u8
  const uint8_t* codebooks;  // [K][256][8] halfs
  const uint16_t* X;         // [B][xs]
  const uint16_t* scales;
  const uint16_t* bias;
  uint16_t* Y;
  long xs, ys;
  int M, B, in_groups, ntiles, nphases, qpp;
};

template <int K, int TPB, int NBT>
struct KpLds {
  static constexpr uint32_t CB = 0;                                        // [K][256][16 B]
  static constexpr uint32_t RED = (uint32_t)K * 4096u;                     // [KR_NW][TPB][NBT][64 lanes][16 B] fp32 partial tiles
  static constexpr uint32_t X = RED + (uint32_t)KR_NW * TPB * NBT * 1024u; // the image of one phase
  static size_t image_bytes(int B, int qpp) { return (size_t)B * qpp * 256; }
};

// NBT = 16-column batch tiles (17 .. 32 rows: two; the A fragments of a k-step feed both)
template <class T, int K, int TPB, int NBT>
__global__ __launch_bounds__(KR_NW * 64) void gemm_kx8_xres_phased_kernel(const KpParams p) {
  using LDS = KpLds<K, TPB, NBT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];
  if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)glds_smem != 0u) __builtin_trap();  // LDS map above starts at 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int arow = lane & 15, kg = lane >> 4;
  const uint32_t B = (uint32_t)p.B;
  const int nquads = p.in_groups >> 4;
  const int grid = (int)gridDim.x;

  if (wave < K * 4)
    __builtin_amdgcn_global_load_lds((ggbl_void_ptr)(p.codebooks + (size_t)wave * 1024 + lane * 16),
                                     (glds_void_ptr)(size_t)(LDS::CB + (uint32_t)wave * 1024u), 16, 0, 0);
  uint32_t brow[NBT], bsw[NBT];  // batch column of this lane's B fragments, per batch tile
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) {
    brow[bt] = (uint32_t)(bt * 16 + arow) < B ? (uint32_t)(bt * 16 + arow) : B - 1u;
    bsw[bt] = kr_swz(brow[bt]);
  }
  typedef typename std::conditional<K == 2, u32x2, uint32_t>::type code_t;
  auto code_ptr = [&](int tile, int quad) -> const code_t* {
    int r = tile * 16 + arow;
    r = r < p.M ? r : p.M - 1;
    return reinterpret_cast<const code_t*>(p.codes + ((size_t)r * p.in_groups + (size_t)quad * 16 + (size_t)kg * 4) * K);
  };
  f32x4 acc[TPB][NBT][2];
#pragma unroll
  for (int ts = 0; ts < TPB; ++ts)
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) acc[ts][bt][0] = acc[ts][bt][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int ph = 0; ph < p.nphases; ++ph) {
    const int q_lo = ph * p.qpp, q_hi = q_lo + p.qpp < nquads ? q_lo + p.qpp : nquads;
    // wave w owns the quads = w (mod 8) whatever the phases are (the single-phase kernel's deal): the summation order of a row does
    // not depend on how the image was cut, i.e. not on the number of rows of the call
    const int q_first = q_lo + ((wave - q_lo) & (KR_NW - 1));
    // the first code word of the phase is requested before its image (it does not depend on it)
    code_t cnext{};
    if ((int)blockIdx.x < p.ntiles && q_first < q_hi) cnext = *code_ptr((int)blockIdx.x, q_first);
    if (ph > 0) __builtin_amdgcn_s_barrier();  // every wave is done with the previous image
    {
      // the image of quads [q_lo, q_hi): chunk c (64 features) of the piece, row b, 16-byte piece sl (stepped as in gemm_kx8_xres_kernel)
      const uint32_t ppc = B * 8u;
      const uint32_t npieces = (uint32_t)(q_hi - q_lo) * 2u * ppc;
      const uint32_t dq = (KR_NW * 64u) / ppc, dr = (KR_NW * 64u) - dq * ppc;
      const uint32_t xs2 = (uint32_t)p.xs * 2u;
      uint32_t q = (uint32_t)wave * 64u + (uint32_t)lane;
      uint32_t c = q / ppc, r = q - c * ppc;
      const uint8_t* xb = (const uint8_t*)p.X + (size_t)q_lo * 256;  // 128 features x 2 bytes per quad
      for (uint32_t q0 = (uint32_t)wave * 64u; q0 < npieces; q0 += KR_NW * 64u) {
        const bool in = q < npieces;
        const uint32_t cc = in ? c : 0u, rr = in ? r : 0u;
        const uint32_t b = rr >> 3, sl = rr & 7u;
        const uint32_t off = (b & 0xffu) * (xs2 & 0xffffffu) + cc * 128u + ((sl ^ kr_swz(b)) << 4);
        __builtin_amdgcn_global_load_lds((ggbl_void_ptr)(xb + off), (glds_void_ptr)(size_t)(LDS::X + q0 * 16u), 16, 0, 0);
        q += KR_NW * 64u;
        c += dq;
        r += dr;
        if (r >= ppc) {
          r -= ppc;
          c += 1u;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the image (and, in phase 0, the codebooks) are in LDS
#pragma unroll
    for (int ts = 0; ts < TPB; ++ts) {
      const int tile = (int)blockIdx.x + ts * grid;
      if (tile >= p.ntiles) break;
      for (int quad = q_first; quad < q_hi; quad += KR_NW) {
        const code_t cw = cnext;
        {  // the next code word of this wave: the next quad of the tile, else the first quad of the block's next tile in this phase
          int nq = quad + KR_NW, nt = tile;
          if (nq >= q_hi) { nq = q_first; nt = tile + grid; }
          if (nt < p.ntiles && nt < (int)blockIdx.x + TPB * grid && nq < q_hi) cnext = *code_ptr(nt, nq);
        }
        uint32_t cwords[2];
        if constexpr (K == 2) { cwords[0] = cw.x; cwords[1] = cw.y; } else { cwords[0] = cw; cwords[1] = 0u; }
        u32x4 w[4][K], xb[4][NBT];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const uint32_t byte = K == 2 ? (cwords[j >> 1] >> (16 * (j & 1) + 8 * k)) & 0xffu : (cwords[0] >> (8 * j)) & 0xffu;
            w[j][k] = *(glds_u32x4_ptr)(size_t)(LDS::CB + (uint32_t)k * 4096u + byte * 16u);
          }
          const uint32_t c = (uint32_t)(quad - q_lo) * 2u + (uint32_t)(kg >> 1), pc = (uint32_t)(kg & 1) * 4u + (uint32_t)j;
#pragma unroll
          for (int bt = 0; bt < NBT; ++bt)
            xb[j][bt] = *(glds_u32x4_ptr)(size_t)(LDS::X + ((c * B + brow[bt]) * 8u + (pc ^ bsw[bt])) * 16u);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt) acc[ts][bt][(j * K + k) & 1] = mfma16<T>(w[j][k], xb[j][bt], acc[ts][bt][(j * K + k) & 1]);
      }
    }
  }
  // ---- the eight K shares of every (row tile, batch tile) meet in LDS; wave ts NBT + bt finishes that pair
#pragma unroll
  for (int ts = 0; ts < TPB; ++ts)
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt)
      *reinterpret_cast<f32x4*>(glds_smem + LDS::RED + (uint32_t)(((wave * TPB + ts) * NBT + bt) * 1024) + (uint32_t)lane * 16u) = acc[ts][bt][0] + acc[ts][bt][1];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wave < TPB * NBT) {
    const int ts = wave / NBT, bt = wave - ts * NBT, tile = (int)blockIdx.x + ts * grid;
    if (tile < p.ntiles) {
      f32x4 v = *reinterpret_cast<const f32x4*>(glds_smem + LDS::RED + (uint32_t)((ts * NBT + bt) * 1024) + (uint32_t)lane * 16u);
#pragma unroll
      for (int w8 = 1; w8 < KR_NW; ++w8)  // wave order, as in the single-phase kernel
        v = v + *reinterpret_cast<const f32x4*>(glds_smem + LDS::RED + (uint32_t)(((w8 * TPB + ts) * NBT + bt) * 1024) + (uint32_t)lane * 16u);
      const int m = tile * 16 + kg * 4;
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
  }
}

static int device_cus() {
  static const int cus = [] {  // (initialised once, thread-safe; every GPU of a node is the same part)
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  return cus;
}

// plan + launch; AQLM_HIP_E_UNSUPPORTED when the layer has more tiles than four (17+ rows: three) per CU, or a phase would hold fewer than 8 quads
template <class T, int K>
static int launch_kx8_xres_phased(KpParams p, int in_features, hipStream_t stream) {
  const int cus = device_cus();
  int tpb = (p.ntiles + cus - 1) / cus;  // tiles per workgroup: 1 .. 4 (their accumulators live in registers across the phases)
  const int nbt = p.B <= 16 ? 1 : 2;
  const int exp_tpb = tuning().kx8_phase_tpb, exp_q = tuning().kx8_phase_quads;  // experiments: tiles per workgroup, quads per phase
  if (exp_tpb >= 1 && exp_tpb <= 4) tpb = exp_tpb;
  if (tpb > (nbt == 1 ? 4 : 3) || p.B > 32) return AQLM_HIP_E_UNSUPPORTED;
  const size_t fixed = (size_t)KpLds<K, 1, 1>::RED + (size_t)KR_NW * tpb * nbt * 1024u;
  const int nquads = in_features / 128;
  const int qmax = (int)((160u * 1024u - fixed) / ((size_t)p.B * 256));  // quads whose image fits
  if (qmax < KR_NW) return AQLM_HIP_E_UNSUPPORTED;
  const int qcap = exp_q >= KR_NW && exp_q < qmax ? exp_q : qmax;
  p.nphases = (nquads + qcap - 1) / qcap;
  p.qpp = (nquads + p.nphases - 1) / p.nphases;
  const size_t lds = fixed + (((size_t)p.B * p.qpp * 256 + 1023) & ~(size_t)1023);
  const int grid = (p.ntiles + tpb - 1) / tpb;
  auto go = [&](auto kern) -> int {
    if (int e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(KR_NW * 64), lds, stream, p);
    return check_hip(hipGetLastError(), "gemm_kx8_xres_phased launch");
  };
  if (nbt == 2) return tpb == 1 ? go(gemm_kx8_xres_phased_kernel<T, K, 1, 2>) : (tpb == 2 ? go(gemm_kx8_xres_phased_kernel<T, K, 2, 2>) : go(gemm_kx8_xres_phased_kernel<T, K, 3, 2>));
  switch (tpb) {
    case 1: return go(gemm_kx8_xres_phased_kernel<T, K, 1, 1>);
    case 2: return go(gemm_kx8_xres_phased_kernel<T, K, 2, 1>);
    case 3: return go(gemm_kx8_xres_phased_kernel<T, K, 3, 1>);
    default: return go(gemm_kx8_xres_phased_kernel<T, K, 4, 1>);
  }
}

namespace aqlm {
// Shared-input launches of the fused K x 8 op at <= 16 rows (called by aqlm_hip_gemv_kx8_multi; arguments validated there):
// AQLM_HIP_E_UNSUPPORTED when the layers' codebooks + X do not fit the LDS together (the caller launches the layers one by one).
int gemm_kx8_xres_multi(const aqlm_hip_segment* segments, int num_segments, const void* X, int in_features, int K, int batch, long xs,
                        int dtype, hipStream_t stream) {
  if (num_segments < 2 || num_segments > KR_MAX_SEG || !tuning().kx8_xres || (K != 1 && K != 2)) return AQLM_HIP_E_UNSUPPORTED;
  if (!(K == 2 ? xres_fits<2>(batch, in_features, xs, num_segments) : xres_fits<1>(batch, in_features, xs, num_segments)) || !aligned16(X) || xs % 8 != 0)
    return AQLM_HIP_E_UNSUPPORTED;
  KrParams<KR_MAX_SEG> p{};
  int tiles = 0;
  for (int k = 0; k < KR_MAX_SEG; ++k) {
    KrSeg& S = p.seg[k];
    if (k < num_segments) {
      const aqlm_hip_segment& sg = segments[k];
      if (!aligned16(sg.codebook)) return AQLM_HIP_E_UNSUPPORTED;
      S.codes = (const uint8_t*)sg.codes;
      S.codebooks = (const uint8_t*)sg.codebook;
      S.scales = (const uint16_t*)sg.scales;
      S.bias = (const uint16_t*)sg.bias;
      S.Y = (uint16_t*)sg.y;
      S.ys = sg.y_row_stride;
      S.M = sg.out_features;
      S.tile0 = tiles;
      tiles += (sg.out_features + 15) / 16;
    } else {
      S = p.seg[0];
      S.tile0 = 0x7fffffff;  // never reached
    }
  }
  p.X = (const uint16_t*)X;
  p.xs = xs;
  p.B = batch;
  p.in_groups = in_features / 8;
  p.ntiles = tiles;
  p.nseg = num_segments;
  if (dtype == AQLM_HIP_F16) return K == 2 ? launch_kx8_xres<F16, 2, KR_MAX_SEG>(p, in_features, stream) : launch_kx8_xres<F16, 1, KR_MAX_SEG>(p, in_features, stream);
  return K == 2 ? launch_kx8_xres<BF16, 2, KR_MAX_SEG>(p, in_features, stream) : launch_kx8_xres<BF16, 1, KR_MAX_SEG>(p, in_features, stream);
}
}  // namespace aqlm

extern "C" int aqlm_hip_gemm_kx8_mfma(const void* codes, const void* codebooks, const void* scales, const void* bias, const void* X,
                                      void* Y, int batch, int out_features, int in_features, int num_codebooks, int in_group_size,
                                      long xs, long ys, int dtype, void* stream_) {
  return aqlm_hip_gemm_kx8_mfma_ws(codes, codebooks, scales, bias, X, Y, batch, out_features, in_features, num_codebooks, in_group_size, xs,
                                   ys, dtype, nullptr, 0, stream_);
}

extern "C" int aqlm_hip_gemm_kx8_mfma_ws(const void* codes, const void* codebooks, const void* scales, const void* bias, const void* X,
                                         void* Y, int batch, int out_features, int in_features, int num_codebooks, int in_group_size,
                                         long xs, long ys, int dtype, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!codes || !codebooks || !scales || !X || !Y) {
    set_last_error("aqlm_hip_gemm_kx8_mfma: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (batch <= 0 || out_features <= 0 || in_features <= 0) {
    set_last_error("aqlm_hip_gemm_kx8_mfma: sizes must be positive");
    return AQLM_HIP_E_INVALID;
  }
  if ((num_codebooks != 1 && num_codebooks != 2) || in_group_size != 8) {
    set_last_error("aqlm_hip_gemm_kx8_mfma: 1 or 2 codebooks of 256 x 8 only, got %d codebooks, group %d", num_codebooks, in_group_size);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_gemm_kx8_mfma: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  KxPlan probe{};
  if (!plan_kx8(std::min(batch, 128), out_features, in_features, probe) || !aligned16(codebooks) || !aligned16(X) || xs % 8 != 0) {
    set_last_error("aqlm_hip_gemm_kx8_mfma: needs in_features %% 128 == 0, >= 384, and 16-B aligned codebooks / X rows");
    return AQLM_HIP_E_UNSUPPORTED;
  }
  // layers of more tiles than CUs at 7+ rows: the phased form first -- every workgroup holds its 2..4 tiles' accumulators and the K shares
  // meet once (measured, 2x8 g8 4096 -> 11008: 11.5 / 11.9 / 12.2 us at 8 / 12 / 16 rows on the single-phase kernel, 10.3 / 10.8 / 11.0 phased;
  // at 4 rows 9.3 vs 10.1, and with one tile per CU the single-phase kernel is ahead: profiles/r05_gemm_kx8_phase_geometry.log)
  const int ntiles_all = (out_features + 15) / 16;
  const bool phased_first = (batch >= 7 && ntiles_all > device_cus() && ntiles_all <= 4 * device_cus()) || tuning().kx8_phase_tpb != 0;
  auto try_phased = [&]() -> int {  // the X-resident kernel in phases: images that do not fit the LDS at once, 17 .. 32 rows, multi-round layers
    if (!(batch <= (tuning().kx8_xres_phased == 2 ? 16 : 32) && tuning().kx8_xres && tuning().kx8_xres_phased && in_features % 128 == 0 && xs > 0 &&
          xs < (1l << 21)))
      return AQLM_HIP_E_UNSUPPORTED;
    KpParams kp{};
    kp.codes = (const uint8_t*)codes;
    kp.codebooks = (const uint8_t*)codebooks;
    kp.X = (const uint16_t*)X;
    kp.scales = (const uint16_t*)scales;
    kp.bias = (const uint16_t*)bias;
    kp.Y = (uint16_t*)Y;
    kp.xs = xs;
    kp.ys = ys;
    kp.M = out_features;
    kp.B = batch;
    kp.in_groups = in_features / 8;
    kp.ntiles = (out_features + 15) / 16;
    if (dtype == AQLM_HIP_F16) return num_codebooks == 2 ? launch_kx8_xres_phased<F16, 2>(kp, in_features, stream) : launch_kx8_xres_phased<F16, 1>(kp, in_features, stream);
    return num_codebooks == 2 ? launch_kx8_xres_phased<BF16, 2>(kp, in_features, stream) : launch_kx8_xres_phased<BF16, 1>(kp, in_features, stream);
  };
  if (phased_first) {
    const int e = try_phased();
    if (e != AQLM_HIP_E_UNSUPPORTED) return e;
  }
  if (batch <= 16 && tuning().kx8_xres && (num_codebooks == 2 ? xres_fits<2>(batch, in_features, xs) : xres_fits<1>(batch, in_features, xs))) {
    // <= 16 rows: X resident in LDS, no per-step synchronisation (round 5)
    KrParams<1> kr{};
    kr.seg[0].codes = (const uint8_t*)codes;
    kr.seg[0].codebooks = (const uint8_t*)codebooks;
    kr.seg[0].scales = (const uint16_t*)scales;
    kr.seg[0].bias = (const uint16_t*)bias;
    kr.seg[0].Y = (uint16_t*)Y;
    kr.seg[0].ys = ys;
    kr.seg[0].M = out_features;
    kr.seg[0].tile0 = 0;
    kr.X = (const uint16_t*)X;
    kr.xs = xs;
    kr.B = batch;
    kr.in_groups = in_features / 8;
    kr.ntiles = (out_features + 15) / 16;
    kr.nseg = 1;
    if (dtype == AQLM_HIP_F16) return num_codebooks == 2 ? launch_kx8_xres<F16, 2, 1>(kr, in_features, stream) : launch_kx8_xres<F16, 1, 1>(kr, in_features, stream);
    return num_codebooks == 2 ? launch_kx8_xres<BF16, 2, 1>(kr, in_features, stream) : launch_kx8_xres<BF16, 1, 1>(kr, in_features, stream);
  }
  if (!phased_first) {
    const int e = try_phased();
    if (e != AQLM_HIP_E_UNSUPPORTED) return e;
  }
  // fp32 partials of the K-split form: [<= KX_MAX_KSPLIT][rows of the slab][out_features]; a workspace too small for a slab's plan
  // simply keeps that slab on the no-split form
  auto can_split = [&](int nb) {
    return workspace != nullptr && (reinterpret_cast<uintptr_t>(workspace) & 15u) == 0 &&
           workspace_bytes >= (size_t)KX_MAX_KSPLIT * nb * out_features * sizeof(float);
  };
  for (int b0 = 0; b0 < batch; b0 += 128) {  // every slab is planned before anything is launched (a tail slab plans differently from the probe)
    KxPlan r{};
    if (!plan_kx8(std::min(128, batch - b0), out_features, in_features, r, can_split(std::min(128, batch - b0)))) {
      set_last_error("aqlm_hip_gemm_kx8_mfma: a slab of %d rows at in_features %d is outside the kernel's plans", std::min(128, batch - b0), in_features);
      return AQLM_HIP_E_UNSUPPORTED;
    }
  }
  for (int b0 = 0; b0 < batch; b0 += 128) {  // slabs of 128 rows (the codes are re-read per slab: they are 2 bits per weight)
    const int nb = std::min(128, batch - b0);
    KxPlan r{};
    plan_kx8(nb, out_features, in_features, r, can_split(nb));
    KxParams kp{};
    kp.codes = (const uint8_t*)codes;
    kp.codebooks = (const uint8_t*)codebooks;
    kp.X = (const uint16_t*)X + (long)b0 * xs;
    kp.scales = (const uint16_t*)scales;
    kp.bias = (const uint16_t*)bias;
    kp.Y = (uint16_t*)Y + (long)b0 * ys;
    kp.xs = xs;
    kp.ys = ys;
    kp.M = out_features;
    kp.B = nb;
    kp.in_groups = in_features / 8;
    kp.nsteps = r.nsteps;
    kp.partial = (float*)workspace;
    kp.ksplit = r.ksplit;
    kp.row_blocks = (out_features + 16 * r.rt - 1) / (16 * r.rt);
    int e;
    if (dtype == AQLM_HIP_F16) e = num_codebooks == 2 ? launch_kx8<F16, 2>(kp, r, stream) : launch_kx8<F16, 1>(kp, r, stream);
    else e = num_codebooks == 2 ? launch_kx8<BF16, 2>(kp, r, stream) : launch_kx8<BF16, 1>(kp, r, stream);
    if (e) return e;
    if (r.ksplit > 1) {  // the slices meet: sum in slice order, scale, bias, one rounding (the 1x16 pipeline's finalize kernel)
      GldsFinalizeParams f{};
      f.partial = (const float*)workspace;
      f.scales = (const uint16_t*)scales;
      f.bias = (const uint16_t*)bias;
      f.Y = (uint16_t*)Y + (long)b0 * ys;
      f.ys = ys;
      f.M = out_features;
      f.B = nb;
      f.ksplit = r.ksplit;
      f.row_blocks = (out_features + 127) / 128;
      f.bchunks = (nb + 7) / 8;
      f.rb_rows = 128;
      const dim3 grid((unsigned)(8 * ((f.row_blocks + 7) / 8) * f.bchunks));
      if (dtype == AQLM_HIP_F16) hipLaunchKernelGGL(gemm_glds_finalize_kernel<F16>, grid, dim3(256), 0, stream, f);
      else hipLaunchKernelGGL(gemm_glds_finalize_kernel<BF16>, grid, dim3(256), 0, stream, f);
      if (int e2 = check_hip(hipGetLastError(), "gemm_kx8 finalize launch")) return e2;
    }
  }
  return 0;
}

extern "C" int aqlm_hip_gemm_1x16_mfma(const void* codes, const void* codebook, const void* scales, const void* bias,
                                       const void* X, void* Y, int batch, int out_features, int in_features,
                                       int in_group_size, long xs, long ys, int dtype, void* workspace,
                                       size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!codes || !codebook || !scales || !X || !Y) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: null pointer argument");
    return AQLM_HIP_E_INVALID;
  }
  if (batch <= 0 || out_features <= 0 || in_features <= 0) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: sizes must be positive");
    return AQLM_HIP_E_INVALID;
  }
  if (in_group_size != 8 && in_group_size != 16) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: only codebooks with 8 or 16 features are supported, got %d",
                   in_group_size);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (dtype != AQLM_HIP_F16 && dtype != AQLM_HIP_BF16) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: AQLM HIP kernels only support float16 and bfloat16 (dtype id %d)", dtype);
    return AQLM_HIP_E_UNSUPPORTED;
  }
  if (in_features % BK != 0 || !aligned16(codes) || !aligned16(codebook) || !aligned16(X) || xs % 8 != 0) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: needs in_features %% 64 == 0 and 16-B aligned codes/codebook/X rows");
    return AQLM_HIP_E_UNSUPPORTED;
  }
  const size_t need = aqlm_hip_workspace_bytes(AQLM_HIP_OP_GEMM_1X16_MFMA, batch, out_features, in_features);
  if (!workspace || workspace_bytes < need) {
    set_last_error("aqlm_hip_gemm_1x16_mfma: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return AQLM_HIP_E_INVALID;
  }
  // gemm_variant: 0 = the slice-scan kernel (gemm_1x16_scan.hip, round 6: codebook slices in LDS, no L2 gathers) up to scan_max_rows
  // where it applies, else round 5's routing (5): 16-row blocks where they pay, else the K-split pipeline; 1 = register-staged kernel of
  // round 1, 2 = 16-row blocks wherever they apply, 3 = K-split pipeline only, 4 = slice-scan kernel at any row count
  int variant = tuning().gemm_variant;
  if ((variant == 4 || (variant == 0 && batch <= tuning().scan_max_rows)) && in_group_size == 8 &&
      scan::workspace_bytes(batch, out_features, in_features) != 0 && aligned16(workspace)) {
    return scan::run(codes, codebook, scales, bias, X, Y, batch, out_features, in_features, xs, ys, dtype, workspace, stream);
  }
  if (variant == 4 || variant == 5) variant = 0;
  // batch > 128 is processed in slabs of 128 columns (codes re-gathered per slab)
  const bool use_glds = variant != 1;
  for (int b0 = 0; b0 < batch; b0 += 128) {
    const int nb = std::min(128, batch - b0);
    R16Plan r16{};
    const bool r16_pays = nb <= R16_ROWS_ANY || (nb <= R16_ROWS_SMALL && (long)out_features * in_features <= R16_SMALL_LAYER);
    if ((variant == 2 || (variant == 0 && r16_pays)) && plan_rows16(nb, in_features, in_group_size, r16)) {
      R16Params rp{};
      rp.codes = (const uint8_t*)codes;
      rp.codebook = (const uint8_t*)codebook;
      rp.X = (const uint16_t*)X + (long)b0 * xs;
      rp.scales = (const uint16_t*)scales;
      rp.bias = (const uint16_t*)bias;
      rp.Y = (uint16_t*)Y + (long)b0 * ys;
      rp.xs = xs;
      rp.ys = ys;
      rp.M = out_features;
      rp.B = nb;
      rp.in_groups = in_features / in_group_size;
      rp.nsteps = r16.nsteps;
      int e;
      if (dtype == AQLM_HIP_F16)
        e = in_group_size == 8 ? launch_rows16<F16, 8>(rp, r16, stream) : launch_rows16<F16, 16>(rp, r16, stream);
      else
        e = in_group_size == 8 ? launch_rows16<BF16, 8>(rp, r16, stream) : launch_rows16<BF16, 16>(rp, r16, stream);
      if (e) return e;
      continue;
    }
    GldsPlan q;
    if (use_glds && plan_glds(nb, out_features, in_features, q)) {
      GldsParams gp{};
      gp.codes = (const uint8_t*)codes;
      gp.codebook = (const uint8_t*)codebook;
      gp.X = (const uint16_t*)X + (long)b0 * xs;
      gp.partial = (float*)workspace;
      gp.scales = (const uint16_t*)scales;
      gp.bias = (const uint16_t*)bias;
      gp.Y = (uint16_t*)Y + (long)b0 * ys;
      gp.xs = xs;
      gp.ys = ys;
      gp.M = out_features;
      gp.B = nb;
      gp.in_groups = in_features / in_group_size;
      gp.row_blocks = q.row_blocks;
      gp.ksplit = q.ksplit;
      gp.chunks_base = q.chunks_base;
      gp.chunks_rem = q.chunks_rem;
      gp.store_nt = tuning().gemm_store_nt;
      gp.dbg = tuning().gemm_debug;
      int e;
      if (dtype == AQLM_HIP_F16)
        e = in_group_size == 8 ? launch_glds<F16, 8>(gp, q, stream) : launch_glds<F16, 16>(gp, q, stream);
      else
        e = in_group_size == 8 ? launch_glds<BF16, 8>(gp, q, stream) : launch_glds<BF16, 16>(gp, q, stream);
      if (e) return e;
      if (q.ksplit > 1) {
        GldsFinalizeParams f{};
        f.partial = (const float*)workspace;
        f.scales = (const uint16_t*)scales;
        f.bias = (const uint16_t*)bias;
        f.Y = (uint16_t*)Y + (long)b0 * ys;
        f.ys = ys;
        f.M = out_features;
        f.B = nb;
        f.ksplit = q.ksplit;
        f.row_blocks = (out_features + 127) / 128;
        f.bchunks = (nb + 7) / 8;
        f.rb_rows = 128;
        const dim3 grid((unsigned)(8 * ((f.row_blocks + 7) / 8) * f.bchunks));
        if (dtype == AQLM_HIP_F16) hipLaunchKernelGGL(gemm_glds_finalize_kernel<F16>, grid, dim3(256), 0, stream, f);
        else hipLaunchKernelGGL(gemm_glds_finalize_kernel<BF16>, grid, dim3(256), 0, stream, f);
        if (int e2 = check_hip(hipGetLastError(), "gemm_glds_finalize launch")) return e2;
      }
      continue;
    }
    const GemmPlan g = plan_gemm(nb, out_features, in_features);
    GemmParams p{};
    p.codes = (const uint8_t*)codes;
    p.codebook = (const uint8_t*)codebook;
    p.X = (const uint16_t*)X + (long)b0 * xs;
    p.partial = (float*)workspace;
    p.M = out_features;
    p.K = in_features;
    p.B = nb;
    p.Bpad = g.Bpad;
    p.in_groups = in_features / in_group_size;
    p.kslice = g.kslice;
    p.ksplit = g.ksplit;
    p.xs = xs;
    p.cb_bytes = 65536 * in_group_size * 2;
    p.x_bytes = (uint32_t)std::min<long>(((long)(nb - 1) * xs + in_features) * 2, 0xffffffffL);
    int e;
    if (dtype == AQLM_HIP_F16)
      e = in_group_size == 8 ? launch_gemm<F16, 8>(p, g.nbt, stream) : launch_gemm<F16, 16>(p, g.nbt, stream);
    else
      e = in_group_size == 8 ? launch_gemm<BF16, 8>(p, g.nbt, stream) : launch_gemm<BF16, 16>(p, g.nbt, stream);
    if (e) return e;
    FinalizeParams f{};
    f.partial = (const float*)workspace;
    f.scales = (const uint16_t*)scales;
    f.bias = (const uint16_t*)bias;
    f.Y = (uint16_t*)Y + (long)b0 * ys;
    f.M = out_features;
    f.B = nb;
    f.Bpad = g.Bpad;
    f.ksplit = g.ksplit;
    f.ys = ys;
    dim3 grid((out_features + 31) / 32, g.nbt);
    if (dtype == AQLM_HIP_F16) hipLaunchKernelGGL(gemm_finalize_kernel<F16>, grid, dim3(256), 0, stream, f);
    else hipLaunchKernelGGL(gemm_finalize_kernel<BF16>, grid, dim3(256), 0, stream, f);
    if (int e2 = check_hip(hipGetLastError(), "gemm_finalize launch")) return e2;
  }
  return 0;
}
