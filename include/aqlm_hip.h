/*
 * aqlm_hip.h -- C ABI of libaqlm_hip.so: the MI355X (gfx950 / CDNA4) implementation of the AQLM
 * additive-codebook dequant-fused matvec / matmul path.
 *
 * This is the drop-in boundary for the ONE hot path of Vahe1994/AQLM that this repository replaces
 * (SURVEY.md section 8).  Each entry point names the reference interface it replaces; paths are relative
 * to the reference tree (inference_lib/src/aqlm/inference_kernels/).  INTEGRATION.md shows the binding
 * a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - raw device pointers, caller-owned, never retained; inputs are never written (one documented exception: the
 *     accumulator cells inside a prepacked buffer, see aqlm_hip_gemv_1x16_packed; the *_cells entries avoid it);
 *   - stream-ordered on `stream` (a hipStream_t passed as void*; NULL = the null stream), asynchronous,
 *     no allocation, no synchronisation -> hipGraph-capturable and re-entrant;
 *   - tensors use the reference's checkpoint layout unchanged:
 *       codes      [out_features][in_features/in_group_size][num_codebooks]  int8 (nbits<=8) / int16 (nbits<=16)
 *                  two's-complement containers holding UNSIGNED indices (utils.py:11-31)
 *       codebooks  [num_codebooks][2**nbits][1][in_group_size]               fp16 / bf16
 *       scales     [out_features]   (the reference's [out,1,1,1], contiguous)  fp16 / bf16
 *       bias       [out_features] or NULL
 *     out_group_size is 1 (as in every reference kernel, kernel_selector.py:27-94);
 *   - every kernel fuses the epilogue  y = acc * scales[row] + bias[row]  (replaces
 *     scale_bias_unflatten_output, cuda_kernel.cpp:95-111); accumulation is fp32;
 *   - dtype: AQLM_HIP_F16 or AQLM_HIP_BF16 for x / y / codebooks / scales / bias alike
 *     (check_use_bfloat16, cuda_kernel.cpp:9-25);
 *   - return 0 on success; a positive value is a hipError_t from the launch; negative values are
 *     AQLM_HIP_E_*.  aqlm_hip_last_error() returns a thread-local human-readable message.
 */
#ifndef AQLM_HIP_H_
#define AQLM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AQLM_HIP_ABI_VERSION 9

#define AQLM_HIP_F16 0
#define AQLM_HIP_BF16 1

#define AQLM_HIP_E_INVALID (-1)     /* bad argument (null pointer, non-positive size, misaligned buffer) */
#define AQLM_HIP_E_UNSUPPORTED (-2) /* shape / scheme outside what the entry point implements */

#define AQLM_HIP_MAX_GEMV_BATCH 8

/* ABI version of the loaded library (== AQLM_HIP_ABI_VERSION of the header it was built from). */
int aqlm_hip_abi_version(void);

/* Thread-local message describing the last non-zero return on this thread ("" if none). */
const char* aqlm_hip_last_error(void);

/*
 * y[b, :] = (W x[b, :]) * scales + bias   for b < batch (1..AQLM_HIP_MAX_GEMV_BATCH), 1x16 scheme
 * (one codebook of 65536 entries, in_group_size 8 or 16).  ONE launch for all batch rows: codes and the
 * gathered codebook vectors are reused across the rows of x.
 *
 * Replaces: code1x16_matvec_cuda<bf16,g> + launcher (cuda_kernel.cu:7-95, 476-521), the per-row host loop of
 *           code1x16_matmat (cuda_kernel.cpp:148-182) and its 3-4 epilogue launches (cuda_kernel.cpp:95-111).
 * x_row_stride / y_row_stride are in elements.  Requirements: in_features % in_group_size == 0.
 */
int aqlm_hip_gemv_1x16(const void* codes_i16, const void* codebook, const void* scales, const void* bias,
                       const void* x, void* y, int out_features, int in_features, int in_group_size, int batch,
                       long x_row_stride, long y_row_stride, int dtype, void* stream);

/*
 * One launch for up to AQLM_HIP_MAX_SEGMENTS 1x16 layers that multiply the SAME x (q/k/v or gate/up of a decoder
 * layer): y_s[b, :] = (W_s x[b, :]) * scales_s + bias_s for every segment s.  Each segment has its own codes,
 * codebook, scales, bias and output; all share in_features, in_group_size, batch, dtype.  Results are bit-identical to
 * num_segments calls of aqlm_hip_gemv_1x16.
 *
 * Replaces: the three (two) consecutive code1x16_matmat calls a decoder layer issues on one hidden state
 *           (cuda_kernel.cpp:148-182 called from inference.py:68-76 once per projection); SURVEY.md section 8(f)2.
 */
typedef struct aqlm_hip_segment {
  const void* codes;     /* [out_features][in_features/in_group_size] int16 (aqlm_hip_gemv_1x16_multi), or the
                            prepacked buffer of aqlm_hip_prepack_1x16 (aqlm_hip_gemv_1x16_packed_multi) */
  const void* codebook;  /* [65536][in_group_size] */
  const void* scales;    /* [out_features] */
  const void* bias;      /* [out_features] or NULL */
  void* y;               /* [batch][out_features], row stride y_row_stride elements */
  long y_row_stride;
  int out_features;
  int reserved;          /* set to 0 */
} aqlm_hip_segment;

#define AQLM_HIP_MAX_SEGMENTS 4

int aqlm_hip_gemv_1x16_multi(const aqlm_hip_segment* segments, int num_segments, const void* x, int in_features,
                             int in_group_size, int batch, long x_row_stride, int dtype, void* stream);

/*
 * Same contract for K x 8-bit schemes (256-entry codebooks held in LDS): num_codebooks in 1..16, any
 * in_group_size that is a multiple of 8 (tuned instances: 1x8 g8, 2x8 g8, 8x8 g32; other shapes run a generic kernel).
 *
 * Replaces: Code2x8MatVec / CodeKx8MatVec + launchers (cuda_kernel.cu:144-233, 296-390, 555-620, 709-758),
 *           code2x8_matmat / code1x8_matmat (cuda_kernel.cpp:387-421, 552-586), and the Triton generic gemv the
 *           reference falls back to for 8x8 etc. (triton_kernel.py:30-205).
 */
int aqlm_hip_gemv_kx8(const void* codes_i8, const void* codebooks, const void* scales, const void* bias,
                      const void* x, void* y, int out_features, int in_features, int num_codebooks,
                      int in_group_size, int batch, long x_row_stride, long y_row_stride, int dtype, void* stream);

/*
 * aqlm_hip_gemv_kx8 for up to AQLM_HIP_MAX_SEGMENTS layers of one K x 8-bit scheme that multiply the same x, in one
 * launch (tuned: 1x8 / 2x8 g8; other schemes run one launch per segment).  segment.codes is int8
 * [out_features][in_features/in_group_size][num_codebooks], segment.codebook is [num_codebooks][256][in_group_size].
 * Results agree with separate aqlm_hip_gemv_kx8 calls to fp32 rounding (a segment may run on a different kernel of
 * the family than it would alone: replicated-LDS vs plain LDS).  2 .. 16 rows of 1x8 / 2x8 g8 (ABI 8): ONE launch of the X-resident
 * fused MFMA kernel over all layers when their codebooks and the X image fit one LDS together (X loaded once, tile walk over the
 * layers back to back) -- bit-identical to the layers' own calls; else one launch per layer.
 * Replaces: consecutive code2x8_matmat / code1x8_matmat calls on one hidden state (cuda_kernel.cpp:387-421, 552-586).
 */
int aqlm_hip_gemv_kx8_multi(const aqlm_hip_segment* segments, int num_segments, const void* x, int in_features,
                            int num_codebooks, int in_group_size, int batch, long x_row_stride, int dtype,
                            void* stream);

/*
 * Fully generic gemv (any num_codebooks, nbits <= 16, any in_group_size, codes in 8- or 16-bit containers):
 * the slow-but-correct path for every scheme without a tuned kernel (the role of triton_kernel.py in the
 * reference, kernel_selector.py:91-94).
 */
int aqlm_hip_gemv_generic(const void* codes, const void* codebooks, const void* scales, const void* bias,
                          const void* x, void* y, int out_features, int in_features, int num_codebooks, int nbits,
                          int in_group_size, int batch, long x_row_stride, long y_row_stride, int dtype,
                          void* stream);

/*
 * W[out_features][in_features] = sum_c codebooks[c][codes[.., c]]  (* scales[row] if scales != NULL), row-major.
 * Replaces: Code1x16Dequant / code1x16_dequant_cuda (cuda_kernel.cu:98-142, 523-553), and with scales the
 *           pybind `code1x16_dequant` (cuda_kernel.cpp:184-227).
 */
int aqlm_hip_dequant_1x16(const void* codes_i16, const void* codebook, const void* scales /* nullable */, void* W,
                          int out_features, int in_features, int in_group_size, int dtype, void* stream);

/*
 * Replaces: Code2x8Dequant / CodeKx8Dequant (cuda_kernel.cu:235-294, 392-468, 622-707, 760-817) and the pybind
 *           `code2x8_dequant` / `code1x8_dequant` (cuda_kernel.cpp:423-448, 588-613).  Unlike the reference's
 *           CodeKx8Dequant there is no read-modify-write of W: all codebooks are summed in registers.
 */
int aqlm_hip_dequant_kx8(const void* codes_i8, const void* codebooks, const void* scales /* nullable */, void* W,
                         int out_features, int in_features, int num_codebooks, int in_group_size, int dtype,
                         void* stream);

/*
 * The same for ANY scheme (num_codebooks, nbits <= 16 in 8- / 16-bit containers, any in_group_size): the role of the
 * reference's torch fallback `_dequantize_weight` (utils.py:43-70) for schemes without a tuned kernel -- used by the
 * large-batch and backward ops of such schemes.
 */
int aqlm_hip_dequant_generic(const void* codes, const void* codebooks, const void* scales /* nullable */, void* W,
                             int out_features, int in_features, int num_codebooks, int nbits, int in_group_size,
                             int dtype, void* stream);

/*
 * Large-batch path: Y[B][out] = (X[B][in] @ W^T) * scales + bias with W dequantised tile-by-tile into LDS and
 * contracted on the matrix cores (v_mfma_f32_16x16x32_f16/bf16: a gathered codebook vector IS a fragment lane); W never
 * touches HBM.  Two kernels behind the entry, chosen by batch and layer size (tuning key `gemm_variant` forces one): a K-split
 * pipeline with fp32 partials in `workspace` + a finalize launch, and (<= 16 rows; <= 64 rows on layers of <= 4096 x 4096) a
 * single launch of 16-row blocks over all of K that uses no workspace.  (`gemm_variant` = 4 / `scan_max_rows` > 0: the slice-scan
 * kernel of aqlm_hip_gemm_1x16_scan instead, where it applies.)
 * Replaces: code1x16_matmat_dequant = Code1x16Dequant + F::linear(cuBLAS) + epilogue (cuda_kernel.cpp:249-301).
 * X and Y are row-major with the given row strides (elements).  workspace: aqlm_hip_workspace_bytes(...) bytes
 * (may be 0 -> NULL allowed).
 */
int aqlm_hip_gemm_1x16_mfma(const void* codes_i16, const void* codebook, const void* scales, const void* bias,
                            const void* X, void* Y, int batch, int out_features, int in_features,
                            int in_group_size, long x_row_stride, long y_row_stride, int dtype, void* workspace,
                            size_t workspace_bytes, void* stream);

/*
 * 1x16, in_group_size 8, at 2 .. any number of rows WITHOUT gathers from L2 (round 6; gemm_1x16_scan.hip): the codebook is cut into 8
 * slices of 8192 entries (128 KiB); a workgroup keeps one slice in LDS, scans the canonical codes [out][in / 8] of its row group and
 * feeds v_mfma_f32_16x16x32 with W fragments whose lanes read their entry when it lives in the workgroup's slice and a zero vector
 * when it does not; x stays in registers (every wave owns a K range), the eight slices' partial sums (x K chunks for long rows) go to
 * `workspace` as fp32 planes [8 * chunks][batch][out] and a second launch adds them in plane order, applies scales + bias and rounds
 * once.  Data-oblivious (no prepacked copy, no dependence on the code histogram), deterministic, batch-invariant; rows are processed
 * in passes of 16.  Needs in_features % 256 == 0, 16-B aligned codes / codebook / X rows / workspace.
 * NOT on a default route: measured slower than the L2-gather kernels of aqlm_hip_gemm_1x16_mfma at Llama layer sizes (8-fold
 * redundant scan + 11.5 us fixed: profiles/r06_scan_kernel_*.log); that entry takes it with tuning key `gemm_variant` = 4 or up to
 * `scan_max_rows` (default 0) rows.
 * Replaces: the per-row relaunch of the matvec for 2 .. 6 rows (cuda_kernel.cpp:165-175) and code1x16_matmat_dequant =
 * Code1x16Dequant + cuBLAS + epilogue above (cuda_kernel.cpp:249-301; cuda_kernel.cu:98-142).
 * workspace: aqlm_hip_gemm_1x16_scan_workspace_bytes(batch, out_features, in_features) bytes (0 = the shape has no plan).
 */
size_t aqlm_hip_gemm_1x16_scan_workspace_bytes(int batch, int out_features, int in_features);
int aqlm_hip_gemm_1x16_scan(const void* codes_i16, const void* codebook, const void* scales, const void* bias, const void* X, void* Y,
                            int batch, int out_features, int in_features, long x_row_stride, long y_row_stride, int dtype,
                            void* workspace, size_t workspace_bytes, void* stream);

/*
 * Load-time repack of 1x16 codes (g 8 or 16) into the slice-bucketed format v7 consumed by aqlm_hip_gemv_1x16_packed (layout:
 * aqlm_amd/csrc/gemv_packed.hip, specification tests/packed_model.py; 4 bytes per code + ~6 bytes per (row, slice),
 * plus 64 bytes per output row of zero-at-rest accumulator cells for the fused finalize).
 * The reference does the analogous thing for its CPU kernel: a one-off permutation of `codes` at first use
 * (inference.py:78-83).
 *
 * The packed kernels give every workgroup one slice of the codebook (`code >> 12`) and the codes that fall into it, so their
 * speed depends on how evenly the codes use the slices; the reference's kernels are data-oblivious (one warp per output row,
 * cuda_kernel.cu:16-27, launcher :496-507), and real checkpoints (k-means + beam search, src/aq.py:286-356) are not uniform.
 * Format v7 therefore balances at pack time, losslessly:
 *   * RELABELLING (AQLM_HIP_PACKED_RELABELLED): the repack counts how often every codebook entry is used and deals the entries
 *     to the slices so that the slices carry equal numbers of codes (longest-processing-time greedy, at most 65536 / slices
 *     entries per slice).  The permutation is kept inside the buffer (aqlm_hip_unpack_1x16 undoes it: bit-exact codes), and the
 *     kernels read a PERMUTED IMAGE of the codebook that also lives in the buffer: the caller writes it with
 *     aqlm_hip_packed_set_codebook (again whenever the codebook changes) -- the `codebook` argument of the matvec entries is then
 *     ignored.  Codes that already use the slices evenly (within 2 %) are not relabelled: no permutation, no image, and the
 *     buffer is what format v6 held.
 *   * VARIABLE GEOMETRY (AQLM_HIP_PACKED_VARGEOM; 16-byte vectors only): a single entry used by more than 1 / 16 of all codes
 *     cannot be balanced by any labelling.  The 256 workgroups are then dealt to the slices in proportion to their work
 *     (slice_groups[s] workgroups for slice s, each owning out_features / slice_groups[s] consecutive rows; >= 8 per slice), instead
 *     of 16 row groups for every slice.  Every row still receives exactly one contribution per slice, so the finalize is
 *     unchanged.  Such buffers run on the single-layer entries (the shared-input / chain / publish entries run them one layer per
 *     launch or refuse with AQLM_HIP_E_UNSUPPORTED); AQLM_HIP_PREPACK_UNIFORM_ONLY asks the repack not to use it.
 *   aqlm_hip_prepack_1x16_bytes  capacity the caller must provide (0: shape not covered -- in_group_size not 8 or 16, more
 *                                input groups than an entry's 12 / 11-bit slot field addresses, or rows whose tables fit no LDS
 *                                image); more than the result needs: the repack uses the tail as scratch.
 *                                in_group_size 16 = the second instantiation (32 slices of 2048 x 32 B).
 *   aqlm_hip_prepack_1x16[_ex]   fills `packed` and `*desc`; desc->used_bytes <= capacity is what has to be kept (the
 *                                buffer may be trimmed / copied; the descriptor travels with it and is also stored in the
 *                                buffer's first bytes).  Synchronises `stream` (load-time call, not graph-capturable).
 *                                AQLM_HIP_E_UNSUPPORTED when even the balanced streams do not fit the capacity (rows that differ
 *                                wildly from one another); `flags`: AQLM_HIP_PREPACK_* below (0 = everything allowed).
 *   aqlm_hip_packed_set_codebook writes the permuted codebook image of a relabelled buffer (no-op for others) and sets
 *                                AQLM_HIP_PACKED_HAS_CODEBOOK in `*desc`; stream-ordered, no synchronisation.
 *   aqlm_hip_packed_desc_read    descriptor from the first sizeof(desc) bytes of a packed buffer copied to the host.
 *   aqlm_hip_unpack_1x16         the inverse: canonical int16 codes [out][in/8] from a packed buffer (lossless).
 *   aqlm_hip_packed_plan_*       the two host-side planning steps of the repack (pure functions, exposed for tests and tools).
 */
#define AQLM_HIP_PACKED_RELABELLED 1u   /* codebook entries were dealt to the slices: permutation + codebook image inside the buffer */
#define AQLM_HIP_PACKED_VARGEOM 2u      /* slice_groups[] is not uniform: the variable-geometry kernels serve the buffer */
#define AQLM_HIP_PACKED_HAS_CODEBOOK 4u /* the codebook image has been written (aqlm_hip_packed_set_codebook) */

#define AQLM_HIP_PREPACK_NO_RELABEL 1   /* keep the checkpoint's labelling (format v6 behaviour) */
#define AQLM_HIP_PREPACK_UNIFORM_ONLY 2 /* 16 row groups for every slice (what the shared-input / publish kernels need) */

typedef struct aqlm_hip_packed_desc {
  uint32_t magic;   /* "AQP7" */
  uint32_t version; /* 7 */
  int32_t out_features, in_features;
  int32_t slices_log2; /* 4: 16 codebook slices of 4096 entries */
  int32_t waves;       /* wave ranges per stream = waves per workgroup of the gemv kernel */
  int32_t steps;       /* KiB steps per wave range */
  int32_t entry_bytes; /* 4 */
  uint64_t used_bytes;
  uint32_t x_copies;   /* rotated copies of x the batch-1 kernel keeps in LDS (1..4); entries name the copy they read */
  float codebook_absmax; /* max |codebook entry| of the layer, set by the CALLER after prepack (prepack does not see the
                            codebook; 0 = unknown).  > 0 enables the fused finalize of aqlm_hip_gemv_1x16_packed[_multi]:
                            it bounds the slice sums, from which the kernel derives an overflow-free fixed-point scale.
                            Must be >= the true maximum (update it when the codebook is retrained); any finite value
                            that is too large only costs resolution (the sums keep ~47 bits below the bound).
                            The value travels as a kernel argument: a hipGraph that captured the launch replays with
                            the bound it was captured with -- re-capture (or capture with a generous bound) if the
                            codebook's range can grow afterwards.  A sum beyond the bound is not wrapped silently:
                            the row's result is NaN (the kernel checks |sum| <= 2 * bound). */
  uint32_t flags;        /* AQLM_HIP_PACKED_* */
  int32_t rows_per_group; /* most rows any workgroup owns (sizes the row tables; uniform geometry: ceil(out_features / row groups)) */
  uint8_t slice_groups[32]; /* workgroups (= row groups) of every slice; their sum is 256 */
} aqlm_hip_packed_desc;

size_t aqlm_hip_prepack_1x16_bytes(int out_features, int in_features, int in_group_size);
int aqlm_hip_prepack_1x16(const void* codes_i16, int out_features, int in_features, int in_group_size, void* packed,
                          size_t packed_bytes, aqlm_hip_packed_desc* desc, void* stream);
int aqlm_hip_prepack_1x16_ex(const void* codes_i16, int out_features, int in_features, int in_group_size, void* packed,
                             size_t packed_bytes, aqlm_hip_packed_desc* desc, int flags, void* stream);
int aqlm_hip_packed_set_codebook(aqlm_hip_packed_desc* desc, void* packed, const void* codebook, void* stream);
int aqlm_hip_packed_desc_read(const void* header_host, size_t header_bytes, aqlm_hip_packed_desc* desc);
int aqlm_hip_unpack_1x16(const aqlm_hip_packed_desc* desc, const void* packed, void* codes_i16, void* stream);
/* Host-side planning steps (no GPU work).  relabel: usage counts of the 65536 entries -> new_of_old[65536] (returns 1 and
 * fills it when the entries should be re-dealt, 0 when the slices are already even within 2 %).  geometry: lane-steps
 * (units of 4 entries) of every slice -> slice_groups[1 << slices_log2] (returns 1 when the result is not uniform). */
int aqlm_hip_packed_plan_relabel(const uint32_t* usage, int slices_log2, uint16_t* new_of_old);
/* The same deal with `force` != 0: also when the slices' total masses are already within 2 % of even -- what the repack calls
 * whenever the 16 x 16 layout is not balanced (label use correlated with the row leaves every global histogram flat; round 6). */
int aqlm_hip_packed_plan_relabel_ex(const uint32_t* usage, int slices_log2, int force, uint16_t* new_of_old);
int aqlm_hip_packed_plan_geometry(const uint64_t* slice_steps, int slices_log2, int out_features, int in_features,
                                  uint8_t* slice_groups);

/*
 * 1x16 g8 matvec for 1..AQLM_HIP_MAX_GEMV_BATCH input rows on prepacked codes: every CU keeps one 64 KiB slice of the
 * codebook in LDS and walks only the codes of that slice; the rows of x share the codes and the gathered codebook
 * vectors.  Same result contract as aqlm_hip_gemv_1x16 (which it replaces for large layers; same reference lines:
 * cuda_kernel.cu:7-95, cuda_kernel.cpp:148-182 incl. the per-row relaunch loop :165-175).  x / y row strides in elements.
 * Finalize: with desc->codebook_absmax > 0 the 16 slice workgroups of an output row add their sums as fixed-point
 * integers into one 64-bit cell INSIDE the packed buffer (order-independent, hence deterministic; the cells are zero at rest
 * and every call leaves them zero), and the last one to arrive writes y -- one kernel, no workspace (`workspace` may be
 * NULL).  The packed buffer is therefore WRITTEN by the call: do not run the same packed buffer on two streams at once.
 * With codebook_absmax == 0 the call falls back to two kernels and needs
 * workspace: aqlm_hip_workspace_bytes(AQLM_HIP_OP_GEMV_1X16_PACKED, batch, out, in) bytes of fp32 partials.
 */
int aqlm_hip_gemv_1x16_packed(const aqlm_hip_packed_desc* desc, void* packed, const void* codebook,
                              const void* scales, const void* bias, const void* x, void* y, int batch,
                              long x_row_stride, long y_row_stride, int dtype, void* workspace, size_t workspace_bytes,
                              void* stream);

/*
 * Re-entrant form of the single-kernel finalize: the accumulator cells are the CALLER's -- `cells` = at least
 * batch * out_features * 8 bytes, 8-B aligned, zero-filled once; every call leaves them zero.  The packed buffer is only
 * read, so the same layer may run on several streams / from several host threads at once, like the reference's stateless
 * launcher (cuda_kernel.cu:505-509); give every stream its own cells (launches on one stream are ordered, so all layers
 * of a stream can share one set).  Requires desc->codebook_absmax > 0.  Same results, bit for bit, as
 * aqlm_hip_gemv_1x16_packed.
 */
int aqlm_hip_gemv_1x16_packed_cells(const aqlm_hip_packed_desc* desc, const void* packed, const void* codebook,
                                    const void* scales, const void* bias, const void* x, void* y, int batch,
                                    long x_row_stride, long y_row_stride, int dtype, void* cells, size_t cells_bytes,
                                    void* stream);

/*
 * aqlm_hip_gemv_1x16_packed that also names the prepacked layer which runs NEXT on the same stream (chain prefetch; no
 * reference counterpart -- the reference launches each layer cold, cuda_kernel.cpp:148-182): a few extra waves of every
 * workgroup request the next layer's entry stream and codebook so that they sit in the GPU's L2 / Infinity Cache when the
 * next launch starts (a batch-1 matvec is bound by the latency of its cold start, not by HBM bandwidth, and HBM idles
 * most of the kernel).  A hint: results are identical to aqlm_hip_gemv_1x16_packed; next_* may be NULL (then it IS that
 * call).  The next layer's buffers are only read.
 */
int aqlm_hip_gemv_1x16_packed_chain(const aqlm_hip_packed_desc* desc, void* packed, const void* codebook,
                                    const void* scales, const void* bias, const void* x, void* y, int batch,
                                    long x_row_stride, long y_row_stride, int dtype, void* workspace,
                                    size_t workspace_bytes, const aqlm_hip_packed_desc* next_desc,
                                    const void* next_packed, const void* next_codebook, void* stream);

/*
 * aqlm_hip_gemv_1x16_packed for up to AQLM_HIP_MAX_SEGMENTS prepacked layers that share x, in one launch (+ one
 * finalize; none when every descriptor carries codebook_absmax: the fused finalize of aqlm_hip_gemv_1x16_packed, which
 * writes into the packed buffers); segment.codes is the prepacked buffer, descs[s] its descriptor.  workspace (two-kernel
 * fallback only): the sum over segments of
 * aqlm_hip_workspace_bytes(AQLM_HIP_OP_GEMV_1X16_PACKED, batch, out_features_s, in_features), 16-B aligned.  Results
 * are bit-identical to separate aqlm_hip_gemv_1x16_packed calls.
 */
int aqlm_hip_gemv_1x16_packed_multi(const aqlm_hip_segment* segments, const aqlm_hip_packed_desc* const* descs,
                                    int num_segments, const void* x, int in_features, int batch, long x_row_stride,
                                    int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* aqlm_hip_gemv_1x16_packed_multi with caller-owned accumulator cells (see aqlm_hip_gemv_1x16_packed_cells): `cells` holds
 * the segments' cells back to back, batch * out_features_s * 8 bytes each, zero-filled once and left zero. */
int aqlm_hip_gemv_1x16_packed_multi_cells(const aqlm_hip_segment* segments, const aqlm_hip_packed_desc* const* descs,
                                          int num_segments, const void* x, int in_features, int batch, long x_row_stride,
                                          int dtype, void* cells, size_t cells_bytes, void* stream);

/*
 * Row-parallel ("in"-split) layers over several MI355X: the finalize of the prepacked matvec fused with a ONE-SHOT
 * all-reduce over xGMI (no reference counterpart -- the reference has no tensor parallelism; BASELINE.json north star:
 * the 70B layer 8192 -> 28672 split over 8 GPUs).  Every rank runs aqlm_hip_gemv_1x16_packed_partials on its shard (the
 * main kernel only: fp32 slice partials [16][batch][out] in `workspace`), then aqlm_hip_xgmi_finalize: it publishes the
 * slice-summed vector in a buffer the peers have mapped (IPC), raises a flag, waits (bounded) for the peers' flags, reads
 * their vectors over xGMI and adds them in rank order, then scale + bias + one rounding -- y is bit-identical on every rank.
 * State per rank: one device allocation of aqlm_hip_xgmi_state_bytes(max_elems) bytes, zero-filled except epoch = 1, laid
 * out as  [pub: 2 x max_elems fp32][flag: 2 x u32 at byte 8 * max_elems][epoch, 2 tickets: 3 x u32 at +64][arrival counters of
 * aqlm_hip_gemv_1x16_packed_publish: 9 x u32 at +80][status: u32 at +128];
 * `peer_pub[r]` / `peer_flag[r]` are DEVICE arrays (world entries) of the peers' pub / flag addresses as mapped into this
 * process (own entries included).  Collective semantics: every rank calls it the same number of times.  On a time-out
 * (spin_limit polls, 0 = default ~seconds) status becomes 1 and y is NaN.  Graph-capturable (the epoch lives in device memory).
 */
typedef struct aqlm_hip_xgmi {
  const void* const* peer_pub;   /* device array [world] of float* */
  const void* const* peer_flag;  /* device array [world] of uint32_t* */
  void* epoch;                   /* this rank's epoch / ticket words (device) */
  void* status;                  /* this rank's status word (device) */
  int rank, world, max_elems;
  uint32_t spin_limit;
} aqlm_hip_xgmi;

size_t aqlm_hip_xgmi_state_bytes(int max_elems);
int aqlm_hip_gemv_1x16_packed_partials(const aqlm_hip_packed_desc* desc, const void* packed, const void* codebook, const void* x,
                                       int batch, long x_row_stride, int dtype, void* workspace, size_t workspace_bytes,
                                       void* stream);
int aqlm_hip_xgmi_finalize(const aqlm_hip_xgmi* xg, const void* partials, const void* scales, const void* bias, void* y,
                           int out_features, int batch, long y_row_stride, int dtype, void* stream);
/*
 * Two launches instead of three: aqlm_hip_gemv_1x16_packed_publish is the shard's matvec with the single-kernel finalize
 * (descriptor with codebook_absmax > 0) whose last-arrival branch writes the row's fp32 total into this rank's pub buffer
 * (`pub_own` / `flag_own` = the rank's own pub and flag addresses, i.e. peer_pub[rank] / peer_flag[rank]) and whose last
 * workgroup raises the flag; then aqlm_hip_xgmi_finalize with partials == NULL runs the reduce only (poll the peers' flags,
 * add the vectors in rank order, scale + bias + one rounding).  AQLM_HIP_E_UNSUPPORTED when the rows do not fit one launch
 * or the codebook range is unknown: use the partials form above.  State words used: epoch + 4 .. + 12 (arrival counters).
 */
int aqlm_hip_gemv_1x16_packed_publish(const aqlm_hip_packed_desc* desc, void* packed, const void* codebook, const void* x,
                                      int batch, long x_row_stride, int dtype, const aqlm_hip_xgmi* xg, void* pub_own,
                                      void* flag_own, void* stream);

/*
 * Batch-1 matvec for 8 x 8-bit schemes (e.g. the 2-bit 8x8 g32 models; in_group_size 8, 16 or 32) through per-token
 * look-up tables in LDS:  lut[j,c,v] = <codebooks[c,v], x_j>,  y[i] = sum lut[j,c,codes[i,j,c]]  -- the formulation of
 * the reference's CPU kernel (numba_kernel.py:37-48) mapped onto LDS slabs.  Same result contract as aqlm_hip_gemv_kx8
 * (which it replaces for num_codebooks == 8; reference route: triton_kernel.py).  workspace:
 * aqlm_hip_workspace_bytes(AQLM_HIP_OP_GEMV_8X8_LUT, in_group_size, out, in) bytes  [note: the `batch` slot carries
 * in_group_size for this op].
 */
int aqlm_hip_gemv_8x8_lut(const void* codes_i8, const void* codebooks, const void* scales, const void* bias,
                          const void* x, void* y, int out_features, int in_features, int in_group_size, int dtype,
                          void* workspace, size_t workspace_bytes, void* stream);

/*
 * aqlm_hip_gemv_8x8_lut for up to AQLM_HIP_MAX_SEGMENTS 8-codebook layers that share x (batch 1) in one launch
 * (+ one finalize).  workspace: the sum over segments of
 * aqlm_hip_workspace_bytes(AQLM_HIP_OP_GEMV_8X8_LUT, in_group_size, out_features_s, in_features).  A row's value does
 * not depend on how the rows are dealt to workgroups, so results equal separate aqlm_hip_gemv_8x8_lut calls bit for bit.
 */
int aqlm_hip_gemv_8x8_lut_multi(const aqlm_hip_segment* segments, int num_segments, const void* x, int in_features,
                                int in_group_size, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same two operations in ONE kernel each (no finalize launch): the slab sums of an output row are added as fixed-point
 * integers into one 64-bit cell (one returning atomic per slab and row; integer adds commute, so the result does not
 * depend on the arrival order: deterministic), and the workgroup that finds all other slabs arrived applies scale and
 * bias, rounds once, writes y and puts the cell back to zero.  The fixed-point unit is derived inside the kernel from
 * max|codebook| and max|x| (both read by every workgroup): no overflow whatever the data; non-finite inputs give NaN.
 * `cells`: out_features (multi: the sum over the segments, in segment order) x 8 bytes, 8-B aligned, ZERO-FILLED by the
 * caller once after allocation; every call leaves them zero.  A set of cells serves one launch at a time (stream order);
 * keep one per layer, or one per stream.
 */
int aqlm_hip_gemv_8x8_lut_fused(const void* codes_i8, const void* codebooks, const void* scales, const void* bias,
                                const void* x, void* y, int out_features, int in_features, int in_group_size, int dtype,
                                void* cells, size_t cells_bytes, void* stream);
int aqlm_hip_gemv_8x8_lut_multi_fused(const aqlm_hip_segment* segments, int num_segments, const void* x, int in_features,
                                      int in_group_size, int dtype, void* cells, size_t cells_bytes, void* stream);

/*
 * Planar layout of 8 x 8-bit codes (ABI 5): [8 codebooks][out_features][jp] bytes, jp = in_features / in_group_size rounded up
 * to 4 (padding bytes zero) -- the checkpoint's [out][in_groups][8] transposed, same size, lossless.  With it a workgroup of the
 * look-up-table matvec serves ONE codebook x 128 input groups: it fetches 16 KiB of codebook (g = 32) instead of all 128 KiB
 * and reads its codes as 128 contiguous bytes per row.  Load-time re-layout, the counterpart of the reference's code
 * permutation for its CPU look-up kernel (inference.py:78-83).  aqlm_hip_8x8_planar_bytes returns 0 for impossible sizes;
 * buffers 8-byte aligned; pack / unpack are plain kernels on `stream` (no sync, capturable).
 */
size_t aqlm_hip_8x8_planar_bytes(int out_features, int in_features, int in_group_size);
int aqlm_hip_8x8_planar_pack(const void* codes_i8, int out_features, int in_features, int in_group_size, void* planar,
                             size_t planar_bytes, void* stream);
int aqlm_hip_8x8_planar_unpack(const void* planar, int out_features, int in_features, int in_group_size, void* codes_i8,
                               void* stream);

/*
 * aqlm_hip_gemv_8x8_lut[_fused] / _multi[_fused] on planar codes (replaces the same reference route, triton_kernel.py:30-125,
 * reached through kernel_selector.py:91-94).  `fused` != 0: `workspace` = zero-at-rest cells as for the _fused entries, and
 * `codebook_absmax` (> 0, a bound of max |codebook entry| over ALL 8 codebooks, e.g. taken at load time) replaces the in-kernel
 * maximum -- a workgroup sees one codebook only, and every workgroup of a row must derive the same fixed-point unit; a bound
 * that is too large only costs resolution, one that is too small gives NaN rows, never a wrong number.  `fused` == 0:
 * `workspace` = aqlm_hip_workspace_bytes(AQLM_HIP_OP_GEMV_8X8_LUT, ...) bytes of fp32 partials, `codebook_absmax` ignored.
 * _multi: segments[k].codes = the segment's planar buffer, codebook_absmax[k] its bound.  Results equal the canonical-layout
 * entries up to fp32 summation order (other slab shape), and _multi equals the single-layer entry bit for bit.
 */
int aqlm_hip_gemv_8x8_lut_planar(const void* planar, const void* codebooks, const void* scales, const void* bias, const void* x,
                                 void* y, int out_features, int in_features, int in_group_size, int dtype, float codebook_absmax,
                                 void* workspace, size_t workspace_bytes, int fused, void* stream);

/*
 * The look-up-table matvec for 1..AQLM_HIP_MAX_GEMV_BATCH input rows in ONE launch (round 5; single-kernel form): row b of x
 * runs as its own set of workgroups (tables are functions of x; what the rows share is the launch and the codes in L2) with its
 * own zero-at-rest cells -- `cells` = batch * out_features * 8 bytes -- and its own output row.  `planar` != 0: `codes` is the
 * planar buffer of aqlm_hip_8x8_planar_pack and codebook_absmax must be > 0; else canonical codes (codebook_absmax ignored).
 * A row's bits equal those of the same row launched alone through the batch-1 entries.
 * Replaces: the per-row loop around the reference's generic gemv for 8x8 (triton_kernel.py:161-182).
 */
int aqlm_hip_gemv_8x8_lut_batch(const void* codes, const void* codebooks, const void* scales, const void* bias, const void* x,
                                void* y, int out_features, int in_features, int in_group_size, int batch, long x_row_stride,
                                long y_row_stride, int dtype, int planar, float codebook_absmax, void* cells, size_t cells_bytes,
                                void* stream);
int aqlm_hip_gemv_8x8_lut_planar_multi(const aqlm_hip_segment* segments, const float* codebook_absmax, int num_segments,
                                       const void* x, int in_features, int in_group_size, int dtype, void* workspace,
                                       size_t workspace_bytes, int fused, void* stream);

/*
 * Large-batch path of the 8-bit schemes (ABI 6): Y[B][out] = (X[B][in] @ W^T) * scales + bias for 1 or 2 codebooks of 256 x 8
 * (1x8 g8, 2x8 g8), W never materialised: the codebooks live in LDS, a block owns 16 output rows over all of K, every lane
 * takes its lanes of the 16 x 32 MFMA fragments straight from the codebook entries its code bytes name (one MFMA per codebook:
 * exact products, fp32 sums -- W is never rounded), X streams through LDS.  No workspace, one launch per 128 batch rows.
 * At <= 32 rows (<= 16: ABI 7) X is RESIDENT in LDS instead: a workgroup loads the X image once (in phases when batch x in_features x 2
 * bytes exceed the LDS, and for 17 .. 32 rows: ABI 8) and walks 16-row tiles with nothing to synchronise inside a tile; a row's bits then depend neither on
 * the other rows of the call nor on their number.
 * Replaces: code2x8_matmat_dequant / code1x8_matmat_dequant = Code2x8Dequant / CodeKx8Dequant + F::linear(cuBLAS) + epilogue
 * (cuda_kernel.cpp:450-484, 615-649; kernels cuda_kernel.cu:235-294, 392-468).
 * AQLM_HIP_E_UNSUPPORTED (the caller dequantises + calls its GEMM, like the reference): other schemes, in_features not a
 * multiple of 128 or < 384, X rows not 16-B aligned.
 */
int aqlm_hip_gemm_kx8_mfma(const void* codes_i8, const void* codebooks, const void* scales, const void* bias, const void* X,
                           void* Y, int batch, int out_features, int in_features, int num_codebooks, int in_group_size,
                           long x_row_stride, long y_row_stride, int dtype, void* stream);

/*
 * 8 codebooks of 256 x 32 (8x8 g32, 2 bits per weight) at 2 .. many batch rows (ABI 8): Y[B][out] = (X[B][in] @ W^T) * scales + bias,
 * W never materialised.  The eight codebooks (128 KiB) live in LDS as 16-byte piece planes; a workgroup owns 16-row output tiles over
 * all of K, a group of 32 features is one k-step of v_mfma_f32_16x16x32 whose A fragment lanes gather their piece of the entry
 * the lane's code byte names -- one MFMA per codebook, so the eight terms of a weight meet in the fp32 accumulator (exact
 * products, fp32 sums; W is never rounded).  Codes in the CHECKPOINT layout [out][in_groups][8] (not the planar copy).  No
 * workspace, one launch per 64 batch rows; a row's bits do not depend on the other rows of the call nor on their number.
 * Replaces: the reference's Triton kernel looped over the rows (triton_kernel.py:161-182) and, above the gemv rule's 6 rows,
 * dequantize_gemm = _dequantize_weight + F.linear (dequantization.py:9-21, utils.py:43-70) for this scheme.
 * AQLM_HIP_E_UNSUPPORTED (the caller takes its other routes): group sizes other than 32, in_features not a multiple of 256 or
 * < 2048, codes / codebooks / X rows not 16-B aligned.
 */
int aqlm_hip_gemm_8x8_mfma(const void* codes_i8, const void* codebooks, const void* scales, const void* bias, const void* X, void* Y,
                           int batch, int out_features, int in_features, int in_group_size, long x_row_stride, long y_row_stride,
                           int dtype, void* stream);

/*
 * aqlm_hip_gemm_kx8_mfma with a workspace (ABI 8): at 49+ batch rows every 16-row block of the no-workspace form pulls all of X
 * (batch x in_features x 2 bytes: 1 MiB at 128 rows of a 4096-wide layer) through its CU's L1; given `workspace` (16-B aligned,
 * aqlm_hip_workspace_bytes(AQLM_HIP_OP_GEMM_KX8_MFMA, batch, out_features, in_features) bytes) the entry takes two row tiles per
 * block and deals the K range to 2 or 4 blocks -- half / a quarter of the X bytes per CU -- whose fp32 tiles a second small kernel
 * sums in slice order before scale, bias and the one rounding (deterministic; results differ from the no-workspace form only in
 * the order of the fp32 sums).  workspace == NULL or too small: exactly aqlm_hip_gemm_kx8_mfma.
 */
int aqlm_hip_gemm_kx8_mfma_ws(const void* codes_i8, const void* codebooks, const void* scales, const void* bias, const void* X,
                              void* Y, int batch, int out_features, int in_features, int num_codebooks, int in_group_size,
                              long x_row_stride, long y_row_stride, int dtype, void* workspace, size_t workspace_bytes, void* stream);

#define AQLM_HIP_OP_GEMM_1X16_MFMA 1
#define AQLM_HIP_OP_GEMM_KX8_MFMA 6
#define AQLM_HIP_OP_GEMV_1X16_PACKED 3
#define AQLM_HIP_OP_GEMV_8X8_LUT 4
#define AQLM_HIP_OP_GEMV_1X16_G16_PACKED 5 /* prepacked codes of 16-element vectors: 32 slices */
size_t aqlm_hip_workspace_bytes(int op, int batch, int out_features, int in_features);

/*
 * 128-bit position-sensitive checksum of a device buffer: out[0] = sum of its 32-bit words, out[1] = sum of word_i * (odd
 * multiplier of i), both mod 2^64 (trailing 1-3 bytes count as one zero-padded word).  `out_u64x2` is 16 bytes of DEVICE memory,
 * overwritten; stream-ordered, no synchronisation (the caller reads it back).
 * What it is for: everything this library DERIVES from the parameters at load time (prepacked / planar codes, the codebook image
 * and range, a dense copy of W) goes stale when a caller overwrites a parameter through a path that leaves no trace on the host
 * (`tensor.data.copy_()` does not bump the version counter).  The reference has no such state -- its launcher reads the live
 * tensors on every call (cuda_kernel.cpp:148-182) -- so the host side re-checks this checksum every few hundred calls
 * (aqlm_amd/inference.py, DERIVED_CHECK_EVERY) and rebuilds what no longer matches.
 */
int aqlm_hip_checksum(const void* data, size_t bytes, void* out_u64x2, void* stream);

/*
 * Tuning / experiment knobs (process-wide, not part of the reference surface; defaults are the shipped
 * configuration).  Unknown keys return AQLM_HIP_E_INVALID.  Keys: struct Tuning in aqlm_amd/csrc/aqlm_common.h.
 */
int aqlm_hip_set_tuning(const char* key, int value);
int aqlm_hip_get_tuning(const char* key, int* value);

#ifdef __cplusplus
}
#endif
#endif /* AQLM_HIP_H_ */
