// Compiled host glue of the decode path: what a QuantizedLinear.forward of <= 8 rows does between "here is x" and
// "the kernel is on torch's current stream", without the interpreter.
//
// Replaces (behaviour, not code): the C++ side of the reference's ops -- code1x16_matmat / code2x8_matmat / code1x8_matmat
// (inference_lib/src/aqlm/inference_kernels/cuda_kernel.cpp:148-182, 387-421, 552-586) and their pybind registration
// (cuda_kernel.cpp:686-699): flatten the input, allocate the output, launch on the current stream.  The Python ops of
// hip_kernel.py do the same through ctypes and stay the general path (every scheme, > 8 rows, autograd, tracing); this
// object is the fast lane of one module: ~17-19 us of interpreter work per eager call become ~5.
//
// A FastLinear never owns the truth: it holds references to the module's parameter tensors and compares their identity and
// version counters on every call; anything unexpected (a parameter rebound or written in place, an input that needs
// grad, another dtype / device, > 8 rows) makes forward() return None and the Python path -- which knows how to rebuild
// derived state -- takes the call.  Links against libaqlm_hip.so (the C ABI of include/aqlm_hip.h); no device code here.
#include <torch/extension.h>

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPGraphsC10Utils.h>

#include <torch/library.h>

#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include <cstring>
#include <string>

#include "../../include/aqlm_hip.h"

namespace {

enum Kind : int { kPacked1x16 = 0, kGemv1x16 = 1, kGemvKx8 = 2, kLutPlanar8x8 = 3 };

struct Watched {  // a parameter of the module: same Python object, same storage, same version as when the lane was built
  PyObject* obj = nullptr;
  const void* data = nullptr;
  uint32_t version = 0;
  bool versioned = false;
};

static Watched watch(const py::dict& params, const char* name, at::Tensor* out) {
  Watched w;
  PyObject* o = PyDict_GetItemString(params.ptr(), name);  // borrowed
  if (o == nullptr || o == Py_None) return w;
  w.obj = o;
  const at::Tensor& t = THPVariable_Unpack(o);
  w.data = t.defined() && t.numel() ? t.data_ptr() : nullptr;
  w.versioned = t.defined() && !t.is_inference();
  w.version = w.versioned ? t._version() : 0;
  if (out) *out = t;
  return w;
}

static bool unchanged(const py::dict& params, const char* name, const Watched& w) {
  PyObject* o = PyDict_GetItemString(params.ptr(), name);
  if (o == Py_None) o = nullptr;
  if (o != w.obj) return false;
  if (o == nullptr) return true;
  const at::Tensor& t = THPVariable_Unpack(o);
  if ((t.numel() ? t.data_ptr() : nullptr) != w.data) return false;
  return !w.versioned || t._version() == w.version;
}

// Accumulator cells of the single-kernel finalize, one zero-at-rest set per (device, stream) -- see hip_kernel.py
// (_packed_cells): the packed buffers are only read, so a layer may run on several streams at once.  Kept until the caller
// releases them (release_stream_cells: a hipGraph captured on the stream may hold the address, which only the caller knows).
constexpr int64_t kCellsBytes = (int64_t)AQLM_HIP_MAX_GEMV_BATCH * 131072 * 8;
static std::mutex g_cells_mu;
static std::map<std::pair<int, void*>, at::Tensor> g_cells;

// ONE registry for the whole process: the Python ops take their cells from here too (stream_cells_tensor below) when the
// extension is loaded, so a stream has one 8 MiB set, not one per code path.
static at::Tensor stream_cells_tensor(const at::Tensor& like, void* stream, int64_t need_bytes) {
  if (need_bytes > kCellsBytes) return at::Tensor();
  std::mutex& mu = g_cells_mu;
  std::map<std::pair<int, void*>, at::Tensor>& cells = g_cells;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_pair((int)like.device().index(), stream);
  auto it = cells.find(key);
  if (it == cells.end()) {
    if (c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None) return at::Tensor();  // cells inside the packed buffer
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(like.device());
    it = cells.emplace(key, at::zeros({kCellsBytes / 8}, like.options().dtype(at::kLong))).first;
  }
  return it->second;
}

// Frees the cell sets of (device, stream) -- device < 0: every device, all_streams: every stream.  For streams that are gone or
// whose captured graphs are gone; a later call on such a stream simply allocates a fresh zero-filled set.  Returns the sets freed.
static int64_t release_stream_cells(int device, int64_t stream, bool all_streams) {
  std::vector<at::Tensor> dead;  // freed outside the lock
  {
    std::lock_guard<std::mutex> lock(g_cells_mu);
    for (auto it = g_cells.begin(); it != g_cells.end();) {
      if ((device < 0 || it->first.first == device) && (all_streams || it->first.second == (void*)(intptr_t)stream)) {
        dead.push_back(std::move(it->second));
        it = g_cells.erase(it);
      } else {
        ++it;
      }
    }
  }
  return (int64_t)dead.size();
}

static void* stream_cells(const at::Tensor& like, void* stream, int64_t need_bytes) {
  const at::Tensor t = stream_cells_tensor(like, stream, need_bytes);
  return t.defined() ? t.data_ptr() : nullptr;
}

class FastGroup;

class FastLinear {
 public:
  // params: the module's _parameters dict (codes, codebooks, scales, bias).  packed / desc_bytes: the prepacked buffer
  // and the bytes of its aqlm_hip_packed_desc (kind 0), or the planar 8x8 codes and the 4 bytes of their codebook bound (kind 3:
  // single-row look-up-table matvec, aqlm_hip_gemv_8x8_lut_planar; the reference reaches Triton here, kernel_selector.py:91-94).
  FastLinear(py::dict params, int kind, c10::optional<at::Tensor> packed, std::string desc_bytes, int64_t in_features,
             int64_t out_features, int64_t num_codebooks, int64_t in_group_size, bool watch_codes, int64_t max_rows,
             int64_t fused_8x8_from_rows = 0)
      : params_(std::move(params)), kind_(kind), in_(in_features), out_(out_features), K_((int)num_codebooks),
        g_((int)in_group_size), watch_codes_(watch_codes),
        max_rows_(max_rows < AQLM_HIP_MAX_GEMV_BATCH ? max_rows : AQLM_HIP_MAX_GEMV_BATCH), fused_8x8_from_rows_(fused_8x8_from_rows) {
    w_codes_ = watch(params_, "codes", &codes_);
    w_cb_ = watch(params_, "codebooks", &codebooks_);
    w_scales_ = watch(params_, "scales", &scales_);
    at::Tensor b;
    w_bias_ = watch(params_, "bias", &b);
    if (b.defined()) bias_ = b;
    TORCH_CHECK(codebooks_.defined() && scales_.defined(), "FastLinear: module without codebooks / scales");
    TORCH_CHECK(codebooks_.is_cuda() && codebooks_.is_contiguous() && scales_.is_contiguous(), "FastLinear: parameters must be contiguous device tensors");
    TORCH_CHECK(codebooks_.scalar_type() == at::kHalf || codebooks_.scalar_type() == at::kBFloat16, "FastLinear: fp16 / bf16 only");
    TORCH_CHECK(scales_.scalar_type() == codebooks_.scalar_type() && (!bias_ || bias_->scalar_type() == codebooks_.scalar_type()),
                "FastLinear: scales / bias dtype must match the codebooks");
    dtype_ = codebooks_.scalar_type() == at::kHalf ? AQLM_HIP_F16 : AQLM_HIP_BF16;
    if (kind_ == kPacked1x16) {
      TORCH_CHECK(packed && packed->is_cuda() && desc_bytes.size() == sizeof(aqlm_hip_packed_desc), "FastLinear: packed buffer + descriptor required");
      packed_ = *packed;
      std::memcpy(&desc_, desc_bytes.data(), sizeof(desc_));
      TORCH_CHECK(desc_.codebook_absmax > 0.f, "FastLinear: the packed lane needs the codebook range (single-kernel finalize)");
    } else if (kind_ == kLutPlanar8x8) {
      TORCH_CHECK(packed && packed->is_cuda() && desc_bytes.size() == sizeof(float), "FastLinear: planar codes + codebook bound required");
      packed_ = *packed;
      std::memcpy(&absmax_, desc_bytes.data(), sizeof(float));
      TORCH_CHECK(absmax_ > 0.f && K_ == 8, "FastLinear: the look-up-table lane needs 8 codebooks and a positive codebook bound");
      // 8x8 g32 with its checkpoint-layout codes still there: from fused_8x8_from_rows rows on the lane launches the fused dequant -> MFMA
      // kernel (aqlm_hip_gemm_8x8_mfma) instead of the table kernel once per row
      if (fused_8x8_from_rows_ > 0 && !(watch_codes_ && g_ == 32 && codes_.defined() && codes_.is_cuda() && codes_.is_contiguous() && codes_.numel() > 0))
        fused_8x8_from_rows_ = 0;
    } else {
      TORCH_CHECK(codes_.defined() && codes_.is_cuda() && codes_.is_contiguous(), "FastLinear: canonical codes required");
    }
  }

  // None when the call is not for this lane (the Python path decides what to do).
  py::object forward(const at::Tensor& x) {
    if (!x.is_cuda() || x.scalar_type() != codebooks_.scalar_type() || x.device() != codebooks_.device() || x.dim() < 1 ||
        x.size(-1) != in_ || (x.requires_grad() && at::GradMode::is_enabled()))
      return py::none();
    const int64_t rows = in_ ? x.numel() / in_ : 0;
    if (rows < 1 || rows > max_rows_) return py::none();
    if (!is_current()) return py::none();
    at::Tensor x2 = x.reshape({rows, in_});
    if (x2.stride(1) != 1 || (rows > 1 && x2.stride(0) % 8 != 0) || (reinterpret_cast<uintptr_t>(x2.data_ptr()) & 15u)) x2 = x2.contiguous();
    at::Tensor y = at::empty({rows, out_}, x.options());
    // PyTorch-ROCm tensors carry the device type "cuda": the guard and stream accessors of that naming
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
    void* stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x.device().index()).stream();
    const void* bias = bias_ ? bias_->data_ptr() : nullptr;
    void* cells = (kind_ == kPacked1x16 || kind_ == kLutPlanar8x8) ? stream_cells(x, stream, rows * out_ * 8) : nullptr;
    if (kind_ == kLutPlanar8x8 && !cells) return py::none();  // (a capture on a stream without cells: the Python path's two-kernel form)
    int rc;
    {
      py::gil_scoped_release nogil;
      if (kind_ == kLutPlanar8x8 && fused_8x8_from_rows_ > 0 && rows >= fused_8x8_from_rows_)
        rc = aqlm_hip_gemm_8x8_mfma(codes_.data_ptr(), codebooks_.data_ptr(), scales_.data_ptr(), bias, x2.data_ptr(), y.data_ptr(), (int)rows,
                                    (int)out_, (int)in_, g_, x2.stride(0), out_, dtype_, stream);
      else if (kind_ == kLutPlanar8x8 && rows > 1)  // one launch of rows x the single-row workgroups
        rc = aqlm_hip_gemv_8x8_lut_batch(packed_.data_ptr(), codebooks_.data_ptr(), scales_.data_ptr(), bias, x2.data_ptr(), y.data_ptr(),
                                         (int)out_, (int)in_, g_, (int)rows, x2.stride(0), out_, dtype_, 1, absmax_, cells, (size_t)kCellsBytes, stream);
      else if (kind_ == kLutPlanar8x8)
        rc = aqlm_hip_gemv_8x8_lut_planar(packed_.data_ptr(), codebooks_.data_ptr(), scales_.data_ptr(), bias, x2.data_ptr(), y.data_ptr(),
                                          (int)out_, (int)in_, g_, dtype_, absmax_, cells, (size_t)kCellsBytes, 1, stream);
      else if (kind_ == kPacked1x16 && cells)
        rc = aqlm_hip_gemv_1x16_packed_cells(&desc_, packed_.data_ptr(), codebooks_.data_ptr(), scales_.data_ptr(), bias, x2.data_ptr(),
                                             y.data_ptr(), (int)rows, x2.stride(0), out_, dtype_, cells, (size_t)kCellsBytes, stream);
      else if (kind_ == kPacked1x16)
        rc = aqlm_hip_gemv_1x16_packed(&desc_, packed_.data_ptr(), codebooks_.data_ptr(), scales_.data_ptr(), bias, x2.data_ptr(),
                                       y.data_ptr(), (int)rows, x2.stride(0), out_, dtype_, nullptr, 0, stream);
      else if (kind_ == kGemv1x16)
        rc = aqlm_hip_gemv_1x16(codes_.data_ptr(), codebooks_.data_ptr(), scales_.data_ptr(), bias, x2.data_ptr(), y.data_ptr(),
                                (int)out_, (int)in_, g_, (int)rows, x2.stride(0), out_, dtype_, stream);
      else
        rc = aqlm_hip_gemv_kx8(codes_.data_ptr(), codebooks_.data_ptr(), scales_.data_ptr(), bias, x2.data_ptr(), y.data_ptr(),
                               (int)out_, (int)in_, K_, g_, (int)rows, x2.stride(0), out_, dtype_, stream);
    }
    if (rc != 0) return py::none();  // the Python path repeats the call and reports the error (or serves it another way)
    std::vector<int64_t> shape(x.sizes().begin(), x.sizes().end());
    shape.back() = out_;
    return py::cast(y.view(shape));
  }

  // the module's parameters are the objects (and versions) this lane was built from
  bool is_current() const {
    return unchanged(params_, "codebooks", w_cb_) && unchanged(params_, "scales", w_scales_) && unchanged(params_, "bias", w_bias_) &&
           (!watch_codes_ || unchanged(params_, "codes", w_codes_));
  }

  int kind() const { return kind_; }

 private:
  friend class FastGroup;
  py::dict params_;
  int kind_;
  int64_t in_, out_;
  int K_, g_;
  bool watch_codes_;
  int64_t max_rows_;
  int64_t fused_8x8_from_rows_ = 0;
  int dtype_ = 0;
  at::Tensor codes_, codebooks_, scales_, packed_;
  c10::optional<at::Tensor> bias_;
  Watched w_codes_, w_cb_, w_scales_, w_bias_;
  aqlm_hip_packed_desc desc_{};
  float absmax_ = 0.f;
};

// Shared-input launch of 2..AQLM_HIP_MAX_SEGMENTS members of one kind (q/k/v, gate/up; aqlm_amd/fusion.py): one check of x, one
// allocation per output, ONE launch of the kind's multi entry -- aqlm_hip_gemv_1x16_packed_multi_cells (the pipelined kernel
// where it applies), aqlm_hip_gemv_1x16_multi or aqlm_hip_gemv_kx8_multi.  The
// parking of the siblings' outputs stays in Python (fusion.SharedInputGroup); this is only its launch.
class FastGroup {
 public:
  explicit FastGroup(std::vector<std::shared_ptr<FastLinear>> members) : m_(std::move(members)) {
    TORCH_CHECK(m_.size() >= 2 && m_.size() <= (size_t)AQLM_HIP_MAX_SEGMENTS, "FastGroup: 2..", AQLM_HIP_MAX_SEGMENTS, " members");
    for (const auto& f : m_) {
      TORCH_CHECK(f && f->kind_ == m_[0]->kind_, "FastGroup: members of one kind (prepacked 1x16, direct 1x16 or K x 8)");
      TORCH_CHECK(f->in_ == m_[0]->in_ && f->dtype_ == m_[0]->dtype_ && f->codebooks_.device() == m_[0]->codebooks_.device() &&
                      f->K_ == m_[0]->K_ && f->g_ == m_[0]->g_,
                  "FastGroup: members must agree on in_features, scheme, dtype and device");
    }
  }

  // list of outputs (member order), or None when the call is not for this lane
  py::object forward(const at::Tensor& x) {
    FastLinear& a = *m_[0];
    if (!x.is_cuda() || x.scalar_type() != a.codebooks_.scalar_type() || x.device() != a.codebooks_.device() || x.dim() < 1 ||
        x.size(-1) != a.in_ || (x.requires_grad() && at::GradMode::is_enabled()))
      return py::none();
    const int64_t rows = a.in_ ? x.numel() / a.in_ : 0;
    if (rows < 1 || rows > a.max_rows_) return py::none();
    if (a.kind_ == kLutPlanar8x8 && rows > 1) return py::none();  // the shared-input table kernel takes one row of x: Python launches the layers one by one
    int64_t total = 0;
    for (const auto& f : m_) {
      if (!f->is_current()) return py::none();
      total += f->out_;
    }
    at::Tensor x2 = x.reshape({rows, a.in_});
    if (x2.stride(1) != 1 || (rows > 1 && x2.stride(0) % 8 != 0) || (reinterpret_cast<uintptr_t>(x2.data_ptr()) & 15u)) x2 = x2.contiguous();
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
    void* stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x.device().index()).stream();
    const bool needs_cells = a.kind_ == kPacked1x16 || a.kind_ == kLutPlanar8x8;
    void* cells = needs_cells ? stream_cells(x, stream, rows * total * 8) : nullptr;
    if (needs_cells && !cells) return py::none();
    const int n = (int)m_.size();
    aqlm_hip_segment seg[AQLM_HIP_MAX_SEGMENTS];
    const aqlm_hip_packed_desc* descs[AQLM_HIP_MAX_SEGMENTS];
    float absmax[AQLM_HIP_MAX_SEGMENTS];
    std::vector<at::Tensor> ys;
    ys.reserve(n);
    for (int k = 0; k < n; ++k) {
      FastLinear& f = *m_[k];
      ys.push_back(at::empty({rows, f.out_}, x.options()));
      seg[k].codes = needs_cells ? f.packed_.data_ptr() : f.codes_.data_ptr();
      absmax[k] = f.absmax_;
      seg[k].codebook = f.codebooks_.data_ptr();
      seg[k].scales = f.scales_.data_ptr();
      seg[k].bias = f.bias_ ? f.bias_->data_ptr() : nullptr;
      seg[k].y = ys[k].data_ptr();
      seg[k].y_row_stride = f.out_;
      seg[k].out_features = (int)f.out_;
      seg[k].reserved = 0;
      descs[k] = &f.desc_;
    }
    int rc;
    {
      py::gil_scoped_release nogil;
      if (a.kind_ == kLutPlanar8x8)
        rc = aqlm_hip_gemv_8x8_lut_planar_multi(seg, absmax, n, x2.data_ptr(), (int)a.in_, a.g_, a.dtype_, cells, (size_t)kCellsBytes, 1, stream);
      else if (a.kind_ == kPacked1x16)
        rc = aqlm_hip_gemv_1x16_packed_multi_cells(seg, descs, n, x2.data_ptr(), (int)a.in_, (int)rows, x2.stride(0), a.dtype_, cells,
                                                   (size_t)kCellsBytes, stream);
      else if (a.kind_ == kGemv1x16)
        rc = aqlm_hip_gemv_1x16_multi(seg, n, x2.data_ptr(), (int)a.in_, a.g_, (int)rows, x2.stride(0), a.dtype_, stream);
      else
        rc = aqlm_hip_gemv_kx8_multi(seg, n, x2.data_ptr(), (int)a.in_, a.K_, a.g_, (int)rows, x2.stride(0), a.dtype_, stream);
    }
    if (rc != 0) return py::none();
    py::list out;
    std::vector<int64_t> shape(x.sizes().begin(), x.sizes().end());
    for (int k = 0; k < n; ++k) {
      shape.back() = m_[k]->out_;
      out.append(py::cast(ys[k].view(shape)));
    }
    return std::move(out);
  }

 private:
  std::vector<std::shared_ptr<FastLinear>> m_;
};


// ---------------------------------------------------------------------------------------------------------------------------
// The reference's stateless ops as compiled dispatcher kernels: aqlm::code1x16_matmat, code2x8_matmat, code1x8_matmat
// (cuda_kernel.cpp:148-182, 387-421, 552-586 are C++ too; its benchmark/matmul_benchmark.py:103 and vLLM-style integrations call
// them directly).  Decode calls -- 1..6 rows (1..8 for K x 8) of a well-formed layer -- are launched from here; everything else
// (more rows -> MFMA / dequant + GEMM, an unknown layer that may want packing, any malformed argument and its error message)
// is handed to the Python implementation of the same op (hip_kernel.py), which stays the definition of the behaviour.
//
// Large 1x16 layers run on the prepacked kernel.  The Python side owns that cache (which layer is packed, eviction, weak
// references to the codes tensors); after it packed a layer it REGISTERS the result here under the identity of the codes
// tensor, and forgets it here when it drops it there.  A hit is validated against (storage pointer, version, shape) of the
// codes and (storage pointer, version) of the codebooks the range was computed from.
struct RawEntry {
  const void* codes_data = nullptr;
  uint32_t codes_version = 0;
  bool codes_versioned = false;
  int64_t out = 0, in_groups = 0;
  const void* cb_data = nullptr;
  uint32_t cb_version = 0;
  bool cb_versioned = false;
  int64_t hits_since_check = 0;
  at::Tensor packed;
  aqlm_hip_packed_desc desc{};
};

static std::mutex g_raw_mu;
static std::unordered_map<const void*, RawEntry> g_raw;  // key: the codes tensor's TensorImpl
// (the dispatcher kernels run without the GIL and from any number of host threads: the mirrors and counters are atomics)
static std::atomic<bool> g_raw_on{true};            // false: every call takes the Python implementation (set_fused_finalize(False), experiments)
static std::atomic<bool> g_raw_prepack{true};       // mirror of hip_kernel.RAW_OP_PREPACK
static std::atomic<int64_t> g_raw_min_codes{500000};  // mirror of hip_kernel.RAW_OP_PREPACK_MIN_CODES
static std::atomic<int64_t> g_raw_gemm_rows{7};     // mirror of hip_kernel.MATMAT_GEMM_MIN_ROWS
static std::atomic<int64_t> g_raw_check_every{256};  // mirror of hip_kernel.RAW_OP_CHECK_EVERY: hits of a registered layer between two visits to Python, which re-checks the checksums of its codes / codebook (writes through `.data` change neither identity nor version)
static PyObject* g_raw_py[3] = {nullptr, nullptr, nullptr};  // Python implementations (leaked on purpose: they outlive the interpreter's teardown order)
static std::atomic<uint64_t> g_raw_served{0};
static std::atomic<uint64_t> g_raw_hits{0};  // calls served from a registered (prepacked) layer: the Python cache reads this as its hit count

static bool versioned(const at::Tensor& t) { return !t.is_inference(); }

static at::Tensor raw_python(int which, const at::Tensor& input, const at::Tensor& codes, const at::Tensor& codebooks,
                             const at::Tensor& scales, const c10::optional<at::Tensor>& bias) {
  py::gil_scoped_acquire gil;
  TORCH_CHECK(g_raw_py[which] != nullptr, "aqlm raw op: no Python implementation installed");
  py::object fn = py::reinterpret_borrow<py::object>(g_raw_py[which]);
  py::object b = bias ? py::cast(*bias) : py::object(py::none());
  return fn(input, codes, codebooks, scales, b).cast<at::Tensor>();
}

struct RawCall {  // what every launch needs once the arguments have been accepted
  at::Tensor x2, y;
  int64_t rows = 0;
  void* stream = nullptr;
  int dtype = 0;
};

// common argument checks of the three ops; false = not a call for this lane
static bool raw_accept(const at::Tensor& input, const at::Tensor& codes, const at::Tensor& codebooks, const at::Tensor& scales,
                       const c10::optional<at::Tensor>& bias, at::ScalarType code_type, int64_t K, int64_t codebook_size, int64_t max_rows,
                       RawCall& c, int64_t& in_features, int64_t& out_features, int& g) {
  if (!g_raw_on.load(std::memory_order_relaxed) || !input.is_cuda() || input.dim() < 1) return false;
  const at::ScalarType st = input.scalar_type();
  if (st != at::kHalf && st != at::kBFloat16) return false;
  if (codebooks.scalar_type() != st || scales.scalar_type() != st || (bias && bias->scalar_type() != st)) return false;
  const auto dev = input.device();
  if (codes.device() != dev || codebooks.device() != dev || scales.device() != dev || (bias && bias->device() != dev)) return false;
  if (codebooks.dim() != 4 || codebooks.size(0) != K || codebooks.size(1) != codebook_size || codebooks.size(2) != 1) return false;
  if (codes.dim() != 3 || codes.size(2) != K || codes.scalar_type() != code_type) return false;
  if (!codes.is_contiguous() || !codebooks.is_contiguous() || !scales.is_contiguous() || (bias && !bias->is_contiguous())) return false;
  g = (int)codebooks.size(3);
  out_features = codes.size(0);
  in_features = codes.size(1) * g;
  if (out_features < 1 || in_features < 1 || input.size(-1) != in_features || scales.numel() != out_features ||
      (bias && bias->numel() != out_features))
    return false;
  if (input.requires_grad() && at::GradMode::is_enabled()) return false;  // (the op has no autograd formula either way; Python decides)
  c.rows = input.numel() / in_features;
  if (c.rows < 1 || c.rows > max_rows) return false;
  c.dtype = st == at::kHalf ? AQLM_HIP_F16 : AQLM_HIP_BF16;
  return true;
}

static void raw_prepare(const at::Tensor& input, int64_t in_features, int64_t out_features, RawCall& c) {
  c.x2 = input.reshape({c.rows, in_features});
  if (c.x2.stride(1) != 1 || (c.rows > 1 && c.x2.stride(0) % 8 != 0) || (reinterpret_cast<uintptr_t>(c.x2.data_ptr()) & 15u)) c.x2 = c.x2.contiguous();
  c.y = at::empty({c.rows, out_features}, input.options());
  c.stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(input.device().index()).stream();
}

static at::Tensor raw_result(const at::Tensor& input, const RawCall& c, int64_t out_features) {
  std::vector<int64_t> shape(input.sizes().begin(), input.sizes().end());
  shape.back() = out_features;
  g_raw_served.fetch_add(1, std::memory_order_relaxed);
  return c.y.view(shape);
}

static at::Tensor raw_code1x16_matmat(const at::Tensor& input, const at::Tensor& codes, const at::Tensor& codebooks,
                                      const at::Tensor& scales, const c10::optional<at::Tensor>& bias) {
  RawCall c;
  int64_t in_features = 0, out_features = 0;
  int g = 0;
  if (!raw_accept(input, codes, codebooks, scales, bias, at::kShort, 1, 65536, g_raw_gemm_rows.load(std::memory_order_relaxed) - 1, c, in_features, out_features, g) ||
      (g != 8 && g != 16))
    return raw_python(0, input, codes, codebooks, scales, bias);
  const void* bias_p = bias ? bias->data_ptr() : nullptr;
  // a layer the Python side has packed and registered?
  at::Tensor packed;
  aqlm_hip_packed_desc desc{};
  bool hit = false;
  {
    std::lock_guard<std::mutex> lock(g_raw_mu);
    auto it = g_raw.find((const void*)codes.unsafeGetTensorImpl());
    if (it != g_raw.end()) {
      RawEntry& e = it->second;
      hit = e.codes_data == codes.data_ptr() && e.out == out_features && e.in_groups == codes.size(1) &&
            e.codes_versioned == versioned(codes) && (!e.codes_versioned || e.codes_version == codes._version()) &&
            e.cb_data == codebooks.data_ptr() && e.cb_versioned == versioned(codebooks) &&
            (!e.cb_versioned || e.cb_version == codebooks._version());
      const int64_t every = g_raw_check_every.load(std::memory_order_relaxed);
      if (hit && every > 0 && ++e.hits_since_check >= every && c10::hip::currentStreamCaptureStatusMayInitCtx() == c10::hip::CaptureStatus::None) {
        e.hits_since_check = 0;
        hit = false;  // this call goes to Python, which verifies the layer's checksums (and re-registers it if it had to repack)
      }
      if (hit) {
        packed = e.packed;
        desc = e.desc;
      }
    }
  }
  const bool direct = !hit && (!g_raw_prepack.load(std::memory_order_relaxed) || out_features * codes.size(1) < g_raw_min_codes.load(std::memory_order_relaxed));
  if (!hit && !direct) return raw_python(0, input, codes, codebooks, scales, bias);  // Python packs (or not) and registers
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
  raw_prepare(input, in_features, out_features, c);
  int rc;
  if (hit) {
    g_raw_hits.fetch_add(1, std::memory_order_relaxed);
    void* cells = stream_cells(input, c.stream, c.rows * out_features * 8);
    if (cells)
      rc = aqlm_hip_gemv_1x16_packed_cells(&desc, packed.data_ptr(), codebooks.data_ptr(), scales.data_ptr(), bias_p, c.x2.data_ptr(),
                                           c.y.data_ptr(), (int)c.rows, c.x2.stride(0), out_features, c.dtype, cells, (size_t)kCellsBytes,
                                           c.stream);
    else
      rc = aqlm_hip_gemv_1x16_packed(&desc, packed.data_ptr(), codebooks.data_ptr(), scales.data_ptr(), bias_p, c.x2.data_ptr(),
                                     c.y.data_ptr(), (int)c.rows, c.x2.stride(0), out_features, c.dtype, nullptr, 0, c.stream);
  } else {
    rc = aqlm_hip_gemv_1x16(codes.data_ptr(), codebooks.data_ptr(), scales.data_ptr(), bias_p, c.x2.data_ptr(), c.y.data_ptr(),
                            (int)out_features, (int)in_features, g, (int)c.rows, c.x2.stride(0), out_features, c.dtype, c.stream);
  }
  if (rc != 0) return raw_python(0, input, codes, codebooks, scales, bias);  // Python repeats the call and reports the error
  return raw_result(input, c, out_features);
}

template <int K, int WHICH>
static at::Tensor raw_codekx8_matmat(const at::Tensor& input, const at::Tensor& codes, const at::Tensor& codebooks,
                                     const at::Tensor& scales, const c10::optional<at::Tensor>& bias) {
  RawCall c;
  int64_t in_features = 0, out_features = 0;
  int g = 0;
  if (!raw_accept(input, codes, codebooks, scales, bias, at::kChar, K, 256, AQLM_HIP_MAX_GEMV_BATCH, c, in_features, out_features, g))
    return raw_python(WHICH, input, codes, codebooks, scales, bias);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
  raw_prepare(input, in_features, out_features, c);
  const int rc = aqlm_hip_gemv_kx8(codes.data_ptr(), codebooks.data_ptr(), scales.data_ptr(), bias ? bias->data_ptr() : nullptr,
                                   c.x2.data_ptr(), c.y.data_ptr(), (int)out_features, (int)in_features, K, g, (int)c.rows, c.x2.stride(0),
                                   out_features, c.dtype, c.stream);
  if (rc != 0) return raw_python(WHICH, input, codes, codebooks, scales, bias);
  return raw_result(input, c, out_features);
}

// Installs the three kernels for the CUDA dispatch key (the schemas are defined by hip_kernel.py, which then registers no
// Python kernel for these names).  `impls`: the Python implementations, the fallback of every call not served here.
static void raw_install(py::object code1x16, py::object code2x8, py::object code1x8) {
  static torch::Library* lib = nullptr;
  TORCH_CHECK(lib == nullptr, "aqlm raw ops: already installed");
  g_raw_py[0] = code1x16.release().ptr();
  g_raw_py[1] = code2x8.release().ptr();
  g_raw_py[2] = code1x8.release().ptr();
  lib = new torch::Library(torch::Library::IMPL, "aqlm", c10::make_optional(c10::DispatchKey::CUDA), __FILE__, __LINE__);  // never destroyed
  lib->impl("code1x16_matmat", TORCH_FN(raw_code1x16_matmat));
  lib->impl("code2x8_matmat", TORCH_FN((raw_codekx8_matmat<2, 1>)));
  lib->impl("code1x8_matmat", TORCH_FN((raw_codekx8_matmat<1, 2>)));
}

static int64_t raw_register(const at::Tensor& codes, const at::Tensor& packed, const std::string& desc_bytes, const at::Tensor& codebooks) {
  TORCH_CHECK(desc_bytes.size() == sizeof(aqlm_hip_packed_desc) && packed.is_cuda() && codes.dim() == 3, "aqlm raw ops: bad registration");
  RawEntry e;
  std::memcpy(&e.desc, desc_bytes.data(), sizeof(e.desc));
  if (!(e.desc.codebook_absmax > 0.f)) return 0;  // the single-kernel finalize needs the codebook range: stay on the Python path
  e.codes_data = codes.data_ptr();
  e.codes_versioned = versioned(codes);
  e.codes_version = e.codes_versioned ? codes._version() : 0;
  e.out = codes.size(0);
  e.in_groups = codes.size(1);
  e.cb_data = codebooks.data_ptr();
  e.cb_versioned = versioned(codebooks);
  e.cb_version = e.cb_versioned ? codebooks._version() : 0;
  e.packed = packed;
  const void* key = (const void*)codes.unsafeGetTensorImpl();
  std::lock_guard<std::mutex> lock(g_raw_mu);
  g_raw[key] = std::move(e);
  return (int64_t)(intptr_t)key;
}

static void raw_forget(int64_t key) {
  at::Tensor keep;  // the packed buffer is released outside the lock
  std::lock_guard<std::mutex> lock(g_raw_mu);
  auto it = g_raw.find((const void*)(intptr_t)key);
  if (it != g_raw.end()) {
    keep = std::move(it->second.packed);
    g_raw.erase(it);
  }
}

static void raw_clear() {
  std::unordered_map<const void*, RawEntry> old;
  std::lock_guard<std::mutex> lock(g_raw_mu);
  old.swap(g_raw);
}

static void raw_config(bool on, bool prepack, int64_t min_codes, int64_t gemm_rows, int64_t check_every) {
  g_raw_check_every = check_every;
  std::lock_guard<std::mutex> lock(g_raw_mu);
  g_raw_on = on;
  g_raw_prepack = prepack;
  g_raw_min_codes = min_codes;
  g_raw_gemm_rows = gemm_rows;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled host glue of aqlm_amd's decode path (see aqlm_amd/csrc_front/front.cpp)";
  m.attr("ABI_VERSION") = AQLM_HIP_ABI_VERSION;
  py::class_<FastLinear, std::shared_ptr<FastLinear>>(m, "FastLinear")
      .def(py::init<py::dict, int, c10::optional<at::Tensor>, std::string, int64_t, int64_t, int64_t, int64_t, bool, int64_t, int64_t>(), py::arg("params"),
           py::arg("kind"), py::arg("packed"), py::arg("desc_bytes"), py::arg("in_features"), py::arg("out_features"),
           py::arg("num_codebooks"), py::arg("in_group_size"), py::arg("watch_codes"), py::arg("max_rows"), py::arg("fused_8x8_from_rows") = 0)
      .def("is_current", &FastLinear::is_current)
      .def("forward", &FastLinear::forward)
      .def("__call__", &FastLinear::forward)
      .def_property_readonly("kind", &FastLinear::kind);
  py::class_<FastGroup>(m, "FastGroup")
      .def(py::init<std::vector<std::shared_ptr<FastLinear>>>(), py::arg("members"))
      .def("forward", &FastGroup::forward)
      .def("__call__", &FastGroup::forward);
  // the same kernels as plain functions (what the reference's pybind module exposes, cuda_kernel.cpp:686-699, and its benchmark calls,
  // benchmark/matmul_benchmark.py:103): no dispatcher in between
  m.def("code1x16_matmat", &raw_code1x16_matmat, py::arg("input"), py::arg("codes"), py::arg("codebooks"), py::arg("scales"), py::arg("bias") = py::none());
  m.def("code2x8_matmat", &raw_codekx8_matmat<2, 1>, py::arg("input"), py::arg("codes"), py::arg("codebooks"), py::arg("scales"), py::arg("bias") = py::none());
  m.def("code1x8_matmat", &raw_codekx8_matmat<1, 2>, py::arg("input"), py::arg("codes"), py::arg("codebooks"), py::arg("scales"), py::arg("bias") = py::none());
  m.def("release_stream_cells", &release_stream_cells, py::arg("device"), py::arg("stream"), py::arg("all_streams"));
  m.def("stream_cells", [](const at::Tensor& like, int64_t stream, int64_t need_bytes) -> py::object {
    const at::Tensor t = stream_cells_tensor(like, (void*)(intptr_t)stream, need_bytes);
    return t.defined() ? py::cast(t) : py::object(py::none());
  }, "the zero-at-rest accumulator cells of (device of `like`, stream): int64 tensor, or None (too large / first use inside a capture)");
  m.def("raw_install", &raw_install, "register the compiled kernels of aqlm::code1x16_matmat / code2x8_matmat / code1x8_matmat (CUDA key)");
  m.def("raw_register", &raw_register, "a packed layer of the raw op's cache -> key");
  m.def("raw_forget", &raw_forget);
  m.def("raw_clear", &raw_clear);
  m.def("raw_config", &raw_config, py::arg("on"), py::arg("prepack"), py::arg("min_codes"), py::arg("gemm_rows"), py::arg("check_every") = 256);
  m.def("raw_served", []() { return g_raw_served.load(); }, "calls launched by the compiled raw ops so far");
  m.def("raw_hits", []() { return g_raw_hits.load(); }, "calls the compiled code1x16_matmat served from a registered prepacked layer");
  m.def("raw_entries", []() { std::lock_guard<std::mutex> lock(g_raw_mu); return g_raw.size(); });
}
