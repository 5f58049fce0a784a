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

#include <map>
#include <mutex>

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
// (_packed_cells): the packed buffers are only read, so a layer may run on several streams at once.  Never freed.
constexpr int64_t kCellsBytes = (int64_t)AQLM_HIP_MAX_GEMV_BATCH * 131072 * 8;

static void* stream_cells(const at::Tensor& like, void* stream, int64_t need_bytes) {
  if (need_bytes > kCellsBytes) return nullptr;
  static std::mutex mu;
  static std::map<std::pair<int, void*>, at::Tensor> cells;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_pair((int)like.device().index(), stream);
  auto it = cells.find(key);
  if (it == cells.end()) {
    if (c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None) return nullptr;  // cells inside the packed buffer
    it = cells.emplace(key, at::zeros({kCellsBytes / 8}, like.options().dtype(at::kLong))).first;
  }
  return it->second.data_ptr();
}

class FastGroup;

class FastLinear {
 public:
  // params: the module's _parameters dict (codes, codebooks, scales, bias).  packed / desc_bytes: the prepacked buffer
  // and the bytes of its aqlm_hip_packed_desc (kind 0), or the planar 8x8 codes and the 4 bytes of their codebook bound (kind 3:
  // single-row look-up-table matvec, aqlm_hip_gemv_8x8_lut_planar; the reference reaches Triton here, kernel_selector.py:91-94).
  FastLinear(py::dict params, int kind, c10::optional<at::Tensor> packed, std::string desc_bytes, int64_t in_features,
             int64_t out_features, int64_t num_codebooks, int64_t in_group_size, bool watch_codes, int64_t max_rows)
      : params_(std::move(params)), kind_(kind), in_(in_features), out_(out_features), K_((int)num_codebooks),
        g_((int)in_group_size), watch_codes_(watch_codes),
        max_rows_(max_rows < AQLM_HIP_MAX_GEMV_BATCH ? max_rows : AQLM_HIP_MAX_GEMV_BATCH) {
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
      max_rows_ = 1;
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
      if (kind_ == kLutPlanar8x8)
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

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled host glue of aqlm_amd's decode path (see aqlm_amd/csrc_front/front.cpp)";
  m.attr("ABI_VERSION") = AQLM_HIP_ABI_VERSION;
  py::class_<FastLinear, std::shared_ptr<FastLinear>>(m, "FastLinear")
      .def(py::init<py::dict, int, c10::optional<at::Tensor>, std::string, int64_t, int64_t, int64_t, int64_t, bool, int64_t>(), py::arg("params"),
           py::arg("kind"), py::arg("packed"), py::arg("desc_bytes"), py::arg("in_features"), py::arg("out_features"),
           py::arg("num_codebooks"), py::arg("in_group_size"), py::arg("watch_codes"), py::arg("max_rows"))
      .def("is_current", &FastLinear::is_current)
      .def("forward", &FastLinear::forward)
      .def("__call__", &FastLinear::forward)
      .def_property_readonly("kind", &FastLinear::kind);
  py::class_<FastGroup>(m, "FastGroup")
      .def(py::init<std::vector<std::shared_ptr<FastLinear>>>(), py::arg("members"))
      .def("forward", &FastGroup::forward)
      .def("__call__", &FastGroup::forward);
}
