// fgmm_pybind.cpp — `flashgmm_amd._native`: the compiled Python boundary over the C ABI (include/flashgmm_amd.h).
//
// Mirror of the reference's pybind11 module definition, compressai/cpp_exts/rans/rans_interface.cpp:961-1036, for the batched entropy-model
// calls: tensors arrive as device addresses (`Tensor.data_ptr()`), the `fgmm_item` array is built here (no ctypes field marshalling, no
// numpy record arrays), the GIL is released across the native call - the reference holds it (no gil_scoped_release anywhere in its
// module) - and the bitstreams come back as `bytes` objects allocated at their final size and filled by the context's host workers
// (fgmm_ctx_take_buffers: one copy, no intermediate buffer object).  Nothing here computes: every function is argument plumbing around
// fgmm_gmc_compress_batch / fgmm_gmc_decompress_batch / fgmm_gmc_compress_head_batch.  flashgmm_amd/_lib.py (ctypes) binds the same
// ABI and stays the fallback.
#include <pybind11/pybind11.h>

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/flashgmm_amd.h"

namespace py = pybind11;

namespace {

const char *status_name(int rc) {
  switch (rc) {
  case FGMM_ERR_INVALID: return "FGMM_ERR_INVALID";
  case FGMM_ERR_NO_DEVICE: return "FGMM_ERR_NO_DEVICE";
  case FGMM_ERR_HIP: return "FGMM_ERR_HIP";
  case FGMM_ERR_NOMEM: return "FGMM_ERR_NOMEM";
  case FGMM_ERR_STREAM: return "FGMM_ERR_STREAM";
  case FGMM_ERR_UNSUPPORTED: return "FGMM_ERR_UNSUPPORTED";
  }
  return "fgmm_status";
}
[[noreturn]] void raise(const char *what, int rc) { // RuntimeError, as pybind11 turns the reference's c10::Error into one
  throw std::runtime_error(std::string(what) + ": " + status_name(rc) + ": " + fgmm_last_error());
}

template <typename T> T *ptr(uintptr_t a) { return reinterpret_cast<T *>(a); }

// N items of one shape given as stacked tensors: item i = base + i * item stride
struct Stacked {
  uintptr_t scales, means, weights; // device; 0 with a fused head
  int64_t item_stride;              // elements between consecutive items of a parameter tensor
  int64_t stride_k, stride_c;
  int dtype, flags;
  int N, M;
  int64_t hw;
};

void fill_items(std::vector<fgmm_item> &it, const Stacked &s) {
  const size_t esz = s.dtype == FGMM_F16 ? 2 : 4;
  for (int i = 0; i < s.N; ++i) {
    fgmm_item &f = it[(size_t)i];
    std::memset(&f, 0, sizeof f);
    const size_t off = (size_t)i * (size_t)s.item_stride * esz;
    f.params.scales = s.scales ? ptr<const void>(s.scales + off) : nullptr;
    f.params.means = s.means ? ptr<const void>(s.means + off) : nullptr;
    f.params.weights = s.weights ? ptr<const void>(s.weights + off) : nullptr;
    f.params.stride_k = s.stride_k, f.params.stride_c = s.stride_c;
    f.params.dtype = s.dtype, f.params.flags = s.flags;
    f.M = s.M, f.K = FGMM_K, f.hw = s.hw;
  }
}

// the bitstreams of a finished compress call -> list of bytes (or of instances of the bytes subclass `cls`: same storage layout),
// filled by the context's workers with the GIL released; the library's buffers are released by that call
py::list take_bytes(fgmm_ctx *ctx, std::vector<fgmm_item> &it, py::handle cls) {
  const size_t n = it.size();
  py::list out(n);
  std::vector<void *> dst(n), src(n);
  std::vector<size_t> len(n);
  py::object bytes_new;
  if (!cls.is_none()) bytes_new = py::reinterpret_borrow<py::object>((PyObject *)&PyBytes_Type).attr("__new__");
  for (size_t i = 0; i < n; ++i) {
    PyObject *o;
    if (cls.is_none()) {
      o = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)it[i].bytes_len); // uninitialised, exactly as long as the bitstream
      if (!o) throw py::error_already_set();
    } else {
      o = bytes_new(cls, py::int_(it[i].bytes_len)).release().ptr(); // bytes.__new__(cls, n): n zero bytes in the instance's own storage
    }
    PyList_SET_ITEM(out.ptr(), (Py_ssize_t)i, o);
    dst[i] = PyBytes_AS_STRING(o), src[i] = it[i].bytes, len[i] = it[i].bytes_len;
  }
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = fgmm_ctx_take_buffers(ctx, dst.data(), src.data(), len.data(), (int)n);
  }
  if (rc) raise("fgmm_ctx_take_buffers", rc);
  return out;
}

// checkpoints of a compress call: ONE bytes object holding every item's notes back to back (16 bytes each) + the counts
py::tuple take_ckpts(fgmm_ctx *ctx, std::vector<fgmm_item> &it) {
  size_t total = 0;
  for (auto &f : it) total += (size_t)f.n_ckpt;
  PyObject *blob = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)(16 * total));
  if (!blob) throw py::error_already_set();
  py::object keep = py::reinterpret_steal<py::object>(blob);
  py::list counts(it.size());
  std::vector<void *> dst, src;
  std::vector<size_t> len;
  char *at = PyBytes_AS_STRING(blob);
  for (size_t i = 0; i < it.size(); ++i) {
    PyList_SET_ITEM(counts.ptr(), (Py_ssize_t)i, PyLong_FromLongLong((long long)it[i].n_ckpt));
    if (it[i].ckpt && it[i].n_ckpt > 0) {
      dst.push_back(at), src.push_back(it[i].ckpt), len.push_back(16 * (size_t)it[i].n_ckpt);
      at += 16 * (size_t)it[i].n_ckpt;
    }
  }
  if (!dst.empty()) {
    const int rc = fgmm_ctx_take_buffers(ctx, dst.data(), src.data(), len.data(), (int)dst.size());
    if (rc) raise("fgmm_ctx_take_buffers", rc);
  }
  for (auto &f : it) f.ckpt = nullptr; // (copied and released by the library)
  return py::make_tuple(keep, counts);
}

// what a finished compress call returned in library-owned buffers, released if this binding does not get to hand it over (an allocation
// that fails half way, a copy that is refused): fgmm_ctx_take_buffers releases what it copies
struct Owned {
  std::vector<fgmm_item> &it;
  ~Owned() {
    for (auto &f : it) {
      fgmm_free(f.bytes), fgmm_free(f.ckpt);
      f.bytes = nullptr, f.ckpt = nullptr;
    }
  }
};

py::tuple finish_compress(fgmm_ctx *ctx, std::vector<fgmm_item> &it, int ckpt_stride, py::handle cls) {
  Owned owned{it};
  py::list strings = take_bytes(ctx, it, cls);
  for (auto &f : it) f.bytes = nullptr; // (copied and released by the library)
  py::list abs_max(it.size());
  for (size_t i = 0; i < it.size(); ++i) PyList_SET_ITEM(abs_max.ptr(), (Py_ssize_t)i, PyLong_FromLong(it[i].abs_max));
  if (ckpt_stride) {
    py::tuple ck = take_ckpts(ctx, it);
    return py::make_tuple(strings, abs_max, ck[0], ck[1]);
  }
  return py::make_tuple(strings, abs_max, py::none(), py::none());
}

// GaussianMixtureConditional.compress for N stacked items (entropy_models.py:833-867, batched):
//   -> (list of N bytes, list of N abs_max, checkpoint blob | None, checkpoint counts | None); y_q and the zero bitmaps are written
//   through yq / zero_bitmap (device float32 [N, M, hw] / HOST int64 [N, M])
py::tuple compress_stacked(uintptr_t ctx_, uintptr_t stream, uintptr_t y, uintptr_t scales, uintptr_t means, uintptr_t weights, int N, int M, int64_t hw,
                           int64_t item_stride, int64_t stride_k, int64_t stride_c, int dtype, int flags, int mode, int clamp_scales, int ckpt_stride,
                           uintptr_t yq, uintptr_t zero_bitmap, py::object bytes_cls) {
  fgmm_ctx *ctx = ptr<fgmm_ctx>(ctx_);
  std::vector<fgmm_item> it((size_t)N);
  fill_items(it, Stacked{scales, means, weights, item_stride, stride_k, stride_c, dtype, flags, N, M, hw});
  for (int i = 0; i < N; ++i) {
    fgmm_item &f = it[(size_t)i];
    f.y = ptr<const float>(y) + (size_t)i * (size_t)M * (size_t)hw;
    f.yq_out = ptr<float>(yq) + (size_t)i * (size_t)M * (size_t)hw;
    f.zero_bitmap = ptr<int64_t>(zero_bitmap) + (size_t)i * (size_t)M;
    f.ckpt_stride = ckpt_stride;
  }
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = fgmm_gmc_compress_batch(ctx, ptr<void>(stream), it.data(), N, mode, clamp_scales);
  }
  if (rc) raise("GaussianMixtureConditional.compress", rc);
  return finish_compress(ctx, it, ckpt_stride, bytes_cls);
}

// ... with the parameter head fused (fgmm_gmc_compress_head_batch): x device float32 [N, c_in, hw]
py::tuple compress_head_stacked(uintptr_t ctx_, uintptr_t stream, uintptr_t y, uintptr_t x, uintptr_t head, int N, int M, int c_in, int64_t hw, int mode,
                                int clamp_scales, int ckpt_stride, uintptr_t yq, uintptr_t zero_bitmap, py::object bytes_cls) {
  fgmm_ctx *ctx = ptr<fgmm_ctx>(ctx_);
  std::vector<fgmm_item> it((size_t)N);
  std::vector<const float *> xs((size_t)N);
  fill_items(it, Stacked{0, 0, 0, 0, 0, 0, FGMM_F32, FGMM_PARAMS_LOGITS, N, M, hw});
  for (int i = 0; i < N; ++i) {
    fgmm_item &f = it[(size_t)i];
    f.y = ptr<const float>(y) + (size_t)i * (size_t)M * (size_t)hw;
    f.yq_out = ptr<float>(yq) + (size_t)i * (size_t)M * (size_t)hw;
    f.zero_bitmap = ptr<int64_t>(zero_bitmap) + (size_t)i * (size_t)M;
    f.ckpt_stride = ckpt_stride;
    xs[(size_t)i] = ptr<const float>(x) + (size_t)i * (size_t)c_in * (size_t)hw;
  }
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = fgmm_gmc_compress_head_batch(ctx, ptr<void>(stream), it.data(), xs.data(), N, ptr<const fgmm_head>(head), mode, clamp_scales);
  }
  if (rc) raise("GaussianMixtureConditional.compress_head_batch", rc);
  return finish_compress(ctx, it, ckpt_stride, bytes_cls);
}

// GaussianMixtureConditional.decompress for N stacked items (entropy_models.py:872-910, batched): the bitstreams are read in place
// (borrowed pointers into the bytes objects, which the caller's list keeps alive across the call); y_hat is written through `y_hat`
// (device float32 [N, M, hw]).  zero_bitmap: HOST int64, row i at zero_bitmap + i * zb_row_stride elements.  ckpt: None, or a sequence
// of N (address, count, stride) triples - the out-of-band notes of the bitstreams that carry them (0, 0, 0 for those that do not).
void decompress_stacked(uintptr_t ctx_, uintptr_t stream, py::sequence strings, py::sequence abs_maxes, uintptr_t zero_bitmap, int64_t zb_row_stride,
                        uintptr_t scales, uintptr_t means, uintptr_t weights, int N, int M, int64_t hw, int64_t item_stride, int64_t stride_k, int64_t stride_c,
                        int dtype, int flags, int mode, int clamp_scales, uintptr_t y_hat, py::object ckpt) {
  fgmm_ctx *ctx = ptr<fgmm_ctx>(ctx_);
  if ((int)py::len(strings) != N || (int)py::len(abs_maxes) != N) throw std::runtime_error("decompress: " + std::to_string(N) + " items in the parameter tensors, " + std::to_string(py::len(strings)) + " bitstreams");
  std::vector<fgmm_item> it((size_t)N);
  fill_items(it, Stacked{scales, means, weights, item_stride, stride_k, stride_c, dtype, flags, N, M, hw});
  for (int i = 0; i < N; ++i) {
    fgmm_item &f = it[(size_t)i];
    py::handle b = strings[(size_t)i];
    if (!PyBytes_Check(b.ptr())) throw std::runtime_error("decompress: bitstream " + std::to_string(i) + " is not a bytes object");
    f.bytes = reinterpret_cast<uint8_t *>(PyBytes_AS_STRING(b.ptr()));
    f.bytes_len = (size_t)PyBytes_GET_SIZE(b.ptr());
    f.abs_max = py::cast<int32_t>(abs_maxes[(size_t)i]);
    f.zero_bitmap = ptr<int64_t>(zero_bitmap) + (size_t)i * (size_t)zb_row_stride;
    f.yq_out = ptr<float>(y_hat) + (size_t)i * (size_t)M * (size_t)hw;
    if (!ckpt.is_none()) {
      py::tuple t = py::cast<py::tuple>(py::cast<py::sequence>(ckpt)[(size_t)i]);
      f.ckpt = ptr<fgmm_ckpt>(py::cast<uintptr_t>(t[0]));
      f.n_ckpt = py::cast<int64_t>(t[1]);
      f.ckpt_stride = py::cast<int32_t>(t[2]);
    }
  }
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = fgmm_gmc_decompress_batch(ctx, ptr<void>(stream), it.data(), N, mode, clamp_scales);
  }
  if (rc) raise("GaussianMixtureConditional.decompress", rc);
}

} // namespace

PYBIND11_MODULE(_native, m) {
  m.doc() = "flashgmm_amd._native: compiled Python boundary of libflashgmm_amd.so (include/flashgmm_amd.h), GIL released across the native calls";
  m.attr("abi_version") = fgmm_abi_version();
  using namespace pybind11::literals;
  m.def("compress_stacked", &compress_stacked, "ctx"_a, "stream"_a, "y"_a, "scales"_a, "means"_a, "weights"_a, "N"_a, "M"_a, "hw"_a, "item_stride"_a, "stride_k"_a,
        "stride_c"_a, "dtype"_a, "flags"_a, "mode"_a, "clamp_scales"_a, "ckpt_stride"_a, "yq"_a, "zero_bitmap"_a, "bytes_cls"_a = py::none());
  m.def("compress_head_stacked", &compress_head_stacked, "ctx"_a, "stream"_a, "y"_a, "x"_a, "head"_a, "N"_a, "M"_a, "c_in"_a, "hw"_a, "mode"_a, "clamp_scales"_a,
        "ckpt_stride"_a, "yq"_a, "zero_bitmap"_a, "bytes_cls"_a = py::none());
  m.def("decompress_stacked", &decompress_stacked, "ctx"_a, "stream"_a, "strings"_a, "abs_maxes"_a, "zero_bitmap"_a, "zb_row_stride"_a, "scales"_a, "means"_a,
        "weights"_a, "N"_a, "M"_a, "hw"_a, "item_stride"_a, "stride_k"_a, "stride_c"_a, "dtype"_a, "flags"_a, "mode"_a, "clamp_scales"_a, "y_hat"_a,
        "ckpt"_a = py::none());
}
