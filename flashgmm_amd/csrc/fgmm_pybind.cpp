// fgmm_pybind.cpp — `flashgmm_amd._native`: the compiled Python boundary over the C ABI (include/flashgmm_amd.h).
//
// Mirror of the reference's pybind11 module definition, compressai/cpp_exts/rans/rans_interface.cpp:961-1036, for the batched entropy-model
// calls: tensors arrive as device addresses (`Tensor.data_ptr()`), the `fgmm_item` array is built here (no ctypes field marshalling, no
// numpy record arrays), the GIL is released across the native call - the reference holds it (no gil_scoped_release anywhere in its
// module) - and the bitstreams come back as `bytes` objects allocated at their final size and filled by the encoders' flush
// (fgmm_sink: no buffer of the library's, no copy after the call).  Nothing here computes: every function is argument plumbing around
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

// The bitstreams of a compress call as Python objects, made by the call itself (include/flashgmm_amd.h: fgmm_sink): when a host
// worker has coded item i, the calling thread - which serves the workers' requests while it waits for them - takes the GIL back for
// as long as it takes to allocate a `bytes` object of the stream's exact size, puts it in the result list and hands its storage to
// the worker's flush: the stream is written once, by the thread that has it in its cache, and nothing is copied afterwards.
// `cls`: a bytes subclass to instantiate instead (same storage layout; its tp_alloc zero-fills, which bytes' own constructor does not).
struct PySink {
  PyObject *list;     // count slots, NULL until the item's stream is there
  PyTypeObject *cls;  // NULL: bytes
  fgmm_sink sink;
  static void *alloc(void *user, int item, size_t nbytes) {
    PySink *s = static_cast<PySink *>(user);
    py::gil_scoped_acquire gil; // (the calling thread's own state, released around the native call)
    PyObject *o;
    if (!s->cls) {
      o = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)nbytes); // uninitialised, exactly as long as the bitstream
    } else {
      o = s->cls->tp_alloc(s->cls, (Py_ssize_t)nbytes); // what bytes.__new__(cls, ...) allocates (zeroed, the terminator included)
      if (o) reinterpret_cast<PyBytesObject *>(o)->ob_shash = -1; // "not hashed yet" (bytes.__new__ sets it the same way)
    }
    void *p = nullptr;
    if (o) {
      PyList_SET_ITEM(s->list, (Py_ssize_t)item, o);
      p = PyBytes_AS_STRING(o);
    } else {
      PyErr_Clear(); // (the call fails with FGMM_ERR_NOMEM)
    }
    return p;
  }
  PySink(py::list &out, py::handle cls_) : list(out.ptr()), cls(cls_.is_none() ? nullptr : reinterpret_cast<PyTypeObject *>(cls_.ptr())), sink{&PySink::alloc, this} {
    if (cls && (!PyType_Check(cls_.ptr()) || !PyType_IsSubtype(cls, &PyBytes_Type))) throw std::runtime_error("bytes_cls must be a subclass of bytes");
  }
};

// Checkpoints of a compress call: ONE bytes object holding every item's notes back to back (16 bytes each); every bitstream object
// (an instance of the bytes subclass CheckpointedBytes, flashgmm_amd/entropy_models.py) gets its `_ck` = (blob, first record, records,
// address of the first, stride) - the ndarray view is made by the object when someone asks for it.
void adopt_ckpts(fgmm_ctx *ctx, std::vector<fgmm_item> &it, py::list &strings, int ckpt_stride) {
  size_t total = 0;
  for (auto &f : it) total += (size_t)f.n_ckpt;
  PyObject *blob = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)(16 * total));
  if (!blob) throw py::error_already_set();
  py::object keep = py::reinterpret_steal<py::object>(blob);
  std::vector<void *> dst, src;
  std::vector<size_t> len;
  char *at = PyBytes_AS_STRING(blob);
  static PyObject *name = PyUnicode_InternFromString("_ck");
  py::object stride = py::int_(ckpt_stride), zero = py::int_(0);
  size_t first = 0;
  for (size_t i = 0; i < it.size(); ++i) {
    const size_t n = it[i].ckpt && it[i].n_ckpt > 0 ? (size_t)it[i].n_ckpt : 0;
    if (n) dst.push_back(at), src.push_back(it[i].ckpt), len.push_back(16 * n);
    py::tuple ck = py::make_tuple(keep, py::int_(first), py::int_(n), n ? py::object(py::int_(reinterpret_cast<uintptr_t>(at))) : zero, stride);
    if (PyObject_SetAttr(PyList_GET_ITEM(strings.ptr(), (Py_ssize_t)i), name, ck.ptr()) != 0) throw py::error_already_set();
    at += 16 * n, first += n;
  }
  if (!dst.empty()) {
    const int rc = fgmm_ctx_take_buffers(ctx, dst.data(), src.data(), len.data(), (int)dst.size());
    if (rc) raise("fgmm_ctx_take_buffers", rc);
  }
  for (auto &f : it) f.ckpt = nullptr; // (copied and released by the library)
}

// what a finished compress call returned in library-owned buffers, released if this binding does not get to hand it over (an allocation
// that fails half way, a copy that is refused): fgmm_ctx_take_buffers releases what it copies
struct Owned {
  std::vector<fgmm_item> &it;
  ~Owned() {
    for (auto &f : it) { // (the bitstreams are in Python objects already: the sink)
      fgmm_free(f.ckpt);
      f.ckpt = nullptr;
    }
  }
};

py::tuple finish_compress(fgmm_ctx *ctx, std::vector<fgmm_item> &it, int ckpt_stride, py::list strings) {
  Owned owned{it};
  for (size_t i = 0; i < it.size(); ++i)
    if (!PyList_GET_ITEM(strings.ptr(), (Py_ssize_t)i)) throw std::runtime_error("compress: item " + std::to_string(i) + " returned no bitstream");
  py::list abs_max(it.size());
  for (size_t i = 0; i < it.size(); ++i) PyList_SET_ITEM(abs_max.ptr(), (Py_ssize_t)i, PyLong_FromLong(it[i].abs_max));
  if (ckpt_stride) adopt_ckpts(ctx, it, strings, ckpt_stride);
  return py::make_tuple(strings, abs_max);
}

// GaussianMixtureConditional.compress for N stacked items (entropy_models.py:833-867, batched):
//   -> (list of N bytes, list of N abs_max); y_q and the zero bitmaps are written through yq / zero_bitmap (device float32 [N, M, hw] /
//   HOST int64 [N, M]).  ckpt_stride > 0: bytes_cls must be CheckpointedBytes, every bitstream object carries its notes (adopt_ckpts)
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
  py::list strings((size_t)N);
  PySink to(strings, bytes_cls);
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = fgmm_gmc_compress_batch_to(ctx, ptr<void>(stream), it.data(), N, mode, clamp_scales, &to.sink);
  }
  if (rc) raise("GaussianMixtureConditional.compress", rc);
  return finish_compress(ctx, it, ckpt_stride, strings);
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
  py::list strings((size_t)N);
  PySink to(strings, bytes_cls);
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = fgmm_gmc_compress_head_batch_to(ctx, ptr<void>(stream), it.data(), xs.data(), N, ptr<const fgmm_head>(head), mode, clamp_scales, &to.sink);
  }
  if (rc) raise("GaussianMixtureConditional.compress_head_batch", rc);
  return finish_compress(ctx, it, ckpt_stride, strings);
}

// GaussianMixtureConditional.decompress for N stacked items (entropy_models.py:872-910, batched): the bitstreams are read in place
// (borrowed pointers into the bytes objects, which the caller's list keeps alive across the call); y_hat is written through `y_hat`
// (device float32 [N, M, hw]).  zero_bitmap: HOST int64, row i at zero_bitmap + i * zb_row_stride elements.  ckpt_cls: None, or the
// bytes subclass whose instances carry out-of-band notes in `_ck` (CheckpointedBytes): (., ., count, address, stride).
void decompress_stacked(uintptr_t ctx_, uintptr_t stream, py::sequence strings, py::sequence abs_maxes, uintptr_t zero_bitmap, int64_t zb_row_stride,
                        uintptr_t scales, uintptr_t means, uintptr_t weights, int N, int M, int64_t hw, int64_t item_stride, int64_t stride_k, int64_t stride_c,
                        int dtype, int flags, int mode, int clamp_scales, uintptr_t y_hat, py::object ckpt_cls) {
  fgmm_ctx *ctx = ptr<fgmm_ctx>(ctx_);
  if ((int)py::len(strings) != N || (int)py::len(abs_maxes) != N) throw std::runtime_error("decompress: " + std::to_string(N) + " items in the parameter tensors, " + std::to_string(py::len(strings)) + " bitstreams");
  std::vector<fgmm_item> it((size_t)N);
  fill_items(it, Stacked{scales, means, weights, item_stride, stride_k, stride_c, dtype, flags, N, M, hw});
  PyTypeObject *cls = ckpt_cls.is_none() ? nullptr : reinterpret_cast<PyTypeObject *>(ckpt_cls.ptr());
  static PyObject *ck_name = PyUnicode_InternFromString("_ck");
  for (int i = 0; i < N; ++i) {
    fgmm_item &f = it[(size_t)i];
    py::object b = strings[(size_t)i];
    if (!PyBytes_Check(b.ptr())) throw std::runtime_error("decompress: bitstream " + std::to_string(i) + " is not a bytes object");
    f.bytes = reinterpret_cast<uint8_t *>(PyBytes_AS_STRING(b.ptr()));
    f.bytes_len = (size_t)PyBytes_GET_SIZE(b.ptr());
    f.abs_max = py::cast<int32_t>(abs_maxes[(size_t)i]);
    f.zero_bitmap = ptr<int64_t>(zero_bitmap) + (size_t)i * (size_t)zb_row_stride;
    f.yq_out = ptr<float>(y_hat) + (size_t)i * (size_t)M * (size_t)hw;
    if (cls && PyObject_TypeCheck(b.ptr(), cls)) {
      PyObject *rec = PyObject_GetAttr(b.ptr(), ck_name);
      if (!rec) throw py::error_already_set();
      py::tuple t = py::cast<py::tuple>(py::reinterpret_steal<py::object>(rec));
      if (t.size() != 5) throw std::runtime_error("decompress: bitstream " + std::to_string(i) + ": malformed checkpoint record");
      f.n_ckpt = py::cast<int64_t>(t[2]);
      f.ckpt = ptr<fgmm_ckpt>(py::cast<uintptr_t>(t[3]));
      f.ckpt_stride = py::cast<int32_t>(t[4]);
    }
  }
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = fgmm_gmc_decompress_batch(ctx, ptr<void>(stream), it.data(), N, mode, clamp_scales);
  }
  if (rc) raise("GaussianMixtureConditional.decompress", rc);
}

// ---- items of any mix of shapes (the sequence form of compress_batch / decompress_batch: ELIC's ten groups per image) --------------
// one item = a tuple of integers; compress: (y, scales, means, weights, M, hw, stride_k, stride_c, yq, zero_bitmap),
// decompress: (scales, means, weights, M, hw, stride_k, stride_c, y_hat, zero_bitmap): device addresses, HOST zero bitmap
void item_common(fgmm_item &f, const py::tuple &t, size_t at, int dtype, int flags) {
  std::memset(&f, 0, sizeof f);
  f.params.scales = ptr<const void>(py::cast<uintptr_t>(t[at]));
  f.params.means = ptr<const void>(py::cast<uintptr_t>(t[at + 1]));
  f.params.weights = ptr<const void>(py::cast<uintptr_t>(t[at + 2]));
  f.M = py::cast<int32_t>(t[at + 3]), f.K = FGMM_K, f.hw = py::cast<int64_t>(t[at + 4]);
  f.params.stride_k = py::cast<int64_t>(t[at + 5]), f.params.stride_c = py::cast<int64_t>(t[at + 6]);
  f.params.dtype = dtype, f.params.flags = flags;
  f.yq_out = ptr<float>(py::cast<uintptr_t>(t[at + 7]));
  f.zero_bitmap = ptr<int64_t>(py::cast<uintptr_t>(t[at + 8]));
}

py::tuple compress_items(uintptr_t ctx_, uintptr_t stream, py::sequence items, int dtype, int flags, int mode, int clamp_scales, int ckpt_stride,
                         py::object bytes_cls) {
  fgmm_ctx *ctx = ptr<fgmm_ctx>(ctx_);
  const size_t n = py::len(items);
  std::vector<fgmm_item> it(n);
  for (size_t i = 0; i < n; ++i) {
    py::tuple t = py::cast<py::tuple>(items[i]);
    if (t.size() != 10) throw std::runtime_error("compress_items: an item is (y, scales, means, weights, M, hw, stride_k, stride_c, yq, zero_bitmap)");
    item_common(it[i], t, 1, dtype, flags);
    it[i].y = ptr<const float>(py::cast<uintptr_t>(t[0]));
    it[i].ckpt_stride = ckpt_stride;
  }
  py::list strings(n);
  PySink to(strings, bytes_cls);
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = fgmm_gmc_compress_batch_to(ctx, ptr<void>(stream), it.data(), (int)n, mode, clamp_scales, &to.sink);
  }
  if (rc) raise("GaussianMixtureConditional.compress", rc);
  return finish_compress(ctx, it, ckpt_stride, strings);
}

void decompress_items(uintptr_t ctx_, uintptr_t stream, py::sequence strings, py::sequence abs_maxes, py::sequence items, int dtype, int flags, int mode,
                      int clamp_scales, py::object ckpt_cls) {
  fgmm_ctx *ctx = ptr<fgmm_ctx>(ctx_);
  const size_t n = py::len(items);
  if (py::len(strings) != n || py::len(abs_maxes) != n) throw std::runtime_error("decompress: " + std::to_string(n) + " items, " + std::to_string(py::len(strings)) + " bitstreams");
  std::vector<fgmm_item> it(n);
  PyTypeObject *cls = ckpt_cls.is_none() ? nullptr : reinterpret_cast<PyTypeObject *>(ckpt_cls.ptr());
  static PyObject *ck_name = PyUnicode_InternFromString("_ck");
  for (size_t i = 0; i < n; ++i) {
    py::tuple t = py::cast<py::tuple>(items[i]);
    if (t.size() != 9) throw std::runtime_error("decompress_items: an item is (scales, means, weights, M, hw, stride_k, stride_c, y_hat, zero_bitmap)");
    fgmm_item &f = it[i];
    item_common(f, t, 0, dtype, flags);
    py::object b = strings[i];
    if (!PyBytes_Check(b.ptr())) throw std::runtime_error("decompress: bitstream " + std::to_string(i) + " is not a bytes object");
    f.bytes = reinterpret_cast<uint8_t *>(PyBytes_AS_STRING(b.ptr()));
    f.bytes_len = (size_t)PyBytes_GET_SIZE(b.ptr());
    f.abs_max = py::cast<int32_t>(abs_maxes[i]);
    if (cls && PyObject_TypeCheck(b.ptr(), cls)) {
      PyObject *rec = PyObject_GetAttr(b.ptr(), ck_name);
      if (!rec) throw py::error_already_set();
      py::tuple c = py::cast<py::tuple>(py::reinterpret_steal<py::object>(rec));
      if (c.size() != 5) throw std::runtime_error("decompress: bitstream " + std::to_string(i) + ": malformed checkpoint record");
      f.n_ckpt = py::cast<int64_t>(c[2]);
      f.ckpt = ptr<fgmm_ckpt>(py::cast<uintptr_t>(c[3]));
      f.ckpt_stride = py::cast<int32_t>(c[4]);
    }
  }
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = fgmm_gmc_decompress_batch(ctx, ptr<void>(stream), it.data(), (int)n, mode, clamp_scales);
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
  m.def("compress_items", &compress_items, "ctx"_a, "stream"_a, "items"_a, "dtype"_a, "flags"_a, "mode"_a, "clamp_scales"_a, "ckpt_stride"_a, "bytes_cls"_a = py::none());
  m.def("decompress_items", &decompress_items, "ctx"_a, "stream"_a, "strings"_a, "abs_maxes"_a, "items"_a, "dtype"_a, "flags"_a, "mode"_a, "clamp_scales"_a,
        "ckpt_cls"_a = py::none());
  m.def("decompress_stacked", &decompress_stacked, "ctx"_a, "stream"_a, "strings"_a, "abs_maxes"_a, "zero_bitmap"_a, "zb_row_stride"_a, "scales"_a, "means"_a,
        "weights"_a, "N"_a, "M"_a, "hw"_a, "item_stride"_a, "stride_k"_a, "stride_c"_a, "dtype"_a, "flags"_a, "mode"_a, "clamp_scales"_a, "y_hat"_a,
        "ckpt_cls"_a = py::none());
}
