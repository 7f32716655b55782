// Host-side plumbing only: the two allocations + the C-ABI call of the drop-in fast paths (CachedSDF.__call__ / query_into,
// ComposedSDF.__call__) without the interpreter in between.  No kernels, no arithmetic: every entry takes the ADDRESS of the
// libpvamd.so entry point it is to call (Python resolves it through ctypes, so PVAMD_LIB and the variant check still decide
// which library runs) and returns None when the arguments are not already what the kernel takes -- the Python path (which
// converts, or raises the descriptive error) handles those.  Optional: without this module the same calls go through ctypes.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPFunctions.h>

namespace {

using cached_fn = int (*)(const void*, const float*, int64_t, float*, float*, uint8_t*, void*);
using composed_fn = int (*)(const void*, int32_t, const float*, int32_t, const float*, int64_t, float*, float*, int32_t*, int32_t, void*);

inline bool takes(const at::Tensor& t, int64_t dev) {
    return t.scalar_type() == at::kFloat && t.is_cuda() && t.get_device() == dev && t.is_contiguous();
}
// the launch goes to the CURRENT device's current stream: the grid's device must be the current one (the Python path switches)
inline bool current(int64_t dev) { return (int64_t)c10::hip::current_device() == dev; }

PyObject* error_class = nullptr;  // _lib.PvamdError once the binding has handed it over (set_error_class): the ctypes path's exception type

[[noreturn]] void fail(const char* what, int rc) {
    const std::string msg = std::string(what) + (rc < 0 ? ": invalid argument (" : ": hipError_t (") + std::to_string(rc) + ")";
    PyErr_SetString(error_class ? error_class : PyExc_RuntimeError, msg.c_str());
    throw py::error_already_set();
}

// val, grad = cached(points): points (..., 3) float32 contiguous on device `dev`, which must be the current device
py::object cached_call(int64_t fn, int64_t desc, int64_t dev, const at::Tensor& p) {
    if (!takes(p, dev) || p.dim() < 1 || p.size(-1) != 3 || !current(dev)) return py::none();
    at::Tensor val, grad;
    int rc;
    {
        py::gil_scoped_release nogil;  // like the ctypes call this replaces: other host threads keep running during the launch
        val = at::empty(p.sizes().slice(0, p.dim() - 1), p.options());
        grad = at::empty_like(p);
        rc = reinterpret_cast<cached_fn>(fn)(reinterpret_cast<const void*>(desc), p.data_ptr<float>(), val.numel(), val.data_ptr<float>(),
                                             grad.data_ptr<float>(), nullptr, c10::hip::getCurrentHIPStream(dev).stream());
    }
    if (rc != 0) fail("pvamd_cached_query", rc);
    return py::make_tuple(std::move(val), std::move(grad));
}

// cached.query_into(points, out_val, out_grad); False = not handled here
bool cached_into(int64_t fn, int64_t desc, int64_t dev, const at::Tensor& p, const at::Tensor& val, const at::Tensor& grad) {
    if (!takes(p, dev) || !takes(val, dev) || !takes(grad, dev) || p.dim() != 2 || p.size(1) != 3 || !current(dev)) return false;
    const int64_t P = p.size(0);
    if (val.dim() != 1 || val.size(0) != P || grad.dim() != 2 || grad.size(0) != P || grad.size(1) != 3) return false;
    int rc;
    {
        py::gil_scoped_release nogil;
        rc = reinterpret_cast<cached_fn>(fn)(reinterpret_cast<const void*>(desc), p.data_ptr<float>(), P, val.data_ptr<float>(),
                                             grad.data_ptr<float>(), nullptr, c10::hip::getCurrentHIPStream(dev).stream());
    }
    if (rc != 0) fail("pvamd_cached_query", rc);
    return true;
}

// val, grad = composed(points) through pvamd_composed_query: `batch` = the transform batch dims ((): none -> flat (P,) / (P, 3))
py::object composed_call(int64_t fn, int64_t grids, int64_t S, int64_t tf, int64_t A, const std::vector<int64_t>& batch, int64_t flags,
                         int64_t dev, const at::Tensor& p) {
    if (!takes(p, dev) || p.dim() < 1 || p.size(-1) != 3 || !current(dev)) return py::none();
    const int64_t P = p.numel() / 3;
    std::vector<int64_t> vs, gs;
    if (batch.empty()) {
        vs = {P};
        gs = {P, 3};
    } else {
        vs = batch;
        vs.insert(vs.end(), p.sizes().begin(), p.sizes().end() - 1);
        gs = batch;
        gs.insert(gs.end(), p.sizes().begin(), p.sizes().end());
    }
    at::Tensor val, grad;
    int rc;
    {
        py::gil_scoped_release nogil;
        val = at::empty(vs, p.options());
        grad = at::empty(gs, p.options());
        rc = reinterpret_cast<composed_fn>(fn)(reinterpret_cast<const void*>(grids), (int32_t)S, reinterpret_cast<const float*>(tf), (int32_t)A,
                                               p.data_ptr<float>(), P, val.data_ptr<float>(), grad.data_ptr<float>(), nullptr, (int32_t)flags,
                                               c10::hip::getCurrentHIPStream(dev).stream());
    }
    if (rc != 0) fail("pvamd_composed_query", rc);
    return py::make_tuple(std::move(val), std::move(grad));
}

}  // namespace

PYBIND11_MODULE(_pvamd_fast, m) {
    m.def("cached_call", &cached_call);
    m.def("cached_into", &cached_into);
    m.def("composed_call", &composed_call);
    m.def("set_error_class", [](py::object cls) { error_class = cls.inc_ref().ptr(); });
}
