// drrt_module.cpp -- the reference-side binding INTEGRATION.md describes, as a complete, compiled example:
// a pybind11 module exposing `TracerC` with the method names and argument order of the reference's
// `drrt.TracerC` (/root/reference/src/drrt.cpp:47-58, include/tracer.h:15-89), every method a thin wrapper
// over the C ABI of include/drrt_hip.h, torch tensors (on the ROCm device) as the exchange type in place
// of enoki arrays.  Errors reach Python as RuntimeError carrying the library's message, as pybind11 did for
// the reference's std::runtime_error.  Build: examples/pybind_drrt/build.py.
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <array>
#include <stdexcept>
#include <tuple>

#include "../../include/drrt_hip.h"

namespace {

void check(int rc) {
  if (rc != DRRT_OK) throw std::runtime_error(drrt_last_error());
}
void* stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

at::Tensor f32(const at::Tensor& t) {
  TORCH_CHECK(t.is_cuda(), "TracerC expects tensors on the cuda (ROCm) device");
  return t.detach().to(at::kFloat).contiguous();
}
at::Tensor rays(const at::Tensor& t, int64_t n = -1) {
  at::Tensor r = f32(t);
  TORCH_CHECK(r.dim() == 2 && r.size(1) == 3 && (n < 0 || r.size(0) == n), "expected an (N,3) ray tensor");
  return r;
}

struct Scratch {            // per call: workspace sized by the library, device-resident stats block
  at::Tensor ws, st;
  Scratch(const at::Tensor& like, size_t n, unsigned flags)
      : ws(at::empty({(int64_t)drrt_workspace_bytes(n, flags) + 512}, like.options().dtype(at::kByte))),
        st(at::empty({3}, like.options().dtype(at::kLong))) {}
  drrt_stats* stats() { return (drrt_stats*)st.data_ptr(); }
};

constexpr unsigned kFlags = DRRT_FLAG_SORT_RAYS;

struct TracerC {
  // Tracer::trace (src/tracer.cpp:35-100)
  std::pair<at::Tensor, at::Tensor> trace(at::Tensor rif, std::array<int, 3> res, at::Tensor pos, at::Tensor vel,
                                          float h, float ds) {
    rif = f32(rif).reshape({-1}); pos = rays(pos); vel = rays(vel, pos.size(0));
    const size_t n = pos.size(0);
    auto xt = at::empty_like(pos), vt = at::empty_like(vel);
    Scratch s(rif, n, kFlags);
    check(drrt_trace_f32(rif.data_ptr<float>(), rif.numel(), res.data(), n, pos.data_ptr<float>(),
                         vel.data_ptr<float>(), h, ds, xt.data_ptr<float>(), vt.data_ptr<float>(), s.stats(),
                         s.ws.data_ptr(), s.ws.numel(), kFlags, stream()));
    return {xt, vt};
  }
  // Tracer::trace_plane (:102-172) -> (xt, vt, failmask)
  std::tuple<at::Tensor, at::Tensor, at::Tensor> trace_pln(at::Tensor rif, std::array<int, 3> res, at::Tensor pos,
                                                           at::Tensor vel, at::Tensor pln_o, at::Tensor pln_d,
                                                           float h, float ds) {
    rif = f32(rif).reshape({-1}); pos = rays(pos);
    const int64_t n = pos.size(0);
    vel = rays(vel, n); pln_o = rays(pln_o, n); pln_d = rays(pln_d, n);
    auto xt = at::empty_like(pos), vt = at::empty_like(vel);
    auto fm = at::empty({n}, rif.options().dtype(at::kByte));
    Scratch s(rif, n, kFlags);
    check(drrt_trace_pln_f32(rif.data_ptr<float>(), rif.numel(), res.data(), n, pos.data_ptr<float>(),
                             vel.data_ptr<float>(), pln_o.data_ptr<float>(), pln_d.data_ptr<float>(), h, ds,
                             xt.data_ptr<float>(), vt.data_ptr<float>(), fm.data_ptr<uint8_t>(), s.stats(),
                             s.ws.data_ptr(), s.ws.numel(), kFlags, stream()));
    return {xt, vt, fm.to(at::kBool)};
  }
  // Tracer::trace_sdf (:244-310)
  std::pair<at::Tensor, at::Tensor> trace_sdf(at::Tensor rif, at::Tensor sdf, std::array<int, 3> res, at::Tensor pos,
                                              at::Tensor vel, float h, float ds) {
    rif = f32(rif).reshape({-1}); sdf = f32(sdf).reshape({-1}); pos = rays(pos); vel = rays(vel, pos.size(0));
    if (sdf.numel() != rif.numel()) throw std::runtime_error("Resolution doesn't match data");   // src/volume.cpp:37
    const size_t n = pos.size(0);
    auto xt = at::empty_like(pos), vt = at::empty_like(vel);
    Scratch s(rif, n, kFlags);
    check(drrt_trace_sdf_f32(rif.data_ptr<float>(), sdf.data_ptr<float>(), rif.numel(), res.data(), n,
                             pos.data_ptr<float>(), vel.data_ptr<float>(), h, ds, xt.data_ptr<float>(),
                             vt.data_ptr<float>(), s.stats(), s.ws.data_ptr(), s.ws.numel(), kFlags, stream()));
    return {xt, vt};
  }
  // Tracer::trace_target (:174-242) -> (xt, vt, dist2)
  std::tuple<at::Tensor, at::Tensor, at::Tensor> trace_target(at::Tensor rif, std::array<int, 3> res, at::Tensor pos,
                                                              at::Tensor vel, at::Tensor target, float h, float ds) {
    rif = f32(rif).reshape({-1}); pos = rays(pos);
    const int64_t n = pos.size(0);
    vel = rays(vel, n); target = rays(target, n);
    auto xt = at::empty_like(pos), vt = at::empty_like(vel);
    auto d2 = at::empty({n}, rif.options());
    Scratch s(rif, n, kFlags);
    check(drrt_trace_target_f32(rif.data_ptr<float>(), rif.numel(), res.data(), n, pos.data_ptr<float>(),
                                vel.data_ptr<float>(), target.data_ptr<float>(), h, ds, xt.data_ptr<float>(),
                                vt.data_ptr<float>(), d2.data_ptr<float>(), s.stats(), s.ws.data_ptr(),
                                s.ws.numel(), kFlags, stream()));
    return {xt, vt, d2};
  }
  // Tracer::trace_cable (:312-382) -> (xt, vt, dist2)
  std::tuple<at::Tensor, at::Tensor, at::Tensor> trace_cable(at::Tensor rif, float radius, float length,
                                                             at::Tensor pos, at::Tensor vel, at::Tensor target,
                                                             float ds) {
    rif = f32(rif).reshape({-1}); pos = rays(pos);
    const int64_t n = pos.size(0);
    vel = rays(vel, n); target = rays(target, n);
    auto xt = at::empty_like(pos), vt = at::empty_like(vel);
    auto d2 = at::empty({n}, rif.options());
    Scratch s(rif, n, 0);
    check(drrt_trace_cable_f32(rif.data_ptr<float>(), rif.numel(), radius, length, n, pos.data_ptr<float>(),
                               vel.data_ptr<float>(), target.data_ptr<float>(), ds, xt.data_ptr<float>(),
                               vt.data_ptr<float>(), d2.data_ptr<float>(), s.stats(), s.ws.data_ptr(),
                               s.ws.numel(), 0, stream()));
    return {xt, vt, d2};
  }
  // Tracer::backtrace (:384-440) -> flat dL/dn
  at::Tensor backtrace(at::Tensor rif, std::array<int, 3> res, at::Tensor xt, at::Tensor vt, at::Tensor dx,
                       at::Tensor dv, float h, float ds) {
    rif = f32(rif).reshape({-1}); xt = rays(xt);
    const int64_t n = xt.size(0);
    vt = rays(vt, n); dx = rays(dx, n); dv = rays(dv, n);
    auto grad = at::empty_like(rif);
    Scratch s(rif, n, kFlags);
    check(drrt_backtrace_f32(rif.data_ptr<float>(), rif.numel(), res.data(), n, xt.data_ptr<float>(),
                             vt.data_ptr<float>(), dx.data_ptr<float>(), dv.data_ptr<float>(), h, ds,
                             grad.data_ptr<float>(), s.stats(), s.ws.data_ptr(), s.ws.numel(), kFlags, stream()));
    return grad;
  }
  // Tracer::backtrace_sdf (:443-509)
  at::Tensor backtrace_sdf(at::Tensor rif, at::Tensor sdf, std::array<int, 3> res, at::Tensor xt, at::Tensor vt,
                           at::Tensor dx, at::Tensor dv, float h, float ds) {
    rif = f32(rif).reshape({-1}); sdf = f32(sdf).reshape({-1}); xt = rays(xt);
    if (sdf.numel() != rif.numel()) throw std::runtime_error("Resolution doesn't match data");
    const int64_t n = xt.size(0);
    vt = rays(vt, n); dx = rays(dx, n); dv = rays(dv, n);
    auto grad = at::empty_like(rif);
    Scratch s(rif, n, kFlags);
    check(drrt_backtrace_sdf_f32(rif.data_ptr<float>(), sdf.data_ptr<float>(), rif.numel(), res.data(), n,
                                 xt.data_ptr<float>(), vt.data_ptr<float>(), dx.data_ptr<float>(),
                                 dv.data_ptr<float>(), h, ds, grad.data_ptr<float>(), s.stats(), s.ws.data_ptr(),
                                 s.ws.numel(), kFlags, stream()));
    return grad;
  }
  // Tracer::backtrace_cable (:511-567) -> dL/d(profile)
  at::Tensor backtrace_cable(at::Tensor rif, float radius, float length, at::Tensor xt, at::Tensor vt, at::Tensor dx,
                             at::Tensor dv, float ds) {
    rif = f32(rif).reshape({-1}); xt = rays(xt);
    const int64_t n = xt.size(0);
    vt = rays(vt, n); dx = rays(dx, n); dv = rays(dv, n);
    auto grad = at::empty_like(rif);
    Scratch s(rif, n, 0);
    check(drrt_backtrace_cable_f32(rif.data_ptr<float>(), rif.numel(), radius, length, n, xt.data_ptr<float>(),
                                   vt.data_ptr<float>(), dx.data_ptr<float>(), dv.data_ptr<float>(), ds,
                                   grad.data_ptr<float>(), s.stats(), s.ws.data_ptr(), s.ws.numel(), 0, stream()));
    return grad;
  }
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "reference-style pybind module over libdrrt_hip.so (example binding, see INTEGRATION.md)";
  py::class_<TracerC>(m, "TracerC")
      .def(py::init<>())
      .def("trace", &TracerC::trace)
      .def("trace_pln", &TracerC::trace_pln)
      .def("trace_sdf", &TracerC::trace_sdf)
      .def("trace_target", &TracerC::trace_target)
      .def("trace_cable", &TracerC::trace_cable)
      .def("backtrace", &TracerC::backtrace)
      .def("backtrace_sdf", &TracerC::backtrace_sdf)
      .def("backtrace_cable", &TracerC::backtrace_cable);
  m.def("version", []() { return std::string(drrt_version()); });
}
