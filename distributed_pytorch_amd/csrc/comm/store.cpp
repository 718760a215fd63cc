// Python surface of the TcpStore (core: store_core.cpp).  Every call releases the GIL: a blocked get must not
// freeze other Python threads, and close() may linger for the other ranks' goodbyes.
#include "comm.h"

#include <pybind11/stl.h>

#include <memory>
#include <stdexcept>

#include "store_core.h"

namespace py = pybind11;
using dpa::store::TcpStore;

namespace {

class PyStore {
 public:
  PyStore(const std::string& host, int port, bool is_master, double timeout_s, double linger_s) {
    py::gil_scoped_release nogil;
    impl_.reset(new TcpStore(host, port, is_master, timeout_s, linger_s));
  }
  ~PyStore() { close(); }
  TcpStore& s() {
    if (!impl_) throw std::runtime_error("TcpStore: closed");
    return *impl_;
  }
  void close() {
    if (!impl_) return;
    py::gil_scoped_release nogil;
    impl_.reset();
  }
  int port() { return s().port(); }
  std::string host() { return s().host(); }
  void set(const std::string& k, const py::bytes& v) {
    std::string val = v;
    py::gil_scoped_release nogil;
    s().client().set(k, val);
  }
  py::bytes get(const std::string& k, double timeout_s) {
    std::string out;
    {
      py::gil_scoped_release nogil;
      out = s().client().get(k, timeout_s);
    }
    return py::bytes(out);
  }
  int64_t add(const std::string& k, int64_t d) {
    py::gil_scoped_release nogil;
    return s().client().add(k, d);
  }
  bool check(const std::string& k) {
    py::gil_scoped_release nogil;
    return s().client().check(k);
  }
  bool remove(const std::string& k) {
    py::gil_scoped_release nogil;
    return s().client().remove(k);
  }
  int64_t num_keys() {
    py::gil_scoped_release nogil;
    return s().client().num_keys();
  }
  void barrier(const std::string& name, int world, double timeout_s) {
    py::gil_scoped_release nogil;
    s().client().barrier(name, world, timeout_s);
  }

 private:
  std::unique_ptr<TcpStore> impl_;
};

}  // namespace

void bind_store(py::module_& m) {
  py::class_<PyStore>(m, "TcpStore")
      .def(py::init<const std::string&, int, bool, double, double>(), py::arg("host"), py::arg("port"),
           py::arg("is_master"), py::arg("timeout_s") = 300.0, py::arg("linger_s") = 10.0)
      .def_property_readonly("port", &PyStore::port)
      .def_property_readonly("host", &PyStore::host)
      .def("set", &PyStore::set)
      .def("get", &PyStore::get, py::arg("key"), py::arg("timeout_s") = -1.0)
      .def("add", &PyStore::add)
      .def("check", &PyStore::check)
      .def("remove", &PyStore::remove)
      .def("num_keys", &PyStore::num_keys)
      .def("barrier", &PyStore::barrier, py::arg("name"), py::arg("world"), py::arg("timeout_s") = -1.0)
      .def("close", &PyStore::close);
}
