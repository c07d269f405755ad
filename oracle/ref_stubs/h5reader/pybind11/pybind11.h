// Stub for pybind11 under the in-place build of the reference's cache reader (oracle/ref_h5_shim.cc): the four feature
// sources name numpy arrays / dicts in constructors the shim never calls (the explicit template instantiations at the end
// of those files compile every member, so the calls must exist); reaching one of them at run time aborts.
#pragma once
#include <sys/types.h>

#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <stdexcept>
#include <string>
#include <vector>
namespace pybind11 {
[[noreturn]] inline void stub_unreachable(const char* what) { std::fprintf(stderr, "pybind11 stub: %s is not available\n", what); std::abort(); }
struct handle {};
struct object : handle {
  template <typename T> T cast() const { stub_unreachable("object::cast"); }
};
struct str : object { str() {} explicit str(const std::string&) {} };
struct dict : object {
  object operator[](const char*) const { stub_unreachable("dict[]"); }
  bool contains(const char*) const { return false; }
};
struct buffer_info { void* ptr = nullptr; ssize_t size = 0; ssize_t ndim = 0; std::vector<ssize_t> shape; std::vector<ssize_t> strides; };
struct array : object { enum { c_style = 1, f_style = 2 }; };
template <typename T, int Flags = 0> class array_t : public array {
 public:
  array_t() {}
  template <int F2> array_t(const array_t<T, F2>&) {}       // c_style <-> default flags
  template <typename S1, typename S2>
  array_t(std::initializer_list<S1>, std::initializer_list<S2>, const T*, handle = handle()) { stub_unreachable("array_t(shape, strides, ptr, base)"); }
  buffer_info request() const { stub_unreachable("array_t::request"); }
};
struct value_error : std::runtime_error { using std::runtime_error::runtime_error; };
}  // namespace pybind11
namespace py = pybind11;
