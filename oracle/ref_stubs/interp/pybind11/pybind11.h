// Stub for <pybind11/pybind11.h>: the headers built in place only NAME these types in declarations.
#pragma once
#include <stdexcept>
namespace pybind11 {
struct array { enum { c_style = 1, f_style = 2 }; };
template <typename T, int Flags = 0> class array_t {};
struct value_error : std::runtime_error { using std::runtime_error::runtime_error; };
}  // namespace pybind11
namespace py = pybind11;
