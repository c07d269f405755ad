// stub of <pybind11/pybind11.h>: irls_optim.h only declares the namespace alias
#pragma once
namespace pybind11 {}
