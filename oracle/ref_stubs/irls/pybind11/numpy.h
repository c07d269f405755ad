// stub of <pybind11/numpy.h>: irls_optim.h only declares the namespace alias
#pragma once
namespace pybind11 {}
