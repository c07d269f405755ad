// Shadows the reference's _pixsfm/src/helpers.h (pybind11 casters): only the `half` alias is needed below the bindings.
#pragma once
#include "pybind11/pybind11.h"
#include "third-party/half.hpp"
using half = half_float::half;
