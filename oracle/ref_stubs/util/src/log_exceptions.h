// Stub for the reference's util/src/log_exceptions.h (which needs glog/pybind11):
// base/src/grid2d.h uses only THROW_CHECK.
#pragma once
#include <stdexcept>
#define THROW_CHECK(cond) do { if (!(cond)) throw std::invalid_argument(#cond); } while (0)
