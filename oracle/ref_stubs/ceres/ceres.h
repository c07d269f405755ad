// Stub for <ceres/ceres.h>: base/src/grid2d.h uses only glog's CHECK_* macros from it.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "Eigen/Core"
#define PXO_STUB_CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s\n", #cond); std::abort(); } } while (0)
#define CHECK_GE(a, b) PXO_STUB_CHECK((a) >= (b))
#define CHECK_LT(a, b) PXO_STUB_CHECK((a) < (b))
