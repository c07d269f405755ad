// Stub for <ceres/cubic_interpolation.h>: nothing from it is used by base/src/grid2d.h.
#pragma once
