// ref_shim.cc -- ORACLE support (test infrastructure only).
//
// Compiles the reference's OWN headers where they lie under /root/reference:
//   pixsfm/base/src/cubic_hermite_spline_simd.h   (the AVX2/F16C/FMA Catmull-Rom kernels)
//   pixsfm/base/src/grid2d.h                      (HWC indexing + border clamping)
//   third-party/half.hpp                          (vendored in the reference)
// against the tiny stub headers in oracle/ref_stubs/ (Eigen/ceres/glog are absent in this
// image; neither header uses their arithmetic).  Output: oracle/_ref/libpxo_ref.so.
// Nothing from the reference is copied into this repository.
//
// The glue below (6 spline calls around Grid2D::GetPointer) restates
// BiCubicInterpolator::EvaluateSIMD, base/src/interpolation.h:177-218, which itself
// cannot be compiled here (it needs Eigen/Ceres/pybind11/HighFive/Boost).
#include <cmath>
#include <cstdint>
#include <vector>

#include "base/src/cubic_hermite_spline_simd.h"
#include "base/src/grid2d.h"

namespace {

template <typename dtype, int C, typename vtype>
void EvaluateSIMD(const dtype* data, int H, int W, double r, double c, double* f, double* dfdr,
                  double* dfdc) {
  using Grid = pixsfm::Grid2D<dtype, C>;
  Grid grid(data, 0, H, 0, W);
  const int row = std::floor(r);
  const int col = std::floor(c);
  std::vector<vtype> fk[4], dk[4];
  for (int j = 0; j < 4; ++j) {
    fk[j].resize(C);
    dk[j].resize(C);
    auto p0 = grid.GetPointer(row - 1 + j, col - 1);
    auto p1 = grid.GetPointer(row - 1 + j, col);
    auto p2 = grid.GetPointer(row - 1 + j, col + 1);
    auto p3 = grid.GetPointer(row - 1 + j, col + 2);
    pixsfm::CubicHermiteSplineSIMD<C>(p0, p1, p2, p3, c - col, fk[j].data(), dk[j].data());
  }
  pixsfm::CubicHermiteSplineSIMD<C>(fk[0].data(), fk[1].data(), fk[2].data(), fk[3].data(),
                                    r - row, f, dfdr);
  if (dfdc != nullptr) {
    pixsfm::CubicHermiteSplineSIMD<C>((vtype*)dk[0].data(), (vtype*)dk[1].data(),
                                      (vtype*)dk[2].data(), (vtype*)dk[3].data(), r - row, dfdc,
                                      (double*)nullptr);
  }
}

template <int C>
int Dispatch(const void* data, int dtype, int H, int W, double r, double c, int use_float_simd,
             double* f, double* dfdr, double* dfdc) {
  if (use_float_simd) {
    if (dtype == 0) EvaluateSIMD<half, C, float>((const half*)data, H, W, r, c, f, dfdr, dfdc);
    else if (dtype == 1) EvaluateSIMD<float, C, float>((const float*)data, H, W, r, c, f, dfdr, dfdc);
    else EvaluateSIMD<double, C, float>((const double*)data, H, W, r, c, f, dfdr, dfdc);
  } else {
    if (dtype == 0) EvaluateSIMD<half, C, double>((const half*)data, H, W, r, c, f, dfdr, dfdc);
    else if (dtype == 1) EvaluateSIMD<float, C, double>((const float*)data, H, W, r, c, f, dfdr, dfdc);
    else EvaluateSIMD<double, C, double>((const double*)data, H, W, r, c, f, dfdr, dfdc);
  }
  return 0;
}

}  // namespace

extern "C" {

// dtype: 0 = half, 1 = float, 2 = double.  Returns -1 for an un-instantiated channel count.
int pxo_ref_bicubic(const void* data, int dtype, int H, int W, int C, double r, double c,
                    int use_float_simd, double* f, double* dfdr, double* dfdc) {
  switch (C) {
    case 8: return Dispatch<8>(data, dtype, H, W, r, c, use_float_simd, f, dfdr, dfdc);
    case 12: return Dispatch<12>(data, dtype, H, W, r, c, use_float_simd, f, dfdr, dfdc);
    case 19: return Dispatch<19>(data, dtype, H, W, r, c, use_float_simd, f, dfdr, dfdc);
    case 64: return Dispatch<64>(data, dtype, H, W, r, c, use_float_simd, f, dfdr, dfdc);
    case 128: return Dispatch<128>(data, dtype, H, W, r, c, use_float_simd, f, dfdr, dfdc);
    default: return -1;
  }
}

// Raw spline entry points (C = 128 only), half/float input -> double output.
void pxo_ref_spline_half128(const uint16_t* p0, const uint16_t* p1, const uint16_t* p2,
                            const uint16_t* p3, double x, double* f, double* dfdx) {
  pixsfm::CubicHermiteSplineSIMD<128>((const half*)p0, (const half*)p1, (const half*)p2,
                                      (const half*)p3, x, f, dfdx);
}
void pxo_ref_spline_double128(const double* p0, const double* p1, const double* p2,
                              const double* p3, double x, double* f, double* dfdx) {
  pixsfm::CubicHermiteSplineSIMD<128>(p0, p1, p2, p3, x, f, dfdx);
}

// Timing helper for the cpu_baseline "reference" flavour: n evaluations of the real
// AVX2 kernels on fp16 16x16x128 patches at given (r,c); returns a checksum.
double pxo_ref_bicubic_many_half128(const uint16_t* arena, int64_t n, int H, int W,
                                    const int64_t* patch_idx, const double* rc, double* out) {
  double acc = 0;
  double f[128], dr[128], dc[128];
  for (int64_t i = 0; i < n; ++i) {
    const half* data = (const half*)(arena + patch_idx[i] * (int64_t)H * W * 128);
    EvaluateSIMD<half, 128, double>(data, H, W, rc[2 * i], rc[2 * i + 1], f, dr, dc);
    if (out) {
      for (int k = 0; k < 128; ++k) {
        out[(i * 3 + 0) * 128 + k] = f[k];
        out[(i * 3 + 1) * 128 + k] = dr[k];
        out[(i * 3 + 2) * 128 + k] = dc[k];
      }
    }
    acc += f[0] + dr[1] + dc[2];
  }
  return acc;
}
// The reference's vendored half.hpp (third-party/half.hpp, 2.2.0) arithmetic the cost-map extraction leans on
// (bundle_adjustment/src/costmap_extractor.h:266-279 subtracts texels in the storage type; FeaturePatch::SetEntry,
// features/src/featurepatch.h:246-248, casts the double cost to the storage type).  Element-wise, n entries.
void pxo_ref_half_sub(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n) {
  const half* ha = (const half*)a;
  const half* hb = (const half*)b;
  half* ho = (half*)out;
  for (int64_t i = 0; i < n; ++i) ho[i] = ha[i] - hb[i];
}
void pxo_ref_half_from_double(const double* v, uint16_t* out, int64_t n) {
  half* ho = (half*)out;
  for (int64_t i = 0; i < n; ++i) ho[i] = half(v[i]);
}
}
