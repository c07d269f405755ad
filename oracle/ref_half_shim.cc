// oracle/ref_half_shim.cc -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// The ONE piece of /root/reference that compiles here from its own sources with nothing added: the vendored
// third-party/half.hpp (2.2.0), a self-contained header (standard library + <immintrin.h> only).  Everything else on the hot
// path includes Eigen / Ceres / COLMAP / HighFive headers that this image does not have, and is therefore UNBUILDABLE here
// (SURVEY 8c); no stand-in headers are written for them.  This file adds extern "C" entry points around the header so that
// ctypes can call it -- it does not replace or restate any part of it.
//
// What it pins: the two half-precision rules the cost-map extraction leans on
//   * bundle_adjustment/src/costmap_extractor.h:266-279 subtracts texels IN THE STORAGE TYPE (half - half -> half);
//   * features/src/featurepatch.h:246-248 (SetEntry) casts the double cost to the storage type (half(double)).
// and the half -> float widening the interpolation reads fp16 texels through.
#include <cstdint>

#include "third-party/half.hpp"

using half_float::half;

extern "C" {
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
void pxo_ref_half_to_float(const uint16_t* a, float* out, int64_t n) {
  const half* ha = (const half*)a;
  for (int64_t i = 0; i < n; ++i) out[i] = (float)ha[i];
}
}
