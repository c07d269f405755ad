/*
 * pxo_interp.c -- ORACLE (test infrastructure only; see pxo.h header).
 * Restates A1-A5 of SURVEY.md section 8a: patch coordinates, Catmull-Rom bicubic
 * interpolation with the reference's exact mixed-precision contract, L2 normalisation
 * with analytic derivative.
 *
 * Build with -ffp-contract=off: every fused multiply-add below is explicit (fma/fmaf)
 * exactly where the reference uses an FMA intrinsic, and nowhere else.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "pxo.h"

/* ---------------------------------------------------------------------------------
 * fp16 <-> fp32 (IEEE binary16, round-to-nearest-even), equivalent to F16C
 * _mm256_cvtph_ps used at base/src/cubic_hermite_spline_simd.h:51-54.
 * --------------------------------------------------------------------------------- */
float pxo_half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while (!(man & 0x400u));
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

uint16_t pxo_float_to_half(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  int32_t exp = (int32_t)((x >> 23) & 0xffu) - 127 + 15;
  uint32_t man = x & 0x7fffffu;
  if (((x >> 23) & 0xffu) == 0xffu) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | (man ? 0x200u | (man >> 13) : 0));
  }
  if (exp >= 31) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf */
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign; /* underflow -> 0 */
    man |= 0x800000u;
    int shift = 14 - exp; /* 14..24 */
    uint32_t half_man = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_man & 1u))) ++half_man;
    return (uint16_t)(sign | half_man);
  }
  uint32_t half = ((uint32_t)exp << 10) | (man >> 13);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) ++half; /* may carry into exp: ok */
  return (uint16_t)(sign | half);
}

void pxo_halfs_to_floats(const uint16_t* h, float* f, int64_t n) {
  for (int64_t i = 0; i < n; ++i) f[i] = pxo_half_to_float(h[i]);
}
void pxo_floats_to_halfs(const float* f, uint16_t* h, int64_t n) {
  for (int64_t i = 0; i < n; ++i) h[i] = pxo_float_to_half(f[i]);
}

static inline float load_lowp(const void* p, int dtype, int i) {
  return dtype == PXO_F16 ? pxo_half_to_float(((const uint16_t*)p)[i]) : ((const float*)p)[i];
}

/* ---------------------------------------------------------------------------------
 * Scalar tail shared by both SIMD overloads:
 * base/src/cubic_hermite_spline_simd.h:105-119 and :176-191.
 * --------------------------------------------------------------------------------- */
static inline void spline_tail(double p0, double p1, double p2, double p3, double x, double* f,
                               double* dfdx) {
  double a = 0.5 * (-p0 + 3.0 * (p1 - p2) + p3);
  double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
  double c = 0.5 * (-p0 + p2);
  double d = p1;
  if (f) *f = d + x * (c + x * (b + x * a));
  if (dfdx) *dfdx = c + x * (2.0 * b + 3.0 * a * x);
}

/* cubic_hermite_spline_simd.h:123-175: 8-lane fp32 body, FMA op order preserved. */
void pxo_spline_lowp(const void* p0v, const void* p1v, const void* p2v, const void* p3v,
                     int dtype, int C, double x, int round_out_to_float, double* f,
                     double* dfdx) {
  const float x2s = (float)(x * x);                 /* :128 */
  const float onehalf = 0.5f, three = 3.0f, four = 4.0f, twofive = 2.5f, min1 = -1.0f;
  const float fourx = (float)(4.0f * x);            /* :135 (float*double -> double -> ps) */
  const float xhalf = (float)(x * 0.5f);            /* :136 */
  const float x2 = x2s;
  const float onefivex2 = 1.5f * x2s;               /* :138 */
  const int C8 = C - (C % 8);
  for (int i = 0; i < C8; ++i) {
    float p0 = load_lowp(p0v, dtype, i), p1 = load_lowp(p1v, dtype, i);
    float p2 = load_lowp(p2v, dtype, i), p3 = load_lowp(p3v, dtype, i);
    float t1 = fmaf(three, p1, -p0);   /* fmsub :149 */
    float t2 = fmaf(three, p2, -p3);   /* :150 */
    float t4 = fmaf(four, p2, -p3);    /* :151 */
    float t5 = fmaf(twofive, p1, -p0); /* :152 */
    float t6 = fmaf(min1, p0, p2);     /* :153 */
    float t3 = t1 - t2;                /* :155 */
    float b = fmaf(onehalf, t4, -t5);  /* :158 */
    if (f) {
      float t7 = fmaf(xhalf, t6, p1);  /* :162 */
      float t8 = fmaf(xhalf, t3, b);
      float rf = fmaf(x2, t8, t7);
      f[i] = (double)rf;               /* _mm256_storeu_ps_T widening :31-41 */
    }
    if (dfdx) {
      float t9 = fmaf(fourx, b, t6);   /* :168 */
      float t10 = onefivex2 * t3;
      float rd = fmaf(onehalf, t9, t10);
      dfdx[i] = (double)rd;
    }
  }
  for (int i = C8; i < C; ++i) { /* :176-191 */
    double p0 = (double)load_lowp(p0v, dtype, i), p1 = (double)load_lowp(p1v, dtype, i);
    double p2 = (double)load_lowp(p2v, dtype, i), p3 = (double)load_lowp(p3v, dtype, i);
    double ff, dd;
    spline_tail(p0, p1, p2, p3, x, &ff, &dd);
    if (f) f[i] = round_out_to_float ? (double)(float)ff : ff;
    if (dfdx) dfdx[i] = round_out_to_float ? (double)(float)dd : dd;
  }
}

/* cubic_hermite_spline_simd.h:56-104: 4-lane fp64 body. */
void pxo_spline_f64(const double* p0v, const double* p1v, const double* p2v, const double* p3v,
                    int C, double x, double* f, double* dfdx) {
  const double x2s = x * x;
  const double onehalf = 0.5, three = 3.0, four = 4.0, twofive = 2.5, min1 = -1.0;
  const double fourx = 4.0 * x, xhalf = x * 0.5, x2 = x2s, onefivex2 = 1.5 * x2s;
  const int C4 = C - (C % 4);
  for (int i = 0; i < C4; ++i) {
    double p0 = p0v[i], p1 = p1v[i], p2 = p2v[i], p3 = p3v[i];
    double t1 = fma(three, p1, -p0);
    double t2 = fma(three, p2, -p3);
    double t4 = fma(four, p2, -p3);
    double t5 = fma(twofive, p1, -p0);
    double t6 = fma(min1, p0, p2);
    double t3 = t1 - t2;
    double b = fma(onehalf, t4, -t5);
    if (f) {
      double t7 = fma(xhalf, t6, p1);
      double t8 = fma(xhalf, t3, b);
      f[i] = fma(x2, t8, t7);
    }
    if (dfdx) {
      double t9 = fma(fourx, b, t6);
      double t10 = onefivex2 * t3;
      dfdx[i] = fma(onehalf, t9, t10);
    }
  }
  for (int i = C4; i < C; ++i)
    spline_tail(p0v[i], p1v[i], p2v[i], p3v[i], x, f ? f + i : NULL, dfdx ? dfdx + i : NULL);
}

/* [upstream] ceres::CubicHermiteSpline (ceres/cubic_interpolation.h, Ceres 2.1):
 *   a = 0.5(-p0 + 3p1 - 3p2 + p3); b = 0.5(2p0 - 5p1 + 4p2 - p3); c = 0.5(-p0 + p2); d = p1
 *   f = d + x(c + x(b + x a));  f' = c + x(2b + 3a x)                                   */
void pxo_spline_ceres(const double* p0v, const double* p1v, const double* p2v,
                      const double* p3v, int C, double x, double* f, double* dfdx) {
  for (int i = 0; i < C; ++i) {
    double p0 = p0v[i], p1 = p1v[i], p2 = p2v[i], p3 = p3v[i];
    double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
    double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
    double c = 0.5 * (-p0 + p2);
    double d = p1;
    if (f) f[i] = d + x * (c + x * (b + x * a));
    if (dfdx) dfdx[i] = c + x * (2.0 * b + 3.0 * a * x);
  }
}

/* Grid2D::GetPointer / GetValue border clamping (base/src/grid2d.h:64-73, :34-38);
 * row_begin = col_begin = 0 (features/src/patch_interpolator.h:104). */
static inline int64_t texel_offset(const pxo_patch* p, int r, int c) {
  int row = r < 0 ? 0 : (r > p->H - 1 ? p->H - 1 : r);
  int col = c < 0 ? 0 : (c > p->W - 1 ? p->W - 1 : c);
  return ((int64_t)p->W * row + col) * p->C;
}

static inline size_t dtype_size(int dtype) { return dtype == PXO_F16 ? 2 : (dtype == PXO_F32 ? 4 : 8); }

static void get_value_f64(const pxo_patch* p, int r, int c, double* out) {
  int64_t off = texel_offset(p, r, c);
  for (int i = 0; i < p->C; ++i) {
    if (p->dtype == PXO_F16) out[i] = (double)pxo_half_to_float(((const uint16_t*)p->data)[off + i]);
    else if (p->dtype == PXO_F32) out[i] = (double)((const float*)p->data)[off + i];
    else out[i] = ((const double*)p->data)[off + i];
  }
}

#define PXO_MAXC 512

static void bicubic_ceres_impl(const pxo_patch* p, double r, double c, double* f, double* dfdr, double* dfdc,
                               double* dfdrc);

/* BiCubicInterpolator::Evaluate (base/src/interpolation.h:220-268 scalar, :177-218 SIMD); dfdrc = the cross
 * derivative the reference produces when a fourth pointer is passed: the derivative output of the vertical spline over
 * the row derivatives. */
static void bicubic_impl(const pxo_patch* p, double r, double c, int use_float_simd, double* f,
                         double* dfdr, double* dfdc, double* dfdrc) {
  const int C = p->C;
  const int row = (int)floor(r); /* :179 */
  const int col = (int)floor(c);
  double h[4][PXO_MAXC], hd[4][PXO_MAXC];
  if (C < 8) { /* :222-227 falls through to the Ceres scalar path */
    bicubic_ceres_impl(p, r, c, f, dfdr, dfdc, dfdrc);
    return;
  }
  const char* base = (const char*)p->data;
  const size_t es = dtype_size(p->dtype);
  for (int j = 0; j < 4; ++j) { /* :185-208: rows row-1 .. row+2, cols col-1 .. col+2 */
    const void* q0 = base + es * texel_offset(p, row - 1 + j, col - 1);
    const void* q1 = base + es * texel_offset(p, row - 1 + j, col);
    const void* q2 = base + es * texel_offset(p, row - 1 + j, col + 1);
    const void* q3 = base + es * texel_offset(p, row - 1 + j, col + 2);
    if (p->dtype == PXO_F64) {
      /* exact overload match: the double body; T = dtype of the f_k buffers */
      pxo_spline_f64((const double*)q0, (const double*)q1, (const double*)q2, (const double*)q3, C,
                     c - col, h[j], hd[j]);
      if (use_float_simd) /* _mm256_storeu_pd_T(float*) :27-30 rounds to float */
        for (int i = 0; i < C; ++i) { h[j][i] = (double)(float)h[j][i]; hd[j][i] = (double)(float)hd[j][i]; }
    } else {
      pxo_spline_lowp(q0, q1, q2, q3, p->dtype, C, c - col, use_float_simd, h[j], hd[j]);
    }
  }
  if (!use_float_simd) { /* f_k are Eigen::Matrix<double>: vertical pass = double body :210-217 */
    pxo_spline_f64(h[0], h[1], h[2], h[3], C, r - row, f, dfdr);
    if (dfdc) pxo_spline_f64(hd[0], hd[1], hd[2], hd[3], C, r - row, dfdc, dfdrc);
  } else { /* f_k are Eigen::Matrix<float>: IN_T = float body, outputs widened into double* f */
    float hf[4][PXO_MAXC], hdf[4][PXO_MAXC];
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < C; ++i) { hf[j][i] = (float)h[j][i]; hdf[j][i] = (float)hd[j][i]; }
    pxo_spline_lowp(hf[0], hf[1], hf[2], hf[3], PXO_F32, C, r - row, 0, f, dfdr);
    if (dfdc) pxo_spline_lowp(hdf[0], hdf[1], hdf[2], hdf[3], PXO_F32, C, r - row, 0, dfdc, dfdrc);
  }
}

void pxo_bicubic(const pxo_patch* p, double r, double c, int use_float_simd, double* f,
                 double* dfdr, double* dfdc) {
  bicubic_impl(p, r, c, use_float_simd, f, dfdr, dfdc, NULL);
}

/* [upstream] ceres::BiCubicInterpolator::Evaluate == interpolation.h:228-268. */
void pxo_bicubic_ceres(const pxo_patch* p, double r, double c, double* f, double* dfdr,
                       double* dfdc) {
  bicubic_ceres_impl(p, r, c, f, dfdr, dfdc, NULL);
}

static void bicubic_ceres_impl(const pxo_patch* p, double r, double c, double* f, double* dfdr, double* dfdc,
                               double* dfdrc) {
  const int C = p->C;
  const int row = (int)floor(r);
  const int col = (int)floor(c);
  double p0[PXO_MAXC], p1[PXO_MAXC], p2[PXO_MAXC], p3[PXO_MAXC];
  double h[4][PXO_MAXC], hd[4][PXO_MAXC];
  for (int j = 0; j < 4; ++j) {
    get_value_f64(p, row - 1 + j, col - 1, p0);
    get_value_f64(p, row - 1 + j, col, p1);
    get_value_f64(p, row - 1 + j, col + 1, p2);
    get_value_f64(p, row - 1 + j, col + 2, p3);
    pxo_spline_ceres(p0, p1, p2, p3, C, c - col, h[j], hd[j]);
  }
  pxo_spline_ceres(h[0], h[1], h[2], h[3], C, r - row, f, dfdr);
  if (dfdc) pxo_spline_ceres(hd[0], hd[1], hd[2], hd[3], C, r - row, dfdc, dfdrc);
}

/* PixelInterpolator::Evaluate (base/src/interpolation.h:642-677). */
void pxo_pixel_interp(const pxo_patch* p, double r, double c, const pxo_interp_cfg* cfg,
                      double* f, double* dfdr, double* dfdc) {
  pxo_pixel_interp_cross(p, r, c, cfg, f, dfdr, dfdc, NULL);
}

/* ... with the cross derivative d2f/drdc (:642-646): it comes straight out of the bicubic and is NOT touched by the
 * L2 normalisation (only f, dfdc and dfdr are, :648-666). */
void pxo_pixel_interp_cross(const pxo_patch* p, double r, double c, const pxo_interp_cfg* cfg,
                            double* f, double* dfdr, double* dfdc, double* dfdrc) {
  const int C = p->C;
  double tmp_r[PXO_MAXC];
  /* the reference always computes dfdr (f and dfdr come out of the same spline call) */
  bicubic_impl(p, r, c, cfg->use_float_simd, f, dfdr ? dfdr : tmp_r, dfdc, dfdrc);
  if (cfg->l2_normalize) {
    double ss = 0.0;
    for (int i = 0; i < C; ++i) ss += f[i] * f[i];
    double norm_inv = 1.0 / sqrt(ss); /* :649, no epsilon guard */
    for (int i = 0; i < C; ++i) f[i] *= norm_inv;
    if (dfdc) { /* :653-659 */
      double dot = 0.0;
      for (int i = 0; i < C; ++i) dfdc[i] *= norm_inv;
      for (int i = 0; i < C; ++i) dot += f[i] * dfdc[i];
      for (int i = 0; i < C; ++i) dfdc[i] -= dot * f[i];
    }
    if (dfdr) { /* :661-666 */
      double dot = 0.0;
      for (int i = 0; i < C; ++i) dfdr[i] *= norm_inv;
      for (int i = 0; i < C; ++i) dot += f[i] * dfdr[i];
      for (int i = 0; i < C; ++i) dfdr[i] -= dot * f[i];
    }
  }
}

/* PatchInterpolator::Evaluate (features/src/patch_interpolator.h:125-135) with
 * FeaturePatch::ToPixelCoordinates (features/src/featurepatch.h:250-255) and the Jet
 * bridge (base/src/interpolation.h:130-140):
 *   u = (x*sx - 0.5 - x0)*up ; v = (y*sy - 0.5 - y0)*up ; Evaluate(r = v, c = u)
 *   d/dx = dfdc * sx*up ; d/dy = dfdr * sy*up                                         */
int pxo_patch_eval(const pxo_patch* p, const double xy[2], const pxo_interp_cfg* cfg, double* f,
                   double* dfdx, double* dfdy) {
  const double u = (xy[0] * p->sx - 0.5 - (double)p->x0) * p->up;
  const double v = (xy[1] * p->sy - 0.5 - (double)p->y0) * p->up;
  double gr[PXO_MAXC], gc[PXO_MAXC];
  const int want = dfdx || dfdy;
  pxo_pixel_interp(p, v, u, cfg, f, want ? gr : NULL, want ? gc : NULL);
  if (dfdx)
    for (int i = 0; i < p->C; ++i) dfdx[i] = gc[i] * (p->sx * p->up);
  if (dfdy)
    for (int i = 0; i < p->C; ++i) dfdy[i] = gr[i] * (p->sy * p->up);
  if (cfg->check_bounds) /* patch_interpolator.h:160-166, IsInsideZeroL: 0 < x < L */
    return (u > 0.0 && u < (double)p->W && v > 0.0 && v < (double)p->H) ? 1 : 0;
  return 1;
}
