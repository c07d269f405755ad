// ref_bench_shim.cc -- ORACLE support (test infrastructure only): the REFERENCE legs of bench.py's CPU baselines.
//
// Compiles the reference's OWN evaluation path where it lies under /root/reference -- residuals/src/feature_reference.h
// (FeatureReferenceCostFunctor), residuals/src/featuremetric.h (FeatureMetric2DCostFunctor), features/src/patch_interpolator.h,
// base/src/interpolation.h (BiCubicInterpolator::EvaluateSIMD, the L2 normalisation and the Jet bridge),
// base/src/cubic_hermite_spline_simd.h, base/src/grid2d.h, base/src/projection.h -- against the stub headers of
// oracle/ref_stubs/interp/ (see ref_residual_shim.cc for what is and is not reference code underneath), with the reference's
// Release flags (CMakeLists.txt:51-75: -O3 -DNDEBUG -mavx2 -mf16c -mfma), and times it with the persistent-thread harness of
// pxo_bench_harness.h.  Output: oracle/_ref/libpxo_ref_bench.so.  Nothing of the reference is copied into this repository.
//
// What a timed unit is:
//   pxo_refbench_ba_residual  FeatureReferenceCostFunctor::operator() on ceres::Jet<double, 10 + K> -- residual and the
//                             128 x (10 + K) Jacobian of one residual block, what ceres::AutoDiffCostFunction::Evaluate
//                             costs Ceres per block and evaluation (the functors are constructed once, outside the timed
//                             region, like Problem::AddResidualBlock does); the loss / corrector are not included;
//   pxo_refbench_ka_edge      FeatureMetric2DCostFunctor::operator() on ceres::Jet<double, 4> (one KA residual block);
//   pxo_refbench_bicubic      BiCubicInterpolator::EvaluateSIMD alone (value + both derivatives, 128 channels).
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "residuals/src/feature_reference.h"
#include "residuals/src/featuremetric.h"

#include "pxo_bench_harness.h"

namespace pixsfm {
template <typename dtype>
FeaturePatch<dtype>::FeaturePatch() : data_ptr_(nullptr) {}

template <typename dtype>
struct ViewPatch : public FeaturePatch<dtype> {
  ViewPatch(const void* data, int H, int W, int C, const int* corner, const double* scale) {
    this->data_ptr_ = const_cast<dtype*>(static_cast<const dtype*>(data));
    this->shape_ = {H, W, C};
    this->corner_[0] = corner[0]; this->corner_[1] = corner[1];
    this->scale_[0] = scale[0]; this->scale_[1] = scale[1];
  }
};
}  // namespace pixsfm

namespace {
using pixsfm::InterpolationConfig;
constexpr int C = 128;
typedef colmap::SimpleRadialCameraModel Cam;      // BASELINE configs[2]: SIMPLE_RADIAL
constexpr int K = (int)Cam::kNumParams, N = 10 + K;
typedef ceres::Jet<double, N> JetBA;
typedef pixsfm::FeatureReferenceCostFunctor<Cam, half, C, 1> BaFunctor;
typedef ceres::Jet<double, 4> JetKA;
typedef pixsfm::FeatureMetric2DCostFunctor<half, C, 1> KaFunctor;

struct BaUser {
  int64_t n;
  const uint16_t* arena; int H, W;
  const int64_t* obs_patch; const int32_t* obs_image; const int32_t* obs_point;
  const int32_t* corners; const double* scales;
  const double* qvec; const double* tvec; const double* xyz; const double* cam_params; int cam_stride;
  const int32_t* image_camera; const double* refs;
  InterpolationConfig cfg;
  int local_copies;
  double sink;
};
struct BaState {
  int64_t first, count;
  std::vector<uint16_t> arena;                      // the thread's own copy of its share (first touch)
  std::vector<double> refs;
  std::vector<std::unique_ptr<pixsfm::ViewPatch<half>>> patches;
  std::vector<std::unique_ptr<BaFunctor>> functors;
  std::vector<JetBA> out;
  double sink;
};

void* ba_init(void* up, int t, int T) {
  BaUser* u = static_cast<BaUser*>(up);
  BaState* s = new BaState();
  pxo_bench_share(u->n, t, T, &s->first, &s->count);
  const size_t pe = (size_t)u->H * u->W * C;
  if (u->local_copies) { s->arena.resize(pe * (size_t)s->count); s->refs.resize((size_t)C * s->count); }
  s->out.resize(C);
  for (int64_t i = 0; i < s->count; ++i) {
    const int64_t o = s->first + i, pi = u->obs_patch[o];
    const uint16_t* data = u->arena + pe * (size_t)pi;
    const double* ref = u->refs + (size_t)C * u->obs_point[o];
    if (u->local_copies) {
      std::memcpy(s->arena.data() + pe * (size_t)i, data, pe * 2);
      std::memcpy(s->refs.data() + (size_t)C * i, ref, sizeof(double) * C);
      data = s->arena.data() + pe * (size_t)i;
      ref = s->refs.data() + (size_t)C * i;
    }
    s->patches.emplace_back(new pixsfm::ViewPatch<half>(data, u->H, u->W, C, u->corners + 2 * pi, u->scales + 2 * pi));
    s->functors.emplace_back(new BaFunctor(*s->patches.back(), u->cfg, ref));
  }
  s->sink = 0;
  return s;
}
void ba_work(void* up, void* sp, int, int) {
  BaUser* u = static_cast<BaUser*>(up);
  BaState* s = static_cast<BaState*>(sp);
  double acc = 0;
  JetBA jq[4], jt[3], jX[3], jk[K];
  for (int64_t i = 0; i < s->count; ++i) {
    const int64_t o = s->first + i;
    const int img = u->obs_image[o], cam = u->image_camera[img];
    const double* q = u->qvec + 4 * img; const double* tv = u->tvec + 3 * img;
    const double* X = u->xyz + 3 * (int64_t)u->obs_point[o]; const double* kp = u->cam_params + (size_t)u->cam_stride * cam;
    for (int a = 0; a < 4; ++a) jq[a] = JetBA(q[a], a);
    for (int a = 0; a < 3; ++a) { jt[a] = JetBA(tv[a], 4 + a); jX[a] = JetBA(X[a], 7 + a); }
    for (int a = 0; a < K; ++a) jk[a] = JetBA(kp[a], 10 + a);
    (*s->functors[i])(jq, jt, jX, jk, s->out.data());
    acc += s->out[0].a + s->out[C - 1].v[N - 1];
  }
  s->sink += acc;
}
void ba_fini(void* up, void* sp) {
  BaUser* u = static_cast<BaUser*>(up);
  BaState* s = static_cast<BaState*>(sp);
  u->sink += s->sink;
  delete s;
}

// ---- interpolation only ----------------------------------------------------------------------------------------------
struct BicUser { int64_t n; const uint16_t* arena; int H, W; const int64_t* patch; const double* rc; int local_copies; double sink; };
struct BicState { int64_t first, count; std::vector<uint16_t> arena; double sink; };
void* bic_init(void* up, int t, int T) {
  BicUser* u = static_cast<BicUser*>(up);
  BicState* s = new BicState();
  pxo_bench_share(u->n, t, T, &s->first, &s->count);
  const size_t pe = (size_t)u->H * u->W * C;
  if (u->local_copies) {
    s->arena.resize(pe * (size_t)s->count);
    for (int64_t i = 0; i < s->count; ++i)
      std::memcpy(s->arena.data() + pe * (size_t)i, u->arena + pe * (size_t)u->patch[s->first + i], pe * 2);
  }
  s->sink = 0;
  return s;
}
void bic_work(void* up, void* sp, int, int) {
  BicUser* u = static_cast<BicUser*>(up);
  BicState* s = static_cast<BicState*>(sp);
  using Grid = pixsfm::Grid2D<half, C>;
  const size_t pe = (size_t)u->H * u->W * C;
  double f[C], dr[C], dc[C], acc = 0;
  for (int64_t i = 0; i < s->count; ++i) {
    const int64_t o = s->first + i;
    const half* data = reinterpret_cast<const half*>(u->local_copies ? s->arena.data() + pe * (size_t)i : u->arena + pe * (size_t)u->patch[o]);
    Grid grid(data, 0, u->H, 0, u->W);
    pixsfm::BiCubicInterpolator<Grid> interp(grid);
    interp.EvaluateSIMD(u->rc[2 * o], u->rc[2 * o + 1], f, dr, dc, nullptr);
    acc += f[0] + dr[1] + dc[2];
  }
  s->sink += acc;
}
void bic_fini(void* up, void* sp) {
  static_cast<BicUser*>(up)->sink += static_cast<BicState*>(sp)->sink;
  delete static_cast<BicState*>(sp);
}

// ---- KA residual blocks ------------------------------------------------------------------------------------------------
struct KaUser {
  int64_t n_edges; const int32_t* src; const int32_t* dst; const double* kp; const int64_t* node_patch;
  const uint16_t* arena; int H, W; const int32_t* corners; const double* scales; InterpolationConfig cfg; double sink;
};
struct KaState {
  int64_t first, count;
  std::vector<std::unique_ptr<pixsfm::ViewPatch<half>>> patches;
  std::vector<std::unique_ptr<KaFunctor>> functors;
  std::vector<JetKA> out;
  double sink;
};
void* ka_init(void* up, int t, int T) {
  KaUser* u = static_cast<KaUser*>(up);
  KaState* s = new KaState();
  pxo_bench_share(u->n_edges, t, T, &s->first, &s->count);
  const size_t pe = (size_t)u->H * u->W * C;
  s->out.resize(C);
  for (int64_t i = 0; i < s->count; ++i) {
    const int64_t e = s->first + i;
    const int64_t pa = u->node_patch[u->src[e]], pb = u->node_patch[u->dst[e]];
    s->patches.emplace_back(new pixsfm::ViewPatch<half>(u->arena + pe * (size_t)pa, u->H, u->W, C, u->corners + 2 * pa, u->scales + 2 * pa));
    s->patches.emplace_back(new pixsfm::ViewPatch<half>(u->arena + pe * (size_t)pb, u->H, u->W, C, u->corners + 2 * pb, u->scales + 2 * pb));
    s->functors.emplace_back(new KaFunctor(*s->patches[2 * i], *s->patches[2 * i + 1], u->cfg));
  }
  s->sink = 0;
  return s;
}
void ka_work(void* up, void* sp, int, int) {
  KaUser* u = static_cast<KaUser*>(up);
  KaState* s = static_cast<KaState*>(sp);
  double acc = 0;
  JetKA a[2], b[2];
  for (int64_t i = 0; i < s->count; ++i) {
    const int64_t e = s->first + i;
    const double* ka = u->kp + 2 * (int64_t)u->src[e]; const double* kb = u->kp + 2 * (int64_t)u->dst[e];
    a[0] = JetKA(ka[0], 0); a[1] = JetKA(ka[1], 1); b[0] = JetKA(kb[0], 2); b[1] = JetKA(kb[1], 3);
    (*s->functors[i])(a, b, s->out.data());
    acc += s->out[0].a + s->out[C - 1].v[3];
  }
  s->sink += acc;
}
void ka_fini(void* up, void* sp) {
  static_cast<KaUser*>(up)->sink += static_cast<KaState*>(sp)->sink;
  delete static_cast<KaState*>(sp);
}

int finish(int rc, const pxo_bench_result& r, double* out) {
  if (rc) return rc;
  out[0] = r.seconds; out[1] = (double)r.passes; out[2] = r.calib_seconds; out[3] = r.pinned;
  return 0;
}
}  // namespace

extern "C" {
// fp16 16 x 16 x 128 patches (any H, W), SIMPLE_RADIAL cameras with `cam_stride` doubles per camera; observations [0, n).
// out[0] = seconds of the timed region, out[1] = passes over the n blocks, out[2] = calibration seconds, out[3] = pinned.
int pxo_refbench_ba_residual(int64_t n, const uint16_t* arena, int H, int W, const int64_t* obs_patch, const int32_t* obs_image,
                             const int32_t* obs_point, const int32_t* corners, const double* scales, const double* qvec,
                             const double* tvec, const double* xyz, const double* cam_params, int cam_stride,
                             const int32_t* image_camera, const double* refs, int l2_normalize, int n_threads,
                             double min_seconds, int local_copies, double* out) {
  BaUser u;
  u.n = n; u.arena = arena; u.H = H; u.W = W; u.obs_patch = obs_patch; u.obs_image = obs_image; u.obs_point = obs_point;
  u.corners = corners; u.scales = scales; u.qvec = qvec; u.tvec = tvec; u.xyz = xyz; u.cam_params = cam_params;
  u.cam_stride = cam_stride; u.image_camera = image_camera; u.refs = refs; u.local_copies = local_copies; u.sink = 0;
  u.cfg.l2_normalize = l2_normalize != 0; u.cfg.use_float_simd = false; u.cfg.check_bounds = false;
  pxo_bench_ops ops = {ba_init, ba_work, ba_fini, nullptr};
  pxo_bench_result r;
  return finish(pxo_bench_run(&ops, &u, n_threads, min_seconds, 0, &r), r, out);
}

int pxo_refbench_bicubic(int64_t n, const uint16_t* arena, int H, int W, const int64_t* patch, const double* rc, int n_threads,
                         double min_seconds, int local_copies, double* out) {
  BicUser u{n, arena, H, W, patch, rc, local_copies, 0.0};
  pxo_bench_ops ops = {bic_init, bic_work, bic_fini, nullptr};
  pxo_bench_result r;
  return finish(pxo_bench_run(&ops, &u, n_threads, min_seconds, 0, &r), r, out);
}

int pxo_refbench_ka_edge(int64_t n_edges, const int32_t* src, const int32_t* dst, const double* kp, const int64_t* node_patch,
                         const uint16_t* arena, int H, int W, const int32_t* corners, const double* scales, int l2_normalize,
                         int n_threads, double min_seconds, double* out) {
  KaUser u;
  u.n_edges = n_edges; u.src = src; u.dst = dst; u.kp = kp; u.node_patch = node_patch; u.arena = arena; u.H = H; u.W = W;
  u.corners = corners; u.scales = scales; u.sink = 0;
  u.cfg.l2_normalize = l2_normalize != 0; u.cfg.use_float_simd = false; u.cfg.check_bounds = false;
  pxo_bench_ops ops = {ka_init, ka_work, ka_fini, nullptr};
  pxo_bench_result r;
  return finish(pxo_bench_run(&ops, &u, n_threads, min_seconds, 0, &r), r, out);
}
}  // extern "C"
