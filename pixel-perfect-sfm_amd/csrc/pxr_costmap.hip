// pxr_costmap.hip -- cost-map extraction for gfx950 (MI355X).
//
// CostMapExtractor::FillPointCostmap (bundle_adjustment/src/costmap_extractor.h:230-358) turns the C-channel
// feature patch of an observation into a 3-channel (cost, dcost/dr, dcost/dc) or 1-channel map of the
// featuremetric error against the point's reference descriptor; the cost-map BA
// (costmap_bundle_optimizer.h) then never touches the features again -- the reference's low-memory path
// (configs/low_memory.yaml).  Here: the case the reference takes without interpolation (cost patch of the
// feature patch's size, no cross derivative, :253-279 / :330-340), per texel
//     res = f - ref;  cost = 0.5 rho(|res|^2)[0]
//     dfdr = 0.5 (f[min(H-1, y+1)] - f[max(0, y-1)]),  dfdc likewise along x   (differences in the STORAGE type)
//     dcost/dr = rho' <res, dfdr>, dcost/dc = rho' <res, dfdc>   where cost > 1e-8;  optional sqrt.
//
// One pass over the feature arena: HBM-bound (H*W*C*sizeof(dtype) + 8 C bytes in, 3 H W sizeof(dtype) out per
// observation).  Mapping: one workgroup per patch, a group of C/8 lanes (one DPP row at C = 128) per texel COLUMN
// walking down the rows with the rows y-1, y, y+1 of its column rolling through registers, so every texel
// is fetched from HBM once; the left / right neighbours come from the L1 lines the adjacent groups fetched.
// Each lane owns 8 channels (one 16-byte load per fp16 texel), the three channel reductions are DPP
// row reductions in fp64 (the reference accumulates in double).
#include <hip/hip_runtime.h>

#include "pxr_device.h"
#include "pxr_interp.h"
#include "pxr_internal.h"

namespace pxr {

struct CostmapArgs {
  const void* fin; const int32_t* cin; const double* sin;     // feature arena: data, corners, scales
  void* fout; int32_t* cout; double* sout;                    // cost-map arena
  int H, W, CO, apply_sqrt;
  int64_t first_out;
  const int64_t* patch;       // [n] feature patch of each cost map
  const int32_t* ref_index;   // [n] row of refs (the 3D point)
  const double* refs;         // [*][C]
  pxr_loss loss;
};

// a - b in the storage type (Eigen expression on Map<Matrix<dtype>>, costmap_extractor.h:266-276), widened
template <typename ST> struct StorageDiff;
template <> struct StorageDiff<_Float16> {
  static __device__ __forceinline__ void run(const Texel8<_Float16>& a, const Texel8<_Float16>& b, double out[8]) {
    // four v_pk_add_f16 with the second operand negated (correctly rounded, like half.hpp 2.2.0); written as inline
    // assembly because the compiler otherwise scalarises the vector subtraction into eight v_sub_f16
    const unsigned xa[4] = {a.raw.x, a.raw.y, a.raw.z, a.raw.w}, xb[4] = {b.raw.x, b.raw.y, b.raw.z, b.raw.w};
    union { unsigned u[4]; half8_t h; } d;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d.u[i]) : "v"(xa[i]), "v"(xb[i]));
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (double)d.h[i];
  }
};
template <> struct StorageDiff<float> {
  static __device__ __forceinline__ void run(const Texel8<float>& a, const Texel8<float>& b, double out[8]) {
    float x[8], y[8];
    a.unpack(x); b.unpack(y);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (double)__fsub_rn(x[i], y[i]);
  }
};
template <> struct StorageDiff<double> {
  static __device__ __forceinline__ void run(const Texel8<double>& a, const Texel8<double>& b, double out[8]) {
    double x[8], y[8];
    a.unpack(x); b.unpack(y);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = __dsub_rn(x[i], y[i]);
  }
};

template <typename ST>
__device__ __forceinline__ void widen8(const Texel8<ST>& t, double out[8]) {
  typename Texel8<ST>::work_t w[8];
  t.unpack(w);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (double)w[i];
}

// FeaturePatch::SetEntry (features/src/featurepatch.h:246-248): dtype(double).  half.hpp has no constructor
// from double, so half storage goes through float -- two roundings, reproduced here.
template <typename OT> __device__ __forceinline__ OT store_cast(double v);
template <> __device__ __forceinline__ _Float16 store_cast<_Float16>(double v) { return (_Float16)(float)v; }
template <> __device__ __forceinline__ float store_cast<float>(double v) { return (float)v; }
template <> __device__ __forceinline__ double store_cast<double>(double v) { return v; }

template <typename ST, typename OT, int C, bool GRAD>
__global__ __launch_bounds__(256) void costmap_kernel(const CostmapArgs a) {
  constexpr int LPO = C / 8;
  const int x = threadIdx.x / LPO, sub = threadIdx.x % LPO;   // blockDim.x = W * LPO
  const int64_t i = blockIdx.x;
  const int64_t pi = a.patch[i];
  const int H = a.H, W = a.W;
  const ST* P = reinterpret_cast<const ST*>(a.fin) + (size_t)pi * H * W * C + sub * 8;
  if (threadIdx.x == 0) {   // CreateShallowCostmapFSet, costmap_extractor.h:382-399: corner and scale are the feature patch's
    const int64_t o = a.first_out + i;
    a.cout[2 * o] = a.cin[2 * pi]; a.cout[2 * o + 1] = a.cin[2 * pi + 1];
    a.sout[2 * o] = a.sin[2 * pi]; a.sout[2 * o + 1] = a.sin[2 * pi + 1];
  }
  double ref[8];
  {
    const double* rp = a.refs + (size_t)a.ref_index[i] * C + sub * 8;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) ref[ch] = rp[ch];
  }
  OT* out = reinterpret_cast<OT*>(a.fout) + (size_t)(a.first_out + i) * H * W * a.CO;
  const int xl = x > 0 ? x - 1 : 0, xr = x < W - 1 ? x + 1 : W - 1;

  Texel8<ST> up, cur, down;
  cur.load(P + (size_t)x * C);
  up = cur;
  down.load(P + (size_t)((H > 1 ? 1 : 0) * W + x) * C);
  for (int y = 0; y < H; ++y) {
    Texel8<ST> nxt, L, R;
    const int y2 = y + 2 < H ? y + 2 : H - 1;
    nxt.load(P + (size_t)(y2 * W + x) * C);
    if (GRAD) { L.load(P + (size_t)(y * W + xl) * C); R.load(P + (size_t)(y * W + xr) * C); }
    double f[8], s = 0.0, br = 0.0, bc = 0.0;
    widen8<ST>(cur, f);
    double res[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) { res[ch] = f[ch] - ref[ch]; s = fma(res[ch], res[ch], s); }
    if (GRAD) {
      double dr[8], dc[8];
      StorageDiff<ST>::run(down, up, dr);
      StorageDiff<ST>::run(R, L, dc);
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) { br = fma(res[ch], 0.5 * dr[ch], br); bc = fma(res[ch], 0.5 * dc[ch], bc); }
    }
    if (LPO == 16) { s = row16_sum(s); if (GRAD) { br = row16_sum(br); bc = row16_sum(bc); } }
    else { s = row8_sum(s); if (GRAD) { br = row8_sum(br); bc = row8_sum(bc); } }
    if (sub == 0) {
      double rho[3];
      loss_eval(a.loss.type, a.loss.a, 1.0, s, rho);
      double cost = 0.5 * rho[0];
      OT* o = out + (size_t)(y * W + x) * a.CO;
      if (GRAD) {
        double dcr = 0.0, dcc = 0.0;
        if (cost > 1.0e-8) {   // costmap_extractor.h:300-318
          dcr = rho[1] * br; dcc = rho[1] * bc;
          if (a.apply_sqrt) { cost = sqrt(cost); dcr *= 0.5 / cost; dcc *= 0.5 / cost; }
        }
        o[0] = store_cast<OT>(cost); o[1] = store_cast<OT>(dcr); o[2] = store_cast<OT>(dcc);
      } else {
        if (a.apply_sqrt) cost = sqrt(cost);   // :351-353
        o[0] = store_cast<OT>(cost);
      }
    }
    up = cur; cur = down; down = nxt;
  }
}

// ---- few-channel features (CHANNELS = 3: the extractor's other registered case, costmap_extractor.h:35-37; 1 alike) -----
// Image intensities: a texel is 3 (or 1) values, so one lane does a whole texel -- residual, storage-type central
// differences of its four neighbours, loss, gate, sqrt, cast -- and nothing crosses lanes.  Workgroup = one cost map.
template <typename ST> __device__ __forceinline__ double storage_diff1(ST a, ST b);
template <> __device__ __forceinline__ double storage_diff1<_Float16>(_Float16 a, _Float16 b) { return (double)(_Float16)(a - b); }
template <> __device__ __forceinline__ double storage_diff1<float>(float a, float b) { return (double)__fsub_rn(a, b); }
template <> __device__ __forceinline__ double storage_diff1<double>(double a, double b) { return __dsub_rn(a, b); }

template <typename ST, typename OT, int C, bool GRAD>
__global__ __launch_bounds__(256) void costmap_small_kernel(const CostmapArgs a) {
  const int64_t i = blockIdx.x;
  const int64_t pi = a.patch[i];
  const int H = a.H, W = a.W;
  const ST* P = reinterpret_cast<const ST*>(a.fin) + (size_t)pi * H * W * C;
  if (threadIdx.x == 0) {   // CreateShallowCostmapFSet, costmap_extractor.h:382-399
    const int64_t o = a.first_out + i;
    a.cout[2 * o] = a.cin[2 * pi]; a.cout[2 * o + 1] = a.cin[2 * pi + 1];
    a.sout[2 * o] = a.sin[2 * pi]; a.sout[2 * o + 1] = a.sin[2 * pi + 1];
  }
  double ref[C];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) ref[ch] = a.refs[(size_t)a.ref_index[i] * C + ch];
  OT* out = reinterpret_cast<OT*>(a.fout) + (size_t)(a.first_out + i) * H * W * a.CO;
  for (int t = threadIdx.x; t < H * W; t += blockDim.x) {
    const int y = t / W, x = t - y * W;
    const ST* c = P + (size_t)t * C;
    double s = 0.0, br = 0.0, bc = 0.0;
    if (GRAD) {
      const ST* up = P + (size_t)((y > 0 ? y - 1 : 0) * W + x) * C;
      const ST* dn = P + (size_t)((y < H - 1 ? y + 1 : H - 1) * W + x) * C;
      const ST* lf = P + (size_t)(y * W + (x > 0 ? x - 1 : 0)) * C;
      const ST* rt = P + (size_t)(y * W + (x < W - 1 ? x + 1 : W - 1)) * C;
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        const double res = (double)c[ch] - ref[ch];
        s = fma(res, res, s);
        br = fma(res, 0.5 * storage_diff1<ST>(dn[ch], up[ch]), br);
        bc = fma(res, 0.5 * storage_diff1<ST>(rt[ch], lf[ch]), bc);
      }
    } else {
#pragma unroll
      for (int ch = 0; ch < C; ++ch) { const double res = (double)c[ch] - ref[ch]; s = fma(res, res, s); }
    }
    double rho[3];
    loss_eval(a.loss.type, a.loss.a, 1.0, s, rho);
    double cost = 0.5 * rho[0];
    OT* o = out + (size_t)t * a.CO;
    if (GRAD) {
      double dcr = 0.0, dcc = 0.0;
      if (cost > 1.0e-8) {   // costmap_extractor.h:300-318
        dcr = rho[1] * br; dcc = rho[1] * bc;
        if (a.apply_sqrt) { cost = sqrt(cost); dcr *= 0.5 / cost; dcc *= 0.5 / cost; }
      }
      o[0] = store_cast<OT>(cost); o[1] = store_cast<OT>(dcr); o[2] = store_cast<OT>(dcc);
    } else {
      if (a.apply_sqrt) cost = sqrt(cost);   // :351-353
      o[0] = store_cast<OT>(cost);
    }
  }
}

// ---- fp16 features, C = 128, square patches: 8 x 8 (low_memory.yaml) and the cost-only 16 x 16 maps --------------------
// (16 x 16 with gradients, the default, has its own kernel further down: costmap_kernel_f16_split)
// The generic kernel above is ALU-bound (fp64 channel math + a 16-lane all-reduce and a one-lane epilogue per
// texel: 0.34 of the HBM peak).  This one
//   * is persistent (two workgroups per CU) and prefetches the NEXT patch's column into registers while the
//     current one is processed: 64 KB per workgroup in flight during the arithmetic;
//   * keeps its whole texel column in registers (vertical neighbours are free) and exchanges the horizontal
//     neighbours through one LDS copy of the patch (16-byte conflict-free reads);
//   * replaces the 3 x PS all-reduces by a reduce-scatter over the DPP row (15 combine steps per 16 row sums
//     instead of 64): lane s of a column ends up with the three sums of ROW s, so the loss, the
//     `cost > 1e-8` branch, the sqrt and the stores run once per column with all 16 lanes busy.
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_masked(double old, double src) {   // lanes outside BANK keep `old`
  union { double d; int i[2]; } o, s, r;
  o.d = old; s.d = src;
  r.i[0] = __builtin_amdgcn_update_dpp(o.i[0], s.i[0], CTRL, 0xf, BANK, false);
  r.i[1] = __builtin_amdgcn_update_dpp(o.i[1], s.i[1], CTRL, 0xf, BANK, false);
  return r.d;
}

// One reduce-scatter step at lane distance D: A belongs to the rows whose index has bit D clear, B to those with
// it set.  Lanes with bit D clear keep A and add the partner's A; the others keep B and add the partner's B.
template <int D>
__device__ __forceinline__ double rs_combine(double A, double B, int sub) {
  if constexpr (D == 8) {          // banks 0,1 <-> banks 2,3, partner = row_ror:8
    const double mine = dpp_masked<0xE4, 0xC>(A, B);
    const double recv = dpp_masked<0x128, 0x3>(dpp_masked<0x128, 0xC>(0.0, B), A);
    return mine + recv;
  } else if constexpr (D == 4) {   // banks 0,2 read lane + 4 (row_shl:4), banks 1,3 read lane - 4 (row_shr:4)
    const double mine = dpp_masked<0xE4, 0xA>(A, B);
    const double recv = dpp_masked<0x104, 0x5>(dpp_masked<0x114, 0xA>(0.0, B), A);
    return mine + recv;
  } else {                         // inside a quad: quad_perm + select
    const bool hi = (sub & D) != 0;
    const double mine = hi ? B : A, send = hi ? A : B;
    return mine + dpp_f64<D == 2 ? 0x4E : 0xB1>(send);
  }
}

struct Sums3 { double s, br, bc; };

template <typename OT, int PS, bool GRAD>
struct CostmapColumn {
  const uint4* col;      // this lane's column: PS texels x 8 channels
  const uint4* sh;       // LDS copy of the patch, [y][x][sub]
  const double* ref;     // 8 channels of the reference
  int xl, xr, sub;

  template <int Y>
  __device__ __forceinline__ Sums3 row() const {
    Texel8<_Float16> cur, up, down;
    cur.raw = col[Y]; up.raw = col[Y > 0 ? Y - 1 : 0]; down.raw = col[Y < PS - 1 ? Y + 1 : PS - 1];
    double f[8], res[8];
    widen8<_Float16>(cur, f);
    Sums3 o = {0.0, 0.0, 0.0};
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) { res[ch] = f[ch] - ref[ch]; o.s = fma(res[ch], res[ch], o.s); }
    if (GRAD) {
      Texel8<_Float16> L, R;
      L.raw = sh[(Y * PS + xl) * 16 + sub]; R.raw = sh[(Y * PS + xr) * 16 + sub];
      double dr[8], dc[8];
      StorageDiff<_Float16>::run(down, up, dr);
      StorageDiff<_Float16>::run(R, L, dc);
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) { o.br = fma(res[ch], dr[ch], o.br); o.bc = fma(res[ch], dc[ch], o.bc); }
    }
    return o;
  }
  // reduce-scattered sums of the rows {J, J + STRIDE, J + 2 STRIDE, ...}: depth-first, so only ~log(PS) partial
  // results are alive at a time
  template <int J, int STRIDE>
  __device__ __forceinline__ Sums3 node() const {
    if constexpr (STRIDE >= PS) {
      Sums3 v = row<J>();
      if constexpr (PS == 8) {    // 8 rows on 16 lanes: both halves of the DPP row hold the row sums
        v.s += dpp_f64<0x128>(v.s);
        if (GRAD) { v.br += dpp_f64<0x128>(v.br); v.bc += dpp_f64<0x128>(v.bc); }
      }
      return v;
    } else {
      const Sums3 a = node<J, 2 * STRIDE>(), b = node<J + STRIDE, 2 * STRIDE>();
      Sums3 o;
      o.s = rs_combine<STRIDE>(a.s, b.s, sub);
      o.br = GRAD ? rs_combine<STRIDE>(a.br, b.br, sub) : 0.0;
      o.bc = GRAD ? rs_combine<STRIDE>(a.bc, b.bc, sub) : 0.0;
      return o;
    }
  }
};

// LDSN: the horizontal neighbours come from an LDS copy of the patch (two barriers per patch, 64 KB per workgroup: two
// workgroups per CU).  !LDSN re-reads them from global memory (the lines the adjacent lane groups just fetched): no LDS,
// no barrier -- measured in round 2 at 10.6 ms per 200k maps against 4.26 ms, at two and at three wavefronts per SIMD
// alike (48 instead of 16 vector-memory instructions per lane and patch; the texture path, not the occupancy, bounds it).
// Kept as a template switch of the body for that record; only the LDS form is instantiated.
template <typename OT, int PS, bool GRAD, bool LDSN>
__device__ __forceinline__ void costmap_f16_body(const CostmapArgs& a, const int64_t n, uint4* sh) {
  constexpr int C = 128;
  const int x = threadIdx.x >> 4, sub = threadIdx.x & 15;
  const _Float16* fin = reinterpret_cast<const _Float16*>(a.fin);
  uint4 col[PS], nxt[PS];
  double ref[8], refn[8];
  int64_t pi = 0, pin = 0;
  auto fetch = [&](int64_t i, uint4* c, double* r, int64_t& p) {
    p = a.patch[i];
    const _Float16* P = fin + (size_t)p * PS * PS * C + (size_t)x * C + sub * 8;
#pragma unroll
    for (int y = 0; y < PS; ++y) c[y] = *reinterpret_cast<const uint4*>(P + (size_t)y * PS * C);
    const double* rp = a.refs + (size_t)a.ref_index[i] * C + sub * 8;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) r[ch] = rp[ch];
  };
  int64_t i = blockIdx.x;
  if (i < n) fetch(i, col, ref, pi);
  for (; i < n; i += gridDim.x) {
    const int64_t inext = i + gridDim.x;
    if (inext < n) fetch(inext, nxt, refn, pin);     // in flight during this patch's arithmetic
    if (GRAD && LDSN) {
      __syncthreads();                               // the previous patch's neighbour reads are done
#pragma unroll
      for (int y = 0; y < PS; ++y) sh[(y * PS + x) * 16 + sub] = col[y];
      __syncthreads();
    }
    const int64_t o_idx = a.first_out + i;
    if (threadIdx.x == 0) {
      a.cout[2 * o_idx] = a.cin[2 * pi]; a.cout[2 * o_idx + 1] = a.cin[2 * pi + 1];
      a.sout[2 * o_idx] = a.sin[2 * pi]; a.sout[2 * o_idx + 1] = a.sin[2 * pi + 1];
    }
    CostmapColumn<OT, PS, GRAD> cc;
    cc.col = col; cc.ref = ref; cc.sub = sub;
    cc.sh = LDSN ? sh : reinterpret_cast<const uint4*>(fin + (size_t)pi * PS * PS * C);   // same [y][x][sub] layout
    cc.xl = x > 0 ? x - 1 : 0; cc.xr = x < PS - 1 ? x + 1 : PS - 1;
    const Sums3 v = cc.template node<0, 1>();        // lane `sub` holds the sums of row `sub` of column x
    if (sub < PS) {
      double rho[3];
      loss_eval(a.loss.type, a.loss.a, 1.0, v.s, rho);
      double cost = 0.5 * rho[0];
      OT* o = reinterpret_cast<OT*>(a.fout) + ((size_t)o_idx * PS * PS + (size_t)sub * PS + x) * a.CO;
      if (GRAD) {
        double dcr = 0.0, dcc = 0.0;
        if (cost > 1.0e-8) {
          dcr = rho[1] * (0.5 * v.br); dcc = rho[1] * (0.5 * v.bc);   // sum(res * 0.5 d) == 0.5 * sum(res * d), exactly
          if (a.apply_sqrt) { cost = sqrt(cost); dcr *= 0.5 / cost; dcc *= 0.5 / cost; }
        }
        o[0] = store_cast<OT>(cost); o[1] = store_cast<OT>(dcr); o[2] = store_cast<OT>(dcc);
      } else {
        if (a.apply_sqrt) cost = sqrt(cost);
        o[0] = store_cast<OT>(cost);
      }
    }
    if (inext < n) {
#pragma unroll
      for (int y = 0; y < PS; ++y) col[y] = nxt[y];
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) ref[ch] = refn[ch];
      pi = pin;
    }
  }
}

template <typename OT, int PS, bool GRAD>
__global__ __launch_bounds__(PS * 16) __attribute__((amdgpu_waves_per_eu(2, 2))) void costmap_kernel_f16(const CostmapArgs a, const int64_t n) {
  __shared__ uint4 sh[GRAD ? PS * PS * 16 : 1];
  costmap_f16_body<OT, PS, GRAD, true>(a, n, sh);
}
// ---- 16 x 16 patches with gradients (the default cost-map configuration): the CHANNELS are split over the wavefronts ------
// Wavefront w owns channels [32 w, 32 w + 32): lane = (k, x), k = lane >> 4 the 8-channel group, x = lane & 15 the texel
// COLUMN, whose 16 rows sit in registers.  So
//   * the left / right neighbours are the adjacent lanes of the DPP row (row_shr:1 / row_shl:1 with the lane's own value as
//     the out-of-range fill = the reference's clamped index): no LDS copy of the patch, no barrier around it;
//   * the channel sums of four rows are reduce-scattered over the four 16-lane rows of the wavefront with the gfx950 half /
//     row exchanges (v_permlane32_swap, v_permlane16_swap: 2 exchanges + 1 add per fp64 pair), DPP-row k ends up with
//     texel row 4 g + k;
//   * the four wavefronts' partial sums (3 x 256 doubles each) meet in 24 KB of LDS (double-buffered: ONE barrier per patch)
//     and thread t finishes texel t: loss, `cost > 1e-8` branch, sqrt and the stores run on all 256 lanes;
//   * ROLLING REFILL: a row register is reloaded with the NEXT patch's row right after its last use (the rows above
//     4 g + 3 once row group g is done), so sixteen 16-byte loads per lane are always in flight (s_waitcnt vmcnt(13..18))
//     and every load has a whole patch period to land, without a second set of registers.  For the counter waits to be
//     exact the memory side of the loop is branch-free (indices past the end are clamped), the loads are pinned where
//     they are written (sched_barrier) and nothing else uses the vector-memory counter in the loop body: the patch index
//     and the reference descriptor are fetched two patches ahead, the descriptor goes through LDS, the corner / scale
//     copy has its own tiny kernel.  The patch base lives in a buffer descriptor (SGPRs), the lane offset in one VGPR.
// Measured on MI355X (200k maps, back to back): 3.10 ms against 3.89 ms for the LDS-staged kernel above = 4.4 TB/s
// algorithmic, 0.55 of the HBM peak.  What bounds it is the vector ALU at the POWER limit: ~1800 vector instructions per
// lane and patch (43 % of them the two-step half -> float -> double conversions of the reference's contract), every one
// of which issues in 4 cycles (tools/valu_rate_probe.hip: fp64 fma / add / cvt, packed fp16 and DPP moves all 4 cycles
// per wavefront, only plain fp32 runs at 2), and the package sits at 1366 W with the shader clock throttled from 2.39 to
// 2.10 GHz while it runs (profiles/r2_costmap_clock_power.txt): 4 waves x ~8200 cycles per patch and CU / 2.10 GHz =
// 3.9 us, x 781 patches per CU = 3.05 ms.  Without the loads the same arithmetic takes 2.84 ms, the loads alone 2.72 ms.
__device__ __forceinline__ double swap_sum32(double A, double B) {   // lanes < 32: A + A(lane + 32); lanes >= 32: B + B(lane - 32)
  union { double d; unsigned u[2]; } a, b;
  a.d = A; b.d = B;
  const auto r0 = __builtin_amdgcn_permlane32_swap(a.u[0], b.u[0], false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(a.u[1], b.u[1], false, false);
  a.u[0] = r0[0]; b.u[0] = r0[1]; a.u[1] = r1[0]; b.u[1] = r1[1];
  return a.d + b.d;
}
__device__ __forceinline__ double swap_sum16(double A, double B) {   // even rows: A + A(row + 1); odd rows: B + B(row - 1)
  union { double d; unsigned u[2]; } a, b;
  a.d = A; b.d = B;
  const auto r0 = __builtin_amdgcn_permlane16_swap(a.u[0], b.u[0], false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(a.u[1], b.u[1], false, false);
  a.u[0] = r0[0]; b.u[0] = r0[1]; a.u[1] = r1[0]; b.u[1] = r1[1];
  return a.d + b.d;
}

// ---- the three 8-channel partial sums of one texel: res = f - ref,  <res, res>,  <res, down - up>,  <res, right - left> ---------
// (differences of neighbours in the STORAGE type: v_pk_add_f16, like the reference's Eigen expression on Map<half>)
// F32 = false: fp64 throughout (the reference's accumulation type): the path of float / double cost maps.
// F32 = true : HALF cost maps (the only shipped configuration, configs/low_memory.yaml: dtype half).  The map is rounded to 2^-11
//   relative on store, so the 8-channel partial sums of a lane are formed in fp32 -- packed v_pk_add_f32 / v_pk_fma_f32 on the
//   widened halves: 54 instead of 88 vector instructions per lane and texel, the conversions half -> float -> double were 43 % of
//   the kernel -- and everything across lanes (16 lanes x 4 wavefronts per texel) stays fp64.  A partial sum of eight products
//   carries ~2e-7 relative; the texel's sum of 16 such partials ~5e-8: the fp16 result differs from the all-fp64 one only where
//   the exact value lies that close to a rounding boundary (measured: tests/test_costmap_*; rounds 4-5 also against a stand-in build of the reference that round 6 removed).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: an array of these is split into registers
template <bool F32> struct TexelRef;
template <> struct TexelRef<false> { double v[8]; __device__ __forceinline__ void set(int ch, double x) { v[ch] = x; } };
template <> struct TexelRef<true> { float v[8]; __device__ __forceinline__ void set(int ch, double x) { v[ch] = (float)x; } };

__device__ __forceinline__ half8_t half_diff8(const u32x4& a, const u32x4& b) {
  union { unsigned u[4]; half8_t h; } d;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d.u[i]) : "v"(a[i]), "v"(b[i]));
  return d.h;
}
template <bool F32>
__device__ __forceinline__ void texel_sums(const u32x4& cu, const u32x4& dn, const u32x4& up, const u32x4& rt,
                                           const u32x4& lf, const TexelRef<F32>& ref, double& ss, double& sr, double& sc) {
  if constexpr (F32) {
    union { unsigned u[4]; half8_t h; } c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.u[i] = cu[i];
    const float8_t f = __builtin_convertvector(c.h, float8_t);
    const float8_t dr = __builtin_convertvector(half_diff8(dn, up), float8_t);
    const float8_t dc = __builtin_convertvector(half_diff8(rt, lf), float8_t);
    f32x2 as = {0.f, 0.f}, ar = {0.f, 0.f}, ac = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 res = {f[2 * i] - ref.v[2 * i], f[2 * i + 1] - ref.v[2 * i + 1]};
      const f32x2 r2 = {dr[2 * i], dr[2 * i + 1]}, c2 = {dc[2 * i], dc[2 * i + 1]};
      as = __builtin_elementwise_fma(res, res, as);
      ar = __builtin_elementwise_fma(res, r2, ar);
      ac = __builtin_elementwise_fma(res, c2, ac);
    }
    ss = (double)(as.x + as.y); sr = (double)(ar.x + ar.y); sc = (double)(ac.x + ac.y);
  } else {
    auto tex = [](const u32x4& v) { Texel8<_Float16> t; t.raw = make_uint4(v.x, v.y, v.z, v.w); return t; };
    double f[8], dr[8], dc[8];
    widen8<_Float16>(tex(cu), f);
    StorageDiff<_Float16>::run(tex(dn), tex(up), dr);
    StorageDiff<_Float16>::run(tex(rt), tex(lf), dc);
    ss = 0.0; sr = 0.0; sc = 0.0;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const double res = f[ch] - ref.v[ch];
      ss = fma(res, res, ss); sr = fma(res, dr[ch], sr); sc = fma(res, dc[ch], sc);
    }
  }
}

// CreateShallowCostmapFSet, costmap_extractor.h:382-399: corner and scale of a cost map are the feature patch's
__global__ void costmap_meta_kernel(const CostmapArgs a, const int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p = a.patch[i], o = a.first_out + i;
  a.cout[2 * o] = a.cin[2 * p]; a.cout[2 * o + 1] = a.cin[2 * p + 1];
  a.sout[2 * o] = a.sin[2 * p]; a.sout[2 * o + 1] = a.sin[2 * p + 1];
}


template <typename OT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void costmap_kernel_f16_split(const CostmapArgs a, const int64_t n) {
  constexpr int PS = 16, C = 128;
  constexpr int ROW_BYTES = PS * C * (int)sizeof(_Float16), PATCH_BYTES = PS * ROW_BYTES;
  __shared__ double part[2][4][3][PS * PS];
  __shared__ double refsh[2][C];
  const int tid = threadIdx.x, w = tid >> 6, k = (tid >> 4) & 3, x = tid & 15;
  const _Float16* fin = reinterpret_cast<const _Float16*>(a.fin);
  const unsigned lane_off = (unsigned)((x * C + 32 * w + 8 * k) * sizeof(_Float16));   // bytes; uniform base + 32-bit lane offset
  const int64_t G = gridDim.x;
  u32x4 col[PS];
  constexpr bool F32 = sizeof(OT) == 2;               // half maps: fp32 partial sums inside a lane (texel_sums)
  TexelRef<F32> ref;
  // The loop body is branch-free on the memory side (indices past the end are clamped to the last patch: one redundant
  // patch load per workgroup at the very end), so the vector-memory counter waits are exact: loads return in order, and a
  // branch around a load would force the compiler to wait for ALL outstanding loads at the next use.
  auto uniform64 = [](int64_t v) {   // the index is the same in every lane: keep it (and the addresses made of it) in SGPRs
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
  };
  int64_t i = blockIdx.x;
  if (i >= n) return;
  auto at = [&](int64_t j) { return j < n ? j : n - 1; };
  int64_t p1 = uniform64(a.patch[at(i + G)]);
  int r1 = __builtin_amdgcn_readfirstlane(a.ref_index[at(i + G)]);
  {
    // same issue order as in the loop (rows 0 .. 15, then the next reference): the wait counts the compiler derives at the
    // loop head are the minimum over both ways into it
    const double* rp = a.refs + (size_t)a.ref_index[i] * C + 32 * w + 8 * k;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) ref.set(ch, rp[ch]);
    const auto P = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(fin + (size_t)uniform64(a.patch[i]) * PS * PS * C), 0, PATCH_BYTES, 0x00020000);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int y = 0; y < PS; ++y) {
      col[y] = __builtin_amdgcn_raw_buffer_load_b128(P, lane_off, y * ROW_BYTES, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double rnext = a.refs[(size_t)r1 * C + (tid & (C - 1))];      // reference descriptor of patch i + G, handed on through LDS
  __builtin_amdgcn_sched_barrier(0);
  int buf = 0;
  for (; i < n; i += G, buf ^= 1) {
    // indices two patches ahead (consumed at the end of this iteration, when they are the OLDEST loads in flight)
    const int64_t i2 = at(i + 2 * G);
    const int64_t p2v = a.patch[i2];
    const int r2v = a.ref_index[i2];
    // the NEXT patch: its rows replace this patch's rows in the registers as soon as the arithmetic below has used them
    // for the last time, so every row load has a whole patch period to land
    // buffer addressing: the patch base lives in four SGPRs, the lane's offset in ONE VGPR for all sixteen rows
    const auto Pn = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(fin + (size_t)p1 * PS * PS * C), 0, PATCH_BYTES, 0x00020000);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double s[4], br[4], bc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int y = 4 * g + r;
        const u32x4 cu = col[y];
        u32x4 lf, rt;
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // clamped horizontal neighbours: a lane without a source keeps its own texel
          lf[q] = (unsigned)__builtin_amdgcn_update_dpp((int)cu[q], (int)cu[q], 0x111, 0xf, 0xf, false);   // row_shr:1 -> x - 1
          rt[q] = (unsigned)__builtin_amdgcn_update_dpp((int)cu[q], (int)cu[q], 0x101, 0xf, 0xf, false);   // row_shl:1 -> x + 1
        }
        texel_sums<F32>(cu, col[y < PS - 1 ? y + 1 : PS - 1], col[y > 0 ? y - 1 : 0], rt, lf, ref, s[r], br[r], bc[r]);
      }
      // rows 4g + 3 and 4g + 4 are still neighbours of the next group; everything above them is dead
      __builtin_amdgcn_sched_barrier(0);                // keep the refill HERE: the scheduler otherwise sinks it to the end
#pragma unroll
      for (int y = (g == 0 ? 0 : 4 * g - 1); y < (g == 3 ? PS : 4 * g + 3); ++y) {
        col[y] = __builtin_amdgcn_raw_buffer_load_b128(Pn, lane_off, y * ROW_BYTES, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // rows (0, 2) and (1, 3) over the half-waves, then the two results over the row pairs: DPP-row k holds texel row 4 g + k
      const int t = (4 * g + k) * PS + x;
      part[buf][w][0][t] = swap_sum16(swap_sum32(s[0], s[2]), swap_sum32(s[1], s[3]));
      part[buf][w][1][t] = swap_sum16(swap_sum32(br[0], br[2]), swap_sum32(br[1], br[3]));
      part[buf][w][2][t] = swap_sum16(swap_sum32(bc[0], bc[2]), swap_sum32(bc[1], bc[3]));
    }
    if (tid < C) refsh[buf][tid] = rnext;
    const int64_t p2 = uniform64(p2v);
    const int r2 = __builtin_amdgcn_readfirstlane(r2v);
    rnext = a.refs[(size_t)r2 * C + (tid & (C - 1))];
    __syncthreads();
    double v[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) v[q] = (part[buf][0][q][tid] + part[buf][1][q][tid]) + (part[buf][2][q][tid] + part[buf][3][q][tid]);
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) ref.set(ch, refsh[buf][32 * w + 8 * k + ch]);
    const int64_t o_idx = a.first_out + i;
    p1 = p2;
    double rho[3];
    loss_eval(a.loss.type, a.loss.a, 1.0, v[0], rho);
    double cost = 0.5 * rho[0], dcr = 0.0, dcc = 0.0;
    if (cost > 1.0e-8) {                               // costmap_extractor.h:300-318
      dcr = rho[1] * (0.5 * v[1]); dcc = rho[1] * (0.5 * v[2]);   // sum(res * 0.5 d) == 0.5 * sum(res * d), exactly
      if (a.apply_sqrt) { cost = sqrt(cost); dcr *= 0.5 / cost; dcc *= 0.5 / cost; }
    }
    OT* o = reinterpret_cast<OT*>(a.fout) + ((size_t)o_idx * PS * PS + tid) * 3;
    o[0] = store_cast<OT>(cost); o[1] = store_cast<OT>(dcr); o[2] = store_cast<OT>(dcc);
  }
}

// ---- 8 x 8 patches with gradients: what the reference's only `strategy: costmaps` configuration uses (configs/low_memory.yaml,
// patch_size 8).  Same channel split over the four wavefronts; a wavefront's 16-lane row now holds the 8 columns x the two
// HALVES of the patch: lane = 16 k + 2 x + yh, four rows (4 yh .. 4 yh + 3) of column x per lane.  Horizontal neighbours are
// two lanes away (row_shr:2 / row_shl:2: the row ends clamp by themselves), the rows across the half boundary come from the
// partner lane (quad_perm [1,0,3,2]) once per patch; one reduce-scatter group per patch leaves DPP-row k with texel row
// 4 yh + k.  Only 16 data registers per lane, so the next patch is simply loaded a whole iteration ahead into a second set.
// Measured on MI355X (200k maps back to back): 0.88 ms against 1.14 ms for the LDS-staged kernel = 4.0 TB/s algorithmic,
// 0.50 of the HBM peak (the per-patch barrier and epilogue weigh four times more than at 16 x 16).
template <typename OT>
__global__ __launch_bounds__(256) void costmap_kernel_f16_split8(const CostmapArgs a, const int64_t n) {
  constexpr int PS = 8, C = 128, RPL = 4;
  constexpr int ROW_BYTES = PS * C * (int)sizeof(_Float16), PATCH_BYTES = PS * ROW_BYTES;
  __shared__ double part[2][4][3][PS * PS];
  __shared__ double refsh[2][C];
  const int tid = threadIdx.x, w = tid >> 6, k = (tid >> 4) & 3, x = (tid >> 1) & 7, yh = tid & 1;
  const _Float16* fin = reinterpret_cast<const _Float16*>(a.fin);
  const unsigned lane_off = (unsigned)(((4 * yh * PS + x) * C + 32 * w + 8 * k) * sizeof(_Float16));
  const int64_t G = gridDim.x;
  auto uniform64 = [](int64_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
  };
  int64_t i = blockIdx.x;
  if (i >= n) return;
  auto at = [&](int64_t j) { return j < n ? j : n - 1; };
  auto load_rows = [&](int64_t p, u32x4* c) {
    const auto P = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(fin + (size_t)p * PS * PS * C), 0, PATCH_BYTES, 0x00020000);
#pragma unroll
    for (int r = 0; r < RPL; ++r) c[r] = __builtin_amdgcn_raw_buffer_load_b128(P, lane_off, r * ROW_BYTES, 0);
  };
  u32x4 col[RPL], nxt[RPL];
  constexpr bool F32 = sizeof(OT) == 2;               // half maps: fp32 partial sums inside a lane (texel_sums)
  TexelRef<F32> ref;
  {
    const double* rp = a.refs + (size_t)a.ref_index[i] * C + 32 * w + 8 * k;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) ref.set(ch, rp[ch]);
  }
  load_rows(uniform64(a.patch[i]), col);
  int64_t p1 = uniform64(a.patch[at(i + G)]);
  int r1 = __builtin_amdgcn_readfirstlane(a.ref_index[at(i + G)]);
  double rnext = a.refs[(size_t)r1 * C + (tid & (C - 1))];
  int buf = 0;
  for (; i < n; i += G, buf ^= 1) {
    const int64_t i2 = at(i + 2 * G);
    const int64_t p2v = a.patch[i2];
    const int r2v = a.ref_index[i2];
    load_rows(p1, nxt);                                // the next patch: a whole iteration to land
    // the rows across the half boundary: the partner lane's bottom row (for the upper half: its top row)
    u32x4 edge;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mine = yh ? (int)col[0][q] : (int)col[RPL - 1][q];          // what the partner needs from this lane
      edge[q] = (unsigned)__builtin_amdgcn_update_dpp(0, mine, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    }
    double s[RPL], br[RPL], bc[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const u32x4 cu = col[r];
      u32x4 lf, rt;
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // clamped horizontal neighbours, two lanes away
        lf[q] = (unsigned)__builtin_amdgcn_update_dpp((int)cu[q], (int)cu[q], 0x112, 0xf, 0xf, false);   // row_shr:2 -> x - 1
        rt[q] = (unsigned)__builtin_amdgcn_update_dpp((int)cu[q], (int)cu[q], 0x102, 0xf, 0xf, false);   // row_shl:2 -> x + 1
      }
      // vertical neighbours: inside the lane, across the half boundary from the partner, clamped at the patch border
      u32x4 up = r > 0 ? col[r - 1] : col[0], dn = r < RPL - 1 ? col[r + 1] : col[RPL - 1];
      if (r == 0) { for (int q = 0; q < 4; ++q) up[q] = yh ? edge[q] : up[q]; }
      if (r == RPL - 1) { for (int q = 0; q < 4; ++q) dn[q] = yh ? dn[q] : edge[q]; }
      texel_sums<F32>(cu, dn, up, rt, lf, ref, s[r], br[r], bc[r]);
    }
    const int t = (4 * yh + k) * PS + x;
    part[buf][w][0][t] = swap_sum16(swap_sum32(s[0], s[2]), swap_sum32(s[1], s[3]));
    part[buf][w][1][t] = swap_sum16(swap_sum32(br[0], br[2]), swap_sum32(br[1], br[3]));
    part[buf][w][2][t] = swap_sum16(swap_sum32(bc[0], bc[2]), swap_sum32(bc[1], bc[3]));
    if (tid < C) refsh[buf][tid] = rnext;
    const int64_t p2 = uniform64(p2v);
    const int r2 = __builtin_amdgcn_readfirstlane(r2v);
    rnext = a.refs[(size_t)r2 * C + (tid & (C - 1))];
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) ref.set(ch, refsh[buf][32 * w + 8 * k + ch]);
    if (tid < PS * PS) {
      double v[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) v[q] = (part[buf][0][q][tid] + part[buf][1][q][tid]) + (part[buf][2][q][tid] + part[buf][3][q][tid]);
      double rho[3];
      loss_eval(a.loss.type, a.loss.a, 1.0, v[0], rho);
      double cost = 0.5 * rho[0], dcr = 0.0, dcc = 0.0;
      if (cost > 1.0e-8) {                               // costmap_extractor.h:300-318
        dcr = rho[1] * (0.5 * v[1]); dcc = rho[1] * (0.5 * v[2]);
        if (a.apply_sqrt) { cost = sqrt(cost); dcr *= 0.5 / cost; dcc *= 0.5 / cost; }
      }
      OT* o = reinterpret_cast<OT*>(a.fout) + ((size_t)(a.first_out + i) * PS * PS + tid) * 3;
      o[0] = store_cast<OT>(cost); o[1] = store_cast<OT>(dcr); o[2] = store_cast<OT>(dcc);
    }
    p1 = p2;
#pragma unroll
    for (int r = 0; r < RPL; ++r) col[r] = nxt[r];
  }
}

// ---- the INTERPOLATING branch of FillPointCostmap (costmap_extractor.h:280-284,341-345) --------------------------------
// Taken by the reference when the cost patch is not of the feature patch's size (CostMapConfig.upsampling_factor != 1) or
// compute_cross_derivative is set: per output texel (y, x) the features are interpolated at the LOCAL patch coordinates
// (x, y) / upsampling_factor with the extractor's InterpolationConfig (PatchInterpolator::EvaluateLocal: bicubic, L2
// normalisation with its chain rule -- unlike the branch above, which reads raw texels), then
//   cost = 0.5 rho(|res|^2)[0];  where cost > 1e-8:  dcost/dr = rho' <res, df/dr>, dcost/dc likewise,
//   d2cost/drdc = 2 rho'' <res, df/dr> <res, df/dc> + rho' (<df/dr, df/dc> + <d2f/drdc, res>)      (:304-308)
// and the sqrt variants (:309-317).  A rarely used configuration (neither pixsfm's default_conf nor low_memory.yaml
// sets it): one workgroup per patch, a group of C / 8 lanes per output texel, the 64 KiB patch served by L1 / L2.
template <typename ST, typename OT, int C, bool FS>
__global__ __launch_bounds__(256) void costmap_interp_kernel(const CostmapArgs a, int Ho, int Wo, double inv_up, int grad, int cross,
                                                             int l2_normalize) {
  constexpr int LPO = C / 8, G = 256 / LPO;
  const int grp = threadIdx.x / LPO, sub = threadIdx.x % LPO;
  const int64_t i = blockIdx.x;
  const int64_t pi = a.patch[i];
  const ST* P = reinterpret_cast<const ST*>(a.fin) + (size_t)pi * a.H * a.W * C;
  if (threadIdx.x == 0) {
    const int64_t o = a.first_out + i;
    a.cout[2 * o] = a.cin[2 * pi]; a.cout[2 * o + 1] = a.cin[2 * pi + 1];
    a.sout[2 * o] = a.sin[2 * pi]; a.sout[2 * o + 1] = a.sin[2 * pi + 1];
  }
  double ref[8];
  {
    const double* rp = a.refs + (size_t)a.ref_index[i] * C + sub * 8;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) ref[ch] = rp[ch];
  }
  OT* out = reinterpret_cast<OT*>(a.fout) + (size_t)(a.first_out + i) * Ho * Wo * a.CO;
  auto rsum = [](double v) { return LPO == 16 ? row16_sum(v) : row8_sum(v); };
  const int n_it = (Ho * Wo + G - 1) / G;                 // uniform trip count: the row reductions need whole groups
  for (int it = 0; it < n_it; ++it) {
    const int t = min(it * G + grp, Ho * Wo - 1);
    const int y = t / Wo, x = t - y * Wo;
    double f[8], fr[8], fc[8], frc[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) frc[ch] = 0.0;
    if (grad) interp8<ST, LPO, true, FS>(P, a.H, a.W, C, sub, x * inv_up, y * inv_up, l2_normalize != 0, f, fr, fc, cross ? frc : nullptr);
    else interp8<ST, LPO, false, FS>(P, a.H, a.W, C, sub, x * inv_up, y * inv_up, l2_normalize != 0, f, fr, fc);
    double s = 0.0, br = 0.0, bc = 0.0, rc = 0.0, xr = 0.0;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const double res = f[ch] - ref[ch];
      s = fma(res, res, s);
      if (grad) { br = fma(res, fr[ch], br); bc = fma(res, fc[ch], bc); rc = fma(fr[ch], fc[ch], rc); xr = fma(frc[ch], res, xr); }
    }
    s = rsum(s);
    if (grad) { br = rsum(br); bc = rsum(bc); if (cross) { rc = rsum(rc); xr = rsum(xr); } }
    if (sub == 0 && it * G + grp < Ho * Wo) {
      double rho[3];
      loss_eval(a.loss.type, a.loss.a, 1.0, s, rho);
      double cost = 0.5 * rho[0];
      OT* o = out + (size_t)t * a.CO;
      if (grad) {
        double dcr = 0.0, dcc = 0.0, dcrc = 0.0;
        if (cost > 1.0e-8) {
          dcr = rho[1] * br; dcc = rho[1] * bc;
          if (cross) dcrc = rho[2] * 2.0 * br * bc + rho[1] * (rc + xr);
          if (a.apply_sqrt) {
            cost = sqrt(cost);
            if (cross) { dcrc *= 0.5 / cost; dcrc += -0.5 * 0.5 / (cost * cost * cost) * dcr * dcc; }
            dcr *= 0.5 / cost; dcc *= 0.5 / cost;
          }
        }
        o[0] = store_cast<OT>(cost); o[1] = store_cast<OT>(dcr); o[2] = store_cast<OT>(dcc);
        if (cross) o[3] = store_cast<OT>(dcrc);
      } else {
        if (a.apply_sqrt) cost = sqrt(cost);
        o[0] = store_cast<OT>(cost);
      }
    }
  }
}

template <typename ST, int C>
static int launch_costmap_interp(pxr_ctx* ctx, int out_dtype, const CostmapArgs& a, int64_t n, int Ho, int Wo, double up,
                                 bool grad, bool cross, const pxr_interp_cfg* cfg) {
  const dim3 grid((unsigned)n), block(256);
#define CMI(OT, FS) hipLaunchKernelGGL((costmap_interp_kernel<ST, OT, C, FS>), grid, block, 0, ctx->stream, a, Ho, Wo, 1.0 / up, \
                                       grad ? 1 : 0, cross ? 1 : 0, cfg->l2_normalize)
  if (cfg->use_float_simd) {
    if (out_dtype == PXR_F16) CMI(_Float16, true); else if (out_dtype == PXR_F32) CMI(float, true); else CMI(double, true);
  } else {
    if (out_dtype == PXR_F16) CMI(_Float16, false); else if (out_dtype == PXR_F32) CMI(float, false); else CMI(double, false);
  }
#undef CMI
  return hip_check(hipGetLastError(), "costmap_interp_kernel launch");
}

template <typename OT>
static int launch_costmap_f16(pxr_ctx* ctx, const CostmapArgs& a, int64_t n, bool grad) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return set_error(PXR_EHIP, "hipGetDeviceProperties failed");
  const int64_t resident = (int64_t)prop.multiProcessorCount * (a.H == 16 ? 2 : 8);
  const dim3 grid((unsigned)(n < resident ? n : resident)), block((unsigned)(a.H * 16));
  if (a.H == 16) {
    if (grad) {   // the channel-split kernel: three workgroups per CU (168 VGPRs, 49 KB of LDS each)
      const int64_t res3 = (int64_t)prop.multiProcessorCount * 3;
      const dim3 g3((unsigned)(n < res3 ? n : res3));
      hipLaunchKernelGGL(costmap_meta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, n);
      hipLaunchKernelGGL((costmap_kernel_f16_split<OT>), g3, dim3(256), 0, ctx->stream, a, n);
    } else hipLaunchKernelGGL((costmap_kernel_f16<OT, 16, false>), grid, block, 0, ctx->stream, a, n);
  } else {
    if (grad) {   // 124 VGPRs: four workgroups per CU resident; twice that many in the grid evens out the tail
      const int64_t res8 = (int64_t)prop.multiProcessorCount * 8;
      const dim3 g8((unsigned)(n < res8 ? n : res8));
      hipLaunchKernelGGL(costmap_meta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, n);
      hipLaunchKernelGGL((costmap_kernel_f16_split8<OT>), g8, dim3(256), 0, ctx->stream, a, n);
    } else hipLaunchKernelGGL((costmap_kernel_f16<OT, 8, false>), grid, block, 0, ctx->stream, a, n);
  }
  return hip_check(hipGetLastError(), "costmap_kernel_f16 launch");
}

template <typename ST, typename OT, int C>
static int launch_costmap(pxr_ctx* ctx, const CostmapArgs& a, int64_t n, bool grad) {
  if constexpr (sizeof(ST) == 2 && C == 128) {
    if (a.H == a.W && (a.H == 16 || a.H == 8)) return launch_costmap_f16<OT>(ctx, a, n, grad);
  }
  const dim3 grid((unsigned)n), block((unsigned)(a.W * (C / 8)));
  if (grad) hipLaunchKernelGGL((costmap_kernel<ST, OT, C, true>), grid, block, 0, ctx->stream, a);
  else hipLaunchKernelGGL((costmap_kernel<ST, OT, C, false>), grid, block, 0, ctx->stream, a);
  return hip_check(hipGetLastError(), "costmap_kernel launch");
}

template <typename ST, typename OT, int C>
static int launch_costmap_small(pxr_ctx* ctx, const CostmapArgs& a, int64_t n, bool grad) {
  if (grad) hipLaunchKernelGGL((costmap_small_kernel<ST, OT, C, true>), dim3((unsigned)n), dim3(256), 0, ctx->stream, a);
  else hipLaunchKernelGGL((costmap_small_kernel<ST, OT, C, false>), dim3((unsigned)n), dim3(256), 0, ctx->stream, a);
  return hip_check(hipGetLastError(), "costmap_small_kernel launch");
}
template <typename ST, int C>
static int launch_costmap_small_o(pxr_ctx* ctx, int out_dtype, const CostmapArgs& a, int64_t n, bool grad) {
  switch (out_dtype) {
    case PXR_F16: return launch_costmap_small<ST, _Float16, C>(ctx, a, n, grad);
    case PXR_F32: return launch_costmap_small<ST, float, C>(ctx, a, n, grad);
    default: return launch_costmap_small<ST, double, C>(ctx, a, n, grad);
  }
}

template <typename ST, int C>
static int launch_costmap_o(pxr_ctx* ctx, int out_dtype, const CostmapArgs& a, int64_t n, bool grad) {
  switch (out_dtype) {
    case PXR_F16: return launch_costmap<ST, _Float16, C>(ctx, a, n, grad);
    case PXR_F32: return launch_costmap<ST, float, C>(ctx, a, n, grad);
    default: return launch_costmap<ST, double, C>(ctx, a, n, grad);
  }
}

}  // namespace pxr

extern "C" int pxr_costmap_extract(pxr_ctx* ctx, pxr_arena* features, pxr_arena* costmaps, int64_t first_out, int64_t n,
                                   const int64_t* d_patch, const int32_t* d_ref_index, const double* d_refs,
                                   const pxr_loss* loss, int as_gradientfield, int apply_sqrt) {
  PXR_REQUIRE(ctx && features && costmaps && loss, "pxr_costmap_extract: NULL argument");
  PXR_REQUIRE(n >= 0 && first_out >= 0 && first_out + n <= costmaps->n,
              "pxr_costmap_extract: cost maps [%lld, %lld) outside the arena of %lld", (long long)first_out,
              (long long)(first_out + n), (long long)costmaps->n);
  PXR_REQUIRE(costmaps->H == features->H && costmaps->W == features->W,
              "pxr_costmap_extract: cost maps are %dx%d, feature patches %dx%d (upsampling_factor != 1 is not supported)",
              costmaps->H, costmaps->W, features->H, features->W);
  PXR_REQUIRE(costmaps->C == (as_gradientfield ? 3 : 1),
              "pxr_costmap_extract: %d cost-map channels, CostMapConfig needs %d (compute_cross_derivative is not supported)",
              costmaps->C, as_gradientfield ? 3 : 1);
  const bool few = features->C == 3 || features->C == 1;
  if (features->C != 128 && features->C != 64 && !few)
    return pxr::set_error(PXR_EUNSUPPORTED, "pxr_costmap_extract: CHANNELS=%d not supported (128, 64, 3, 1)", features->C);
  if (!few && features->W * (features->C / 8) > 256)
    return pxr::set_error(PXR_EUNSUPPORTED, "pxr_costmap_extract: patches wider than %d texels are not supported "
                          "(dense maps: slice dense_cut_size windows with pxr_arena_extract first)", 256 / (features->C / 8));
  if (n == 0) return PXR_OK;
  PXR_REQUIRE(d_patch && d_ref_index && d_refs, "pxr_costmap_extract: NULL argument");
  PXR_HIP(hipSetDevice(ctx->device));
  pxr::CostmapArgs a;
  a.fin = features->d_data; a.cin = features->d_corners; a.sin = features->d_scales;
  a.fout = costmaps->d_data; a.cout = costmaps->d_corners; a.sout = costmaps->d_scales;
  a.H = features->H; a.W = features->W; a.CO = costmaps->C; a.apply_sqrt = apply_sqrt;
  a.first_out = first_out; a.patch = d_patch; a.ref_index = d_ref_index; a.refs = d_refs; a.loss = *loss;
  const bool grad = as_gradientfield != 0;
  const int od = costmaps->dtype;
  if (few) {
    if (features->dtype == PXR_F16) return features->C == 3 ? pxr::launch_costmap_small_o<_Float16, 3>(ctx, od, a, n, grad) : pxr::launch_costmap_small_o<_Float16, 1>(ctx, od, a, n, grad);
    if (features->dtype == PXR_F32) return features->C == 3 ? pxr::launch_costmap_small_o<float, 3>(ctx, od, a, n, grad) : pxr::launch_costmap_small_o<float, 1>(ctx, od, a, n, grad);
    return features->C == 3 ? pxr::launch_costmap_small_o<double, 3>(ctx, od, a, n, grad) : pxr::launch_costmap_small_o<double, 1>(ctx, od, a, n, grad);
  }
  if (features->dtype == PXR_F16 && features->C == 128) return pxr::launch_costmap_o<_Float16, 128>(ctx, od, a, n, grad);
  if (features->dtype == PXR_F16) return pxr::launch_costmap_o<_Float16, 64>(ctx, od, a, n, grad);
  if (features->dtype == PXR_F32 && features->C == 128) return pxr::launch_costmap_o<float, 128>(ctx, od, a, n, grad);
  if (features->dtype == PXR_F32) return pxr::launch_costmap_o<float, 64>(ctx, od, a, n, grad);
  if (features->C == 128) return pxr::launch_costmap_o<double, 128>(ctx, od, a, n, grad);
  return pxr::launch_costmap_o<double, 64>(ctx, od, a, n, grad);
}

extern "C" int pxr_costmap_extract_ex(pxr_ctx* ctx, pxr_arena* features, pxr_arena* costmaps, int64_t first_out, int64_t n,
                                      const int64_t* d_patch, const int32_t* d_ref_index, const double* d_refs,
                                      const pxr_loss* loss, int as_gradientfield, int apply_sqrt, const pxr_interp_cfg* cfg,
                                      double upsampling_factor, int compute_cross_derivative) {
  PXR_REQUIRE(ctx && features && costmaps && loss && cfg, "pxr_costmap_extract_ex: NULL argument");
  PXR_REQUIRE(upsampling_factor > 0.0, "pxr_costmap_extract_ex: upsampling_factor must be positive");
  PXR_REQUIRE(!compute_cross_derivative || as_gradientfield, "pxr_costmap_extract_ex: the cross derivative is part of the gradient field");
  const int Ho = (int)(features->H * (upsampling_factor + 1.0e-6)), Wo = (int)(features->W * (upsampling_factor + 1.0e-6));   // costmap_extractor.h:385-390
  const int CO = as_gradientfield ? (compute_cross_derivative ? 4 : 3) : 1;                                                    // GetEffectiveChannels, :52-61
  if (Ho == features->H && Wo == features->W && !compute_cross_derivative) {   // the no-interpolation branch (:253-279, :330-340)
    int rc = pxr_costmap_extract(ctx, features, costmaps, first_out, n, d_patch, d_ref_index, d_refs, loss, as_gradientfield, apply_sqrt);
    if (rc == PXR_OK) costmaps->up = upsampling_factor;
    return rc;
  }
  PXR_REQUIRE(n >= 0 && first_out >= 0 && first_out + n <= costmaps->n,
              "pxr_costmap_extract_ex: cost maps [%lld, %lld) outside the arena of %lld", (long long)first_out,
              (long long)(first_out + n), (long long)costmaps->n);
  PXR_REQUIRE(costmaps->H == Ho && costmaps->W == Wo && costmaps->C == CO,
              "pxr_costmap_extract_ex: the cost-map arena is %dx%dx%d, this configuration needs %dx%dx%d", costmaps->H,
              costmaps->W, costmaps->C, Ho, Wo, CO);
  PXR_REQUIRE(Ho >= 1 && Wo >= 1, "pxr_costmap_extract_ex: empty cost maps");
  if (features->C != 128 && features->C != 64)
    return pxr::set_error(PXR_EUNSUPPORTED, "pxr_costmap_extract_ex: CHANNELS=%d not supported by the interpolating branch (128, 64; "
                          "3 / 1 channels: only cost maps of the patch size without the cross derivative)", features->C);
  costmaps->up = upsampling_factor;                      // SetUpsamplingFactor, :399
  if (n == 0) return PXR_OK;
  PXR_REQUIRE(d_patch && d_ref_index && d_refs, "pxr_costmap_extract_ex: NULL argument");
  PXR_HIP(hipSetDevice(ctx->device));
  pxr::CostmapArgs a;
  a.fin = features->d_data; a.cin = features->d_corners; a.sin = features->d_scales;
  a.fout = costmaps->d_data; a.cout = costmaps->d_corners; a.sout = costmaps->d_scales;
  a.H = features->H; a.W = features->W; a.CO = CO; a.apply_sqrt = apply_sqrt;
  a.first_out = first_out; a.patch = d_patch; a.ref_index = d_ref_index; a.refs = d_refs; a.loss = *loss;
  const bool grad = as_gradientfield != 0, cross = compute_cross_derivative != 0;
  const int od = costmaps->dtype;
  const double up = upsampling_factor;
  if (features->dtype == PXR_F16 && features->C == 128) return pxr::launch_costmap_interp<_Float16, 128>(ctx, od, a, n, Ho, Wo, up, grad, cross, cfg);
  if (features->dtype == PXR_F16) return pxr::launch_costmap_interp<_Float16, 64>(ctx, od, a, n, Ho, Wo, up, grad, cross, cfg);
  if (features->dtype == PXR_F32 && features->C == 128) return pxr::launch_costmap_interp<float, 128>(ctx, od, a, n, Ho, Wo, up, grad, cross, cfg);
  if (features->dtype == PXR_F32) return pxr::launch_costmap_interp<float, 64>(ctx, od, a, n, Ho, Wo, up, grad, cross, cfg);
  if (features->C == 128) return pxr::launch_costmap_interp<double, 128>(ctx, od, a, n, Ho, Wo, up, grad, cross, cfg);
  return pxr::launch_costmap_interp<double, 64>(ctx, od, a, n, Ho, Wo, up, grad, cross, cfg);
}
