// pxr_chol_core.h -- the latency-oriented core of the dense Cholesky (csrc/pxr_chol.hip): ONE workgroup of eight
// wavefronts factors a 64 x 64 SPD tile and carries up to 48 further rows (panel rows below the tile, or identity rows
// that come out as inv(L)^T) through the same sweep, so that neither a triangular solve against an explicit inverse nor
// a second pass is on the critical path.
//
// The serial chain of a Cholesky is one reciprocal square root per pivot; a single wavefront issues one instruction
// every 4 cycles (8 for fp64 arithmetic), so the chain wavefront's INSTRUCTION COUNT is the second limit.  Hence:
//   * wavefront 0 holds the current 8-column panel of the tile with LANE = ROW (8 doubles per lane).  Pivot rows and
//     the scaled columns are fetched with v_readlane into scalar registers -- no LDS round trip, no barrier, no branch
//     inside a panel; two pivots per step from independent reciprocal square roots;
//   * the rank-8 update of the trailing columns is fp64 MFMA (v_mfma_f64_16x16x4, two per 16 x 16 block) on
//     accumulators that stay in registers for the whole tile: wavefront w < 4 owns block row w (blocks (w, 0..w), the
//     diagonal one in full so the pivot rows are symmetric-complete), the "row" wavefronts 5, 6, 7 own 16 extra rows x 64
//     columns each (wavefront 4 would share wavefront 0's SIMD: it only keeps the barriers company);
//   * per panel there are exactly two workgroup barriers: panel -> LDS (operand layout) -> MFMA of the ONE block column
//     the next panel lives in -> its 8 columns back to LDS -> wavefront 0.  All other block updates, the stores of L and
//     the whole of the extra rows' work run one panel behind, overlapped with wavefront 0's next pivot chain
//     (double-buffered panel).
// Accumulator layout (transposed product): lane (i = lane & 15, g = lane >> 4), register t of block (rb, cb) is the
// matrix element (row 16 rb + i, column 16 cb + g + 4 t) -- lanes run along rows, contiguous in memory.
#pragma once
#include <hip/hip_runtime.h>

namespace pxr {
namespace cholcore {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NB = 64;        // tile
constexpr int PW = 8;         // panel width
constexpr int ROW_WAVE0 = 5;  // first row wavefront
constexpr int ROW_WAVES = 3;  // 48 extra rows per workgroup

struct FactorLds {
  double P[2][PW][NB];            // panel p (buffer p & 1) in operand order: P[k][row] = L[row][8 p + k], zero above the diagonal
  double Dinv[2][NB];             // Dinv[buf][j] = 1 / L[8 p + j][8 p + j], j < 8 (the panel's diagonal factor itself is P[j][8 p + c])
  double Nx[PW][NB];              // the NEXT panel's columns as the accumulators hold them: Nx[c][row]
  double Tn[ROW_WAVES][PW][16];   // per row wavefront: its 16 rows x 8 panel columns on the way between accumulator and lane = row order
};

struct RowSink {                  // where a wavefront's results go (wave-uniform)
  enum { kNone = 0, kInverse = 1, kPanel = 2 };
  double* l_diag;     // wavefront 1 writes L (rows >= column, both < nb) here when non-null: l_diag[row + col * 64]
  double* rows_out;   // kInverse: rows_out[col * 64 + row] (row-major inv(L): the extra rows are identity rows);
  int mode;           // kPanel:   rows_out[row + col * ld] for row < rows_valid, col < nb
  int ld, rows_valid, nb;
#ifdef PXR_CHOL_PROFILE
  long long* prof;    // [wave][panel][4] clock64 stamps: phase-1 done, past barrier A, phase-2 done, past barrier B
#endif
};
#ifdef PXR_CHOL_PROFILE
#define CHOL_STAMP(k) do { if (lane == 0) sink.prof[(wave * 8 + p) * 4 + (k)] = clock64(); } while (0)
#else
#define CHOL_STAMP(k) do { } while (0)
#endif

// lanes of ONE wavefront exchanging data through LDS: the hardware runs a wavefront's LDS accesses in order, but the
// compiler reasons per thread (it forwarded a lane's own earlier store to a load of a slot another lane had rewritten) --
// a wavefront-scope fence makes the exchange visible to it; it costs no instruction
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// 1 / sqrt(x) for finite x > 0: v_rsq_f64 + one second-order correction (the device library's formula without its
// class test for 0 / inf, which the callers exclude: three instructions off the chain per pivot)
__device__ __forceinline__ double rsqrt_pos(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y0), y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

// blocks (w, 0..w) of the tile at `a` (column-major LOWER storage, leading dimension lda), mirrored into the upper triangle
// of the diagonal block, identity beyond nb
__device__ __forceinline__ void load_diag_blocks(d4 (&acc)[4], const double* __restrict__ a, int lda, int nb, int w, int lane) {
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    acc[cb] = d4{0.0, 0.0, 0.0, 0.0};
    if (cb > w) continue;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = 16 * w + i, col = 16 * cb + g + 4 * t;
      const int hi = row > col ? row : col, lo = row > col ? col : row;
      double v = (row == col) ? 1.0 : 0.0;
      if (hi < nb) v = a[(size_t)hi + (size_t)lo * lda];
      acc[cb][t] = v;
    }
  }
}

// 16 rows of the identity: global row index 16 vb + i
__device__ __forceinline__ void identity_rows(d4 (&acc)[4], int vb, int lane) {
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[cb][t] = (16 * cb + g + 4 * t == 16 * vb + i) ? 1.0 : 0.0;
}

// 16 rows x 64 columns of the matrix at `a` (element (0, 0) = first row, first column of the tile's block column):
// rows >= rows_valid and columns >= nb read as zero
__device__ __forceinline__ void load_rows(d4 (&acc)[4], const double* __restrict__ a, int lda, int nb, int rows_valid, int lane) {
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = 16 * cb + g + 4 * t;
      acc[cb][t] = (i < rows_valid && col < nb) ? a[(size_t)i + (size_t)col * lda] : 0.0;
    }
}

// columns of panel pn out of the accumulators of block column pn >> 1 into Nx (wavefronts w >= pn >> 1)
__device__ __forceinline__ void export_panel(FactorLds& lds, const d4 (&acc)[4], int w, int i, int g, int pn) {
  const int bcn = pn >> 1, hn = pn & 1;       // pn is a compile-time constant at every call site (the sweep is unrolled)
  if (w < bcn) return;
  lds.Nx[g][16 * w + i] = acc[bcn][2 * hn];
  lds.Nx[g + 4][16 * w + i] = acc[bcn][2 * hn + 1];
}

// blk(rows of this wavefront, block column cb) -= rows . P_cols^T for panel buffer `buf`; pr[] = this wavefront's row operand
__device__ __forceinline__ void mfma_block(const FactorLds& lds, d4& blk, int buf, int cb, int i, int g, const double (&pr)[2]) {
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const double pc = -lds.P[buf][4 * h2 + g][16 * cb + i];
    blk = __builtin_amdgcn_mfma_f64_16x16x4f64(pc, pr[h2], blk, 0, 0, 0);
  }
}

// wavefront 0: the 8 pivots of panel p, two per step.  Rows piv, piv + 1 of the diagonal block's first two live columns
// come in with v_readlane; both pivots are formed from INDEPENDENT reciprocal square roots (rsqrt(d00) and
// rsqrt(d00 d11 - d10^2)); every lane scales its own row, and the multipliers of the remaining columns (the scaled
// entries of rows piv + 2 .. base + 7, i.e. other lanes' results) are fetched with v_readlane again.
__device__ __forceinline__ int factor_panel(FactorLds& lds, int p, int lane, int bad) {
  const int buf = p & 1, base = PW * p;
  double a[PW], dinv = 0.0;
#pragma unroll
  for (int c = 0; c < PW; ++c) a[c] = lds.Nx[c][lane];
#pragma unroll
  for (int j = 0; j < PW; j += 2) {
    const int piv = base + j;
    const double d00 = readlane_f64(a[j], piv), d10 = readlane_f64(a[j + 1], piv), d11 = readlane_f64(a[j + 1], piv + 1);
    const double det = fma(d00, d11, -(d10 * d10));
    const bool ok0 = d00 > 0.0, ok1 = det > 0.0;
    bad = (!ok0 && bad == 0) ? piv + 1 : bad;
    bad = (!ok1 && bad == 0) ? piv + 2 : bad;
    const double r0 = rsqrt_pos(d00), rd = rsqrt_pos(det);
    const double inv0 = ok0 ? r0 : 1.0;
    const double inv1 = ok1 ? rd * (d00 * inv0) : 1.0;                      // 1 / sqrt(d11 - d10^2 / d00) = sqrt(d00) / sqrt(det)
    const double l10 = d10 * inv0;
    const double l0 = a[j] * inv0;
    const double l1 = fma(-l0, l10, a[j + 1]) * inv1;
#pragma unroll
    for (int c = j + 2; c < PW; ++c) {
      const double u = readlane_f64(l0, base + c);                          // L[base + c][piv]
      const double w = readlane_f64(l1, base + c);                          // L[base + c][piv + 1]
      a[c] = fma(-l1, w, fma(-l0, u, a[c]));
    }
    dinv = lane == j ? inv0 : dinv;                                         // lanes j, j + 1 keep the reciprocal diagonal for the extra rows
    dinv = lane == j + 1 ? inv1 : dinv;
    a[j] = lane >= piv ? l0 : 0.0;
    a[j + 1] = lane >= piv + 1 ? l1 : 0.0;
  }
  lds.Dinv[buf][lane] = dinv;                                               // one conflict-free store (lanes >= 8: unused slots)
#pragma unroll
  for (int k = 0; k < PW; ++k) lds.P[buf][k][lane] = a[k];
  return bad;
}

// wavefront 1, one panel behind: panel p of L out of the LDS copy
__device__ __forceinline__ void store_diag_panel(const FactorLds& lds, int p, int lane, const RowSink& sink) {
  if (!sink.l_diag || lane < PW * p || lane >= sink.nb) return;           // one exec-mask branch; rows above the panel's diagonal block hold zeros
  double* out = sink.l_diag + lane + (PW * p) * NB;
#pragma unroll
  for (int k = 0; k < PW; ++k)
    if (PW * p + k < sink.nb) out[k * NB] = lds.P[p & 1][k][lane];   // wave-uniform test; entries above the diagonal inside the
                                                                                  // 8 x 8 block are stored as the zeros P holds (upper part: never read)
}

// a row wavefront (16 extra rows): the panel-p part of its triangular solve, then its rank-8 updates
__device__ __forceinline__ void extra_rows(FactorLds& lds, d4 (&acc)[4], int v, int i, int g, int p, const RowSink& sink) {
  const int bc = p >> 1, h = p & 1, buf = p & 1, bcn = (p + 1) >> 1;
  lds.Tn[v][g][i] = acc[bc][2 * h];
  lds.Tn[v][g + 4][i] = acc[bc][2 * h + 1];
  wave_lds_fence();
  double a[PW], l[PW];
#pragma unroll
  for (int c = 0; c < PW; ++c) a[c] = lds.Tn[v][c][i];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    l[j] = a[j] * lds.Dinv[buf][j];
#pragma unroll
    for (int c = j + 1; c < PW; ++c) a[c] = fma(-l[j], lds.P[buf][j][PW * p + c], a[c]);    // L[8 p + c][8 p + j]: same address in every lane
  }
  // row operand of the MFMAs: lane (i, g) needs L[row i][4 h2 + g] -- back through the wavefront's own LDS patch (a select
  // chain over l[] makes the compiler index a scratch array)
  wave_lds_fence();
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < PW; ++j) lds.Tn[v][j][i] = l[j];
  }
  wave_lds_fence();
  double pr[2];
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) pr[h2] = lds.Tn[v][4 * h2 + g][i];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
    if (cb >= bcn) mfma_block(lds, acc[cb], buf, cb, i, g, pr);
  if (g == 0 && i < sink.rows_valid) {                                     // one exec-mask branch; the column test is wave-uniform
    const size_t stride = sink.mode == RowSink::kInverse ? NB : (size_t)sink.ld;
    const int ncol = sink.mode == RowSink::kInverse ? NB : sink.nb;
    double* out = sink.rows_out + (size_t)i + (size_t)(PW * p) * stride;
#pragma unroll
    for (int j = 0; j < PW; ++j)
      if (PW * p + j < ncol) out[(size_t)j * stride] = l[j];
  }
}

// The sweep.  Wavefronts 0..3: acc = blocks (w, 0..w) of the tile; wavefronts 5..7: acc = 16 extra rows x 64 columns
// (sink.mode kNone: an idle row wavefront).  Returns (wavefront 0) 0, or 1 + the first non-positive pivot.
__device__ __forceinline__ int factor_tile(FactorLds& lds, d4 (&acc)[4], int wave, int lane, const RowSink& sink) {
  const int i = lane & 15, g = lane >> 4;
  const bool row_wave = wave >= ROW_WAVE0 && sink.mode != RowSink::kNone;
  int bad = 0;
  if (wave < 4) export_panel(lds, acc, wave, i, g, 0);
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NB / PW; ++p) {
    if (wave == 0) {
      bad = factor_panel(lds, p, lane, bad);
    } else if (wave < 4) {
      if (p > 0) {                                   // the blocks right of the next panel's block column, one panel behind
        const int q = p - 1, qn = (q + 1) >> 1;
        double pr[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) pr[h2] = lds.P[q & 1][4 * h2 + g][16 * wave + i];
#pragma unroll
        for (int cb = 1; cb < 4; ++cb)
          if (cb > qn && cb <= wave) mfma_block(lds, acc[cb], q & 1, cb, i, g, pr);
        if (wave == 1) store_diag_panel(lds, q, lane, sink);
      }
    } else if (row_wave && p > 0) {
      extra_rows(lds, acc, wave - ROW_WAVE0, i, g, p - 1, sink);
    }
    CHOL_STAMP(0);
    __syncthreads();
    CHOL_STAMP(1);
    if (wave < 4 && p + 1 < NB / PW) {               // the one block the next panel needs, then hand its columns over
      const int bcn = (p + 1) >> 1;
      if (wave >= bcn) {
        double pr[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) pr[h2] = lds.P[p & 1][4 * h2 + g][16 * wave + i];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          if (cb == bcn) mfma_block(lds, acc[cb], p & 1, cb, i, g, pr);
      }
      export_panel(lds, acc, wave, i, g, p + 1);
    }
    CHOL_STAMP(2);
    __syncthreads();
    CHOL_STAMP(3);
  }
  if (wave == 1) store_diag_panel(lds, NB / PW - 1, lane, sink);
  if (row_wave) extra_rows(lds, acc, wave - ROW_WAVE0, i, g, NB / PW - 1, sink);
  return bad;
}

}  // namespace cholcore
}  // namespace pxr
