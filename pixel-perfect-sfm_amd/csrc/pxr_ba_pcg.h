// pxr_ba_pcg.h -- interface between the LM loop (pxr_ba_solve.hip) and the iterative Schur solver
// (pxr_ba_pcg.hip).  Not part of the C-ABI.
#pragma once
#include "pxr_ba_solve.h"

namespace pxr {

constexpr int PCG_GS = 18;   // rows of the largest preconditioner block: 6 pose + PXR_KPAD intrinsics columns

struct PcgArgs {
  hipStream_t st;
  SolveDev dv;                       // problem description at the linearisation point
  // structure (device)
  const ImgChunk* chunks; int n_chunks;   // chunks of the image-ordered observation slots
  const int4* so;                    // slot -> {observation, point, first partner, partners (0 = constant point)}
  const int64_t* pt_ptr; const int* part_obs; const int4* obs_cols;   // per point: observations and their column descriptors
  // current linearisation (device)
  const double* W; const double* T; const double* gp; const double* gc; const double* damp_c;
  const double* Ublk;                // [n_images][DC][DC] camera-side blocks (this rank's part)
  // preconditioner blocks (device): every column in exactly one block
  int n_groups; const int2* col_group; const int* group_size; const int* group_cols;
  // work space (device)
  double* u;                         // [n_points][3]
  double* Mloc;                      // [n_images][DC][DC]
  double* Gm;                        // [n_groups][PCG_GS][PCG_GS]
  double *x, *r, *p, *q, *z, *b;     // [n_c]
  double* cgs;                       // [8] scalars + [8] control block of the device-resident loop
  double* cg_part;                   // [8][ceil(n_c / 256)] per-block partial sums of the dot products
  int* d_fail;
};

struct PcgResult {
  int iterations;    // conjugate-gradient iterations performed
  bool ok;           // false: LINEAR_SOLVER_FAILURE (the LM loop treats the step as invalid)
  double x_dot_r;    // x . (b - S x) of the returned x (recurrence residual): corrects the model cost change of an inexact step
};

// Solves (U + D_c / radius - sum W T W^T) x = g_c - sum W T g_p approximately; x is left in a.x.
int pcg_solve(PcgArgs& a, double inv_radius, const pxr_lm_options* opt,
              const std::function<int(double*, int64_t)>& allreduce, PcgResult* res);
// diag(U) from the per-image blocks (atomic accumulation into a zeroed vector)
int pcg_diag_from_blocks(hipStream_t st, const SolveDev& d, const double* Ublk, double* diag);

}  // namespace pxr
