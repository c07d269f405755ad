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
  // deterministic mode (ordered partial sums instead of floating-point atomics; all null / 0 otherwise)
  bool det = false;
  const int* chunk_ptr = nullptr;    // [n_images + 1] first chunk of every image (chunks are image-major)
  const int* col_ent_ptr = nullptr;  // [n_c + 1] per reduced-system column: its (image, local column) entries, images ascending
  const int2* col_ent = nullptr;
  double* wpart = nullptr;           // [n_chunks][DC] a chunk's part of W u
  double* mpart = nullptr;           // [n_chunks][DC][DC] a chunk's part of an image's own block
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
// deterministic linearisation of the block form: k_img left every chunk's NE = dc (dc + 1) / 2 + dc sums in kpart[chunk][ne_max];
// the image's block of U and its gradient part are the ordered sums over its chunks, diag(U) and g_c the ordered sums over the
// columns' entries
int pcg_blocks_from_partials(hipStream_t st, const SolveDev& d, const int* kchunk_ptr, const double* kpart, int ne_max, double* Ublk,
                             double* gimg /* [n_images][DC] */, const int* col_ent_ptr, const int2* col_ent, double* diag, double* gc);

}  // namespace pxr
