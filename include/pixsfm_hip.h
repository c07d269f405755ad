/*
 * pixsfm_hip.h -- C-ABI of the MI355X-native featuremetric refinement engine
 * (libpixsfm_hip.so).  Drop-in boundary for pixsfm's keypoint-adjustment (KA) and
 * bundle-adjustment (BA) hot path.
 *
 * The reference has no C plugin ABI: its seam is the pybind11 module pixsfm._pixsfm
 * (pixsfm/_pixsfm/bindings.cc:34-63) and, at the finest grain,
 * ceres::CostFunction::Evaluate(parameters, residuals, jacobians) [upstream Ceres]
 * called once per residual block.  This ABI is the batched, flat-array replacement of
 * that seam: plain pointers and sizes, no torch / pybind / Eigen types.  Each entry
 * point cites the reference interface it replaces.  INTEGRATION.md shows the binding a
 * pixsfm maintainer would add on the reference side.
 *
 * Conventions
 *  - every function returns 0 on success, a negative PXR_E* code on failure;
 *    pxr_last_error() returns a thread-local message (the reference throws C++
 *    exceptions mapped to Python, util/src/log_exceptions.h:52-116).
 *  - pointers prefixed d_ are DEVICE pointers (HBM), h_ are host pointers.
 *  - all work is enqueued on the context's HIP stream; functions that return host
 *    values synchronise that stream.
 *  - parameters are optimised IN PLACE in the caller's (device) arrays, like the
 *    reference does in caller memory (featuremetric_keypoint_optimizer.h:195-196,
 *    feature_reference_bundle_optimizer.h:111-114).
 *  - layouts: patches H x W x C channel-fastest (features/src/featurepatch.cc:160-163);
 *    qvec w-first (COLMAP); cam_params padded to PXR_KPAD doubles per camera.
 */
#ifndef PIXSFM_HIP_H_
#define PIXSFM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PXR_KPAD 12 /* camera parameter slots per camera (FULL_OPENCV has 12) */
#define PXR_OBS_REC 8 /* doubles per fused observation record */

enum { PXR_OK = 0, PXR_EINVAL = -1, PXR_EHIP = -2, PXR_ENOMEM = -3, PXR_EUNSUPPORTED = -4,
       PXR_ENUMERIC = -5 };
enum { PXR_F16 = 0, PXR_F32 = 1, PXR_F64 = 2 };
/* COLMAP 3.8 camera model ids (CAMERA_MODEL_SWITCH_CASES, residuals/src/feature_reference.h:232) */
enum { PXR_SIMPLE_PINHOLE = 0, PXR_PINHOLE = 1, PXR_SIMPLE_RADIAL = 2, PXR_RADIAL = 3,
       PXR_OPENCV = 4, PXR_OPENCV_FISHEYE = 5, PXR_FULL_OPENCV = 6, PXR_FOV = 7,
       PXR_SIMPLE_RADIAL_FISHEYE = 8, PXR_RADIAL_FISHEYE = 9, PXR_THIN_PRISM_FISHEYE = 10 };
/* ceres loss functions reachable through options.loss
 * (keypoint_adjustment_options.h:53, bundle_adjustment_options.h:49) */
enum { PXR_LOSS_TRIVIAL = 0, PXR_LOSS_CAUCHY = 1, PXR_LOSS_HUBER = 2, PXR_LOSS_SOFTL1 = 3 };

typedef struct pxr_ctx pxr_ctx;
typedef struct pxr_arena pxr_arena;

/* InterpolationConfig (base/src/interpolation.h:39-51, base/bindings.cc:134-154).
 * Only mode = BICUBIC with one node is on the hot path; other modes are rejected. */
typedef struct {
  int32_t l2_normalize;   /* default 1 */
  int32_t use_float_simd; /* default 0: fp32 horizontal pass, fp64 vertical pass.  1: all splines in fp32 (interpolation.h:177-218 with
                           * the float instantiation); pxr_ba_solve then evaluates from the texels in that arithmetic -- the
                           * Gram-matrix cache (exact fp64 algebra) is not used for such a solve */
  int32_t check_bounds;   /* default 0.  1: PatchInterpolator::Evaluate reports whether 0 < u < W, 0 < v < H
                           * (patch_interpolator.h:125-135,160-166).  Like in the reference this FAILS an evaluation
                           * only where the functor passes it on: the BA functor WITHOUT a reference descriptor, i.e.
                           * the cost-map BA (feature_reference.h:128-130) -- the block's squared norm is NaN, the
                           * solvers reject a non-finite trial cost and report PXR_TERM_FAILURE for a non-finite
                           * initial cost, like Ceres when Evaluate returns false.  With a reference descriptor
                           * (feature_reference.h:132-136) and in keypoint adjustment (featuremetric.h:61,
                           * feature_reference.h:59) the functors return true regardless: no effect. */
} pxr_interp_cfg;

typedef struct {
  int32_t type; /* PXR_LOSS_* */
  double a;     /* scale (Cauchy(0.25) is the reference default) */
} pxr_loss;

/* ---- context / errors ------------------------------------------------------------- */
int pxr_version(void);
const char* pxr_last_error(void);
/* device: HIP device ordinal; stream: a hipStream_t (e.g. torch's current stream
 * handle) or NULL for the device's default stream. */
int pxr_ctx_create(int device, void* stream, pxr_ctx** out);
int pxr_ctx_destroy(pxr_ctx* ctx);
int pxr_ctx_sync(pxr_ctx* ctx);
/* plain device-memory plumbing so that a host without torch can drive the library */
int pxr_malloc(pxr_ctx* ctx, size_t bytes, void** d_ptr);
int pxr_free(pxr_ctx* ctx, void* d_ptr);
int pxr_memcpy_h2d(pxr_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int pxr_memcpy_d2h(pxr_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int pxr_memset(pxr_ctx* ctx, void* d_dst, int value, size_t bytes);
/* HIP-event timer on the context's stream: start/stop bracket enqueued work;
 * pxr_timer_stop synchronises and returns elapsed milliseconds. */
int pxr_timer_start(pxr_ctx* ctx);
int pxr_timer_stop(pxr_ctx* ctx, double* ms);

/* ---- patch arena --------------------------------------------------------------------
 * Flat HBM-resident replacement of the FeaturePatch/FeatureMap/FeatureSet/FeatureView
 * container hierarchy (features/src/featurepatch.h:40-156, featureview.cc:44-55): n
 * patches of identical H x W x C and dtype, patch i at data + i*H*W*C, with per-patch
 * corner_ (x0,y0) and scale_ (sx,sy) (upsampling_factor_ is 1 for feature patches). */
int pxr_arena_create(pxr_ctx* ctx, int dtype, int C, int H, int W, int64_t n_patches,
                     void* d_data_or_null /* adopt caller's device buffer, non-owning */,
                     pxr_arena** out);
int pxr_arena_destroy(pxr_arena* a);
/* copy `count` patches (host, HWC) and their metadata; corners int32[count][2],
 * scales double[count][2].  h_patches may be NULL to only set metadata. */
int pxr_arena_upload(pxr_arena* a, int64_t first, int64_t count, const void* h_patches,
                     const int32_t* h_corners, const double* h_scales);
/* The same from `count` SEPARATE host patches (h_patch_ptrs[i]: H*W*C elements, HWC) -- the FeaturePatch objects a
 * FeatureView hands out one by one (features/src/featureview.cc:44-55, featurepatch.h:40-156): gathered by all host cores
 * into pinned staging buffers and uploaded double-buffered, without a stacked host copy of the set. */
int pxr_arena_upload_gather(pxr_arena* a, int64_t first, int64_t count, const void* const* h_patch_ptrs,
                            const int32_t* h_corners, const double* h_scales);
/* Sparse patch producer on the device: replaces FeatureExtractor.tensor_to_fmap's
 * sparse branch + extract_patches_torch/_numpy (pixsfm/features/extractor.py:152-199,
 * features/extract_patches.py:13-44) and the GPU -> CPU -> optimiser copies behind them.
 * d_fmap: ONE image's dense feature map on the device, torch layout [C][h][w], src_dtype PXR_F16 or
 * PXR_F32, C = the arena's channels (128 / 64: CNN features; 3 / 1: the `image` model's colour / grey values,
 * features/models/image.py).  d_keypoints [n][2]: COLMAP image coordinates.  scale =
 * (w / image_w, h / image_h); corner = clip((int)(kp * scale - 8), 0, (w, h) - 16 - 1) (C truncation,
 * like astype(np.int32)); l2_normalize: torch.nn.functional.normalize over channels (fp32,
 * eps 1e-12) before the cast to the arena dtype (extractor.py:173-175).  Fills patches
 * [first, first + n) of the arena (square patches of side ps <= 16; "8" / "16" above read ps / 2 and
 * ps) with their corners and scales; asynchronous on the context's stream. */
int pxr_arena_extract(pxr_ctx* ctx, pxr_arena* a, int64_t first, int64_t n, const void* d_fmap,
                      int src_dtype, int h, int w, const double* d_keypoints, double image_w,
                      double image_h, int l2_normalize);
void* pxr_arena_data(pxr_arena* a);     /* device pointer of patch 0 */
int32_t* pxr_arena_corners(pxr_arena* a); /* device int32[n][2] */
double* pxr_arena_scales(pxr_arena* a);   /* device double[n][2] */
int64_t pxr_arena_size(pxr_arena* a);

/* ---- BA evaluation -------------------------------------------------------------------
 * Batched replacement of FeatureReferenceCostFunctor / ...ConstantPoseCostFunctor
 * ::operator() under ceres::AutoDiffCostFunction
 * (residuals/src/feature_reference.h:71-148,157-207), one unit = one observation =
 * one residual block of C residuals.  All arrays are device pointers. */
typedef struct {
  int64_t n_obs;
  const int32_t* d_obs_image; /* [n_obs] index into qvec/tvec/image_camera */
  const int32_t* d_obs_point; /* [n_obs] index into xyz/refs */
  const int64_t* d_obs_patch; /* [n_obs] index into the arena */
  int32_t n_images;
  const int32_t* d_image_camera; /* [n_images] */
  const double* d_qvec;          /* [n_images][4] */
  const double* d_tvec;          /* [n_images][3] */
  int32_t n_cameras;
  const int32_t* d_cam_model;    /* [n_cameras] PXR_* model id */
  const double* d_cam_params;    /* [n_cameras][PXR_KPAD] */
  int64_t n_points;
  const double* d_xyz;           /* [n_points][3] */
  const double* d_refs;          /* [n_points][C]  Reference::descriptor (references.h:65); NULL = no reference
                                    is subtracted (cost-map BA, costmap_bundle_optimizer.h:104-119) */
} pxr_ba_view;

/* Fused evaluation.  Per observation i writes the record d_rec[i][0..7]:
 *   [0] s = r.r   [1] gx.gx  [2] gx.gy  [3] gy.gy  [4] gx.r  [5] gy.r  [6] x  [7] y
 * where r = f(x,y) - ref (C residuals), gx = dr/dx, gy = dr/dy (image coordinates) and
 * (x,y) = WorldToPixel(...) (base/src/projection.h:60-75).  The C x n Jacobian of the
 * block is J = [gx gy] * d(x,y)/d(params) (Jet bridge, base/src/interpolation.h:130-140),
 * so J^T J, J^T r and the robustifier's corrector are functions of this record and of
 * the 2 x n projection Jacobian only; the 128-row Jacobian is never materialised.
 * with_jacobian = 0 evaluates only s, x, y (trial-point cost evaluation).
 * Optional materialised outputs for parity checks (NULL to skip):
 *   d_r [n_obs][C], d_gx [n_obs][C], d_gy [n_obs][C]. */
int pxr_ba_eval(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view,
                const pxr_interp_cfg* cfg, int with_jacobian, double* d_rec, double* d_r,
                double* d_gx, double* d_gy);

/* Projection Jacobians for parity checks: d_P [n_obs][2][10 + PXR_KPAD], columns
 * q(4, ambient) t(3) X(3) k(PXR_KPAD); row 0 = dx/d., row 1 = dy/d. */
int pxr_ba_projection_jacobian(pxr_ctx* ctx, const pxr_ba_view* view, double* d_P);

/* Sum of 0.5 * rho(s_i) over the records (ceres cost), robustifier A20. */
int pxr_ba_cost(pxr_ctx* ctx, const double* d_rec, int64_t n_obs, const pxr_loss* loss,
                double* h_cost);


/* ---- BA solve -------------------------------------------------------------------------
 * Replaces BundleOptimizer::SolveProblem -> ceres::Solve
 * (bundle_adjustment/src/bundle_optimizer.h:172-245) for the feature-reference cost:
 * trust-region Levenberg-Marquardt [upstream Ceres 2.1 semantics: Jacobi scaling, LM
 * diagonal clamp, step acceptance, radius update, tolerances] with the point blocks
 * eliminated (Schur complement, what DENSE_SCHUR / SPARSE_SCHUR do, :181-191) and a dense
 * Cholesky of the reduced camera system on the GPU. */
enum { PXR_TERM_CONVERGENCE = 0, PXR_TERM_NO_CONVERGENCE = 1, PXR_TERM_FAILURE = 2 };

typedef struct {
  int32_t max_iterations;        /* 100 (bundle_adjustment_options.h:54)                */
  double function_tolerance;     /* 0                                                   */
  double gradient_tolerance;     /* 0                                                   */
  double parameter_tolerance;    /* 0                                                   */
  double initial_radius;         /* 1e4  [upstream ceres::Solver::Options defaults]     */
  double max_radius;             /* 1e16                                                */
  double min_radius;             /* 1e-32                                               */
  double min_relative_decrease;  /* 1e-3                                                */
  double min_lm_diagonal;        /* 1e-6                                                */
  double max_lm_diagonal;        /* 1e32                                                */
  int32_t max_consecutive_invalid_steps; /* 10 (bundle_adjustment_options.h:56)         */
  int32_t jacobi_scaling;        /* 1                                                   */
  int32_t use_inner_iterations;  /* BA: refine every variable point on its own after each trust-region step
                                    (bundle_adjustment/main.py:43 default True; [upstream Ceres]
                                    coordinate descent); ignored by pxr_ka_solve          */
  double inner_iteration_tolerance; /* 1e-3 [upstream]: disabled once the relative gain falls below */
  /* linear solver of the reduced camera system (BA only; bundle_optimizer.h:180-191 overrides the user's choice
   * by image count: <= 50 DENSE_SCHUR, <= 1000 SPARSE_SCHUR -- both = exact Schur complement + Cholesky, here the
   * dense GPU Cholesky -- and above ITERATIVE_SCHUR + SCHUR_JACOBI) */
  int32_t linear_solver;         /* PXR_LINEAR_AUTO: that rule on view->n_images; _DIRECT / _ITERATIVE force one   */
  int32_t max_linear_solver_iterations; /* 200 (bundle_adjustment_options.h:55)                                   */
  double eta;                    /* 0.1 [upstream Solver::Options::eta]: CG stops at i (Q_i - Q_{i-1}) / Q_i < eta,
                                    Q = x.Sx/2 - x.b (the only test the LM strategy leaves enabled [upstream])    */
  double linear_r_tolerance;     /* -1 = off like [upstream]; > 0: also stop at |r| <= tol |b| (parity tests)     */
} pxr_lm_options;
enum { PXR_LINEAR_AUTO = 0, PXR_LINEAR_DIRECT = 1, PXR_LINEAR_ITERATIVE = 2 };
#define PXR_MAX_IMAGES_DIRECT 1000 /* kMaxNumImagesDirectSparseSolver, bundle_optimizer.h:179 */

typedef struct {
  int32_t iterations;      /* LM iterations attempted                                   */
  int32_t num_successful;
  int32_t termination;     /* PXR_TERM_*                                                */
  int32_t num_camera_unknowns; /* size of the reduced camera system                     */
  int64_t num_point_unknowns;
  double initial_cost, final_cost, final_radius;
  double total_ms;         /* wall time of the LM loop (set-up excluded)                */
  double setup_ms;         /* host-side index construction + allocations                */
  int32_t linear_solver;   /* PXR_LINEAR_DIRECT / _ITERATIVE actually used (BA)         */
  int32_t collective_kib;   /* several ranks, direct solver: KiB moved per [S | rhs] all-reduce (packed upper triangle), else 0 */
  int64_t linear_iterations; /* pxr_ba_solve: conjugate-gradient iterations summed over the LM attempts (iterative solver);
                                pxr_ka_solve: node stencils (4 x 4 texels x C) interpolated over the solve -- its
                                algorithmic traffic is that count x 16 x C x sizeof(storage type)                 */
  int32_t accumulation;      /* how the sums over observations / points / ranks were formed in THIS solve: 1 = fixed-point integers
                                (the deterministic default: same bits for every run, launch shape and rank count), 2 = ordered
                                partial sums (the iterative solver's deterministic form: same bits per rank count), 0 = floating-
                                point atomics (pxr_set_deterministic(ctx, 0), or a direct solve WITHOUT Jacobi scaling, whose
                                columns no single grid can serve -- pxr_get_deterministic() alone does not tell)        */
  int32_t initial_us;        /* pxr_ba_solve: microseconds of total_ms spent BEFORE the first LM iteration -- the evaluation at the initial
                                point, the unscaled linearisation the Jacobi scaling is read from and the scaled one (device idle at
                                the end of it: the fixed-point check synchronises); 0 for pxr_ka_solve                    */
} pxr_lm_summary;

/* ceres::IterationCallback of the BA solve (the `solver.callbacks` of pixsfm's option dicts, base/src/callbacks.h; the
 * reference also keeps Ctrl-C responsive this way, util/src/py_interrupt.h): called on the host by pxr_ba_solve after the
 * initial evaluation (iteration 0) and after every LM iteration.  Return 0 to continue, 1 to abort (SOLVER_ABORT: the solve
 * ends with PXR_TERM_FAILURE, parameters as of the last accepted step), 2 to stop as converged (SOLVER_TERMINATE_SUCCESSFULLY).
 * With several ranks every rank calls its own callback and all must return the same value.  pxr_ka_solve runs its LM loops
 * inside one kernel and does not call back. */
typedef struct {
  int32_t iteration;            /* 0 = the initial evaluation                                            */
  int32_t step_is_valid;        /* the linear solve succeeded and the model cost decreased              */
  int32_t step_is_successful;   /* the step was accepted                                                */
  double cost;                  /* cost after this iteration (unchanged when the step was not accepted) */
  double cost_change;           /* cost before - candidate cost (0 for iteration 0 / invalid steps)     */
  double relative_decrease;     /* cost_change / model cost change                                      */
  double trust_region_radius;   /* radius the NEXT iteration starts with                                */
  double step_norm;
} pxr_iteration_summary;
typedef int (*pxr_iteration_callback)(const pxr_iteration_summary* summary, void* user);
int pxr_set_iteration_callback(pxr_ctx* ctx, pxr_iteration_callback fn /* NULL removes it */, void* user);

/* Deterministic mode: ON by default since round 5 (pxr_set_deterministic(ctx, 0) or PXR_DETERMINISTIC=0 in the environment of a
 * new context opt out).  Ceres sums each parameter block's Jacobian rows in a fixed order: the reference is run-to-run reproducible
 * for a fixed thread count (residuals/src/feature_reference.h:91-93 through AutoDiffCostFunction).  With the mode on, pxr_ba_solve
 * (direct solver) and pxr_ka_solve produce bit-identical results from run to run, for every launch shape AND for every number of
 * ranks: matrix / vector accumulations round each addend to a fixed-point grid and add 64-bit integers (associative; the grid is
 * derived from the measured diagonal of the previous linearisation and every linearisation is checked for overflow and repeated on
 * a coarser grid if need be), scalar sums are four 40-bit limbs per scalar, and several ranks all-reduce the integers (ncclInt64).
 * The iterative solver (> 1000 images) sums ORDERED PARTIALS instead -- every chunk of an image's observations leaves its part in a
 * buffer, the parts are added per image and per reduced-system column in a fixed order -- which gives the same bits on every run
 * for a given number of ranks (the ranks' parts are all-reduced as doubles).  The grids need the Jacobi scaling of the options (the
 * default); a direct solve without it falls back to floating-point atomics: fast, but the order of the additions -- hence the last
 * bits, hence now and then an accept / reject decision of the trust-region loop -- varies from run to run.  Costs ~3 % of an LM
 * iteration (1 % with the iterative solver), 8 % of a KA solve (bench.py reports both modes). */
int pxr_set_deterministic(pxr_ctx* ctx, int on);
int pxr_get_deterministic(pxr_ctx* ctx);

/* Gram-matrix cache of pxr_ba_solve: ON by default since round 5 (pxr_set_gram_cache(ctx, 0) or PXR_GRAM_CACHE=0 in the environment
 * of a new context switch it off).
 * The solver only consumes the 64-byte record of a residual block (pxr_ba_eval), and bicubic interpolation is linear in the
 * sixteen texels of the stencil: with the stencil's Gram matrix G = T T^t (16 x 16, over the channels) and D = T ref the
 * record is a set of quadratic / linear forms in the Catmull-Rom weights of the fractional position.  With the cache on,
 * pxr_ba_solve builds G and D once per observation (1 408 bytes each, HBM of the context, grow-only; rebuilt for the
 * observations whose projection moves to another texel) and evaluates every LM iteration from them instead of from the
 * 4 x 4 x C texels: ~3x less HBM traffic per iteration (the evaluation at the initial point of a solve still reads the
 * texels: the first step usually moves most projections to another texel).  The algebra on G is exact in double precision where the reference
 * interpolates with an fp32 horizontal pass (cubic_hermite_spline_simd.h), so a record differs from pxr_ba_eval's by that
 * pass's own rounding (1e-7 of the descriptor norm per channel): costs agree to ~1e-9 relative on sums over many blocks,
 * trajectories to the solver's tolerances; pxr_ba_eval itself, the other solvers and cost maps are unaffected.  Used for
 * feature patches of 128 / 64 channels in fp16 / fp32 storage with reference descriptors; otherwise the flag is ignored.
 * pxr_set_gram_cache(ctx, 0) also releases the storage.
 * Memory: 1 408 bytes per observation (+ 12 per observation of bookkeeping) in a grow-only buffer of the context -- 1.4 GB at 1M
 * observations -- allocated by the first solve that uses it and kept until the flag is cleared or the context is destroyed.  The
 * inner iterations (pxr_lm_options.use_inner_iterations) use the same buffer whenever the patches qualify, flag or no flag (set
 * PXR_INNER_NO_CACHE=1 to keep them from it); if it cannot be allocated they rebuild their matrices at every call instead. */
int pxr_set_gram_cache(pxr_ctx* ctx, int on);
int pxr_get_gram_cache(pxr_ctx* ctx);
/* The records of pxr_ba_eval(with_jacobian = 1) through that cache, outside a solve (parity tests, bench): reset != 0
 * (or a cache that does not fit the view yet) starts from an empty cache -- every observation's G is built -- otherwise the
 * matrices of the previous call are reused for the observations still in their cell.  The cache is keyed by observation
 * index: reuse it only with the same arena, observations and reference descriptors.  h_rebuilt (may be NULL; a non-NULL
 * pointer synchronises): how many observations' matrices this call built. */
int pxr_ba_eval_gram(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, const pxr_interp_cfg* cfg, int reset,
                     double* d_rec, int32_t* h_rebuilt);

/* ---- multi-GPU: one process per GPU ---------------------------------------------
 * With N ranks every rank holds ALL images and cameras (replicated) and a disjoint shard of the points with
 * all their observations, patches and references; the only exchange of the BA path is an in-place
 * all-reduce(sum) of doubles on the context's stream: [S | rhs] per linear solve (direct solver) or one
 * camera-sized vector per conjugate-gradient iteration (iterative solver), diag(U) | g_c per linearisation and
 * 16 scalars per LM attempt.  KA / reference extraction shard over independent sub-problems / points and
 * need no collective during the solve (base/src/parallel_optimizer.h:77-211 is the reference's shape).
 *
 * Native path: an RCCL communicator owned by the context.  Rank 0 calls pxr_comm_unique_id and hands the 128
 * bytes to the other ranks by any side channel (torch.distributed broadcast, a file, MPI, ...); then EVERY
 * rank calls pxr_comm_init (collective).  librccl is resolved at run time (the copy already loaded by the
 * process, e.g. PyTorch's, else /opt/rocm/lib/librccl.so.1), so the library has no link-time dependency. */
#define PXR_COMM_ID_BYTES 128
int pxr_comm_unique_id(void* h_id /* [PXR_COMM_ID_BYTES] */);
int pxr_comm_init(pxr_ctx* ctx, const void* h_id, int rank, int nranks);
int pxr_comm_destroy(pxr_ctx* ctx);
/* rank / size of the context: set by pxr_comm_init, or by pxr_comm_set_rank when the collective is a
 * caller-supplied callback (below); (0, 1) by default. */
int pxr_comm_set_rank(pxr_ctx* ctx, int rank, int nranks);
int pxr_comm_rank(pxr_ctx* ctx, int* rank, int* nranks);
/* in-place sum over the ranks of `count` doubles at device pointer d_buf, ordered with the context's stream
 * (ncclAllReduce on that stream); a no-op on a context without communicator / with one rank. */
int pxr_comm_allreduce_sum(pxr_ctx* ctx, double* d_buf, int64_t count);
/* Diagnostics.  pxr_comm_force(ctx, 1) (or PXR_FORCE_COLLECTIVE=1 in the environment when the context is created): a context
 * whose communicator has ONE rank still takes the multi-rank branch of the solvers -- pack, ncclAllReduce(ncclInt64 /
 * ncclFloat64) on the context's stream, unpack -- so that the native collective path can be exercised and timed on a single
 * GPU; results must be bit-identical to the plain one-rank solve (tests/test_zz_multi_rank_gpu.py).
 * pxr_comm_stats: ncclAllReduce calls / payload bytes issued through the context so far (NULL pointers are skipped). */
int pxr_comm_force(pxr_ctx* ctx, int on);
int pxr_comm_stats(pxr_ctx* ctx, int64_t* calls, int64_t* bytes, int reset);

/* Caller-supplied collective (same contract as pxr_comm_allreduce_sum), for hosts that bring their own
 * transport -- the CPU/gloo tests, MPI.  Passed to pxr_ba_solve it takes precedence over the context's
 * communicator; NULL = use the communicator (or run single-GPU). */
typedef int (*pxr_allreduce_fn)(void* user, double* d_buf, int64_t count);

/* Parameterisation (host arrays, bundle_optimizer.h:335-453):
 *   h_pose_const[img]      1 = qvec,tvec constant (HasConstantPose / !refine_extrinsics)
 *   h_tvec_const_mask[img] bit a = tvec[a] constant (SubsetManifold, ConstantTvec)
 *   h_cam_const_mask[cam]  bit a = camera parameter a constant (SubsetManifold / constant camera)
 *   h_point_const[pt]      1 = constant point
 * The qvec / tvec / cam_params / xyz device arrays of `view` are updated IN PLACE. */
int pxr_ba_solve(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, const pxr_interp_cfg* cfg,
                 const pxr_loss* loss, const uint8_t* h_pose_const, const uint8_t* h_tvec_const_mask,
                 const uint16_t* h_cam_const_mask, const uint8_t* h_point_const,
                 const pxr_lm_options* options, pxr_allreduce_fn allreduce, void* allreduce_user,
                 pxr_lm_summary* summary);


/* ---- BA reference extraction -----------------------------------------------------------
 * Replaces ReferenceExtractor::Run (bundle_adjustment/src/reference_extractor.h:125-318) +
 * RobustMeanIRLS (base/src/irls_optim.h:24-71) for N_NODES = 1: per point, descriptors of all
 * its observations at the CURRENT projection (view->d_refs is ignored), IRLS robust mean with
 * `loss` (ReferenceConfig.loss, Cauchy(0.25)) and `iters` (100) iterations, reference = the
 * observation descriptor closest to the robust mean.  Outputs (device): d_refs_out
 * [n_points][C], d_ref_obs_out [n_points] (index of the chosen observation, -1 if the point has
 * none), optional d_robust_mean_out [n_points][C].  Channels: 128, 64, and 3 / 1 (image intensities; the registered
 * FeatureReferenceBundleOptimizer cases (128,1) (64,1) (3,1) (1,1), feature_reference_bundle_optimizer.h:13-16 -- below 8
 * channels the interpolation is the scalar Ceres bicubic the reference falls back to, interpolation.h:222-268). */
int pxr_ba_compute_references(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view,
                              const pxr_interp_cfg* cfg, const pxr_loss* loss, int iters,
                              double* d_refs_out, int64_t* d_ref_obs_out, double* d_robust_mean_out,
                              double* d_obs_desc_out /* NULL, or [n_obs][C]: the per-observation descriptors
                                                        (ReferenceConfig::keep_observations, reference_extractor.h:60,259-265) */);

/* ---- cost maps (the reference's low-memory BA) -----------------------------
 * Replaces CostMapExtractor::RunSubset / FillPointCostmap (bundle_adjustment/src/costmap_extractor.h:177-358)
 * for CostMapConfig.upsampling_factor = 1, compute_cross_derivative = false on sparse patches: cost map
 * first_out + i of `costmaps` (C = 3 [cost, dcost/dr, dcost/dc] when as_gradientfield, else 1 [cost]; H, W as
 * the features; any dtype -- the reference's binding uses the features', bundle_adjustment/bindings.cc:21)
 * = featuremetric error of every texel of feature patch d_patch[i] against reference descriptor
 * d_refs[d_ref_index[i]] (the observation's 3D point), robustified with `loss` (CostMapConfig.loss), central
 * differences in the storage type, optional sqrt (CostMapConfig.apply_sqrt); corner and scale are copied from
 * the feature patch.  The cost-map BA (CostMapBundleOptimizer, costmap_bundle_optimizer.h) is then
 * pxr_ba_solve on the `costmaps` arena with view->d_refs = NULL and cfg->l2_normalize = 0
 * (bundle_adjustment/main.py:270): the residual block is the interpolated 3- (or 1-) channel texel. 
 * Feature channels: 128, 64, and 3 / 1 (image intensities; costmap_extractor.h:35-37 registers 128 and 3) -- one lane per
 * texel there.  pxr_costmap_extract_ex's interpolating branch (upsampling_factor != 1, cross derivative) takes 128 / 64 only. */
int pxr_costmap_extract(pxr_ctx* ctx, pxr_arena* features, pxr_arena* costmaps, int64_t first_out, int64_t n,
                        const int64_t* d_patch, const int32_t* d_ref_index, const double* d_refs,
                        const pxr_loss* loss, int as_gradientfield, int apply_sqrt);

/* The general form: CostMapConfig.upsampling_factor and compute_cross_derivative (costmap_extractor.h:42-45).  With
 * upsampling_factor = 1 and no cross derivative this IS pxr_costmap_extract.  Otherwise the reference interpolates
 * (FillPointCostmap :280-284, :341-345): output texel (y, x) = the features evaluated at the local patch coordinates
 * (x, y) / upsampling_factor by PatchInterpolator::EvaluateLocal under `cfg` (bicubic; L2 normalisation applies here),
 * 4 channels [cost, dcost/dr, dcost/dc, d2cost/drdc] with the cross derivative (:304-317).  The `costmaps` arena must be
 * int(H (f + 1e-6)) x int(W (f + 1e-6)) x {1, 3, 4} (CreateShallowCostmapFSet :385-390, GetEffectiveChannels :52-61) and
 * receives the upsampling factor (SetUpsamplingFactor :399; pxr_arena_set_upsampling), which the cost-map BA's coordinate
 * transform u = (x sx - 0.5 - x0) f then honours (featurepatch.h:250-255).  Note that the reference's own
 * CostMapBundleOptimizer takes 1 or 3 channels only (costmap_bundle_optimizer.h:9-14): 4-channel maps can be produced,
 * not optimised over -- here too (pxr_ba_eval rejects them). */
int pxr_costmap_extract_ex(pxr_ctx* ctx, pxr_arena* features, pxr_arena* costmaps, int64_t first_out, int64_t n,
                           const int64_t* d_patch, const int32_t* d_ref_index, const double* d_refs,
                           const pxr_loss* loss, int as_gradientfield, int apply_sqrt, const pxr_interp_cfg* cfg,
                           double upsampling_factor, int compute_cross_derivative);
/* FeaturePatch::SetUpsamplingFactor / UpsamplingFactor (features/src/featurepatch.h:208-216) for every patch of an arena;
 * 1 by default.  Only cost-map arenas (1 / 3 channels) may carry another value. */
int pxr_arena_set_upsampling(pxr_arena* a, double upsampling_factor);
double pxr_arena_upsampling(pxr_arena* a);

/* PatchInterpolator::Evaluate / InterpolateNodes, batched (A5, features/src/patch_interpolator.h:86-135;
 * `_features.PatchInterpolator(config).interpolate_nodes(fpatch, xy)`, features/bindings.cc): the normalised
 * bicubic descriptor of arena patch d_patch[i] at keypoint d_kp[i] (COLMAP image coordinates) into
 * d_desc [n][C], and optionally its Jacobian with respect to the keypoint, d_J [n][C][2] (d/dx, d/dy).
 * Channels 128, 64, 3, 1 (pxr_nearest_references below: the same). */
int pxr_interpolate(pxr_ctx* ctx, pxr_arena* arena, const pxr_interp_cfg* cfg, int64_t n, const double* d_kp,
                    const int64_t* d_patch, double* d_desc, double* d_J /* may be NULL */);

/* FindNearestReferences (localization/src/nearest_references.h:20-52): for each 2D-3D correspondence i the
 * query descriptor f(patch[d_patch[i]], d_kp[i]) is compared with its candidates -- rows
 * d_cand_index[d_cand_ptr[i] .. d_cand_ptr[i+1]) of d_cand_desc [*][C] (d_cand_index NULL: the rows
 * themselves), i.e. the observation descriptors of its 3D point (Reference::observations).  d_best[i] =
 * row of the nearest candidate (squared L2, first minimum wins; -1 without candidates), optional
 * d_best_dist[i], optional copy of the winner into d_out_desc [n][C]. */
int pxr_nearest_references(pxr_ctx* ctx, pxr_arena* arena, const pxr_interp_cfg* cfg, int64_t n,
                           const double* d_kp, const int64_t* d_patch, const int64_t* d_cand_ptr,
                           const int64_t* d_cand_index, const double* d_cand_desc, int64_t* d_best,
                           double* d_best_dist, double* d_out_desc);

/* ---- KA ---------------------------------------------------------------------------------
 * Batched replacement of FeatureMetricKeypointOptimizer::RunParallel / RunSubset
 * (keypoint_adjustment/src/featuremetric_keypoint_optimizer.h:69-137) for the flat edge list
 * the host builds from TopologicalKeypointOptimizer::SetUp + AddIntraResiduals
 * (topological_keypoint_optimizer.h:97-175, featuremetric_keypoint_optimizer.h:158-202):
 * one residual block per edge, r = f(patch[src], kp[src]) - f(patch[dst], kp[dst])
 * (FeatureMetric2DCostFunctor, residuals/src/featuremetric.h:44-63), ScaledLoss(loss, weight).
 * Sub-problems (ParallelOptimizer groups, base/src/parallel_optimizer.h:77-211) are given in
 * CSR form; each is solved by ONE workgroup that runs the whole bounded LM loop
 * (KeypointOptimizerBase::SolveProblem + ParameterizeKeypoints, keypoint_optimizer.h:77-157)
 * on the device.  All arrays are device pointers; d_kp is refined IN PLACE. */
typedef struct {
  int64_t n_nodes;
  double* d_kp;                   /* [n_nodes][2] keypoints in COLMAP image coordinates   */
  const int64_t* d_node_patch;    /* [n_nodes] patch index in the arena                   */
  const uint8_t* d_node_const;    /* [n_nodes] 1 = constant (roots, KeypointAdjustmentSetup); 2 = variable WITHOUT box
                                     bounds: a match destination outside RunSubset's nodes_in_problem, which
                                     ParameterizeKeypoints never visits (keypoint_optimizer.h:117); 0 = variable */
  int64_t n_edges;
  const int32_t* d_edge_src;      /* [n_edges]                                            */
  const int32_t* d_edge_dst;
  const double* d_edge_w;         /* ScaledLoss weight (similarity if weight_by_sim, else 1) */
  int32_t n_problems;
  const int64_t* d_prob_node_ptr; /* [n_problems + 1] into d_prob_nodes (ascending node ids) */
  const int32_t* d_prob_nodes;
  const int64_t* d_prob_edge_ptr; /* [n_problems + 1] into d_prob_edges                    */
  const int32_t* d_prob_edges;
  /* Unary reference terms (localization QKA, A8): one FeatureReference2DCostFunctor block
   * r = f(patch[node], kp[node]) - ref per entry (residuals/src/feature_reference.h:20-60,
   * localization/src/query_keypoint_optimizer.h:122-139); several entries may name the same
   * node (stacked correspondences, single_query_keypoint_optimizer.h:124-170).  n_unary = 0
   * (and NULL pointers) for plain KA. */
  int64_t n_unary;
  const int32_t* d_unary_node;    /* [n_unary]                                            */
  const double* d_unary_ref;      /* [n_unary][C] reference descriptors                   */
  const double* d_unary_w;        /* [n_unary] ScaledLoss weight; NULL = 1                */
  const int64_t* d_prob_unary_ptr;/* [n_problems + 1] into d_prob_unary                   */
  const int32_t* d_prob_unary;
  /* Optional (NULL: every sub-problem is a Ceres problem of its own, as above).  d_prob_group[i] = id of the label group --
   * the ceres::Problem of the reference, ONE trust region / line search / termination (keypoint_optimizer.h:77-104) -- that
   * sub-problem i is a CHUNK of.  The chunks of a group are consecutive sub-problems and share no variable (whole tracks
   * each); every chunk still gets its own workgroup, and the group's scalars -- cost, model cost change, probe costs, step
   * norms -- are summed over its workgroups, so that a group of 1000 keypoints (configs/low_memory.yaml:
   * max_kps_per_problem 1000) runs on 20 workgroups instead of one with the decisions of ONE problem.  Per-chunk summaries
   * then carry the group's iteration counts and the chunk's part of the costs.  A group may have at most as many chunks as
   * the launch keeps resident (~448 on an MI355X; larger groups: PXR_EUNSUPPORTED). */
  const int32_t* d_prob_group;    /* [n_problems] non-decreasing group ids, or NULL          */
} pxr_ka_view;

/* Per-edge evaluation for parity checks: d_cost [n_edges] = 0.5 w rho(|r|^2); optional
 * d_r [n_edges][C], d_J1 / d_J2 [n_edges][C][2] (dr/dkp_src, dr/dkp_dst, row-major). */
int pxr_ka_eval(pxr_ctx* ctx, pxr_arena* arena, const pxr_ka_view* view, const pxr_interp_cfg* cfg,
                const pxr_loss* loss, double* d_cost, double* d_r, double* d_J1, double* d_J2);

/* Solve every sub-problem.  bound: KeypointOptimizerOptions::bound (4.0,
 * keypoint_adjustment/main.py:78).  h_summaries: NULL or [n_problems] per-problem summaries;
 * *total accumulates them like AccumulateSummaries (util/src/statistics.h:131-160). */
int pxr_ka_solve(pxr_ctx* ctx, pxr_arena* arena, const pxr_ka_view* view, const pxr_interp_cfg* cfg,
                 const pxr_loss* loss, double bound, const pxr_lm_options* options,
                 pxr_lm_summary* h_summaries, pxr_lm_summary* total);

/* ---- match-graph labelling (host code) ----------------------------------
 * ComputeTrackLabels / ComputeScoreLabels / ComputeRootLabels (base/src/graph.cc:126-256), the
 * pre-processing KeypointAdjuster.refine runs before the optimisers (keypoint_adjustment/main.py:111-118),
 * over flat HOST arrays: node_image [n_nodes] (FeatureNode::image_id), edges in Graph order (for node in
 * nodes: for match in node.out_matches) as edge_src / edge_dst / edge_sim.  Sequential by construction
 * (every union depends on the earlier ones), native so that it does not become the bottleneck. */
int pxr_graph_track_labels(int64_t n_nodes, const int32_t* node_image, int64_t n_edges,
                           const int64_t* edge_src, const int64_t* edge_dst, const double* edge_sim,
                           int64_t* track_labels /* [n_nodes] */, int64_t* n_tracks_out /* may be NULL */);
int pxr_graph_score_labels(int64_t n_nodes, int64_t n_edges, const int64_t* edge_src,
                           const int64_t* edge_dst, const double* edge_sim, const int64_t* track_labels,
                           double* scores /* [n_nodes] */);
int pxr_graph_root_labels(int64_t n_nodes, const int64_t* track_labels, const double* scores,
                          uint8_t* is_root /* [n_nodes] */);
/* The same three labellings ON THE DEVICE for a flat graph that already lives in HBM (csrc/pxr_graph_gpu.hip): connected
 * components by label hooking, one wavefront per component replaying the reference's ordered, image-constrained
 * union-find, scores summed in the reference's order (bit-identical), roots by per-track maximum.  All arrays are device
 * pointers; matches in Graph order (ascending source node); d_scores / d_is_root may be NULL (roots need scores). */
int pxr_graph_labels_device(pxr_ctx* ctx, int64_t n_nodes, const int32_t* d_node_image, int64_t n_edges,
                            const int64_t* d_edge_src, const int64_t* d_edge_dst, const double* d_edge_sim,
                            int64_t* d_track_labels, double* d_scores, uint8_t* d_is_root, int64_t* h_n_tracks);
/* ---- BA problem construction (host code; A16, A17) ----------------------------------------------------------
 * BundleOptimizer::SetUp on a flat scene (bundle_adjustment/src/bundle_optimizer.h:139-165): AddImageToProblem for the
 * images of the setup (:247-275, incl. the min_track_length filter), AddPointToProblem for the extra variable /
 * constant points (:279-317: their observations in images OUTSIDE the setup, whose cameras become constant unless
 * residuals of setup images use them), FeatureReferenceBundleOptimizer::AddResiduals
 * (feature_reference_bundle_optimizer.h:90-149), then ParameterizePoints / Images / Cameras (:335-453).
 * Scene (HOST arrays, what a binding reads off colmap::Reconstruction): images 0 .. n_images-1 with their camera and
 * their points2D (rows p2d_ptr[i] .. p2d_ptr[i+1] of p2d_point3D; -1 = no 3D point), cameras (COLMAP model id), points
 * 0 .. n_points-1 with their tracks in Track().Elements() order (track_ptr / track_image / track_p2d);
 * has_patch [n_p2d] (NULL: every observation has a feature patch; a missing one is an error unless
 * skip_missing_patches -- the extractors' GetVisibleObservations, reference_extractor.h:171-213).
 * Setup (BundleAdjustmentSetup, bundle_adjustment_options.h:28-42): in_setup / const_pose / tvec_mask (bit a = tvec[a]
 * constant) per image, variable_point / constant_point per point, constant_camera per camera; the four refine_* flags
 * and min_track_length of BundleOptimizerOptions (:66-90).
 * Outputs: the residual blocks (capacity n_p2d) as (image, point2D index, point3D), ordered by point and inside a point
 * like its track; per image whether it takes part, whether its pose is constant, its constant translation components;
 * per camera -1 (not in the problem) or the bit mask of constant parameters (all bits: the block is constant); per
 * point -1 (no residual block), 0 variable, 1 constant.  With inner iterations every variable point is in group 0
 * (:350-355).  The linear solver follows the number of images IN THE SETUP (:180-191), see pxr_lm_options. */
int pxr_ba_build_problem(int32_t n_images, const int32_t* image_camera, const int64_t* p2d_ptr, const int64_t* p2d_point3D,
                         int32_t n_cameras, const int32_t* cam_model, int64_t n_points, const int64_t* track_ptr,
                         const int32_t* track_image, const int32_t* track_p2d, const uint8_t* has_patch,
                         const uint8_t* in_setup, const uint8_t* const_pose, const uint8_t* tvec_mask,
                         const uint8_t* variable_point, const uint8_t* constant_point, const uint8_t* constant_camera,
                         int refine_focal_length, int refine_principal_point, int refine_extra_params, int refine_extrinsics,
                         int min_track_length, int skip_missing_patches, int64_t* n_obs, int32_t* obs_image, int32_t* obs_p2d,
                         int64_t* obs_point, uint8_t* image_in_problem, uint8_t* pose_is_const, uint8_t* tvec_mask_out,
                         int32_t* camera_mask, int8_t* point_role);

/* Residual-block selection of TopologicalKeypointOptimizer::SetUp + AddIntraResiduals (A12,
 * topological_keypoint_optimizer.h:97-175, featuremetric_keypoint_optimizer.h:158-202): intra-track matches,
 * minus keypoint aliases, optionally root edges only, plus root-regularisation blocks; weights = similarity
 * (weight_by_sim) or 1 / root_regularize_weight.  HOST arrays; node_feature [n_nodes] = FeatureNode::feature_idx;
 * edges grouped by ascending source node (Graph order); nodes_in_problem NULL = every node;
 * out_src / out_dst / out_w have capacity 3 * n_edges, *n_out receives the number of blocks.  The output is what
 * pxr_ka_view's d_edge_src / d_edge_dst / d_edge_w take. */
int pxr_ka_build_edges(int64_t n_nodes, const int32_t* node_image, const int32_t* node_feature,
                       int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst, const double* edge_sim,
                       const int64_t* track_labels, const uint8_t* root_labels, const int64_t* nodes_in_problem,
                       int64_t n_in_problem, int weight_by_sim, int root_edges_only,
                       double root_regularize_weight, int64_t* out_src, int64_t* out_dst, double* out_w,
                       int64_t* n_out);

/* Dense SPD solve used for the reduced camera system (what Ceres' DENSE_SCHUR / SPARSE_SCHUR
 * Cholesky does on the CPU, bundle_optimizer.h:181-191): blocked right-looking Cholesky +
 * substitution in hand-written HIP.  d_a: n x n row-major, UPPER triangle filled (overwritten
 * by the factor); d_b: right-hand side -> solution.  *h_info = 0 or the 1-based index of the
 * first non-positive pivot. */
int pxr_dense_spd_solve(pxr_ctx* ctx, double* d_a, int n, double* d_b, int* h_info);

#ifdef __cplusplus
}
#endif
#endif /* PIXSFM_HIP_H_ */
