"""Thin Python host layer over the C-ABI: context, device arrays, patch arena, BA problem.

Everything numerical happens in libpixsfm_hip.so on the GPU.  numpy is used only to stage
host arrays in and out.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import F16, F32, F64, KPAD, OBS_REC, BaView, InterpCfg, Loss, check

_NP2DT = {np.dtype(np.float16): F16, np.dtype(np.float32): F32, np.dtype(np.float64): F64}


class Context:
    """One HIP device + stream.  `stream` may be a raw hipStream_t handle (int), e.g.
    torch.cuda.current_stream().cuda_stream, so torch events see the kernels."""

    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.pxr_ctx_create(int(device), C.c_void_p(stream or 0), C.byref(h)), "pxr_ctx_create")
        self.handle = h
        self.device = device

    def sync(self):
        check(self.lib.pxr_ctx_sync(self.handle), "pxr_ctx_sync")

    def timer_start(self):
        check(self.lib.pxr_timer_start(self.handle), "pxr_timer_start")

    def timer_stop(self):
        ms = C.c_double()
        check(self.lib.pxr_timer_stop(self.handle, C.byref(ms)), "pxr_timer_stop")
        return ms.value

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pxr_ctx_destroy(self.handle)
            self.handle = None

    # -- multi-GPU ----------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from rank 0 (ncclGetUniqueId) that every rank passes to comm_init."""
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        check(_lib.load().pxr_comm_unique_id(buf), "pxr_comm_unique_id")
        return bytes(buf.raw)

    def comm_init(self, unique_id, rank, nranks):
        """Collective: creates this context's RCCL communicator (ncclCommInitRank on its device)."""
        assert len(unique_id) == _lib.COMM_ID_BYTES
        check(self.lib.pxr_comm_init(self.handle, C.c_char_p(bytes(unique_id)), int(rank), int(nranks)), "pxr_comm_init")

    def comm_set_rank(self, rank, nranks):
        """Rank / size when the collective is a caller-supplied callback (gloo tests, MPI)."""
        check(self.lib.pxr_comm_set_rank(self.handle, int(rank), int(nranks)), "pxr_comm_set_rank")

    def comm_rank(self):
        r, n = C.c_int(), C.c_int()
        check(self.lib.pxr_comm_rank(self.handle, C.byref(r), C.byref(n)), "pxr_comm_rank")
        return r.value, n.value

    @property
    def deterministic(self):
        """Order-independent accumulation in pxr_ba_solve / pxr_ka_solve: bit-identical results from run to run (direct BA solver
        and KA: for every number of ranks too).  ON by default since round 5 (pxr_set_deterministic; PXR_DETERMINISTIC=0 switches it
        off for every new context)."""
        return bool(self.lib.pxr_get_deterministic(self.handle))

    @deterministic.setter
    def deterministic(self, on):
        check(self.lib.pxr_set_deterministic(self.handle, int(bool(on))), "pxr_set_deterministic")

    @property
    def gram_cache(self):
        """pxr_ba_solve evaluates the residual blocks' records from cached Gram matrices of the 4 x 4 stencils instead of from
        the texels: ~3x less HBM traffic per LM iteration, records differ from pxr_ba_eval's by the rounding of the reference's
        fp32 horizontal pass.  ON by default since round 5 (pxr_set_gram_cache; PXR_GRAM_CACHE=0 switches it off for every new
        context)."""
        return bool(self.lib.pxr_get_gram_cache(self.handle))

    @gram_cache.setter
    def gram_cache(self, on):
        check(self.lib.pxr_set_gram_cache(self.handle, int(bool(on))), "pxr_set_gram_cache")

    def set_iteration_callbacks(self, callbacks):
        """ceres::IterationCallback objects for the BA solves of this context (pxr_set_iteration_callback): each callable gets
        the iteration summary (attributes iteration, step_is_valid, step_is_successful, cost, cost_change, relative_decrease,
        trust_region_radius, step_norm) and returns None / 0 to continue, 1 to abort, 2 to terminate successfully; the largest
        answer wins.  An empty list removes the hook -- and re-raises an exception a callback raised during the solve (which
        was aborted at that iteration)."""
        callbacks = list(callbacks or [])
        if not callbacks:
            check(self.lib.pxr_set_iteration_callback(self.handle, None, None), "pxr_set_iteration_callback")
            self._iter_cb = None
            pending, self._iter_exc = getattr(self, "_iter_exc", None), None
            if pending is not None:          # an exception raised inside a callback aborted the solve: it surfaces here
                raise pending
            return
        self._iter_exc = None

        def hook(summary, _user):
            rc = 0
            try:
                # a copy: the solver's summary lives on its stack, a callback may keep what it is given
                snap = type(summary.contents).from_buffer_copy(summary.contents)
                for cb in callbacks:
                    ans = cb(snap)
                    rc = max(rc, int(getattr(ans, "value", ans) or 0))
            except BaseException as e:       # noqa: BLE001 -- ctypes would print and swallow it; abort the solve instead
                self._iter_exc = e
                return 1
            return rc
        self._iter_cb = _lib.ITERATION_CALLBACK(hook)            # kept alive as long as it is installed
        check(self.lib.pxr_set_iteration_callback(self.handle, C.cast(self._iter_cb, C.c_void_p), None), "pxr_set_iteration_callback")

    def comm_force(self, on=True):
        """Diagnostics: with a ONE-rank communicator the solvers still take their multi-rank branch (pack, ncclAllReduce on the
        context's stream, unpack); results must equal the plain solve bit for bit (pxr_comm_force)."""
        check(self.lib.pxr_comm_force(self.handle, int(bool(on))), "pxr_comm_force")

    def comm_stats(self, reset=False):
        """(ncclAllReduce calls, payload bytes) issued through this context so far."""
        calls, nbytes = C.c_int64(), C.c_int64()
        check(self.lib.pxr_comm_stats(self.handle, C.byref(calls), C.byref(nbytes), int(bool(reset))), "pxr_comm_stats")
        return calls.value, nbytes.value

    def comm_destroy(self):
        check(self.lib.pxr_comm_destroy(self.handle), "pxr_comm_destroy")

    def allreduce_sum(self, array):
        """In-place sum over the ranks of a float64 DeviceArray through the context's communicator."""
        assert array.dtype == np.float64
        check(self.lib.pxr_comm_allreduce_sum(self.handle, array.ptr, array.nbytes // 8), "pxr_comm_allreduce_sum")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- device arrays ------------------------------------------------------------------
    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype):
        a = DeviceArray(self, shape, dtype)
        check(self.lib.pxr_memset(self.handle, a.ptr, 0, a.nbytes), "pxr_memset")
        return a

    def to_device(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype)
        a = DeviceArray(self, host.shape, host.dtype)
        a.upload(host)
        return a


class DeviceArray:
    """A typed HBM buffer owned through pxr_malloc/pxr_free."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if hasattr(shape, "__len__") else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        check(ctx.lib.pxr_malloc(ctx.handle, self.nbytes, C.byref(p)), "pxr_malloc")
        self.ptr = p

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.nbytes == self.nbytes, (host.shape, self.shape)
        check(self.ctx.lib.pxr_memcpy_h2d(self.ctx.handle, self.ptr, host.ctypes.data, self.nbytes), "pxr_memcpy_h2d")

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(self.ctx.lib.pxr_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes), "pxr_memcpy_d2h")
        return out

    def free(self):
        if getattr(self, "ptr", None) and self.ctx.handle:
            self.ctx.lib.pxr_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PatchArena:
    """HBM-resident patches (n, H, W, C) channel-fastest + per-patch corner / scale.

    Replaces the FeaturePatch / FeatureMap / FeatureSet / FeatureView containers
    (pixsfm/features/src/featurepatch.h:40-156, featureview.cc:44-55) on the device.
    """

    def __init__(self, ctx, n, H, W, channels, dtype=np.float16, device_ptr=None):
        self.ctx = ctx
        self.n, self.H, self.W, self.C = int(n), int(H), int(W), int(channels)
        self.dtype = np.dtype(dtype)
        h = C.c_void_p()
        check(ctx.lib.pxr_arena_create(ctx.handle, _NP2DT[self.dtype], self.C, self.H, self.W, self.n,
                                       C.c_void_p(device_ptr or 0), C.byref(h)), "pxr_arena_create")
        self.handle = h

    @classmethod
    def from_numpy(cls, ctx, patches, corners, scales=None):
        patches = np.ascontiguousarray(patches)
        n, H, W, ch = patches.shape
        a = cls(ctx, n, H, W, ch, patches.dtype)
        a.upload(0, patches, corners, scales)
        return a

    def upload(self, first, patches=None, corners=None, scales=None):
        count = len(patches) if patches is not None else len(corners)
        if patches is not None:
            patches = np.ascontiguousarray(patches, dtype=self.dtype)
            assert patches.shape[1:] == (self.H, self.W, self.C)
        if corners is not None:
            corners = np.ascontiguousarray(corners, dtype=np.int32).reshape(count, 2)
        if scales is None and corners is not None:
            scales = np.ones((count, 2))
        if scales is not None:
            scales = np.ascontiguousarray(scales, dtype=np.float64).reshape(count, 2)
        check(self.ctx.lib.pxr_arena_upload(
            self.handle, int(first), int(count),
            None if patches is None else patches.ctypes.data,
            None if corners is None else corners.ctypes.data,
            None if scales is None else scales.ctypes.data), "pxr_arena_upload")

    @classmethod
    def from_patch_pointers(cls, ctx, pointers, shape, dtype, corners, scales):
        """An arena filled from SEPARATE host patches (pointers[i]: address of patch i, `shape` = (H, W, C) elements of
        `dtype`): pxr_arena_upload_gather -- all host cores gather into pinned staging buffers, the upload is
        double-buffered; no stacked host copy of the whole set.  The caller keeps the patches alive during the call."""
        pointers = np.ascontiguousarray(pointers, dtype=np.uint64)
        n = len(pointers)
        a = cls(ctx, n, shape[0], shape[1], shape[2], dtype)
        corners = np.ascontiguousarray(corners, dtype=np.int32).reshape(n, 2)
        scales = np.ascontiguousarray(scales, dtype=np.float64).reshape(n, 2)
        check(ctx.lib.pxr_arena_upload_gather(a.handle, 0, n, pointers.ctypes.data, corners.ctypes.data, scales.ctypes.data),
              "pxr_arena_upload_gather")
        return a

    def extract(self, first, fmap, keypoints, image_size, l2_normalize=True):
        """Fill patches [first, first + len(keypoints)) from ONE image's dense feature map that is
        already on the device (FeatureExtractor.tensor_to_fmap sparse branch, extractor.py:152-199,
        without the GPU -> CPU -> GPU round trip).

        fmap: a CUDA/ROCm tensor-like exposing __cuda_array_interface__ (e.g. a contiguous
        torch.cuda tensor) of shape (C, h, w) or (1, C, h, w), float16 or float32.
        keypoints: (n, 2) COLMAP image coordinates (host array or DeviceArray).
        image_size: (width, height) of the image the keypoints live in."""
        cai = fmap.__cuda_array_interface__
        shape = tuple(cai["shape"])
        if len(shape) == 4 and shape[0] == 1:
            shape = shape[1:]
        if len(shape) != 3 or shape[0] != self.C:
            raise ValueError("feature map must be (C=%d, h, w); got %r" % (self.C, tuple(cai["shape"])))
        if cai.get("strides") is not None:
            item = int(cai["typestr"][2:])
            full = tuple(cai["shape"])
            expect = tuple(int(np.prod(full[i + 1:])) * item for i in range(len(full)))
            if tuple(cai["strides"]) != expect:
                raise ValueError("feature map must be contiguous (call .contiguous())")
        src = {"<f2": _lib.F16, "<f4": _lib.F32}.get(cai["typestr"])
        if src is None:
            raise ValueError("feature map dtype %s not supported (float16 / float32)" % cai["typestr"])
        if isinstance(keypoints, DeviceArray):
            d_kp, n = keypoints, keypoints.shape[0]
        else:
            kp = np.ascontiguousarray(keypoints, dtype=np.float64).reshape(-1, 2)
            d_kp, n = self.ctx.to_device(kp, np.float64), len(kp)
        check(self.ctx.lib.pxr_arena_extract(self.ctx.handle, self.handle, int(first), int(n), C.c_void_p(cai["data"][0]),
                                             src, int(shape[1]), int(shape[2]), d_kp.ptr, float(image_size[0]),
                                             float(image_size[1]), int(bool(l2_normalize))), "pxr_arena_extract")
        self.ctx.sync()          # the temporary keypoint upload may be released after this
        return n

    def download(self, first=0, count=None):
        """(patches, corners, scales) of a range of the arena as numpy arrays (tests / debugging)."""
        count = self.n - first if count is None else count
        patches = np.empty((count, self.H, self.W, self.C), dtype=self.dtype)
        corners = np.empty((count, 2), dtype=np.int32)
        scales = np.empty((count, 2), dtype=np.float64)
        lib, h = self.ctx.lib, self.ctx.handle
        pb = patches[0].nbytes if count else 0
        check(lib.pxr_memcpy_d2h(h, patches.ctypes.data, C.c_void_p(self.data_ptr + pb * first), patches.nbytes), "d2h")
        check(lib.pxr_memcpy_d2h(h, corners.ctypes.data, C.c_void_p(lib.pxr_arena_corners(self.handle) + 8 * first),
                                 corners.nbytes), "d2h")
        check(lib.pxr_memcpy_d2h(h, scales.ctypes.data, C.c_void_p(lib.pxr_arena_scales(self.handle) + 16 * first),
                                 scales.nbytes), "d2h")
        return patches, corners, scales

    @property
    def data_ptr(self):
        return self.ctx.lib.pxr_arena_data(self.handle)

    @property
    def upsampling_factor(self):
        """FeaturePatch::UpsamplingFactor of every patch of the arena (cost maps; 1 for feature patches)."""
        return float(self.ctx.lib.pxr_arena_upsampling(self.handle))

    @upsampling_factor.setter
    def upsampling_factor(self, value):
        check(self.ctx.lib.pxr_arena_set_upsampling(self.handle, float(value)), "pxr_arena_set_upsampling")

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx.lib.pxr_arena_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def interpolate(ctx, arena, cfg, keypoints, patch_idx, jacobian=False):
    """PatchInterpolator::Evaluate, batched (pxr_interpolate): descriptors (n, C) of arena patches patch_idx at
    the keypoints (COLMAP image coordinates) and, with jacobian=True, their (n, C, 2) derivatives d/d(x, y)."""
    kp = ctx.to_device(np.ascontiguousarray(keypoints, dtype=np.float64).reshape(-1, 2), np.float64)
    n = kp.shape[0]
    pidx = ctx.to_device(np.ascontiguousarray(patch_idx, dtype=np.int64), np.int64)
    desc = ctx.empty((n, arena.C), np.float64)
    J = ctx.empty((n, arena.C, 2), np.float64) if jacobian else None
    check(ctx.lib.pxr_interpolate(ctx.handle, arena.handle, C.byref(cfg), n, kp.ptr, pidx.ptr, desc.ptr,
                                  J.ptr if J else None), "pxr_interpolate")
    return desc.download(), (J.download() if J else None)


def nearest_references(ctx, arena, cfg, keypoints, patch_idx, cand_ptr, cand_desc, cand_index=None, want_desc=False):
    """FindNearestReferences (localization/src/nearest_references.h:20-52) on the device.
    keypoints (n, 2), patch_idx (n,) arena patches of the query keypoints; candidates of correspondence i:
    rows cand_index[cand_ptr[i]:cand_ptr[i+1]] of cand_desc (a DeviceArray or host array (m, C); cand_index None:
    the rows themselves).  Returns (best row per correspondence, squared distances, winners (n, C) | None)."""
    kp = ctx.to_device(np.ascontiguousarray(keypoints, dtype=np.float64).reshape(-1, 2), np.float64)
    n = kp.shape[0]
    pidx = ctx.to_device(np.ascontiguousarray(patch_idx, dtype=np.int64), np.int64)
    cptr = ctx.to_device(np.ascontiguousarray(cand_ptr, dtype=np.int64), np.int64)
    cidx = None if cand_index is None else ctx.to_device(np.ascontiguousarray(cand_index, dtype=np.int64), np.int64)
    cdesc = cand_desc if isinstance(cand_desc, DeviceArray) else ctx.to_device(np.ascontiguousarray(cand_desc, dtype=np.float64), np.float64)
    best = ctx.empty((n,), np.int64)
    dist = ctx.empty((n,), np.float64)
    out = ctx.empty((n, arena.C), np.float64) if want_desc else None
    check(ctx.lib.pxr_nearest_references(ctx.handle, arena.handle, C.byref(cfg), n, kp.ptr, pidx.ptr, cptr.ptr,
                                         cidx.ptr if cidx else None, cdesc.ptr, best.ptr, dist.ptr,
                                         out.ptr if out else None), "pxr_nearest_references")
    return best.download(), dist.download(), (out.download() if out else None)


def interp_cfg(l2_normalize=True, use_float_simd=False, check_bounds=False, mode="BICUBIC", nodes=None,
               ncc_normalize=False, **_):
    """InterpolationConfig (pixsfm/base/main.py:1-7).  Only the hot-path configuration is accepted."""
    if str(mode).upper() != "BICUBIC":
        raise ValueError("InterpolatorType %r is outside the accelerated path (BICUBIC only)" % (mode,))
    if nodes is not None and [list(map(float, n)) for n in nodes] != [[0.0, 0.0]]:
        raise ValueError("only a single interpolation node [[0, 0]] is supported (N_NODES = 1)")
    if ncc_normalize:
        raise ValueError("ncc_normalize needs N_NODES > 1, which is outside the accelerated path")
    return InterpCfg(int(l2_normalize), int(use_float_simd), int(check_bounds))


def make_loss(name="cauchy", params=(0.25,)):
    name = str(name).lower()
    if name not in _lib.LOSS_IDS:
        raise ValueError("unsupported loss %r (have %s)" % (name, sorted(_lib.LOSS_IDS)))
    a = float(params[0]) if (params is not None and len(params)) else 1.0
    return Loss(_lib.LOSS_IDS[name], a)


def lm_options(max_iterations=100, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0,
               initial_radius=1e4, max_radius=1e16, min_radius=1e-32, min_relative_decrease=1e-3,
               min_lm_diagonal=1e-6, max_lm_diagonal=1e32, max_consecutive_invalid_steps=10, jacobi_scaling=1,
               use_inner_iterations=False, inner_iteration_tolerance=1e-3, linear_solver="auto",
               max_linear_solver_iterations=200, eta=0.1, linear_r_tolerance=-1.0, **ignored):
    """Subset of ceres::Solver::Options the engine honours (pixsfm/base/main.py:9-22 +
    bundle_adjustment_options.h:48-64); unknown keys are ignored like pyceres ignores nothing --
    callers should pass only what they need."""
    return _lib.LMOptions(int(max_iterations), float(function_tolerance), float(gradient_tolerance),
                          float(parameter_tolerance), float(initial_radius), float(max_radius), float(min_radius),
                          float(min_relative_decrease), float(min_lm_diagonal), float(max_lm_diagonal),
                          int(max_consecutive_invalid_steps), int(jacobi_scaling), int(bool(use_inner_iterations)),
                          float(inner_iteration_tolerance), _linear_solver_id(linear_solver),
                          int(max_linear_solver_iterations), float(eta), float(linear_r_tolerance))


def _linear_solver_id(name):
    """"auto": by image count like bundle_optimizer.h:180-191 (<= 1000 images: Schur complement + Cholesky;
    more: ITERATIVE_SCHUR).  Ceres' names are accepted: DENSE_SCHUR / SPARSE_SCHUR -> direct, ITERATIVE_SCHUR."""
    if isinstance(name, int):
        return int(name)
    key = str(name).lower()
    table = {"auto": _lib.LINEAR_AUTO, "direct": _lib.LINEAR_DIRECT, "dense_schur": _lib.LINEAR_DIRECT,
             "sparse_schur": _lib.LINEAR_DIRECT, "iterative": _lib.LINEAR_ITERATIVE,
             "iterative_schur": _lib.LINEAR_ITERATIVE}
    if key not in table:
        raise ValueError("unknown linear solver %r (auto, direct, iterative)" % (name,))
    return table[key]


class BAProblem:
    """Device-resident flat arrays of a bundle-adjustment problem (pxr_ba_view).

    problem: dict with obs_image, obs_point, obs_patch, image_camera, qvec, tvec,
    cam_model, cam_params (n_cams x KPAD), xyz, refs (n_points x C; None: no reference is
    subtracted -- the cost-map BA, costmap_bundle_optimizer.h:104-119).
    """

    def __init__(self, ctx, arena, problem):
        self.ctx, self.arena = ctx, arena
        g = problem
        self.n_obs = len(g["obs_image"])
        self.n_images = len(g["image_camera"])
        self.n_cameras = len(g["cam_model"])
        self.n_points = len(g["xyz"])
        cam_params = np.zeros((self.n_cameras, KPAD))
        cp = np.asarray(g["cam_params"], dtype=np.float64)
        cam_params[:, :cp.shape[1]] = cp
        self.d = {
            "obs_image": ctx.to_device(g["obs_image"], np.int32),
            "obs_point": ctx.to_device(g["obs_point"], np.int32),
            "obs_patch": ctx.to_device(g["obs_patch"], np.int64),
            "image_camera": ctx.to_device(g["image_camera"], np.int32),
            "qvec": ctx.to_device(g["qvec"], np.float64),
            "tvec": ctx.to_device(g["tvec"], np.float64),
            "cam_model": ctx.to_device(g["cam_model"], np.int32),
            "cam_params": ctx.to_device(cam_params, np.float64),
            "xyz": ctx.to_device(g["xyz"], np.float64),
            "refs": ctx.to_device(g["refs"], np.float64) if g.get("refs") is not None else None,
        }
        self.rec = ctx.empty((self.n_obs, OBS_REC), np.float64)
        self.view = self._view()

    def _view(self):
        d = self.d
        return BaView(self.n_obs, d["obs_image"].ptr, d["obs_point"].ptr, d["obs_patch"].ptr,
                      self.n_images, d["image_camera"].ptr, d["qvec"].ptr, d["tvec"].ptr,
                      self.n_cameras, d["cam_model"].ptr, d["cam_params"].ptr,
                      self.n_points, d["xyz"].ptr, d["refs"].ptr if d["refs"] is not None else None)

    def extract_costmaps(self, loss, as_gradientfield=True, apply_sqrt=False, dtype=None, out=None, first_out=0,
                         upsampling_factor=1.0, compute_cross_derivative=False, cfg=None):
        """CostMapExtractor (bundle_adjustment/src/costmap_extractor.h:177-358) on the GPU: one cost map per
        observation -- the featuremetric error of every texel of its feature patch against the reference of
        its 3D point (this problem's `refs`, e.g. after compute_references()).  Returns a PatchArena of
        n_obs maps, 3 channels [cost, dcost/dr, dcost/dc] (as_gradientfield) or 1, dtype = the features' unless
        given; map i belongs to observation i.  out: an existing arena to write into, maps first_out .. first_out + n_obs - 1
        (several problems -- chunks of a scene too large for the device -- can share one cost-map arena)."""
        ctx, a = self.ctx, self.arena
        if self.d["refs"] is None:
            raise ValueError("cost maps need reference descriptors")
        up = float(upsampling_factor)
        co = (4 if compute_cross_derivative else 3) if as_gradientfield else 1          # GetEffectiveChannels
        if out is None:
            out = PatchArena(ctx, self.n_obs, int(a.H * (up + 1.0e-6)), int(a.W * (up + 1.0e-6)), co,
                             a.dtype if dtype is None else dtype)
        if up == 1.0 and not compute_cross_derivative:
            check(ctx.lib.pxr_costmap_extract(ctx.handle, a.handle, out.handle, int(first_out), self.n_obs, self.d["obs_patch"].ptr,
                                              self.d["obs_point"].ptr, self.d["refs"].ptr, C.byref(loss),
                                              int(bool(as_gradientfield)), int(bool(apply_sqrt))), "pxr_costmap_extract")
        else:       # CostMapConfig.upsampling_factor / compute_cross_derivative: the interpolating branch, under `cfg`
            cfg = cfg if cfg is not None else interp_cfg()
            check(ctx.lib.pxr_costmap_extract_ex(ctx.handle, a.handle, out.handle, int(first_out), self.n_obs,
                                                 self.d["obs_patch"].ptr, self.d["obs_point"].ptr, self.d["refs"].ptr,
                                                 C.byref(loss), int(bool(as_gradientfield)), int(bool(apply_sqrt)),
                                                 C.byref(cfg), up, int(bool(compute_cross_derivative))), "pxr_costmap_extract_ex")
        return out

    def costmap_problem(self, costmaps):
        """The cost-map BA problem over this problem's parameters (CostMapBundleOptimizer::AddResiduals,
        costmap_bundle_optimizer.h:76-132): the SAME device qvec / tvec / cam_params / xyz arrays (refined in
        place by either problem), observation i reads cost map i, no reference descriptor."""
        p = object.__new__(BAProblem)
        p.ctx, p.arena = self.ctx, costmaps
        p.n_obs, p.n_images, p.n_cameras, p.n_points = self.n_obs, self.n_images, self.n_cameras, self.n_points
        p.d = dict(self.d)
        p.d["obs_patch"] = self.ctx.to_device(np.arange(self.n_obs, dtype=np.int64), np.int64)
        p.d["refs"] = None
        p.rec = self.ctx.empty((self.n_obs, OBS_REC), np.float64)
        p.view = p._view()
        return p

    def eval(self, cfg, with_jacobian=True, materialize=False):
        """Launch the fused residual kernel.  Returns device arrays (rec, r, gx, gy)."""
        ctx = self.ctx
        r = gx = gy = None
        if materialize:
            r = ctx.empty((self.n_obs, self.arena.C), np.float64)
            if with_jacobian:
                gx = ctx.empty((self.n_obs, self.arena.C), np.float64)
                gy = ctx.empty((self.n_obs, self.arena.C), np.float64)
        check(ctx.lib.pxr_ba_eval(ctx.handle, self.arena.handle, C.byref(self.view), C.byref(cfg),
                                  int(with_jacobian), self.rec.ptr,
                                  r.ptr if r else None, gx.ptr if gx else None, gy.ptr if gy else None),
              "pxr_ba_eval")
        return self.rec, r, gx, gy

    def eval_gram(self, cfg, reset=True, sync=True):
        """The records of eval(with_jacobian=True) through the Gram-matrix cache of the context (pxr_ba_eval_gram).  Returns
        (rec, number of observations whose matrices this call built -- None without `sync`)."""
        built = C.c_int32(0)
        check(self.ctx.lib.pxr_ba_eval_gram(self.ctx.handle, self.arena.handle, C.byref(self.view), C.byref(cfg), int(bool(reset)),
                                            self.rec.ptr, C.byref(built) if sync else None), "pxr_ba_eval_gram")
        return self.rec, (built.value if sync else None)

    def projection_jacobian(self, out=None):
        """The 2 x (10+K) projection Jacobian P of every observation (k_jac); `out` re-uses a device array."""
        P = out if out is not None else self.ctx.empty((self.n_obs, 2, 10 + KPAD), np.float64)
        check(self.ctx.lib.pxr_ba_projection_jacobian(self.ctx.handle, C.byref(self.view), P.ptr),
              "pxr_ba_projection_jacobian")
        return P

    def solve(self, cfg, loss, pose_const, tvec_const_mask, cam_const_mask, point_const, options=None,
              allreduce=None):
        """Run the GPU LM (pxr_ba_solve); parameters are refined in place on the device.

        allreduce: optional callable(device_ptr:int, count:int) summing `count` doubles in place
        over the ranks (see parallel.make_allreduce); None for one GPU.
        Returns the summary dict; use params() to fetch the refined parameters.
        """
        ctx = self.ctx
        pc = np.ascontiguousarray(pose_const, dtype=np.uint8)
        tm = np.ascontiguousarray(tvec_const_mask, dtype=np.uint8)
        cm = np.ascontiguousarray(cam_const_mask, dtype=np.uint16)
        ptc = np.ascontiguousarray(point_const, dtype=np.uint8)
        if len(pc) != self.n_images or len(tm) != self.n_images or len(cm) != self.n_cameras \
                or len(ptc) != self.n_points:
            raise ValueError("parameterisation arrays do not match the problem dimensions")
        opts = options or lm_options()
        summ = _lib.LMSummary()
        cb = None
        if allreduce is not None:
            def _cb(user, ptr, count):
                try:
                    allreduce(ptr, count)
                    return 0
                except Exception as e:  # noqa: BLE001 -- must not unwind through C
                    import sys
                    print("all-reduce callback failed: %r" % (e,), file=sys.stderr)
                    return 1
            cb = _lib.ALLREDUCE_FN(_cb)
        check(ctx.lib.pxr_ba_solve(ctx.handle, self.arena.handle, C.byref(self.view), C.byref(cfg), C.byref(loss),
                                   pc.ctypes.data, tm.ctypes.data, cm.ctypes.data, ptc.ctypes.data,
                                   C.byref(opts), C.cast(cb, C.c_void_p) if cb else None, None, C.byref(summ)),
              "pxr_ba_solve")
        return summ.as_dict()

    def compute_references(self, cfg, loss, iters=100, keep_mean=False, keep_observations=False):
        """ReferenceExtractor.run on the GPU: fills this problem's `refs` in place (device) and
        returns (ref_obs indices, robust means | None) as numpy arrays.  keep_observations
        (reference_extractor.h:60): the per-observation descriptors stay on the device in
        self.obs_desc (n_obs x C), e.g. as candidates for nearest_references()."""
        ctx = self.ctx
        ref_obs = ctx.empty((self.n_points,), np.int64)
        mean = ctx.empty((self.n_points, self.arena.C), np.float64) if keep_mean else None
        self.obs_desc = ctx.empty((self.n_obs, self.arena.C), np.float64) if keep_observations else None
        check(ctx.lib.pxr_ba_compute_references(ctx.handle, self.arena.handle, C.byref(self.view), C.byref(cfg),
                                                C.byref(loss), int(iters), self.d["refs"].ptr, ref_obs.ptr,
                                                mean.ptr if mean else None,
                                                self.obs_desc.ptr if self.obs_desc else None), "pxr_ba_compute_references")
        return ref_obs.download(), (mean.download() if mean else None)

    def params(self):
        """Download (qvec, tvec, cam_params, xyz)."""
        d = self.d
        return d["qvec"].download(), d["tvec"].download(), d["cam_params"].download(), d["xyz"].download()

    def cost(self, loss):
        out = C.c_double()
        check(self.ctx.lib.pxr_ba_cost(self.ctx.handle, self.rec.ptr, self.n_obs, C.byref(loss), C.byref(out)),
              "pxr_ba_cost")
        return out.value
