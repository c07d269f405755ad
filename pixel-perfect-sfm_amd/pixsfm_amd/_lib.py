"""ctypes loader of libpixsfm_hip.so (the C-ABI declared in include/pixsfm_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, an exception
is raised.  Nothing under oracle/ is ever imported from the product package.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PXR_HIP_LIB") or os.path.join(HERE, "libpixsfm_hip.so")   # PXR_HIP_LIB: an alternative build (debugging)

KPAD = 12
OBS_REC = 8
F16, F32, F64 = 0, 1, 2
CAMERA_MODEL_IDS = {"SIMPLE_PINHOLE": 0, "PINHOLE": 1, "SIMPLE_RADIAL": 2, "RADIAL": 3, "OPENCV": 4, "OPENCV_FISHEYE": 5,
                    "FULL_OPENCV": 6, "FOV": 7, "SIMPLE_RADIAL_FISHEYE": 8, "RADIAL_FISHEYE": 9, "THIN_PRISM_FISHEYE": 10}
LOSS_IDS = {"trivial": 0, "cauchy": 1, "huber": 2, "soft_l1": 3}


class PixsfmHipError(RuntimeError):
    code = 0


PXR_EUNSUPPORTED = -4


class InterpCfg(C.Structure):
    _fields_ = [("l2_normalize", C.c_int32), ("use_float_simd", C.c_int32), ("check_bounds", C.c_int32)]


class Loss(C.Structure):
    _fields_ = [("type", C.c_int32), ("a", C.c_double)]


class BaView(C.Structure):
    _fields_ = [("n_obs", C.c_int64), ("d_obs_image", C.c_void_p), ("d_obs_point", C.c_void_p),
                ("d_obs_patch", C.c_void_p), ("n_images", C.c_int32), ("d_image_camera", C.c_void_p),
                ("d_qvec", C.c_void_p), ("d_tvec", C.c_void_p), ("n_cameras", C.c_int32),
                ("d_cam_model", C.c_void_p), ("d_cam_params", C.c_void_p), ("n_points", C.c_int64),
                ("d_xyz", C.c_void_p), ("d_refs", C.c_void_p)]


class LMOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double), ("max_consecutive_invalid_steps", C.c_int32),
                ("jacobi_scaling", C.c_int32), ("use_inner_iterations", C.c_int32),
                ("inner_iteration_tolerance", C.c_double), ("linear_solver", C.c_int32),
                ("max_linear_solver_iterations", C.c_int32), ("eta", C.c_double), ("linear_r_tolerance", C.c_double)]


class LMSummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("num_successful", C.c_int32), ("termination", C.c_int32),
                ("num_camera_unknowns", C.c_int32), ("num_point_unknowns", C.c_int64),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_radius", C.c_double),
                ("total_ms", C.c_double), ("setup_ms", C.c_double), ("linear_solver", C.c_int32),
                ("collective_kib", C.c_int32), ("linear_iterations", C.c_int64), ("accumulation", C.c_int32),
                ("initial_us", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)


class KaView(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("d_kp", C.c_void_p), ("d_node_patch", C.c_void_p),
                ("d_node_const", C.c_void_p), ("n_edges", C.c_int64), ("d_edge_src", C.c_void_p),
                ("d_edge_dst", C.c_void_p), ("d_edge_w", C.c_void_p), ("n_problems", C.c_int32),
                ("d_prob_node_ptr", C.c_void_p), ("d_prob_nodes", C.c_void_p), ("d_prob_edge_ptr", C.c_void_p),
                ("d_prob_edges", C.c_void_p), ("n_unary", C.c_int64), ("d_unary_node", C.c_void_p),
                ("d_unary_ref", C.c_void_p), ("d_unary_w", C.c_void_p), ("d_prob_unary_ptr", C.c_void_p),
                ("d_prob_unary", C.c_void_p), ("d_prob_group", C.c_void_p)]

# every symbol include/pixsfm_hip.h declares (checked by tests/test_cabi_and_host.py)
_SIGNATURES = {
    "pxr_version": (C.c_int, []),
    "pxr_last_error": (C.c_char_p, []),
    "pxr_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "pxr_ctx_destroy": (C.c_int, [C.c_void_p]),
    "pxr_ctx_sync": (C.c_int, [C.c_void_p]),
    "pxr_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pxr_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pxr_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pxr_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pxr_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "pxr_timer_start": (C.c_int, [C.c_void_p]),
    "pxr_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "pxr_arena_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                   C.POINTER(C.c_void_p)]),
    "pxr_arena_destroy": (C.c_int, [C.c_void_p]),
    "pxr_arena_upload": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pxr_arena_upload_gather": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pxr_arena_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_double, C.c_double, C.c_int]),
    "pxr_arena_data": (C.c_void_p, [C.c_void_p]),
    "pxr_arena_corners": (C.c_void_p, [C.c_void_p]),
    "pxr_arena_scales": (C.c_void_p, [C.c_void_p]),
    "pxr_arena_size": (C.c_int64, [C.c_void_p]),
    "pxr_ba_eval": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BaView), C.POINTER(InterpCfg), C.c_int,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pxr_ba_eval_gram": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BaView), C.POINTER(InterpCfg), C.c_int, C.c_void_p,
                                   C.POINTER(C.c_int32)]),
    "pxr_ba_projection_jacobian": (C.c_int, [C.c_void_p, C.POINTER(BaView), C.c_void_p]),
    "pxr_ba_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BaView), C.POINTER(InterpCfg), C.POINTER(Loss),
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(LMOptions),
                               C.c_void_p, C.c_void_p, C.POINTER(LMSummary)]),
    "pxr_ba_compute_references": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BaView), C.POINTER(InterpCfg),
                                            C.POINTER(Loss), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pxr_costmap_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.POINTER(Loss), C.c_int, C.c_int]),
    "pxr_costmap_extract_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.POINTER(Loss), C.c_int, C.c_int, C.POINTER(InterpCfg), C.c_double, C.c_int]),
    "pxr_arena_set_upsampling": (C.c_int, [C.c_void_p, C.c_double]),
    "pxr_arena_upsampling": (C.c_double, [C.c_void_p]),
    "pxr_interpolate": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(InterpCfg), C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "pxr_nearest_references": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(InterpCfg), C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pxr_ka_eval": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(KaView), C.POINTER(InterpCfg), C.POINTER(Loss),
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pxr_ka_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(KaView), C.POINTER(InterpCfg), C.POINTER(Loss),
                               C.c_double, C.POINTER(LMOptions), C.c_void_p, C.POINTER(LMSummary)]),
    "pxr_dense_spd_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "pxr_graph_track_labels": (C.c_int, [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_int64)]),
    "pxr_graph_score_labels": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pxr_graph_root_labels": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pxr_graph_labels_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "pxr_ka_build_edges": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "pxr_ba_build_problem": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64),
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pxr_ba_cost": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(Loss), C.POINTER(C.c_double)]),
    "pxr_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pxr_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "pxr_comm_destroy": (C.c_int, [C.c_void_p]),
    "pxr_set_iteration_callback": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pxr_set_deterministic": (C.c_int, [C.c_void_p, C.c_int]),
    "pxr_get_deterministic": (C.c_int, [C.c_void_p]),
    "pxr_set_gram_cache": (C.c_int, [C.c_void_p, C.c_int]),
    "pxr_get_gram_cache": (C.c_int, [C.c_void_p]),
    "pxr_comm_set_rank": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pxr_comm_rank": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pxr_comm_allreduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pxr_comm_force": (C.c_int, [C.c_void_p, C.c_int]),
    "pxr_comm_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
}
LINEAR_AUTO, LINEAR_DIRECT, LINEAR_ITERATIVE = 0, 1, 2
COMM_ID_BYTES = 128

_lib = None


class IterationSummary(C.Structure):
    """pxr_iteration_summary"""
    _fields_ = [("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32), ("cost", C.c_double),
                ("cost_change", C.c_double), ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double), ("step_norm", C.c_double)]


ITERATION_CALLBACK = C.CFUNCTYPE(C.c_int, C.POINTER(IterationSummary), C.c_void_p)


def declared_symbols():
    return sorted(_SIGNATURES)


def load():
    """Load libpixsfm_hip.so; raises PixsfmHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PixsfmHipError(
            "libpixsfm_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: when PyTorch-ROCm is installed it must load ITS libamdhip64 first
    # (loading ours first makes a later torch.cuda initialisation fail with "No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().pxr_last_error()
        err = PixsfmHipError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))
        err.code = int(rc)
        raise err
