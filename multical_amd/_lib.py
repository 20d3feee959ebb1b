"""ctypes binding of the C ABI in include/mcba.h (libmcba.so, built by multical_amd.build).

The library is HIP-only.  Importing this module never falls back to a CPU implementation: if the shared object
is missing, `load()` raises with the build command; if no gfx950 device is present, `mcba_create` fails.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCBA_LIB_PATH") or os.path.join(HERE, "_build", "libmcba.so")   # (override: kernel experiments)

MCBA_VERSION = 3
MOTION_STATIC, MOTION_ROLLING, MOTION_HAND_EYE = 0, 1, 2
CAMERA_PINHOLE, CAMERA_FISHEYE = 0, 1
LOSSES = dict(linear=0, soft_l1=1, huber=2, cauchy=3, arctan=4)
TR_SOLVERS = dict(exact=0, lsmr=1)     # mcba_options.tr_solver (MCBA_TR_*)
OPT_BITS = dict(camera_poses=1, board_poses=2, motion=4, cameras=8, boards=16)
PARAM_ORDER = ["camera_poses", "board_poses", "motion", "cameras", "boards"]   # calibration.py:146-153

c_double_p = C.POINTER(C.c_double)
c_uint8_p = C.POINTER(C.c_uint8)
c_int32_p = C.POINTER(C.c_int32)


class Problem(C.Structure):
  _fields_ = [
    ("version", C.c_int32), ("n_cameras", C.c_int32), ("n_frames", C.c_int32), ("n_boards", C.c_int32),
    ("n_points", C.c_int32),
    ("points", c_double_p), ("point_valid", c_uint8_p), ("inlier_mask", c_uint8_p),
    ("board_sizes", c_int32_p), ("camera_valid", c_uint8_p), ("frame_valid", c_uint8_p), ("board_valid", c_uint8_p),
    ("motion", C.c_int32), ("camera_model", C.c_int32), ("n_dist", C.c_int32),
    ("image_heights", c_double_p), ("fix_aspect", c_uint8_p), ("base_wrt_gripper", c_double_p),
    ("optimize", C.c_uint32), ("x_full", c_double_p),
    ("frame_begin", C.c_int32), ("frame_end", C.c_int32),
    ("camera_n_dist", c_int32_p),
    ("camera_fisheye", c_uint8_p),
    ("points_f32", C.POINTER(C.c_float)),
  ]


class Options(C.Structure):
  _fields_ = [("ftol", C.c_double), ("xtol", C.c_double), ("gtol", C.c_double), ("max_nfev", C.c_int32),
              ("loss", C.c_int32), ("f_scale", C.c_double), ("verbose", C.c_int32), ("tr_solver", C.c_int32)]


class Result(C.Structure):
  _fields_ = [("cost", C.c_double), ("initial_cost", C.c_double), ("optimality", C.c_double), ("nfev", C.c_int32),
              ("njev", C.c_int32), ("status", C.c_int32), ("iterations", C.c_int32), ("solve_seconds", C.c_double),
              ("linearize_seconds", C.c_double)]


class RoundReport(C.Structure):   # mcba_round_report
  _fields_ = [("rms_all", C.c_double), ("rms_inliers", C.c_double), ("n_all", C.c_int64), ("n_inliers", C.c_int64),
              ("quantiles", C.c_double * 5), ("f_scale", C.c_double), ("threshold", C.c_double), ("n_kept", C.c_int64),
              ("n_valid", C.c_int64), ("solve", Result)]


LOG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p)

# every symbol include/mcba.h declares: (name, restype, argtypes)
H = C.c_void_p
SYMBOLS = [
  ("mcba_last_error", C.c_char_p, []),
  ("mcba_full_size", C.c_int32, [C.POINTER(Problem), C.POINTER(C.c_int64)]),
  ("mcba_create", C.c_int32, [C.POINTER(Problem), C.c_void_p, C.POINTER(H)]),
  ("mcba_destroy", C.c_int32, [H]),
  ("mcba_release_cached_memory", C.c_int32, []),
  ("mcba_device_info", C.c_int32, [H, C.c_char_p, C.c_size_t]),
  ("mcba_num_params", C.c_int32, [H, C.POINTER(C.c_int64)]),
  ("mcba_num_residuals", C.c_int32, [H, C.POINTER(C.c_int64)]),
  ("mcba_set_inliers", C.c_int32, [H, c_uint8_p]),
  ("mcba_set_allreduce", C.c_int32, [H, ALLREDUCE_FN, C.c_void_p]),
  ("mcba_set_shard_root", C.c_int32, [H, C.c_int32]),
  ("mcba_set_shard_rank", C.c_int32, [H, C.c_int32, C.c_int32]),
  ("mcba_allreduce_stats", C.c_int32, [H, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                       C.c_int32, C.POINTER(C.c_int32)]),
  ("mcba_set_log", C.c_int32, [H, LOG_FN, C.c_void_p]),
  ("mcba_residuals", C.c_int32, [H, c_double_p, c_double_p]),
  ("mcba_jacobian", C.c_int32, [H, c_double_p, c_int32_p, c_double_p, c_int32_p]),
  ("mcba_reprojection_error", C.c_int32, [H, c_double_p, c_double_p, c_uint8_p]),
  ("mcba_project", C.c_int32, [H, c_double_p, c_double_p]),
  ("mcba_project_model", C.c_int32, [H, c_double_p, C.c_int32, c_double_p]),
  ("mcba_align_poses_robust", C.c_int32, [C.c_int32, C.POINTER(C.c_int64), c_double_p, c_double_p, c_uint8_p, C.c_double,
                                          C.c_int32, c_double_p, c_uint8_p, c_uint8_p]),
  ("mcba_align_poses_indexed", C.c_int32, [C.c_int32, C.POINTER(C.c_int64), c_double_p, C.c_int64, c_int32_p, c_double_p, C.c_int64,
                                           c_int32_p, c_uint8_p, C.c_double, C.c_int32, c_double_p, c_uint8_p, c_uint8_p]),
  ("mcba_error_stats", C.c_int32, [H, c_double_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), c_double_p,
                                   C.POINTER(C.c_int64), c_double_p]),
  ("mcba_error_count", C.c_int32, [H, C.c_int32, C.POINTER(C.c_int64)]),
  ("mcba_reject_outliers", C.c_int32, [H, c_double_p, C.c_double, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
  ("mcba_get_inliers", C.c_int32, [H, c_uint8_p]),
  ("mcba_adjust_outliers", C.c_int32, [H, c_double_p, C.POINTER(Options), C.c_int32, C.c_double, C.c_double, C.c_double,
                                       C.c_double, C.POINTER(RoundReport), c_uint8_p]),
  ("mcba_normal_equations", C.c_int32, [H, c_double_p, C.POINTER(Options), c_double_p, c_double_p, c_double_p]),
  ("mcba_normal_equations_device", C.c_int32, [H, C.POINTER(Options)]),
  ("mcba_synchronize", C.c_int32, [H]),
  ("mcba_dense_hessian", C.c_int32, [H, c_double_p]),
  ("mcba_solve", C.c_int32, [H, c_double_p, C.POINTER(Options), C.POINTER(Result)]),
  ("mcba_time_linearize", C.c_int32, [H, c_double_p, C.POINTER(Options), C.c_int32, c_double_p]),
  ("mcba_time_residuals", C.c_int32, [H, c_double_p, C.c_int32, c_double_p]),
  ("mcba_time_lsmr_iteration", C.c_int32, [H, c_double_p, C.c_int32, c_double_p]),
  ("mcba_rccl_unique_id", C.c_int32, [C.POINTER(C.c_uint8)]),
  ("mcba_rccl_init", C.c_int32, [H, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]),
  ("mcba_rccl_shutdown", C.c_int32, [H]),
  ("mcba_rccl_version", C.c_int32, [C.POINTER(C.c_int32)]),
  ("mcba_set_mfma", C.c_int32, [H, C.c_int32]),
  ("mcba_debug_set_lin_grid", C.c_int32, [H, C.c_int32]),
  ("mcba_debug_set_frame_groups", C.c_int32, [H, C.c_int32]),
  ("mcba_debug_pipe_probe", C.c_int32, [C.c_int32, C.POINTER(C.c_double)]),
  ("mcba_debug_dot3", C.c_int32, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
  ("mcba_debug_dispatch_probe", C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_longlong)]),
  ("mcba_debug_xcd_probe", C.c_int32, [C.c_int32, C.POINTER(C.c_longlong)]),
  ("mcba_debug_gn_step", C.c_int32, [H, C.c_double, c_double_p, c_double_p, c_double_p]),
  ("mcba_debug_mfma_probe", C.c_int32, [c_double_p, c_double_p]),
  ("mcba_debug_chol", C.c_int32, [H, C.c_int32, c_double_p, c_double_p, C.c_double, C.c_int32, c_double_p]),
  ("mcba_debug_linearize_profile", C.c_int32, [H, c_double_p, C.POINTER(C.c_longlong)]),
  ("mcba_debug_lsmr_products", C.c_int32, [H, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
  ("mcba_debug_lsmr_fused_products", C.c_int32, [H, c_double_p, c_double_p, c_double_p, c_double_p]),
  ("mcba_debug_lsmr_info", C.c_int32, [H, C.POINTER(C.c_int64)]),
  ("mcba_debug_set_lsmr_fused", C.c_int32, [H, C.c_int32]),
  ("mcba_debug_set_switch", C.c_int32, [C.c_char_p, C.c_char_p]),
  ("mcba_debug_lsmr_solve", C.c_int32, [H, c_double_p, C.POINTER(Options), C.c_double, c_double_p, C.c_int32, c_double_p, c_double_p, c_double_p]),
  ("mcba_debug_lsmr_trace", C.c_int32, [H, C.c_int32, c_double_p, C.POINTER(C.c_int32)]),
  ("mcba_debug_set_lsmr_trace", C.c_int32, [H, C.c_int32]),
  ("mcba_debug_set_lsmr_grid", C.c_int32, [H, C.c_int32]),
  ("mcba_debug_set_allreduce_trace", C.c_int32, [H, C.c_int32]),
  ("mcba_debug_set_lsmr_masks_form", C.c_int32, [H, C.c_int32]),
]

_lib = None


def load():
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise RuntimeError(f"{LIB_PATH} is missing: build the HIP back-end first (python -m multical_amd.build). "
                       "multical_amd has no CPU fallback.")
  lib = C.CDLL(LIB_PATH)
  for name, restype, argtypes in SYMBOLS:
    try:
      fn = getattr(lib, name)
    except AttributeError:
      # only a side-by-side build loaded through MCBA_LIB_PATH (A/B measurements against an older library) may lack an
      # entry point; the product library must export everything
      if os.environ.get("MCBA_LIB_PATH"):
        continue
      raise
    fn.restype = restype
    fn.argtypes = argtypes
  _lib = lib
  return lib


def set_switch(name, value):
  """Experiment / path-forcing switch of the library (mcba_debug_set_switch): tests and profiling scripts only; process-wide, before
  the first Handle.  The product library does not read these from the environment."""
  rc = load().mcba_debug_set_switch(name.encode(), str(value).encode())
  if rc != 0:
    raise RuntimeError(load().mcba_last_error().decode("utf-8", "replace"))


class McbaError(RuntimeError):
  pass


def check(rc):
  if rc != 0:
    msg = load().mcba_last_error().decode("utf-8", "replace")
    # scipy raises ValueError for non-finite initial residuals (least_squares.py:844-845); keep that convention
    if "not finite" in msg:
      raise ValueError(msg)
    raise McbaError(msg)
