"""Lowering of a `Calibration` to the flat problem description of include/mcba.h, and the handle wrapper.

`lower()` is duck-typed: it reads the attributes that both multical_amd.calibration.Calibration and the reference's
multical.optimization.calibration.Calibration expose (calibration.py:43-61,146-153), so the very same code path serves
the host mirror in this package and the drop-in patch of the real multical (multical_amd.dropin).

Nothing here computes residuals on the host: every numeric method forwards to libmcba.so (HIP, gfx950).
"""
import ctypes as C
import time
from types import SimpleNamespace

import numpy as np

from . import _lib
from ._lib import (Problem, Options, Result, RoundReport, LOSSES, OPT_BITS, PARAM_ORDER, MOTION_STATIC, MOTION_ROLLING,
                   MOTION_HAND_EYE, CAMERA_PINHOLE, CAMERA_FISHEYE, check)


def _ptr(a, ctype):
  return a.ctypes.data_as(C.POINTER(ctype))


def _u8(a):
  a = np.asarray(a)
  if a.dtype == np.bool_ and a.flags.c_contiguous:
    return a.view(np.uint8)          # masks are bool tables of millions of slots: reinterpret, do not copy
  if a.dtype == np.uint8 and a.flags.c_contiguous:
    return a
  return np.ascontiguousarray(a.astype(np.uint8))


def _f64(a):
  return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _class_name(obj):
  return type(obj).__name__


def _optimize_bits(optimize):
  bits = 0
  for k in PARAM_ORDER:
    if optimize[k] is True:     # calibration.py:160  `self.optimize[k] is True`
      bits |= OPT_BITS[k]
  return bits


def lower(calib):
  """Calibration -> SimpleNamespace of C-contiguous numpy arrays + scalars (keeps them alive for the C call)."""
  points = calib.point_table.points
  C_, F, B, P = calib.point_table.valid.shape
  p = SimpleNamespace()
  p.shape = (C_, F, B, P)
  # float32 detections (cv2's corner dtype, which the reference's table keeps: tables.py:15-17) go up as they are and are
  # widened on the device -- exactly, as numpy promotes them in `reprojected.points - point_table.points`
  points = np.asarray(points)
  p.points_f32 = np.ascontiguousarray(points) if points.dtype == np.float32 else None
  p.points = None if p.points_f32 is not None else _f64(points)
  p.point_valid = _u8(calib.point_table.valid)
  p.inlier_mask = None if calib.inlier_mask is None else _u8(calib.inlier_mask)
  p.board_sizes = np.ascontiguousarray(np.array([b.num_points for b in calib.boards], dtype=np.int32))
  p.camera_valid = _u8(calib.camera_poses.valid)
  p.board_valid = _u8(calib.board_poses.valid)
  p.frame_valid = _u8(calib.motion.valid)

  motion = calib.motion
  mname = _class_name(motion)
  p.base_wrt_gripper = None
  if mname == "RollingFrames":
    p.motion = MOTION_ROLLING
  elif mname == "HandEye":
    p.motion = MOTION_HAND_EYE
    p.base_wrt_gripper = _f64(motion.base_wrt_gripper.poses)
  elif mname == "StaticFrames":
    p.motion = MOTION_STATIC
  else:
    raise TypeError(f"unsupported motion model {mname}")

  cams = list(calib.cameras)
  fisheye = [_class_name(c) == "CameraFisheye" or getattr(c, "model", None) == "fisheye" for c in cams]
  p.camera_model = CAMERA_FISHEYE if all(fisheye) else CAMERA_PINHOLE
  # every Camera is an independent object (optimization/parameters.py:54-85): the projection family (Camera / CameraFisheye)
  # and the distortion size (models `standard` / `rational` / `thin_prism` / `tilted`, or a 4-coefficient file) may differ
  # from camera to camera; the library pads to the largest block and freezes the coefficients a camera does not have
  p.camera_fisheye = None
  if any(fisheye) != all(fisheye):
    p.camera_fisheye = np.ascontiguousarray(np.array(fisheye, dtype=np.uint8))
  nds = [int(np.asarray(c.dist).size) for c in cams]
  p.n_dist = max(nds)
  p.camera_n_dist = None
  if len(set(nds)) != 1:
    if all(fisheye):
      raise ValueError(f"fisheye cameras carry 4 distortion coefficients, got {sorted(set(nds))}")
    p.camera_n_dist = np.ascontiguousarray(np.array(nds, dtype=np.int32))
  p.image_heights = _f64([c.image_size[1] for c in cams])
  p.fix_aspect = _u8([bool(c.fix_aspect) for c in cams])

  p.optimize = _optimize_bits(calib.optimize)
  # all five parameter blocks in reference order (calibration.py:146-153), enabled or not.  (`param_vec` is a cached
  # property of every block object: the vectors are shared with Calibration.param_vec, nothing is converted twice.)
  objs = calib.param_objects
  vecs = [np.asarray(objs[k].param_vec, dtype=np.float64).ravel() for k in PARAM_ORDER]
  p.x_full = _f64(np.concatenate(vecs))
  p.block_sizes = [int(v.size) for v in vecs]
  p.n_params = sum(n for k, n in zip(PARAM_ORDER, p.block_sizes) if calib.optimize[k] is True)
  return p


def _to_struct(p, frame_range=None):
  C_, F, B, P = p.shape
  s = Problem()
  s.version = _lib.MCBA_VERSION
  s.n_cameras, s.n_frames, s.n_boards, s.n_points = C_, F, B, P
  s.points = None if p.points is None else _ptr(p.points, C.c_double)
  s.points_f32 = None if getattr(p, "points_f32", None) is None else _ptr(p.points_f32, C.c_float)
  s.point_valid = _ptr(p.point_valid, C.c_uint8)
  s.inlier_mask = None if p.inlier_mask is None else _ptr(p.inlier_mask, C.c_uint8)
  s.board_sizes = _ptr(p.board_sizes, C.c_int32)
  s.camera_valid = _ptr(p.camera_valid, C.c_uint8)
  s.frame_valid = _ptr(p.frame_valid, C.c_uint8)
  s.board_valid = _ptr(p.board_valid, C.c_uint8)
  s.motion, s.camera_model, s.n_dist = p.motion, p.camera_model, p.n_dist
  s.image_heights = _ptr(p.image_heights, C.c_double)
  s.fix_aspect = _ptr(p.fix_aspect, C.c_uint8)
  s.base_wrt_gripper = None if p.base_wrt_gripper is None else _ptr(p.base_wrt_gripper, C.c_double)
  s.optimize = p.optimize
  s.x_full = _ptr(p.x_full, C.c_double)
  s.frame_begin, s.frame_end = (-1, -1) if frame_range is None else frame_range
  s.camera_n_dist = None if getattr(p, "camera_n_dist", None) is None else _ptr(p.camera_n_dist, C.c_int32)
  s.camera_fisheye = None if getattr(p, "camera_fisheye", None) is None else _ptr(p.camera_fisheye, C.c_uint8)
  return s


def make_options(tolerance=1e-4, f_scale=1.0, max_iterations=100, loss='linear', xtol=1e-8, gtol=1e-8, verbose=2,
                 tr_solver='exact'):
  if loss not in LOSSES:
    raise ValueError(f"`loss` must be one of {list(LOSSES)} or a callable.")   # scipy's message
  if tr_solver not in _lib.TR_SOLVERS:
    raise ValueError("`tr_solver` must be 'exact' or 'lsmr'.")
  o = Options()
  o.tr_solver = _lib.TR_SOLVERS[tr_solver]
  o.ftol, o.xtol, o.gtol = float(tolerance), float(xtol), float(gtol)
  o.max_nfev = int(max_iterations)
  o.loss = LOSSES[loss]
  o.f_scale = float(f_scale)
  o.verbose = int(verbose)
  return o


STATUS_MESSAGES = {   # scipy/optimize/_lsq/least_squares.py TERMINATION_MESSAGES
  -1: "Improper input parameters status returned from `leastsq`",
  0: "The maximum number of function evaluations is exceeded.",
  1: "`gtol` termination condition is satisfied.",
  2: "`ftol` termination condition is satisfied.",
  3: "`xtol` termination condition is satisfied.",
  4: "Both `ftol` and `xtol` termination conditions are satisfied.",
}


class Handle(object):
  """Owns one mcba_handle (device tables of one Calibration).  Not picklable by design: Calibration objects hold no
  handle (calibration.py:222-226 pickles only the 8 constructor fields); handles are created per call."""

  def __init__(self, calib_or_problem, frame_range=None, stream=None):
    self.lib = _lib.load()
    self.problem = calib_or_problem if isinstance(calib_or_problem, SimpleNamespace) else lower(calib_or_problem)
    self._struct = _to_struct(self.problem, frame_range)
    h = C.c_void_p()
    check(self.lib.mcba_create(C.byref(self._struct), C.c_void_p(stream) if stream else None, C.byref(h)))
    self.h = h
    n = C.c_int64()
    check(self.lib.mcba_num_params(self.h, C.byref(n)))
    self.n_params = n.value
    self._log_cb = self._allreduce_cb = None    # the currently installed ctypes thunks (kept alive, one per kind)
    self._jac_pattern = None                    # (shape, indices, indptr, keep) of mcba_jacobian for the current inlier set
    self.shape = self.problem.shape

  def close(self):
    if getattr(self, "h", None):
      self.lib.mcba_destroy(self.h)
      self.h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  def __enter__(self):
    return self

  def __exit__(self, *a):
    self.close()

  # --- info ---------------------------------------------------------------------------------------------------
  @property
  def n_residuals(self):
    n = C.c_int64()
    check(self.lib.mcba_num_residuals(self.h, C.byref(n)))
    return n.value

  def device_info(self):
    buf = C.create_string_buffer(256)
    check(self.lib.mcba_device_info(self.h, buf, 256))
    return buf.value.decode()

  def set_mfma(self, on):
    check(self.lib.mcba_set_mfma(self.h, 1 if on else 0))

  def set_lin_grid(self, grid):
    check(self.lib.mcba_debug_set_lin_grid(self.h, int(grid)))

  def set_frame_groups(self, nw):
    """experiment: bind the views of a frame to nw waves (0 = the product's largest-first list of views)."""
    check(self.lib.mcba_debug_set_frame_groups(self.h, int(nw)))

  def set_inliers(self, mask):
    self._jac_pattern = None
    if mask is None:
      check(self.lib.mcba_set_inliers(self.h, None))
    else:
      m = _u8(mask)
      assert m.shape == tuple(self.shape), f"inlier mask shape {m.shape} != {self.shape}"
      check(self.lib.mcba_set_inliers(self.h, _ptr(m, C.c_uint8)))

  def _x(self, x):
    x = _f64(x)
    assert x.shape == (self.n_params,), f"inconsistent parameter sizes, got {x.size}, expected {self.n_params}"
    return x

  # --- evaluation ---------------------------------------------------------------------------------------------
  def residuals(self, x):
    x = self._x(x)
    r = np.empty(self.n_residuals)
    check(self.lib.mcba_residuals(self.h, _ptr(x, C.c_double), _ptr(r, C.c_double)))
    return r

  def jacobian(self, x):
    """scipy.sparse.csr_matrix [n_residuals, n_params] in the reference's sparsity pattern (calibration.py:173-196).
    The pattern (column indices, row pointers) is fetched once per inlier set; later calls download the values only."""
    from scipy.sparse import csr_matrix
    x = self._x(x)
    nnz = C.c_int32()
    check(self.lib.mcba_jacobian(self.h, None, C.byref(nnz), None, None))
    k = nnz.value
    m = self.n_residuals
    vals = np.empty((m, k))
    st = self._jac_pattern
    if st is None or st[0] != (m, k):
      cols = np.empty((m // 2, k), dtype=np.int32)
      check(self.lib.mcba_jacobian(self.h, _ptr(x, C.c_double), C.byref(nnz), _ptr(vals, C.c_double), _ptr(cols, C.c_int32)))
      indices = np.repeat(cols, 2, axis=0).ravel()
      if (cols < 0).any():   # ragged camera blocks: slots of coefficients a camera's model does not have carry column -1
        keep = indices >= 0
        counts = keep.reshape(m, k).sum(axis=1)
        st = ((m, k), indices[keep], np.concatenate([[0], np.cumsum(counts)]), keep)
      else:
        st = ((m, k), indices, np.arange(0, m * k + 1, k), None)
      self._jac_pattern = st
    else:
      check(self.lib.mcba_jacobian(self.h, _ptr(x, C.c_double), C.byref(nnz), _ptr(vals, C.c_double), None))
    _, indices, indptr, keep = st
    data = vals.ravel() if keep is None else vals.ravel()[keep]
    return csr_matrix((data, indices, indptr), shape=(m, self.n_params))

  def reprojection_error(self, x):
    x = self._x(x)
    err = np.empty(self.shape)
    valid = np.empty(self.shape, dtype=np.uint8)
    check(self.lib.mcba_reprojection_error(self.h, _ptr(x, C.c_double), _ptr(err, C.c_double), _ptr(valid, C.c_uint8)))
    return err, valid.astype(bool)

  def project(self, x):
    x = self._x(x)
    out = np.empty(tuple(self.shape) + (2,))
    check(self.lib.mcba_project(self.h, _ptr(x, C.c_double), _ptr(out, C.c_double)))
    return out

  def project_model(self, x, max_iterations=4):
    """Calibration.projected (calibration.py:113-119): projection without the measured points (rolling shutter: scan
    time iterated from the projected row, motion/rolling_frames.py:125-133)."""
    x = self._x(x)
    out = np.empty(tuple(self.shape) + (2,))
    check(self.lib.mcba_project_model(self.h, _ptr(x, C.c_double), int(max_iterations), _ptr(out, C.c_double)))
    return out

  def normal_equations(self, x, loss='linear', f_scale=1.0):
    x = self._x(x)
    opt = make_options(loss=loss, f_scale=f_scale)
    cost = C.c_double()
    g = np.empty(self.n_params)
    diag = np.empty(self.n_params)
    check(self.lib.mcba_normal_equations(self.h, _ptr(x, C.c_double), C.byref(opt), C.byref(cost),
                                         _ptr(g, C.c_double), _ptr(diag, C.c_double)))
    return cost.value, g, diag

  def normal_equations_device(self, loss='linear', f_scale=1.0):
    """The same evaluation at the x already on the device, enqueued without host transfer or synchronisation (results
    stay in HBM: read them with dense_hessian / debug_gn_step, or continue with a solve)."""
    opt = make_options(loss=loss, f_scale=f_scale)
    check(self.lib.mcba_normal_equations_device(self.h, C.byref(opt)))

  def synchronize(self):
    check(self.lib.mcba_synchronize(self.h))

  def dense_hessian(self):
    H = np.empty((self.n_params, self.n_params))
    check(self.lib.mcba_dense_hessian(self.h, _ptr(H, C.c_double)))
    return H

  def debug_gn_step(self, reg):
    gn = np.empty(self.n_params)
    gh = np.empty(self.n_params)
    si = np.empty(self.n_params)
    check(self.lib.mcba_debug_gn_step(self.h, float(reg), _ptr(gn, C.c_double), _ptr(gh, C.c_double),
                                      _ptr(si, C.c_double)))
    return gn, gh, si

  def debug_chol(self, S, rhs, reg=0.0, blocked=False):
    S, rhs = _f64(S), _f64(rhs)
    out = np.empty(rhs.size)
    check(self.lib.mcba_debug_chol(self.h, int(rhs.size), _ptr(S, C.c_double), _ptr(rhs, C.c_double), float(reg),
                                   int(blocked), _ptr(out, C.c_double)))
    return out

  # --- outlier loop on the device ---------------------------------------------------------------------------------
  def error_stats(self, x, quantiles=(0, 0.25, 0.5, 0.75, 1), inliers_only=False):
    """error_stats(Calibration.reprojection_error) of the reference (calibration.py:304-310) computed on the device:
    returns (mse, rms, quantiles, n).  Quantiles follow numpy's default 'linear' method exactly: virtual index
    (n-1) q, floor / ceil order statistics from the device radix select, numpy's _lerp on the host."""
    x = self._x(x)
    q = np.asarray(quantiles, dtype=np.float64).reshape(-1)
    n = C.c_int64()
    ssq = C.c_double()
    # the ranks depend on n: known on the host for a single-GPU handle; a sharded handle needs a first pass for it
    check(self.lib.mcba_error_count(self.h, int(inliers_only), C.byref(n)))
    if n.value < 0 or q.size == 0:
      check(self.lib.mcba_error_stats(self.h, _ptr(x, C.c_double), int(inliers_only), 0, None, None, C.byref(n),
                                      C.byref(ssq)))
    nv = n.value
    if nv == 0:   # the reference substitutes a single zero (calibration.py:306-307)
      return 0.0, 0.0, np.zeros(q.shape), 1
    if q.size == 0:
      mse = ssq.value / nv
      return mse, float(np.sqrt(mse)), q, nv
    virt = (nv - 1) * q
    lo = np.floor(virt).astype(np.int64)
    hi = np.minimum(lo + 1, nv - 1)
    lo = np.clip(lo, 0, nv - 1)
    gamma = virt - np.floor(virt)
    ranks = np.ascontiguousarray(np.stack([lo, hi], axis=1).ravel())
    vals = np.empty(ranks.size)
    check(self.lib.mcba_error_stats(self.h, _ptr(x, C.c_double), int(inliers_only), int(ranks.size),
                                    ranks.ctypes.data_as(C.POINTER(C.c_int64)), _ptr(vals, C.c_double), C.byref(n),
                                    C.byref(ssq)))
    a, b = vals[0::2], vals[1::2]
    diff = b - a
    out = a + diff * gamma                                   # numpy.lib._function_base_impl._lerp
    out = np.where(gamma >= 0.5, b - diff * (1 - gamma), out)
    mse = ssq.value / nv
    return mse, float(np.sqrt(mse)), out, nv

  def reject_outliers(self, x, threshold):
    """inliers = (err < threshold) & valid on the device; returns (n_inliers, n_valid)."""
    x = self._x(x)
    self._jac_pattern = None
    ni, nvv = C.c_int64(), C.c_int64()
    check(self.lib.mcba_reject_outliers(self.h, _ptr(x, C.c_double), float(threshold), C.byref(ni), C.byref(nvv)))
    return ni.value, nvv.value

  def adjust_outliers(self, x0, num_adjustments=3, outlier=None, scale=None, tolerance=1e-4, f_scale=1.0, max_iterations=100,
                      loss='linear', xtol=1e-8, gtol=1e-8, tr_solver='exact'):
    """Calibration.adjust_outliers (calibration.py:254-268) in ONE library call (mcba_adjust_outliers): `num_adjustments` rounds of
    {report, f_scale from `scale` = (quantile, factor) or None, rejection at `outlier` = (quantile, factor) or None, solve} and
    the final report.  Returns (x, rounds, inlier mask); rounds[i] = namespace(rms, rms_inliers, n, n_inliers, quantiles,
    f_scale, threshold, n_kept, n_valid, solve result); the last entry is the final report (no solve)."""
    x = self._x(x0).copy()
    self._jac_pattern = None
    opt = make_options(tolerance, f_scale, max_iterations, loss, xtol, gtol, 2, tr_solver)
    rounds = (RoundReport * (num_adjustments + 1))()
    mask = np.empty(self.shape, dtype=np.bool_)
    oq, of = outlier if outlier is not None else (0.0, -1.0)
    sq, sf = scale if scale is not None else (0.0, -1.0)
    check(self.lib.mcba_adjust_outliers(self.h, _ptr(x, C.c_double), C.byref(opt), int(num_adjustments), float(oq), float(of),
                                        float(sq), float(sf), rounds, _ptr(mask.view(np.uint8), C.c_uint8)))
    out = []
    for i, r in enumerate(rounds):
      res = r.solve
      solve = None if i == num_adjustments else SimpleNamespace(
        cost=res.cost, initial_cost=res.initial_cost, optimality=res.optimality, nfev=res.nfev, njev=res.njev, status=res.status,
        iterations=res.iterations, message=STATUS_MESSAGES.get(res.status, ""), solve_seconds=res.solve_seconds)
      out.append(SimpleNamespace(rms=r.rms_all, rms_inliers=r.rms_inliers, n=int(r.n_all), n_inliers=int(r.n_inliers),
                                 quantiles=np.array(list(r.quantiles)), f_scale=r.f_scale, threshold=r.threshold,
                                 n_kept=int(r.n_kept), n_valid=int(r.n_valid), solve=solve))
    return x, out, mask

  def get_inliers(self):
    m = np.empty(self.shape, dtype=np.bool_)      # the device writes 0 / 1 bytes: a bool array without a second copy
    check(self.lib.mcba_get_inliers(self.h, _ptr(m.view(np.uint8), C.c_uint8)))
    return m

  def linearize_profile(self, x, with_tmat=False):
    x = self._x(x)
    C_, F, B, P = self.shape
    nv = F * C_ * B
    out = np.zeros((nv + (nv + 7) // 8, 8), dtype=np.int64)   # (+ one row per k_tmat workgroup, experimental builds)
    check(self.lib.mcba_debug_linearize_profile(self.h, _ptr(x, C.c_double), out.ctypes.data_as(C.POINTER(C.c_longlong))))
    return (out[:nv], out[nv:]) if with_tmat else out[:nv]

  # --- solve --------------------------------------------------------------------------------------------------
  def set_log(self, fn):
    """fn(iteration, nfev, cost, cost_reduction, step_norm, optimality) or None."""
    if fn is None:
      cb = C.cast(None, _lib.LOG_FN)
    else:
      cb = _lib.LOG_FN(lambda ctx, it, nfev, cost, red, step, opt: fn(it, nfev, cost, red, step, opt))
    check(self.lib.mcba_set_log(self.h, cb, None))
    self._log_cb = cb        # replaces (and releases) the previous thunk

  def set_shard_root(self, is_root):
    check(self.lib.mcba_set_shard_root(self.h, 1 if is_root else 0))

  def set_shard_rank(self, rank, world):
    """rank of this handle among the `world` handles of one frame-sharded problem (rank 0 = root)."""
    check(self.lib.mcba_set_shard_rank(self.h, int(rank), int(world)))

  def allreduce_stats(self, reset=True, cap=4096):
    """(calls, doubles, sizes): collectives this (frame-sharded) handle issued since the last reset; `sizes` lists the
    element counts in issue order (negative = max reduction)."""
    calls, doubles, n = C.c_int64(), C.c_int64(), C.c_int32()
    sizes = (C.c_int64 * cap)()
    check(self.lib.mcba_allreduce_stats(self.h, 1 if reset else 0, C.byref(calls), C.byref(doubles), sizes, cap, C.byref(n)))
    return calls.value, doubles.value, [int(sizes[i]) for i in range(n.value)]

  def set_allreduce(self, fn):
    """fn(device_ptr:int, count:int, op:int, stream:int) -> int (0 = ok); see multical_amd.distributed."""
    if fn is None:
      cb = C.cast(None, _lib.ALLREDUCE_FN)
    else:
      def tramp(ctx, buf, count, op, stream):
        try:
          return int(fn(buf or 0, count, op, stream or 0) or 0)
        except Exception as e:   # never let an exception cross the C boundary
          import traceback
          traceback.print_exc()
          return 1
      cb = _lib.ALLREDUCE_FN(tramp)
    check(self.lib.mcba_set_allreduce(self.h, cb, None))
    self._allreduce_cb = cb

  # --- native RCCL all-reduce (multical_amd.distributed.init_native_allreduce drives these) ----------------------
  @staticmethod
  def rccl_unique_id():
    """128 opaque bytes from ncclGetUniqueId (call on ONE rank, hand to all)."""
    buf = (C.c_uint8 * 128)()
    check(_lib.load().mcba_rccl_unique_id(buf))
    return bytes(buf)

  def rccl_init(self, unique_id, rank, world):
    """Collective: builds this handle's RCCL communicator; afterwards every reduction is an in-place ncclAllReduce on
    the handle's stream."""
    assert len(unique_id) == 128
    buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
    check(self.lib.mcba_rccl_init(self.h, buf, int(rank), int(world)))

  @staticmethod
  def rccl_version():
    """ncclGetVersion of the librccl the native path binds (0: not available)."""
    v = C.c_int32()
    check(_lib.load().mcba_rccl_version(C.byref(v)))
    return int(v.value)

  def rccl_shutdown(self):
    check(self.lib.mcba_rccl_shutdown(self.h))

  def solve(self, x0, tolerance=1e-4, f_scale=1.0, max_iterations=100, loss='linear', xtol=1e-8, gtol=1e-8,
            verbose=2, tr_solver='exact'):
    """mcba_solve.  tr_solver = 'exact': regularised Gauss-Newton steps from the exact normal equations (Schur + Cholesky):
    fast, ends at the converged optimum.  tr_solver = 'lsmr': scipy's own trust-region step gn_h = lsmr(J_h, f, damp) with the
    Jacobian products on the device: the reference's trajectory and end point."""
    x = self._x(x0).copy()
    opt = make_options(tolerance, f_scale, max_iterations, loss, xtol, gtol, verbose, tr_solver)
    res = Result()
    check(self.lib.mcba_solve(self.h, _ptr(x, C.c_double), C.byref(opt), C.byref(res)))
    return SimpleNamespace(x=x, cost=res.cost, initial_cost=res.initial_cost, optimality=res.optimality,
                           nfev=res.nfev, njev=res.njev, status=res.status, iterations=res.iterations,
                           message=STATUS_MESSAGES.get(res.status, ""), solve_seconds=res.solve_seconds,
                           linearize_seconds=res.linearize_seconds, success=res.status > 0)

  def lsmr_products(self, x, v=None, u=None):
    """(J(x) v, J(x)^T u) through the matrix-free kernels the lsmr mode iterates with (mcba_debug_lsmr_products; unscaled columns,
    linear loss, reference residual order) -- test hook: compare with `jacobian(x) @ v` / `jacobian(x).T @ u`."""
    x = self._x(x)
    jv = jtu = None
    pv = pu = pjv = pjtu = None
    if v is not None:
      v = self._x(v)
      jv = np.empty(self.n_residuals)
      pv, pjv = _ptr(v, C.c_double), _ptr(jv, C.c_double)
    if u is not None:
      u = _f64(u)
      assert u.shape == (self.n_residuals,)
      jtu = np.empty(self.n_params)
      pu, pjtu = _ptr(u, C.c_double), _ptr(jtu, C.c_double)
    check(self.lib.mcba_debug_lsmr_products(self.h, _ptr(x, C.c_double), pv, pu, pjv, pjtu))
    return jv, jtu

  def lsmr_fused_products(self, x, v):
    """(J(x) v, J(x)^T J(x) v) through k_lsmr_fused2 + k_lsmr_gather3, the two kernels of the default solver's LSMR iteration
    (mcba_debug_lsmr_fused_products) -- test hook."""
    x, v = self._x(x), self._x(v)
    jv, w = np.empty(self.n_residuals), np.empty(self.n_params)
    check(self.lib.mcba_debug_lsmr_fused_products(self.h, _ptr(x, C.c_double), _ptr(v, C.c_double), _ptr(jv, C.c_double), _ptr(w, C.c_double)))
    return jv, w

  def set_lsmr_fused(self, mode):
    """A/B switch of the LSMR iteration: -1 (default) = automatic (3 on static / hand-eye rigs, 2 with rolling shutter or boards=True),
    3 = two launches with the per-observation state cached, 2 = two launches (k_lsmr_fused2 / k_lsmr_gather3), 1 = three (k_lsmr_fused /
    k_lsmr_gather2 / k_lsmr_update2), 0 = the six-launch form of round 4."""
    check(self.lib.mcba_debug_set_lsmr_fused(self.h, int(mode)))

  def lsmr_solve(self, x, damp, scale=None, maxiter=0, loss='linear', f_scale=1.0):
    """ONE call of the device's LSMR solve (mcba_debug_lsmr_solve) on the linearisation at x: returns (gn_h, scale, info) with
    info = dict(istop, itn, normr, normar, normA, condA, normx, normb) -- scipy's `lsmr` return tuple -- test hook.  scale = None:
    scipy's Jacobian scaling of a first iterate; else the column scaling d to use (J_h = J diag(d)); maxiter > 0: scipy's `maxiter`."""
    x = self._x(x)
    scale_in = None if scale is None else self._x(scale)
    gn, scale, out = np.empty(self.n_params), np.empty(self.n_params), np.empty(8)
    opt = make_options(loss=loss, f_scale=f_scale)
    check(self.lib.mcba_debug_lsmr_solve(self.h, _ptr(x, C.c_double), C.byref(opt), C.c_double(damp),
                                         None if scale_in is None else _ptr(scale_in, C.c_double), C.c_int32(int(maxiter)), _ptr(gn, C.c_double),
                                         _ptr(scale, C.c_double), _ptr(out, C.c_double)))
    keys = ("istop", "itn", "normr", "normar", "normA", "condA", "normx", "normb")
    info = {k: (int(v) if k in ("istop", "itn") else float(v)) for k, v in zip(keys, out)}
    return gn, scale, info

  def set_lsmr_trace(self, scalars=True):
    """ask the next lsmr-mode solves for normr .. normx of every LSMR call (lsmr_trace) -- test / profiling hook"""
    check(self.lib.mcba_debug_set_lsmr_trace(self.h, 1 if scalars else 0))

  def lsmr_trace(self):
    """The LSMR calls of the last `solve(tr_solver='lsmr')`: a list of dicts (iteration, damp, Delta, istop, itn, normr, normar,
    normA, condA, normx); the last five are NaN unless set_lsmr_trace() preceded the solve."""
    n = C.c_int32()
    check(self.lib.mcba_debug_lsmr_trace(self.h, 0, None, C.byref(n)))
    rows = np.zeros((max(n.value, 1), 10))
    check(self.lib.mcba_debug_lsmr_trace(self.h, n.value, _ptr(rows, C.c_double), C.byref(n)))
    keys = ("iteration", "damp", "Delta", "istop", "itn", "normr", "normar", "normA", "condA", "normx")
    return [{k: (int(v) if k in ("iteration", "istop", "itn") else float(v)) for k, v in zip(keys, r)} for r in rows[:n.value]]

  def set_lsmr_masks_form(self, on=True):
    """A/B switch: the LSMR product kernel reads masks / observations / board points from the frame-major tables (its form with
    boards=True) instead of the compacted observation tables -- test / profiling hook"""
    check(self.lib.mcba_debug_set_lsmr_masks_form(self.h, 1 if on else 0))

  def set_allreduce_trace(self, cap):
    """how many collective sizes allreduce_stats() keeps (default 4096) -- the collective-sequence tests record whole lsmr solves"""
    check(self.lib.mcba_debug_set_allreduce_trace(self.h, int(cap)))

  def set_lsmr_grid(self, grid):
    """persistent workgroups of the LSMR product kernels (default 2048): only the summation order changes -- experiment hook"""
    check(self.lib.mcba_debug_set_lsmr_grid(self.h, int(grid)))

  def lsmr_iterations(self):
    """LSMR iterations of the last `solve(tr_solver='lsmr')` on this handle."""
    n = C.c_int64()
    check(self.lib.mcba_debug_lsmr_info(self.h, C.byref(n)))
    return int(n.value)

  def solve_scipy(self, x0, tolerance=1e-4, f_scale=1.0, max_iterations=100, loss='linear', verbose=2):
    """The reference's OWN solver call (optimization/calibration.py:209-210) with the device functions plugged in:

        scipy.optimize.least_squares(evaluate, x0, jac_sparsity=S, verbose=2, x_scale='jac', f_scale=f_scale,
                                     ftol=tolerance, max_nfev=max_iterations, method='trf', loss=loss)

    `evaluate` = mcba_residuals, and `jac_sparsity=S` (finite differences over S) is replaced by `jac` = mcba_jacobian, the
    analytic Jacobian in the very same pattern S.  Everything else -- TRF, LSMR steps, `x_scale='jac'`, the termination tests,
    the verbose table on stdout -- is scipy's, so the trajectory and the END POINT are the reference's: within 1e-6 px of the
    reference's final RMS with identical `nfev` / `status` wherever the reference's own end point is reproducible to that level
    (profiles/parity_table.md, tests/test_gpu_protocol.py).  The price is scipy's host-side LSMR (74 % of the
    reference's wall time); `solve` (mcba_solve) is the fast route and ends at the converged optimum instead."""
    from scipy.optimize import least_squares
    t0 = time.perf_counter()
    first = {}

    def fun(x):
      r = self.residuals(x)
      first.setdefault("r0", r)
      return r

    res = least_squares(fun, self._x(x0), jac=self.jacobian, verbose=verbose, x_scale='jac', f_scale=f_scale, ftol=tolerance,
                        max_nfev=max_iterations, method='trf', loss=loss)
    from scipy.optimize._lsq.least_squares import construct_loss_function
    r0 = first["r0"]
    if loss == 'linear':
      c0 = 0.5 * float(r0 @ r0)
    else:
      c0 = float(construct_loss_function(r0.size, loss, f_scale)(r0, cost_only=True))
    return SimpleNamespace(x=res.x, cost=float(res.cost), initial_cost=c0, optimality=float(res.optimality), nfev=int(res.nfev),
                           njev=int(res.njev), status=int(res.status), iterations=int(res.njev) - 1,
                           message=res.message, solve_seconds=time.perf_counter() - t0, linearize_seconds=0.0,
                           success=bool(res.success))

  # --- measurement --------------------------------------------------------------------------------------------
  def time_linearize(self, x, repeats=20, loss='linear', f_scale=1.0):
    x = self._x(x)
    opt = make_options(loss=loss, f_scale=f_scale)
    ms = C.c_double()
    check(self.lib.mcba_time_linearize(self.h, _ptr(x, C.c_double), C.byref(opt), int(repeats), C.byref(ms)))
    return ms.value

  def time_lsmr_iteration(self, x, repeats=50):
    """(ms of k_lsmr_fused2, ms of k_lsmr_gather3): the two launches of one LSMR iteration of the default solver, HIP events."""
    x = self._x(x)
    ms = np.zeros(2)
    check(self.lib.mcba_time_lsmr_iteration(self.h, _ptr(x, C.c_double), repeats, _ptr(ms, C.c_double)))
    return float(ms[0]), float(ms[1])

  def time_residuals(self, x, repeats=20):
    x = self._x(x)
    ms = C.c_double()
    check(self.lib.mcba_time_residuals(self.h, _ptr(x, C.c_double), int(repeats), C.byref(ms)))
    return ms.value


def release_cached_memory():
  """Return the device buffers, pinned buffers and streams that closed handles left in the library's cache to the runtime."""
  check(_lib.load().mcba_release_cached_memory())


def mfma_probe(V):
  lib = _lib.load()
  V = _f64(V)
  assert V.shape == (4, 32)
  out = np.empty((16, 16))
  check(lib.mcba_debug_mfma_probe(_ptr(V, C.c_double), _ptr(out, C.c_double)))
  return out
