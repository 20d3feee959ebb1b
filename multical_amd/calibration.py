"""Host mirror of multical.optimization.calibration.Calibration (optimization/calibration.py:43-310) whose numerical
work runs on the MI355X back-end.

Same constructor, same immutable-copy style, same parameter ordering and the same public methods as the reference
class, so that callers written against the reference (Workspace.calibrate, workspace.py:228-247) work unchanged:

    bundle_adjust(tolerance, f_scale, max_iterations, loss)   calibration.py:199-212   -> mcba_solve
    reprojected / reprojection_error / reprojection_inliers    calibration.py:124-141   -> mcba_project / _error
    reject_outliers / adjust_outliers / report                 calibration.py:234-300
    enable / copy / params / with_params / param_vec           calibration.py:144-171,214-232

Objects hold no device state: like the reference only the 8 constructor fields are pickled (calibration.py:222-226);
device handles live in a small module-level cache keyed on the observation table.
"""
import contextlib
import logging
import os
from functools import cached_property

import numpy as np

from . import parameters
from .backend import Handle, lower
from .structs import Struct, Table, struct, choose, subset

logger = logging.getLogger("calibration")   # same logger name as multical/io/logging.py:11


def info(msg):
  logger.info(msg)


class LogWriter(object):
  """io/logging.py:53-68: file-like object that forwards what scipy prints with verbose=2 to the "calibration" logger."""

  def __init__(self, level=logging.INFO, ignore_newline=True):
    self.level, self.ignore_newline = level, ignore_newline

  def write(self, message, *args, **kwargs):
    if message != '\n' or not self.ignore_newline:
      logger._log(self.level, message, args, **kwargs)

  def flush(self):
    pass

  @staticmethod
  def info(ignore_newline=True):
    return LogWriter(logging.INFO, ignore_newline)


# Which solver `Calibration.bundle_adjust` uses (all run residuals and Jacobian on the MI355X):
#   "lsmr"    DEFAULT.  mcba_solve with scipy's OWN trust-region step on the device (tr_solver = lsmr): gn_h = lsmr(J_h, f, damp) with
#             the two Jacobian products as HIP kernels and scipy's driver restated line by line -- the reference's trajectory and
#             END POINT (within max(1e-6 px, the reference's own reproducibility), identical nfev / status), without the
#             host-side LSMR (an hour of one core at the north-star rig).
#   "native"  mcba_solve: trust-region driver with exact Schur / Cholesky steps, everything on the device, ~100 x faster than
#             "lsmr".  Ends at the CONVERGED optimum (at or below the reference's cost) -- NOT at the reference's end point: on
#             weakly determined rigs the principal point ends tens of pixels away from the reference's (profiles/parity_table.md).
#   "scipy"   the reference's own call scipy.optimize.least_squares(method='trf', x_scale='jac', ...) on mcba_residuals +
#             mcba_jacobian (Handle.solve_scipy): the reference's end point as well, at the price of scipy's host-side LSMR.
SOLVERS = ("lsmr", "native", "scipy")
_default_solver = [os.environ.get("MULTICAL_AMD_SOLVER", "lsmr").lower()]


def set_solver(name):
  """Select the solver of every following `bundle_adjust` / `adjust_outliers` / `Workspace.calibrate`; returns the previous one."""
  if name not in SOLVERS:
    raise ValueError(f"unknown solver {name!r}, options are {SOLVERS}")
  prev, _default_solver[0] = _default_solver[0], name
  if prev != name:
    import logging
    logging.getLogger("calibration").info("multical_amd: bundle adjustment solver '%s' -> '%s'", prev, name)
  return prev


def get_solver():
  return _default_solver[0]


def fused_outlier_loop_off():
  """MULTICAL_AMD_FUSED_LOOP=0: adjust_outliers as a Python loop over report / reject_outliers / bundle_adjust on new Calibration
  objects (the reference's structure, step by step) instead of one mcba_adjust_outliers call."""
  return os.environ.get("MULTICAL_AMD_FUSED_LOOP", "1") == "0"


default_optimize = struct(cameras=False, boards=False, camera_poses=True, board_poses=True, motion=True)


def select_threshold(quantile=0.75, factor=5.0):
  """calibration.py:37-40.  The returned callable still works on an error array like the reference's; it is tagged with
  its parameters so that adjust_outliers can evaluate the quantile on the device without downloading the errors."""
  def f(reprojection_error):
    return np.quantile(reprojection_error, quantile) * factor
  f.quantile, f.factor = quantile, factor
  return f


def error_stats(errors):
  """calibration.py:304-310."""
  if len(errors) == 0:
    errors = np.zeros((1, 1), np.float32)
  mse = np.square(errors).mean()
  quantiles = np.array([np.quantile(errors, n) for n in [0, 0.25, 0.5, 0.75, 1]])
  return struct(mse=mse, rms=np.sqrt(mse), quantiles=quantiles, n=errors.size)


# ---------------------------------------------------------------------------------------------------------------
# device-handle cache (never pickled, never part of a Calibration)
# ---------------------------------------------------------------------------------------------------------------
class _HandleCache(object):
  def __init__(self, capacity=2):
    self.capacity = capacity
    # entries: dict(key, points, valid, x_full, handle, mask).  `points` / `valid` are the ORIGINAL arrays of the point
    # table the handle was built from, held strongly: identity (`is`) is the fast test, and holding the reference keeps
    # an id from being recycled by a later table of the same shape (float32 detections are widened into a copy, so the
    # handle itself would not keep the original alive).  A table with equal content but another identity (e.g.
    # `point_table._extend(valid=...)` keeps `points`, replaces `valid`) is compared by value.  The inlier mask OBJECT is
    # a token compared with `is` first: hashing 2.6 MB of mask bytes on every lookup cost 0.6 ms x 16 lookups per
    # Workspace.calibrate.
    self.entries = []

  @staticmethod
  def _same_array(a, b):
    return a is b or (a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b))

  def get(self, calib):
    prob = getattr(calib, "_mcba_problem", None)     # Calibration objects are immutable: lower once per object
    if prob is None:
      prob = lower(calib)
      try:
        calib._mcba_problem = prob                   # (not part of __getstate__: never pickled)
      except AttributeError:
        pass
    key = (prob.shape, prob.optimize, prob.motion, prob.camera_model, prob.n_dist,
           None if prob.camera_n_dist is None else prob.camera_n_dist.tobytes(),
           None if prob.camera_fisheye is None else prob.camera_fisheye.tobytes(), prob.fix_aspect.tobytes(),
           prob.camera_valid.tobytes(), prob.frame_valid.tobytes(), prob.board_valid.tobytes(),
           prob.board_sizes.tobytes(), prob.image_heights.tobytes())
    points, valid = calib.point_table.points, calib.point_table.valid
    for i, e in enumerate(self.entries):
      h = e["handle"]
      if e["key"] == key and h.h and self._same_array(e["points"], points) and self._same_array(e["valid"], valid) \
          and np.array_equal(self._constants(calib, prob, e["x_full"]), self._constants(calib, prob, prob.x_full)) \
          and (prob.base_wrt_gripper is None or np.array_equal(prob.base_wrt_gripper, h.problem.base_wrt_gripper)):
        mask, tok = calib.inlier_mask, e["mask"]
        if mask is not tok:
          if mask is None or tok is None or not np.array_equal(mask, tok):
            h.set_inliers(mask)
          e["mask"] = mask
        return h, prob
    h = Handle(prob)
    self.entries.append(dict(key=key, points=points, valid=valid, x_full=prob.x_full, handle=h, mask=calib.inlier_mask))
    while len(self.entries) > self.capacity:
      self.entries.pop(0)["handle"].close()
    return h, prob

  @staticmethod
  def _constants(calib, prob, x_full):
    """values of the DISABLED blocks (the only part of x_full the device keeps)."""
    out, pos = [], 0
    for k, n in zip(parameters_order(), prob.block_sizes):
      if calib.optimize[k] is not True:
        out.append(x_full[pos:pos + n])
      pos += n
    return np.concatenate(out) if out else np.zeros(0)

  def note_inliers(self, handle, mask):
    """the device already holds `mask` (set by reject_outliers): remember its token so it is not uploaded again."""
    for e in self.entries:
      if e["handle"] is handle:
        e["mask"] = mask

  def invalidate_inliers(self, handle):
    """a library call that rewrites the device mask (mcba_reject_outliers, mcba_adjust_outliers) FAILED part-way: the
    device mask is unknown, so the next lookup must upload its own (a fresh token never compares equal)."""
    self.note_inliers(handle, _UNKNOWN_MASK)

  def clear(self):
    for e in self.entries:
      e["handle"].close()
    self.entries = []


def parameters_order():
  return ["camera_poses", "board_poses", "motion", "cameras", "boards"]


_UNKNOWN_MASK = np.zeros(0, dtype=np.uint8)   # token of "the device mask is in an unknown state"
handle_cache = _HandleCache()


class Calibration(parameters.Parameters):
  def __init__(self, cameras, boards, point_table, camera_poses, board_poses, motion, inlier_mask=None,
               optimize=default_optimize):
    self.cameras = cameras
    self.boards = boards
    self.point_table = point_table
    self.camera_poses = camera_poses
    self.board_poses = board_poses
    self.motion = motion
    self.optimize = optimize
    self.inlier_mask = inlier_mask

    assert len(self.cameras) == self.size.cameras
    assert camera_poses.size == self.size.cameras
    assert board_poses.size == self.size.boards

  # --- shapes and masks (calibration.py:63-81) --------------------------------------------------------------
  @cached_property
  def size(self):
    cameras, rig_poses, boards, points = self.point_table.valid.shape
    return struct(cameras=cameras, rig_poses=rig_poses, boards=boards, points=points)

  @cached_property
  def valid(self):
    valid = (np.expand_dims(self.camera_poses.valid, [1, 2]) & np.expand_dims(self.motion.valid, [0, 2]) &
             np.expand_dims(self.board_poses.valid, [0, 1]))
    return self.point_table.valid & np.expand_dims(valid, valid.ndim)

  @cached_property
  def inliers(self):
    return choose(self.inlier_mask, self.valid)

  # --- parameters (calibration.py:144-171) ------------------------------------------------------------------
  @cached_property
  def param_objects(self):
    return struct(camera_poses=self.camera_poses, board_poses=self.board_poses, motion=self.motion,
                  cameras=self.cameras, boards=self.boards)

  @cached_property
  def params(self):
    all_params = self.param_objects._map(lambda p: p.param_vec)
    return all_params._filterWithKey(lambda k: self.optimize[k] is True)

  def with_params(self, params):
    updated = {k: self.param_objects[k].with_param_vec(param_vec) for k, param_vec in params.items()}
    return self.copy(**updated)

  def enable(self, **flags):
    for k in flags.keys():
      assert k in self.optimize, f"unknown option {k}, options are {list(self.optimize.keys())}"
    return self.copy(optimize=self.optimize._extend(**flags))

  def __getstate__(self):
    attrs = ['cameras', 'boards', 'point_table', 'camera_poses', 'board_poses', 'motion', 'inlier_mask', 'optimize']
    return subset(self.__dict__, attrs)

  def __setstate__(self, d):
    self.__dict__.update(d)

  def copy(self, **k):
    d = self.__getstate__()
    d.update(k)
    return Calibration(**d)

  def with_master(self, camera):
    """calibration.py:99-104: re-gauge so that `camera` (name or index) sits at the origin."""
    if isinstance(camera, str):
      camera = self.camera_poses.names.index(camera)
    return self.transform_views(self.camera_poses.poses[int(camera)])

  def transform_views(self, t):
    """calibration.py:107-112: cameras by t^-1, time poses by t -- the reprojections do not change."""
    return self.copy(camera_poses=self.camera_poses.post_transform(np.linalg.inv(t)),
                     motion=self.motion.pre_transform(t))

  # --- device evaluation ------------------------------------------------------------------------------------
  def _handle(self):
    return handle_cache.get(self)[0]

  @cached_property
  def reprojected(self):
    """calibration.py:124-130 (rolling-shutter scan time from the measured points)."""
    h = self._handle()
    points = h.project(self.param_vec)
    _, valid = h.reprojection_error(self.param_vec)
    # reprojected.valid of the reference = pose validity & board-point validity (no detection mask)
    C, F, B, P = self.point_table.valid.shape
    pose_valid = (np.expand_dims(self.camera_poses.valid, [1, 2, 3]) & np.expand_dims(self.motion.valid, [0, 2, 3]) &
                  np.expand_dims(self.board_poses.valid, [0, 1, 3]))
    sizes = np.array([b.num_points for b in self.boards])
    board_valid = np.arange(P)[None, :] < sizes[:, None]
    return Table.create(points=points, valid=pose_valid & board_valid[None, None])

  @cached_property
  def projected(self):
    """calibration.py:113-119: projected points to each image WITHOUT the measured points (rolling shutter: the scan
    time is iterated from the projected row, RollingFrames.max_iterations passes) -- the table the GUI draws."""
    h = self._handle()
    points = h.project_model(self.param_vec, getattr(self.motion, "max_iterations", 4))
    return Table.create(points=points, valid=self.reprojected.valid)

  def _errors(self):
    """tables.reprojection_error (tables.py:244-249) of (reprojected, point_table) on the device."""
    return self._handle().reprojection_error(self.param_vec)

  @cached_property
  def reprojection_error(self):
    err, mask = self._errors()
    return err[mask]

  @cached_property
  def reprojection_inliers(self):
    err, mask = self._errors()
    # calibration.py:138-141: point_table with valid := inliers, masked with reprojected.valid
    return err[self.reprojected.valid & choose(self.inliers, self.valid)]

  def residuals(self, param_vec=None):
    """`evaluate` of calibration.py:204-206."""
    return self._handle().residuals(self.param_vec if param_vec is None else param_vec)

  def jacobian(self, param_vec=None):
    return self._handle().jacobian(self.param_vec if param_vec is None else param_vec)

  # --- solve (calibration.py:199-212) -----------------------------------------------------------------------
  def bundle_adjust(self, tolerance=1e-4, f_scale=1.0, max_iterations=100, loss='linear', return_result=False,
                    xtol=1e-8, gtol=1e-8, solver=None):
    """Non-linear least squares on point reprojection error, solved on the GPU.

    Keeps the reference's signature and semantics.  solver = "lsmr" (default, see `set_solver`) / "native" (mcba_solve): the iteration
    table scipy prints with verbose=2 is emitted in the same format through the "calibration" logger (calibration.py:208,
    io/logging.py:53-68).  solver = "scipy": the reference's own `least_squares` call on the device residuals + analytic
    Jacobian (`Handle.solve_scipy`), scipy's own table redirected to the logger exactly as calibration.py:208 does."""
    solver = get_solver() if solver is None else solver
    if solver not in SOLVERS:
      raise ValueError(f"unknown solver {solver!r}, options are {SOLVERS}")
    h = self._handle()
    if solver == "scipy":
      with contextlib.redirect_stdout(LogWriter.info()):
        res = h.solve_scipy(self.param_vec, tolerance=tolerance, f_scale=f_scale, max_iterations=max_iterations, loss=loss,
                            verbose=2)
      out = self.with_param_vec(res.x)
      return (out, res) if return_result else out
    rows = []

    def log_row(it, nfev, cost, red, step, opt):
      red_s = " " * 15 if np.isnan(red) else f"{red:^15.2e}"
      step_s = " " * 15 if np.isnan(step) else f"{step:^15.2e}"
      rows.append(f"{it:^15}{nfev:^15}{cost:^15.4e}{red_s}{step_s}{opt:^15.2e}")

    h.set_log(log_row)
    info("{:^15}{:^15}{:^15}{:^15}{:^15}{:^15}".format("Iteration", "Total nfev", "Cost", "Cost reduction",
                                                       "Step norm", "Optimality"))
    try:
      res = h.solve(self.param_vec, tolerance=tolerance, f_scale=f_scale, max_iterations=max_iterations, loss=loss,
                    xtol=xtol, gtol=gtol, verbose=2, tr_solver="lsmr" if solver == "lsmr" else "exact")
    finally:
      h.set_log(None)        # cached handles outlive the call: do not keep the closure over `rows` installed
    for r in rows:
      info(r)
    info(res.message)
    info(f"Function evaluations {res.nfev}, initial cost {res.initial_cost:.4e}, final cost {res.cost:.4e}, "
         f"first-order optimality {res.optimality:.2e}.")
    out = self.with_param_vec(res.x)
    return (out, res) if return_result else out

  # --- outliers (calibration.py:234-268) --------------------------------------------------------------------
  def reject_outliers_quantile(self, quantile=0.95, factor=1.0):
    threshold = np.quantile(self.reprojection_error, quantile)
    return self.reject_outliers(threshold=threshold * factor)

  def _select(self, selector):
    """selector(self.reprojection_error); tagged select_threshold closures are evaluated on the device
    (exact numpy 'linear' quantile from radix-selected order statistics), anything else gets the error array."""
    if hasattr(selector, "quantile") and hasattr(selector, "factor"):
      # report() has usually just selected this very quantile (0.75 is one of its five): same object, same errors
      stats = self.__dict__.get("_overall_stats")
      if stats is not None:
        hit = [v for qq, v in zip(stats[0], stats[1].quantiles) if qq == selector.quantile]
        if hit:
          return float(hit[0]) * selector.factor
      _, _, q, _ = self._handle().error_stats(self.param_vec, quantiles=[selector.quantile])
      return float(q[0]) * selector.factor
    return selector(self.reprojection_error)

  def reject_outliers(self, threshold):
    """calibration.py:240-252, evaluated on the device; only the new mask (uint8) comes back."""
    h = self._handle()
    try:
      n_in, n_valid = h.reject_outliers(self.param_vec, threshold)
      inliers = h.get_inliers()
    except BaseException:
      handle_cache.invalidate_inliers(h)
      raise
    num_outliers = n_valid - n_in
    inlier_percent = 100.0 * n_in / max(n_valid, 1)
    info(f"Rejecting {num_outliers} outliers with error > {threshold:.2f} pixels, "
         f"keeping {n_in} / {n_valid} inliers, ({inlier_percent:.2f}%)")
    out = self.copy(inlier_mask=inliers)
    handle_cache.note_inliers(h, inliers)
    return out

  def _adjust_outliers_in_one_call(self, num_adjustments, select_scale, select_outliers, kwargs):
    """The whole loop inside the library (mcba_adjust_outliers): no Calibration objects, no re-lowering and no rotation-vector <->
    matrix round trips between the rounds.  The log lines are the reference's, emitted afterwards in the reference's order."""
    h = self._handle()
    rows = []

    def log_row(it, nfev, cost, red, step, opt):
      red_s = " " * 15 if np.isnan(red) else f"{red:^15.2e}"
      step_s = " " * 15 if np.isnan(step) else f"{step:^15.2e}"
      rows.append((it, f"{it:^15}{nfev:^15}{cost:^15.4e}{red_s}{step_s}{opt:^15.2e}"))

    h.set_log(log_row)
    try:
      x, rounds, mask = h.adjust_outliers(
        self.param_vec, num_adjustments,
        outlier=None if select_outliers is None else (select_outliers.quantile, select_outliers.factor),
        scale=None if select_scale is None else (select_scale.quantile, select_scale.factor),
        tr_solver="lsmr" if get_solver() == "lsmr" else "exact", **kwargs)
    except BaseException:
      handle_cache.invalidate_inliers(h)   # a rejection may already have rewritten the device mask
      raise
    finally:
      h.set_log(None)
    tables, cur = [], None                      # the iteration rows of every solve (a solve starts with iteration 0)
    for it, line in rows:
      if it == 0:
        cur = []
        tables.append(cur)
      cur.append(line)
    has_mask = self.inlier_mask is not None

    def report_line(stage, r, with_inliers):
      n_all = max(r.n, 1)                       # (the reference substitutes a single zero for an empty error set)
      if with_inliers:
        info(f"{stage} reprojection RMS={r.rms_inliers:.3f} ({r.rms:.3f}), n={r.n_inliers} ({n_all}), quantiles={r.quantiles}")
      else:
        info(f"{stage} reprojection RMS={r.rms:.3f}, n={n_all}, quantiles={r.quantiles}")

    for i, r in enumerate(rounds[:-1]):
      report_line(f"Adjust_outliers {i}:", r, has_mask)
      if select_scale is not None:
        info(f"Auto scaling for outliers influence at {r.f_scale:.2f} pixels")
      if select_outliers is not None:
        num_outliers = r.n_valid - r.n_kept
        info(f"Rejecting {num_outliers} outliers with error > {r.threshold:.2f} pixels, "
             f"keeping {r.n_kept} / {r.n_valid} inliers, ({100.0 * r.n_kept / max(r.n_valid, 1):.2f}%)")
        has_mask = True
      info("{:^15}{:^15}{:^15}{:^15}{:^15}{:^15}".format("Iteration", "Total nfev", "Cost", "Cost reduction", "Step norm",
                                                         "Optimality"))
      for line in (tables[i] if i < len(tables) else []):
        info(line)
      info(r.solve.message)
      info(f"Function evaluations {r.solve.nfev}, initial cost {r.solve.initial_cost:.4e}, final cost {r.solve.cost:.4e}, "
           f"first-order optimality {r.solve.optimality:.2e}.")
    report_line("Adjust_outliers end:", rounds[-1], has_mask)
    out = self.with_param_vec(x)
    if select_outliers is not None and num_adjustments > 0:
      out = out.copy(inlier_mask=mask)
      handle_cache.note_inliers(h, mask)
    final = rounds[-1]
    out.__dict__["_overall_stats"] = ((0.0, 0.25, 0.5, 0.75, 1.0),
                                      struct(mse=final.rms ** 2, rms=final.rms, quantiles=final.quantiles, n=max(final.n, 1)))
    return out

  def adjust_outliers(self, num_adjustments=3, select_scale=None, select_outliers=None, **kwargs):
    info(f"Beginning adjustments ({num_adjustments}) enabled: {dict(self.optimize)}, options: {kwargs}")
    tagged = lambda f: f is None or (hasattr(f, "quantile") and hasattr(f, "factor"))
    if (get_solver() in ("native", "lsmr") and tagged(select_scale) and tagged(select_outliers) and not fused_outlier_loop_off()
        and set(kwargs) <= {"loss", "tolerance", "f_scale", "max_iterations", "xtol", "gtol"}):
      return self._adjust_outliers_in_one_call(num_adjustments, select_scale, select_outliers, kwargs)
    for i in range(num_adjustments):
      self.report(f"Adjust_outliers {i}:")
      f_scale = (None if select_scale is None else self._select(select_scale)) or 1.0
      if select_scale is not None:
        info(f"Auto scaling for outliers influence at {f_scale:.2f} pixels")
      if select_outliers is not None:
        self = self.reject_outliers(self._select(select_outliers))
      self = self.bundle_adjust(f_scale=f_scale, **kwargs)
    self.report("Adjust_outliers end:")
    return self

  def error_statistics(self, inliers_only=False, quantiles=(0, 0.25, 0.5, 0.75, 1)):
    """error_stats(self.reprojection_error / reprojection_inliers) without downloading the error table.  (A Calibration is
    immutable: the statistics over all valid points are kept on the object, like the reference's cached properties.)"""
    key = tuple(float(q) for q in quantiles)
    if not inliers_only:
      kept = self.__dict__.get("_overall_stats")
      if kept is not None and kept[0] == key:
        return kept[1]
    mse, rms, q, n = self._handle().error_stats(self.param_vec, quantiles=quantiles, inliers_only=inliers_only)
    out = struct(mse=mse, rms=rms, quantiles=q, n=n)
    if not inliers_only:
      self.__dict__["_overall_stats"] = (key, out)
    return out

  def report(self, stage=""):
    overall = self.error_statistics(False)
    # (the reference's report prints RMS and n of the inliers, never their quantiles: calibration.py:296-300)
    inliers = self.error_statistics(True, quantiles=()) if self.inlier_mask is not None else overall
    if self.inlier_mask is not None:
      info(f"{stage} reprojection RMS={inliers.rms:.3f} ({overall.rms:.3f}), "
           f"n={inliers.n} ({overall.n}), quantiles={overall.quantiles}")
    else:
      info(f"{stage} reprojection RMS={overall.rms:.3f}, n={overall.n}, quantiles={overall.quantiles}")


# ---------------------------------------------------------------------------------------------------------------
# builder from a synthetic rig (multical_amd.synthetic.make_rig)
# ---------------------------------------------------------------------------------------------------------------
def from_rig(rig, which='init'):
  from .board import Board
  from .camera import Camera, CameraFisheye
  from .motion import StaticFrames, RollingFrames, HandEye
  from .parameters import ParamList
  from .pose_set import PoseSet

  src = getattr(rig, which)
  C, F, B, P = rig.valid.shape
  cam_names = [f"cam{i}" for i in range(C)]
  board_names = [f"board{i}" for i in range(B)]
  frame_names = [f"frame{i}" for i in range(F)]
  cameras = []
  for c in src.cameras:
    if c.model == 'fisheye':
      cameras.append(CameraFisheye(c.image_size, c.intrinsic, c.dist, fix_aspect=c.fix_aspect, has_skew=c.has_skew))
    else:
      cameras.append(Camera(c.image_size, c.intrinsic, c.dist, model=c.model, fix_aspect=c.fix_aspect,
                            has_skew=c.has_skew))
  boards = [Board(p, name=n) for p, n in zip(rig.board_points, board_names)]
  kind = rig.cfg["motion"]
  if kind == 'static':
    motion = StaticFrames(Table.create(poses=src.rig, valid=rig.frame_valid), frame_names)
  elif kind == 'rolling':
    motion = RollingFrames(src.rig, src.rig_end, rig.frame_valid, frame_names)
  else:
    he = src.hand_eye
    motion = HandEye(Table.create(poses=he.base_wrt_gripper, valid=rig.frame_valid), he.world_wrt_base,
                     he.gripper_wrt_camera, frame_names)
  calib = Calibration(ParamList(cameras, cam_names), ParamList(boards, board_names),
                      Table.create(points=rig.points, valid=rig.valid),
                      PoseSet(Table.create(poses=src.camera_poses, valid=rig.camera_valid), cam_names),
                      PoseSet(Table.create(poses=src.board_poses, valid=rig.board_valid), board_names), motion)
  return calib.enable(**rig.optimize)
