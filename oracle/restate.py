"""TEST INFRASTRUCTURE -- numpy/scipy CPU restatement of the reference's bundle-adjustment hot path.

This is the ORACLE: a checker for the HIP back-end.  It is never imported by multical_amd/ (the
product fails loudly without the HIP library); only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg use it.

What it restates (every function cites the reference file:line it follows, paths relative to
/root/reference/multical/):
  * parameter packing                    optimization/parameters.py:44-50,88-106, calibration.py:146-171
  * residual closure `evaluate`          optimization/calibration.py:204-206
  * projection chain                     motion/static_frames.py:10-34, motion/rolling_frames.py:15-41,115-133,
                                         motion/hand_eye.py:43-46, tables.py:284-304,385-405, transform/matrix.py:26-29
  * camera models                        camera.py:124-171, camera_fisheye.py:113-160  (cv2 formulas: oracle/shims/cv2)
  * Jacobian sparsity                    optimization/calibration.py:173-196, parameters.py:109-150
  * solver call                          optimization/calibration.py:199-212  (REAL scipy.optimize.least_squares)
  * outlier loop / error statistics      optimization/calibration.py:37-40,134-141,234-268,290-310, tables.py:239-249

Third-party pieces: scipy (installed, used as-is, like the reference does); cv2.projectPoints and
cv2.fisheye.projectPoints (opencv-contrib-python >=4.5,<=4.7, NOT installed) are restated in
oracle/shims/cv2/__init__.py from OpenCV's published formulas.

PARITY PIN: `tests/test_oracle_vs_reference.py` runs this file against the unmodified reference executed
in-container through oracle/refload.py (residuals bit-for-bit, sparsity pattern identical, identical
scipy trajectory), and tests/golden/*.npz hold outputs of the real reference for the GPU box where
/root/reference does not exist.  The cv2 boundary itself is "parity unpinned" (no OpenCV binary here).
"""
import os
import sys
import contextlib
import io
from types import SimpleNamespace

import numpy as np
from scipy import optimize
from scipy.sparse import lil_matrix
from scipy.spatial.transform import Rotation as R

_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def _cv2():
  """The oracle's numpy restatement of the two OpenCV projection functions (oracle/shims/cv2)."""
  import importlib.util
  name = "_oracle_cv2_restatement"
  if name in sys.modules:
    return sys.modules[name]
  spec = importlib.util.spec_from_file_location(name, os.path.join(_SHIMS, "cv2", "__init__.py"))
  mod = importlib.util.module_from_spec(spec)
  sys.modules[name] = mod
  spec.loader.exec_module(mod)
  return mod


# --------------------------------------------------------------------------------------------------
# transform/rtvec.py, transform/matrix.py
# --------------------------------------------------------------------------------------------------
def rtvec_to_matrix(rtvec):
  """transform/rtvec.py:24-27 + matrix.join (matrix.py:33-39)."""
  rtvec = np.asarray(rtvec, dtype=np.float64)
  rvec, tvec = rtvec[..., 0:3], rtvec[..., 3:6]
  rot = R.from_rotvec(rvec).as_matrix()
  m = np.zeros(rtvec.shape[:-1] + (4, 4))
  m[..., :3, :3] = rot
  m[..., :3, 3] = tvec
  m[..., 3, 3] = 1.0
  return m


def rtvec_from_matrix(m):
  """transform/rtvec.py:29-32 (as_rotvec canonicalises the angle to [0, pi])."""
  rot, t = m[..., :3, :3], m[..., :3, 3]
  rvec = R.from_matrix(rot).as_rotvec()
  return np.hstack([rvec, t])


def transform_homog(t, points):
  """transform/matrix.py:21-29."""
  padding = np.ones([*points.shape[:-1], 1])
  hp = np.concatenate([points, padding], axis=points.ndim - 1)
  hp = np.expand_dims(hp, points.ndim)
  return (t @ hp).squeeze(points.ndim)[..., :3]


# --------------------------------------------------------------------------------------------------
# cameras: camera.py:124-171, camera_fisheye.py:113-160
# --------------------------------------------------------------------------------------------------
class OracleCamera(object):
  def __init__(self, image_size, intrinsic, dist, model='standard', fix_aspect=False, has_skew=False):
    self.image_size = tuple(image_size)
    self.intrinsic = np.asarray(intrinsic, dtype=np.float64)
    self.dist = np.zeros(5) if dist is None else np.asarray(dist, dtype=np.float64)
    self.model = model            # 'fisheye' selects CameraFisheye behaviour
    self.fix_aspect = fix_aspect
    self.has_skew = has_skew

  def project(self, points):
    cv2 = _cv2()
    f = cv2.fisheye.projectPoints if self.model == 'fisheye' else cv2.projectPoints
    projected, _ = f(points.reshape(-1, 1, 3), np.zeros(3), np.zeros(3), self.intrinsic, self.dist)
    return projected.reshape(*points.shape[:-1], 2)

  @property
  def param_vec(self):
    """camera.py:144-155: focal(2) | principal point(2) | skew(1) | dist."""
    f = np.array([self.intrinsic[0, 0], self.intrinsic[1, 1]])
    if self.fix_aspect:
      f = np.array([f.mean(), f.mean()])
    skew = self.intrinsic[0, 1] if self.has_skew else 0.0
    pp = np.array([self.intrinsic[0, 2], self.intrinsic[1, 2]])
    return np.concatenate([f, pp, np.array([skew]), self.dist.ravel()])

  def with_param_vec(self, p):
    """camera.py:157-171."""
    fx, fy = (p[0], p[1]) if not self.fix_aspect else (p[0], p[0])
    px, py, skew = p[2], p[3], p[4]
    K = np.array([[fx, skew, px], [0, fy, py], [0, 0, 1]])
    return OracleCamera(self.image_size, K, p[5:].reshape(self.dist.shape), self.model, self.fix_aspect, self.has_skew)


# --------------------------------------------------------------------------------------------------
# Calibration: optimization/calibration.py
# --------------------------------------------------------------------------------------------------
DEFAULT_OPTIMIZE = dict(cameras=False, boards=False, camera_poses=True, board_poses=True, motion=True)
PARAM_ORDER = ["camera_poses", "board_poses", "motion", "cameras", "boards"]   # calibration.py:146-153


class OracleCalibration(object):
  """Plain-array mirror of multical.optimization.Calibration (calibration.py:43-61).

  cameras        list[OracleCamera]
  board_points   list of [P_b,3] arrays (adjusted_points; float32 for charuco like the reference)
  points, valid  [C,F,B,P,2] / [C,F,B,P]           (point_table)
  camera_poses, camera_valid  [C,4,4] / [C]
  board_poses, board_valid    [B,4,4] / [B]
  motion         SimpleNamespace(kind='static'|'rolling'|'hand_eye', valid [F], poses | pose_start/pose_end |
                                 base_wrt_gripper/world_wrt_base/gripper_wrt_camera)
  """

  def __init__(self, cameras, board_points, points, valid, camera_poses, camera_valid,
               board_poses, board_valid, motion, inlier_mask=None, optimize=None):
    self.cameras = cameras
    self.board_points = board_points
    self.points = points
    self.point_valid = valid
    self.camera_poses = camera_poses
    self.camera_valid = camera_valid
    self.board_poses = board_poses
    self.board_valid = board_valid
    self.motion = motion
    self.inlier_mask = inlier_mask
    self.optimize = dict(DEFAULT_OPTIMIZE) if optimize is None else dict(optimize)
    C, F, B, P = valid.shape
    assert len(cameras) == C and camera_poses.shape[0] == C and board_poses.shape[0] == B

  def copy(self, **k):
    d = dict(cameras=self.cameras, board_points=self.board_points, points=self.points, valid=self.point_valid,
             camera_poses=self.camera_poses, camera_valid=self.camera_valid, board_poses=self.board_poses,
             board_valid=self.board_valid, motion=self.motion, inlier_mask=self.inlier_mask, optimize=self.optimize)
    d.update(k)
    return OracleCalibration(**d)

  def enable(self, **flags):
    for k in flags:
      assert k in self.optimize, f"unknown option {k}, options are {list(self.optimize.keys())}"
    o = dict(self.optimize)
    o.update(flags)
    return self.copy(optimize=o)

  # --- masks (calibration.py:69-81) -----------------------------------------------------------------
  @property
  def valid(self):
    v = (np.expand_dims(self.camera_valid, [1, 2]) & np.expand_dims(self.motion.valid, [0, 2]) &
         np.expand_dims(self.board_valid, [0, 1]))
    return self.point_valid & np.expand_dims(v, v.ndim)

  @property
  def inliers(self):
    return self.valid if self.inlier_mask is None else self.inlier_mask

  # --- board / world points (calibration.py:83-90, tables.py:385-405) ------------------------------
  @property
  def stacked_boards(self):
    padded = max(p.shape[0] for p in self.board_points)
    pts = np.stack([np.pad(p.astype(np.float64), [(0, padded - p.shape[0]), (0, 0)]) for p in self.board_points])
    val = np.stack([np.arange(padded) < p.shape[0] for p in self.board_points])
    return pts, val

  @property
  def world_points(self):
    pts, val = self.stacked_boards
    wp = transform_homog(np.expand_dims(self.board_poses, 1), pts)
    return wp, np.expand_dims(self.board_valid, 1) & val

  # --- projection (static_frames.py:10-34, rolling_frames.py:15-41,115-133, hand_eye.py:43-46) ------
  def _frame_tables(self):
    m = self.motion
    if m.kind == 'static':
      return [m.poses]
    if m.kind == 'rolling':
      return [m.pose_start, m.pose_end]
    if m.kind == 'hand_eye':
      return [(m.gripper_wrt_camera @ m.base_wrt_gripper) @ m.world_wrt_base]
    raise ValueError(m.kind)

  def _transform(self, frame_poses, world_points):
    # tables.expand_views: T[c,f] = cam[c] @ rig[f]; transform_points over expanded dims (2,3)/(0,1)
    view = np.expand_dims(self.camera_poses, 1) @ np.expand_dims(frame_poses, 0)          # [C,F,4,4]
    return transform_homog(np.expand_dims(view, (2, 3)), np.expand_dims(world_points, (0, 1)))  # [C,F,B,P,3]

  def _project_cameras(self, local_points):
    return np.stack([cam.project(p) for cam, p in zip(self.cameras, local_points)])

  def reprojected(self):
    """calibration.py:124-130: projection with rolling-shutter scan time from the OBSERVED points."""
    wp, wvalid = self.world_points
    view_valid = np.expand_dims(self.camera_valid, 1) & np.expand_dims(self.motion.valid, 0)        # [C,F]
    valid = np.expand_dims(view_valid, (2, 3)) & np.expand_dims(wvalid, (0, 1))
    tabs = self._frame_tables()
    if self.motion.kind == 'rolling':
      heights = np.array([cam.image_size[1] for cam in self.cameras])
      times = self.points[..., 1] / np.expand_dims(heights, (1, 2, 3))                                # rolling_frames.py:15-19
      start, end = self._transform(tabs[0], wp), self._transform(tabs[1], wp)
      t = np.expand_dims(times, times.ndim)
      local = start * (1 - t) + end * t                                                               # interpolate.py:6-8
    else:
      local = self._transform(tabs[0], wp)
    return self._project_cameras(local), valid

  def projected(self, max_iterations=4):
    """calibration.py:113-119 -> motion.project(cameras, camera_poses, world_points) WITHOUT estimates: what the GUI /
    interface/view_table.py:43-52 consume.  Static / hand-eye: the plain projection.  Rolling shutter
    (motion/rolling_frames.py:115-133): scan time 0.5 for the first pass, then `max_iterations` fixed-point passes with the
    scan time taken from the PROJECTED row of the previous pass."""
    wp, wvalid = self.world_points
    view_valid = np.expand_dims(self.camera_valid, 1) & np.expand_dims(self.motion.valid, 0)
    valid = np.expand_dims(view_valid, (2, 3)) & np.expand_dims(wvalid, (0, 1))
    tabs = self._frame_tables()
    if self.motion.kind != 'rolling':
      return self._project_cameras(self._transform(tabs[0], wp)), valid
    heights = np.array([cam.image_size[1] for cam in self.cameras])
    start, end = self._transform(tabs[0], wp), self._transform(tabs[1], wp)

    def project_at(times):
      t = np.expand_dims(times, times.ndim)
      return self._project_cameras(start * (1 - t) + end * t)

    points = project_at(np.full(valid.shape, 0.5))                                 # rolling_frames.py:119-120
    for _ in range(max_iterations):                                                # :128-131
      points = project_at(points[..., 1] / np.expand_dims(heights, (1, 2, 3)))     # rolling_times(cameras, points)
    return points, valid

  # --- parameters (calibration.py:146-171, parameters.py:44-50,88-106) -----------------------------
  def _blocks(self):
    m = self.motion
    if m.kind == 'static':
      motion = [rtvec_from_matrix(m.poses).ravel()]
    elif m.kind == 'rolling':
      motion = [rtvec_from_matrix(m.pose_start).ravel(), rtvec_from_matrix(m.pose_end).ravel()]
    else:
      motion = [rtvec_from_matrix(m.world_wrt_base), rtvec_from_matrix(m.gripper_wrt_camera)]
    return dict(
      camera_poses=[rtvec_from_matrix(self.camera_poses).ravel()],
      board_poses=[rtvec_from_matrix(self.board_poses).ravel()],
      motion=motion,
      cameras=[cam.param_vec for cam in self.cameras],
      boards=[np.asarray(p).ravel() for p in self.board_points])

  @property
  def param_vec(self):
    blocks = self._blocks()
    parts = [a for k in PARAM_ORDER if self.optimize[k] is True for a in blocks[k]]
    return np.concatenate([np.asarray(p, dtype=np.float64).ravel() for p in parts])

  def with_param_vec(self, x):
    blocks = self._blocks()
    upd, i = {}, 0
    total = sum(a.size for k in PARAM_ORDER if self.optimize[k] is True for a in blocks[k])
    assert x.size == total, f"inconsistent parameter sizes, got {x.size}, expected {total}"
    for k in PARAM_ORDER:
      if self.optimize[k] is not True:
        continue
      vals = []
      for a in blocks[k]:
        vals.append(x[i:i + a.size])
        i += a.size
      if k == 'camera_poses':
        upd['camera_poses'] = rtvec_to_matrix(vals[0].reshape(-1, 6))
      elif k == 'board_poses':
        upd['board_poses'] = rtvec_to_matrix(vals[0].reshape(-1, 6))
      elif k == 'motion':
        m = self.motion
        if m.kind == 'static':
          upd['motion'] = SimpleNamespace(kind='static', valid=m.valid, poses=rtvec_to_matrix(vals[0].reshape(-1, 6)))
        elif m.kind == 'rolling':
          upd['motion'] = SimpleNamespace(kind='rolling', valid=m.valid,
                                          pose_start=rtvec_to_matrix(vals[0].reshape(-1, 6)),
                                          pose_end=rtvec_to_matrix(vals[1].reshape(-1, 6)))
        else:
          upd['motion'] = SimpleNamespace(kind='hand_eye', valid=m.valid, base_wrt_gripper=m.base_wrt_gripper,
                                          world_wrt_base=rtvec_to_matrix(vals[0]),
                                          gripper_wrt_camera=rtvec_to_matrix(vals[1]))
      elif k == 'cameras':
        upd['cameras'] = [cam.with_param_vec(v) for cam, v in zip(self.cameras, vals)]
      elif k == 'boards':
        upd['board_points'] = [v.reshape(-1, 3) for v in vals]
    return self.copy(**upd)

  # --- residual closure (calibration.py:204-206) ----------------------------------------------------
  def evaluate(self, x):
    calib = self.with_param_vec(x)
    proj, _ = calib.reprojected()
    return (proj - calib.points)[self.inliers].ravel()

  # --- sparsity (calibration.py:173-196, parameters.py:109-150) -------------------------------------
  @property
  def sparsity_matrix(self):
    inl = self.inliers
    mask_coords = np.broadcast_to(np.expand_dims(inl, -1), [*inl.shape, 2])
    indices = np.arange(mask_coords.size).reshape(*mask_coords.shape)

    def point_indexes(i, axis, enabled=True):
      return np.take(indices, i, axis=axis).ravel() if enabled else None

    def pose_mapping(valid, axis):
      return [(6, point_indexes(i, axis, ok)) for i, ok in enumerate(valid)]

    m = self.motion
    if m.kind == 'static':
      motion = pose_mapping(m.valid, 1)
    elif m.kind == 'rolling':
      motion = pose_mapping(m.valid, 1) + pose_mapping(m.valid, 1)
    else:
      motion = [(12, indices)]

    cam_params = np.concatenate([cam.param_vec for cam in self.cameras]).reshape(len(self.cameras), -1)
    mappings = dict(
      camera_poses=pose_mapping(self.camera_valid, 0),
      board_poses=pose_mapping(self.board_valid, 2),
      motion=motion,
      cameras=[(p.size, point_indexes(i, 0)) for i, p in enumerate(cam_params)],
      boards=[(p.size, point_indexes(i, 3)) for b in self.board_points
              for i, p in enumerate(np.asarray(b).ravel().reshape(-1, 3))])

    params = sum([mappings[k] for k in PARAM_ORDER if self.optimize[k] is True], [])
    total = sum(n for n, _ in params)
    sparsity = lil_matrix((mask_coords.size, total), dtype='int16')
    count = 0
    for n, idx in params:
      if idx is not None:
        sparsity[idx.reshape(1, -1), (count + np.arange(n)).reshape(-1, 1)] = 1
      count += n
    return sparsity[mask_coords.ravel()]

  # --- solver call (calibration.py:199-212) ---------------------------------------------------------
  def bundle_adjust(self, tolerance=1e-4, f_scale=1.0, max_iterations=100, loss='linear', return_result=False,
                    log=None, **lsq_overrides):
    kw = dict(jac_sparsity=self.sparsity_matrix, verbose=2, x_scale='jac', f_scale=f_scale, ftol=tolerance,
              max_nfev=max_iterations, method='trf', loss=loss)
    kw.update(lsq_overrides)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
      res = optimize.least_squares(self.evaluate, self.param_vec, **kw)
    if log is not None:
      log.append(buf.getvalue())
    out = self.with_param_vec(res.x)
    return (out, res) if return_result else out

  # --- errors / outliers (tables.py:239-249, calibration.py:134-141,240-268,304-310) ----------------
  def reprojection_error_table(self):
    proj, pvalid = self.reprojected()
    mask = pvalid & self.point_valid
    err = np.linalg.norm(proj - self.points, axis=-1)
    err[~mask] = 0
    return err, mask

  @property
  def reprojection_error(self):
    err, mask = self.reprojection_error_table()
    return err[mask]

  @property
  def reprojection_inliers(self):
    proj, pvalid = self.reprojected()
    mask = pvalid & self.inliers
    err = np.linalg.norm(proj - self.points, axis=-1)
    return err[mask]

  def reject_outliers(self, threshold):
    err, valid = self.reprojection_error_table()
    inliers = (err < threshold) & valid
    return self.copy(inlier_mask=inliers)

  def adjust_outliers(self, num_adjustments=3, select_scale=None, select_outliers=None, **kwargs):
    calib = self
    for _ in range(num_adjustments):
      f_scale = (None if select_scale is None else select_scale(calib.reprojection_error)) or 1.0
      if select_outliers is not None:
        calib = calib.reject_outliers(select_outliers(calib.reprojection_error))
      calib = calib.bundle_adjust(f_scale=f_scale, **kwargs)
    return calib


def select_threshold(quantile=0.75, factor=5.0):
  """calibration.py:37-40."""
  return lambda err: np.quantile(err, quantile) * factor


def error_stats(errors):
  """calibration.py:304-310."""
  if len(errors) == 0:
    errors = np.zeros((1, 1), np.float32)
  mse = np.square(errors).mean()
  q = np.array([np.quantile(errors, n) for n in [0, 0.25, 0.5, 0.75, 1]])
  return SimpleNamespace(mse=mse, rms=np.sqrt(mse), quantiles=q, n=errors.size)


# --------------------------------------------------------------------------------------------------
# builders from a synthetic rig (multical_amd.synthetic.make_rig output; plain numpy only)
# --------------------------------------------------------------------------------------------------
def from_rig(rig, which='init'):
  src = getattr(rig, which)
  cams = [OracleCamera(c.image_size, c.intrinsic, c.dist, c.model, c.fix_aspect, c.has_skew) for c in src.cameras]
  kind = rig.cfg["motion"]
  if kind == 'static':
    motion = SimpleNamespace(kind='static', valid=rig.frame_valid, poses=src.rig)
  elif kind == 'rolling':
    motion = SimpleNamespace(kind='rolling', valid=rig.frame_valid, pose_start=src.rig, pose_end=src.rig_end)
  else:
    he = src.hand_eye
    motion = SimpleNamespace(kind='hand_eye', valid=rig.frame_valid, base_wrt_gripper=he.base_wrt_gripper,
                             world_wrt_base=he.world_wrt_base, gripper_wrt_camera=he.gripper_wrt_camera)
  return OracleCalibration(cams, rig.board_points, rig.points, rig.valid, src.camera_poses, rig.camera_valid,
                           src.board_poses, rig.board_valid, motion, optimize=rig.optimize)


# ---- interface/view_table.py (the numeric part of the GUI's reprojection tables) -------------------------
VIEW_TABLE_AXES = dict(overall=None, views=(2, 3), board_views=(3,), boards=(0, 1, 3), cameras=(1, 2, 3), frames=(0, 2, 3))


def reprojection_tables(calib, inlier_only=False):
  """view_table.py:19-52: error of `projected` (NOT `reprojected`: no measured scan time) against the point table, reduced
  per axis: detected, outliers, mse, rms and the five quantiles (nanquantile over the masked errors)."""
  proj, pvalid = calib.projected()
  valid = calib.inliers if inlier_only else calib.point_valid
  valid = pvalid & valid                                                      # tables.py:244-249
  error = np.linalg.norm(proj - calib.points, axis=-1)
  error[~valid] = 0
  out = {}
  for k, axis in VIEW_TABLE_AXES.items():
    n = valid.sum(axis=axis)
    mse = np.square(error).sum(axis=axis) / np.maximum(n, 1)
    e = error.copy()
    e[~valid] = np.nan
    with np.errstate(all='ignore'):
      import warnings
      with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        q = np.nanquantile(e, [0, 0.25, 0.5, 0.75, 1.0], axis=axis)
    out[k] = dict(detected=n, outliers=(valid & ~calib.inliers).sum(axis=axis), mse=mse, rms=np.sqrt(mse), min=q[0],
                  lower_q=q[1], median=q[2], upper_q=q[3], max=q[4])
  return out
