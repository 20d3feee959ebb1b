"""Parameter-space comparison of two calibrations of the same rig.

The bundle adjustment holds no pose fixed (optimization/calibration.py:146-161): X_cam = camera[c] . rig[f] . board[b] . X is
invariant under  camera -> camera T^-1, rig -> T rig  (the freedom `Calibration.with_master` removes at export time,
calibration.py:99-112) and under  rig -> rig S^-1, board -> S board.  Two solvers that reach the same reprojections may sit at
different points of that 12-dimensional null space, so raw parameter vectors are not comparable; `canonical` moves a
calibration to the gauge "first valid camera at the origin, first valid board at the origin" and `parameter_deltas` reports
physical differences in that gauge: focal length (relative), principal point (px), distortion coefficients (absolute),
rotation angle (degrees) and translation (in board units, metres for the example boards) of camera, frame and board poses.
"""
import numpy as np

from .structs import struct


def _first_valid(valid):
  idx = np.flatnonzero(np.asarray(valid))
  return int(idx[0]) if idx.size else 0


def canonical(calib):
  """struct(cameras [C,4,4], frames [F,4,4] (rolling shutter: + frames_end), boards [B,4,4], valid masks, K, dist) of `calib`
  in the gauge camera[c0] = I, board[b0] = I (c0 / b0 = first valid camera / board)."""
  cams, boards = np.asarray(calib.camera_poses.poses), np.asarray(calib.board_poses.poses)
  T0 = cams[_first_valid(calib.camera_poses.valid)]
  S0 = boards[_first_valid(calib.board_poses.valid)]
  T0i, S0i = np.linalg.inv(T0), np.linalg.inv(S0)
  motion = calib.motion
  out = struct(cameras=cams @ T0i, boards=S0i @ boards, camera_valid=np.asarray(calib.camera_poses.valid),
               board_valid=np.asarray(calib.board_poses.valid), frame_valid=np.asarray(motion.valid))
  if hasattr(motion, "pose_start"):
    out = out._extend(frames=T0 @ np.asarray(motion.pose_start) @ S0, frames_end=T0 @ np.asarray(motion.pose_end) @ S0)
  else:
    out = out._extend(frames=T0 @ np.asarray(motion.frame_poses.poses) @ S0)
  cameras = list(calib.cameras)
  out = out._extend(K=np.stack([np.asarray(c.intrinsic, dtype=np.float64) for c in cameras]),
                    dist=[np.asarray(c.dist, dtype=np.float64).ravel() for c in cameras])
  return out


def _pose_delta(A, B, valid):
  """(max rotation angle in degrees, max translation distance) between corresponding valid poses."""
  A, B = np.asarray(A)[valid], np.asarray(B)[valid]
  if A.shape[0] == 0:
    return 0.0, 0.0
  R = np.einsum('nij,nkj->nik', A[:, :3, :3], B[:, :3, :3])          # R_A R_B^T
  cos = np.clip((np.trace(R, axis1=1, axis2=2) - 1.0) / 2.0, -1.0, 1.0)
  # (small angles: arccos loses half the digits; use the sine of the skew part instead)
  skew = 0.5 * np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1)
  ang = np.arctan2(np.linalg.norm(skew, axis=1), cos)
  return float(np.degrees(ang.max())), float(np.linalg.norm(A[:, :3, 3] - B[:, :3, 3], axis=1).max())


def parameter_deltas(a, b):
  """Largest physical differences between calibrations `a` and `b` of the same rig, after moving both to the common gauge."""
  ca, cb = canonical(a), canonical(b)
  fa = np.stack([ca.K[:, 0, 0], ca.K[:, 1, 1]], axis=1)
  fb = np.stack([cb.K[:, 0, 0], cb.K[:, 1, 1]], axis=1)
  d = struct(focal_rel=float(np.abs(fa / fb - 1.0).max()),
             principal_px=float(np.abs(ca.K[:, :2, 2] - cb.K[:, :2, 2]).max()),
             dist_abs=float(max(np.abs(x - y).max() for x, y in zip(ca.dist, cb.dist))))
  d = d._extend(**dict(zip(("camera_deg", "camera_t"), _pose_delta(ca.cameras, cb.cameras, ca.camera_valid & cb.camera_valid))))
  fdeg, ft = _pose_delta(ca.frames, cb.frames, ca.frame_valid & cb.frame_valid)
  if "frames_end" in ca and "frames_end" in cb:
    edeg, et = _pose_delta(ca.frames_end, cb.frames_end, ca.frame_valid & cb.frame_valid)
    fdeg, ft = max(fdeg, edeg), max(ft, et)
  d = d._extend(frame_deg=fdeg, frame_t=ft)
  d = d._extend(**dict(zip(("board_deg", "board_t"), _pose_delta(ca.boards, cb.boards, ca.board_valid & cb.board_valid))))
  return d
