"""TEST INFRASTRUCTURE (oracle harness): build a *real* reference `Calibration` from a synthetic rig.

Uses the unmodified reference classes (loaded through oracle/refload.py): Camera / CameraFisheye,
CharucoBoard / AprilGrid, PoseSet, StaticFrames / RollingFrames / HandEye, ParamList, Calibration.
`tables.initialise_poses` is bypassed on purpose (it breaks under numpy 2.x: `np.bool`, tables.py:363;
and BA parity wants identical initial values on both sides anyway) -- SURVEY.md section 7, hard part 7.
"""
import numpy as np
from . import refload

BOARD_ARGS = {
  "charuco_16x22": ("charuco", dict(size=(16, 22), square_length=0.025, marker_length=0.01875, aruco_dict='4X4_1000')),
  "charuco_10x10": ("charuco", dict(size=(10, 10), square_length=0.040, marker_length=0.032, aruco_dict='5X5_1000',
                                    min_rows=3, min_points=9)),
  "aprilgrid_9x9": ("aprilgrid", dict(size=(9, 9), tag_length=0.06, tag_spacing=0.3)),
  "charuco_25x35": ("charuco", dict(size=(25, 35), square_length=0.020, marker_length=0.015, aruco_dict='5X5_1000')),
}


def reference_calibration(rig, which='init'):
  ref = refload.load()
  from multical.board import CharucoBoard, AprilGrid
  from multical.camera import Camera
  from multical.camera_fisheye import CameraFisheye
  from multical.optimization.calibration import Calibration
  from multical.optimization.parameters import ParamList
  from multical.optimization.pose_set import PoseSet
  from multical.motion import StaticFrames, RollingFrames, HandEye
  from structs.numpy import Table

  src = getattr(rig, which)
  C, F, B, P = rig.valid.shape
  cam_names = [f"cam{i}" for i in range(C)]
  board_names = [f"board{i}" for i in range(B)]
  frame_names = [f"frame{i}" for i in range(F)]

  cameras = []
  for c in src.cameras:
    if c.model == 'fisheye':
      cameras.append(CameraFisheye(c.image_size, c.intrinsic.copy(), c.dist.copy(),
                                   fix_aspect=c.fix_aspect, has_skew=c.has_skew))
    else:
      cameras.append(Camera(c.image_size, c.intrinsic.copy(), c.dist.copy(), model=c.model,
                            fix_aspect=c.fix_aspect, has_skew=c.has_skew))

  boards = []
  for i, name in enumerate(rig.cfg["boards"]):
    kind, kw = BOARD_ARGS[name]
    kw = dict(kw)
    if kind == "charuco":
      kw["aruco_offset"] = 50 * i if rig.cfg["boards"].count(name) > 1 else 0
      b = CharucoBoard(**kw)
    else:
      b = AprilGrid(**kw)
    assert np.array_equal(np.asarray(b.adjusted_points), rig.board_points[i]), "board geometry mismatch"
    boards.append(b)

  point_table = Table.create(points=rig.points, valid=rig.valid)
  camera_poses = PoseSet(Table.create(poses=src.camera_poses, valid=rig.camera_valid), cam_names)
  board_poses = PoseSet(Table.create(poses=src.board_poses, valid=rig.board_valid), board_names)

  kind = rig.cfg["motion"]
  if kind == 'static':
    motion = StaticFrames(Table.create(poses=src.rig, valid=rig.frame_valid), frame_names)
  elif kind == 'rolling':
    motion = RollingFrames(src.rig, src.rig_end, rig.frame_valid, frame_names)
  else:
    he = src.hand_eye
    motion = HandEye(Table.create(poses=he.base_wrt_gripper, valid=rig.frame_valid),
                     he.world_wrt_base, he.gripper_wrt_camera, frame_names)

  calib = Calibration(ParamList(cameras, cam_names), ParamList(boards, board_names), point_table,
                      camera_poses, board_poses, motion)
  return calib.enable(**rig.optimize), ref
