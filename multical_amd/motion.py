"""Motion models as parameter blocks (multical/motion/): how the rig pose of a frame is parameterised.

  StaticFrames   one pose per frame                                     static_frames.py:29-42
  RollingFrames  start / end pose per frame, linear in the scan time    rolling_frames.py:66-166
  HandEye        rig[f] = gripper_wrt_camera @ base_wrt_gripper[f] @ world_wrt_base   motion/hand_eye.py:14-107

The projection itself (`MotionModel.project`, motion_model.py:1-8) is not implemented on the host: it is part of the
HIP kernels (multical_amd/csrc/mcba_kernels.h: slot_forward) and reached through Calibration.reprojected.
"""
from functools import cached_property
import numpy as np
from . import transform
from .parameters import Parameters
from .pose_set import PoseSet
from .structs import Table, struct


class MotionModel(object):
  pass


class StaticFrames(PoseSet, MotionModel):
  @staticmethod
  def init(pose_table, names=None):
    return StaticFrames(pose_table, names)

  @property
  def frame_poses(self):
    return self.pose_table


class RollingFrames(MotionModel, Parameters):
  def __init__(self, pose_start, pose_end, valid, names=None, max_iterations=4):
    self.pose_start = np.asarray(pose_start)
    self.pose_end = np.asarray(pose_end)
    self.valid = np.asarray(valid)
    self.names = names or [str(i) for i in range(self.pose_start.shape[0])]
    self.max_iterations = max_iterations

  @staticmethod
  def init(pose_table, names=None, max_iterations=4):
    return RollingFrames(pose_table.poses, pose_table.poses, pose_table.valid, names, max_iterations)

  @property
  def size(self):
    return self.pose_start.shape[0]

  @property
  def frame_poses(self):
    return Table.create(poses=self.pose_start, valid=self.valid)

  def pre_transform(self, t):
    return self.copy(pose_start=t @ self.pose_start, pose_end=t @ self.pose_end)

  def post_transform(self, t):
    return self.copy(pose_start=self.pose_start @ t, pose_end=self.pose_end @ t)

  @cached_property
  def params(self):
    return [transform.from_matrix(self.pose_start).ravel(), transform.from_matrix(self.pose_end).ravel()]

  def with_params(self, params):
    start, end = [transform.to_matrix(m.reshape(-1, 6)) for m in params]
    return self.copy(pose_start=start, pose_end=end)

  def __getstate__(self):
    return dict(pose_start=self.pose_start, pose_end=self.pose_end, valid=self.valid, names=self.names,
                max_iterations=self.max_iterations)

  def __setstate__(self, d):
    self.__dict__.update(d)

  def copy(self, **k):
    d = self.__getstate__()
    d.update(k)
    return self.__class__(**d)


class HandEye(Parameters, MotionModel):
  def __init__(self, base_wrt_gripper, world_wrt_base, gripper_wrt_camera, names=None):
    self.base_wrt_gripper = base_wrt_gripper if isinstance(base_wrt_gripper, Table) else Table(base_wrt_gripper)
    n = self.base_wrt_gripper.poses.shape[0]
    self.names = names or [str(i) for i in range(n)]
    self.world_wrt_base = np.asarray(world_wrt_base)
    self.gripper_wrt_camera = np.asarray(gripper_wrt_camera)

  @property
  def size(self):
    return self.base_wrt_gripper.poses.shape[0]

  @property
  def valid(self):
    return self.base_wrt_gripper.valid

  @property
  def pose_table(self):
    poses = (self.gripper_wrt_camera @ self.base_wrt_gripper.poses) @ self.world_wrt_base
    return Table.create(poses=poses, valid=self.valid)

  frame_poses = pose_table

  def pre_transform(self, t):
    return self.copy(gripper_wrt_camera=t @ self.gripper_wrt_camera)

  def post_transform(self, t):
    return self.copy(world_wrt_base=self.world_wrt_base @ t)

  @cached_property
  def params(self):
    return struct(world_wrt_base=transform.from_matrix(self.world_wrt_base),
                  gripper_wrt_camera=transform.from_matrix(self.gripper_wrt_camera))

  def with_params(self, params):
    return self.copy(world_wrt_base=transform.to_matrix(params.world_wrt_base),
                     gripper_wrt_camera=transform.to_matrix(params.gripper_wrt_camera))

  def __getstate__(self):
    return dict(base_wrt_gripper=self.base_wrt_gripper, gripper_wrt_camera=self.gripper_wrt_camera,
                world_wrt_base=self.world_wrt_base, names=self.names)

  def __setstate__(self, d):
    self.__dict__.update(d)

  def copy(self, **k):
    d = self.__getstate__()
    d.update(k)
    return self.__class__(**d)
