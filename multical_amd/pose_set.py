"""PoseSet: named set of 4x4 poses + validity as a 6-DoF parameter block (multical/optimization/pose_set.py:12-72)."""
from functools import cached_property
import numpy as np
from . import transform
from .parameters import Parameters
from .structs import Table


class PoseSet(Parameters):
  def __init__(self, pose_table, names=None):
    self.pose_table = pose_table if isinstance(pose_table, Table) else Table(pose_table)
    self.names = names or [str(i) for i in range(self.size)]

  @property
  def size(self):
    return self.poses.shape[0]

  @property
  def valid(self):
    return self.pose_table.valid

  @property
  def poses(self):
    return self.pose_table.poses

  def __getitem__(self, k):
    if isinstance(k, str):
      if k not in self.names:
        raise KeyError(f"pose {k} not found in {self.names}")
      return self.poses[self.names.index(k)]
    return self.poses[k]

  def pre_transform(self, t):
    return self.copy(pose_table=self.pose_table._extend(poses=t @ self.poses))

  def post_transform(self, t):
    return self.copy(pose_table=self.pose_table._extend(poses=self.poses @ t))

  @cached_property
  def params(self):
    return transform.from_matrix(self.poses).ravel()

  def with_params(self, params):
    m = transform.to_matrix(params.reshape(-1, transform.size))
    return self.copy(pose_table=self.pose_table._update(poses=m))

  def __getstate__(self):
    return dict(pose_table=self.pose_table, names=self.names)

  def __setstate__(self, d):
    self.__dict__.update(d)

  def copy(self, **k):
    d = self.__getstate__()
    d.update(k)
    return self.__class__(**d)
