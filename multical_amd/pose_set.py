"""Host mirror of the reference's pose block: N rigid transforms with a validity flag each, exposed to the optimiser as
N x (rotation vector | translation).  Mirrors the public surface of multical/optimization/pose_set.py:12-72 (`poses`,
`valid`, `names`, `params` / `with_params`, `pre_transform` / `post_transform`, item access by index or name) so that
code written against the reference's class reads the same; the storage is a Table with fields `poses`, `valid`.
"""
from functools import cached_property

import numpy as np

from . import transform
from .parameters import Parameters
from .structs import Table


def _as_table(t):
  return t if isinstance(t, Table) else Table(t)


class PoseSet(Parameters):
  STATE = ("pose_table", "names")

  def __init__(self, pose_table, names=None):
    self.pose_table = _as_table(pose_table)
    count = self.pose_table.poses.shape[0]
    self.names = list(names) if names else [str(i) for i in range(count)]

  # -- read access ----------------------------------------------------------------------------------------------
  poses = property(lambda self: self.pose_table.poses)      # [N, 4, 4]
  valid = property(lambda self: self.pose_table.valid)      # [N] bool; invalid poses stay in x with zero Jacobian columns
  size = property(lambda self: self.pose_table.poses.shape[0])

  def index_of(self, key):
    if not isinstance(key, str):
      return key
    try:
      return self.names.index(key)
    except ValueError:
      raise KeyError(f"pose {key} not found in {self.names}") from None

  def __getitem__(self, key):
    return self.poses[self.index_of(key)]

  # -- parameter block (pose_set.py:51-57): rows of rtvec.from_matrix, row-major --------------------------------
  @cached_property
  def params(self):
    return transform.from_matrix(self.poses).reshape(-1)

  def with_params(self, params):
    rt = np.asarray(params).reshape(self.size, transform.size)
    return self._replace_poses(transform.to_matrix(rt), keep_valid=True)

  # -- rigid re-gauging (pose_set.py:45-49) ------------------------------------------------------------------------
  def pre_transform(self, t):
    return self._replace_poses(t @ self.poses)

  def post_transform(self, t):
    return self._replace_poses(self.poses @ t)

  def _replace_poses(self, poses, keep_valid=False):
    table = self.pose_table._update(poses=poses) if keep_valid else self.pose_table._extend(poses=poses)
    return self.copy(pose_table=table)

  # -- value semantics --------------------------------------------------------------------------------------------
  def __getstate__(self):
    return {k: getattr(self, k) for k in self.STATE}

  def __setstate__(self, state):
    self.__dict__.update(state)

  def copy(self, **changes):
    return type(self)(**{**self.__getstate__(), **changes})
