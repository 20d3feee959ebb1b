"""SE(3) <-> rotation-vector conversions of the host mirror.

Same conventions and the same third-party call as the reference (multical/transform/rtvec.py:16-32,
multical/transform/matrix.py:33-44): scipy's Rotation, read-back canonicalised to an angle in [0, pi]."""
import numpy as np
from scipy.spatial.transform import Rotation as R

size = 6


def split(rtvec):
  assert rtvec.shape[-1] == size, f"split - bad shape: {rtvec.shape}"
  return rtvec[..., 0:3], rtvec[..., 3:6]


def to_matrix(rtvec):
  rtvec = np.asarray(rtvec, dtype=np.float64)
  rvec, tvec = split(rtvec)
  m = np.zeros(rtvec.shape[:-1] + (4, 4))
  m[..., :3, :3] = R.from_rotvec(rvec).as_matrix()
  m[..., :3, 3] = tvec
  m[..., 3, 3] = 1.0
  return m


def from_matrix(m):
  m = np.asarray(m, dtype=np.float64)
  assert m.shape[-2:] == (4, 4)
  rvec = R.from_matrix(m[..., :3, :3]).as_rotvec()
  return np.hstack([rvec, m[..., :3, 3]])
