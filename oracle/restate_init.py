"""TEST INFRASTRUCTURE -- numpy/scipy CPU restatement of the reference's pose-graph INITIALISATION tables:
the producer of the bundle adjustment's inputs.  A checker for multical_amd.tables (HIP); never imported by the product.

What it restates (paths relative to /root/reference/multical/):
  tables.initialise_poses               tables.py:353-377
  tables.estimate_relative_poses(_inv)  tables.py:207-230          (graph.select_pairs: graph.py:7-33)
  tables.pattern_overlaps               tables.py:134-148
  tables.estimate_transform             tables.py:153-176
  tables.relative_between(_inv / _n)    tables.py:326-345
  matrix.align_transforms_robust        transform/matrix.py:140-153 (test_outlier :135-137, error_transform :66-67)
  matrix.mean_robust                    transform/matrix.py:109-113 -> common.mean_robust / cluster (transform/common.py:6-21:
                                        scipy `linkage(whiten(v), 'ward')` + `fcluster(maxclust)` + most common cluster)
  rtvec.from_matrix / to_matrix         transform/rtvec.py:24-32   (scipy Rotation, like the reference)
Tables are plain dicts {poses [..., 4, 4], valid [...]}.

PARITY PIN: tests/test_oracle_vs_reference.py::test_initialise_poses_restatement runs this file against the unmodified
reference (`np.bool` restored for the two lines of the reference that still use it, tables.py:363 and matrix.py:145).
"""
from collections import Counter

import numpy as np
from scipy.cluster.hierarchy import linkage, fcluster
from scipy.cluster.vq import whiten
from scipy.spatial.transform import Rotation as R


# ---- transform/rtvec.py, transform/matrix.py -----------------------------------------------------------------------
def rtvec_from_matrix(m):
  """rtvec.py:29-32."""
  return np.hstack([R.from_matrix(m[..., :3, :3]).as_rotvec(), m[..., :3, 3]])


def rtvec_to_matrix(rtvec):
  """rtvec.py:24-27 + matrix.join."""
  rtvec = np.asarray(rtvec, dtype=np.float64)
  m = np.zeros(rtvec.shape[:-1] + (4, 4))
  m[..., :3, :3] = R.from_rotvec(rtvec[..., 0:3]).as_matrix()
  m[..., :3, 3] = rtvec[..., 3:6]
  m[..., 3, 3] = 1.0
  return m


def cluster(vectors, min_clusters=3, cluster_size=10):
  """transform/common.py:6-15."""
  Z = linkage(whiten(vectors), 'ward')
  n_clust = max(vectors.shape[0] / cluster_size, min_clusters)
  clusters = fcluster(Z, t=n_clust, criterion='maxclust')
  cc = Counter(clusters[clusters >= 0])
  most = cc.most_common(n=1)[0][0]
  return clusters == most


def mean_robust_vectors(vectors):
  """transform/common.py:18-21."""
  return vectors[cluster(vectors)].mean(axis=0) if len(vectors) > 1 else vectors[0]


def mean_robust(m):
  """transform/matrix.py:109-113."""
  return rtvec_to_matrix(mean_robust_vectors(rtvec_from_matrix(m)))


def relative_to(source, dest):
  """matrix.py:60-61."""
  return dest @ np.linalg.inv(source)


def error_transform(t, source, dest):
  """matrix.py:66-67."""
  return np.linalg.norm(t @ source - dest, axis=(1, 2))


def align_transforms_robust(m1, m2, valid=None, threshold=1.5):
  """matrix.py:140-153 (align_transforms_mean :78-79, test_outlier :135-137)."""
  mask = np.ones(m1.shape[0], dtype=bool) if valid is None else valid
  m = mean_robust(relative_to(m1[mask], m2[mask]))
  errs = error_transform(m, m1, m2)
  inliers = (errs < np.quantile(errs, 0.75) * threshold) & mask
  m = mean_robust(relative_to(m1[inliers], m2[inliers]))
  return m, inliers


# ---- tables.py ------------------------------------------------------------------------------------------------------
def table(poses, valid):
  return dict(poses=np.asarray(poses, dtype=np.float64), valid=np.asarray(valid, dtype=bool))


def inverse(t):
  return table(np.linalg.inv(t["poses"]), t["valid"])


def index_select(t, i, axis):
  return table(np.take(t["poses"], i, axis=axis), np.take(t["valid"], i, axis=axis))


def pattern_overlaps(t, num_points, axis=0):
  """tables.py:134-148 (num_points: the pose table's per-entry detection count)."""
  n = t["valid"].shape[axis]
  overlaps = np.zeros([n, n])
  for i in range(n):
    for j in range(i + 1, n):
      vi, vj = np.take(t["valid"], i, axis=axis), np.take(t["valid"], j, axis=axis)
      ni, nj = np.take(num_points, i, axis=axis), np.take(num_points, j, axis=axis)
      has_pose = vi & vj
      weight = np.min([ni, nj], axis=0)
      overlaps[i, j] = overlaps[j, i] = np.sum(has_pose.astype(np.float32) * weight)
  return overlaps


def select_pairs(overlaps, hop_penalty=0.8):
  """graph.py:7-33."""
  overlaps = overlaps.copy()
  n = overlaps.shape[0]
  master = np.argmax(overlaps.sum(1))
  weight = (np.arange(n) == master).astype(np.float32).reshape(n, 1)
  overlaps[:, master] = 0
  pairs = []
  while len(pairs) + 1 < n:
    i = np.unravel_index(np.argmax(overlaps * weight), overlaps.shape)
    overlap = (overlaps * weight)[i]
    if overlap <= 0:
      break
    parent, child = i
    overlaps[:, child] = 0
    weight[child] = weight[parent] * hop_penalty
    pairs.append((int(parent), int(child)))
  return int(master), pairs


def estimate_transform(t, i, j, axis=0):
  """tables.py:153-176 (without the log lines)."""
  ti, tj = index_select(t, i, axis), index_select(t, j, axis)
  valid = (ti["valid"] & tj["valid"]).ravel()
  m, _ = align_transforms_robust(ti["poses"].reshape(-1, 4, 4), tj["poses"].reshape(-1, 4, 4), valid=valid)
  return m


def fill_poses(pose_dict, n):
  """tables.py:178-183."""
  poses = np.broadcast_to(np.eye(4), (n, 4, 4)).copy()
  valid = np.zeros(n, dtype=bool)
  for k in sorted(pose_dict):
    poses[k] = pose_dict[k]
    valid[k] = True
  return table(poses, valid)


def estimate_relative_poses(t, num_points, axis=0, hop_penalty=0.9):
  """tables.py:207-227."""
  n = t["valid"].shape[axis]
  overlaps = pattern_overlaps(t, num_points, axis=axis)
  master, pairs = select_pairs(overlaps, hop_penalty)
  pose_dict = {master: np.eye(4)}
  for parent, child in pairs:
    pose_dict[child] = estimate_transform(t, parent, child, axis=axis) @ pose_dict[parent]
  rel = fill_poses(pose_dict, n)
  return table(rel["poses"] @ np.linalg.inv(rel["poses"][0]), rel["valid"])


def estimate_relative_poses_inv(t, num_points, axis=2, hop_penalty=0.9):
  """tables.py:229-230."""
  return inverse(estimate_relative_poses(inverse(t), num_points, axis=axis, hop_penalty=hop_penalty))


def relative_between(t1, t2):
  """tables.py:326-332."""
  valid = np.nonzero(t1["valid"] & t2["valid"])
  if valid[0].size == 0:
    return np.eye(4), False
  m, _ = align_transforms_robust(t1["poses"][valid], t2["poses"][valid])
  return m, True


def relative_between_n(t1, t2, axis=0, inv=False):
  """tables.py:334-345."""
  n = t1["valid"].shape[axis]
  poses, valid = [], []
  for k in range(n):
    a, b = index_select(t1, k, axis), index_select(t2, k, axis)
    if inv:
      m, ok = relative_between(inverse(a), inverse(b))
      m = np.linalg.inv(m)
    else:
      m, ok = relative_between(a, b)
    poses.append(m)
    valid.append(ok)
  return table(np.stack(poses), np.array(valid))


def initialise_poses(pose_table, num_points, camera_poses=None):
  """tables.py:353-377: camera / board / rig-pose tables from the per-view board poses [C, F, B]."""
  camera = estimate_relative_poses(pose_table, num_points, axis=0)
  if camera_poses is not None:
    camera = table(camera_poses, np.ones(camera_poses.shape[0], dtype=bool))
  board = estimate_relative_poses_inv(pose_table, num_points, axis=2)
  binv = inverse(board)
  board_relative = table(pose_table["poses"] @ binv["poses"][None, None], pose_table["valid"] & binv["valid"][None, None])
  expanded = table(np.broadcast_to(camera["poses"][:, None, None], board_relative["poses"].shape),
                   np.broadcast_to(camera["valid"][:, None, None], board_relative["valid"].shape))
  times = relative_between_n(expanded, board_relative, axis=1, inv=True)
  return dict(times=times, camera=camera, board=board)
