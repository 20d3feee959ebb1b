"""Initialisation tables of the pose graph (multical/tables.py:134-230,326-377) with the numeric core on the MI355X.

The reference builds the bundle adjustment's starting point from the per-view board poses [cameras, frames, boards]:
relative camera poses and relative board poses through a spanning tree of pairwise robust alignments
(`estimate_relative_poses`, tables.py:207-230) and one rig pose per frame (`relative_between_n`, tables.py:337-345).  Every
alignment is `matrix.align_transforms_robust` (transform/matrix.py:140-153): relative poses -> robust mean (Ward clustering
of the whitened rotation-vector | translation 6-vectors, transform/common.py:6-21) -> upper-quartile outlier test -> robust
mean.  Here all alignments of a stage run as ONE batch on the device (mcba_align_poses_robust: a workgroup per problem);
what stays on the host is the control logic on tiny arrays -- the overlap matrix, the greedy spanning tree
(graph.select_pairs, graph.py:7-33), chaining the pair transforms along the tree -- and 4x4 matrix products.

Tables are `structs.Table`s with `poses [..., 4, 4]`, `valid [...]` (+ `num_points` for the pose table), like the
reference's.  `make_point_table` (tables.py:68-81) is data marshalling of ragged detections and stays in numpy.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .structs import Table, struct


def _f64(a):
  return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def align_transforms_robust_ragged(A, B, sizes, mask=None, threshold=1.5, invert=False):
  """The device call on an already concatenated batch: A, B [total, 4, 4], sizes [P] entries per problem (in order), mask
  [total] or None.  Returns (transforms [P,4,4], valid [P] bool, inlier flags [total] bool)."""
  lib = _lib.load()
  sizes = np.asarray(sizes, dtype=np.int64).reshape(-1)
  P = int(sizes.size)
  if P == 0:
    return np.zeros((0, 4, 4)), np.zeros(0, dtype=bool), np.zeros(0, dtype=bool)
  offsets = np.ascontiguousarray(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64))
  total = int(offsets[-1])
  A = _f64(A).reshape(-1, 4, 4) if total else np.zeros((1, 4, 4))
  B = _f64(B).reshape(-1, 4, 4) if total else np.zeros((1, 4, 4))
  assert A.shape[0] == max(total, 1) and B.shape[0] == max(total, 1)
  if mask is not None:
    mask = np.ascontiguousarray(np.asarray(mask).astype(np.uint8).reshape(-1)) if total else np.zeros(1, dtype=np.uint8)
  out = np.empty((P, 4, 4))
  valid = np.empty(P, dtype=np.uint8)
  inl = np.empty(max(total, 1), dtype=np.uint8)
  dp = C.POINTER(C.c_double)
  up = C.POINTER(C.c_uint8)
  import os, time
  t0 = time.perf_counter()
  check(lib.mcba_align_poses_robust(P, offsets.ctypes.data_as(C.POINTER(C.c_int64)), A.ctypes.data_as(dp), B.ctypes.data_as(dp),
                                    None if mask is None else mask.ctypes.data_as(up), float(threshold), 1 if invert else 0,
                                    out.ctypes.data_as(dp), valid.ctypes.data_as(up), inl.ctypes.data_as(up)))
  if os.environ.get("MCBA_TIMING"):
    print(f"[mcba_align_poses_robust] {P} problems, {total} entries, largest {int(sizes.max())}: {(time.perf_counter() - t0) * 1e3:.2f} ms",
          flush=True)
  return out, valid.astype(bool), inl[:total].astype(bool)


def align_transforms_robust_indexed(table, index_a, index_b, sizes, mask=None, threshold=1.5, invert=False):
  """The same device batch with the pairs given as indices into ONE pose table [N, 4, 4] (mcba_align_poses_indexed): entry k is
  the pair (table[index_a[k]], table[index_b[k]]).  Returns (transforms [P,4,4], valid [P] bool, inlier flags [total] bool)."""
  lib = _lib.load()
  sizes = np.asarray(sizes, dtype=np.int64).reshape(-1)
  P = int(sizes.size)
  if P == 0:
    return np.zeros((0, 4, 4)), np.zeros(0, dtype=bool), np.zeros(0, dtype=bool)
  offsets = np.ascontiguousarray(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64))
  total = int(offsets[-1])
  table = _f64(table).reshape(-1, 4, 4)
  ia = np.ascontiguousarray(np.asarray(index_a).reshape(-1), dtype=np.int32) if total else np.zeros(1, dtype=np.int32)
  ib = np.ascontiguousarray(np.asarray(index_b).reshape(-1), dtype=np.int32) if total else np.zeros(1, dtype=np.int32)
  assert ia.size == max(total, 1) and ib.size == max(total, 1) and table.shape[0] > 0
  if mask is not None:
    mask = np.ascontiguousarray(np.asarray(mask).astype(np.uint8).reshape(-1)) if total else np.zeros(1, dtype=np.uint8)
  out = np.empty((P, 4, 4))
  valid = np.empty(P, dtype=np.uint8)
  inl = np.empty(max(total, 1), dtype=np.uint8)
  dp, up, ip = C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
  tp = table.ctypes.data_as(dp)
  check(lib.mcba_align_poses_indexed(P, offsets.ctypes.data_as(C.POINTER(C.c_int64)), tp, table.shape[0], ia.ctypes.data_as(ip), tp,
                                     table.shape[0], ib.ctypes.data_as(ip), None if mask is None else mask.ctypes.data_as(up),
                                     float(threshold), 1 if invert else 0, out.ctypes.data_as(dp), valid.ctypes.data_as(up),
                                     inl.ctypes.data_as(up)))
  return out, valid.astype(bool), inl[:total].astype(bool)


def align_transforms_robust_batch(problems, threshold=1.5, invert=False):
  """problems: list of (m1 [n,4,4], m2 [n,4,4], mask [n] bool or None).  Returns (transforms [P,4,4], valid [P] bool,
  list of inlier masks) -- per problem exactly matrix.align_transforms_robust(m1, m2, valid=mask, threshold)
  (invert=True: tables.relative_between_inv: inputs and result inverted)."""
  P = len(problems)
  if P == 0:
    return np.zeros((0, 4, 4)), np.zeros(0, dtype=bool), []
  sizes = [int(np.asarray(p[0]).shape[0]) for p in problems]
  total = sum(sizes)
  A = np.concatenate([np.asarray(p[0], dtype=np.float64).reshape(-1, 4, 4) for p in problems]) if total else np.zeros((0, 4, 4))
  B = np.concatenate([np.asarray(p[1], dtype=np.float64).reshape(-1, 4, 4) for p in problems]) if total else np.zeros((0, 4, 4))
  mask = None
  if any(p[2] is not None for p in problems):
    mask = np.concatenate([np.ones(n, dtype=np.uint8) if p[2] is None else np.asarray(p[2]).astype(np.uint8)
                           for p, n in zip(problems, sizes)]) if total else np.zeros(0, dtype=np.uint8)
  out, valid, inl = align_transforms_robust_ragged(A, B, sizes, mask, threshold, invert)
  offsets = np.concatenate([[0], np.cumsum(sizes)])
  return out, valid, [inl[offsets[i]:offsets[i + 1]] for i in range(P)]


# ---- host control logic on [n, n] matrices (tables.py:134-148, graph.py:7-33) -------------------------------------------
def pattern_overlaps(table, axis=0):
  """tables.py:134-148: overlaps[i, j] = sum over the entries both i and j see of min(num_points).  The reference sums
  `has_pose.astype(float32) * weight` per pair; the terms are integers, so the float32 sum is EXACT (independent of its order)
  while it stays below 2^24 -- then one row of min() per index replaces the n (n - 1) / 2 np.take pairs (6 ms at 16 cameras x 1000
  frames x 5 boards); larger sums keep the reference's expression."""
  n = table.valid.shape[axis]
  V = np.moveaxis(np.asarray(table.valid), axis, 0).reshape(n, -1)
  W = np.moveaxis(np.asarray(table.num_points), axis, 0).reshape(n, -1)
  Wv = np.where(V, W, 0).astype(np.int64)          # min(w_i, w_j) over the common entries = min of the masked weights
  overlaps = np.zeros([n, n])
  if n > 0 and int(Wv.sum(axis=1).max(initial=0)) < (1 << 24):
    for i in range(n):
      overlaps[i] = np.minimum(Wv[i][None], Wv).sum(axis=1)
    np.fill_diagonal(overlaps, 0.0)
    return overlaps
  for i in range(n):
    for j in range(i + 1, n):
      has_pose = V[i] & V[j]
      weight = np.minimum(W[i], W[j])
      overlaps[i, j] = overlaps[j, i] = np.sum(has_pose.astype(np.float32) * weight)
  return overlaps


def select_pairs(overlaps, hop_penalty=0.8):
  overlaps = overlaps.copy()
  n = overlaps.shape[0]
  master = int(np.argmax(overlaps.sum(1)))
  weight = (np.arange(n) == master).astype(np.float32).reshape(n, 1)
  overlaps[:, master] = 0
  pairs = []
  while len(pairs) + 1 < n:
    w = overlaps * weight
    parent, child = np.unravel_index(np.argmax(w), w.shape)
    if w[parent, child] <= 0:
      break
    overlaps[:, child] = 0
    weight[child] = weight[parent] * hop_penalty
    pairs.append((int(parent), int(child)))
  return master, pairs


def inverse_poses(m):
  """(R | t) -> (R^T | -R^T t): the closed form agrees with the reference's np.linalg.inv (tables.py:229-230 use) to the last
  bits (1e-16) and costs a tenth of it."""
  m = np.asarray(m, dtype=np.float64)
  out = np.zeros(m.shape)
  out[..., :3, :3] = np.swapaxes(m[..., :3, :3], -1, -2)
  t = m[..., :3, 3]
  out[..., :3, 3] = -((m[..., 0, :3] * t[..., 0:1] + m[..., 1, :3] * t[..., 1:2]) + m[..., 2, :3] * t[..., 2:3])
  out[..., 3, 3] = 1.0
  return out


def inverse(table):
  """tables.inverse (tables.py:229-230 use)."""
  return table._extend(poses=inverse_poses(table.poses))


def estimate_relative_poses(table, axis=0, hop_penalty=0.9, of_inverse=False):
  """tables.py:207-227: all pair alignments of the spanning tree in one device batch.
  of_inverse: the result for `inverse(table)` without forming that table -- the device inverts the gathered entries as it loads
  them (`invert`: relative_between_inv semantics, inputs AND result inverted) and the handful of results is inverted back here
  (the closed-form inverse of all 80 000 poses of a 16 x 1000 x 5 table was a third of the host time of an initialisation)."""
  n = table.valid.shape[axis]
  master, pairs = select_pairs(pattern_overlaps(table, axis=axis), hop_penalty)
  if pairs:
    # the pose table goes to the device as it is and the batch [pairs x entries] as two lists of indices into it (per-pair np.take
    # + a concatenation of the problems were three host copies of every side, 10 ms at 15 pairs x 5 000 entries, and 19 MB of
    # upload where the table has 10)
    shape3 = np.asarray(table.valid).shape
    idx = np.moveaxis(np.arange(int(np.prod(shape3)), dtype=np.int32).reshape(shape3), axis, 0).reshape(n, -1)
    V = np.moveaxis(np.asarray(table.valid), axis, 0).reshape(n, -1)
    par = np.fromiter((p for p, _ in pairs), dtype=np.intp, count=len(pairs))
    chi = np.fromiter((c for _, c in pairs), dtype=np.intp, count=len(pairs))
    ts, ok, _ = align_transforms_robust_indexed(table.poses, idx[par], idx[chi], np.full(len(pairs), V.shape[1], dtype=np.int64),
                                                (V[par] & V[chi]).reshape(-1), invert=of_inverse)
    if of_inverse:
      ts = inverse_poses(ts)
  else:
    ts, ok = np.zeros((0, 4, 4)), np.zeros(0, dtype=bool)
  pose_dict = {master: np.eye(4)}
  for (parent, child), t, good in zip(pairs, ts, ok):
    if not good:
      # no common entry, or none that passes the upper-quartile test of the first pass: the reference's
      # align_transforms_robust takes the mean of an empty set there (transform/matrix.py:140-153) and fails; the device
      # reports the case instead of a pose (an identity transform must never be chained into the spanning tree)
      raise ValueError(f"estimate_relative_poses (axis={axis}): no usable common poses for pair ({parent}, {child})")
    pose_dict[child] = t @ pose_dict[parent]
  poses = np.broadcast_to(np.eye(4), (n, 4, 4)).copy()
  valid = np.zeros(n, dtype=bool)
  for k in sorted(pose_dict):
    poses[k] = pose_dict[k]
    valid[k] = True
  return Table.create(poses=poses @ np.linalg.inv(poses[0]), valid=valid)


def estimate_relative_poses_inv(table, axis=2, hop_penalty=0.9):
  """tables.py:229-230."""
  return inverse(estimate_relative_poses(table, axis=axis, hop_penalty=hop_penalty, of_inverse=True))


def relative_between_n(table1, table2, axis=0, inv=False):
  """tables.py:337-345: one alignment per index of `axis`, restricted to the entries valid in both tables -- a ragged
  batch of small problems (at most cameras x boards entries each) on the device.  The batch is cut out of the tables with ONE
  boolean selection (the per-index `np.take` of the first version cost 0.7 s for the 1000 frames of a 16 x 1000 x 5 table --
  more than the device work of the whole initialisation)."""
  n = table1.valid.shape[axis]
  v = np.moveaxis(np.asarray(table1.valid) & np.asarray(table2.valid), axis, 0).reshape(n, -1)      # [n, entries]
  p1 = np.moveaxis(np.asarray(table1.poses), axis, 0).reshape(n, -1, 4, 4)
  p2 = np.moveaxis(np.asarray(table2.poses), axis, 0).reshape(n, -1, 4, 4)
  poses, valid, _ = align_transforms_robust_ragged(p1[v], p2[v], v.sum(axis=1), None, invert=inv)   # (k, entry) order
  return Table.create(poses=poses, valid=valid)


def initialise_poses(pose_table, camera_poses=None):
  """tables.py:353-377: camera / board / rig-pose tables from the per-view board poses [C, F, B]."""
  camera = estimate_relative_poses(pose_table, axis=0)
  if camera_poses is not None:
    camera = Table.create(poses=np.asarray(camera_poses, dtype=np.float64), valid=np.ones(len(camera_poses), dtype=bool))
  board = estimate_relative_poses_inv(pose_table, axis=2)
  binv = inverse(board)
  # cam @ rig @ board = pose  ->  cam @ rig = board_relative = pose @ board^-1, then one alignment per frame between the camera
  # poses (expanded over frames and boards, tables.py:366-372) and board_relative over the entries valid in both:
  # `relative_between_n(expanded, board_relative, axis=1, inv=True)`.  Only those entries are ever read, so only they are formed
  # (the two full [C, F, B, 4, 4] tables, the broadcast copy and two boolean selections were 14 ms of host time at 16 x 1000 x 5,
  # where 15 455 of the 80 000 entries take part), in the order the reference's selection gives them: (frame | camera, board).
  poses = np.asarray(pose_table.poses, dtype=np.float64)
  C_, F, B = poses.shape[:3]
  v = (np.asarray(pose_table.valid) & binv.valid[None, None] & np.asarray(camera.valid)[:, None, None])
  vf = np.moveaxis(v, 1, 0).reshape(F, C_ * B)
  f_idx, e_idx = np.nonzero(vf)
  c_idx, b_idx = e_idx // B, e_idx % B
  p1 = np.asarray(camera.poses, dtype=np.float64)[c_idx]
  p2 = poses[c_idx, f_idx, b_idx] @ binv.poses[b_idx]
  rig, rig_valid, _ = align_transforms_robust_ragged(p1, p2, vf.sum(axis=1), None, invert=True)
  times = Table.create(poses=rig, valid=rig_valid)
  return struct(times=times, camera=camera, board=board)


def make_point_table(detections, boards):
  """tables.py:68-81: ragged per-image detections (corners [k, 2], ids [k]) -> dense table [C, F, B, P] (+ mask).

  dtype: the reference fills one array per image with `values.dtype` (fill_sparse, tables.py:15-21) and stacks them
  (make_nd_table -> Table.stack -> np.stack), so the table carries numpy's PROMOTION of all per-image corner dtypes: float32
  when every detection is float32 (cv2.aruco), float64 as soon as one image contributes float64 corners -- e.g. an empty
  detection built as np.zeros([0, 2]).  One scatter for the whole table instead of one per image."""
  num_points = int(np.max([b.num_points for b in boards]))
  C_, F, B = len(detections), len(detections[0]), len(detections[0][0])
  flat = [detections[c][f][b] for c in range(C_) for f in range(F) for b in range(B)]
  corners = [np.asarray(d.corners) for d in flat]
  ids = [np.asarray(d.ids, dtype=np.int64).reshape(-1) for d in flat]
  dtype = np.result_type(*[a.dtype for a in corners])
  counts = np.array([i.size for i in ids], dtype=np.int64)
  points = np.zeros((C_ * F * B, num_points, 2), dtype=dtype)
  valid = np.zeros((C_ * F * B, num_points), dtype=bool)
  if counts.sum() > 0:
    image = np.repeat(np.arange(C_ * F * B), counts)
    point = np.concatenate(ids)
    points[image, point] = np.concatenate([a.reshape(-1, 2) for a, i in zip(corners, ids) if i.size]).astype(dtype, copy=False)
    valid[image, point] = True
  return Table.create(points=points.reshape(C_, F, B, num_points, 2), valid=valid.reshape(C_, F, B, num_points))
