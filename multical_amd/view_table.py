"""Per-axis reprojection statistics of a calibration: the numeric part of the reference's GUI tables
(multical/interface/view_table.py:19-52 `masked_quantile`, `reprojection_statistics`, `reprojection_tables`; the Qt table
models around them are GUI and out of scope).

The inputs are the two things the HIP back-end produces for this consumer: `Calibration.projected` (the projection
WITHOUT the measured points -- rolling shutter: scan time iterated from the projected row; mcba_project_model) and
`Calibration.inliers`.  The statistics themselves are the reference's numpy reductions over the reference's axes, so a
calibration solved on the GPU fills the same tables `multical vis` shows.  Works on the mirror classes and, being duck-
typed, on the reference's own objects.
"""
import math

import numpy as np

from .structs import Table, struct

# view_table.py:39-40: the table axes are (camera, frame, board, point)
SUM_AXES = dict(overall=None, views=(2, 3), board_views=(3,), boards=(0, 1, 3), cameras=(1, 2, 3), frames=(0, 2, 3))


def masked_quantile(error, mask, quantiles, axis=None):
  """view_table.py:19-23"""
  error = error.copy()
  error[~mask] = math.nan
  return np.nanquantile(error, quantiles, axis=axis)


def reprojection_statistics(error, valid, inlier, axis=None):
  """view_table.py:26-37"""
  n = valid.sum(axis=axis)
  mse = np.square(error).sum(axis=axis) / np.maximum(n, 1)
  outliers = (valid & ~inlier).sum(axis=axis)
  mn, lq, median, uq, mx = masked_quantile(error, valid, [0, 0.25, 0.5, 0.75, 1.0], axis=axis)
  return Table.create(detected=n, outliers=outliers, mse=mse, rms=np.sqrt(mse), min=mn, lower_q=lq, median=median,
                      upper_q=uq, max=mx)


def table_reprojection_error(reprojected, point_table):
  """tables.reprojection_error (tables.py:244-249): per-slot error, zero outside the common mask"""
  valid = np.asarray(reprojected.valid) & np.asarray(point_table.valid)
  error = np.linalg.norm(np.asarray(point_table.points, dtype=np.float64) - np.asarray(reprojected.points), axis=-1)
  error[~valid] = 0
  return error, valid


def reprojection_tables(calib, inlier_only=False):
  """view_table.py:43-52"""
  points, valid = calib.point_table.points, calib.point_table.valid
  if inlier_only:
    valid = calib.inliers
  error, valid = table_reprojection_error(calib.projected, struct(points=points, valid=valid))
  return struct(**{k: reprojection_statistics(error, valid, calib.inliers, axis=axis) for k, axis in SUM_AXES.items()})
