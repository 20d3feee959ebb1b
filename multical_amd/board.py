"""Calibration targets as data (multical/board/): the [P_b, 3] point table and its optional adjustment block.

Detection / drawing / pose estimation (cv2.aruco, apriltags) are image processing upstream of the hot path and are
out of scope; the geometry generators match board/charuco.py:56-58 (float32 chessboard corners) and
board/aprilgrid.py:78-83 + aprilgrid_detector.py:44-55.
"""
from functools import cached_property
import numpy as np
from .parameters import Parameters
from .structs import choose
from . import synthetic


class Board(Parameters):
  def __init__(self, points, adjusted_points=None, name=None):
    self._points = np.asarray(points)
    self.adjusted_points = choose(adjusted_points, self._points)
    self.name = name

  @property
  def points(self):
    return self._points

  @property
  def num_points(self):
    return len(self._points)

  @property
  def ids(self):
    return np.arange(self.num_points)

  @cached_property
  def params(self):
    return np.asarray(self.adjusted_points)

  def with_params(self, params):
    return self.copy(adjusted_points=params)

  def __getstate__(self):
    return dict(points=self._points, adjusted_points=self.adjusted_points, name=self.name)

  def __setstate__(self, d):
    self.__init__(**d)

  def copy(self, **k):
    d = self.__getstate__()
    d.update(k)
    return self.__class__(**d)


class CharucoBoard(Board):
  def __init__(self, size=None, square_length=None, marker_length=None, adjusted_points=None, points=None, name=None,
               **ignored):
    self.size = None if size is None else tuple(size)
    self.square_length, self.marker_length = square_length, marker_length
    pts = points if points is not None else synthetic.charuco_points(self.size, square_length)
    super().__init__(pts, adjusted_points, name)

  def __getstate__(self):
    return dict(size=self.size, square_length=self.square_length, marker_length=self.marker_length,
                adjusted_points=self.adjusted_points, points=self._points, name=self.name)


class AprilGrid(Board):
  def __init__(self, size=None, tag_length=None, tag_spacing=None, adjusted_points=None, points=None, name=None,
               **ignored):
    self.size = None if size is None else tuple(size)
    self.tag_length, self.tag_spacing = tag_length, tag_spacing
    pts = points if points is not None else synthetic.aprilgrid_points(self.size, tag_length, tag_spacing)
    super().__init__(pts, adjusted_points, name)

  def __getstate__(self):
    return dict(size=self.size, tag_length=self.tag_length, tag_spacing=self.tag_spacing,
                adjusted_points=self.adjusted_points, points=self._points, name=self.name)
