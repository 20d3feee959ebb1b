"""Cameras as parameter blocks (multical/camera.py:27-171, multical/camera_fisheye.py:28-160).

Only the bundle-adjustment face of the reference classes is mirrored: intrinsic matrix + distortion <-> parameter
vector [fx fy | cx cy | skew | dist].  `project` lives in the HIP kernels (csrc/mcba_math.h: project_point) -- there is
no OpenCV in this package.  Intrinsic calibration (cv2.calibrateCamera*) is upstream of the hot path and out of scope.
"""
from functools import cached_property
import numpy as np
from .parameters import Parameters
from .structs import struct

DIST_SIZES = dict(standard=(4, 5), rational=(8,), thin_prism=(12,), tilted=(14,))


class Camera(Parameters):
  model_names = ("standard", "rational", "tilted", "thin_prism")

  def __init__(self, image_size, intrinsic, dist, model='standard', fix_aspect=False, has_skew=False):
    assert model in self.model_names, f"unknown camera model {model} options are {list(self.model_names)}"
    self.model = model
    self.image_size = tuple(image_size)
    self.intrinsic = np.asarray(intrinsic, dtype=np.float64)
    self.dist = np.zeros(5) if dist is None else np.asarray(dist, dtype=np.float64)
    self.fix_aspect = fix_aspect
    self.has_skew = has_skew

  @property
  def focal_length(self):
    return np.array([self.intrinsic[0, 0], self.intrinsic[1, 1]])

  @property
  def principle_point(self):
    return np.array([self.intrinsic[0, 2], self.intrinsic[1, 2]])

  @property
  def skew(self):
    return self.intrinsic[0, 1] if self.has_skew else 0.0

  @cached_property
  def params(self):
    f = self.focal_length
    if self.fix_aspect:
      f = np.array([f.mean(), f.mean()])
    return struct(focal_length=f, principle_point=self.principle_point, skew=np.array([self.skew]),
                  dist=np.asarray(self.dist))

  def with_params(self, params):
    f = params.focal_length
    fx, fy = f if not self.fix_aspect else (f[0], f[0])
    px, py = params.principle_point
    skew, = params.skew
    intrinsic = np.array([[fx, skew, px], [0, fy, py], [0, 0, 1]])
    return self.copy(intrinsic=intrinsic, dist=params.dist)

  def scale_image(self, factor):
    intrinsic = self.intrinsic.copy()
    intrinsic[:2] *= factor
    return self.copy(intrinsic=intrinsic)

  def __getstate__(self):
    return dict(image_size=self.image_size, intrinsic=self.intrinsic, dist=self.dist, fix_aspect=self.fix_aspect,
                has_skew=self.has_skew, model=self.model)

  def __setstate__(self, d):
    self.__dict__.update(d)

  def copy(self, **k):
    d = self.__getstate__()
    d.update(k)
    return self.__class__(**d)

  def __repr__(self):
    return f"{type(self).__name__}(image_size={self.image_size}, intrinsic={self.intrinsic.tolist()}, dist={self.dist.tolist()})"


class CameraFisheye(Camera):
  """Kannala-Brandt fisheye (camera_fisheye.py:28): same parameter block, cv2.fisheye.projectPoints forward model."""
  model_names = ("standard", "fix_k1", "fix_k2", "fix_k3", "fix_k4")
