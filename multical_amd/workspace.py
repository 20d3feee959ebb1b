"""Minimal mirror of multical.workspace.Workspace for the optimisation phase only (workspace.py:228-247).

Image loading, detection, intrinsic calibration, pose initialisation, export and the pickle checkpoint are upstream /
downstream of the hot path and stay with the reference; a Workspace here is seeded with an
initial Calibration and reproduces `calibrate`'s enable -> adjust_outliers sequence and argument mapping.
"""
from .calibration import Calibration, select_threshold


class Workspace(object):
  def __init__(self, initialisation=None, name="calibration"):
    self.name = name
    self.calibrations = {}
    if initialisation is not None:
      self.calibrations["initialisation"] = initialisation

  @property
  def initialisation(self) -> Calibration:
    return self.calibrations["initialisation"]

  @property
  def latest_calibration(self) -> Calibration:
    return list(self.calibrations.values())[-1]

  def calibrate(self, name="calibration", camera_poses=True, motion=True, board_poses=True, cameras=False, boards=False,
                loss='linear', tolerance=1e-4, num_adjustments=3, quantile=0.75, auto_scale=None,
                outlier_threshold=5.0) -> Calibration:
    calib = self.latest_calibration.enable(cameras=cameras, boards=boards, camera_poses=camera_poses, motion=motion,
                                           board_poses=board_poses)
    calib = calib.adjust_outliers(
      loss=loss, tolerance=tolerance, num_adjustments=num_adjustments,
      select_outliers=select_threshold(quantile=quantile, factor=outlier_threshold),
      select_scale=select_threshold(quantile=quantile, factor=auto_scale) if auto_scale is not None else None)
    self.calibrations[name] = calib
    return calib
