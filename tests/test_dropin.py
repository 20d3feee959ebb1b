"""The drop-in itself (multical_amd/dropin.py): patching Calibration.bundle_adjust of multical
(optimization/calibration.py:199-212) so that Workspace.calibrate (workspace.py:228-247), Calibration.adjust_outliers
(calibration.py:254-268) and HandEyeCalibration.bundle_adjust (optimization/hand_eye.py:73-75) reach the HIP back-end.

CPU (build box, reference importable): install() / uninstall() patch and restore the REAL reference class; the patched
method lowers the reference's own objects and reaches mcba_create, which fails loudly without a GPU (no CPU fallback).
GPU: the same patch applied to a module exposing the mirror Calibration runs the reference-shaped call chains end to end
and reproduces the reference's golden results.
"""
import types

import numpy as np
import pytest

from multical_amd import dropin, synthetic, calibration
from util import load_golden, mirror, gpu_available


@pytest.mark.needs_reference
def test_install_patches_and_uninstall_restores_the_reference_class():
  from oracle import refload, build_reference
  refload.load()
  import multical.optimization.calibration as refcal
  original = refcal.Calibration.bundle_adjust
  try:
    cls = dropin.install()
    assert cls is refcal.Calibration
    assert refcal.Calibration.bundle_adjust is dropin.bundle_adjust
    assert refcal.Calibration._scipy_bundle_adjust is original
    dropin.install()                                   # idempotent: the saved original is not overwritten by the patch
    assert refcal.Calibration._scipy_bundle_adjust is original
    # signature of the reference method is kept (calibration.py:199)
    import inspect
    assert list(inspect.signature(dropin.bundle_adjust).parameters) == list(inspect.signature(original).parameters)
    if not gpu_available():
      # the patched method lowers the REAL reference objects and reaches the C ABI; without a GPU it must fail loudly
      rig = synthetic.make_rig("tiny_handeye")
      ref_calib, _ = build_reference.reference_calibration(rig)
      with pytest.raises(RuntimeError, match="no HIP device|GPU-only"):
        ref_calib.bundle_adjust()
      # ... also through the reference's own callers: adjust_outliers and HandEyeCalibration.bundle_adjust
      from multical.optimization.hand_eye import HandEyeCalibration
      he = HandEyeCalibration(ref_calib, np.linalg.inv(rig.init.hand_eye.base_wrt_gripper), rig.init.rig)
      with pytest.raises(RuntimeError, match="no HIP device|GPU-only"):
        he.bundle_adjust()
      with pytest.raises(RuntimeError, match="no HIP device|GPU-only"):
        ref_calib.adjust_outliers(num_adjustments=1, select_outliers=refcal.select_threshold(0.75, 5.0))
  finally:
    dropin.uninstall()
  assert refcal.Calibration.bundle_adjust is original
  assert not hasattr(refcal.Calibration, "_scipy_bundle_adjust")


@pytest.mark.needs_reference
def test_install_from_env(monkeypatch):
  from oracle import refload
  refload.load()
  import multical.optimization.calibration as refcal
  original = refcal.Calibration.bundle_adjust
  monkeypatch.setenv("MULTICAL_BACKEND", "scipy")
  assert dropin.install_from_env() is None and refcal.Calibration.bundle_adjust is original
  monkeypatch.setenv("MULTICAL_BACKEND", "hip")
  try:
    assert dropin.install_from_env() is refcal.Calibration
    assert refcal.Calibration.bundle_adjust is dropin.bundle_adjust
  finally:
    dropin.uninstall()
  assert refcal.Calibration.bundle_adjust is original
  monkeypatch.setenv("MULTICAL_BACKEND", "hip-native")
  try:
    assert dropin.install_from_env() is refcal.Calibration
    assert refcal.Calibration.bundle_adjust is dropin.bundle_adjust_native
  finally:
    dropin.uninstall()
  monkeypatch.setenv("MULTICAL_BACKEND", "hip-lsmr")
  try:
    assert dropin.install_from_env() is refcal.Calibration
    assert refcal.Calibration.bundle_adjust is dropin.bundle_adjust      # the default: the reference's end point
  finally:
    dropin.uninstall()
  monkeypatch.setenv("MULTICAL_BACKEND", "hip-scipy")
  try:
    assert dropin.install_from_env() is refcal.Calibration
    assert refcal.Calibration.bundle_adjust is dropin.bundle_adjust_scipy
    assert refcal.Calibration._scipy_bundle_adjust is original
  finally:
    dropin.uninstall()
  assert refcal.Calibration.bundle_adjust is original


@pytest.mark.needs_reference
def test_scipy_mode_keeps_the_signature_and_fails_loudly_without_a_gpu():
  """dropin.install(mode="scipy"): the reference's own least_squares call with mcba_residuals + mcba_jacobian plugged in
  (calibration.py:208-210).  Same signature; on the GPU-less build box it reaches mcba_create on the REAL reference objects
  and fails there -- it never falls back to the reference's CPU evaluate."""
  import inspect
  from oracle import refload, build_reference
  refload.load()
  import multical.optimization.calibration as refcal
  original = refcal.Calibration.bundle_adjust
  with pytest.raises(ValueError, match="unknown mode"):
    dropin.install(mode="cpu")
  assert refcal.Calibration.bundle_adjust is original
  try:
    dropin.install(mode="scipy")
    assert refcal.Calibration.bundle_adjust is dropin.bundle_adjust_scipy
    assert list(inspect.signature(dropin.bundle_adjust_scipy).parameters) == list(inspect.signature(original).parameters)
    if not gpu_available():
      rig = synthetic.make_rig("tiny_handeye")
      ref_calib, _ = build_reference.reference_calibration(rig)
      with pytest.raises(RuntimeError, match="no HIP device|GPU-only"):
        ref_calib.bundle_adjust()
  finally:
    dropin.uninstall()
  assert refcal.Calibration.bundle_adjust is original


def test_solver_selection_of_the_mirror_is_validated():
  assert calibration.get_solver() in calibration.SOLVERS
  prev = calibration.set_solver("scipy")
  try:
    assert calibration.get_solver() == "scipy"
    with pytest.raises(ValueError, match="unknown solver"):
      calibration.set_solver("cpu")
    assert calibration.get_solver() == "scipy"
  finally:
    calibration.set_solver(prev)
  rig = synthetic.make_rig("tiny_handeye")
  with pytest.raises(ValueError, match="unknown solver"):
    mirror(rig).bundle_adjust(solver="cg")


class _PlainCalibration(calibration.Calibration):
  """The mirror Calibration with a bundle_adjust that must never run: stands in for the scipy method that install()
  replaces (the real reference class is not importable on the GPU box)."""

  def bundle_adjust(self, tolerance=1e-4, f_scale=1.0, max_iterations=100, loss='linear'):
    raise AssertionError("the un-patched bundle_adjust was called")

  def copy(self, **k):
    d = self.__getstate__()
    d.update(k)
    return _PlainCalibration(**d)


def _as_plain(c):
  return _PlainCalibration(**c.__getstate__())


class _HandEyeCalibration(object):
  """Shape of multical.optimization.hand_eye.HandEyeCalibration (hand_eye.py:13-20,73-79,93-100): holds a Calibration
  with a HandEye motion model and forwards bundle_adjust / adjust_outliers to it through copy()."""

  def __init__(self, calib, gripper_wrt_base, world_wrt_camera):
    self.calib, self.gripper_wrt_base, self.world_wrt_camera = calib, gripper_wrt_base, world_wrt_camera

  def bundle_adjust(self):
    return self.copy(calib=self.calib.bundle_adjust())

  def adjust_outliers(self, **kwargs):
    return self.copy(calib=self.calib.adjust_outliers(**kwargs))

  def copy(self, **k):
    d = dict(gripper_wrt_base=self.gripper_wrt_base, world_wrt_camera=self.world_wrt_camera, calib=self.calib)
    d.update(k)
    return self.__class__(**d)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["lsmr", "native"])
def test_dropin_runs_the_reference_call_chains_on_the_gpu(mode):
  """install() (default mode "lsmr": the reference's end point) and install(mode="native") (exact steps: the converged optimum)
  on fixtures whose reference end point and converged optimum agree to 1e-6 px."""
  mod = types.SimpleNamespace(Calibration=_PlainCalibration)
  try:
    assert (dropin.install(calibration_module=mod) if mode == "lsmr" else
            dropin.install(calibration_module=mod, mode=mode)) is _PlainCalibration
    assert _PlainCalibration.bundle_adjust is (dropin.bundle_adjust if mode == "lsmr" else dropin.bundle_adjust_native)
    # Calibration.bundle_adjust + adjust_outliers (what Workspace.calibrate drives) against the cfg1 reference golden
    g, rig = load_golden("cfg1")
    c = _as_plain(mirror(rig))
    x_before = c.param_vec.copy()
    out = c.bundle_adjust()
    assert isinstance(out, _PlainCalibration) and out is not c
    assert np.array_equal(c.param_vec, x_before)                                   # pure function of self
    assert abs(calibration.error_stats(out.reprojection_error).rms - float(g["ba_rms"])) < 1e-6
    ao = c.adjust_outliers(num_adjustments=3, select_outliers=calibration.select_threshold(0.75, 5.0), loss='linear',
                           tolerance=1e-4)
    assert np.array_equal(ao.inliers, g["ao_inliers"])
    assert abs(calibration.error_stats(ao.reprojection_error).rms - float(g["ao_rms"])) < 1e-6
    assert abs(calibration.error_stats(ao.reprojection_inliers).rms - float(g["ao_rms_inliers"])) < 1e-6
    # HandEyeCalibration.bundle_adjust / adjust_outliers (optimization/hand_eye.py:73-79)
    g, rig = load_golden("tiny_handeye")
    he = _HandEyeCalibration(_as_plain(mirror(rig)), np.linalg.inv(rig.init.hand_eye.base_wrt_gripper), rig.init.rig)
    he2 = he.bundle_adjust()
    assert isinstance(he2, _HandEyeCalibration) and he2.calib is not he.calib
    assert abs(calibration.error_stats(he2.calib.reprojection_error).rms - float(g["ba_rms"])) < 1e-6
    he3 = he.adjust_outliers(num_adjustments=3, select_outliers=calibration.select_threshold(0.75, 5.0))
    assert np.array_equal(he3.calib.inliers, g["ao_inliers"])
    assert abs(calibration.error_stats(he3.calib.reprojection_inliers).rms -
               float(g["ao_tight_rms_inliers" if mode == "native" else "ao_rms_inliers"])) < 1e-6
  finally:
    dropin.uninstall(calibration_module=mod)
  with pytest.raises(AssertionError, match="un-patched"):
    _as_plain(mirror(rig)).bundle_adjust()


@pytest.mark.gpu
def test_dropin_scipy_mode_reproduces_the_reference_end_point_on_the_gpu():
  """dropin.install(mode="scipy") on the reference-shaped call chains: bundle_adjust and the complete adjust_outliers loop land
  on the reference's END points (1e-6 px; cfg1 = BASELINE configs[0], hand-eye) with identical inlier masks."""
  mod = types.SimpleNamespace(Calibration=_PlainCalibration)
  try:
    assert dropin.install(calibration_module=mod, mode="scipy") is _PlainCalibration
    g, rig = load_golden("cfg1")
    c = _as_plain(mirror(rig))
    x_before = c.param_vec.copy()
    out = c.bundle_adjust()
    assert isinstance(out, _PlainCalibration) and np.array_equal(c.param_vec, x_before)
    assert abs(calibration.error_stats(out.reprojection_error).rms - float(g["ba_rms"])) < 1e-6
    ao = c.adjust_outliers(num_adjustments=3, select_outliers=calibration.select_threshold(0.75, 5.0), loss='linear',
                           tolerance=1e-4)
    assert np.array_equal(ao.inliers, g["ao_inliers"])
    assert abs(calibration.error_stats(ao.reprojection_inliers).rms - float(g["ao_rms_inliers"])) < 1e-6
    g, rig = load_golden("tiny_handeye")
    he = _HandEyeCalibration(_as_plain(mirror(rig)), np.linalg.inv(rig.init.hand_eye.base_wrt_gripper), rig.init.rig)
    he2 = he.bundle_adjust()
    assert abs(calibration.error_stats(he2.calib.reprojection_error).rms - float(g["ba_rms"])) < 1e-6
  finally:
    dropin.uninstall(calibration_module=mod)


@pytest.mark.gpu
def test_dropin_log_table_has_the_scipy_format():
  """scipy's verbose=2 table reaches the "calibration" logger like the reference's redirect_stdout(LogWriter.info())
  (calibration.py:208, io/logging.py:53-68): header, one row per iteration, termination message, summary."""
  import logging
  lines = []

  class Grab(logging.Handler):
    def emit(self, rec):
      lines.append(rec.getMessage())

  mod = types.SimpleNamespace(Calibration=_PlainCalibration)
  log = logging.getLogger("calibration")
  hd = Grab()
  log.addHandler(hd)
  log.setLevel(logging.INFO)
  try:
    dropin.install(calibration_module=mod)
    g, rig = load_golden("cfg1")
    _as_plain(mirror(rig)).bundle_adjust()
  finally:
    dropin.uninstall(calibration_module=mod)
    log.removeHandler(hd)
  ref = [l for l in str(g["ba_log"]).splitlines() if l.strip()]
  header = [l for l in lines if "Iteration" in l]
  assert len(header) == 1 and header[0].split() == ref[0].split()
  rows = [l for l in lines if l.strip() and l.split()[0].isdigit()]
  ref_rows = [l for l in ref if l.split()[0].isdigit()]
  assert len(rows) == len(ref_rows)
  for a, b in zip(rows, ref_rows):            # same iteration / nfev counters, costs to the printed precision
    ta, tb = a.split(), b.split()
    assert ta[:2] == tb[:2] and ta[2] == tb[2]
  assert any("`ftol` termination condition is satisfied." in l for l in lines)
  assert any(l.startswith("Function evaluations") for l in lines)
