"""Drop-in for an installed multical: route Calibration.bundle_adjust (optimization/calibration.py:199-212) through
the MI355X back-end while keeping every other line of multical untouched.

    import multical_amd.dropin as dropin
    dropin.install()                # mode="lsmr": the reference's own trajectory and END POINT, everything on the device;
                                    # or set MULTICAL_BACKEND=hip and call dropin.install_from_env()
    dropin.install(mode="native")   # exact Schur / Cholesky steps: the converged optimum, ~100 x faster (MULTICAL_BACKEND=hip-native)
    dropin.install(mode="scipy")    # the reference's own scipy driver on the device fun + jac (MULTICAL_BACKEND=hip-scipy)

The DEFAULT reproduces the reference's answer (final RMS within max(1e-6 px, the reference's own reproducibility), identical
nfev / status); `native` does NOT -- it walks the same valley further down and may end tens of pixels of principal point away
from the reference's end point on weakly determined rigs (profiles/parity_table.md).

After `install()`, `Workspace.calibrate` (workspace.py:228-247), `Calibration.adjust_outliers` (calibration.py:254-268)
and `HandEyeCalibration.bundle_adjust` (optimization/hand_eye.py:73-75) reach the GPU through the unchanged call chain.
`bundle_adjust` keeps the reference's signature, returns `self.with_param_vec(res.x)` exactly like the reference, does
not mutate `self`, stores nothing on the Calibration (pickling is unaffected, calibration.py:222-226) and writes the
scipy-style iteration table to the same "calibration" logger.  See INTEGRATION.md.
"""
import os

import numpy as np

from .backend import Handle, lower

_HEADER = "{:^15}{:^15}{:^15}{:^15}{:^15}{:^15}".format("Iteration", "Total nfev", "Cost", "Cost reduction", "Step norm",
                                                        "Optimality")


def _format_row(it, nfev, cost, red, step, opt):
  red_s = " " * 15 if np.isnan(red) else f"{red:^15.2e}"
  step_s = " " * 15 if np.isnan(step) else f"{step:^15.2e}"
  return f"{it:^15}{nfev:^15}{cost:^15.4e}{red_s}{step_s}{opt:^15.2e}"


def bundle_adjust(self, tolerance=1e-4, f_scale=1.0, max_iterations=100, loss='linear'):
  """Replacement body of multical.optimization.calibration.Calibration.bundle_adjust (same signature), the DEFAULT of install():
  mcba_solve with scipy's OWN trust-region step (tr_solver = lsmr) -- scipy's TRF driver and its LSMR restated line by line, the
  two Jacobian products as HIP kernels: the reference's trajectory and END POINT, everything on the device."""
  return _bundle_adjust_device(self, tolerance, f_scale, max_iterations, loss, 'lsmr')


def bundle_adjust_native(self, tolerance=1e-4, f_scale=1.0, max_iterations=100, loss='linear'):
  """Opt-in replacement body (install(mode="native")): exact regularised Gauss-Newton steps from the Schur-reduced normal
  equations.  Ends at the CONVERGED optimum -- at or below the reference's cost, NOT at the reference's end point
  (DESIGN.md section 2)."""
  return _bundle_adjust_device(self, tolerance, f_scale, max_iterations, loss, 'exact')


def _bundle_adjust_device(self, tolerance, f_scale, max_iterations, loss, _tr_solver):
  import logging
  log = logging.getLogger("calibration")        # multical/io/logging.py:11
  rows = []
  with Handle(lower(self)) as h:
    h.set_log(lambda *a: rows.append(_format_row(*a)))
    res = h.solve(self.param_vec, tolerance=tolerance, f_scale=f_scale, max_iterations=max_iterations, loss=loss,
                  tr_solver=_tr_solver)
  log.info(_HEADER)
  for r in rows:
    log.info(r)
  log.info(res.message)
  log.info(f"Function evaluations {res.nfev}, initial cost {res.initial_cost:.4e}, final cost {res.cost:.4e}, "
           f"first-order optimality {res.optimality:.2e}.")
  return self.with_param_vec(res.x)


def bundle_adjust_scipy(self, tolerance=1e-4, f_scale=1.0, max_iterations=100, loss='linear'):
  """Replacement body of Calibration.bundle_adjust that keeps the reference's OWN solver call (calibration.py:208-210) and
  only swaps the functions it calls: `evaluate` -> mcba_residuals, `jac_sparsity=S` (finite differences) -> `jac` =
  mcba_jacobian (analytic, same pattern S).  Same driver, same options, same stdout redirection: the reference's trajectory
  and END POINT (final RMS within 1e-6 px, identical nfev / status where the reference reproduces itself to that level:
  profiles/parity_table.md), with scipy's LSMR on the host."""
  import contextlib
  try:
    from multical.io.logging import LogWriter       # the reference's own writer when multical is importable
  except Exception:
    from .calibration import LogWriter
  with Handle(lower(self)) as h:
    with contextlib.redirect_stdout(LogWriter.info()):
      res = h.solve_scipy(self.param_vec, tolerance=tolerance, f_scale=f_scale, max_iterations=max_iterations, loss=loss,
                          verbose=2)
  return self.with_param_vec(res.x)


bundle_adjust_lsmr = bundle_adjust     # (round-4 name)


MODES = {"lsmr": bundle_adjust, "native": bundle_adjust_native, "scipy": bundle_adjust_scipy}


def _reprojection_tables(self):
  """(errors, mask) of tables.reprojection_error(self.reprojected, self.point_table) on the device (tables.py:244-249)."""
  with Handle(lower(self)) as h:
    return h.reprojection_error(self.param_vec)


def install(calibration_module=None, patch_errors=False, mode="lsmr"):
  """Patch `Calibration.bundle_adjust` of multical (or of the given module object).  Returns the patched class.
  mode = "lsmr" (default): mcba_solve with scipy's TRF + LSMR step on the device -- the reference's end point.
  mode = "native": mcba_solve with exact steps (fastest; the converged optimum, not the reference's end point).
  mode = "scipy": the reference's own scipy driver on the device residuals + analytic Jacobian (host-side LSMR).
  patch_errors=True additionally evaluates `reprojection_error` / `reject_outliers` on the device."""
  if mode not in MODES:
    raise ValueError(f"unknown mode {mode!r}, options are {sorted(MODES)}")
  import logging
  logging.getLogger("calibration").info(
    "multical_amd: Calibration.bundle_adjust -> HIP back-end, solver '%s' (%s)", mode,
    {"lsmr": "scipy's TRF + LSMR steps restated on the device: the reference's end point; ~100x the time of 'native'",
     "native": "exact Schur / Cholesky steps: the converged optimum, not the reference's end point",
     "scipy": "scipy's own driver on the device residuals + analytic Jacobian: LSMR on the host"}[mode])
  if calibration_module is None:
    import multical.optimization.calibration as calibration_module
  cls = calibration_module.Calibration
  if not hasattr(cls, "_scipy_bundle_adjust"):
    cls._scipy_bundle_adjust = cls.bundle_adjust
  cls.bundle_adjust = MODES[mode]
  if patch_errors:
    from functools import cached_property

    def reprojection_error(self):
      err, mask = _reprojection_tables(self)
      return err[mask]

    prop = cached_property(reprojection_error)
    prop.__set_name__(cls, "reprojection_error")
    cls.reprojection_error = prop
  return cls


def uninstall(calibration_module=None):
  if calibration_module is None:
    import multical.optimization.calibration as calibration_module
  cls = calibration_module.Calibration
  if hasattr(cls, "_scipy_bundle_adjust"):
    cls.bundle_adjust = cls._scipy_bundle_adjust
    del cls._scipy_bundle_adjust


def install_from_env():
  """MULTICAL_BACKEND=hip (or hip-lsmr) -> install(): the reference's end point on the device; hip-native -> install(mode="native");
  hip-scipy -> install(mode="scipy"); anything else keeps the reference's scipy path untouched."""
  backend = os.environ.get("MULTICAL_BACKEND", "scipy").lower().replace("_", "-")
  if backend in ("hip", "hip-lsmr"):
    return install()
  if backend == "hip-native":
    return install(mode="native")
  if backend == "hip-scipy":
    return install(mode="scipy")
  return None
