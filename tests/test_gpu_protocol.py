"""GPU (MI355X): solver-level parity protocol of SURVEY.md section 7 against the reference's ACTUAL end points.

The fixtures hold, per case, outputs of the unmodified reference (oracle/make_golden.py):
  ba_rms / ao_rms*      the reference's default-tolerance end point (bundle_adjust / adjust_outliers)
  *_pert_*              the same reference call repeated with N(0, 1e-12 px) noise on its residual function: the
                        reference's own end point moves by up to 1e-3 px (forward differences with h ~ 1.5e-8 amplify the
                        noise 1e8-fold, LSMR-truncated steps + ftol = 1e-4 stop before convergence).  `spread` below is
                        max |perturbed - unperturbed| over those runs: the resolution to which "the reference's end
                        point" is defined at all.
  *_tight_*             converged optimum of the reference's own residual function (exact trust-region solver on 3-point
                        differences of the reference's `evaluate`)

Protocol
  (B)  the HIP `fun` (+ analytic `jac`, or scipy's own finite differences with the reference's sparsity) under the
       reference's own driver scipy.optimize.least_squares(method='trf', x_scale='jac', ...) exactly as
       optimization/calibration.py:209-210: |RMS - ba_rms| <= max(1e-6 px, 3 * spread), i.e. 1e-6 px wherever the
       reference's end point is defined to 1e-6 px.
  (C)  converged optima: scipy-driven exact solve on the HIP functions AND the native solver at tight tolerance are
       within 1e-6 px of the reference's converged optimum.
  (N)  native solver at the reference's default tolerance: final RMS between the converged optimum and the reference's
       end point (+ its spread): never worse than the reference, never "better" than its optimum.
"""
import json

import numpy as np
import pytest

from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
from oracle import restate
from util import load_golden, mirror, oracle, rel_col_error, GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _exact_step_solver():
  """This module pins the properties of the EXACT-step solver (solver = "native": converged optima, fused outlier loop, timing
  paths) unless a test names another one; the product default is "lsmr" (tests/test_gpu_lsmr.py, test_device_lsmr_mode_*)."""
  prev = calibration.set_solver("native")
  yield
  calibration.set_solver(prev)

PROTOCOL_CASES = ["tiny", "tiny_rolling", "tiny_fisheye", "tiny_handeye", "tiny_rational", "tiny_thin_prism",
                  "tiny_tilted", "tiny_edge", "tiny_fixintr", "tiny_pin4", "cfg1", "tiny_softl1", "tiny_huber",
                  "tiny_boards", "tiny_bigboard", "tiny_fishmix"]
# fixtures whose reference end point is reproducible to better than 1e-6 px (spread < 3e-7): plain 1e-6 assertion
WELL_DEFINED = ["tiny_handeye", "tiny_fixintr", "cfg1", "tiny_huber"]
# over-parameterised distortion models on 8 frames: a flat valley that neither the reference's own tight polish nor any
# other solver bottoms out in 400 evaluations (ba_tight_status 0); no converged optimum to compare with
FLAT_VALLEY = ["tiny_rational", "tiny_thin_prism", "tiny_tilted"]
# fixtures on which the reference's own driver with the HIP fun + jac reproduces the reference's end point to 1e-6 px
B_TIGHT = WELL_DEFINED + ["tiny_thin_prism"]


def spread_of(g, key="ba"):
  return float(np.abs(g[f"{key}_pert_rms"] - g[f"{key}_rms"]).max())


def rms_of(h, x):
  e, v = h.reprojection_error(x)
  return float(np.sqrt(np.mean(e[v] ** 2)))


def scipy_args(g):
  kw = json.loads(str(g["ba_kwargs_json"]))
  return dict(x_scale='jac', ftol=kw.get('tolerance', 1e-4), max_nfev=kw.get('max_iterations', 100), method='trf',
              loss=kw.get('loss', 'linear'), f_scale=kw.get('f_scale', 1.0))


@pytest.mark.parametrize("name", PROTOCOL_CASES)
def test_protocol_b_scipy_driven_with_hip_fun_and_jac(name, record_property):
  """SURVEY 7 protocol (B): the reference's own solver call with the HIP residuals and the HIP analytic Jacobian."""
  from scipy.optimize import least_squares
  g, rig = load_golden(name)
  with Handle(mirror(rig)) as h:
    res = least_squares(h.residuals, g["x0"], jac=h.jacobian, **scipy_args(g))
    rms = rms_of(h, res.x)
  ref, spread = float(g["ba_rms"]), spread_of(g)
  # (B_TIGHT: measured 5e-14 ... 7e-8 px in profiles/parity_table.md; elsewhere the reference's own end point is only
  #  defined to its spread over ten perturbed re-runs)
  tol = 1e-6 if name in B_TIGHT else max(1e-6, 3 * spread)
  record_property("delta_rms_px", abs(rms - ref))
  record_property("nfev", (int(res.nfev), int(g["ba_nfev"])))
  assert res.status == int(g["ba_status"]) or name not in WELL_DEFINED
  assert abs(rms - ref) <= tol, (name, abs(rms - ref), spread, res.nfev, int(g["ba_nfev"]))
  if name in WELL_DEFINED:
    assert res.nfev == int(g["ba_nfev"])
    assert spread < 3e-7


@pytest.mark.parametrize("name", PROTOCOL_CASES)
def test_protocol_b_scipy_driven_with_hip_fun_and_reference_finite_differences(name):
  """the drop-in with the smallest change: only `evaluate` runs on the GPU, scipy differentiates it numerically with
  the reference's sparsity pattern (calibration.py:173-196) -- the reference's trajectory up to round-off."""
  from scipy.optimize import least_squares
  from scipy.sparse import csr_matrix
  g, rig = load_golden(name)
  S = csr_matrix(oracle(rig).sparsity_matrix)
  with Handle(mirror(rig)) as h:
    res = least_squares(h.residuals, g["x0"], jac_sparsity=S, **scipy_args(g))
    rms = rms_of(h, res.x)
  ref, spread = float(g["ba_rms"]), spread_of(g)
  tol = 1e-6 if name in WELL_DEFINED else max(1e-6, 3 * spread)
  assert abs(rms - ref) <= tol, (name, abs(rms - ref), spread)
  if name in WELL_DEFINED:
    assert res.nfev == int(g["ba_nfev"]) and res.status == int(g["ba_status"])


@pytest.mark.parametrize("name", [n for n in PROTOCOL_CASES if n not in FLAT_VALLEY])
def test_protocol_c_converged_optimum(name):
  """both the scipy-driven HIP functions (exact trust-region solver) and the native HIP solver, run to tight tolerance,
  end within 1e-6 px of the converged optimum of the reference's own residual function."""
  from scipy.optimize import least_squares
  g, rig = load_golden(name)
  kw = scipy_args(g)
  c = mirror(rig)
  tight = float(g["ba_tight_rms"])
  with Handle(c) as h:
    res = least_squares(h.residuals, g["ba_x_raw"], jac=lambda x: h.jacobian(x).toarray(), x_scale='jac', tr_solver='exact',
                        ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400, method='trf', loss=kw["loss"], f_scale=kw["f_scale"])
    assert abs(rms_of(h, res.x) - tight) < 1e-6
    assert res.cost == pytest.approx(float(g["ba_tight_cost"]), rel=1e-9)
    nat = h.solve(g["x0"], tolerance=1e-14, xtol=1e-14, gtol=1e-14, max_iterations=400, loss=kw["loss"],
                  f_scale=kw["f_scale"])
    assert abs(rms_of(h, nat.x) - tight) < 1e-6, (name, rms_of(h, nat.x) - tight)
    assert nat.cost == pytest.approx(float(g["ba_tight_cost"]), rel=1e-8)


@pytest.mark.parametrize("name", PROTOCOL_CASES)
def test_protocol_n_native_solver_between_optimum_and_reference(name):
  """mcba_solve at the reference's default tolerance: never above the reference's end point (+ its own spread), never
  below the converged optimum; on the well-defined fixtures that pins it to 1e-6 px of the reference."""
  g, rig = load_golden(name)
  kw = scipy_args(g)
  with Handle(mirror(rig)) as h:
    res = h.solve(g["x0"], tolerance=kw["ftol"], loss=kw["loss"], f_scale=kw["f_scale"], max_iterations=kw["max_nfev"])
    rms = rms_of(h, res.x)
  ref, spread = float(g["ba_rms"]), spread_of(g)
  # (robust losses: both solvers stop on ftol = 1e-4 at slightly different points of a flat approach; the cost is
  #  compared to a thousandth of ftol, the linear loss -- where the native solver always ends lower -- strictly)
  assert res.cost <= float(g["ba_cost"]) * (1 + (1e-9 if kw["loss"] == "linear" else 1e-6))
  if name in WELL_DEFINED and kw["loss"] == "linear":
    assert abs(rms - ref) < 1e-6 and res.nfev == int(g["ba_nfev"]) and res.status == int(g["ba_status"])
  if kw["loss"] == "linear":     # (RMS is the minimised quantity only for the linear loss)
    assert rms <= ref + max(1e-6, 3 * spread)
    if name not in FLAT_VALLEY:
      assert rms >= float(g["ba_tight_rms"]) - 1e-6


def test_arctan_loss_is_not_worse_than_the_reference():
  """arctan on data with gross outliers is non-convex and the reference stops after 5 evaluations on xtol (its end point
  moves by 16 px under 1e-12 px perturbations, `ba_pert_rms`): only the cost is comparable."""
  g, rig = load_golden("tiny_arctan")
  with Handle(mirror(rig)) as h:
    res = h.solve(g["x0"], loss='arctan', f_scale=3.0)
  assert res.cost <= float(g["ba_cost"]) * (1 + 1e-9)
  assert spread_of(g) > 1.0


@pytest.mark.parametrize("name", ["tiny_rolling", "tiny_fisheye", "tiny_tilted", "tiny_handeye", "tiny_pin4"])
def test_device_jacobian_against_three_point_differences_of_the_oracle(name):
  """h.jacobian (k_jacobian on the GPU) against 3-point differences of the ORACLE's evaluate (error ~1e-9 relative):
  2e-7 per column, every camera model / motion model."""
  from scipy.optimize._numdiff import approx_derivative, group_columns
  from scipy.sparse import csr_matrix
  g, rig = load_golden(name)
  oc = oracle(rig)
  S = csr_matrix(oc.sparsity_matrix)
  J3 = csr_matrix(approx_derivative(oc.evaluate, g["x0"], method='3-point', sparsity=(S, group_columns(S))))
  with Handle(mirror(rig)) as h:
    J = h.jacobian(g["x0"])
    # and the fused pass agrees with it: g = J^T f, diag = diag(J^T J)
    cost, grad, diag = h.normal_equations(g["x0"])
  assert rel_col_error(J, J3) < 2e-7
  r = oc.evaluate(g["x0"])
  assert np.abs(J3.T @ r - grad).max() <= 1e-6 * np.abs(grad).max()


# -----------------------------------------------------------------------------------------------------------------
# Workspace.calibrate(loss, auto_scale) (workspace.py:239-244, calibration.py:254-268)
# -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny_autoscale", "tiny_autoscale_huber"])
def test_auto_scale_through_workspace_calibrate(name):
  """select_scale = select_threshold(quantile, factor=auto_scale): the f_scale of every round is the device quantile x
  factor, the inlier masks after three rounds equal the reference's, and the final RMS is within the reference's own
  spread of the reference's end point."""
  import logging
  from multical_amd import Workspace
  g, rig = load_golden(name)
  kw = json.loads(str(g["ao_kwargs_json"]))
  lines = []

  class Grab(logging.Handler):
    def emit(self, rec):
      lines.append(rec.getMessage())

  log = logging.getLogger("calibration")
  hd = Grab()
  log.addHandler(hd)
  log.setLevel(logging.INFO)
  try:
    ws = Workspace(mirror(rig))
    out = ws.calibrate(cameras=rig.optimize["cameras"], camera_poses=rig.optimize["camera_poses"], loss=kw["loss"],
                       auto_scale=kw["auto_scale"])
  finally:
    log.removeHandler(hd)
  mine = [l for l in lines if l.startswith("Auto scaling")]
  ref = [l for l in str(g["ao_log"]).splitlines() if l.startswith("Auto scaling")]
  assert len(mine) == 3 and mine[0] == ref[-3]          # the first round starts from identical errors: identical f_scale
  assert np.array_equal(out.inliers, g["ao_inliers"]) or int((out.inliers != g["ao_inliers"]).sum()) <= int(g["ao_pert_mask_diff"].max())
  rms_inl = out.error_statistics(True).rms
  spread = float(np.abs(g["ao_pert_rms_inliers"] - g["ao_rms_inliers"]).max())
  # one-sided: with huber + rolling shutter the reference's LSMR rounds run into max_nfev = 100 (ao_nfev = [27 100 100])
  # and leave the inlier RMS at 0.68 px; the exact normal-equation solve converges to the noise level (0.28 px)
  assert rms_inl <= float(g["ao_rms_inliers"]) + max(1e-6, 3 * spread), (rms_inl, float(g["ao_rms_inliers"]), spread)
  if int(np.max(g["ao_nfev"])) < 100:      # the reference converged: two-sided, to the reference's own reproducibility
    assert abs(rms_inl - float(g["ao_rms_inliers"])) <= max(2e-4, 3 * spread)
  assert 0.2 < rms_inl < float(g["ao_rms_inliers"]) + 0.01


@pytest.mark.parametrize("name,loss,f_scale", [("tiny_softl1", "soft_l1", 1.5), ("tiny_huber", "huber", 2.0)])
def test_robust_loss_solve_matches_reference(name, loss, f_scale):
  """solve-level robust losses through the mirror entry point Calibration.bundle_adjust(loss, f_scale)."""
  g, rig = load_golden(name)
  out, res = mirror(rig).bundle_adjust(loss=loss, f_scale=f_scale, return_result=True)
  rms = calibration.error_stats(out.reprojection_error).rms
  # both solvers stop on ftol = 1e-4 at slightly different points of the same approach: the cost (the minimised quantity)
  # agrees to a hundredth of ftol; the RMS -- not what a robust loss minimises -- to 1e-4 px at default tolerance and to
  # 1e-6 px once both are converged (below).  The scipy-driven protocol (B) is within 1e-6 px at default tolerance.
  assert res.cost <= float(g["ba_cost"]) * (1 + 1e-6)
  # (soft_l1: the reference stops 2e-3 px short of its own converged optimum `ba_tight_rms`; the exact normal-equation
  #  steps land on the optimum: the result lies between the two)
  ref, tight, tol = float(g["ba_rms"]), float(g["ba_tight_rms"]), max(1e-4, 3 * spread_of(g))
  assert min(ref, tight) - tol <= rms <= max(ref, tight) + tol
  if name == "tiny_huber":   # reproducible reference end point (spread 9e-8 px): 1e-6 px
    tight = out.bundle_adjust(loss=loss, f_scale=f_scale, tolerance=1e-14, xtol=1e-14, gtol=1e-14, max_iterations=300)
    assert abs(calibration.error_stats(tight.reprojection_error).rms - float(g["ba_tight_rms"])) < 1e-6


# -----------------------------------------------------------------------------------------------------------------
# BASELINE configurations: reference trajectories of the complete outlier loop (result-only fixtures)
# -----------------------------------------------------------------------------------------------------------------
def load_big(name):
  import os
  g = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))
  rig = synthetic.make_rig(str(g["config"]))
  # the rig is regenerated from its seed: make sure it is the one the reference ran on
  assert tuple(g["shape"]) == rig.valid.shape and int(g["valid_count"]) == int(rig.valid.sum())
  assert float(g["points_sum"]) == pytest.approx(float(rig.points.sum()), rel=1e-13)
  return g, rig


@pytest.mark.parametrize("name", ["cfg2", "cfg3_40", "cfg4_40", "cfg5_40", "manypairs"])
def test_baseline_configs_against_reference_trajectories(name):
  """BASELINE configs[1] at full size (4 x 200, 84 s per reference solve), configs[2..4] at 40 frames and the 16-camera x
  10-board rig (160 (camera, board) pairs: the many-pairs path of the shared assembly): residuals at the
  start point, the bundle adjustment from the start point, and Workspace.calibrate's complete outlier loop against the
  unmodified reference: identical inlier masks after three rounds; final RMS (all / inliers) within 1e-6 px of the
  converged optimum of the reference's residual function and within the reference's own spread of its end point."""
  from multical_amd import Workspace
  g, rig = load_big(name)
  c = mirror(rig)
  assert np.array_equal(c.param_vec, g["x0"])
  with Handle(c) as h:
    r = h.residuals(g["x0"])
    assert r.size == int(g["r0_size"]) and np.abs(r[:64] - g["r0_head"]).max() < 1e-9
    assert r @ r == pytest.approx(float(g["r0_sq"]), rel=1e-11)
    assert abs(rms_of(h, g["x0"]) - float(g["rms0"])) < 1e-9
    assert abs(rms_of(h, g["ba_x_raw"]) - float(g["ba_rms"])) < 1e-9      # the reference's solution evaluates identically
    # bundle_adjust from the start point (1 % gross outliers still in)
    res = h.solve(g["x0"])
    rms = rms_of(h, res.x)
    assert res.cost <= float(g["ba_cost"]) * (1 + 1e-9)
    # (no lower bound here: with the gross outliers still in, the dense Gauss-Newton polish behind `ba_tight_*` of the big
    #  fixtures is not guaranteed to have bottomed out; the outlier-free optimum below is)
    assert rms <= float(g["ba_rms"]) + max(1e-6, 3 * spread_of(g))
  out = Workspace(c).calibrate(cameras=rig.optimize["cameras"], camera_poses=rig.optimize["camera_poses"])
  mask = np.unpackbits(g["ao_inliers_packed"])[:rig.valid.size].reshape(rig.valid.shape).astype(bool)
  assert np.array_equal(out.inliers, mask)
  rms_all, rms_inl = out.error_statistics(False).rms, out.error_statistics(True).rms
  sp_all = float(np.abs(g["ao_pert_rms"] - g["ao_rms"]).max())
  sp_inl = float(np.abs(g["ao_pert_rms_inliers"] - g["ao_rms_inliers"]).max())
  assert float(g["ao_tight_rms_inliers"]) - 1e-6 <= rms_inl <= float(g["ao_rms_inliers"]) + max(1e-6, 3 * sp_inl)
  # (the RMS over ALL valid points is dominated by the rejected 5-50 px outliers, which the last solves do not see: it
  #  is compared loosely here and to 1e-6 px after the tight polish below)
  assert abs(rms_all - float(g["ao_rms"])) <= max(5e-4, 3 * sp_all)
  tight = out.bundle_adjust(tolerance=1e-14, xtol=1e-14, gtol=1e-14, max_iterations=200)
  d_inl = tight.error_statistics(True).rms - float(g["ao_tight_rms_inliers"])
  if abs(d_inl) < 1e-6:
    assert abs(tight.error_statistics(False).rms - float(g["ao_tight_rms"])) < 1e-6
  else:
    # cfg4_40 (16 cameras on a cube, 40 frames: weakly determined intrinsics): the Levenberg-Marquardt polish of the
    # reference's residual function behind `ao_tight_*` creeps along a flat valley and stops 3e-6 px ABOVE the point the
    # exact normal-equation solve reaches.  Accept a lower optimum only: the ORACLE's own residual function (bit-identical
    # to the reference) must confirm the lower cost at our solution.
    assert -1e-5 < d_inl < 0
    oc = restate.from_rig(rig).copy(inlier_mask=mask)
    r = oc.evaluate(tight.param_vec)
    assert 0.5 * r @ r <= float(g["ao_tight_cost"])


@pytest.mark.parametrize("name", ["cfg2", "cfg3_40", "cfg4_40", "cfg5_40", "manypairs"])
def test_baseline_configs_scipy_driven(name):
  """protocol (B) on the BASELINE-sized fixtures (configs[1] at full size, configs[2..4] at 40 frames, the 160-pair rig)
  THROUGH THE PRODUCT ROUTE `Calibration.bundle_adjust(solver="scipy")`: the reference's own solver call -- scipy TRF / LSMR --
  driven by the HIP fun + analytic jac from the reference's start point ends within 1e-6 px of the reference's end point,
  with the reference's number of function evaluations and status (profiles/parity_table.md: measured 4e-8 ... 7e-8 px; 6e-7 px
  on the 160-pair rig, whose own reproducibility is 1.6e-6 px)."""
  g, rig = load_big(name)
  out, res = mirror(rig).bundle_adjust(solver="scipy", return_result=True)
  rms = calibration.error_stats(out.reprojection_error).rms
  tol = 1e-6 if name != "manypairs" else max(1e-6, 3 * spread_of(g))
  assert abs(rms - float(g["ba_rms"])) <= tol, (rms, float(g["ba_rms"]), spread_of(g))
  assert res.nfev == int(g["ba_nfev"]) and res.status == int(g["ba_status"])


# -----------------------------------------------------------------------------------------------------------------
# the scipy-driven route as a PRODUCT mode: Calibration.bundle_adjust(solver="scipy"),
# calibration.set_solver("scipy") for Workspace.calibrate, dropin.install(mode="scipy")
# -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", B_TIGHT)
def test_product_scipy_mode_reproduces_the_reference_end_point(name):
  """north_star tolerance, two-sided: final RMS within 1e-6 px of the reference's, identical nfev and status, on every fixture
  whose reference end point is defined to that level (hand-eye, fixed intrinsics, cfg1 = BASELINE configs[0], huber loss,
  thin-prism model)."""
  g, rig = load_golden(name)
  kw = scipy_args(g)
  c = mirror(rig)
  x_before = c.param_vec.copy()
  out, res = c.bundle_adjust(tolerance=kw["ftol"], f_scale=kw["f_scale"], max_iterations=kw["max_nfev"], loss=kw["loss"],
                             solver="scipy", return_result=True)
  assert np.array_equal(c.param_vec, x_before) and out is not c
  rms = calibration.error_stats(out.reprojection_error).rms
  assert abs(rms - float(g["ba_rms"])) <= 1e-6, (name, rms - float(g["ba_rms"]))
  assert res.nfev == int(g["ba_nfev"]) and res.status == int(g["ba_status"])
  assert res.cost == pytest.approx(float(g["ba_cost"]), rel=1e-7)
  # the native solver on the same object: never above this end point (it ends at the converged optimum)
  nat, nres = c.bundle_adjust(tolerance=kw["ftol"], f_scale=kw["f_scale"], max_iterations=kw["max_nfev"], loss=kw["loss"],
                              solver="native", return_result=True)
  assert nres.cost <= res.cost * (1 + 1e-6)


def test_product_scipy_mode_logs_scipys_own_table_and_drives_workspace_calibrate():
  """set_solver("scipy"): Workspace.calibrate -> adjust_outliers -> bundle_adjust runs the reference's loop with the reference's
  solver; scipy's verbose=2 output reaches the "calibration" logger through redirect_stdout (calibration.py:208) and equals the
  reference's own log to the printed precision; masks identical, inlier RMS within 1e-6 px of the reference's END point."""
  import logging
  from multical_amd import Workspace
  g, rig = load_golden("cfg1")
  lines = []

  class Grab(logging.Handler):
    def emit(self, rec):
      lines.append(rec.getMessage())

  log = logging.getLogger("calibration")
  hd = Grab()
  log.addHandler(hd)
  log.setLevel(logging.INFO)
  prev = calibration.set_solver("scipy")
  try:
    assert calibration.get_solver() == "scipy"
    c = mirror(rig)
    out = c.bundle_adjust()
    ba_lines = list(lines)
    ao = Workspace(c).calibrate(cameras=rig.optimize["cameras"], camera_poses=rig.optimize["camera_poses"])
  finally:
    calibration.set_solver(prev)
    log.removeHandler(hd)
  assert abs(calibration.error_stats(out.reprojection_error).rms - float(g["ba_rms"])) <= 1e-6
  ref_rows = [l.split() for l in str(g["ba_log"]).splitlines() if l.strip() and l.split()[0].isdigit()]
  rows = [l.split() for l in "\n".join(ba_lines).splitlines() if l.strip() and l.split()[0].isdigit()]
  assert len(rows) == len(ref_rows)
  for a, b in zip(rows, ref_rows):     # iteration, nfev, cost to the printed digits; cost reduction / step norm to a percent
    assert a[:3] == b[:3], (a, b)      # (analytic instead of forward-difference Jacobian: the last printed digit may differ)
    for u, v in zip(a[3:5], b[3:5]):
      assert float(u) == pytest.approx(float(v), rel=1e-2), (a, b)
  assert any("`ftol` termination condition is satisfied." in l for l in ba_lines)
  assert np.array_equal(ao.inliers, g["ao_inliers"])
  assert abs(ao.error_statistics(True).rms - float(g["ao_rms_inliers"])) <= 1e-6
  assert abs(ao.error_statistics(False).rms - float(g["ao_rms"])) <= 1e-6


# -----------------------------------------------------------------------------------------------------------------
# solver = "lsmr": scipy's TRF + LSMR step restated on the device (mcba_options.tr_solver = MCBA_TR_LSMR)
# -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", B_TIGHT + ["cfg2", "cfg3_40", "cfg4_40", "cfg5_40", "manypairs"])
def test_device_lsmr_mode_reproduces_the_reference_end_point(name):
  """The reference's solver is scipy's TRF with its LSMR trust-region step (scipy chooses tr_solver='lsmr' for the sparse
  Jacobian of calibration.py:209-210).  mcba_solve(tr_solver = lsmr) restates that driver and scipy's lsmr() line by line with
  the two Jacobian products J_h v / J_h^T u as matrix-free HIP kernels: final RMS within 1e-6 px of the REFERENCE's end point,
  identical nfev and status, on every fixture whose reference end point is defined to that level -- the BASELINE-sized ones
  included -- without scipy's host-side LSMR (seconds to hours)."""
  big = name in ("cfg2", "cfg3_40", "cfg4_40", "cfg5_40", "manypairs")
  g, rig = load_big(name) if big else load_golden(name)
  kw = dict(x_scale='jac', ftol=1e-4, max_nfev=100, method='trf', loss='linear', f_scale=1.0) if big else scipy_args(g)
  c = mirror(rig)
  out, res = c.bundle_adjust(tolerance=kw["ftol"], f_scale=kw["f_scale"], max_iterations=kw["max_nfev"], loss=kw["loss"],
                             solver="lsmr", return_result=True)
  rms = calibration.error_stats(out.reprojection_error).rms
  # 1e-6 px wherever the reference's end point is defined to 1e-6 px; elsewhere the reference's OWN reproducibility (max over ten
  # re-runs with 1e-12 px noise: 3.5e-6 px at cfg2 ... 8e-6 at cfg4_40): a Golub-Kahan process that has lost orthogonality
  # amplifies rounding-level differences between two implementations of the same recurrences to 1e-4 relative in the weakly
  # determined components of the step (tests/test_oracle.py::test_scipys_lsmr_step_is_not_reproducible_beyond_rounding_noise)
  tol = 1e-6 if name in WELL_DEFINED else max(1e-6, spread_of(g))
  assert abs(rms - float(g["ba_rms"])) <= tol, (name, rms - float(g["ba_rms"]), spread_of(g))
  assert res.nfev == int(g["ba_nfev"]) and res.status == int(g["ba_status"])
  assert res.cost == pytest.approx(float(g["ba_cost"]), rel=2e-6)


@pytest.mark.parametrize("name", [n for n in PROTOCOL_CASES if n not in B_TIGHT])
def test_device_lsmr_mode_within_the_references_own_spread(name):
  """the fixtures whose reference end point moves by 1e-5 ... 1e-3 px under 1e-12 px perturbations of its own residual
  function: the device LSMR mode lands inside that spread (like the scipy mode), never on the far side of the valley the
  exact solver walks down (tiny_rational: 2.17 px after 98 evaluations; here the reference's 2.82 px after ~9)."""
  g, rig = load_golden(name)
  kw = scipy_args(g)
  with Handle(mirror(rig)) as h:
    res = h.solve(g["x0"], tolerance=kw["ftol"], loss=kw["loss"], f_scale=kw["f_scale"], max_iterations=kw["max_nfev"],
                  tr_solver="lsmr")
    rms = rms_of(h, res.x)
  ref, spread = float(g["ba_rms"]), spread_of(g)
  assert abs(rms - ref) <= max(1e-6, 3 * spread), (name, rms - ref, spread, res.nfev, int(g["ba_nfev"]))
  assert res.nfev <= 2 * int(g["ba_nfev"]) + 5


def test_device_lsmr_mode_through_the_dropin_and_workspace():
  """dropin.install() / the default solver of the mirror (both "lsmr"): the reference-shaped call chains land on the reference's
  END points."""
  import types
  from multical_amd import dropin, Workspace
  from test_dropin import _PlainCalibration, _as_plain
  g, rig = load_golden("cfg1")
  mod = types.SimpleNamespace(Calibration=_PlainCalibration)
  try:
    dropin.install(calibration_module=mod)
    assert _PlainCalibration.bundle_adjust is dropin.bundle_adjust is dropin.MODES["lsmr"]
    out = _as_plain(mirror(rig)).bundle_adjust()
    assert abs(calibration.error_stats(out.reprojection_error).rms - float(g["ba_rms"])) < 1e-6
  finally:
    dropin.uninstall(calibration_module=mod)
  import os
  assert os.environ.get("MULTICAL_AMD_SOLVER") is None
  prev = calibration.set_solver("lsmr")     # (this module's fixture selected "native")
  try:
    ao = Workspace(mirror(rig)).calibrate(cameras=rig.optimize["cameras"], camera_poses=rig.optimize["camera_poses"])
  finally:
    calibration.set_solver(prev)
  assert np.array_equal(ao.inliers, g["ao_inliers"])
  assert abs(ao.error_statistics(True).rms - float(g["ao_rms_inliers"])) <= 1e-6
