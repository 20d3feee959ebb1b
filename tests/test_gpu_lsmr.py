"""GPU (MI355X): the lsmr mode (mcba_options.tr_solver = MCBA_TR_LSMR) below and above the solver level.

  * kernel level: the matrix-free products J v / J^T u of k_lsmr_jv / k_lsmr_jtu / k_lsmr_gather against the analytic Jacobian
    mcba_jacobian returns (itself pinned to the reference's finite differences and to 3-point differences of the oracle:
    tests/test_gpu_parity.py, tests/test_gpu_protocol.py), on every motion / camera model and with boards=True;
  * END POINTS of the unmodified reference at the BASELINE configurations' STATED sizes (tests/golden/cfg*_endpoint.npz,
    oracle/make_endpoint.py: hours of one host core each): solver = "lsmr" reproduces them with identical nfev / status.
"""
import os

import numpy as np
import pytest

from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
from util import load_golden, mirror, GOLDEN

pytestmark = pytest.mark.gpu

PRODUCT_CASES = ["tiny", "tiny_rolling", "tiny_fisheye", "tiny_handeye", "tiny_rational", "tiny_thin_prism", "tiny_tilted",
                 "tiny_edge", "tiny_fixintr", "tiny_pin4", "tiny_boards", "tiny_bigboard", "tiny_fishmix", "tiny_fishmix5", "cfg1"]


@pytest.mark.parametrize("name", PRODUCT_CASES)
def test_lsmr_products_against_the_jacobian(name):
  """J v and J^T u of the lsmr mode's kernels = h.jacobian(x) @ v and .T @ u to rounding (1e-12 of sum |J_ij| |v_j|): the row
  pairs, the That chain of the pose blocks, the per-view reduction, the gather over the views of every parameter (incl. the
  board-point block of boards=True and invalid / ragged blocks, whose columns must come back exactly zero)."""
  g, rig = load_golden(name)
  c = mirror(rig)
  rng = np.random.default_rng(5)
  for x in (c.param_vec, c.param_vec + 1e-3 * rng.normal(size=c.param_vec.size)):
    with Handle(c) as h:
      J = h.jacobian(x)
      v = rng.normal(size=h.n_params)
      u = rng.normal(size=h.n_residuals)
      jv, jtu = h.lsmr_products(x, v, u)
      jv_only, _ = h.lsmr_products(x, v, None)
      _, jtu_only = h.lsmr_products(x, None, u)
    A = abs(J)
    assert np.abs(jv - J @ v).max() <= 1e-12 * (A @ np.abs(v)).max()
    scale = A.T @ np.abs(u)
    assert np.abs(jtu - J.T @ u).max() <= 1e-12 * scale.max()
    assert np.all(jtu[scale == 0] == 0)                   # structurally zero columns (skew, invalid poses, fix_aspect)
    assert np.array_equal(jv, jv_only) and np.array_equal(jtu, jtu_only)


@pytest.mark.parametrize("name", PRODUCT_CASES)
def test_lsmr_two_launch_kernels_against_the_jacobian(name):
  """The kernels the DEFAULT solver iterates with -- k_lsmr_fused2 (both products of a Golub-Kahan step from one evaluation of the
  analytic rows) and k_lsmr_gather3 -- against the analytic Jacobian: J v to 1e-12, J^T (J v) to 1e-11 (of sum |J^T| |J v|)."""
  g, rig = load_golden(name)
  c = mirror(rig)
  rng = np.random.default_rng(8)
  x = c.param_vec + 1e-3 * rng.normal(size=c.param_vec.size)
  with Handle(c) as h:
    J = h.jacobian(x)
    v = rng.normal(size=h.n_params)
    jv, w = h.lsmr_fused_products(x, v)
    jv2, w2 = h.lsmr_fused_products(x, v)
  A = abs(J)
  ref = J @ v
  assert np.abs(jv - ref).max() <= 1e-12 * (A @ np.abs(v)).max()
  scale = A.T @ np.abs(ref)
  assert np.abs(w - J.T @ ref).max() <= 1e-11 * scale.max()
  assert np.all(w[scale == 0] == 0)
  assert np.array_equal(jv, jv2) and np.array_equal(w, w2)          # deterministic


def test_lsmr_products_after_an_outlier_rejection():
  """the products follow the CURRENT inlier set (view lists, residual order, obs_index of the board-point gather)"""
  g, rig = load_golden("tiny_boards")
  c = mirror(rig)
  x = c.param_vec
  rng = np.random.default_rng(6)
  with Handle(c) as h:
    e, valid = h.reprojection_error(x)
    h.reject_outliers(x, float(np.quantile(e[valid.astype(bool)], 0.9)))
    J = h.jacobian(x)
    assert J.shape[0] == h.n_residuals < g["r0"].size
    v, u = rng.normal(size=h.n_params), rng.normal(size=h.n_residuals)
    jv, jtu = h.lsmr_products(x, v, u)
  A = abs(J)
  assert np.abs(jv - J @ v).max() <= 1e-12 * (A @ np.abs(v)).max()
  assert np.abs(jtu - J.T @ u).max() <= 1e-12 * (A.T @ np.abs(u)).max()


def test_lsmr_mode_with_adjusted_board_points():
  """boards=True (board/charuco.py:112-117, sparsity calibration.py:188-190) under the lsmr mode: the reference's end point of
  `tiny_boards` within the reference's own spread, like every other fixture of test_device_lsmr_mode_*."""
  g, rig = load_golden("tiny_boards")
  with Handle(mirror(rig)) as h:
    res = h.solve(g["x0"], tr_solver="lsmr")
    e, v = h.reprojection_error(res.x)
    assert h.lsmr_iterations() > 0
  rms = float(np.sqrt(np.mean(e[v.astype(bool)] ** 2)))
  ref, spread = float(g["ba_rms"]), float(np.abs(g["ba_pert_rms"] - g["ba_rms"]).max())
  assert abs(rms - ref) <= max(1e-6, 3 * spread), (rms - ref, spread, res.nfev, int(g["ba_nfev"]))
  assert res.nfev <= 2 * int(g["ba_nfev"]) + 5


def load_endpoint(cfg):
  path = os.path.join(GOLDEN, f"{cfg}_endpoint.npz")
  if not os.path.exists(path):
    pytest.skip(f"{path} not generated (oracle/make_endpoint.py)")
  g = dict(np.load(path, allow_pickle=False))
  rig = synthetic.make_rig(str(g["config"]))
  assert tuple(g["shape"]) == rig.valid.shape and int(g["valid_count"]) == int(rig.valid.sum())
  assert float(g["points_sum"]) == pytest.approx(float(rig.points.sum()), rel=1e-13)
  return g, rig


def endpoint_spread(g):
  return float(np.abs(g["ba_pert_rms"] - g["ba_rms"]).max()) if "ba_pert_rms" in g else 0.0


@pytest.mark.parametrize("cfg", ["cfg3", "cfg5", "cfg4"])
def test_lsmr_mode_reproduces_the_reference_end_point_at_full_size(cfg, record_property):
  """BASELINE configs[2] (8 x 500 x 2 rolling shutter: the rig bench.py measures), configs[4] (6 x 400 x 5 fisheye) and configs[3]
  (16 x 1000 x 5) AT THEIR STATED SIZE: the unmodified reference's `Calibration.bundle_adjust()` end point (final RMS, nfev,
  status, cost) is reproduced by solver = "lsmr" within max(1e-6 px, the reference's own spread under 1e-12 px perturbations);
  the exact-step default solver's distance from that end point is reported beside it."""
  g, rig = load_endpoint(cfg)
  c = mirror(rig)
  assert np.array_equal(c.param_vec, g["x0"])
  out, res = c.bundle_adjust(solver="lsmr", return_result=True)
  rms = calibration.error_stats(out.reprojection_error).rms
  spread = endpoint_spread(g)
  native, nres = c.bundle_adjust(solver="native", return_result=True)
  rms_native = calibration.error_stats(native.reprojection_error).rms
  record_property("lsmr_minus_reference_px", rms - float(g["ba_rms"]))
  record_property("native_minus_reference_px", rms_native - float(g["ba_rms"]))
  record_property("reference_spread_px", spread)
  print(f"{cfg}: reference {float(g['ba_rms']):.9f} px (nfev {int(g['ba_nfev'])}, {float(g['ba_seconds']):.0f} s on one core), "
        f"lsmr {rms - float(g['ba_rms']):+.2e} px in {res.solve_seconds * 1e3:.1f} ms, native {rms_native - float(g['ba_rms']):+.2e} px "
        f"in {nres.solve_seconds * 1e3:.2f} ms, reference spread {spread:.1e}")
  assert abs(rms - float(g["ba_rms"])) <= max(1e-6, spread), (cfg, rms - float(g["ba_rms"]), spread)
  assert res.nfev == int(g["ba_nfev"]) and res.status == int(g["ba_status"])
  cost_spread = float(np.abs(g["ba_pert_cost"] - g["ba_cost"]).max()) if "ba_pert_cost" in g else 0.0
  assert abs(res.cost - float(g["ba_cost"])) <= max(2e-6 * float(g["ba_cost"]), cost_spread), (res.cost, float(g["ba_cost"]), cost_spread)
  # the exact solver ends at or below the reference's cost (converged optimum of the same function), never above its end point
  assert rms_native <= float(g["ba_rms"]) + max(1e-6, spread)


@pytest.mark.parametrize("cfg", ["cfg3", "cfg5", "cfg4"])
def test_workspace_calibrate_at_full_size_against_the_reference(cfg):
  """Workspace.calibrate's outlier loop (workspace.py:228-247 -> calibration.py:254-268) at the stated size under solver="lsmr":
  the reference's inlier mask after three rounds, bit for bit, and its inlier RMS."""
  from multical_amd import Workspace
  g, rig = load_endpoint(cfg)
  if "ao_inliers_packed" not in g:
    pytest.skip("adjust_outliers end point of the reference not generated")
  prev = calibration.set_solver("lsmr")
  try:
    ao = Workspace(mirror(rig)).calibrate(cameras=rig.optimize["cameras"], camera_poses=rig.optimize["camera_poses"])
  finally:
    calibration.set_solver(prev)
  ref_mask = np.unpackbits(g["ao_inliers_packed"])[:rig.valid.size].reshape(rig.valid.shape).astype(bool)
  diff = int(np.sum(ao.inliers != ref_mask))
  assert diff <= 2, diff        # (an observation whose error sits within rounding of the threshold may flip)
  assert abs(ao.error_statistics(True).rms - float(g["ao_rms_inliers"])) <= max(1e-6, endpoint_spread(g))


@pytest.mark.parametrize("name", ["cfg1", "tiny_handeye", "tiny_rolling", "tiny_fisheye", "tiny_boards", "tiny_edge", "tiny_fishmix"])
def test_lsmr_iteration_forms_agree(name):
  """The LSMR iteration exists in three forms -- two launches (default: k_lsmr_fused2 / k_lsmr_gather3), three (k_lsmr_fused) and the
  six launches of round 4 (one kernel per product + two scalar kernels): the same recurrences with different summation orders.
  They must take the same trust-region trajectory wherever the reference's end point is defined to 1e-6 px, and stay inside the
  reference's own spread elsewhere."""
  g, rig = load_golden(name)
  spread = float(np.abs(g["ba_pert_rms"] - g["ba_rms"]).max())
  out = {}
  with Handle(mirror(rig)) as h:
    for mode in (0, 1, 2):
      h.set_lsmr_fused(mode)
      res = h.solve(g["x0"], tr_solver="lsmr")
      e, v = h.reprojection_error(res.x)
      out[mode] = (res.nfev, res.status, float(np.sqrt(np.mean(e[v] ** 2))), h.lsmr_iterations())
  ref = float(g["ba_rms"])
  for mode in (0, 1, 2):
    assert abs(out[mode][2] - ref) <= max(1e-6, 3 * spread), (name, mode, out)
  if spread < 3e-7:
    assert out[0][:2] == out[1][:2] == out[2][:2] == (int(g["ba_nfev"]), int(g["ba_status"])), (name, out)
    assert max(abs(out[m][2] - out[0][2]) for m in (1, 2)) <= 1e-6, (name, out)


@pytest.mark.parametrize("name", ["tiny_rolling", "tiny_boards", "cfg1"])
def test_lsmr_solve_is_bit_repeatable(name):
  """No atomics and no order-dependent reductions anywhere in the lsmr route: every workgroup folds partial sums in a fixed order
  and the scalar recurrences are bit-identical in all of them, so two solves of the same problem return the SAME bits -- also
  across handles -- although thousands of LSMR iterations amplify any rounding difference to the 1e-6 px level."""
  g, rig = load_golden(name)
  c = mirror(rig)
  xs, its = [], []
  for _ in range(2):
    with Handle(c) as h:
      for _ in range(2):
        res = h.solve(g["x0"], tr_solver="lsmr")
        xs.append(res.x)
        its.append((res.nfev, res.status, h.lsmr_iterations(), res.cost))
  assert all(np.array_equal(xs[0], x) for x in xs[1:]), [float(np.abs(xs[0] - x).max()) for x in xs[1:]]
  assert all(it == its[0] for it in its), its


@pytest.mark.parametrize("name", ["tiny_autoscale", "tiny_autoscale_huber", "tiny", "tiny_rolling", "tiny_handeye", "tiny_edge"])
def test_workspace_calibrate_under_the_default_solver(name):
  """Workspace.calibrate (workspace.py:228-247: enable -> 3 x {report, auto-scale, reject, bundle_adjust} -> report) under the DEFAULT
  solver -- robust losses with `auto_scale` included -- against the reference's own run of the same loop: the inlier mask after three
  rounds (up to the mask differences the reference's own perturbed re-runs show) and the inlier RMS within the reference's spread."""
  import json
  from multical_amd import Workspace
  g, rig = load_golden(name)
  kw = json.loads(str(g["ao_kwargs_json"])) if "ao_kwargs_json" in g else {}
  assert calibration.get_solver() == "lsmr"
  out = Workspace(mirror(rig)).calibrate(cameras=rig.optimize["cameras"], camera_poses=rig.optimize["camera_poses"],
                                         loss=kw.get("loss", "linear"), auto_scale=kw.get("auto_scale", None))
  allowed = int(g["ao_pert_mask_diff"].max()) if "ao_pert_mask_diff" in g else 0
  assert int((out.inliers != g["ao_inliers"]).sum()) <= allowed, (int((out.inliers != g["ao_inliers"]).sum()), allowed)
  spread = float(np.abs(g["ao_pert_rms_inliers"] - g["ao_rms_inliers"]).max())
  rms_inl = out.error_statistics(True).rms
  assert abs(rms_inl - float(g["ao_rms_inliers"])) <= max(1e-6, 3 * spread), (rms_inl, float(g["ao_rms_inliers"]), spread)


@pytest.mark.parametrize("cfg", ["cfg5", "cfg3"])
def test_workspace_calibrate_with_a_robust_loss_at_full_size(cfg):
  """Workspace.calibrate(loss='soft_l1', auto_scale=2.0) (workspace.py:239-244: the soft margin re-derived from the error quantile in
  every round) at the stated size under the default solver, against the unmodified reference's run of the same loop."""
  import json
  from multical_amd import Workspace
  g, rig = load_endpoint(cfg)
  if "aor_inliers_packed" not in g:
    pytest.skip("robust outlier loop of the reference not generated (oracle/make_endpoint.py aor)")
  kw = json.loads(str(g["aor_kwargs_json"]))
  out = Workspace(mirror(rig)).calibrate(cameras=rig.optimize["cameras"], camera_poses=rig.optimize["camera_poses"],
                                         loss=kw["loss"], auto_scale=kw["auto_scale"])
  ref_mask = np.unpackbits(g["aor_inliers_packed"])[:rig.valid.size].reshape(rig.valid.shape).astype(bool)
  allowed = 2 + (int(g["aor_pert_mask_diff"].max()) if "aor_pert_mask_diff" in g else 0)
  assert int(np.sum(out.inliers != ref_mask)) <= allowed
  spread = float(np.abs(g["aor_pert_rms_inliers"] - g["aor_rms_inliers"]).max()) if "aor_pert_rms_inliers" in g else 0.0
  rms_inl = out.error_statistics(True).rms
  print(f"{cfg}: reference inlier RMS {float(g['aor_rms_inliers']):.9f} (nfev {g['aor_nfev']}, {float(g['aor_seconds']):.0f} s), "
        f"here {rms_inl - float(g['aor_rms_inliers']):+.2e}, reference spread {spread:.1e}")
  assert abs(rms_inl - float(g["aor_rms_inliers"])) <= max(1e-5, 3 * spread)
