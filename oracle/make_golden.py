"""TEST INFRASTRUCTURE: generate tests/golden/*.npz by running the REAL reference in this container.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden [case ...]

The reference has no golden vectors of its own, and /root/reference does not exist on
the GPU box, so outputs of the unmodified reference code (loaded through oracle/refload.py; cv2 projection
restated, see oracle/shims/cv2) are committed as fixtures together with this script.

Per case the fixture holds the inputs (synthetic rig arrays, multical_amd.synthetic.rig_to_arrays) and
  x0            Calibration.param_vec                                  (parameters.py:44-46)
  r0            evaluate(x0)                                           (calibration.py:204-206)
  J_*           2-point finite-difference Jacobian at x0 exactly as scipy.least_squares builds it from
                Calibration.sparsity_matrix (calibration.py:173-196)   [CSR; small cases only]
  err0 / rms0   reprojection_error / error_stats at x0                 (calibration.py:134-136,304-310)
  ba_*          Calibration.bundle_adjust(**ba_kwargs) result          (calibration.py:199-212)
  ao_*          Calibration.adjust_outliers(num_adjustments=3, select_outliers=select_threshold(.75, 5))
                as Workspace.calibrate drives it                       (workspace.py:228-247)
"""
import io
import os
import sys
import json
import contextlib
import copy

import numpy as np

from multical_amd import synthetic
from . import build_reference

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ---- rig mutators (edge cases the reference's asserts / masks define) ----------------------------
def _edge(rig):
  rig.frame_valid = rig.frame_valid.copy()
  rig.frame_valid[3] = False                 # invalid rig pose: identity in x, zero Jacobian columns
  rig.init.rig[3] = np.eye(4)                # tables.py:176-181: invalid poses are stored as identity
  rig.valid[:, 5] = False                    # a valid frame without any observation
  rig.points[:, 5] = 0
  rig.camera_valid = rig.camera_valid.copy()
  rig.camera_valid[2] = False
  rig.init.camera_poses[2] = np.eye(4)
  rig.init.cameras[1].fix_aspect = True      # camera.py:147-148,159-160
  # a camera calibrated with CALIB_FIX_ASPECT_RATIO has fx == fy; with fx != fy the reference's own
  # `reprojection_error` (raw intrinsics) and `evaluate` (param_vec -> mean focal length) would disagree
  rig.init.cameras[1].intrinsic = rig.init.cameras[1].intrinsic.copy()
  rig.init.cameras[1].intrinsic[1, 1] = rig.init.cameras[1].intrinsic[0, 0]
  rig.init.cameras[0].has_skew = True        # camera.py:139-141 (skew is carried but cv2 ignores it)
  rig.init.cameras[0].intrinsic = rig.init.cameras[0].intrinsic.copy()
  rig.init.cameras[0].intrinsic[0, 1] = 0.7
  return rig


def _adjust_board(rig):
  rig.optimize = dict(rig.optimize, boards=True)
  return rig


CASES = {
  # name: (rig config name | dict, mutator, bundle_adjust kwargs, run adjust_outliers?, store J?)
  "tiny": ("tiny", None, {}, True, True),
  "tiny_rolling": ("tiny_rolling", None, {}, True, True),
  "tiny_fisheye": ("tiny_fisheye", None, {}, True, True),
  "tiny_handeye": ("tiny_handeye", None, {}, True, True),
  "tiny_rational": ("tiny_rational", None, {}, False, True),
  "tiny_thin_prism": (dict(synthetic.CONFIGS["tiny_rational"], model="thin_prism", seed=17), None, {}, False, True),
  "tiny_tilted": ("tiny_tilted", None, {}, False, True),
  "tiny_edge": (dict(cameras=3, frames=8, boards=["charuco_10x10", "charuco_10x10"], motion="static",
                     model="standard", optimize_cameras=True, layout="stereo", seed=21), _edge, {}, True, True),
  "tiny_fixintr": (dict(synthetic.CONFIGS["tiny"], optimize_cameras=False, seed=22), None, {}, False, True),
  "tiny_softl1": ("tiny", None, dict(loss='soft_l1', f_scale=1.5), False, False),
  "tiny_huber": ("tiny", None, dict(loss='huber', f_scale=2.0), False, False),
  "tiny_arctan": ("tiny", None, dict(loss='arctan', f_scale=3.0), False, False),
  "tiny_boards": ("tiny", _adjust_board, {}, False, True),
  "cfg1": ("cfg1", None, {}, True, False),
}


def fd_jacobian(calib, x0, f0):
  """The Jacobian scipy.optimize.least_squares(jac='2-point', jac_sparsity=S) evaluates at x0
  (scipy/optimize/_lsq/least_squares.py:153-163 + _numdiff.py)."""
  from scipy.optimize._numdiff import approx_derivative, group_columns
  from scipy.sparse import csr_matrix
  S = csr_matrix(calib.sparsity_matrix)
  groups = group_columns(S)

  def fun(x):
    c = calib.with_param_vec(x)
    return (c.reprojected.points - c.point_table.points)[calib.inliers].ravel()

  J = approx_derivative(fun, x0, rel_step=None, method='2-point', f0=f0,
                        bounds=(-np.inf, np.inf), sparsity=(S, groups))
  return csr_matrix(J), int(groups.max()) + 1


def run_case(name):
  cfg, mutate, ba_kwargs, run_ao, store_j = CASES[name]
  rig = synthetic.make_rig(cfg)
  rig.name = name
  if mutate is not None:
    rig = mutate(rig)
  calib, ref = build_reference.reference_calibration(rig)
  error_stats = ref.optimization_calibration.error_stats
  select_threshold = ref.optimization_calibration.select_threshold

  out = synthetic.rig_to_arrays(rig)
  x0 = calib.param_vec

  def evaluate(c0, x):
    c = c0.with_param_vec(x)
    return (c.reprojected.points - c.point_table.points)[c0.inliers].ravel()

  r0 = evaluate(calib, x0)
  out["x0"], out["r0"] = x0, r0
  out["inliers0"] = calib.inliers
  err0 = calib.reprojection_error
  out["err0"] = err0
  out["rms0"] = error_stats(err0).rms

  if store_j:
    J, n_groups = fd_jacobian(calib, x0, r0)
    out["J_data"], out["J_indices"], out["J_indptr"] = J.data, J.indices, J.indptr
    out["J_shape"] = np.array(J.shape)
    out["J_groups"] = n_groups
  else:
    rng = np.random.default_rng(0)
    J, n_groups = fd_jacobian(calib, x0, r0)
    v = rng.normal(size=x0.size)
    out["Jv_v"], out["Jv"] = v, J @ v
    out["JTr"] = J.T @ r0
    out["J_groups"] = n_groups

  # --- bundle_adjust through the reference entry point, scipy's table captured via the logger ------
  log = io.StringIO()
  import logging
  handler = logging.StreamHandler(log)
  logger = logging.getLogger("calibration")
  logger.addHandler(handler)
  logger.setLevel(logging.INFO)
  logger.propagate = False

  # the reference returns only the new Calibration; recover scipy's counters by wrapping least_squares
  from scipy import optimize
  results = []
  real_lsq = optimize.least_squares

  def spy(*a, **k):
    res = real_lsq(*a, **k)
    results.append(res)
    return res

  optimize.least_squares = spy
  try:
    ba = calib.bundle_adjust(**ba_kwargs)
    res = results[-1]
    out["ba_kwargs_json"] = np.array(json.dumps(ba_kwargs))
    out["ba_x"] = ba.param_vec          # canonicalised by with_param_vec -> from_matrix (rtvec.py:29-32)
    out["ba_x_raw"] = res.x             # scipy's res.x (raw rotation vectors)
    out["ba_cost"], out["ba_optimality"] = res.cost, res.optimality
    out["ba_nfev"], out["ba_njev"], out["ba_status"] = res.nfev, res.njev, res.status
    out["ba_rms"] = error_stats(ba.reprojection_error).rms
    out["ba_log"] = np.array(log.getvalue())

    if run_ao:
      results.clear()
      ao = calib.adjust_outliers(num_adjustments=3, select_outliers=select_threshold(quantile=0.75, factor=5.0),
                                 loss='linear', tolerance=1e-4)
      out["ao_x"] = ao.param_vec
      out["ao_inliers"] = ao.inliers
      out["ao_rms"] = error_stats(ao.reprojection_error).rms
      out["ao_rms_inliers"] = error_stats(ao.reprojection_inliers).rms
      out["ao_nfev"] = np.array([r.nfev for r in results])
      out["ao_cost"] = np.array([r.cost for r in results])
      # converged optimum of the reference's residual function on the final inlier set: the value any converged
      # solver must reach (hard part 1).  scipy's default LSMR trust-region solver does not converge
      # tightly on these problems (hundreds of evaluations, status 0), so the polish uses scipy's exact (SVD)
      # trust-region solver with 3-point differences of the reference's own `evaluate`.
      res_t = real_lsq(lambda x: evaluate(ao, x), ao.param_vec, jac='3-point', x_scale='jac', tr_solver='exact',
                       ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400, method='trf')
      tight = ao.with_param_vec(res_t.x)
      out["ao_tight_rms"] = error_stats(tight.reprojection_error).rms
      out["ao_tight_rms_inliers"] = error_stats(tight.reprojection_inliers).rms
      out["ao_tight_cost"] = res_t.cost
      out["ao_tight_status"] = res_t.status
      out["ao_tight_optimality"] = res_t.optimality
  finally:
    optimize.least_squares = real_lsq
    logger.removeHandler(handler)

  os.makedirs(GOLDEN_DIR, exist_ok=True)
  path = os.path.join(GOLDEN_DIR, f"{name}.npz")
  np.savez_compressed(path, **out)
  print(f"{name}: n={x0.size} m={r0.size} rms0={float(out['rms0']):.4f} ba_rms={float(out['ba_rms']):.6f} "
        f"nfev={int(out['ba_nfev'])} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main(argv):
  names = argv or list(CASES)
  for n in names:
    run_case(n)


if __name__ == "__main__":
  sys.dont_write_bytecode = True
  main(sys.argv[1:])
