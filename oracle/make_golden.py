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
                as Workspace.calibrate drives it                       (workspace.py:228-247); `ao_kwargs_json` holds the
                Workspace.calibrate arguments (loss, auto_scale) of the case
  *_tight_*     the CONVERGED optimum of the reference's own residual function (scipy's exact trust-region solver /
                a dense Gauss-Newton polish on 3-point differences of the reference's `evaluate`)
  *_pert_*      SELF-SENSITIVITY of the reference: the same reference call repeated with N(0, 1e-12 px) noise added to
                its residual function (a few ulp of a 2000 px coordinate).  scipy's forward differences (h ~ 1.5e-8)
                amplify that noise 1e8-fold into the Jacobian, and LSMR-truncated steps + ftol=1e-4 stop the run before
                convergence, so the reference's default-tolerance end point is only defined up to this spread.

Big configurations (cfg2, cfg3_40, cfg4_40, cfg5_40) are stored RESULT-ONLY: the rig is regenerated from its seed by
multical_amd.synthetic.make_rig (checksums of the observation table are kept in the fixture).
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
  "tiny_pin4": ("tiny_pin4", None, {}, True, True),
  # Workspace.calibrate(loss=..., auto_scale=...) (workspace.py:239-244): f_scale = quantile_0.75(errors) * auto_scale
  "tiny_autoscale": ("tiny", None, dict(loss='soft_l1', f_scale=1.5), True, False),
  "tiny_autoscale_huber": ("tiny_rolling", None, dict(loss='huber', f_scale=2.0), True, False),
  # a board with more points than one 512-slot compaction segment of the HIP kernels (25 x 35 charuco: 816 corners)
  "tiny_bigboard": ("tiny_bigboard", None, {}, True, True),
  # pinhole AND fisheye cameras in one rig (missing 3): independent Camera / CameraFisheye objects in the
  # reference's ParamList; 4-coefficient pinhole + fisheye = equal block sizes, the mix the reference itself can solve
  "tiny_fishmix": ("tiny_fishmix", None, {}, True, True),
}

AO_KWARGS = {   # Workspace.calibrate arguments of the adjust_outliers run (default: loss='linear', no auto_scale)
  "tiny_autoscale": dict(loss='soft_l1', auto_scale=2.0),
  "tiny_autoscale_huber": dict(loss='huber', auto_scale=1.0),
}

# result-only goldens of the BASELINE configurations: (config name, frames or None = as configured)
BIG_CASES = {
  "cfg2": ("cfg2", None),
  "cfg3_40": ("cfg3_40", None),
  "cfg4_40": ("cfg4_40", None),
  "cfg5_40": ("cfg5_40", None),
  "manypairs": ("tiny_manypairs", None),   # 16 cameras x 10 boards = 160 (camera, board) pairs
}
# evaluation-only goldens of the BASELINE configurations AT THEIR STATED SIZE (one reference `evaluate` costs seconds
# there, a reference solve hours): residual checksums, a strided sample of the residual vector, error statistics at x0
# and at a seeded perturbed point x1, and a central-difference directional derivative of the reference's cost
FULL_CASES = {
  "cfg3_full": "cfg3",     # 8 x 500 x 2 rolling shutter: the rig bench.py measures
  "cfg4_full": "cfg4",     # 16 x 1000 x 5
  "cfg5_full": "cfg5",     # 6 x 400 x 5 fisheye hand-eye
}
N_PERT = int(os.environ.get("MCBA_GOLDEN_NPERT", "10"))   # perturbed re-runs per reference call (oracle/make_pert.py widened the older
                                                          # fixtures to the same 10; the 160-pair rig keeps 3: 44 reference solves of
                                                          # 138 k residuals are hours)
PERT_SIGMA = 1e-12    # px


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


class _Spy(object):
  """Wraps scipy.optimize.least_squares (the call at calibration.py:209-210): records the OptimizeResults and, when
  `sigma` > 0, adds N(0, sigma) noise to the reference's residual function (self-sensitivity runs)."""

  def __init__(self, sigma=0.0, seed=0):
    from scipy import optimize
    self.optimize = optimize
    self.real = optimize.least_squares
    self.results = []
    self.sigma = sigma
    self.rng = np.random.default_rng(seed)

  def __enter__(self):
    def spy(fun, x0, *a, **k):
      f = fun
      if self.sigma > 0:
        f = lambda x: (lambda r: r + self.rng.normal(size=r.size) * self.sigma)(fun(x))
      res = self.real(f, x0, *a, **k)
      self.results.append(res)
      return res
    self.optimize.least_squares = spy
    return self

  def __exit__(self, *a):
    self.optimize.least_squares = self.real


def _evaluate(c0, x):
  c = c0.with_param_vec(x)
  return (c.reprojected.points - c.point_table.points)[c0.inliers].ravel()


def _dense_polish(calib, x, iters=60):
  """Converged optimum of the reference's residual function on the inlier set of `calib`, for problems too large for
  scipy's exact (SVD) trust-region solver: Levenberg-Marquardt on the DENSE normal equations of the reference's own
  `evaluate`, Jacobian by scipy's 3-point differences with the reference's sparsity (column scaling like x_scale='jac';
  the damping starts at 1e-8 in the scaled space -- the 12 gauge directions have zero gradient -- and adapts).  Stops
  when two consecutive accepted steps reduce the cost by less than 1e-14 relative.  Independent of the HIP code."""
  from scipy.optimize._numdiff import approx_derivative, group_columns
  from scipy.sparse import csr_matrix
  S = csr_matrix(calib.sparsity_matrix)
  groups = group_columns(S)
  fun = lambda v: _evaluate(calib, v)
  f = fun(x)
  cost = 0.5 * f @ f
  lam, small = 1e-8, 0
  for it in range(iters):
    J = csr_matrix(approx_derivative(fun, x, method='3-point', f0=f, sparsity=(S, groups)))
    H = (J.T @ J).toarray()
    g = J.T @ f
    d = np.sqrt(np.diag(H))
    d[d == 0] = 1
    Hs = H / d[:, None] / d[None, :]
    accepted = False
    for _ in range(12):
      step = -np.linalg.solve(Hs + lam * np.eye(x.size), g / d) / d
      fn = fun(x + step)
      cn = 0.5 * fn @ fn
      if cn <= cost:
        accepted = True
        break
      lam *= 10.0
    if not accepted:
      break
    rel = (cost - cn) / cost
    x, f, cost = x + step, fn, cn
    lam = max(lam * 0.1, 1e-12)
    small = small + 1 if rel < 1e-14 else 0
    if small >= 2:
      break
  return x, cost


def _reference_runs(calib, ref, out, ba_kwargs, ao_kwargs, run_ao, exact_polish):
  """bundle_adjust / adjust_outliers through the reference entry points + tight optima + self-sensitivity."""
  error_stats = ref.optimization_calibration.error_stats
  select_threshold = ref.optimization_calibration.select_threshold
  log = io.StringIO()
  import logging
  handler = logging.StreamHandler(log)
  logger = logging.getLogger("calibration")
  logger.addHandler(handler)
  logger.setLevel(logging.INFO)
  logger.propagate = False

  def polish(c, x, loss='linear', f_scale=1.0):
    if exact_polish:
      res_t = _Spy().real(lambda v: _evaluate(c, v), x, jac='3-point', x_scale='jac', tr_solver='exact', loss=loss,
                          f_scale=f_scale, ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400, method='trf')
      return res_t.x, res_t.cost, res_t.status, res_t.optimality
    assert loss == 'linear'
    xt, cost = _dense_polish(c, x)
    return xt, cost, -1, np.nan

  def ao_args():
    kw = dict(ao_kwargs)
    auto_scale = kw.pop("auto_scale", None)
    return dict(num_adjustments=3, select_outliers=select_threshold(quantile=0.75, factor=5.0),
                select_scale=select_threshold(quantile=0.75, factor=auto_scale) if auto_scale is not None else None,
                loss=kw.get("loss", 'linear'), tolerance=1e-4)

  try:
    with _Spy() as spy:
      ba = calib.bundle_adjust(**ba_kwargs)
      res = spy.results[-1]
    out["ba_kwargs_json"] = np.array(json.dumps(ba_kwargs))
    out["ba_x"] = ba.param_vec          # canonicalised by with_param_vec -> from_matrix (rtvec.py:29-32)
    out["ba_x_raw"] = res.x             # scipy's res.x (raw rotation vectors)
    out["ba_cost"], out["ba_optimality"] = res.cost, res.optimality
    out["ba_nfev"], out["ba_njev"], out["ba_status"] = res.nfev, res.njev, res.status
    out["ba_rms"] = error_stats(ba.reprojection_error).rms
    out["ba_log"] = np.array(log.getvalue())
    xt, cost_t, status_t, opt_t = polish(calib, res.x, ba_kwargs.get("loss", "linear"), ba_kwargs.get("f_scale", 1.0))
    out["ba_tight_x"], out["ba_tight_cost"], out["ba_tight_status"] = xt, cost_t, status_t
    out["ba_tight_rms"] = error_stats(calib.with_param_vec(xt).reprojection_error).rms
    pert = []
    for k in range(N_PERT):
      with _Spy(PERT_SIGMA, seed=100 + k) as spy:
        bp = calib.bundle_adjust(**ba_kwargs)
        pert.append((error_stats(bp.reprojection_error).rms, spy.results[-1].nfev, spy.results[-1].cost))
    out["ba_pert_rms"] = np.array([p[0] for p in pert])
    out["ba_pert_nfev"] = np.array([p[1] for p in pert])
    out["ba_pert_cost"] = np.array([p[2] for p in pert])

    if run_ao:
      with _Spy() as spy:
        ao = calib.adjust_outliers(**ao_args())
        results = list(spy.results)
      out["ao_kwargs_json"] = np.array(json.dumps(ao_kwargs))
      out["ao_x"] = ao.param_vec
      out["ao_x_raw"] = results[-1].x
      out["ao_inliers"] = ao.inliers
      out["ao_rms"] = error_stats(ao.reprojection_error).rms
      out["ao_rms_inliers"] = error_stats(ao.reprojection_inliers).rms
      out["ao_nfev"] = np.array([r.nfev for r in results])
      out["ao_cost"] = np.array([r.cost for r in results])
      out["ao_log"] = np.array(log.getvalue())
      if ao_kwargs.get("loss", "linear") == "linear":
        # converged optimum of the reference's residual function on the final inlier set: the value any converged
        # solver must reach (hard part 1).  scipy's default LSMR trust-region solver does not converge
        # tightly on these problems (hundreds of evaluations, status 0), so the polish uses scipy's exact (SVD)
        # trust-region solver with 3-point differences of the reference's own `evaluate` (dense Gauss-Newton on the
        # same differences for the big configurations).
        xt, cost_t, status_t, opt_t = polish(ao, results[-1].x)
        tight = ao.with_param_vec(xt)
        out["ao_tight_rms"] = error_stats(tight.reprojection_error).rms
        out["ao_tight_rms_inliers"] = error_stats(tight.reprojection_inliers).rms
        out["ao_tight_cost"] = cost_t
        out["ao_tight_status"] = status_t
        out["ao_tight_optimality"] = opt_t
      pert = []
      for k in range(N_PERT):
        with _Spy(PERT_SIGMA, seed=200 + k):
          ap = calib.adjust_outliers(**ao_args())
        pert.append((error_stats(ap.reprojection_error).rms, error_stats(ap.reprojection_inliers).rms,
                     int(np.sum(ap.inliers != ao.inliers))))
      out["ao_pert_rms"] = np.array([p[0] for p in pert])
      out["ao_pert_rms_inliers"] = np.array([p[1] for p in pert])
      out["ao_pert_mask_diff"] = np.array([p[2] for p in pert])
  finally:
    logger.removeHandler(handler)


def run_case(name):
  cfg, mutate, ba_kwargs, run_ao, store_j = CASES[name]
  rig = synthetic.make_rig(cfg)
  rig.name = name
  if mutate is not None:
    rig = mutate(rig)
  calib, ref = build_reference.reference_calibration(rig)
  error_stats = ref.optimization_calibration.error_stats

  out = synthetic.rig_to_arrays(rig)
  x0 = calib.param_vec
  r0 = _evaluate(calib, x0)
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

  _reference_runs(calib, ref, out, ba_kwargs, AO_KWARGS.get(name, {}), run_ao, exact_polish=True)

  os.makedirs(GOLDEN_DIR, exist_ok=True)
  path = os.path.join(GOLDEN_DIR, f"{name}.npz")
  np.savez_compressed(path, **out)
  print(f"{name}: n={x0.size} m={r0.size} rms0={float(out['rms0']):.4f} ba_rms={float(out['ba_rms']):.6f} "
        f"nfev={int(out['ba_nfev'])} pert spread {np.abs(out['ba_pert_rms'] - out['ba_rms']).max():.1e} "
        f"-> {path} ({os.path.getsize(path) / 1024:.0f} KiB)", flush=True)


def run_big_case(name):
  """Result-only golden of a BASELINE configuration: the rig is regenerated from its seed on the test side."""
  import time
  cfg, frames = BIG_CASES[name]
  rig = synthetic.make_rig(cfg, frames=frames)
  rig.name = name
  calib, ref = build_reference.reference_calibration(rig)
  error_stats = ref.optimization_calibration.error_stats
  out = dict(config=np.array(cfg), frames=np.array(rig.valid.shape[1]),
             points_sum=np.array(rig.points.sum()), points_abs_sum=np.array(np.abs(rig.points).sum()),
             valid_count=np.array(int(rig.valid.sum())), shape=np.array(rig.valid.shape))
  x0 = calib.param_vec
  t0 = time.time()
  r0 = _evaluate(calib, x0)
  out["x0"] = x0
  out["r0_sum"], out["r0_sq"], out["r0_size"] = r0.sum(), r0 @ r0, r0.size
  out["r0_head"] = r0[:64]
  out["rms0"] = error_stats(calib.reprojection_error).rms
  full = {}
  _reference_runs(calib, ref, full, {}, {}, True, exact_polish=False)
  for k, v in full.items():
    if k == "ao_inliers":
      out["ao_inliers_packed"] = np.packbits(v.ravel())
    else:
      out[k] = v
  out["seconds"] = time.time() - t0
  path = os.path.join(GOLDEN_DIR, f"{name}.npz")
  np.savez_compressed(path, **out)
  print(f"{name}: n={x0.size} m={r0.size} ba_rms={float(out['ba_rms']):.6f} ao_rms_inliers={float(out['ao_rms_inliers']):.6f} "
        f"tight {float(out['ao_tight_rms_inliers']):.6f} pert spread ba {np.abs(out['ba_pert_rms'] - out['ba_rms']).max():.1e} "
        f"ao {np.abs(out['ao_pert_rms_inliers'] - out['ao_rms_inliers']).max():.1e} mask diffs {out['ao_pert_mask_diff']} "
        f"in {out['seconds']:.0f} s -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)", flush=True)


def run_mixed_case(name="tiny_mixed"):
  """Cameras of DIFFERENT distortion models in one rig.  Every reference object accepts that (a ParamList of independent
  Camera objects, optimization/parameters.py:54-85) and `evaluate` / `reprojection_error` work, but `bundle_adjust` does
  not: Calibration.sparsity_matrix reshapes the cameras block to [n_cameras, -1] (calibration.py:179), which raises for a
  ragged block.  The fixture pins what the reference CAN compute -- x0, r0, err0, rms0, a dense 2-point Jacobian of its
  `evaluate` -- and records the exception of its bundle_adjust."""
  from scipy.optimize._numdiff import approx_derivative
  rig = synthetic.make_rig(name)
  rig.name = name
  calib, ref = build_reference.reference_calibration(rig)
  error_stats = ref.optimization_calibration.error_stats
  out = synthetic.rig_to_arrays(rig)
  x0 = calib.param_vec
  r0 = _evaluate(calib, x0)
  out["x0"], out["r0"], out["inliers0"] = x0, r0, calib.inliers
  out["err0"] = calib.reprojection_error
  out["rms0"] = error_stats(out["err0"]).rms
  out["J_dense"] = approx_derivative(lambda v: _evaluate(calib, v), x0, method='2-point', f0=r0)
  try:
    calib.bundle_adjust()
    out["ba_error"] = np.array("")
  except Exception as e:   # noqa: BLE001 -- the point of the fixture
    out["ba_error"] = np.array(f"{type(e).__name__}: {e}")
  path = os.path.join(GOLDEN_DIR, f"{name}.npz")
  np.savez_compressed(path, **out)
  print(f"{name}: n={x0.size} m={r0.size} rms0={float(out['rms0']):.4f} reference bundle_adjust -> {str(out['ba_error'])!r} "
        f"-> {path} ({os.path.getsize(path) / 1024:.0f} KiB)", flush=True)


def full_case_points(x0, seed=7):
  """The seeded perturbed point x1 and direction v of the full-size goldens (shared with the tests)."""
  rng = np.random.default_rng(seed)
  x1 = x0 + rng.normal(size=x0.size) * 1e-3 * np.maximum(1.0, np.abs(x0)) * 1e-1
  v = rng.normal(size=x0.size) * np.maximum(1e-3, 1e-3 * np.abs(x0))
  return x1, v


def run_full_case(name):
  """Evaluation-only golden of a BASELINE configuration at its stated size, from the real reference: what the bench
  measures (cfg3) and the 8- / 2-GPU configurations are pinned by `evaluate` (calibration.py:204-206),
  `reprojection_error` + `error_stats` (calibration.py:134-136,304-310) at two points."""
  import time
  cfg = FULL_CASES[name]
  rig = synthetic.make_rig(cfg)
  calib, ref = build_reference.reference_calibration(rig)
  error_stats = ref.optimization_calibration.error_stats
  t0 = time.time()
  out = dict(config=np.array(cfg), frames=np.array(rig.valid.shape[1]), shape=np.array(rig.valid.shape),
             points_sum=np.array(rig.points.sum()), points_abs_sum=np.array(np.abs(rig.points).sum()),
             valid_count=np.array(int(rig.valid.sum())))
  x0 = calib.param_vec
  x1, v = full_case_points(x0)
  out["x0"], out["x1"], out["v"] = x0, x1, v
  for tag, x in (("0", x0), ("1", x1)):
    r = _evaluate(calib, x)
    c = calib.with_param_vec(x)
    es = error_stats(c.reprojection_error)
    stride = max(1, r.size // 4096)
    out["r%s_size" % tag], out["r%s_sum" % tag], out["r%s_sq" % tag] = r.size, r.sum(), r @ r
    out["r%s_abs_sum" % tag] = np.abs(r).sum()
    out["r%s_head" % tag], out["r%s_stride" % tag], out["r%s_sample" % tag] = r[:64], stride, r[::stride]
    # weighted checksum: sensitive to any permutation of the residual order
    out["r%s_wsum" % tag] = float(np.dot(r, np.cos(np.arange(r.size) * 0.001)))
    out["rms%s" % tag], out["mse%s" % tag], out["n%s" % tag] = es.rms, es.mse, es.n
    out["quantiles%s" % tag] = np.asarray(es.quantiles)
  h = 1e-4
  fp, fm = _evaluate(calib, x0 + h * v), _evaluate(calib, x0 - h * v)
  out["dd_h"], out["dd"] = h, (0.5 * fp @ fp - 0.5 * fm @ fm) / (2 * h)    # ~ g(x0) . v
  out["seconds"] = time.time() - t0
  path = os.path.join(GOLDEN_DIR, f"{name}.npz")
  np.savez_compressed(path, **out)
  print(f"{name}: n={x0.size} m={int(out['r0_size'])} rms0={float(out['rms0']):.6f} rms1={float(out['rms1']):.6f} "
        f"dd={float(out['dd']):.6e} in {out['seconds']:.0f} s -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)", flush=True)


def main(argv):
  names = argv or list(CASES)
  for n in names:
    if n == "tiny_mixed":
      run_mixed_case(n)
    elif n == "tiny_fishmix5":
      run_mixed_case(n)
    elif n in FULL_CASES:
      run_full_case(n)
    elif n in BIG_CASES:
      run_big_case(n)
    else:
      run_case(n)


if __name__ == "__main__":
  sys.dont_write_bytecode = True
  main(sys.argv[1:])
