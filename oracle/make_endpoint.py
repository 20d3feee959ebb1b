"""TEST INFRASTRUCTURE: END POINTS of the unmodified reference at the BASELINE configurations' STATED sizes.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_endpoint cfg3 ba          # Calibration.bundle_adjust()
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_endpoint cfg3 ao          # adjust_outliers as Workspace.calibrate drives it
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_endpoint cfg3 aor         # ... with loss='soft_l1', auto_scale=2.0
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_endpoint cfg3 pert 100 101   # self-sensitivity re-runs (seeds)
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_endpoint cfg3 tight       # converged optimum of the reference's residual function
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_endpoint merge cfg3       # parts -> tests/golden/cfg3_endpoint.npz

Runs `Calibration.bundle_adjust` (/root/reference/multical/optimization/calibration.py:199-212) and
`Calibration.adjust_outliers` (:254-268, arguments of workspace.py:228-247) of the real reference (oracle/refload.py) on the
seeded synthetic rig of a BASELINE configuration at full size -- hours of one host core -- and records what the reference
returned, scipy's iteration table, the number of `evaluate` calls, wall time and peak memory, all MEASURED (BASELINE.md only
extrapolated them).  The rig is regenerated from its seed on the test side (checksums of the observation table are stored).
"""
import io
import os
import sys
import json
import time
import resource
import logging

import numpy as np

from multical_amd import synthetic
from . import build_reference
from .make_golden import _Spy, PERT_SIGMA, GOLDEN_DIR

PART_DIR = os.path.join(GOLDEN_DIR, "_parts")


def _rig(cfg):
  rig = synthetic.make_rig(cfg)
  calib, ref = build_reference.reference_calibration(rig)
  head = dict(config=np.array(cfg), shape=np.array(rig.valid.shape), points_sum=np.array(rig.points.sum()),
              points_abs_sum=np.array(np.abs(rig.points).sum()), valid_count=np.array(int(rig.valid.sum())))
  return rig, calib, ref, head


class _Count(object):
  """Counts `Calibration.with_param_vec` calls (= calls of the reference's `evaluate` closure)."""

  def __init__(self, ref):
    self.cls = ref.optimization_calibration.Calibration
    self.n = 0

  def __enter__(self):
    self.real = self.cls.with_param_vec
    outer = self

    def counted(this, x):
      outer.n += 1
      return outer.real(this, x)
    self.cls.with_param_vec = counted
    return self

  def __exit__(self, *a):
    self.cls.with_param_vec = self.real


def _log():
  log = io.StringIO()
  handler = logging.StreamHandler(log)
  logger = logging.getLogger("calibration")
  logger.addHandler(handler)
  logger.setLevel(logging.INFO)
  logger.propagate = False
  return log


def _save(cfg, stage, out):
  os.makedirs(PART_DIR, exist_ok=True)
  path = os.path.join(PART_DIR, f"{cfg}_{stage}.npz")
  np.savez_compressed(path, **out)
  print(f"[{time.strftime('%H:%M:%S')}] {cfg} {stage} -> {path}", flush=True)


def run_ba(cfg):
  rig, calib, ref, out = _rig(cfg)
  error_stats = ref.optimization_calibration.error_stats
  log = _log()
  out["x0"] = calib.param_vec
  t0 = time.time()
  with _Spy() as spy, _Count(ref) as cnt:
    ba = calib.bundle_adjust()
    res = spy.results[-1]
  out["ba_seconds"] = time.time() - t0
  out["ba_evaluate_calls"] = cnt.n
  out["ba_peak_rss_gb"] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20
  out["ba_x"], out["ba_x_raw"] = ba.param_vec, res.x
  out["ba_cost"], out["ba_optimality"] = res.cost, res.optimality
  out["ba_nfev"], out["ba_njev"], out["ba_status"] = res.nfev, res.njev, res.status
  es = error_stats(ba.reprojection_error)
  out["ba_rms"], out["ba_quantiles"] = es.rms, np.asarray(es.quantiles)
  out["ba_log"] = np.array(log.getvalue())
  out["host"] = np.array(json.dumps(dict(cpu_count=os.cpu_count(), numpy=np.__version__,
                                         scipy=__import__("scipy").__version__)))
  print(log.getvalue(), flush=True)
  print(f"{cfg} ba: rms {float(out['ba_rms']):.9f} nfev {res.nfev} status {res.status} {out['ba_seconds']:.0f} s "
        f"{cnt.n} evaluate calls, peak {out['ba_peak_rss_gb']:.1f} GB", flush=True)
  _save(cfg, "ba", out)


def run_ao(cfg):
  rig, calib, ref, out = _rig(cfg)
  error_stats = ref.optimization_calibration.error_stats
  select_threshold = ref.optimization_calibration.select_threshold
  log = _log()
  t0 = time.time()
  with _Spy() as spy, _Count(ref) as cnt:
    ao = calib.adjust_outliers(num_adjustments=3, select_outliers=select_threshold(quantile=0.75, factor=5.0),
                               select_scale=None, loss='linear', tolerance=1e-4)
    results = list(spy.results)
  out["ao_seconds"] = time.time() - t0
  out["ao_evaluate_calls"] = cnt.n
  out["ao_peak_rss_gb"] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20
  out["ao_x"], out["ao_x_raw"] = ao.param_vec, results[-1].x
  out["ao_inliers_packed"] = np.packbits(ao.inliers.ravel())
  out["ao_rms"] = error_stats(ao.reprojection_error).rms
  out["ao_rms_inliers"] = error_stats(ao.reprojection_inliers).rms
  out["ao_nfev"] = np.array([r.nfev for r in results])
  out["ao_status"] = np.array([r.status for r in results])
  out["ao_cost"] = np.array([r.cost for r in results])
  out["ao_log"] = np.array(log.getvalue())
  print(log.getvalue(), flush=True)
  print(f"{cfg} ao: rms {float(out['ao_rms']):.9f} inliers {float(out['ao_rms_inliers']):.9f} nfev {out['ao_nfev']} "
        f"{out['ao_seconds']:.0f} s", flush=True)
  _save(cfg, "ao", out)


def run_ao_robust(cfg, loss='soft_l1', auto_scale=2.0):
  """Workspace.calibrate(loss=..., auto_scale=...) (workspace.py:239-244): the outlier loop with a robust loss whose soft margin is
  re-derived from the error quantile in every round (calibration.py:259-266)."""
  rig, calib, ref, out = _rig(cfg)
  error_stats = ref.optimization_calibration.error_stats
  select_threshold = ref.optimization_calibration.select_threshold
  log = _log()
  t0 = time.time()
  with _Spy() as spy:
    ao = calib.adjust_outliers(num_adjustments=3, select_outliers=select_threshold(quantile=0.75, factor=5.0),
                               select_scale=select_threshold(quantile=0.75, factor=auto_scale), loss=loss, tolerance=1e-4)
    results = list(spy.results)
  out["aor_kwargs_json"] = np.array(json.dumps(dict(loss=loss, auto_scale=auto_scale)))
  out["aor_seconds"] = time.time() - t0
  out["aor_x"], out["aor_x_raw"] = ao.param_vec, results[-1].x
  out["aor_inliers_packed"] = np.packbits(ao.inliers.ravel())
  out["aor_rms"] = error_stats(ao.reprojection_error).rms
  out["aor_rms_inliers"] = error_stats(ao.reprojection_inliers).rms
  out["aor_nfev"] = np.array([r.nfev for r in results])
  out["aor_status"] = np.array([r.status for r in results])
  out["aor_cost"] = np.array([r.cost for r in results])
  out["aor_log"] = np.array(log.getvalue())
  print(log.getvalue(), flush=True)
  print(f"{cfg} ao {loss} auto_scale {auto_scale}: rms {float(out['aor_rms']):.9f} inliers {float(out['aor_rms_inliers']):.9f} "
        f"nfev {out['aor_nfev']} status {out['aor_status']} {out['aor_seconds']:.0f} s", flush=True)
  _save(cfg, "aor", out)


def run_aor_pert(cfg, seeds, loss='soft_l1', auto_scale=2.0):
  """the reference's own reproducibility of the robust outlier loop: `aor` repeated with N(0, 1e-12 px) on its residual function"""
  rig, calib, ref, out = _rig(cfg)
  error_stats = ref.optimization_calibration.error_stats
  select_threshold = ref.optimization_calibration.select_threshold
  base_path = os.path.join(PART_DIR, f"{cfg}_aor.npz")
  if not os.path.exists(base_path):      # (parts of an earlier session are gone: the merged fixture holds the same arrays)
    base_path = os.path.join(GOLDEN_DIR, f"{cfg}_endpoint.npz")
  base = np.load(base_path)
  base_mask = np.unpackbits(base["aor_inliers_packed"])[:rig.valid.size].reshape(rig.valid.shape).astype(bool)
  for s in seeds:
    t0 = time.time()
    with _Spy(PERT_SIGMA, seed=s) as spy:
      ap = calib.adjust_outliers(num_adjustments=3, select_outliers=select_threshold(quantile=0.75, factor=5.0),
                                 select_scale=select_threshold(quantile=0.75, factor=auto_scale), loss=loss, tolerance=1e-4)
      nfev = [r.nfev for r in spy.results]
    part = dict(out, seed=s, rms=error_stats(ap.reprojection_error).rms, rms_inliers=error_stats(ap.reprojection_inliers).rms,
                mask_diff=int(np.sum(ap.inliers != base_mask)), nfev=np.array(nfev), seconds=time.time() - t0)
    print(f"{cfg} aor pert {s}: inliers rms {float(part['rms_inliers']):.9f} mask diff {part['mask_diff']} nfev {nfev} {part['seconds']:.0f} s", flush=True)
    _save(cfg, f"aorpert{s}", part)


def run_pert(cfg, seeds):
  """The reference's own reproducibility at this size: the same call with N(0, 1e-12 px) added to its residual function
  (oracle/make_golden.py explains why that moves the end point)."""
  rig, calib, ref, out = _rig(cfg)
  error_stats = ref.optimization_calibration.error_stats
  for s in seeds:
    t0 = time.time()
    with _Spy(PERT_SIGMA, seed=s) as spy:
      bp = calib.bundle_adjust()
      res = spy.results[-1]
    part = dict(out, seed=s, rms=error_stats(bp.reprojection_error).rms, nfev=res.nfev, status=res.status,
                cost=res.cost, x_raw=res.x, seconds=time.time() - t0)
    print(f"{cfg} pert {s}: rms {float(part['rms']):.9f} nfev {res.nfev} status {res.status} {part['seconds']:.0f} s", flush=True)
    _save(cfg, f"pert{s}", part)


def run_tight(cfg, max_iter=40):
  """SURVEY 7, protocol C at the STATED size: the CONVERGED optimum of the reference's own residual function (`evaluate`,
  calibration.py:204-206, on the reference's classes), reached from the reference's end point by Levenberg-Marquardt on the dense
  normal equations.  The Jacobian that steers the iteration is the analytic one of tests/hostmath (the reference's 72 evaluations per
  3-point Jacobian would take 5 minutes each here); the optimum belongs to the residual function, and it is VERIFIED with the
  reference alone: at the final point the gradient J_fd^T f from scipy's 3-point differences of `evaluate` over the reference's
  sparsity predicts a further cost reduction (`ba_tight_fd_predicted_rel`) far below the 1e-6 px the tests resolve.  Adds ba_tight_*
  to tests/golden/<cfg>_endpoint.npz."""
  from scipy.linalg import cho_factor, cho_solve
  from scipy.optimize._numdiff import approx_derivative, group_columns
  from scipy.sparse import csr_matrix
  from multical_amd import calibration as mirror_calibration
  from .make_golden import _evaluate
  here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, os.path.join(here, "tests"))
  from hostmath_lib import HostMath
  rig, calib, ref, head = _rig(cfg)
  error_stats = ref.optimization_calibration.error_stats
  path = os.path.join(GOLDEN_DIR, f"{cfg}_endpoint.npz")
  g = dict(np.load(path, allow_pickle=False))
  hm = HostMath(mirror_calibration.from_rig(rig))
  fun = lambda v: _evaluate(calib, v)
  x = np.array(g["ba_x_raw"], dtype=np.float64)
  t0 = time.time()
  f = fun(x)
  print(f"[{time.strftime('%H:%M:%S')}] {cfg} tight: one reference evaluate {time.time() - t0:.1f} s, m = {f.size}, n = {x.size}; "
        f"|f_ref - f_host|_max = {np.abs(f - hm.residuals(x)).max():.2e}", flush=True)
  cost = 0.5 * f @ f
  n = x.size
  lam, small = 1e-8, 0

  def scaled_system(J, f):
    H = (J.T @ J).toarray()
    gvec = J.T @ f
    d = np.sqrt(np.diag(H))
    d[d == 0] = 1
    H /= d[:, None]
    H /= d[None, :]
    return H, gvec, d

  for it in range(max_iter):
    H, gvec, d = scaled_system(hm.jacobian(x), f)
    accepted = False
    for _ in range(12):
      A = H.copy()
      A[np.diag_indices(n)] += lam
      step = -cho_solve(cho_factor(A, overwrite_a=True, check_finite=False), gvec / d) / d
      fn = fun(x + step)
      cn = 0.5 * fn @ fn
      if cn <= cost:
        accepted = True
        break
      lam *= 10.0
    if not accepted:
      print(f"  iteration {it}: no acceptable step (lam {lam:.1e}): converged to rounding", flush=True)
      break
    rel = (cost - cn) / cost
    print(f"[{time.strftime('%H:%M:%S')}]   iteration {it}: cost {cn:.12e} relative reduction {rel:.2e} |step|_inf {np.abs(step).max():.2e} lam {lam:.1e}", flush=True)
    x, f, cost = x + step, fn, cn
    lam = max(lam * 0.1, 1e-10)
    small = small + 1 if rel < 1e-15 else 0
    if small >= 2:
      break
  # verification with the reference alone: what would a Gauss-Newton step on ITS OWN 3-point differences still gain?
  S = csr_matrix(calib.sparsity_matrix)
  t1 = time.time()
  J_fd = csr_matrix(approx_derivative(fun, x, method='3-point', f0=f, sparsity=(S, group_columns(S))))
  H, gvec, d = scaled_system(J_fd, f)
  H[np.diag_indices(n)] += 1e-10
  gs = gvec / d
  predicted = 0.5 * gs @ cho_solve(cho_factor(H, overwrite_a=True, check_finite=False), gs)
  tight = calib.with_param_vec(x)
  out = dict(ba_tight_x=x, ba_tight_cost=np.array(cost), ba_tight_rms=np.array(error_stats(tight.reprojection_error).rms),
             ba_tight_fd_predicted_rel=np.array(predicted / cost), ba_tight_fd_gradient_inf=np.array(np.abs(gs).max()),
             ba_tight_seconds=np.array(time.time() - t0))
  print(f"{cfg} tight: rms {float(out['ba_tight_rms']):.12f} (end point {float(g['ba_rms']):.12f}), cost {cost:.12e}; the reference's own 3-point "
        f"Jacobian ({time.time() - t1:.0f} s) predicts a further relative cost reduction of {predicted / cost:.2e}; {time.time() - t0:.0f} s", flush=True)
  _save(cfg, "tight", dict(head, **out))
  g.update(out)
  np.savez_compressed(path, **g)
  print(f"{cfg}: ba_tight_* -> {path}", flush=True)


def merge(cfg):
  import glob
  out = {}
  path = os.path.join(GOLDEN_DIR, f"{cfg}_endpoint.npz")
  if os.path.exists(path):               # parts of earlier sessions are not kept: start from what the fixture already holds
    out.update({k: v for k, v in np.load(path, allow_pickle=False).items()})
  for stage in ("ba", "ao", "aor", "tight"):
    p = os.path.join(PART_DIR, f"{cfg}_{stage}.npz")
    if os.path.exists(p):
      out.update({k: v for k, v in np.load(p).items()})
  aps = sorted(glob.glob(os.path.join(PART_DIR, f"{cfg}_aorpert*.npz")))
  if aps and "aor_rms_inliers" in out:
    ps = [np.load(p) for p in aps]
    old = set(int(v) for v in out.get("aor_pert_seed", []))
    new = [p for p in ps if int(p["seed"]) not in old]
    out["aor_pert_seed"] = np.concatenate([np.asarray(out.get("aor_pert_seed", []), dtype=np.int64), [int(p["seed"]) for p in new]])
    keep = len(out["aor_pert_seed"]) - len(new)
    out["aor_pert_rms_inliers"] = np.concatenate([np.asarray(out.get("aor_pert_rms_inliers", []), dtype=np.float64)[:keep], [float(p["rms_inliers"]) for p in new]])
    out["aor_pert_mask_diff"] = np.concatenate([np.asarray(out.get("aor_pert_mask_diff", []), dtype=np.int64)[:keep], [int(p["mask_diff"]) for p in new]])
  perts = sorted(glob.glob(os.path.join(PART_DIR, f"{cfg}_pert*.npz")))
  if perts:
    ps = [np.load(p) for p in perts]
    out["ba_pert_seed"] = np.array([int(p["seed"]) for p in ps])
    out["ba_pert_rms"] = np.array([float(p["rms"]) for p in ps])
    out["ba_pert_nfev"] = np.array([int(p["nfev"]) for p in ps])
    out["ba_pert_status"] = np.array([int(p["status"]) for p in ps])
    out["ba_pert_cost"] = np.array([float(p["cost"]) for p in ps])
  np.savez_compressed(path, **out)
  print(f"{cfg}: {sorted(out)} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
  sys.dont_write_bytecode = True
  if sys.argv[1] == "merge":
    for c in sys.argv[2:]:
      merge(c)
  else:
    cfg, stage = sys.argv[1], sys.argv[2]
    if stage == "ba":
      run_ba(cfg)
    elif stage == "ao":
      run_ao(cfg)
    elif stage == "aor":
      run_ao_robust(cfg)
    elif stage == "aorpert":
      run_aor_pert(cfg, [int(s) for s in sys.argv[3:]])
    elif stage == "pert":
      run_pert(cfg, [int(s) for s in sys.argv[3:]])
    elif stage == "tight":
      run_tight(cfg)
    else:
      raise SystemExit(f"unknown stage {stage}")
