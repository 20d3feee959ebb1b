"""TEST INFRASTRUCTURE: where does scipy's OWN algorithm end when its Jacobian products are (nearly) exact?

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_exact_products cfg3_40 [n_double [n_longdouble]]
    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_exact_products cfg3 1 1          # full size: ~1 h

The reference's end point (`Calibration.bundle_adjust`, optimization/calibration.py:199-212) is the result of four to six trust-region
steps whose Gauss-Newton directions come from scipy's LSMR stopped at atol = btol = 1e-6 after hundreds of Golub-Kahan steps WITHOUT
reorthogonalisation.  On these Jacobians the bidiagonalisation loses orthogonality after 20 - 40 steps; from there on the rounding
errors of the two sparse products J_h v / J_h^T u decide when the stopping rule fires and how converged the step is.  This script runs
scipy's algorithm -- tests/lsmr_emulation.trf_lsmr, a transcription of scipy's trf_no_bounds that is checked bit for bit against
scipy.optimize.least_squares in tests/test_host.py -- on the REFERENCE's residual function (oracle/refload.py; analytic Jacobian from
tests/hostmath) in two arithmetics:

  double      scipy.sparse's products as the reference runs them (sequential double accumulation), with the rows of the problem in
              different orders: the run-to-run spread of the reference's arithmetic;
  longdouble  the same algorithm, the same scipy.sparse.linalg.lsmr, but the two products accumulated in 80-bit extended precision and
              rounded to double once: what the algorithm returns when the products are exact to rounding.

The device forms its products by per-lane partial sums and tree reductions, i.e. with errors of a few ulp instead of sqrt(nnz per
column) ulp: its end points coincide with the `longdouble` cluster (tests/test_gpu_lsmr.py::test_default_solver_lands_on_scipys_exact_
product_end_point), and the distance of that cluster from the reference's single run (-2.4e-6 px at 8 x 40 x 2) is the footprint of
the reference's own product rounding.  Results -> tests/golden/exact_products.json.
"""
import os
import sys
import json
import time

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(HERE, "tests"))
OUT = os.path.join(HERE, "tests", "golden", "exact_products.json")
LD = np.longdouble


def longdouble_solver(x, scale, damp, J, f):
  """scipy.sparse.linalg.lsmr on J_h = J diag(scale) with both products accumulated in extended precision"""
  from scipy.sparse.linalg import lsmr, LinearOperator
  Jl = J.astype(LD)
  JlT = Jl.T.tocsr()
  dl = scale.astype(LD)
  mv = lambda v: np.asarray(Jl @ (np.ravel(v).astype(LD) * dl), dtype=np.float64)
  rmv = lambda u: np.asarray(dl * (JlT @ np.ravel(u).astype(LD)), dtype=np.float64)
  return lsmr(LinearOperator(J.shape, matvec=mv, rmatvec=rmv, dtype=np.float64), f, damp=damp)


def run(name, n_double, n_long):
  from multical_amd import synthetic, calibration as mirror_calibration
  from hostmath_lib import HostMath
  from lsmr_emulation import trf_lsmr
  from . import build_reference
  from .make_golden import _evaluate, GOLDEN_DIR
  full = name in ("cfg3", "cfg4", "cfg5")
  g = dict(np.load(os.path.join(GOLDEN_DIR, f"{name}_endpoint.npz" if full else f"{name}.npz"), allow_pickle=False))
  rig = synthetic.make_rig(str(g["config"]))
  calib, ref = build_reference.reference_calibration(rig)
  error_stats = ref.optimization_calibration.error_stats
  hm = HostMath(mirror_calibration.from_rig(rig))
  x0 = np.array(g["x0"])
  assert np.array_equal(calib.param_vec, x0)
  ref_rms = float(g["ba_rms"])
  rms = lambda x: float(error_stats(calib.with_param_vec(x).reprojection_error).rms)
  runs = []
  for kind, count in (("double", n_double), ("longdouble", n_long)):
    for seed in range(count):
      perm = np.random.default_rng(seed).permutation(hm.m) if seed > 0 else np.arange(hm.m)
      fun = lambda x: _evaluate(calib, x)[perm]          # the reference's residual function (calibration.py:204-206)
      jac = lambda x: hm.jacobian(x)[perm]
      calls = []
      t0 = time.time()
      res = trf_lsmr(fun, jac, x0, solver="scipy" if kind == "double" else longdouble_solver, calls=calls)
      row = dict(arithmetic=kind, row_order_seed=seed, nfev=int(res["nfev"]), status=int(res["status"]), rms=rms(res["x"]),
                 calls=[(c["istop"], c["itn"]) for c in calls], seconds=time.time() - t0)
      row["rms_minus_reference"] = row["rms"] - ref_rms
      runs.append(row)
      print(f"[{time.strftime('%H:%M:%S')}] {name} {kind} order {seed}: rms - reference {row['rms_minus_reference']:+.3e} nfev {row['nfev']} "
            f"calls {row['calls']} {row['seconds']:.0f} s", flush=True)
      _store(name, ref_rms, int(g["ba_nfev"]), runs)


def _store(name, ref_rms, ref_nfev, runs):
  data = json.load(open(OUT)) if os.path.exists(OUT) else {}
  entry = dict(reference_rms=ref_rms, reference_nfev=ref_nfev, runs=runs)
  for kind in ("double", "longdouble"):
    d = [r["rms"] for r in runs if r["arithmetic"] == kind]
    if d:
      entry[f"{kind}_mean_rms"] = float(np.mean(d))
      entry[f"{kind}_mean_minus_reference"] = float(np.mean(d)) - ref_rms
      entry[f"{kind}_spread"] = float(np.max(d) - np.min(d))
  data[name] = entry
  with open(OUT, "w") as fh:
    json.dump(data, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
  sys.dont_write_bytecode = True
  run(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3, int(sys.argv[3]) if len(sys.argv) > 3 else 3)
