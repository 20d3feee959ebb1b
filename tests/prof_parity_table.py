"""profiles/parity_table.md: per fixture the reference's end point, its own reproducibility (10 perturbed runs), and the two
HIP routes to the same problem -- the reference's scipy driver on the HIP fun + jac (protocol B) and the native solver --
with the number of function evaluations of each.  Run on the GPU box:  python tests/prof_parity_table.py > gpurun_out/parity_table.md"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from scipy.optimize import least_squares
from multical_amd import synthetic
from multical_amd.backend import Handle
from util import load_golden, mirror, GOLDEN

SMALL = ["cfg1", "tiny_handeye", "tiny_fixintr", "tiny_huber", "tiny", "tiny_rolling", "tiny_fisheye", "tiny_rational",
         "tiny_thin_prism", "tiny_tilted", "tiny_edge", "tiny_pin4", "tiny_softl1", "tiny_boards", "tiny_bigboard"]
BIG = ["cfg2", "cfg3_40", "cfg4_40", "cfg5_40", "manypairs"]


def rms_of(h, x):
  e, v = h.reprojection_error(x)
  return float(np.sqrt(np.mean(e[v] ** 2)))


rows = []
for name in SMALL + BIG:
  if name in BIG:
    g = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))
    rig = synthetic.make_rig(str(g["config"]))
  else:
    g, rig = load_golden(name)
  kw = json.loads(str(g["ba_kwargs_json"])) if "ba_kwargs_json" in g else {}
  loss, f_scale = kw.get("loss", "linear"), kw.get("f_scale", 1.0)
  with Handle(mirror(rig)) as h:
    t0 = time.time()
    res = least_squares(h.residuals, g["x0"], jac=h.jacobian, x_scale='jac', ftol=kw.get("tolerance", 1e-4),
                        max_nfev=kw.get("max_iterations", 100), method='trf', loss=loss, f_scale=f_scale)
    rms_b, t_b = rms_of(h, res.x), time.time() - t0
    nat = h.solve(g["x0"], tolerance=kw.get("tolerance", 1e-4), loss=loss, f_scale=f_scale, max_iterations=kw.get("max_iterations", 100))
    rms_n = rms_of(h, nat.x)
  ref = float(g["ba_rms"])
  pert = np.asarray(g["ba_pert_rms"])
  rows.append(dict(name=name, ref=ref, ref_nfev=int(g["ba_nfev"]), spread=float(np.abs(pert - ref).max()), sigma=float(pert.std()),
                   n_pert=int(pert.size), tight=float(g["ba_tight_rms"]) if "ba_tight_rms" in g else float("nan"),
                   rms_b=rms_b, nfev_b=int(res.nfev), rms_n=rms_n, nfev_n=int(nat.nfev), loss=loss))
  print(f"# {name} done ({t_b:.1f} s scipy-driven)", file=sys.stderr, flush=True)

print("# Parity table: final reprojection RMS (px) at the reference's default tolerance (ftol = 1e-4, max_nfev = 100)\n")
print("Reference = unmodified `Calibration.bundle_adjust` (tests/golden/*.npz, oracle/make_golden.py).  `spread` = max |RMS of a")
print("perturbed reference run - RMS of the reference run| over N re-runs of the reference with N(0, 1e-12 px) noise on its own")
print("residual function (oracle/make_pert.py): the resolution to which the reference's end point is defined.  `converged` =")
print("optimum of the reference's residual function (tight polish).  B = the reference's own scipy driver on the HIP `fun` +")
print("analytic `jac` (protocol B); N = the native HIP solver.  |d| columns are |RMS - reference RMS|.\n")
print("| fixture | loss | reference RMS | nfev | spread (max) | spread (sigma) | N runs | converged RMS | B: RMS | B: nfev | B: \\|d\\| | N: RMS | N: nfev | N: \\|d\\| | N within 1e-6 px |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
  db, dn = abs(r["rms_b"] - r["ref"]), abs(r["rms_n"] - r["ref"])
  print(f"| {r['name']} | {r['loss']} | {r['ref']:.9f} | {r['ref_nfev']} | {r['spread']:.1e} | {r['sigma']:.1e} | {r['n_pert']} | {r['tight']:.9f} | "
        f"{r['rms_b']:.9f} | {r['nfev_b']} | {db:.1e} | {r['rms_n']:.9f} | {r['nfev_n']} | {dn:.1e} | {'yes' if dn <= 1e-6 else 'no'} |")
json.dump(rows, open(os.path.join("gpurun_out", "parity_table.json"), "w"), indent=1)
