"""Round 6, review item 1 (continued): which half of the default solver carries its offset from scipy's end point?

  part C  scipy's trust-region algebra (tests/lsmr_emulation.trf_lsmr) around the DEVICE's LSMR call (mcba_debug_lsmr_solve on the
          linearisation, scaling and damping scipy's driver hands over): if this lands where scipy lands, the offset is in the device's
          trust-region driver; if it lands where the device's solve lands, it is in the LSMR solve / the matrix-free products.
  part D  the first k Golub-Kahan steps (maxiter = k): device vs scipy's lsmr on mcba_jacobian's matrix, scalars and solution.

Run on the GPU box:  python profiles/scripts/prof_lsmr_bisect.py [names] > gpurun_out/r06_lsmr_bisect.json
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from scipy.sparse.linalg import lsmr
from multical_amd import synthetic
from multical_amd.backend import Handle
from util import load_golden, mirror, GOLDEN
from lsmr_emulation import trf_lsmr, scaled_operator

SMALL = ["cfg1", "tiny_handeye", "tiny_fixintr"]
BIG = ["cfg2", "cfg3_40", "cfg4_40", "cfg5_40", "manypairs"]
names = sys.argv[1].split(",") if len(sys.argv) > 1 else SMALL + BIG


def load(name):
  if name in SMALL:
    return load_golden(name)
  g = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))
  return g, synthetic.make_rig(str(g["config"]))


def rms_of(h, x):
  e, v = h.reprojection_error(x)
  return float(np.sqrt(np.mean(e[v] ** 2)))


out = {}
for name in names:
  g, rig = load(name)
  ref = float(g["ba_rms"])
  with Handle(mirror(rig)) as h:
    def device_call(x, scale, damp, J, f):
      gn, _, info = h.lsmr_solve(x, damp, scale=scale)
      return (gn, info["istop"], info["itn"], info["normr"], info["normar"], info["normA"], info["condA"], info["normx"])
    row = {}
    for label, solver in (("scipy_tr_scipy_lsmr", "scipy"), ("scipy_tr_device_lsmr", device_call)):
      calls = []
      res = trf_lsmr(h.residuals, h.jacobian, g["x0"], solver=solver, calls=calls)
      row[label] = dict(nfev=res["nfev"], status=res["status"], d_rms=rms_of(h, res["x"]) - ref, calls=[(c["istop"], c["itn"]) for c in calls])
    r = h.solve(g["x0"], tr_solver="lsmr")
    row["device_tr_device_lsmr"] = dict(nfev=r.nfev, status=r.status, d_rms=rms_of(h, r.x) - ref, calls=[(c["istop"], c["itn"]) for c in h.lsmr_trace()])
    # part D: the first k steps
    x0 = g["x0"]
    J, f = h.jacobian(x0), h.residuals(x0)
    si = np.asarray(J.power(2).sum(axis=0)).ravel() ** 0.5
    si[si == 0] = 1
    d = 1 / si
    damp = 0.01
    short = []
    for k in (1, 2, 5, 10, 20, 40, 80):
      gn, scale, info = h.lsmr_solve(x0, damp, maxiter=k)
      s = lsmr(scaled_operator(J, d), f, damp=damp, maxiter=k)
      rel = lambda a, b: float(abs(a - b) / abs(b)) if b != 0 else float(abs(a))
      short.append(dict(maxiter=k, istop=(info["istop"], int(s[1])), itn=(info["itn"], int(s[2])), normr=rel(info["normr"], s[3]), normar=rel(info["normar"], s[4]),
                        normA=rel(info["normA"], s[5]), condA=rel(info["condA"], s[6]), normx=rel(info["normx"], s[7]),
                        x=float(np.linalg.norm(gn - s[0]) / np.linalg.norm(s[0])), scale=float(np.abs(scale / d - 1).max())))
    row["first_steps"] = short
  out[name] = row
  print(name, {k: ("%+.2e" % v["d_rms"], v["calls"]) for k, v in row.items() if k != "first_steps"}, file=sys.stderr, flush=True)
  print(name, "first steps:", [(s["maxiter"], "%.1e" % s["normA"], "%.1e" % s["normr"], "%.1e" % s["x"]) for s in short], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
