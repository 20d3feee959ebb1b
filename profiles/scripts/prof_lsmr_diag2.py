"""first trust-region steps of the device LSMR mode against scipy's, evaluation point by evaluation point"""
import json, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic
from multical_amd.backend import Handle
from util import load_golden, mirror, GOLDEN
from scipy.optimize import least_squares

name = sys.argv[1] if len(sys.argv) > 1 else "tiny_thin_prism"
if name in ("cfg2", "cfg3_40", "cfg4_40", "cfg5_40", "manypairs"):
  g = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)); rig = synthetic.make_rig(str(g["config"]))
else:
  g, rig = load_golden(name)
with Handle(mirror(rig)) as h:
  xs = []

  def fun(x):
    xs.append(x.copy())
    return h.residuals(x)

  least_squares(fun, g["x0"], jac=h.jacobian, x_scale='jac', ftol=1e-4, max_nfev=100, method='trf')
  for k in range(2, min(len(xs), 6) + 1):
    r = h.solve(g["x0"], tr_solver="lsmr", max_iterations=k)
    d = r.x - xs[k - 1]
    step = xs[k - 1] - xs[k - 2]
    print(f"after {k - 1} trial step(s): |x_device - x_scipy| = {np.abs(d).max():.3e} (rel. to the step: {np.linalg.norm(d) / max(np.linalg.norm(step), 1e-300):.3e}), "
          f"|step| = {np.linalg.norm(step):.6e}, worst entries {np.argsort(-np.abs(d))[:5]}")
