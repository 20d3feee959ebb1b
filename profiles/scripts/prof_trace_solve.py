import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
rig = synthetic.make_rig("cfg3"); c = calibration.from_rig(rig); x0 = c.param_vec
with Handle(c) as h:
    h.solve(x0)
    rng = np.random.default_rng(1)
    x1 = x0 + 1e-3 * rng.normal(size=x0.size)
    h.solve(x1, tolerance=1e-15, xtol=1e-15, gtol=1e-15, max_iterations=12)
    print("=====", file=sys.stderr, flush=True)
    h.solve(x1, tolerance=1e-15, xtol=1e-15, gtol=1e-15, max_iterations=12)
