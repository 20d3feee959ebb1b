import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
rig = synthetic.make_rig("cfg3"); c = calibration.from_rig(rig); x0 = c.param_vec
rng = np.random.default_rng(5)
x1 = x0 + 2e-2 * rng.normal(size=x0.size)
with Handle(c) as h:
    res = h.solve(x1, max_iterations=60)
    t0 = time.perf_counter(); res = h.solve(x1, max_iterations=60); dt = time.perf_counter() - t0
    print("far start: nfev", res.nfev, "njev", res.njev, "status", res.status, "cost", res.initial_cost, "->", res.cost, "ms/trial", dt / (res.nfev - 1) * 1e3)
