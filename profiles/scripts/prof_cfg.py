"""rocprofv3 target: repeated solves of one BASELINE config (python tests/prof_cfg.py cfg4)."""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
rig = synthetic.make_rig(name); c = calibration.from_rig(rig); x0 = c.param_vec
with Handle(c) as h:
    h.solve(x0)
    t0 = time.perf_counter(); res = h.solve(x0); dt = time.perf_counter() - t0
    print(name, "solve ms", dt * 1e3, "nfev", res.nfev, "njev", res.njev, "status", res.status, "ms/trial", dt / max(res.nfev - 1, 1) * 1e3)
    res = h.solve(x0, tolerance=1e-15, xtol=1e-15, gtol=1e-15, max_iterations=21)
    t0 = time.perf_counter(); res = h.solve(x0, tolerance=1e-15, xtol=1e-15, gtol=1e-15, max_iterations=21); dt = time.perf_counter() - t0
    print(name, "long solve: nfev", res.nfev, "ms per trial step", dt / max(res.nfev - 1, 1) * 1e3)
