"""wall clock of repeated solves of one config (clock ramp / first-call effects): python tests/prof_solve_repeat.py cfg3"""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
rig = synthetic.make_rig(name); c = calibration.from_rig(rig); x0 = c.param_vec
with Handle(c) as h:
  for k in range(8):
    t0 = time.perf_counter(); res = h.solve(x0); dt = time.perf_counter() - t0
    print(name, "solve", k, "ms %.3f" % (dt * 1e3), "nfev", res.nfev, "device linearize ms", getattr(res, "linearize_ms", None), flush=True)
  for k in range(3):
    h.time_linearize(x0, 50)
    t0 = time.perf_counter(); res = h.solve(x0); dt = time.perf_counter() - t0
    print(name, "after 50 linearisations: solve ms %.3f" % (dt * 1e3), flush=True)
