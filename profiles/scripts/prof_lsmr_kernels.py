"""rocprofv3 target: one solver="lsmr" solve of a BASELINE config (python tests/prof_lsmr_kernels.py cfg3)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
rig = synthetic.make_rig(name); c = calibration.from_rig(rig); x0 = c.param_vec
with Handle(c) as h:
    h.solve(x0, tr_solver="lsmr", max_iterations=2)
    t0 = time.perf_counter(); res = h.solve(x0, tr_solver="lsmr"); dt = time.perf_counter() - t0
    print(name, "lsmr solve ms", dt * 1e3, "nfev", res.nfev, "status", res.status)
