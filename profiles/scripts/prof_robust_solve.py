"""A robust-loss solve (soft_l1) of the 200-frame north-star rig under both solvers: input of profiles/scripts/quick_trace_cmd.sh."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
c = calibration.from_rig(synthetic.make_rig("cfg3", frames=200))
x0 = c.param_vec
with Handle(c) as h:
  for k in range(2):
    t0 = time.perf_counter(); r = h.solve(x0, tr_solver="lsmr", loss="soft_l1", f_scale=1.0); dt = time.perf_counter() - t0
    print("robust lsmr solve %.2f ms nfev %d its %d" % (1e3 * dt, r.nfev, h.lsmr_iterations()))
  t0 = time.perf_counter(); r = h.solve(x0, loss="soft_l1", f_scale=1.0); print("robust native %.2f ms nfev %d" % (1e3 * (time.perf_counter() - t0), r.nfev))
