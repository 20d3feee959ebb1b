"""The default solver alone (tr_solver = "lsmr", automatic iteration form) for a kernel trace: one warm-up solve + N timed solves of a BASELINE rig.
   rocprofv3 --kernel-trace --stats -- python profiles/scripts/prof_lsmr_default.py cfg3 [frames] [solves]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else None
solves = int(sys.argv[3]) if len(sys.argv) > 3 else 3
c = calibration.from_rig(synthetic.make_rig(cfg, frames=frames) if frames else synthetic.make_rig(cfg))
x0 = c.param_vec
with Handle(c) as h:
  h.solve(x0, tr_solver="lsmr")
  for k in range(solves):
    t0 = time.perf_counter()
    res = h.solve(x0, tr_solver="lsmr")
    dt = time.perf_counter() - t0
    print(cfg, "solve %d: %.2f ms  nfev %d  LSMR iterations %d" % (k, 1e3 * dt, res.nfev, h.lsmr_iterations()), flush=True)
