"""Grid of the persistent LSMR product kernel (debug switch MCBA_LSMR_GRID, one process per value):
python profiles/scripts/prof_lsmr_grid.py <grid> [cfg]"""
import sys, time, json
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration, _lib
from multical_amd.backend import Handle
grid = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
_lib.set_switch("MCBA_LSMR_GRID", grid)
c = calibration.from_rig(synthetic.make_rig(cfg))
x0 = c.param_vec
with Handle(c) as h:
  h.solve(x0, tr_solver="lsmr")
  ts = []
  for _ in range(3):
    t0 = time.perf_counter(); r = h.solve(x0, tr_solver="lsmr"); ts.append(time.perf_counter() - t0)
  t = sorted(ts)[1]
  print(cfg, "grid", grid, json.dumps(dict(seconds=t, nfev=r.nfev, lsmr_iterations=h.lsmr_iterations(),
                                           us_per_lsmr_iteration=t / h.lsmr_iterations() * 1e6)), flush=True)
