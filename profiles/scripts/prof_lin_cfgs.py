"""k_linearize time (HIP events) at several BASELINE configs: python tests/prof_lin_cfgs.py cfg2 cfg3 cfg4"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
for name in sys.argv[1:] or ["cfg2", "cfg3", "cfg4", "cfg5"]:
  rig = synthetic.make_rig(name); c = calibration.from_rig(rig); x0 = c.param_vec
  with Handle(c) as h:
    h.time_linearize(x0, 5)
    print(name, "n_obs", h.n_residuals // 2, "linearize us", round(h.time_linearize(x0, 30) * 1e3, 2), flush=True)
