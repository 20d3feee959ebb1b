import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
for cfg in (sys.argv[1:] or ["cfg3"]):
  c = calibration.from_rig(synthetic.make_rig(cfg))
  with Handle(c) as h:
    h.set_lsmr_fused(2)
    print(cfg, h.time_lsmr_iteration(c.param_vec, repeats=100), file=sys.stderr)
