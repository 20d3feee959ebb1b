"""A/B of k_linearize's observation source (TEST_MASKS=1: masks + slot tables; default: compacted tables): dominant kernel by HIP events and the whole evaluation step."""
import sys, os, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration, _lib
from multical_amd.backend import Handle
if os.environ.get("TEST_MASKS") == "1":
  _lib.set_switch("MCBA_LIN_COMPACT", "0")
for cfg, frames in (("cfg3", 500), ("cfg4", 1000), ("cfg5", 400), ("cfg2", 200)):
  c = calibration.from_rig(synthetic.make_rig(cfg, frames=frames))
  with Handle(c) as h:
    x0 = c.param_vec
    h.normal_equations(x0)
    import time
    tl = h.time_linearize(x0, 200)
    best = 1e9
    for rep in range(5):
      h.synchronize(); t0 = time.perf_counter()
      for k in range(200): h.normal_equations_device()
      h.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
    ts = best * 1e3
    print(cfg, frames, "k_linearize %.2f us  step %.2f us" % (1e3 * tl, 1e3 * ts), flush=True)
