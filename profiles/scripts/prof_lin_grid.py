"""Persistent grid of k_linearize (mcba_debug_set_lin_grid; 0 = automatic): dominant kernel by HIP events and the whole evaluation step."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
for cfg, frames in (("cfg3", 500), ("cfg4", 1000), ("cfg5", 400)):
  c = calibration.from_rig(synthetic.make_rig(cfg, frames=frames))
  with Handle(c) as h:
    x0 = c.param_vec
    for grid in (0, 2048, 3072, 4096, 5120, 6144, 8192, 16384):
      h.set_lin_grid(grid)
      h.normal_equations(x0)
      tl = h.time_linearize(x0, 200)
      best = 1e9
      for rep in range(5):
        h.synchronize(); t0 = time.perf_counter()
        for k in range(200): h.normal_equations_device()
        h.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
      print(cfg, frames, "grid", grid, "k_linearize %.2f us  step %.2f us" % (1e3 * tl, 1e6 * best), flush=True)
