import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
rig = synthetic.make_rig("cfg3"); c = calibration.from_rig(rig); x0 = c.param_vec
with Handle(c) as h:
    h.error_stats(x0)
    t0 = time.perf_counter()
    for _ in range(20): h.error_stats(x0)
    print("error_stats ms", (time.perf_counter() - t0) / 20 * 1e3)
    t0 = time.perf_counter()
    for _ in range(20): h.error_stats(x0, quantiles=[0.75])
    print("error_stats 1 quantile ms", (time.perf_counter() - t0) / 20 * 1e3)
