"""The LSMR iteration of the parity route at the north-star rig: two-launch form (k_lsmr_fused2 / k_lsmr_gather3), the same with the
per-observation state cached (round 6 experiment: mcba_debug_set_lsmr_fused(h, 3)), three-launch form
(k_lsmr_fused) and the six-launch form of round 4, same handle, same solve.   python profiles/scripts/prof_lsmr_iter.py [cfg3|cfg4|cfg5 ...]
(under `rocprofv3 --kernel-trace --stats` the per-kernel times of both forms land in one trace)"""
import sys, time, json
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle

for cfg in (sys.argv[1:] or ["cfg3"]):
  rig = synthetic.make_rig(cfg)
  c = calibration.from_rig(rig)
  x0 = c.param_vec
  with Handle(c) as h:
    out = {}
    for fused in (2, 3, 12, 1, 0, 2):       # (12: form 2 reading the frame-major tables -- the only source until round 6)
      h.set_lsmr_masks_form(fused == 12)
      h.set_lsmr_fused(2 if fused == 12 else fused)
      h.solve(x0, tr_solver="lsmr")
      ts = []
      for _ in range(3):
        t0 = time.perf_counter(); r = h.solve(x0, tr_solver="lsmr"); ts.append(time.perf_counter() - t0)
      t = sorted(ts)[1]
      itn = h.lsmr_iterations()
      e, v = h.reprojection_error(r.x)
      k_us = [1e3 * ms for ms in h.time_lsmr_iteration(x0, repeats=200)] if fused >= 2 else None
      out[{3: "two_launch_cached_state", 2: "two_launch", 12: "two_launch_masks_form", 1: "three_launch", 0: "six_launch"}[fused]] = dict(product_gather_kernel_us=k_us, seconds=t, nfev=r.nfev, status=r.status, lsmr_iterations=itn,
                                                      us_per_lsmr_iteration=t / max(itn, 1) * 1e6, cost=r.cost,
                                                      rms_px=float(np.sqrt(np.mean(e[v] ** 2))))
    print(cfg, json.dumps(out), flush=True)
