"""k_tmat phase stamps (library built with -DMCBA_EXP_TMAT_PROF): python tests/prof_tmat.py [cfg]"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
rig = synthetic.make_rig(name); c = calibration.from_rig(rig); x0 = c.param_vec
with Handle(c) as h:
  h.time_linearize(x0, 5)
  for rep in range(2):
    _, tm = h.linearize_profile(x0, with_tmat=True)
  tm = tm[tm[:, 0] > 0]
  print("k_tmat workgroups", len(tm))
  if len(tm):
    d = np.diff(tm[:, :4], axis=1)
    for i, n in enumerate(["zero + local poses", "columns + chains", "table stores"]):
      print(f"{n:20s} mean {d[:, i].mean():9.0f} cyc  median {np.median(d[:, i]):9.0f}  max {d[:, i].max()}")
    print("lifetime mean", (tm[:, 3] - tm[:, 0]).mean(), "cyc; wall span (100 MHz ticks)", tm[:, 4].max() - tm[:, 4].min())
    print("first start .. last end (cycles)", tm[:, 3].max() - tm[:, 0].min())
