import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
for F in [60, 125, 250, 500, 1000, 2000]:
    rig = synthetic.make_rig("cfg3", frames=F); c = calibration.from_rig(rig); x0 = c.param_vec
    with Handle(c) as h:
        ms = h.time_linearize(x0, 20); msr = h.time_residuals(x0, 20)
        p = h.linearize_profile(x0); act = p[p[:,4] > 0]
        print(F, "views", len(p), "active", len(act), "obs", h.n_residuals//2, "lin ms %.4f"%ms, "resid ms %.4f"%msr, "life mean", int(act[:,7].mean()))
