import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
rig = synthetic.make_rig("cfg3"); c = calibration.from_rig(rig); x0 = c.param_vec
with Handle(c) as h:
    h.time_linearize(x0, 5)
    p = h.linearize_profile(x0)
    act = p[p[:, 4] > 0]
    print("views active", len(act), "of", len(p), "mean count", act[:,4].mean())
    names = ["setup", "rows", "stage+mfma", "epilogue"]
    for i, n in enumerate(names): print(f"{n:12s} mean {act[:, i].mean():10.0f} cyc  median {np.median(act[:, i]):10.0f}  max {act[:, i].max()}")
    print("setup: That columns", act[:,5].mean(), " clear+mask loads", act[:,6].mean(), " compaction", (act[:,0]-act[:,5]-act[:,6]).mean())
    print("lifetime mean", act[:, 7].mean())
    # clock rate of s_memtime: span vs measured time
    print("linearize ms", h.time_linearize(x0, 20))
