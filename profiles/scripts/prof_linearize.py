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
    print("epilogue: reduce + S store", act[:,5].mean(), " Y", act[:,6].mean(), " M + record stores", (act[:,3]-act[:,5]-act[:,6]).mean())
    print("lifetime mean", act[:, 7].mean())
    # clock rate of s_memtime: span vs measured time
    print("linearize ms", h.time_linearize(x0, 20))
    for g in (2048, 3072, 4096, 4401, 6000):
        h.set_lin_grid(g); h.time_linearize(x0, 3)
        print("grid", g, "linearize ms", h.time_linearize(x0, 20))
    h.set_lin_grid(0)
