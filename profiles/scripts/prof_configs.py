"""Full-size BASELINE configs on the GPU: timings + final RMS of the complete Workspace.calibrate sequence."""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration, Workspace
from multical_amd.backend import Handle
import logging
for name in sys.argv[1:] or ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"]:
    t = time.time(); rig = synthetic.make_rig(name); tg = time.time() - t
    c = calibration.from_rig(rig); x0 = c.param_vec
    t = time.time(); h = Handle(c); tc = time.time() - t
    lin = h.time_linearize(x0, 10); res = h.time_residuals(x0, 10)
    t = time.time(); r = h.solve(x0); ts = time.time() - t
    print(f"{name}: shape {rig.valid.shape} n={h.n_params} obs={h.n_residuals//2} gen {tg:.1f}s create {tc*1e3:.0f} ms | linearize {lin*1e3:.1f} us residual {res*1e3:.1f} us | "
          f"solve {ts*1e3:.1f} ms nfev {r.nfev} njev {r.njev} status {r.status} cost {r.initial_cost:.3e}->{r.cost:.6e}", flush=True)
    h.close()
    ws = Workspace(c)
    t = time.time(); out = ws.calibrate(cameras=rig.optimize["cameras"], camera_poses=rig.optimize["camera_poses"]); tw = time.time() - t
    st = out.error_statistics(False); si = out.error_statistics(True)
    print(f"   Workspace.calibrate: {tw*1e3:.1f} ms  rms all {st.rms:.6f} inliers {si.rms:.6f} n {si.n}/{st.n}", flush=True)
    calibration.handle_cache.clear()
