"""cProfile of the whole Workspace.calibrate sequence at a BASELINE config (host-side overhead around the kernels)."""
import sys, time, cProfile, pstats, io
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration, Workspace
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
rig = synthetic.make_rig(name)
if "--float32" in sys.argv:      # corners as cv2 detects them and as the reference's point table keeps them (tables.py:15-17)
  import numpy as np
  rig.points = rig.points.astype(np.float32)
  print("point table: float32")
if "--solver" in sys.argv:      # "lsmr" (the product default: the reference's end point) or "native" (exact steps)
  calibration.set_solver(sys.argv[sys.argv.index("--solver") + 1])
print("solver:", calibration.get_solver())
c = calibration.from_rig(rig)
ws = Workspace(c); ws.calibrate(cameras=rig.optimize["cameras"], camera_poses=rig.optimize["camera_poses"])
calibration.handle_cache.clear()
ws = Workspace(c)
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
out = ws.calibrate(cameras=rig.optimize["cameras"], camera_poses=rig.optimize["camera_poses"])
pr.disable()
print("calibrate ms", (time.perf_counter() - t) * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
