"""k_chol_blk phase stamps at ns = 140 (mcba_debug_chol mode 4) and ms per call of the automatic path."""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
c = calibration.from_rig(synthetic.make_rig("tiny"))
rng = np.random.default_rng(0)
with Handle(c) as h:
  for ns in (140,):
    M = rng.normal(size=(ns + 20, ns)); S = M.T @ M / ns + 0.1 * np.eye(ns); rhs = rng.normal(size=ns)
    ref = np.linalg.solve(S + 0.05 * np.eye(ns), rhs)
    p = h.debug_chol(S, rhs, reg=0.05, blocked=0)
    print("ns", ns, "err", np.abs(p - ref).max() / np.abs(ref).max())
    h.debug_chol(S, rhs, reg=0.05, blocked=4)
    st = h.debug_chol(S, rhs, reg=0.05, blocked=4)
    names = ["-", "first diagonal tile || load of the other tiles", "panels", "trailing + look-ahead factor", "back substitution", "load of the first tile (wave 0)", "-", "-"]
    for n, v in zip(names, st[:8]): print(f"  {n:48s} {v:10.0f} cycles")
    print("  total", sum(st[:6]))
