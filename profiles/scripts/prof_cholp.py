"""Phase cycles of the panel Cholesky (k_cholp_panel, summed over its panels) + wall time of the whole factor / solve."""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
c = calibration.from_rig(synthetic.make_rig("tiny"))
rng = np.random.default_rng(0)
with Handle(c) as h:
    for ns in (200, 286, 400, 700):
        M = rng.normal(size=(ns + 20, ns)); S = M.T @ M / ns + 0.1 * np.eye(ns); rhs = rng.normal(size=ns)
        h.debug_chol(S, rhs, reg=0.05, blocked=7)
        st = h.debug_chol(S, rhs, reg=0.05, blocked=7)[:6]
        print("ns %4d k_cholp_panel cycles (100 MHz? shader clock): load %d  first factor %d  panel solve %d  trailing+look-ahead %d  invert %d  store %d | total %d"
              % ((ns,) + tuple(int(v) for v in st) + (int(sum(st)),)), flush=True)
