"""Where the device LSMR mode and scipy part: LSMR iteration counts / stopping reasons per trust-region iteration.
python tests/prof_lsmr_diag.py cfg3_40"""
import json, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic
from multical_amd.backend import Handle
from util import load_golden, mirror, GOLDEN
import scipy.optimize._lsq.trf as trf

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_40"
if name in ("cfg2", "cfg3_40", "cfg4_40", "cfg5_40", "manypairs"):
  g = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)); rig = synthetic.make_rig(str(g["config"]))
else:
  g, rig = load_golden(name)
real = trf.lsmr
calls = []


def spy(A, b, damp=0.0, **kw):
  out = real(A, b, damp=damp, **kw)
  calls.append((damp * damp, out[2], out[1], out[3], out[4], out[5], out[6], out[7]))
  return out


trf.lsmr = spy
with Handle(mirror(rig)) as h:
  rs = h.solve_scipy(g["x0"], verbose=2)
  print("scipy mode: nfev", rs.nfev, "cost %.12e" % rs.cost)
  for i, c in enumerate(calls):
    print("  scipy lsmr call %d: reg_term %.17g itn %d istop %d normr %.10e normar %.6e normA %.6e condA %.4e normx %.10e" % ((i,) + c))
  # (the product library takes experiment switches through mcba_debug_set_switch only -- not from the environment -- and latches them
  #  on first use: the per-call (istop, itn, normr ...) of a solve now come back as values, Handle.lsmr_trace())
  h.set_lsmr_trace(True)
  rows = []
  h.set_log(lambda *a: rows.append(a))
  rl = h.solve(g["x0"], tr_solver="lsmr")
  print("device lsmr: nfev", rl.nfev, "cost %.12e" % rl.cost)
  for c in h.lsmr_trace():
    print("  device lsmr call %(iteration)d: damp %(damp).17g itn %(itn)d istop %(istop)d normr %(normr).10e normar %(normar).6e normA %(normA).6e condA %(condA).4e normx %(normx).10e" % c)
  for r in rows:
    print("  ", r)
