import json, os, subprocess, sys
ROOT="/root/repo"
CODE = """
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
import numpy as np, time
out = {}
for cfg in ("cfg3","cfg4","cfg5","cfg2"):
  c = calibration.from_rig(synthetic.make_rig(cfg))
  with Handle(c) as h:
    h.set_lsmr_fused(2)
    h.time_lsmr_iteration(c.param_vec, repeats=50)
    k = [1e3 * v for v in h.time_lsmr_iteration(c.param_vec, repeats=300)]
    h.solve(c.param_vec, tr_solver="lsmr")
    ts=[]
    for _ in range(3):
      t0=time.perf_counter(); r=h.solve(c.param_vec, tr_solver="lsmr"); ts.append(time.perf_counter()-t0)
    out[cfg] = dict(kernels=k, solve_ms=sorted(ts)[1]*1e3, itn=h.lsmr_iterations(), cost=r.cost)
print(json.dumps(out))
"""
for variant in ("", "f2_DEPTH1"):
  lib = os.path.join(ROOT, "multical_amd", "_build" + ("_" + variant if variant else ""), "libmcba.so")
  env = dict(os.environ, MCBA_LIB_PATH=lib)
  r = subprocess.run([sys.executable, "-c", CODE % (ROOT, ROOT)], env=env, capture_output=True, text=True)
  print(variant or "depth2(product)", r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-400:])
