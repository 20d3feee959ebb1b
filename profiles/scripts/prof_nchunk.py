"""Chunk sums of the shared part (k_assemble -> k_shared_final): evaluation step against MCBA_NCHUNK_TARGET = (pair, chunk) workgroups aimed at.
   python profiles/scripts/prof_nchunk.py  (one subprocess per value: the switch is read once per process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r"""
import sys, time, os
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from multical_amd import synthetic, calibration, _lib
from multical_amd.backend import Handle
t = os.environ.get("TEST_NCHUNK")
if t: _lib.set_switch("MCBA_NCHUNK_TARGET", t)
for cfg in ("cfg3", "cfg4", "cfg5", "cfg2"):
  c = calibration.from_rig(synthetic.make_rig(cfg))
  with Handle(c) as h:
    x0 = c.param_vec
    h.normal_equations(x0)
    best = 1e9
    for rep in range(5):
      h.synchronize(); t0 = time.perf_counter()
      for k in range(200): h.normal_equations_device()
      h.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
    print("target", t or "default", cfg, "step %%.2f us" %% (1e6 * best), flush=True)
""" % (ROOT, ROOT)
for target in (sys.argv[1:] or ["", "512", "1024", "2048", "4096"]):
  r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, TEST_NCHUNK=target), capture_output=True, text=True)
  print(r.stdout.strip() if r.returncode == 0 else r.stderr[-400:], flush=True)
