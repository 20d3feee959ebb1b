"""Round-3 A/B timings on the GPU box: evaluation step, dominant kernel and LM iteration per configuration, for a list of
environment variants (each in its own process: the switches are read once).

    python tests/prof_r3.py [cfg3 cfg4 ...] -- VAR=VAL,VAR2=VAL2 ...      (variant "base" = no switch)
"""
import json, os, subprocess, sys, time

CODE = r'''
import sys, time, json, numpy as np, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle, make_options, check, _ptr
name = sys.argv[1]
rig = synthetic.make_rig(name); c = calibration.from_rig(rig); x0 = c.param_vec
out = dict(cfg=name)
with Handle(c) as h:
  cost, g, d = h.normal_equations(x0)
  opt = make_options()
  for _ in range(20): check(h.lib.mcba_normal_equations_device(h.h, C.byref(opt)))
  h.synchronize()
  ts = []
  for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(200): check(h.lib.mcba_normal_equations_device(h.h, C.byref(opt)))
    h.synchronize()
    ts.append((time.perf_counter() - t0) / 200 * 1e6)
  out["step_us"] = sorted(ts)[2]
  out["lin_us"] = h.time_linearize(x0, 50) * 1e3
  rng = np.random.default_rng(1)
  x1 = x0 + 1e-3 * rng.normal(size=x0.size)
  h.solve(x1, tolerance=1e-15, xtol=1e-15, gtol=1e-15, max_iterations=41)
  ts = []
  for rep in range(3):
    t0 = time.perf_counter(); res = h.solve(x1, tolerance=1e-15, xtol=1e-15, gtol=1e-15, max_iterations=41); dt = time.perf_counter() - t0
    ts.append(dt / max(res.nfev - 1, 1) * 1e6)
  out["lm_iter_us"] = sorted(ts)[1]; out["nfev"] = res.nfev; out["njev"] = res.njev; out["cost"] = res.cost
  out["lm_lin_us"] = sorted(ts)[1] * max(res.nfev - 1, 1) / max(res.njev, 1)
  t0 = time.perf_counter(); res = h.solve(x0); out["solve_ms"] = (time.perf_counter() - t0) * 1e3; out["solve_nfev"] = res.nfev
  out["solve_cost"] = res.cost
print("RESULT" + json.dumps(out))
'''

def main():
  args = sys.argv[1:]
  cfgs = args[:args.index("--")] if "--" in args else (args or ["cfg3"])
  variants = args[args.index("--") + 1:] if "--" in args else ["base"]
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  for cfg in cfgs:
    for var in variants:
      env = dict(os.environ)
      if var != "base":
        for kv in var.split(","):
          k, v = kv.split("=")
          env[k] = v
      p = subprocess.run([sys.executable, "-c", CODE, cfg], cwd=root, env=env, capture_output=True, text=True, timeout=900)
      line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
      if p.returncode != 0 or not line:
        print(cfg, var, "FAILED", p.stderr[-1500:], flush=True)
        continue
      r = json.loads(line[0][6:])
      print(f"{cfg:6s} {var:40s} step {r['step_us']:8.2f} us  k_linearize {r['lin_us']:7.2f} us  LM trial step {r['lm_iter_us']:8.1f} us, per linearisation {r['lm_lin_us']:8.1f} us "
            f"(nfev {r['nfev']}, njev {r['njev']}, cost {r['cost']:.9e})  default solve {r['solve_ms']:.2f} ms nfev {r['solve_nfev']} cost {r['solve_cost']:.9e}", flush=True)

if __name__ == "__main__":
  main()
