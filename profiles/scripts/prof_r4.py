"""Round-4 A/B timings on the GPU box (each variant in its own process: the switches / the library path are read once).

    python tests/prof_r4.py ab  [cfg3 cfg4 ...] -- base=<libmcba.so> new=<libmcba.so> ...   evaluation step + k_linearize + LM step
    python tests/prof_r4.py frames [cfg3 cfg4 ...]      k_linearize with the views of a frame bound to nw waves
                                                        (mcba_debug_set_frame_groups: the load-balance bound of a
                                                        frame-level linearisation)
"""
import json, os, subprocess, sys

AB = r'''
import sys, time, json, numpy as np, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle, make_options, check
name = sys.argv[1]
rig = synthetic.make_rig(name); c = calibration.from_rig(rig); x0 = c.param_vec
out = dict(cfg=name)
with Handle(c) as h:
  cost, g, d = h.normal_equations(x0)
  out["cost"] = cost; out["gsum"] = float(np.abs(g).sum()); out["dsum"] = float(d.sum())
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
  out["lin_us"] = sorted(h.time_linearize(x0, 50) * 1e3 for _ in range(3))[1]
  rng = np.random.default_rng(1)
  x1 = x0 + 1e-3 * rng.normal(size=x0.size)
  h.solve(x1, tolerance=1e-15, xtol=1e-15, gtol=1e-15, max_iterations=41)
  ts = []
  for rep in range(3):
    t0 = time.perf_counter(); res = h.solve(x1, tolerance=1e-15, xtol=1e-15, gtol=1e-15, max_iterations=41); dt = time.perf_counter() - t0
    ts.append(dt / max(res.nfev - 1, 1) * 1e6)
  out["lm_iter_us"] = sorted(ts)[1]; out["nfev"] = res.nfev; out["lm_cost"] = res.cost
print("RESULT" + json.dumps(out))
'''

FRAMES = r'''
import sys, json, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
name = sys.argv[1]
rig = synthetic.make_rig(name); c = calibration.from_rig(rig); x0 = c.param_vec
out = dict(cfg=name, rows=[])
with Handle(c) as h:
  ref = h.normal_equations(x0)
  for nw in [0, 1, 2, 4, 8, 16, 0]:
    h.set_frame_groups(nw)
    cost, g, d = h.normal_equations(x0)
    assert cost == ref[0] and np.array_equal(g, ref[1]) and np.array_equal(d, ref[2]), nw   # same records, same assembly
    us = sorted(h.time_linearize(x0, 50) * 1e3 for _ in range(3))[1]
    out["rows"].append((nw, us))
print("RESULT" + json.dumps(out))
'''


def run(code, cfg, env):
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  p = subprocess.run([sys.executable, "-c", code, cfg], cwd=root, env=env, capture_output=True, text=True, timeout=900)
  line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
  if p.returncode != 0 or not line:
    print(cfg, "FAILED", p.stderr[-1500:], flush=True)
    return None
  return json.loads(line[0][6:])


def main():
  mode, args = sys.argv[1], sys.argv[2:]
  cfgs = args[:args.index("--")] if "--" in args else (args or ["cfg3"])
  if mode == "frames":
    for cfg in cfgs:
      r = run(FRAMES, cfg, dict(os.environ))
      if r:
        for nw, us in r["rows"]:
          label = "largest-first list of views (product)" if nw == 0 else f"views of a frame on {nw} wave(s) of its own"
          print(f"{cfg:6s} {label:45s} k_linearize {us:7.2f} us", flush=True)
    return
  variants = args[args.index("--") + 1:] if "--" in args else ["new="]
  for cfg in cfgs:
    for var in variants:
      tag, _, rest = var.partition("=")
      env = dict(os.environ)
      for kv in rest.split(","):
        if not kv:
          continue
        if "=" in kv:
          k, v = kv.split("=")
          env[k] = v
        else:
          env["MCBA_LIB_PATH"] = os.path.abspath(kv)
      r = run(AB, cfg, env)
      if r:
        print(f"{cfg:6s} {tag:12s} step {r['step_us']:8.2f} us  k_linearize {r['lin_us']:7.2f} us  LM trial step {r['lm_iter_us']:8.1f} us (nfev {r['nfev']})  "
              f"cost {r['cost']:.12e} |g|1 {r['gsum']:.12e} sum diag {r['dsum']:.12e} lm cost {r['lm_cost']:.9e}", flush=True)


if __name__ == "__main__":
  main()
