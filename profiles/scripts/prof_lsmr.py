"""solver = "lsmr" (scipy's TRF + LSMR step on the device) against the scipy mode and the reference's golden end points:
python tests/prof_lsmr.py [fixture ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic
from multical_amd.backend import Handle
from util import load_golden, mirror, GOLDEN

BIG = ["cfg2", "cfg3_40", "cfg4_40", "cfg5_40", "manypairs"]
names = sys.argv[1:] or ["cfg1", "tiny_handeye", "tiny_fixintr", "tiny_huber", "tiny_thin_prism", "tiny", "tiny_rolling", "tiny_fisheye",
                         "tiny_rational", "tiny_tilted", "tiny_edge", "tiny_pin4", "tiny_softl1", "tiny_bigboard", "tiny_fishmix"] + BIG


def rms_of(h, x):
  e, v = h.reprojection_error(x)
  return float(np.sqrt(np.mean(e[v] ** 2)))


print("| fixture | reference RMS | nfev | status | lsmr mode: RMS | nfev | status | \\|d\\| | s | scipy mode: RMS | nfev | \\|d\\| | s |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for name in names:
  if name in BIG:
    g = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))
    rig = synthetic.make_rig(str(g["config"]))
  else:
    g, rig = load_golden(name)
  kw = json.loads(str(g["ba_kwargs_json"])) if "ba_kwargs_json" in g else {}
  a = dict(tolerance=kw.get("tolerance", 1e-4), max_iterations=kw.get("max_iterations", 100), loss=kw.get("loss", "linear"),
           f_scale=kw.get("f_scale", 1.0))
  with Handle(mirror(rig)) as h:
    h.solve(g["x0"], tr_solver="lsmr", **a)
    t0 = time.perf_counter(); rl = h.solve(g["x0"], tr_solver="lsmr", **a); tl = time.perf_counter() - t0
    rms_l = rms_of(h, rl.x)
    rs, ts, rms_s = None, 0.0, float("nan")
    if os.environ.get("LSMR_NO_SCIPY") != "1":
      t0 = time.perf_counter(); rs = h.solve_scipy(g["x0"], verbose=0, **a); ts = time.perf_counter() - t0
      rms_s = rms_of(h, rs.x)
  ref = float(g["ba_rms"])
  print(f"| {name} | {ref:.9f} | {int(g['ba_nfev'])} | {int(g['ba_status'])} | {rms_l:.9f} | {rl.nfev} | {rl.status} | {abs(rms_l - ref):.1e} | {tl:.3f} | "
        f"{rms_s:.9f} | {rs.nfev if rs else '-'} | {abs(rms_s - ref):.1e} | {ts:.2f} |", flush=True)
