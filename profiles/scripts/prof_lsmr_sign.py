"""Round 6, review item 1: where does the default solver's end point sit relative to scipy's -- at the level of ONE lsmr() call, of
the per-trust-region-iteration (istop, itn) sequence, and as a DISTRIBUTION over summation orders?

  part A  per fixture: scipy's trf_no_bounds + scipy.sparse.linalg.lsmr on the device's residuals / analytic Jacobian
          (tests/lsmr_emulation.trf_lsmr: every call's x, scale, damp and return tuple logged), then the DEVICE's lsmr_solve
          (mcba_debug_lsmr_solve) on the same linearisation, scale and damp, call by call; then the device's whole solve with its trace.
  part B  the device solve under different summation orders (product grid 1024 .. 4096 x the three iteration forms) at cfg3_40 / cfg3:
          the distribution of (RMS - reference RMS) next to the reference's own perturbed re-runs.

Run on the GPU box:  python profiles/scripts/prof_lsmr_sign.py [A|B|AB] > gpurun_out/r06_lsmr_sign.json
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic
from multical_amd.backend import Handle
from util import load_golden, mirror, GOLDEN
from lsmr_emulation import trf_lsmr

SMALL = ["cfg1", "tiny_handeye", "tiny_fixintr"]
BIG = ["cfg2", "cfg3_40", "cfg4_40", "cfg5_40", "manypairs"]
what = sys.argv[1] if len(sys.argv) > 1 else "AB"
names = sys.argv[2].split(",") if len(sys.argv) > 2 else SMALL + BIG


def load(name):
  if name in SMALL:
    return load_golden(name)
  g = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))
  return g, synthetic.make_rig(str(g["config"]))


def rms_of(h, x):
  e, v = h.reprojection_error(x)
  return float(np.sqrt(np.mean(e[v] ** 2)))


KEYS = ("istop", "itn", "normr", "normar", "normA", "condA", "normx")
out = {}
if "A" in what:
  for name in names:
    g, rig = load(name)
    ref = float(g["ba_rms"])
    with Handle(mirror(rig)) as h:
      calls = []
      t0 = time.time()
      res = trf_lsmr(h.residuals, h.jacobian, g["x0"], solver="scipy", calls=calls)
      row = dict(reference_rms=ref, reference_nfev=int(g["ba_nfev"]),
                 scipy=dict(nfev=res["nfev"], status=res["status"], d_rms=rms_of(h, res["x"]) - ref, seconds=time.time() - t0,
                            calls=[{k: c[k] for k in KEYS} for c in calls]))
      dev_calls = []
      for c in calls:   # the device's LSMR on the very linearisation, scaling and damping of every scipy call
        gn, scale, info = h.lsmr_solve(c["x"], c["damp"], scale=c["scale"])
        info["gn_rel_diff"] = float(np.linalg.norm(gn - c["gn_h"]) / np.linalg.norm(c["gn_h"]))
        info["scale_rel_diff"] = float(np.abs(scale / c["scale"] - 1).max())
        dev_calls.append(info)
      row["device_calls_on_scipys_iterates"] = dev_calls
      h.set_lsmr_trace(True)
      t0 = time.time()
      r = h.solve(g["x0"], tr_solver="lsmr")
      row["device"] = dict(nfev=r.nfev, status=r.status, d_rms=rms_of(h, r.x) - ref, seconds=time.time() - t0,
                           calls=[{k: c[k] for k in KEYS + ("damp", "Delta")} for c in h.lsmr_trace()])
      row["scipy_damp_Delta"] = [dict(damp=c["damp"], Delta=c["Delta"]) for c in calls]
    out[name] = row
    print(name, "scipy", [(c["istop"], c["itn"]) for c in row["scipy"]["calls"]], "%+.2e" % row["scipy"]["d_rms"],
          "| device on scipy's iterates", [(c["istop"], c["itn"]) for c in dev_calls],
          "| device solve", [(c["istop"], c["itn"]) for c in row["device"]["calls"]], "%+.2e" % row["device"]["d_rms"], file=sys.stderr, flush=True)

if "B" in what:
  dist = {}
  for name in ([n for n in names if n in ("cfg3_40", "cfg4_40", "cfg5_40", "cfg2", "manypairs")] + (["cfg3"] if "cfg3" in names or len(sys.argv) <= 2 else [])):
    if name == "cfg3":
      g = dict(np.load(os.path.join(GOLDEN, "cfg3_endpoint.npz"), allow_pickle=False))
      rig = synthetic.make_rig("cfg3")
    else:
      g, rig = load(name)
    ref = float(g["ba_rms"])
    rows = []
    with Handle(mirror(rig)) as h:
      for form in (2, 1, 0):
        h.set_lsmr_fused(form)
        for grid in (1024, 1536, 2048, 3072, 4096):
          h.set_lsmr_grid(grid)
          r = h.solve(g["x0"], tr_solver="lsmr")
          rows.append(dict(form=form, grid=grid, nfev=r.nfev, status=r.status, d_rms=rms_of(h, r.x) - ref,
                           calls=[(c["istop"], c["itn"]) for c in h.lsmr_trace()]))
    d = np.array([r["d_rms"] for r in rows])
    pert = (np.asarray(g["ba_pert_rms"]) - ref).tolist() if "ba_pert_rms" in g else []
    dist[name] = dict(reference_rms=ref, runs=rows, mean=float(d.mean()), std=float(d.std()), min=float(d.min()), max=float(d.max()),
                      n_negative=int((d < 0).sum()), n=int(d.size), reference_perturbed_minus_reference=pert)
    print(name, "device distribution: mean %+.2e std %.1e min %+.2e max %+.2e, negative %d / %d | reference's own perturbed runs: %s" %
          (d.mean(), d.std(), d.min(), d.max(), (d < 0).sum(), d.size, ["%+.1e" % p for p in pert]), file=sys.stderr, flush=True)
  out["distribution"] = dist
print(json.dumps(out, indent=1))
