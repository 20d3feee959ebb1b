"""Diagnostic sweep on a real GPU (not a pytest file): prints parity numbers for every stage; never stops early."""
import sys, os, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle, mfma_probe
from hostmath_lib import HostMath
from util import load_golden, mirror, oracle, golden_jacobian, rel_col_error, ALL_CASES, SMALL_CASES
from oracle import restate


def section(name):
  print(f"\n=== {name} ===", flush=True)


def guarded(fn, *a):
  try:
    fn(*a)
  except Exception:
    traceback.print_exc()
    print("!! FAILED", fn.__name__, a, flush=True)


def probe():
  section("mfma probe")
  rng = np.random.default_rng(0)
  V = rng.normal(size=(4, 32))
  out = mfma_probe(V)
  ref = V[:, :16].T @ V[:, 16:]
  print("A^T B maxdiff", np.abs(out - ref).max(), " (transposed?)", np.abs(out - ref.T).max())


def case(name):
  section(f"case {name}")
  g, rig = load_golden(name)
  c = mirror(rig)
  hm = HostMath(c)
  x0 = g["x0"]
  with Handle(c) as h:
    print(h.device_info(), "n", h.n_params, "m", h.n_residuals)
    r = h.residuals(x0)
    print("residual vs reference golden:", np.abs(r - g["r0"]).max(), " vs hostmath:", np.abs(r - hm.residuals(x0)).max())
    err, valid = h.reprojection_error(x0)
    print("reproj err vs golden:", np.abs(err[valid] - g["err0"]).max(), valid.sum(), g["err0"].size)
    proj = h.project(x0)
    print("project finite:", np.isfinite(proj).all())
    J = h.jacobian(x0)
    Jh = hm.jacobian(x0)
    print("jacobian vs hostmath:", np.abs((J - Jh)).max(), "scale", np.abs(Jh).max())
    if "J_data" in g:
      print("jacobian vs reference FD (rel col):", rel_col_error(J, golden_jacobian(g)))
    Hh, gh_, costh = hm.normal_equations(x0)
    for mf in (0, 1):
      h.set_mfma(mf)
      cost, grad, diag = h.normal_equations(x0)
      H = h.dense_hessian()
      print(f"normal eq (mfma={mf}): cost rel", abs(cost - costh) / costh, "g rel", np.abs(grad - gh_).max() / np.abs(gh_).max(),
            "diag rel", np.abs(diag - np.diag(Hh)).max() / np.abs(np.diag(Hh)).max(),
            "H rel", np.abs(H - Hh).max() / np.abs(Hh).max(), "H asym", np.abs(H - H.T).max() / np.abs(H).max())
    # regularised GN step
    for reg in (1e-3, 1e-8):
      gn, ghs, si = h.debug_gn_step(reg)
      si_ref = np.sqrt(np.diag(Hh)); si_ref[si_ref == 0] = 1
      d = 1 / si_ref
      Hs = Hh * d[:, None] * d[None, :]
      ref = np.linalg.solve(Hs + reg * np.eye(h.n_params), d * gh_)
      print(f"gn step reg={reg}: rel err", np.abs(gn - ref).max() / np.abs(ref).max(), "scale_inv rel", np.abs(si - si_ref).max() / si_ref.max())
    # solve
    for mf in (0, 1):
      h.set_mfma(mf)
      rows = []
      h.set_log(lambda *a: rows.append(a))
      t = time.time()
      res = h.solve(x0)
      dt = time.time() - t
      cres = c.with_param_vec(res.x)
      e, v = h.reprojection_error(res.x)
      rms = np.sqrt(np.mean(e[v] ** 2))
      print(f"solve (mfma={mf}): nfev {res.nfev} njev {res.njev} status {res.status} cost {res.cost:.10e} (ref {float(g['ba_cost']):.10e}, ref nfev {int(g['ba_nfev'])}) "
            f"rms {rms:.10f} ref {float(g['ba_rms']):.10f} diff {rms - float(g['ba_rms']):.3e} time {dt*1e3:.1f} ms solve_s {res.solve_seconds*1e3:.2f} ms lin {res.linearize_seconds*1e3:.3f} ms")
      for row in rows:
        print("     ", row[0], row[1], f"{row[2]:.6e}", f"{row[3]:.3e}", f"{row[4]:.3e}", f"{row[5]:.3e}")
    # tight solve vs tight reference
    res = h.solve(x0, tolerance=1e-12, xtol=1e-12, gtol=1e-12, max_iterations=200)
    e, v = h.reprojection_error(res.x)
    print(f"tight solve: nfev {res.nfev} status {res.status} cost {res.cost:.12e} rms {np.sqrt(np.mean(e[v]**2)):.12f}")


def outliers(name):
  section(f"adjust_outliers {name}")
  g, rig = load_golden(name)
  c = mirror(rig)
  t = time.time()
  ao = c.adjust_outliers(num_adjustments=3, select_outliers=calibration.select_threshold(0.75, 5.0), loss='linear', tolerance=1e-4)
  dt = time.time() - t
  rms = calibration.error_stats(ao.reprojection_error).rms
  rmsi = calibration.error_stats(ao.reprojection_inliers).rms
  print(f"rms {rms:.10f} ref {float(g['ao_rms']):.10f} diff {rms-float(g['ao_rms']):.3e} | tight ref {float(g['ao_tight_rms']):.10f} diff {rms-float(g['ao_tight_rms']):.3e}")
  print(f"rms inliers {rmsi:.10f} ref {float(g['ao_rms_inliers']):.10f} diff {rmsi-float(g['ao_rms_inliers']):.3e}; inlier masks equal: {np.array_equal(ao.inliers, g['ao_inliers'])} ({(ao.inliers != g['ao_inliers']).sum()} differ) time {dt:.3f}s")


def timing(name, frames=None):
  section(f"timing {name} frames={frames}")
  rig = synthetic.make_rig(name, frames=frames)
  c = mirror(rig)
  x0 = c.param_vec
  t = time.time()
  h = Handle(c)
  print("create", time.time() - t, "s; n", h.n_params, "m", h.n_residuals, "slots", np.prod(h.shape))
  for mf in (0, 1):
    h.set_mfma(mf)
    print(f"linearize mfma={mf}: {h.time_linearize(x0, 10):.4f} ms")
  print(f"residuals: {h.time_residuals(x0, 10):.4f} ms")
  t = time.time()
  res = h.solve(x0)
  print(f"solve: {time.time()-t:.4f}s nfev {res.nfev} njev {res.njev} iters {res.iterations} status {res.status} cost {res.cost:.6e} lin_s {res.linearize_seconds:.5f}")
  e, v = h.reprojection_error(res.x)
  print("rms", np.sqrt(np.mean(e[v] ** 2)))
  h.close()


if __name__ == "__main__":
  guarded(probe)
  names = sys.argv[1:] or ["tiny", "tiny_rolling", "tiny_fisheye", "tiny_handeye", "tiny_rational", "tiny_thin_prism", "tiny_tilted", "tiny_edge", "tiny_fixintr", "cfg1"]
  for n in names:
    guarded(case, n)
  for n in ["tiny", "tiny_rolling", "tiny_handeye", "cfg1"]:
    guarded(outliers, n)
  guarded(timing, "cfg2")
  guarded(timing, "cfg3", 100)
  guarded(timing, "cfg3")
