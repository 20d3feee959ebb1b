"""Per-kernel timing of one full solve at the north-star rig (run under rocprofv3 --kernel-trace --stats)."""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
rig = synthetic.make_rig("cfg3"); c = calibration.from_rig(rig); x0 = c.param_vec
with Handle(c) as h:
    h.solve(x0)
    t0 = time.perf_counter(); res = h.solve(x0); dt = time.perf_counter() - t0
    print("solve s", dt, "nfev", res.nfev, "njev", res.njev, "status", res.status, "cost", res.cost, "iters/s", (res.nfev - 1) / dt)
    print("linearize ms", h.time_linearize(x0, 50))
    rng = np.random.default_rng(1)
    x1 = x0 + 1e-3 * rng.normal(size=x0.size)
    h.solve(x1)
    t0 = time.perf_counter(); res = h.solve(x1); dt = time.perf_counter() - t0
    print("perturbed solve s", dt, "nfev", res.nfev, "njev", res.njev, "status", res.status, "cost", res.cost,
          "ms/iter", dt / max(res.nfev - 1, 1) * 1e3)
    rng = np.random.default_rng(0)
    for _ in range(3):
        t0 = time.perf_counter(); res = h.solve(x1); dt = time.perf_counter() - t0
        print("  repeat: ms/iter", dt / max(res.nfev - 1, 1) * 1e3, "total ms", dt * 1e3)
    res = h.solve(x1, tolerance=1e-15, xtol=1e-15, gtol=1e-15, max_iterations=41)
    t0 = time.perf_counter(); res = h.solve(x1, tolerance=1e-15, xtol=1e-15, gtol=1e-15, max_iterations=41); dt = time.perf_counter() - t0
    print("long solve: nfev", res.nfev, "njev", res.njev, "status", res.status, "ms per trial step", dt / max(res.nfev - 1, 1) * 1e3)
    for ns in (140,):
        M = rng.normal(size=(ns + 20, ns)); S = M.T @ M / ns + 0.1 * np.eye(ns); rhs = rng.normal(size=ns)
        st = h.debug_chol(S, rhs, reg=0.05, blocked=4)[:8]
        print("k_chol_blk cycles: load(barrier) %d  first diag %d  panel %d  syrk + next diag %d  back %d | load batch 1 %d  batch 2 %d" % tuple(st[:7]))
        for mode in (0, 3):
            h.debug_chol(S, rhs, reg=0.05, blocked=mode)
            t0 = time.perf_counter()
            for _ in range(20): h.debug_chol(S, rhs, reg=0.05, blocked=mode)
            print("chol mode", mode, "ms/call incl. copies", (time.perf_counter() - t0) / 20 * 1e3)
