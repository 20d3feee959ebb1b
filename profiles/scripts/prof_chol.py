"""Device Cholesky paths at a few sizes: ms per call including the two small copies (mcba_debug_chol)."""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
c = calibration.from_rig(synthetic.make_rig("tiny"))
rng = np.random.default_rng(0)
with Handle(c) as h:
    for ns in (70, 140, 159, 200, 286, 400, 963):
        M = rng.normal(size=(ns + 20, ns)); S = M.T @ M / ns + 0.1 * np.eye(ns); rhs = rng.normal(size=ns)
        ref = np.linalg.solve(S + 0.05 * np.eye(ns), rhs)
        out = []
        for mode in (0, 1, 6):
            try:
                p = h.debug_chol(S, rhs, reg=0.05, blocked=mode)
                t0 = time.perf_counter()
                for _ in range(10): h.debug_chol(S, rhs, reg=0.05, blocked=mode)
                out.append("mode %d: %.3f ms (err %.1e)" % (mode, (time.perf_counter() - t0) / 10 * 1e3, np.abs(p - ref).max() / np.abs(ref).max()))
            except Exception as e:
                out.append("mode %d: %s" % (mode, str(e)[:40]))
        print("ns", ns, " | ".join(out), flush=True)
