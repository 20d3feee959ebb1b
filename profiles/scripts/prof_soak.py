"""Soak: many solves on one handle (and re-created handles) -- every result must repeat bit for bit, nothing may hang.
    python tests/prof_soak.py [n_small] [n_big]"""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_big = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for name, n in (("tiny_rolling", n_small), ("tiny_handeye", n_small // 3), ("cfg2", n_big), ("cfg3", n_big)):
    rig = synthetic.make_rig(name); c = calibration.from_rig(rig); x0 = c.param_vec
    t0 = time.perf_counter()
    ref = None
    for rep in range(3):                      # three handles in a row (the resource cache recycles buffers, streams, pinned words)
        with Handle(c) as h:
            for i in range(max(n // 3, 1)):
                r = h.solve(x0)
                key = (r.nfev, r.status, float(r.cost), r.x.tobytes())
                if ref is None: ref = key
                assert key == ref, (name, rep, i, r.nfev, r.status, r.cost)
    print("%-14s %4d solves identical (nfev %d, status %d, cost %.9e) in %.2f s" % (name, 3 * max(n // 3, 1), ref[0], ref[1], ref[2], time.perf_counter() - t0), flush=True)
# solver = "lsmr": the device-resident LSMR iteration (iterations enqueued ahead of a progress word) must repeat bit for bit too
for name, n in (("tiny_rolling", max(n_small // 10, 3)), ("cfg2", max(n_big // 6, 3)), ("cfg3", max(n_big // 6, 3)), ("cfg5", max(n_big // 6, 3))):
    # (round 6: cfg3 = compacted observation tables, rolling shutter; cfg5 / cfg2 = the cached per-observation state, hand-eye / static)
    rig = synthetic.make_rig(name); c = calibration.from_rig(rig); x0 = c.param_vec
    t0 = time.perf_counter()
    ref = None
    for rep in range(3):
        with Handle(c) as h:
            for i in range(max(n // 3, 1)):
                r = h.solve(x0, tr_solver="lsmr")
                key = (r.nfev, r.status, float(r.cost), r.x.tobytes())
                if ref is None: ref = key
                assert key == ref, (name, "lsmr", rep, i, r.nfev, r.status, r.cost)
    print("%-14s %4d lsmr solves identical (nfev %d, status %d, cost %.9e) in %.2f s" % (name, 3 * max(n // 3, 1), ref[0], ref[1], ref[2], time.perf_counter() - t0), flush=True)
# pose-graph initialisation: staged alignment kernels (rounds enqueued ahead of a progress word), recycled streams / buffers
from multical_amd import tables as mtables
from multical_amd.structs import Table
for name in ("cfg3", "cfg4"):
    rig = synthetic.make_rig(name)
    pt = synthetic.make_pose_table(rig, seed=5)
    tab = Table.create(poses=pt["poses"], valid=pt["valid"], num_points=pt["num_points"])
    t0 = time.perf_counter()
    ref = None
    for i in range(max(n_big // 3, 3)):
        got = mtables.initialise_poses(tab)
        key = tuple(got[k].poses.tobytes() for k in ("camera", "board", "times"))
        if ref is None: ref = key
        assert key == ref, (name, "initialise_poses", i)
    print("%-14s %4d initialisations identical in %.2f s" % (name, max(n_big // 3, 3), time.perf_counter() - t0), flush=True)
