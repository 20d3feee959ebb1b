"""Initialisation tables: multical_amd.tables.initialise_poses on the device vs the oracle restatement
(= the reference, numpy / scipy on one host core) on synthetic pose tables of the BASELINE shapes."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from multical_amd import synthetic, tables as mtables
from multical_amd.structs import Table
from oracle import restate_init
for name in sys.argv[1:] or ["cfg2", "cfg3", "cfg4"]:
  rig = synthetic.make_rig(name)
  pt = synthetic.make_pose_table(rig, seed=5)
  tab = Table.create(poses=pt["poses"], valid=pt["valid"], num_points=pt["num_points"])
  mtables.initialise_poses(tab)
  t0 = time.perf_counter(); got = mtables.initialise_poses(tab); t_dev = time.perf_counter() - t0
  t0 = time.perf_counter(); want = restate_init.initialise_poses(restate_init.table(pt["poses"], pt["valid"]), pt["num_points"]); t_ref = time.perf_counter() - t0
  err = max(np.abs(got[k].poses - want[k]["poses"]).max() for k in ("camera", "board", "times"))
  print(f"{name}: pose table {pt['valid'].shape}, device {t_dev * 1e3:.1f} ms, reference (oracle, 1 core) {t_ref * 1e3:.0f} ms, "
        f"max |pose difference| {err:.1e}", flush=True)
