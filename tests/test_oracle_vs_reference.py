"""CPU, build box only: pin the oracle against the UNMODIFIED reference executed in-container (oracle/refload.py)."""
import numpy as np
import pytest

from multical_amd import synthetic
from oracle import restate

pytestmark = pytest.mark.needs_reference


@pytest.mark.parametrize("name", ["tiny", "tiny_rolling", "tiny_fisheye", "tiny_handeye", "tiny_tilted"])
def test_restatement_is_bit_identical_to_reference(name):
  from oracle import build_reference
  rig = synthetic.make_rig(name)
  ref_calib, ref = build_reference.reference_calibration(rig)
  oc = restate.from_rig(rig)
  x = ref_calib.param_vec
  assert np.array_equal(x, oc.param_vec)

  def evaluate(v):
    c = ref_calib.with_param_vec(v)
    return (c.reprojected.points - c.point_table.points)[ref_calib.inliers].ravel()

  rng = np.random.default_rng(0)
  for _ in range(2):
    assert np.array_equal(evaluate(x), oc.evaluate(x))
    x = x + rng.normal(0, 1e-3, x.shape)
  assert (ref_calib.sparsity_matrix.tocsr() != oc.sparsity_matrix.tocsr()).nnz == 0
  assert np.array_equal(ref_calib.reprojection_error, oc.reprojection_error)


def test_reference_bundle_adjust_equals_oracle():
  from oracle import build_reference
  rig = synthetic.make_rig("tiny_fisheye")
  ref_calib, ref = build_reference.reference_calibration(rig)
  a = ref_calib.bundle_adjust()
  b = restate.from_rig(rig).bundle_adjust()
  assert np.array_equal(a.param_vec, b.param_vec)


def test_mirror_lowering_accepts_reference_objects():
  """multical_amd.backend.lower() is duck-typed: the reference's own Calibration lowers to the same flat problem."""
  from oracle import build_reference
  from multical_amd import calibration
  from multical_amd.backend import lower
  for name in ["tiny_rolling", "tiny_handeye", "tiny_fisheye"]:
    rig = synthetic.make_rig(name)
    ref_calib, _ = build_reference.reference_calibration(rig)
    pa, pb = lower(ref_calib), lower(calibration.from_rig(rig))
    for k in ["points", "point_valid", "board_sizes", "camera_valid", "frame_valid", "board_valid", "x_full",
              "image_heights", "fix_aspect"]:
      assert np.array_equal(getattr(pa, k), getattr(pb, k)), k
    assert (pa.motion, pa.camera_model, pa.n_dist, pa.optimize, pa.n_params) == \
           (pb.motion, pb.camera_model, pb.n_dist, pb.optimize, pb.n_params)


@pytest.mark.parametrize("name,master", [("cfg1", None), ("cfg1", "cam1"), ("tiny_rolling", "cam0"), ("tiny_handeye", None)])
def test_export_json_equals_reference_export(name, master):
  """SURVEY 8(f) result formats: multical_amd.export.export_json on the mirror Calibration == the reference's
  export_json (io/export_calib.py:81-97) on the reference Calibration of the same rig, with and without a master
  camera (with_master / transform_views, calibration.py:99-112).  Lists of floats are compared exactly."""
  import json
  from types import SimpleNamespace
  from oracle import build_reference
  from multical_amd import calibration as mcal, export as mexport
  rig = synthetic.make_rig(name)
  ref_calib, ref = build_reference.reference_calibration(rig)
  from multical.io.export_calib import export_json as ref_export_json
  from structs.struct import to_dicts
  C = rig.valid.shape[0]
  names = SimpleNamespace(camera=[f"cam{i}" for i in range(C)])
  filenames = [[f"cam{c}/img{f}.png" for f in range(3)] for c in range(C)]
  want = to_dicts(ref_export_json(ref_calib, names, filenames, master=master))
  mine = mcal.from_rig(rig)
  mine.camera_poses.names = list(names.camera)
  got = mexport.export_json(mine, names, filenames, master=master)
  assert json.loads(json.dumps(got)) == json.loads(json.dumps(want))


@pytest.mark.parametrize("name", ["tiny_rolling", "tiny", "tiny_handeye"])
def test_projected_restatement_is_bit_identical_to_reference(name):
  """Calibration.projected (calibration.py:113-119; rolling shutter: the t = 0.5 start + max_iterations fixed-point
  passes of motion/rolling_frames.py:125-133): the oracle's restatement against the unmodified reference."""
  from oracle import build_reference
  rig = synthetic.make_rig(name)
  ref_calib, ref = build_reference.reference_calibration(rig)
  want = ref_calib.projected
  got, valid = restate.from_rig(rig).projected()
  assert np.array_equal(np.asarray(want.valid), valid)
  assert np.array_equal(np.asarray(want.points)[valid], got[valid])


@pytest.mark.parametrize("name,frames", [("tiny_rolling", None), ("tiny_fisheye", None), ("cfg4", 12), ("cfg2", 30),
                                         ("cfg5", 30)])
def test_initialise_poses_restatement_is_bit_identical_to_reference(name, frames):
  """tables.initialise_poses (tables.py:353-377) -- relative camera / board poses through the overlap
  spanning tree and the per-frame rig poses, all through matrix.align_transforms_robust / the Ward-cluster robust mean --
  restated in oracle/restate_init.py, against the unmodified reference on a synthetic pose table with outliers."""
  from oracle import refload, restate_init
  refload.load()
  if not hasattr(np, "bool"):
    np.bool = bool                     # the reference still spells it np.bool (tables.py:363, transform/matrix.py:145)
  from multical import tables as ref_tables
  from structs.numpy import Table
  rig = synthetic.make_rig(name, frames=frames)
  pt = synthetic.make_pose_table(rig, seed=5)
  ref_in = Table.create(poses=pt["poses"].copy(), valid=pt["valid"].copy(), num_points=pt["num_points"].copy())
  want = ref_tables.initialise_poses(ref_in)
  got = restate_init.initialise_poses(restate_init.table(pt["poses"], pt["valid"]), pt["num_points"])
  for k, rk in [("camera", want.camera), ("board", want.board), ("times", want.times)]:
    assert np.array_equal(np.asarray(rk.valid), got[k]["valid"]), k
    assert np.array_equal(np.asarray(rk.poses), got[k]["poses"]), k


@pytest.mark.parametrize("empty_dtype", [np.float32, np.float64])
def test_make_point_table_equals_the_reference(empty_dtype):
  """tables.make_point_table (tables.py:12-20,68-81) on ragged float32 detections with empty images: same points, same mask and
  the SAME dtype as the reference's fill_sparse + Table.stack -- float32 when every image is float32, float64 as soon as one
  (e.g. an empty detection built as float64 zeros) is not."""
  from oracle import refload
  from multical_amd import tables as mtables
  refload.load()
  import multical.tables as rtables
  from structs.struct import struct
  rng = np.random.default_rng(11)
  C, F, B = 3, 5, 2
  sizes = [81, 324]
  boards = [struct(num_points=n) for n in sizes]
  dets = []
  for c in range(C):
    cam = []
    for f in range(F):
      frame = []
      for b in range(B):
        k = 0 if (c + f + b) % 4 == 0 else int(rng.integers(1, sizes[b]))
        ids = np.sort(rng.choice(sizes[b], size=k, replace=False)).astype(np.int32)
        if k == 0:
          corners = np.zeros([0, 2], dtype=empty_dtype)
        else:
          corners = rng.uniform(0, 2000, size=(k, 2)).astype(np.float32)
        frame.append(struct(corners=corners, ids=ids))
      cam.append(frame)
    dets.append(cam)
  ref = rtables.make_point_table(dets, boards)
  got = mtables.make_point_table(dets, boards)
  assert got.points.dtype == ref.points.dtype == (np.float32 if empty_dtype == np.float32 else np.float64)
  assert got.points.shape == ref.points.shape == (C, F, B, 324, 2)
  assert np.array_equal(got.points, ref.points) and np.array_equal(got.valid, ref.valid)
  assert got.valid.dtype == ref.valid.dtype == bool
  # the float32 table goes to the device as it is and is widened there exactly as numpy widens it (mcba_problem.points_f32)
  assert np.array_equal(got.points.astype(np.float64), np.asarray(ref.points, dtype=np.float64))


def test_exact_product_end_points_regenerate_here():
  """tests/golden/exact_products.json, regenerated on the spot for 6 x 40 x 5 (hand-eye, fisheye): scipy's own trf + lsmr on the REFERENCE's
  residual function, once with scipy.sparse's double products and once with the products accumulated in 80-bit precision
  (oracle/make_exact_products.py).  The committed clusters come back, and the 80-bit run ends below the double runs -- the footprint of the reference's own product rounding (profiles/r06_lsmr_sign.md)."""
  import json
  import os
  import sys
  here = os.path.dirname(os.path.abspath(__file__))
  sys.path.insert(0, here)
  from multical_amd import synthetic, calibration as mirror_calibration
  from hostmath_lib import HostMath
  from lsmr_emulation import trf_lsmr
  from oracle import build_reference
  from oracle.make_golden import _evaluate
  from oracle.make_exact_products import longdouble_solver
  xp = json.load(open(os.path.join(here, "golden", "exact_products.json")))["cfg5_40"]
  g = dict(np.load(os.path.join(here, "golden", "cfg5_40.npz"), allow_pickle=False))
  rig = synthetic.make_rig(str(g["config"]))
  calib, ref = build_reference.reference_calibration(rig)
  error_stats = ref.optimization_calibration.error_stats
  hm = HostMath(mirror_calibration.from_rig(rig))
  fun = lambda x: _evaluate(calib, x)
  out = {}
  for kind, solver in (("double", "scipy"), ("longdouble", longdouble_solver)):
    res = trf_lsmr(fun, hm.jacobian, g["x0"], solver=solver)
    out[kind] = float(error_stats(calib.with_param_vec(res["x"]).reprojection_error).rms)
    want = [r for r in xp["runs"] if r["arithmetic"] == kind and r["row_order_seed"] == 0][0]
    assert res["nfev"] == want["nfev"] == int(g["ba_nfev"])
    print(f"{kind}: rms - reference {out[kind] - float(g['ba_rms']):+.3e} (committed run: {want['rms_minus_reference']:+.3e})")
    # (the committed value comes back bit for bit on this container; a BLAS that splits its dot products differently moves a run inside
    #  its cluster: the assertion is the cluster, not the bits)
    cluster = [r["rms"] for r in xp["runs"] if r["arithmetic"] == kind]
    assert min(cluster) - 5e-7 <= out[kind] <= max(cluster) + 5e-7, (kind, out[kind] - float(g["ba_rms"]))
  dbl = [r["rms"] for r in xp["runs"] if r["arithmetic"] == "double"]
  assert out["longdouble"] < out["double"] - 2e-7 and out["longdouble"] < min(dbl) - 2e-7, (out, dbl)
  assert abs(out["double"] - float(g["ba_rms"])) <= 1e-6          # scipy's arithmetic on the analytic Jacobian = the reference's end point
