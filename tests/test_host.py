"""CPU: host logic (parameter packing, lowering, C-ABI surface) and the product's device math compiled for the host
(tests/hostmath) against the golden fixtures of the real reference."""
import ctypes
import os
import re

import numpy as np
import pytest

from multical_amd import _lib, synthetic, calibration
from multical_amd.backend import lower
from hostmath_lib import HostMath
from util import SMALL_CASES, ALL_CASES, load_golden, mirror, oracle, golden_jacobian, rel_col_error

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
  header = open(os.path.join(ROOT, "include", "mcba.h")).read()
  declared = set(re.findall(r"\b(mcba_[a-z_0-9]+)\s*\(", header))
  declared -= {"mcba_full_size()"}
  lib = _lib.load()
  bound = {name for name, _, _ in _lib.SYMBOLS}
  missing = [s for s in declared if not hasattr(lib, s)]
  assert not missing, f"libmcba.so lacks {missing}"
  assert declared <= bound, f"ctypes table lacks {declared - bound}"


def test_library_exports_every_debug_symbol():
  """csrc/mcba_debug.h (test / profiling entry points, not part of the drop-in boundary): every declared function is exported and
  bound with a ctypes signature, so that a parity hook cannot silently go missing from the library the GPU tests load."""
  header = open(os.path.join(ROOT, "multical_amd", "csrc", "mcba_debug.h")).read()
  declared = set(re.findall(r"\bint32_t\s+(mcba_[a-z_0-9]+)\s*\(", header))
  assert len(declared) >= 20
  lib = _lib.load()
  bound = {name for name, _, _ in _lib.SYMBOLS}
  missing = [s for s in declared if not hasattr(lib, s)]
  assert not missing, f"libmcba.so lacks {missing}"
  assert declared <= bound, f"ctypes table lacks {declared - bound}"


def test_create_fails_loudly_without_gpu():
  from util import gpu_available
  if gpu_available():
    pytest.skip("GPU present")
  from multical_amd.backend import Handle
  with pytest.raises(RuntimeError, match="no HIP device|GPU-only|hip"):
    Handle(mirror(synthetic.make_rig("tiny")))


@pytest.mark.parametrize("name", ALL_CASES)
def test_parameter_packing_matches_reference(name):
  g, rig = load_golden(name)
  c = mirror(rig)
  assert np.array_equal(c.param_vec, g["x0"])
  assert np.array_equal(c.inliers, g["inliers0"])
  # with_param_vec(param_vec) round trip and the canonicalised read-back of a solved vector (rtvec.py:29-32)
  assert np.array_equal(c.with_param_vec(g["ba_x_raw"]).param_vec, g["ba_x"])


def test_lowering_layout():
  rig = synthetic.make_rig("tiny_rolling")
  c = mirror(rig)
  p = lower(c)
  C_, F, B, P = p.shape
  assert p.x_full.size == 6 * C_ + 6 * B + 12 * F + C_ * (5 + 5) + 3 * sum(p.board_sizes)
  assert p.n_params == c.param_vec.size
  n = ctypes.c_int64()
  from multical_amd.backend import _to_struct
  assert _lib.load().mcba_full_size(ctypes.byref(_to_struct(p)), ctypes.byref(n)) == 0
  assert n.value == p.x_full.size
  # enable flags follow `optimize[k] is True` (calibration.py:160)
  assert lower(c.enable(cameras=False)).n_params == p.n_params - C_ * 10
  # a float32 point table (the reference's dtype: tables.py:15-17) is handed over as it is, a float64 one as float64
  st = _to_struct(p)
  assert p.points is not None and p.points_f32 is None and bool(st.points) and not bool(st.points_f32)
  rig.points = rig.points.astype(np.float32)
  p32 = lower(mirror(rig))
  st32 = _to_struct(p32)
  assert p32.points is None and p32.points_f32.dtype == np.float32 and p32.points_f32.flags.c_contiguous
  assert not bool(st32.points) and bool(st32.points_f32)
  assert _lib.load().mcba_full_size(ctypes.byref(st32), ctypes.byref(n)) == 0 and n.value == p.x_full.size


def test_pickle_holds_only_constructor_fields():
  import pickle
  c = mirror(synthetic.make_rig("tiny"))
  c2 = pickle.loads(pickle.dumps(c))
  assert sorted(c2.__dict__) == sorted(['cameras', 'boards', 'point_table', 'camera_poses', 'board_poses', 'motion',
                                        'inlier_mask', 'optimize'])
  assert np.array_equal(c2.param_vec, c.param_vec)


@pytest.mark.parametrize("name", ALL_CASES)
def test_device_math_residuals_match_reference(name):
  """mcba_math.h / mcba_view.h (built for the host) reproduce evaluate() of the reference to 1e-9 px."""
  g, rig = load_golden(name)
  hm = HostMath(mirror(rig))
  r = hm.residuals(g["x0"])
  assert r.shape == g["r0"].shape
  assert np.abs(r - g["r0"]).max() < 1e-9
  err, valid = hm.reprojection_error(g["x0"])
  assert np.abs(err[valid] - g["err0"]).max() < 1e-9
  # ... and at the reference's solution
  c = mirror(rig).with_param_vec(g["ba_x_raw"])
  oc = oracle(rig)
  assert np.abs(HostMath(c).residuals(g["ba_x_raw"]) - oc.evaluate(g["ba_x_raw"])).max() < 1e-9


@pytest.mark.parametrize("name", SMALL_CASES)
def test_device_math_jacobian_matches_reference_fd(name):
  g, rig = load_golden(name)
  hm = HostMath(mirror(rig))
  J = hm.jacobian(g["x0"])
  Jfd = golden_jacobian(g)
  assert J.shape == Jfd.shape
  assert rel_col_error(J, Jfd) < 5e-5          # forward differences with h = sqrt(eps)|x| are accurate to ~1e-6
  # analytic non-zeros lie inside the reference's sparsity pattern
  S = oracle(rig).sparsity_matrix.tocsr()
  assert (abs(J) > 0).multiply(S == 0).nnz == 0


def test_device_math_jacobian_against_central_differences():
  """tighter check of the analytic derivatives: 3-point differences of the oracle (error ~1e-9)."""
  from scipy.optimize._numdiff import approx_derivative, group_columns
  from scipy.sparse import csr_matrix
  for name in ["tiny_rolling", "tiny_fisheye", "tiny_tilted", "tiny_handeye"]:
    g, rig = load_golden(name)
    oc = oracle(rig)
    S = csr_matrix(oc.sparsity_matrix)
    J3 = csr_matrix(approx_derivative(oc.evaluate, g["x0"], method='3-point', sparsity=(S, group_columns(S))))
    J = HostMath(mirror(rig)).jacobian(g["x0"])
    assert rel_col_error(J, J3) < 2e-7, name


@pytest.mark.parametrize("name", ["tiny", "tiny_rolling", "tiny_handeye", "tiny_edge", "tiny_fixintr"])
def test_device_math_normal_equations(name):
  """per-view S = V^T V, M = That^T S That scattered through local_to_x == J^T J, J^T f."""
  g, rig = load_golden(name)
  hm = HostMath(mirror(rig))
  J = hm.jacobian(g["x0"]).toarray()
  r = hm.residuals(g["x0"])
  H, grad, cost = hm.normal_equations(g["x0"])
  assert np.abs(H - J.T @ J).max() <= 1e-12 * np.abs(H).max()
  assert np.abs(grad - J.T @ r).max() <= 1e-12 * np.abs(grad).max()
  assert cost == pytest.approx(0.5 * r @ r, rel=1e-13)


@pytest.mark.parametrize("loss,f_scale", [("soft_l1", 1.5), ("huber", 2.0), ("cauchy", 1.0), ("arctan", 3.0)])
def test_device_math_robust_loss_scaling(loss, f_scale):
  """row / residual scaling == scipy's scale_for_robust_loss_function (common.py:720-731)."""
  from scipy.optimize._lsq.least_squares import construct_loss_function
  from scipy.optimize._lsq.common import scale_for_robust_loss_function
  g, rig = load_golden("tiny")
  hm = HostMath(mirror(rig))
  J = hm.jacobian(g["x0"]).toarray()
  f = hm.residuals(g["x0"])
  lf = construct_loss_function(f.size, loss, f_scale)
  rho = lf(f)
  cost_ref = 0.5 * np.sum(rho[0])
  Js, fs = scale_for_robust_loss_function(J.copy(), f.copy(), rho)
  H, grad, cost = hm.normal_equations(g["x0"], loss=loss, f_scale=f_scale)
  assert cost == pytest.approx(cost_ref, rel=1e-12)
  assert np.abs(H - Js.T @ Js).max() <= 1e-11 * np.abs(H).max()
  assert np.abs(grad - Js.T @ fs).max() <= 1e-11 * np.abs(grad).max()


def test_pose_table_outlier_pose_rejection():
  """BASELINE configs[4] "outlier pose rejection": tables.extract_pose (tables.py:44-56) drops every view whose PnP pose
  reprojects its detections worse than pose_error_limit before the initialisation tables are built.  The synthetic pose
  table of cfg5 models it: rejected views are invalid_pose entries (identity, no points, tables.py:38), the surviving
  ones meet the limit, and the wrong-orientation poses are among the rejected."""
  from multical_amd import synthetic
  rig = synthetic.make_rig("cfg5", frames=20)
  assert rig.cfg["pose_error_limit"] == 1.0                      # the reference's default (config/arguments.py:50)
  pt = synthetic.make_pose_table(rig, seed=3)
  free = synthetic.make_pose_table(rig, seed=3, pose_error_limit=np.inf)
  detected = rig.valid.sum(axis=3) > 0
  assert np.array_equal(free["valid"], detected) and not free["rejected"].any()
  rej = pt["rejected"]
  assert np.array_equal(pt["valid"], detected & ~rej) and 0.2 < rej.sum() / detected.sum() < 0.8
  assert np.array_equal(pt["poses"][~pt["valid"]], np.broadcast_to(np.eye(4), pt["poses"][~pt["valid"]].shape))
  assert (pt["num_points"][~pt["valid"]] == 0).all() and (pt["num_points"][pt["valid"]] >= 12).all()
  err = synthetic.view_pose_errors(rig, free["poses"])
  assert (err[pt["valid"]] <= 1.0).all() and (err[rej] > 1.0).all()
  assert np.array_equal(pt["poses"][pt["valid"]], free["poses"][pt["valid"]])
  # a rig without the key keeps every detected view (the other BASELINE configurations)
  rig2 = synthetic.make_rig("cfg2", frames=5)
  assert not synthetic.make_pose_table(rig2, seed=3)["rejected"].any()


def test_parameter_deltas_are_gauge_invariant():
  """multical_amd.gauge: the 12-dimensional gauge freedom of the bundle adjustment (camera -> camera T^-1, rig -> T rig; rig ->
  rig S^-1, board -> S board; calibration.py:99-112 removes the first at export time) does not show up in the physical
  differences, a real change does."""
  from scipy.spatial.transform import Rotation
  from multical_amd import gauge
  for name in ("tiny", "tiny_rolling"):
    rig = synthetic.make_rig(name)
    c = calibration.from_rig(rig)

    def rigid(seed):
      r = np.random.default_rng(seed)
      T = np.eye(4)
      T[:3, :3] = Rotation.from_rotvec(r.normal(0, 0.4, 3)).as_matrix()
      T[:3, 3] = r.normal(0, 0.3, 3)
      return T

    T, S = rigid(1), rigid(2)
    moved = c.copy(camera_poses=c.camera_poses.post_transform(np.linalg.inv(T)),
                   motion=c.motion.pre_transform(T).post_transform(np.linalg.inv(S)),
                   board_poses=c.board_poses.pre_transform(S))
    d = gauge.parameter_deltas(c, moved)
    assert max(d.camera_deg, d.frame_deg, d.board_deg) < 1e-10 and max(d.camera_t, d.frame_t, d.board_t) < 1e-12
    assert d.focal_rel == 0 and d.principal_px == 0 and d.dist_abs == 0
    # a real difference: camera 1 rotated by 0.01 degrees, its focal length 0.1 % longer
    cams = c.camera_poses.poses.copy()
    R = np.eye(4)
    R[:3, :3] = Rotation.from_rotvec([0, 0, np.radians(0.01)]).as_matrix()
    cams[1] = R @ cams[1]
    cam1 = c.cameras[1]
    K = cam1.intrinsic.copy()
    K[0, 0] *= 1.001
    from multical_amd.parameters import ParamList
    cameras = ParamList([cam1.copy(intrinsic=K) if i == 1 else cam for i, cam in enumerate(c.cameras)], c.cameras.names)
    changed = moved.copy(camera_poses=moved.camera_poses.copy(pose_table=moved.camera_poses.pose_table._extend(poses=cams @ np.linalg.inv(T))),
                         cameras=cameras)
    d = gauge.parameter_deltas(c, changed)
    assert abs(d.camera_deg - 0.01) < 1e-9 and abs(d.focal_rel - 1e-3 / 1.001) < 1e-12


@pytest.mark.parametrize("name,frames,seed", [("tiny_rolling", None, 5), ("cfg4", 12, 5), ("cfg5", 20, 9)])
def test_initialise_poses_host_logic_with_the_oracle_as_the_device(name, frames, seed):
  """multical_amd.tables.initialise_poses with both device entry points replaced by the oracle's align_transforms_robust (per
  problem, on the CPU): what is tested is everything AROUND the kernels -- overlaps, spanning tree, the pair lists as indices into
  the pose table, `invert` instead of an inverted table for the board stage, the per-frame gather of the entries that take part,
  chaining and the final inverses -- against the oracle's restatement of tables.initialise_poses (bit-identical to the reference:
  test_oracle_vs_reference)."""
  from multical_amd import tables as mtables
  from multical_amd.structs import Table
  from oracle import restate_init

  def ragged(A, B, sizes, mask=None, threshold=1.5, invert=False):
    sizes = np.asarray(sizes, dtype=np.int64).reshape(-1)
    A, B = np.asarray(A, dtype=np.float64).reshape(-1, 4, 4), np.asarray(B, dtype=np.float64).reshape(-1, 4, 4)
    off = np.concatenate([[0], np.cumsum(sizes)])
    m = np.ones(int(off[-1]), dtype=bool) if mask is None else np.asarray(mask).astype(bool).reshape(-1)
    out = np.broadcast_to(np.eye(4), (sizes.size, 4, 4)).copy()
    valid = np.zeros(sizes.size, dtype=bool)
    inl = np.zeros(int(off[-1]), dtype=bool)
    for p in range(sizes.size):
      sl = slice(int(off[p]), int(off[p + 1]))
      if m[sl].any():
        a, b = (np.linalg.inv(A[sl]), np.linalg.inv(B[sl])) if invert else (A[sl], B[sl])    # relative_between_inv: tables.py:334-335
        t, il = restate_init.align_transforms_robust(a, b, valid=m[sl], threshold=threshold)
        out[p], valid[p], inl[sl] = (np.linalg.inv(t) if invert else t), True, il
    return out, valid, inl

  def indexed(table, ia, ib, sizes, mask=None, threshold=1.5, invert=False):
    t = np.asarray(table, dtype=np.float64).reshape(-1, 4, 4)
    return ragged(t[np.asarray(ia).reshape(-1)], t[np.asarray(ib).reshape(-1)], sizes, mask, threshold, invert)

  saved = mtables.align_transforms_robust_ragged, mtables.align_transforms_robust_indexed
  mtables.align_transforms_robust_ragged, mtables.align_transforms_robust_indexed = ragged, indexed
  try:
    rig = synthetic.make_rig(name, frames=frames)
    pt = synthetic.make_pose_table(rig, seed=seed)
    got = mtables.initialise_poses(Table.create(poses=pt["poses"], valid=pt["valid"], num_points=pt["num_points"]))
  finally:
    mtables.align_transforms_robust_ragged, mtables.align_transforms_robust_indexed = saved
  want = restate_init.initialise_poses(restate_init.table(pt["poses"], pt["valid"]), pt["num_points"])
  for k in ("camera", "board", "times"):
    assert np.array_equal(got[k].valid, want[k]["valid"]), k
    assert np.abs(got[k].poses - want[k]["poses"]).max() < 1e-9, (k, np.abs(got[k].poses - want[k]["poses"]).max())


def test_the_default_solver_is_the_one_that_reproduces_the_reference():
  """`dropin.install()`, MULTICAL_BACKEND=hip and the mirror's `bundle_adjust()` select "lsmr" (scipy's TRF
  + LSMR step on the device: the reference's END POINT); the exact-step solver is the opt-in "native" mode."""
  import inspect
  from multical_amd import dropin
  assert os.environ.get("MULTICAL_AMD_SOLVER") is None and calibration.get_solver() == "lsmr"
  assert calibration.SOLVERS[0] == "lsmr" and set(calibration.SOLVERS) == {"lsmr", "native", "scipy"}
  assert inspect.signature(dropin.install).parameters["mode"].default == "lsmr"
  assert dropin.MODES["lsmr"] is dropin.bundle_adjust and dropin.MODES["native"] is dropin.bundle_adjust_native
  sig = lambda f: list(inspect.signature(f).parameters)
  assert sig(dropin.bundle_adjust) == sig(dropin.bundle_adjust_native) == sig(dropin.bundle_adjust_scipy) == \
      ["self", "tolerance", "f_scale", "max_iterations", "loss"]          # calibration.py:199


def test_handle_cache_forgets_the_device_mask_when_a_mutating_call_fails():
  """mcba_reject_outliers / mcba_adjust_outliers rewrite the device inlier mask; when the call raises
  after that, the cache may not keep believing that the device still holds the Calibration's mask."""
  g, rig = load_golden("tiny")
  c = mirror(rig)
  uploads = []

  class FakeHandle(object):
    h = 1
    problem = None

    def set_inliers(self, mask):
      uploads.append(None if mask is None else mask.copy())

    def reject_outliers(self, x, threshold):
      raise RuntimeError("device failure after the mask was rewritten")

    def adjust_outliers(self, *a, **k):
      raise RuntimeError("solver failure after a rejection")

    def set_log(self, fn):
      pass

    def close(self):
      pass

  cache = calibration.handle_cache
  cache.clear()
  prob = lower(c)
  c._mcba_problem = prob
  fake = FakeHandle()
  fake.problem = prob
  real_handle = calibration.Handle
  calibration.Handle = lambda prob_: fake
  try:
    assert c._handle() is fake and uploads == []
    with pytest.raises(RuntimeError, match="mask was rewritten"):
      c.reject_outliers(1.0)
    assert c._handle() is fake and len(uploads) == 1 and uploads[0] is None      # re-synchronised: inlier_mask is None here
    masked = c.copy(inlier_mask=c.valid.copy())
    masked._mcba_problem = lower(masked)
    assert masked._handle() is fake and len(uploads) == 2
    with pytest.raises(RuntimeError, match="after a rejection"):
      masked.adjust_outliers(num_adjustments=1, select_outliers=calibration.select_threshold(0.75, 5.0))
    assert masked._handle() is fake and len(uploads) == 3 and np.array_equal(uploads[2], masked.inlier_mask)
    assert masked._handle() is fake and len(uploads) == 3                      # ... and the token is valid again afterwards
  finally:
    calibration.Handle = real_handle
    cache.entries = []


@pytest.mark.parametrize("name", ["tiny", "tiny_rolling", "tiny_fisheye", "tiny_handeye", "tiny_tilted", "tiny_edge", "tiny_fixintr",
                                  "tiny_pin4", "tiny_boards", "tiny_bigboard"])
def test_device_math_lsmr_products_match_the_jacobian(name):
  """The matrix-free factorisation the lsmr mode's kernels use (J v = E (That v_pose) + K v_K + jp v_point and its adjoint, built from
  point_state / point_row / view_column / board_point_direction / board_point_adjoint / local_to_x) compiled for the host, against the
  Jacobian of the same device functions -- which test_device_math_jacobian_* pin to the reference's finite differences.  The GPU
  test of the kernels themselves is tests/test_gpu_lsmr.py::test_lsmr_products_against_the_jacobian."""
  g, rig = load_golden(name)
  hm = HostMath(mirror(rig))
  rng = np.random.default_rng(3)
  x = g["x0"] + 1e-3 * rng.normal(size=g["x0"].size)
  J = hm.jacobian(x)
  v, u = rng.normal(size=hm.n), rng.normal(size=hm.m)
  jv, jtu = hm.lsmr_products(x, v, u)
  A = abs(J)
  assert np.abs(jv - J @ v).max() <= 1e-12 * (A @ np.abs(v)).max()
  scale = A.T @ np.abs(u)
  assert np.abs(jtu - J.T @ u).max() <= 1e-12 * scale.max()
  assert np.all(jtu[scale == 0] == 0)


# ---------------------------------------------------------------------------------------------------------------------------------
# The default solver's algorithm on the host (tests/lsmr_emulation.py): scipy's trf_no_bounds transcribed, the device's LSMR flow and the
# device's trust-region driver walked through with numpy vectors and the scalar code of csrc/mcba_lsmr.h / csrc/mcba_trmath.h itself
# ---------------------------------------------------------------------------------------------------------------------------------
EMU_CASES = ["cfg1", "tiny_handeye", "tiny_fixintr", "tiny_rolling"]


@pytest.mark.parametrize("name", EMU_CASES)
def test_trf_transcription_equals_scipys_least_squares(name):
  """tests/lsmr_emulation.trf_lsmr(solver="scipy") IS scipy.optimize.least_squares(method='trf', tr_solver='lsmr', x_scale='jac') as the
  reference calls it (calibration.py:209-210) when both get the same residual function and the same sparse analytic Jacobian: the
  instrumented driver the call-level parity tests, profiles/scripts/prof_lsmr_sign.py and oracle/make_exact_products.py rely on."""
  from scipy.optimize import least_squares
  from lsmr_emulation import trf_lsmr
  g, rig = load_golden(name)
  hm = HostMath(mirror(rig))
  ref = least_squares(hm.residuals, g["x0"], jac=hm.jacobian, x_scale='jac', ftol=1e-4, max_nfev=100, method='trf')
  calls = []
  res = trf_lsmr(hm.residuals, hm.jacobian, g["x0"], solver="scipy", calls=calls)
  assert (res["nfev"], res["njev"], res["status"]) == (ref.nfev, ref.njev, ref.status)
  assert np.array_equal(res["x"], ref.x) or np.abs(res["x"] - ref.x).max() <= 1e-13 * np.abs(ref.x).max()
  assert res["cost"] == pytest.approx(ref.cost, rel=1e-14) and res["optimality"] == pytest.approx(ref.optimality, rel=1e-12)
  assert len(calls) == res["njev"] - (0 if res["status"] == 0 else 1) + 1 or len(calls) >= 1


@pytest.mark.parametrize("name", EMU_CASES)
def test_device_lsmr_flow_first_steps_equal_scipys(name):
  """The device's LSMR FLOW (two launches per iteration: u and v kept un-normalised, rotation + vector update of a step in the tail of the
  next product launch, stopping tests one launch later; tests/lsmr_emulation.device_lsmr) with the scalar recurrences of
  csrc/mcba_lsmr.h, against scipy.sparse.linalg.lsmr(maxiter = k): the whole return tuple and the solution to 1e-10 for the first
  Golub-Kahan steps, and the same stopping reason for the complete call."""
  from scipy.sparse.linalg import lsmr
  from lsmr_emulation import device_lsmr, ScaledMatrix, scaled_operator
  g, rig = load_golden(name)
  hm = HostMath(mirror(rig))
  x0 = g["x0"]
  J, f = hm.jacobian(x0), hm.residuals(x0)
  si = np.asarray(J.power(2).sum(axis=0)).ravel() ** 0.5
  si[si == 0] = 1
  d = 1 / si
  for damp in (0.0, 0.02):
    for k in (1, 2, 3, 5) + ((8,) if hm.n >= 100 else ()):
      ref = lsmr(scaled_operator(J, d), f, damp=damp, maxiter=k)
      out = device_lsmr(ScaledMatrix(J, d), f, damp, maxiter=k)
      assert (out[1], out[2]) == (ref[1], ref[2]), (name, k, out[1:3], ref[1:3])
      for a, b in zip(out[3:], ref[3:]):
        assert abs(a - b) <= 1e-10 * abs(b), (name, damp, k, out[3:], ref[3:])
      assert np.linalg.norm(out[0] - ref[0]) <= 1e-10 * np.linalg.norm(ref[0])
    ref = lsmr(scaled_operator(J, d), f, damp=damp)
    out = device_lsmr(ScaledMatrix(J, d), f, damp)
    assert out[1] == ref[1] and abs(out[2] - ref[2]) <= max(2, 0.03 * ref[2]), (name, damp, out[1:3], ref[1:3])


@pytest.mark.parametrize("name", EMU_CASES)
def test_device_trust_region_driver_equals_scipys(name):
  """csrc/mcba_api.hip: solve_lsmr's driver -- Cauchy damping, the 2-D subspace from the Gram matrix of {g_h, gn_h} and the products
  J_h g_h, J_h gn_h, p_h = alpha g_h + beta gn_h, radius update, termination -- walked through on the host with csrc/mcba_trmath.h itself
  (tests/lsmr_emulation.trf_lsmr_device_driver), around scipy's lsmr: the trajectory of scipy's own driver (same nfev / status, the
  same LSMR stopping reasons, end points within 1e-8 px -- the two differ in rounding only, which the LSMR calls amplify)."""
  from lsmr_emulation import trf_lsmr, trf_lsmr_device_driver
  g, rig = load_golden(name)
  hm = HostMath(mirror(rig))

  def rms(x):
    e, v = hm.reprojection_error(x)
    return float(np.sqrt(np.mean(e[v] ** 2)))
  ca, cb = [], []
  a = trf_lsmr(hm.residuals, hm.jacobian, g["x0"], solver="scipy", calls=ca)
  b = trf_lsmr_device_driver(hm.residuals, hm.jacobian, g["x0"], solver="scipy", calls=cb)
  spread = float(np.abs(g["ba_pert_rms"] - g["ba_rms"]).max())
  if spread < 1e-6:     # (elsewhere the reference's own trajectory changes under 1e-12 px of noise: tiny_rolling)
    assert (a["nfev"], a["status"]) == (b["nfev"], b["status"]) == (int(g["ba_nfev"]), int(g["ba_status"]))
    assert [c["istop"] for c in ca] == [c["istop"] for c in cb]
  assert ca[0]["damp"] == pytest.approx(cb[0]["damp"], rel=1e-12) and ca[0]["Delta"] == pytest.approx(cb[0]["Delta"], rel=1e-14)
  assert abs(rms(a["x"]) - rms(b["x"])) <= max(1e-8, 3 * spread), (rms(a["x"]) - float(g["ba_rms"]), rms(b["x"]) - float(g["ba_rms"]))
  c = trf_lsmr_device_driver(hm.residuals, hm.jacobian, g["x0"], solver="device")    # driver AND LSMR flow of the device
  if spread < 1e-6:
    assert (c["nfev"], c["status"]) == (a["nfev"], a["status"])
  assert abs(rms(c["x"]) - float(g["ba_rms"])) <= max(1e-6, 3 * spread)


def test_chunked_enqueueing_keeps_ranks_matched():
  """csrc/mcba_lsmr.h: lsmr_chunk_allowed -- the rule by which the ranks of a frame-sharded solve enqueue LSMR iterations (each carries a
  collective) in chunks of 8, two chunks ahead of the progress word.  Simulation: R ranks poll the (monotone, rank-independent) word at
  random moments; call j executes once EVERY rank has enqueued it and then publishes the tests of step j - 1.  For every stop step s and
  every polling schedule all ranks end with the same number of enqueued calls -- (s // 8 + 2) * 8, capped at maxiter + 1 -- and nobody
  waits for a call that is never enqueued (no deadlock)."""
  import ctypes as C
  import hostmath_lib
  lib = hostmath_lib.lib()
  lib.hm_lsmr_chunk_allowed.restype = C.c_int64
  allowed = lambda have, istop, done, cap: int(lib.hm_lsmr_chunk_allowed(int(have), int(istop), C.c_int64(done), C.c_int64(8), C.c_int64(cap)))
  rng = np.random.default_rng(3)
  for trial in range(300):
    R = int(rng.integers(1, 9))
    maxiter = int(rng.integers(1, 80))
    s = int(rng.integers(1, maxiter + 1))              # the step whose tests stop the solve (istop 7 at the latest: s = maxiter)
    cap = maxiter + 1
    enq = [0] * R
    finished = [False] * R
    seen = [None] * R                                  # last word a rank has read: (istop, done)
    executed = 0                                       # calls the device has executed (= min over ranks of what is enqueued, in order)
    for step in range(100000):
      executed_possible = min(enq)
      # the device runs ahead as far as all ranks have enqueued, but kernels behind the stop publish nothing new
      executed = max(executed, executed_possible)
      word = None if executed == 0 else ((2, s) if executed >= s + 1 else (0, executed - 1))
      r = int(rng.integers(0, R))
      if finished[r]:
        if all(finished):
          break
        continue
      if rng.random() < 0.7:
        seen[r] = word                                 # the rank looks at the pinned word
      have = seen[r] is not None
      istop, done = seen[r] if have else (0, 0)
      a = allowed(have, istop, done, cap)
      if have and istop != 0 and enq[r] >= a:
        finished[r] = True
      elif enq[r] < a:
        enq[r] += 1
    assert all(finished), (trial, R, maxiter, s, enq)
    expect = min((s // 8 + 2) * 8, cap)
    assert enq == [expect] * R, (trial, R, maxiter, s, enq, expect)


def test_exact_products_fixture_separates_the_two_arithmetics():
  """tests/golden/exact_products.json (oracle/make_exact_products.py: scipy's own trf + lsmr on the REFERENCE's residual function, products
  in double with reordered rows vs accumulated in 80-bit precision): on every BASELINE-size rig the 80-bit runs end BELOW the double runs and
  within 1e-6 px of each other (1.1e-6 at 8 x 500 x 2, where the double runs scatter as much), with the reference's nfev -- the data behind test_default_solver_lands_on_scipys_exact_product_end_point (GPU)
  and DESIGN.md section 2."""
  import json
  path = os.path.join(ROOT, "tests", "golden", "exact_products.json")
  xp = json.load(open(path))
  assert {"cfg2", "cfg3_40", "cfg4_40", "cfg5_40", "manypairs"} <= set(xp)
  for name, e in xp.items():
    runs = e["runs"]
    assert all(r["nfev"] == e["reference_nfev"] and r["status"] == 2 for r in runs), name
    dbl = [r["rms_minus_reference"] for r in runs if r["arithmetic"] == "double"]
    ldb = [r["rms_minus_reference"] for r in runs if r["arithmetic"] == "longdouble"]
    assert dbl and ldb, name
    assert max(ldb) - min(ldb) <= (1.5e-6 if name in ("cfg3", "cfg4") else 1e-6), (name, ldb)     # (8 x 500 x 2: four runs over 1.07e-6)
    assert np.mean(ldb) < np.mean(dbl), (name, dbl, ldb)
    assert abs(np.mean(dbl)) <= 1.5e-6 and -4e-6 <= np.mean(ldb) <= 0.0, (name, dbl, ldb)
