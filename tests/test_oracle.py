"""CPU: the oracle (oracle/restate.py) against the golden fixtures produced by the REAL reference."""
import numpy as np
import pytest

from oracle import restate
from util import ALL_CASES, load_golden, oracle, golden_jacobian


@pytest.mark.parametrize("name", ALL_CASES)
def test_oracle_matches_reference_fixtures(name):
  g, rig = load_golden(name)
  oc = oracle(rig)
  assert np.array_equal(oc.param_vec, g["x0"])                      # parameter packing, bit-exact
  assert np.array_equal(oc.inliers, g["inliers0"])
  r0 = oc.evaluate(g["x0"])
  assert np.array_equal(r0, g["r0"])                                # same numpy/scipy arithmetic -> bit-exact
  assert np.array_equal(oc.reprojection_error, g["err0"])
  assert restate.error_stats(oc.reprojection_error).rms == pytest.approx(float(g["rms0"]), rel=0, abs=0)


@pytest.mark.parametrize("name", ["tiny", "tiny_handeye", "tiny_edge"])
def test_oracle_bundle_adjust_matches_reference(name):
  g, rig = load_golden(name)
  oc = oracle(rig)
  out, res = oc.bundle_adjust(return_result=True)
  assert res.nfev == int(g["ba_nfev"]) and res.status == int(g["ba_status"])
  assert np.array_equal(res.x, g["ba_x_raw"])
  assert np.array_equal(out.param_vec, g["ba_x"])
  assert restate.error_stats(out.reprojection_error).rms == float(g["ba_rms"])


def test_oracle_sparsity_matches_fd_pattern():
  g, rig = load_golden("tiny_rolling")
  oc = oracle(rig)
  S = oc.sparsity_matrix.tocsr()
  J = golden_jacobian(g)
  assert S.shape == J.shape
  # every finite-difference non-zero lies inside the reference's sparsity pattern
  assert (abs(J) > 0).multiply(S == 0).nnz == 0


def test_oracle_outlier_loop_matches_reference():
  g, rig = load_golden("tiny")
  oc = oracle(rig)
  ao = oc.adjust_outliers(num_adjustments=3, select_outliers=restate.select_threshold(0.75, 5.0),
                          loss='linear', tolerance=1e-4)
  assert np.array_equal(ao.inliers, g["ao_inliers"])
  assert np.array_equal(ao.param_vec, g["ao_x"])
  assert restate.error_stats(ao.reprojection_error).rms == float(g["ao_rms"])
