"""Shared helpers of the test-suite (TEST INFRASTRUCTURE)."""
import os

import numpy as np

from multical_amd import synthetic, calibration
from oracle import restate

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL_CASES = ["tiny", "tiny_rolling", "tiny_fisheye", "tiny_handeye", "tiny_rational", "tiny_thin_prism",
               "tiny_tilted", "tiny_edge", "tiny_fixintr", "tiny_pin4", "tiny_bigboard", "tiny_fishmix"]
ALL_CASES = SMALL_CASES + ["cfg1"]


def load_golden(name):
  g = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))
  rig = synthetic.rig_from_arrays(g)
  return g, rig


def golden_jacobian(g):
  from scipy.sparse import csr_matrix
  return csr_matrix((g["J_data"], g["J_indices"], g["J_indptr"]), shape=tuple(g["J_shape"]))


def mirror(rig):
  return calibration.from_rig(rig)


def oracle(rig):
  return restate.from_rig(rig)


def rel_col_error(J, Jref):
  """max over columns of |J - Jref|_inf / |Jref|_inf (absolute where the reference column is zero)."""
  D = np.abs((J - Jref).toarray()).max(axis=0)
  s = np.abs(Jref.toarray()).max(axis=0)
  return np.where(s > 0, D / np.where(s > 0, s, 1), D).max()


def gpu_available():
  try:
    import torch
    return torch.cuda.is_available()
  except Exception:
    return False


def sub_rig(rig, frames):
  """The first `frames` frames of a synthetic rig as a rig of its own (same cameras, boards, initial guess): lets the
  CPU oracle check a slice of a full-size BASELINE configuration in seconds."""
  from multical_amd import synthetic
  arrs = dict(synthetic.rig_to_arrays(rig))
  for k in ("points", "valid"):
    arrs[k] = arrs[k][:, :frames]
  arrs["frame_valid"] = arrs["frame_valid"][:frames]
  for prefix in ("init_", "truth_"):
    for k in ("rig", "rig_end", "he_base_wrt_gripper"):
      if prefix + k in arrs:
        arrs[prefix + k] = arrs[prefix + k][:frames]
  return synthetic.rig_from_arrays(arrs)
