"""TEST INFRASTRUCTURE: ctypes wrapper of tests/hostmath (g++ build of the product's __host__ __device__ functions)."""
import ctypes as C
import os
import subprocess

import numpy as np

from multical_amd import _lib
from multical_amd.backend import lower, _to_struct, _ptr, _f64

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostmath", "hostmath.cpp")
OUT_DIR = os.path.join(HERE, "hostmath", "_build")
LIB = os.path.join(OUT_DIR, "libmcba_hostmath.so")


def build(force=False):
  os.makedirs(OUT_DIR, exist_ok=True)
  csrc = os.path.join(os.path.dirname(HERE), "multical_amd", "csrc")
  deps = [SRC] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
  if force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", LIB, SRC])
  return LIB


_h = None


def lib():
  global _h
  if _h is None:
    _h = C.CDLL(build())
    _h.hm_last_error.restype = C.c_char_p
  return _h


def _check(rc):
  if rc != 0:
    raise RuntimeError(lib().hm_last_error().decode())


class HostMath(object):
  def __init__(self, calib, frame_range=None):
    self.prob = lower(calib)
    self.struct = _to_struct(self.prob, frame_range)
    n, m, k = C.c_int64(), C.c_int64(), C.c_int32()
    _check(lib().hm_sizes(C.byref(self.struct), C.byref(n), C.byref(m), C.byref(k)))
    self.n, self.m, self.row_nnz = n.value, m.value, k.value

  def residuals(self, x):
    x = _f64(x)
    r = np.zeros(self.m)
    _check(lib().hm_residuals(C.byref(self.struct), _ptr(x, C.c_double), _ptr(r, C.c_double), None, None))
    return r

  def reprojection_error(self, x):
    x = _f64(x)
    err = np.zeros(self.prob.shape)
    valid = np.zeros(self.prob.shape, dtype=np.uint8)
    _check(lib().hm_residuals(C.byref(self.struct), _ptr(x, C.c_double), None, _ptr(err, C.c_double),
                              _ptr(valid, C.c_uint8)))
    return err, valid.astype(bool)

  def jacobian(self, x):
    from scipy.sparse import csr_matrix
    x = _f64(x)
    k, m = self.row_nnz, self.m
    vals = np.zeros((m, k))
    cols = np.zeros((m // 2, k), dtype=np.int32)
    _check(lib().hm_jacobian(C.byref(self.struct), _ptr(x, C.c_double), k, _ptr(vals, C.c_double), _ptr(cols, C.c_int32)))
    indices = np.repeat(cols, 2, axis=0).ravel()
    if (cols < 0).any():   # ragged camera blocks: slots of coefficients a camera's model does not have
      keep = indices >= 0
      counts = keep.reshape(m, k).sum(axis=1)
      return csr_matrix((vals.ravel()[keep], indices[keep], np.concatenate([[0], np.cumsum(counts)])), shape=(m, self.n))
    return csr_matrix((vals.ravel(), indices, np.arange(0, m * k + 1, k)), shape=(m, self.n))

  def normal_equations(self, x, loss='linear', f_scale=1.0):
    x = _f64(x)
    H = np.zeros((self.n, self.n))
    g = np.zeros(self.n)
    cost = C.c_double()
    _check(lib().hm_normal_equations(C.byref(self.struct), _ptr(x, C.c_double), _lib.LOSSES[loss], C.c_double(f_scale),
                                     _ptr(H, C.c_double), _ptr(g, C.c_double), C.byref(cost)))
    return H, g, cost.value

  def lsmr_products(self, x, v, u):
    """(J v, J^T u) through the matrix-free factorisation the lsmr mode's kernels use (device functions, serial)."""
    x, v, u = _f64(x), _f64(v), _f64(u)
    jv, jtu = np.zeros(self.m), np.zeros(self.n)
    _check(lib().hm_lsmr_products(C.byref(self.struct), _ptr(x, C.c_double), _ptr(v, C.c_double), _ptr(u, C.c_double),
                                  _ptr(jv, C.c_double), _ptr(jtu, C.c_double)))
    return jv, jtu
