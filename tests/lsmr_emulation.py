"""TEST INFRASTRUCTURE (CPU): the device's LSMR solve and trust-region driver, emulated step by step with numpy vectors.

`device_lsmr` walks through exactly what the two-launch iteration does on the GPU (csrc/mcba_api.hip: lsmr_solve,
k_lsmr_fused2 / k_lsmr_gather3): the prologue, u kept UN-normalised (1 / beta applied where it is read), v kept un-normalised
(1 / alpha applied where it is read), rotation + vector update of a step in the tail of the NEXT product launch, the stopping
tests of a step in front of the next gather -- with the scalar recurrences of csrc/mcba_lsmr.h themselves (compiled into
tests/hostmath by g++).  Only the summation order of the products and norms differs from the device (numpy / scipy.sparse).
`trf_lsmr` is scipy's trf_no_bounds (scipy/optimize/_lsq/trf.py:401-560, tr_solver='lsmr', x_scale='jac') with the LSMR solve
pluggable, so that scipy's own `lsmr` and the emulation can be swapped on the same Jacobian and every call's
(istop, itn, normr, normar, normA, condA, normx) compared.  Used by tests/test_host.py; never imported by multical_amd.
"""
import ctypes as C

import numpy as np
from numpy.linalg import norm
from scipy.optimize._lsq.common import (solve_trust_region_2d, update_tr_radius, check_termination, minimize_quadratic_1d)
from scipy.sparse.linalg import lsmr as scipy_lsmr, LinearOperator

import hostmath_lib


class LsmrState(object):
  """The LS_* block of csrc/mcba_lsmr.h on the host."""

  def __init__(self):
    self.lib = hostmath_lib.lib()
    self.lib.hm_lsmr_state_test.restype = C.c_int32
    self.L = np.zeros(self.lib.hm_lsmr_nslots())
    self.p = self.L.ctypes.data_as(C.POINTER(C.c_double))

  def slot(self, name):
    i = self.lib.hm_lsmr_slot(name.encode())
    assert i >= 0, name
    return float(self.L[i])

  def init(self, alpha, beta, damp, normb, maxiter):
    self.lib.hm_lsmr_state_init(self.p, C.c_double(alpha), C.c_double(beta), C.c_double(damp), C.c_double(normb), C.c_double(maxiter))

  def beta(self, u2):
    self.lib.hm_lsmr_state_beta(self.p, C.c_double(u2))

  def rotate(self, v2):
    self.lib.hm_lsmr_state_rotate(self.p, C.c_double(v2))

  def test(self, x2):
    return int(self.lib.hm_lsmr_state_test(self.p, C.c_double(x2)))


def device_lsmr(A, b, damp, maxiter=None, sumsq=None):
  """The two-launch device iteration on a LinearOperator / sparse matrix A.  Returns scipy's tuple
  (x, istop, itn, normr, normar, normA, condA, normx).  sumsq(vector) = the squared norm (default: np.dot)."""
  sumsq = sumsq or (lambda a: float(np.dot(a, a)))
  m, n = A.shape
  maxiter = min(m, n) if maxiter is None else maxiter
  st = LsmrState()
  # ---- prologue (lsmr_solve): u = b, beta = |u|, v = A^T (u / beta), alpha = |v|, v /= alpha, h = v
  u = np.array(b, dtype=np.float64)
  normb = np.sqrt(sumsq(u))
  beta = normb
  x = np.zeros(n)
  if beta > 0:
    u = u * (1.0 / beta)
    v = A.T @ u
    alpha = np.sqrt(sumsq(v))
  else:
    v = np.zeros(n)
    alpha = 0.0
  if alpha > 0:
    v = (1.0 / alpha) * v
  if alpha * beta == 0 or normb == 0:
    return x, 0, 0, beta, alpha * beta, alpha, 1.0, 0.0
  h = v.copy()
  hbar = np.zeros(n)
  st.init(alpha, beta, damp, normb, float(maxiter))
  pending = False
  vstore = v                 # what the v buffer holds: v_raw of the last gather (or the normalised v of the prologue)
  uhat = u                   # what the u buffer holds: the un-normalised uhat (or the normalised u of the prologue)
  x2 = 0.0
  enqueued = 0
  while enqueued <= maxiter:
    # ---- k_lsmr_fused2: head
    alpha, inv_alpha, v2 = st.slot("alpha"), st.slot("inv_alpha"), 0.0
    inv_beta_old = st.slot("inv_beta")
    if pending:
      v2 = sumsq(vstore)
      inv_alpha = 1.0
      if st.slot("skipv") == 0.0:
        alpha = np.sqrt(v2)
        if alpha > 0:
          inv_alpha = 1.0 / alpha
    # body: uhat <- A (v_raw / alpha) - alpha (uhat_old / beta_old)
    vn = vstore * inv_alpha
    uhat = A @ vn - alpha * (uhat * inv_beta_old)
    u2 = sumsq(uhat)
    # tail: rotation + vector update of the step whose v_raw was pending
    if pending:
      st.rotate(v2)
      hbar = st.slot("c_hbar") * hbar + h
      x = x + st.slot("c_x") * hbar
      h = st.slot("c_h") * h + vstore * st.slot("inv_alpha")
      x2 = sumsq(x)
    # ---- k_lsmr_gather3: stopping tests of the completed step, then beta and the new v_raw
    istop = st.test(x2) if st.slot("itn") > 0 else 0
    if istop != 0:
      break
    st.beta(u2)
    if st.slot("skipv") == 0.0:
      vstore = (A.T @ uhat) * st.slot("inv_beta") - st.slot("beta") * vn
    else:
      vstore = vn
    pending = True
    enqueued += 1
  return (x, istop, int(st.slot("itn")), st.slot("normr"), st.slot("normar"), st.slot("norma"), st.slot("conda"), float(np.sqrt(x2)))


def scaled_operator(J, d):
  """J_h = J diag(d) as scipy forms it for a sparse Jacobian (common.py: right_multiplied_operator)"""
  m, n = J.shape
  return LinearOperator((m, n), matvec=lambda x: J @ (np.ravel(x) * d), rmatvec=lambda x: d * (J.T @ np.ravel(x)), dtype=np.float64)


class ScaledMatrix(object):
  """J diag(d) with `@` and `.T @` (what device_lsmr needs), products formed as the device forms them: J (d v), d (J^T u)"""

  def __init__(self, J, d):
    self.J, self.d, self.shape = J, d, J.shape
    self.JT = J.T.tocsr()

  def __matmul__(self, v):
    return self.J @ (self.d * v)

  @property
  def T(self):
    outer = self

    class _T(object):
      def __matmul__(self, u):
        return outer.d * (outer.JT @ u)
    return _T()


def trf_lsmr(fun, jac, x0, solver="scipy", ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=100, calls=None, sumsq=None):
  """scipy's trf_no_bounds with tr_solver='lsmr', x_scale='jac', linear loss, statement by statement with scipy's own helper functions
  (bit-identical to scipy.optimize.least_squares: tests/test_host.py); solver = "scipy" (scipy.sparse.linalg.lsmr), "device"
  (device_lsmr) or a callable(x, scale, damp, J, f) returning scipy's tuple.  calls (a list) receives one dict per LSMR call."""
  from scipy.linalg import qr
  from scipy.optimize._lsq.common import (compute_grad, compute_jac_scale, right_multiplied_operator, build_quadratic_1d,
                                          evaluate_quadratic)
  x = np.array(x0, dtype=np.float64)
  f = fun(x)
  nfev, njev = 1, 1
  J = jac(x)
  m, n = J.shape
  cost = 0.5 * np.dot(f, f)
  g = compute_grad(J, f)
  scale, scale_inv = compute_jac_scale(J)
  Delta = norm(x * scale_inv)
  if Delta == 0:
    Delta = 1.0
  status, iteration, step_norm, actual_reduction = None, 0, None, None
  while True:
    g_norm = norm(g, ord=np.inf)
    if g_norm < gtol:
      status = 1
    if status is not None or nfev == max_nfev:
      break
    d = scale
    g_h = d * g
    J_h = right_multiplied_operator(J, d)
    a, b = build_quadratic_1d(J_h, g_h, -g_h)
    to_tr = Delta / norm(g_h)
    ag_value = minimize_quadratic_1d(a, b, 0, to_tr)[1]
    reg_term = -ag_value / Delta**2
    damp = (0.0**2 + reg_term)**0.5
    if callable(solver):          # e.g. the device's own LSMR call on this linearisation (mcba_debug_lsmr_solve)
      out = solver(x=x, scale=d, damp=damp, J=J, f=f)
    elif solver == "scipy":
      out = scipy_lsmr(J_h, f, damp=damp)
    else:
      out = device_lsmr(ScaledMatrix(J, d), f, damp, sumsq=sumsq)
    gn_h = out[0]
    if calls is not None:
      calls.append(dict(iteration=iteration, x=x.copy(), scale=d.copy(), gn_h=np.array(gn_h), damp=damp, Delta=Delta, istop=int(out[1]),
                        itn=int(out[2]), normr=float(out[3]), normar=float(out[4]), normA=float(out[5]), condA=float(out[6]),
                        normx=float(out[7])))
    S = np.vstack((g_h, gn_h)).T
    S, _ = qr(S, mode='economic')
    JS = J_h.dot(S)
    B_S = np.dot(JS.T, JS)
    g_S = S.T.dot(g_h)
    actual_reduction = -1
    while actual_reduction <= 0 and nfev < max_nfev:
      p_S, _ = solve_trust_region_2d(B_S, g_S, Delta)
      step_h = S.dot(p_S)
      predicted_reduction = -evaluate_quadratic(J_h, g_h, step_h)
      step = d * step_h
      x_new = x + step
      f_new = fun(x_new)
      nfev += 1
      step_h_norm = norm(step_h)
      if not np.all(np.isfinite(f_new)):
        Delta = 0.25 * step_h_norm
        continue
      cost_new = 0.5 * np.dot(f_new, f_new)
      actual_reduction = cost - cost_new
      Delta_new, ratio = update_tr_radius(Delta, actual_reduction, predicted_reduction, step_h_norm, step_h_norm > 0.95 * Delta)
      step_norm = norm(step)
      status = check_termination(actual_reduction, cost, step_norm, norm(x), ratio, ftol, xtol)
      if status is not None:
        break
      Delta = Delta_new
    if actual_reduction > 0:
      x, f, cost = x_new, f_new, cost_new
      J = jac(x)
      njev += 1
      g = compute_grad(J, f)
      scale, scale_inv = compute_jac_scale(J, scale_inv)
    else:
      step_norm, actual_reduction = 0, 0
    iteration += 1
  return dict(x=x, cost=cost, nfev=nfev, njev=njev, status=status or 0, optimality=g_norm)


class TrBlock(object):
  """The TR_* scalar block of csrc/mcba_trmath.h on the host."""

  def __init__(self):
    self.lib = hostmath_lib.lib()
    self.lib.hm_tr_reg_term.restype = C.c_double
    self.lib.hm_tr_update_radius.restype = C.c_double
    self.S = np.zeros(self.lib.hm_tr_nslots())
    self.p = self.S.ctypes.data_as(C.POINTER(C.c_double))

  def __setitem__(self, name, v):
    self.S[self.lib.hm_tr_slot(name.encode())] = v

  def __getitem__(self, name):
    return float(self.S[self.lib.hm_tr_slot(name.encode())])


def trf_lsmr_device_driver(fun, jac, x0, solver="scipy", ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=100, calls=None):
  """csrc/mcba_api.hip: solve_lsmr step by step on the host -- the device's trust-region DRIVER (Cauchy damping, 2-D subspace from the
  Gram matrix of {g_h, gn_h} and the products J_h g_h, J_h gn_h, trial step p_h = alpha g_h + beta gn_h, radius update, termination)
  with the scalar algebra of csrc/mcba_trmath.h itself (compiled into tests/hostmath); the LSMR solve is pluggable as in trf_lsmr."""
  tr = TrBlock()
  lib = tr.lib
  x = np.array(x0, dtype=np.float64)
  f = fun(x)
  J = jac(x)
  nfev, njev, iteration, status = 1, 1, 0, -100
  cost = 0.5 * np.dot(f, f)
  g = J.T @ f
  scale_inv = np.asarray(J.power(2).sum(axis=0)).ravel() ** 0.5
  scale_inv[scale_inv == 0] = 1
  first = True
  Delta = 0.0
  while True:
    d = 1 / scale_inv
    g_h = d * g
    g_norm, gg, xs = np.abs(g).max(), float(np.dot(g_h, g_h)), float(np.dot(x * scale_inv, x * scale_inv))
    if first:
      Delta = np.sqrt(xs) or 1.0
      first = False
    if g_norm < gtol:
      status = 1
    if status != -100 or nfev >= max_nfev:
      break
    Jg = J @ (d * g_h)
    Q00 = float(np.dot(Jg, Jg))
    reg_term = lib.hm_tr_reg_term(C.c_double(Q00), C.c_double(gg), C.c_double(Delta), C.c_double(0.0)) if gg > 0 else 0.0
    damp = np.sqrt(reg_term)
    if callable(solver):
      out = solver(x=x, scale=d, damp=damp, J=J, f=f)
    elif solver == "scipy":
      out = scipy_lsmr(scaled_operator(J, d), f, damp=damp)
    else:
      out = device_lsmr(ScaledMatrix(J, d), f, damp)
    gn = np.array(out[0])
    if calls is not None:
      calls.append(dict(iteration=iteration, damp=damp, Delta=Delta, istop=int(out[1]), itn=int(out[2])))
    Jgn = J @ (d * gn)
    tr["reg"], tr["q00"] = reg_term, float(np.dot(Jg, Jg))
    tr["d00"], tr["d01"], tr["d11"] = float(np.dot(g_h, g_h)), float(np.dot(g_h, gn)), float(np.dot(gn, gn))
    tr["gnorm"], tr["gh2"], tr["xs2"] = g_norm, gg, xs
    lib.hm_tr_subspace(tr.p, 1, C.c_double(float(np.dot(Jg, Jgn))), C.c_double(float(np.dot(Jgn, Jgn))))
    actual_reduction, cost_new = -1.0, cost
    while actual_reduction <= 0 and nfev < max_nfev:
      lib.hm_tr_trial(tr.p, C.c_double(Delta))
      p = tr["alpha"] * g_h + tr["beta"] * gn
      x_new = x + d * p
      f_new = fun(x_new)
      nfev += 1
      step_h_norm = np.sqrt(np.dot(p, p))
      cost_new = 0.5 * np.dot(f_new, f_new)
      if not np.isfinite(cost_new):
        Delta = 0.25 * step_h_norm
        continue
      actual_reduction = cost - cost_new
      ratio = C.c_double(0.0)
      Delta_new = lib.hm_tr_update_radius(C.c_double(Delta), C.c_double(actual_reduction), C.c_double(tr["pred"]), C.c_double(step_h_norm),
                                          int(step_h_norm > 0.95 * Delta), C.byref(ratio))
      status = lib.hm_tr_check_termination(C.c_double(actual_reduction), C.c_double(cost), C.c_double(norm(d * p)), C.c_double(norm(x)),
                                           ratio, C.c_double(ftol), C.c_double(xtol))
      if status != -100:
        break
      Delta = Delta_new
    if actual_reduction > 0:
      x, f, cost = x_new, f_new, cost_new
      J = jac(x)
      njev += 1
      g = J.T @ f
      scale_inv = np.maximum(np.asarray(J.power(2).sum(axis=0)).ravel() ** 0.5, scale_inv)
    iteration += 1
  return dict(x=x, cost=cost, nfev=nfev, njev=njev, status=0 if status == -100 else status, optimality=g_norm)
