"""Numpy prototype of the native solver's algorithm (test helper, CPU only).

scipy's `trf_no_bounds` (scipy/optimize/_lsq/trf.py:401-560) with ONE change: the regularised Gauss-Newton
step that scipy obtains with LSMR (`lsmr(J_h, f, damp=sqrt(reg_term))`) is computed exactly from the damped
normal equations (J_h^T J_h + reg I) p = J_h^T f -- which is what the HIP back-end does with a Schur
complement + Cholesky.  Everything else (x_scale='jac', 2-D subspace, radius update, termination) uses
scipy's own helper functions, so this file documents exactly which behaviour the C++ driver mirrors.
"""
import numpy as np
from numpy.linalg import norm
from scipy.optimize._lsq.common import (solve_trust_region_2d, update_tr_radius, check_termination,
                                        minimize_quadratic_1d)

REG_FLOOR = 1e-14


def trf_exact(fun, jac, x0, ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=100, log=None, reg_floor=REG_FLOOR):
  x = x0.copy()
  f = fun(x)
  nfev, njev = 1, 1
  J = jac(x)
  cost = 0.5 * f @ f
  H = (J.T @ J)
  H = H.toarray() if hasattr(H, "toarray") else np.asarray(H)
  g = J.T @ f
  scale_inv = np.sqrt(np.diag(H)).copy()
  scale_inv[scale_inv == 0] = 1
  scale = 1 / scale_inv
  Delta = norm(x0 * scale_inv) or 1.0
  status, iteration, step_norm, actual_reduction = None, 0, None, None
  n = x.size
  while True:
    g_norm = norm(g, ord=np.inf)
    if g_norm < gtol:
      status = 1
    if log is not None:
      log.append((iteration, nfev, cost, actual_reduction, step_norm, g_norm))
    if status is not None or nfev == max_nfev:
      break
    d = scale
    g_h = d * g
    H_h = H * d[:, None] * d[None, :]
    a = 0.5 * (g_h @ H_h @ g_h)
    b = -(g_h @ g_h)
    to_tr = Delta / norm(g_h)
    ag_value = minimize_quadratic_1d(a, b, 0, to_tr)[1]
    reg_term = -ag_value / Delta**2
    A = H_h + max(reg_term, reg_floor) * np.eye(n)
    L = np.linalg.cholesky(A)
    gn_h = np.linalg.solve(L.T, np.linalg.solve(L, g_h))
    S = np.vstack((g_h, gn_h)).T
    S, _ = np.linalg.qr(S)
    B_S = S.T @ H_h @ S
    g_S = S.T @ g_h
    actual_reduction = -1
    while actual_reduction <= 0 and nfev < max_nfev:
      p_S, _ = solve_trust_region_2d(B_S, g_S, Delta)
      step_h = S @ p_S
      predicted_reduction = -(0.5 * p_S @ B_S @ p_S + g_S @ p_S)
      step = d * step_h
      x_new = x + step
      f_new = fun(x_new)
      nfev += 1
      step_h_norm = norm(step_h)
      if not np.all(np.isfinite(f_new)):
        Delta = 0.25 * step_h_norm
        continue
      cost_new = 0.5 * f_new @ f_new
      actual_reduction = cost - cost_new
      Delta_new, ratio = update_tr_radius(Delta, actual_reduction, predicted_reduction, step_h_norm,
                                          step_h_norm > 0.95 * Delta)
      step_norm = norm(step)
      status = check_termination(actual_reduction, cost, step_norm, norm(x), ratio, ftol, xtol)
      if status is not None:
        break
      Delta = Delta_new
    if actual_reduction > 0:
      x, f, cost = x_new, f_new, cost_new
      J = jac(x)
      njev += 1
      H = (J.T @ J)
      H = H.toarray() if hasattr(H, "toarray") else np.asarray(H)
      g = J.T @ f
      scale_inv = np.maximum(np.sqrt(np.diag(H)), scale_inv)
      scale = 1 / scale_inv
    else:
      step_norm, actual_reduction = 0, 0
    iteration += 1
  return dict(x=x, cost=cost, nfev=nfev, njev=njev, status=status or 0, optimality=g_norm)
