// mcba_trmath.h -- the scalar algebra of scipy's trust-region-reflective driver without bounds
// (scipy/optimize/_lsq/trf.py:trf_no_bounds, common.py): Cauchy regularisation, the 2-D subspace problem, the radius
// update and the termination tests.  __host__ __device__: the single-GPU driver evaluates it in one-thread kernels
// between the vector kernels (k_tr_reg and the head of k_vec_step in mcba_solver_kernels.h) so that an iteration needs one host
// synchronisation instead of three; the host runs the SAME source for rejected trial steps and for frame-sharded
// handles.  No std:: containers or std::complex here.
#pragma once
#include "mcba_math.h"

namespace mcba {

constexpr double TR_REG_FLOOR = 1e-10;   // floor of the Levenberg-Marquardt damping in the scaled space (gauge null space)

// layout of the scalar block shared by the driver kernels and the host (doubles, device: scal[0 .. 32))
enum TrSlot {
  TR_GNORM = 0,      // |g|_inf
  TR_GH2 = 1,        // |g_h|^2
  TR_XS2 = 2,        // |x * scale_inv|^2
  TR_COST_NEW = 3,   // trial cost (filled by the host from the k_cost partials)
  TR_Q00 = 4,        // g_h^T H_h g_h
  TR_REG = 5,        // damping used by the Gauss-Newton solve
  TR_DELTA = 6,      // trust radius the step was computed for
  TR_D00 = 7, TR_D01 = 8, TR_D11 = 9,   // g_h.g_h, g_h.gn, gn.gn
  TR_ALPHA = 10, TR_BETA = 11,          // p_h = alpha g_h + beta gn
  TR_INFO = 12,      // Cholesky pivot report (0 = ok)
  TR_PRED = 13,      // predicted reduction of the step
  TR_BS0 = 14, TR_BS1 = 15, TR_BS2 = 19,   // 2-D model in the orthonormal basis (B_S symmetric: b00, b01, b11)
  TR_COST = 16, TR_COUNT = 17,          // cost / observation count of the linearisation
  TR_GS0 = 18,       // gradient in the basis: (|g_h|, 0)
  TR_CM0 = 20,       // [20 .. 24): Cm, row-major 2 x 2: (alpha, beta) = Cm p_S
  TR_NSLOTS = 32
};

struct cplx { double re, im; };
MCBA_HD cplx c_mul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
MCBA_HD cplx c_sub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
MCBA_HD double c_abs(cplx a) { return hypot(a.re, a.im); }
MCBA_HD cplx c_div(cplx a, cplx b) {
  const double den = b.re * b.re + b.im * b.im;
  return {(a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den};
}

// scipy common.py minimize_quadratic_1d (c = 0): argmin of t (a t + b) over [lb, ub]
MCBA_HD void tr_minimize_quadratic_1d(double a, double b, double lb, double ub, double* t_out, double* y_out) {
  double t[3] = {lb, ub, 0};
  int n = 2;
  if (a != 0) {
    const double ext = -0.5 * b / a;
    if (lb < ext && ext < ub) t[n++] = ext;
  }
  double best = INFINITY;
  for (int i = 0; i < n; ++i) {
    const double y = t[i] * (a * t[i] + b);
    if (y < best) { best = y; *t_out = t[i]; }
  }
  *y_out = best;
}

// damping of the Gauss-Newton step from the Cauchy step (trf.py:338-340, build_quadratic_1d + minimize_quadratic_1d)
// floor: the exact normal-equation solve needs a positive definite reduced system (no pose is fixed: gauge null space);
// scipy's LSMR branch takes the value as it is (floor = 0)
MCBA_HD double tr_reg_term(double q00, double gh2, double Delta, double floor = TR_REG_FLOOR) {
  double tmin, ag_value;
  tr_minimize_quadratic_1d(0.5 * q00, -gh2, 0.0, Delta / sqrt(gh2), &tmin, &ag_value);
  const double reg_term = -ag_value / (Delta * Delta);
  return reg_term > floor ? reg_term : floor;
}

// real roots of a polynomial of degree <= 4 (coefficients highest power first); Durand-Kerner + Newton polish
// (stands in for numpy.roots in solve_trust_region_2d, common.py:178-181)
MCBA_HD int tr_real_roots(const double* coeffs_in, int ncoef, double* roots) {
  int start = 0;
  while (start < ncoef && coeffs_in[start] == 0.0) ++start;   // numpy.roots strips leading zeros
  int deg = ncoef - start - 1;
  if (deg <= 0) return 0;
  double c[5];
  for (int i = 0; i <= deg; ++i) c[i] = coeffs_in[start + i];
  int nroots = 0;
  while (deg > 0 && c[deg] == 0.0) { roots[nroots++] = 0.0; --deg; }   // trailing zeros -> roots at 0
  if (deg == 0) return nroots;
  cplx z[4];
  const double lead = c[0];
  double bound = 0;
  for (int i = 1; i <= deg; ++i) bound = fmax(bound, fabs(c[i] / lead));
  bound = 1.0 + bound;
  for (int i = 0; i < deg; ++i) {
    const double ang = 0.4 + 2.0 * 3.14159265358979323846 * i / deg;
    z[i] = {bound * 0.7 * cos(ang), bound * 0.7 * sin(ang)};
  }
  for (int it = 0; it < 500; ++it) {
    double change = 0;
    for (int i = 0; i < deg; ++i) {
      cplx den = {lead, 0.0};
      for (int j = 0; j < deg; ++j)
        if (j != i) den = c_mul(den, c_sub(z[i], z[j]));
      if (c_abs(den) == 0) den = {1e-300, 0.0};
      cplx v = {c[0], 0.0};
      for (int k = 1; k <= deg; ++k) { v = c_mul(v, z[i]); v.re += c[k]; }
      const cplx dz = c_div(v, den);
      z[i] = c_sub(z[i], dz);
      change = fmax(change, c_abs(dz) / (1.0 + c_abs(z[i])));
    }
    if (change < 1e-15) break;
  }
  for (int i = 0; i < deg; ++i) {
    if (fabs(z[i].im) > 1e-7 * (1.0 + fabs(z[i].re))) continue;
    double x = z[i].re;
    for (int it = 0; it < 4; ++it) {   // Newton polish on the real axis
      double v = c[0], dv = 0;
      for (int k = 1; k <= deg; ++k) { dv = dv * x + v; v = v * x + c[k]; }
      if (dv == 0) break;
      const double nx = x - v / dv;
      if (!isfinite(nx)) break;
      x = nx;
    }
    roots[nroots++] = x;
  }
  return nroots;
}

// scipy common.py solve_trust_region_2d: min 0.5 p^T B p + g^T p, |p| <= Delta
MCBA_HD void tr_solve_2d(const double B[3] /*b00,b01,b11*/, const double g[2], double Delta, double p[2]) {
  const double b00 = B[0], b01 = B[1], b11 = B[2];
  const double det = b00 * b11 - b01 * b01;
  if (b00 > 0 && det > 0) {   // Cholesky succeeds <=> positive definite
    const double p0 = -(b11 * g[0] - b01 * g[1]) / det, p1 = -(-b01 * g[0] + b00 * g[1]) / det;
    if (p0 * p0 + p1 * p1 <= Delta * Delta) { p[0] = p0; p[1] = p1; return; }
  }
  const double a = b00 * Delta * Delta, b = b01 * Delta * Delta, c = b11 * Delta * Delta;
  const double dd = g[0] * Delta, f = g[1] * Delta;
  const double coeffs[5] = {-b + dd, 2 * (a - c + f), 6 * b, 2 * (-a + c + f), -b - dd};
  double roots[8];
  const int nr = tr_real_roots(coeffs, 5, roots);
  double best = INFINITY;
  p[0] = 0; p[1] = -Delta;   // t -> infinity limit of the parametrisation, as a safe fallback candidate
  for (int i = -1; i < nr; ++i) {
    double p0 = 0.0, p1 = -Delta;
    if (i >= 0) {
      const double t = roots[i], q = 1 + t * t;
      p0 = Delta * 2 * t / q;
      p1 = Delta * (1 - t * t) / q;
    }
    const double val = 0.5 * (p0 * (b00 * p0 + b01 * p1) + p1 * (b01 * p0 + b11 * p1)) + g[0] * p0 + g[1] * p1;
    if (val < best) { best = val; p[0] = p0; p[1] = p1; }
  }
}

// 2-D subspace model in an orthonormal basis [e0 e1] = [g_h gn] Cm of span{g_h, gn} (Gram-Schmidt on the Gram matrix);
// the quadratic forms with gn follow from (H_h + reg I) gn = g_h:  g_h^T H_h gn = |g_h|^2 - reg g_h.gn,
// gn^T H_h gn = g_h.gn - reg |gn|^2.  S is the TrSlot block: reads TR_Q00, TR_REG, TR_D00..TR_D11; writes BS, GS0, CM.
// (explicit = true: gn is an INEXACT solution -- LSMR -- and the forms g_h^T H_h gn, gn^T H_h gn are supplied by the caller,
//  computed from the products J_h g_h, J_h gn like scipy's B_S = (J_h S)^T (J_h S), trf.py:484-487)
MCBA_HD void tr_subspace(double* S, bool explicit_forms = false, double Q01_in = 0.0, double Q11_in = 0.0) {
  const double reg = S[TR_REG], d00 = S[TR_D00], d01 = S[TR_D01], d11 = S[TR_D11];
  const double Q00 = S[TR_Q00], Q01 = explicit_forms ? Q01_in : d00 - reg * d01, Q11 = explicit_forms ? Q11_in : d01 - reg * d11;
  const double n0 = sqrt(d00);
  const double proj = d01 / d00;
  const double nw2 = d11 - proj * d01;
  double* Cm = S + TR_CM0;
  S[TR_GS0] = n0;
  if (nw2 > 1e-28 * d11 && isfinite(nw2)) {
    const double nw = sqrt(nw2);
    Cm[0] = 1.0 / n0; Cm[1] = -proj / nw;
    Cm[2] = 0.0;      Cm[3] = 1.0 / nw;
    S[TR_BS0] = Q00 / d00;
    S[TR_BS1] = (Q01 - proj * Q00) / (n0 * nw);
    S[TR_BS2] = (Q11 - 2 * proj * Q01 + proj * proj * Q00) / nw2;
  } else {   // gn parallel to the gradient: 1-D subspace
    Cm[0] = 1.0 / n0; Cm[1] = 0; Cm[2] = 0; Cm[3] = 0;
    S[TR_BS0] = Q00 / d00; S[TR_BS1] = 0; S[TR_BS2] = 1.0;
  }
}

// trial step for radius Delta in the subspace of S: writes TR_ALPHA, TR_BETA, TR_PRED, TR_DELTA
MCBA_HD void tr_trial(double* S, double Delta) {
  const double BS[3] = {S[TR_BS0], S[TR_BS1], S[TR_BS2]}, gS[2] = {S[TR_GS0], 0.0};
  const double* Cm = S + TR_CM0;
  double pS[2];
  tr_solve_2d(BS, gS, Delta, pS);
  S[TR_PRED] = -(0.5 * (pS[0] * (BS[0] * pS[0] + BS[1] * pS[1]) + pS[1] * (BS[1] * pS[0] + BS[2] * pS[1])) + gS[0] * pS[0] +
                 gS[1] * pS[1]);
  S[TR_ALPHA] = Cm[0] * pS[0] + Cm[1] * pS[1];
  S[TR_BETA] = Cm[2] * pS[0] + Cm[3] * pS[1];
  S[TR_DELTA] = Delta;
}

// scipy common.py update_tr_radius
MCBA_HD void tr_update_radius(double& Delta, double actual, double predicted, double step_norm, bool bound_hit,
                              double& ratio) {
  if (predicted > 0) ratio = actual / predicted;
  else if (predicted == 0 && actual == 0) ratio = 1;
  else ratio = 0;
  if (ratio < 0.25) Delta = 0.25 * step_norm;
  else if (ratio > 0.75 && bound_hit) Delta *= 2.0;
}

// scipy common.py check_termination; -100 stands for None
MCBA_HD int tr_check_termination(double dF, double F, double dx_norm, double x_norm, double ratio, double ftol,
                                 double xtol) {
  const bool f_ok = dF < ftol * F && ratio > 0.25;
  const bool x_ok = dx_norm < xtol * (xtol + x_norm);
  if (f_ok && x_ok) return 4;
  if (f_ok) return 2;
  if (x_ok) return 3;
  return -100;
}

}  // namespace mcba
