// Scalar recurrences of scipy.sparse.linalg.lsmr (sparse/linalg/_isolve/lsmr.py:300-420), restated for ONE thread of the device:
// the state of an LSMR solve lives in a block of doubles in HBM (LS_*), the vector kernels of an iteration read their
// coefficients from it, and two small kernels per iteration (k_lsmr_scal_a / k_lsmr_scal_b, mcba_solver_kernels.h) advance it
// -- the host enqueues iterations ahead and only watches a progress word in pinned memory (mcba_api.hip: lsmr_solve).
// Every expression is evaluated in scipy's order WITHOUT contraction into fused multiply-adds (the iteration amplifies rounding
// differences: tests/test_oracle.py::test_scipys_lsmr_step_is_not_reproducible_beyond_rounding_noise).
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LSMR_HD __host__ __device__ inline
#else
#define LSMR_HD inline   // (tests/hostmath: the same source under g++, which never contracts on plain x86-64)
#endif

enum LsmrSlot : int {
  LS_ALPHA = 0, LS_BETA, LS_INV_BETA, LS_INV_ALPHA, LS_C_HBAR, LS_C_X, LS_C_H,   // read by the vector kernels
  LS_SKIPV,                    // beta == 0: no new v in this iteration (lsmr.py:318 "if beta > 0")
  LS_ISTOP, LS_ITN, LS_MAXITER, LS_DAMP, LS_NORMB,
  LS_ZETABAR, LS_ALPHABAR, LS_RHO, LS_RHOBAR, LS_CBAR, LS_SBAR, LS_BETADD, LS_BETAD, LS_RHODOLD, LS_TAUTILDEOLD, LS_THETATILDE,
  LS_ZETA, LS_DD, LS_NORMA2, LS_MAXRBAR, LS_MINRBAR, LS_NORMR, LS_NORMA, LS_CONDA, LS_NORMAR,
  LS_PENDING,                  // two-launch iteration: a new v_raw (and its |.|^2 partials) waits for its rotation + vector update
  LS_X2,                       // |x|^2 the last stopping tests saw (normx of scipy's return tuple: mcba_debug_lsmr_solve, the call trace)
  LS_NSLOTS
};

// progress word of a solve in pinned host memory: [call id : 24 | istop : 8 | completed iterations : 32]
LSMR_HD unsigned long long lsmr_progress_word(unsigned long long call, int istop, long long itn) {
  return ((call & 0xffffffull) << 40) | ((unsigned long long)(istop & 0xff) << 32) | (unsigned long long)(itn & 0xffffffffll);
}

// Frame-sharded solves enqueue LSMR iterations in CHUNKS, two chunks ahead of the progress word (mcba_api.hip: lsmr_solve): calls are
// numbered from 1, call j publishes the stopping tests of step j - 1, and chunk k + 1 (calls (k + 1) chunk + 1 .. (k + 2) chunk) goes out
// once the word of call k * chunk has arrived WITHOUT a stop.  Given the last word a rank has seen -- have_word, istop, done = the step
// the word reports -- this returns how many calls the rank may have enqueued in total.  The word is monotone and identical on all ranks,
// so whenever a rank looks, the answer converges to the same number: (floor(s / chunk) + 2) * chunk for a stop at step s, capped at
// maxiter + 1 calls (tests/test_host.py::test_chunked_enqueueing_keeps_ranks_matched).
LSMR_HD long long lsmr_chunk_allowed(bool have_word, int istop, long long done, long long chunk, long long cap) {
  long long allowed = 2 * chunk;                                              // chunks 0 and 1 need no word
  if (have_word) allowed = ((istop != 0 ? done : done + 1) / chunk + 2) * chunk;   // (done + 1 = the call whose word this is)
  return allowed < cap ? allowed : cap;
}

// lsmr.py:_sym_ortho
LSMR_HD void lsmr_sym_ortho(double a, double b, double& c, double& s, double& r) {
#pragma clang fp contract(off)
  const double sa = a > 0 ? 1.0 : (a < 0 ? -1.0 : 0.0), sb = b > 0 ? 1.0 : (b < 0 ? -1.0 : 0.0);
  if (b == 0) { c = sa; s = 0; r = fabs(a); }
  else if (a == 0) { c = 0; s = sb; r = fabs(b); }
  else if (fabs(b) > fabs(a)) { const double tau = a / b; s = sb / sqrt(1 + tau * tau); c = s * tau; r = b / s; }
  else { const double tau = b / a; c = sa / sqrt(1 + tau * tau); s = c * tau; r = a / c; }
}

// state in front of the first iteration (lsmr.py:262-298); alpha, beta, normb come from the two products of the prologue
LSMR_HD void lsmr_state_init(double* L, double alpha, double beta, double damp, double normb, double maxiter) {
#pragma clang fp contract(off)
  for (int i = 0; i < LS_NSLOTS; ++i) L[i] = 0.0;
  L[LS_ALPHA] = alpha; L[LS_BETA] = beta; L[LS_INV_ALPHA] = 1.0; L[LS_INV_BETA] = 1.0;
  L[LS_MAXITER] = maxiter; L[LS_DAMP] = damp; L[LS_NORMB] = normb;
  L[LS_ZETABAR] = alpha * beta; L[LS_ALPHABAR] = alpha; L[LS_RHO] = 1; L[LS_RHOBAR] = 1; L[LS_CBAR] = 1; L[LS_SBAR] = 0;
  L[LS_BETADD] = beta; L[LS_BETAD] = 0; L[LS_RHODOLD] = 1; L[LS_TAUTILDEOLD] = 0; L[LS_THETATILDE] = 0; L[LS_ZETA] = 0; L[LS_DD] = 0;
  L[LS_NORMA2] = alpha * alpha; L[LS_MAXRBAR] = 0; L[LS_MINRBAR] = 1e+100; L[LS_NORMA] = sqrt(alpha * alpha); L[LS_CONDA] = 1;
  L[LS_NORMR] = beta; L[LS_NORMAR] = alpha * beta;
}

// u = A v - alpha u is formed: beta = |u| (lsmr.py:316-318)
LSMR_HD void lsmr_state_beta(double* L, double u2) {
#pragma clang fp contract(off)
  const double beta = sqrt(u2);
  L[LS_BETA] = beta;
  L[LS_SKIPV] = beta > 0 ? 0.0 : 1.0;
  L[LS_INV_BETA] = beta > 0 ? 1.0 / beta : 1.0;
}

// v = A^T u - beta v is formed (v2 = its squared norm): alpha, the rotations and the coefficients of the vector update
// (lsmr.py:320-389)
LSMR_HD void lsmr_state_rotate(double* L, double v2) {
#pragma clang fp contract(off)
  const double itn = L[LS_ITN] + 1.0;
  L[LS_ITN] = itn;
  double alpha = L[LS_ALPHA], inv_alpha = 1.0;
  const double beta = L[LS_BETA], damp = L[LS_DAMP];
  if (L[LS_SKIPV] == 0.0) {
    alpha = sqrt(v2);
    if (alpha > 0) inv_alpha = 1.0 / alpha;
  }
  L[LS_ALPHA] = alpha;
  L[LS_INV_ALPHA] = inv_alpha;
  double alphabar = L[LS_ALPHABAR], rho = L[LS_RHO], rhobar = L[LS_RHOBAR], cbar = L[LS_CBAR], sbar = L[LS_SBAR];
  double zeta = L[LS_ZETA], zetabar = L[LS_ZETABAR], betadd = L[LS_BETADD], betad = L[LS_BETAD], rhodold = L[LS_RHODOLD];
  double tautildeold = L[LS_TAUTILDEOLD], thetatilde = L[LS_THETATILDE], dd = L[LS_DD], normA2 = L[LS_NORMA2];
  double maxrbar = L[LS_MAXRBAR], minrbar = L[LS_MINRBAR];
  double chat, shat, alphahat;
  lsmr_sym_ortho(alphabar, damp, chat, shat, alphahat);
  const double rhoold = rho;
  double c, sn;
  lsmr_sym_ortho(alphahat, beta, c, sn, rho);
  const double thetanew = sn * alpha;
  alphabar = c * alpha;
  const double rhobarold = rhobar, zetaold = zeta;
  const double thetabar = sbar * rho, rhotemp = cbar * rho;
  lsmr_sym_ortho(cbar * rho, thetanew, cbar, sbar, rhobar);
  zeta = cbar * zetabar;
  zetabar = -sbar * zetabar;
  L[LS_C_HBAR] = -(thetabar * rho / (rhoold * rhobarold));
  L[LS_C_X] = zeta / (rho * rhobar);
  L[LS_C_H] = -(thetanew / rho);
  const double betaacute = chat * betadd, betacheck = -shat * betadd;
  const double betahat = c * betaacute;
  betadd = -sn * betaacute;
  const double thetatildeold = thetatilde;
  double ctildeold, stildeold, rhotildeold;
  lsmr_sym_ortho(rhodold, thetabar, ctildeold, stildeold, rhotildeold);
  thetatilde = stildeold * rhobar;
  rhodold = ctildeold * rhobar;
  betad = -stildeold * betad + ctildeold * betahat;
  tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
  const double taud = (zeta - thetatilde * tautildeold) / rhodold;
  dd = dd + betacheck * betacheck;
  L[LS_NORMR] = sqrt(dd + (betad - taud) * (betad - taud) + betadd * betadd);
  normA2 = normA2 + beta * beta;
  L[LS_NORMA] = sqrt(normA2);
  normA2 = normA2 + alpha * alpha;
  maxrbar = fmax(maxrbar, rhobarold);
  if (itn > 1) minrbar = fmin(minrbar, rhobarold);
  L[LS_CONDA] = fmax(maxrbar, rhotemp) / fmin(minrbar, rhotemp);
  L[LS_NORMAR] = fabs(zetabar);
  L[LS_ALPHABAR] = alphabar; L[LS_RHO] = rho; L[LS_RHOBAR] = rhobar; L[LS_CBAR] = cbar; L[LS_SBAR] = sbar;
  L[LS_ZETA] = zeta; L[LS_ZETABAR] = zetabar; L[LS_BETADD] = betadd; L[LS_BETAD] = betad; L[LS_RHODOLD] = rhodold;
  L[LS_TAUTILDEOLD] = tautildeold; L[LS_THETATILDE] = thetatilde; L[LS_DD] = dd; L[LS_NORMA2] = normA2;
  L[LS_MAXRBAR] = maxrbar; L[LS_MINRBAR] = minrbar;
}

// x is updated (x2 = |x|^2): the stopping tests of the iteration (lsmr.py:391-420, atol = btol = 1e-6, conlim = 1e8: the call of
// scipy/optimize/_lsq/trf.py:481); returns scipy's istop, 0 = carry on
LSMR_HD int lsmr_state_test(const double* L, double x2) {
#pragma clang fp contract(off)
  const double atol = 1e-6, btol = 1e-6, ctol = 1 / 1e8;
  const double normx = sqrt(x2), normr = L[LS_NORMR], normA = L[LS_NORMA], normb = L[LS_NORMB], normar = L[LS_NORMAR];
  const double test1 = normr / normb;
  const double test2 = (normA * normr) != 0 ? normar / (normA * normr) : INFINITY;
  const double test3 = 1 / L[LS_CONDA];
  const double t1 = test1 / (1 + normA * normx / normb);
  const double rtol = btol + atol * normA * normx / normb;
  int istop = 0;
  if (L[LS_ITN] >= L[LS_MAXITER]) istop = 7;
  if (1 + test3 <= 1) istop = 6;
  if (1 + test2 <= 1) istop = 5;
  if (1 + t1 <= 1) istop = 4;
  if (test3 <= ctol) istop = 3;
  if (test2 <= atol) istop = 2;
  if (test1 <= rtol) istop = 1;
  return istop;
}
