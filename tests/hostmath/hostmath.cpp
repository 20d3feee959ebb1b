// TEST INFRASTRUCTURE -- serial CPU harness around the product's own device functions.
//
// Compiles multical_amd/csrc/mcba_view.h + mcba_math.h + mcba_lower.h with g++ (they are __host__ __device__ /
// plain C++) and runs them in plain loops, so that the formulas and index maps the HIP kernels use can be checked
// against the oracle on the GPU-less build box.  It mirrors the kernels' algebra (per-view S = V^T V, M = That^T S That,
// scatter through local_to_x) but none of their parallel structure.  NEVER loaded by multical_amd: the product has
// no CPU path.  Built by __graft_entry__.build() into tests/hostmath/_build/libmcba_hostmath.so.
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../multical_amd/csrc/mcba_lower.h"
#include "../../multical_amd/csrc/mcba_view.h"
#include "../../multical_amd/csrc/mcba_lsmr.h"
#include "../../multical_amd/csrc/mcba_trmath.h"

using namespace mcba;

namespace {
thread_local std::string g_err;

struct Host {
  HostProblem hp;
  Tables t{};
  std::vector<double> board_points, pose, cam, view;
  void init(const mcba_problem* p) {
    lower_problem(p, hp);
    const Dims& d = hp.d;
    board_points.assign((size_t)d.B * d.P * 3, 0.0);
    pose.assign((size_t)d.n_pose * POSE_STRIDE, 0.0);
    cam.assign((size_t)d.C * CAM_STRIDE, 0.0);
    view.assign((size_t)d.views() * d.view_stride(), 0.0);
    t.obs = hp.obs.data(); t.inlier = hp.inlier.data(); t.evalid = hp.evalid.data(); t.obs_index = hp.obs_index.data();
    t.view_count = hp.view_count.data(); t.board_off = hp.board_off.data(); t.full2act = hp.full2act.data();
    t.xfull = hp.xfull.data(); t.bwg = hp.bwg.data(); t.img_h = hp.img_h.data(); t.fix_aspect = hp.fix_aspect.data();
    t.board_points = board_points.data(); t.pose = pose.data(); t.cam = cam.data(); t.view = view.data();
  }
  std::vector<double> xint;   // ragged camera blocks: the caller's vector scattered into the padded internal layout
  void eval_tables(const double* x) {
    const Dims& d = hp.d;
    if (!hp.ext2int.empty()) {
      xint.assign((size_t)d.n, 0.0);
      for (int i = 0; i < hp.n_ext; ++i) xint[hp.ext2int[i]] = x[i];
      x = xint.data();
    }
    const int items = d.n_pose + d.C + d.B * d.P;
    for (int i = 0; i < items; ++i) prep_item(d, t, x, i);
    const int nv = d.views() * (d.motion == MOTION_ROLLING ? 2 : 1);
    for (int i = 0; i < nv; ++i) view_item(d, t, i);
  }
};

template <int ND, int FISH, bool ROLL>
void residuals_t(Host& h, double* r, double* err, uint8_t* valid) {
  const Dims& d = h.hp.d;
  for (int s = 0; s < d.slots(); ++s) {
    const int p = s % d.P, v = s / d.P;
    const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
    const double2 ob = h.t.obs[s];
    double uv[2], Xs[3], Xe[3], tr;
    slot_forward<ND, FISH, ROLL, false>(d, h.t, v, c, b, p, ob, uv, nullptr, nullptr, Xs, Xe, tr);
    const double ex = uv[0] - ob.x, ey = uv[1] - ob.y;
    const int idx = h.t.obs_index[s];
    if (r && idx >= 0) { r[2 * idx] = ex; r[2 * idx + 1] = ey; }
    if (err) {
      const size_t ri = (((size_t)c * d.F + f) * d.B + b) * d.P + p;
      const bool ok = h.t.evalid[s] != 0;
      err[ri] = ok ? std::sqrt(ex * ex + ey * ey) : 0.0;
      valid[ri] = ok;
    }
  }
}

template <int ND, int FISH, bool ROLL>
void jacobian_t(Host& h, int row_nnz, double* vals, int32_t* cols) {
  const Dims& d = h.hp.d;
  constexpr int DE = ROLL ? 12 : 6, KIA = 4 + ND, NV = DE + KIA + 1;
  for (int s = 0; s < d.slots(); ++s) {
    const int idx = h.t.obs_index[s];
    if (idx < 0) continue;
    const int p = s % d.P, v = s / d.P;
    const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
    Dims dl = d;
    dl.loss = 0;
    double vr[2 * NV], jp[6];
    point_rows<ND, FISH, ROLL, true>(dl, h.t, v, c, b, p, h.t.obs[s], vr, jp);
    double* o0 = vals + (size_t)(2 * idx) * row_nnz;
    double* o1 = o0 + row_nnz;
    int32_t* oc = cols + (size_t)idx * row_nnz;
    int pos = 0;
    const int order[4] = {0, d.NPB - 1, 1, 2};
    for (int oi = 0; oi < d.NPB; ++oi) {
      const int k = order[oi];
      if (local_to_x(d, f, c, b, 6 * k) < 0) continue;
      for (int jj = 0; jj < 6; ++jj) {
        double col[12];
        view_column(d, h.t, f, c, b, 6 * k + jj, col);
        double a0 = 0, a1 = 0;
        for (int a = 0; a < DE; ++a) { a0 += vr[a] * col[a]; a1 += vr[NV + a] * col[a]; }
        o0[pos] = a0; o1[pos] = a1; oc[pos] = local_to_x(d, f, c, b, 6 * k + jj);
        ++pos;
      }
    }
    if (d.off_cameras >= 0) {
      const int base = d.off_cameras + c * (5 + ND);
      for (int q = 0; q < 5 + ND; ++q) {
        const int lq = q < 4 ? q : q - 1;
        const bool skew = q == 4;
        o0[pos] = skew ? 0.0 : vr[DE + lq];
        o1[pos] = skew ? 0.0 : vr[NV + DE + lq];
        oc[pos] = base + q;
        ++pos;
      }
    }
    if (d.off_boards >= 0) {
      const int base = d.off_boards + 3 * (h.t.board_off[b] + p);
      for (int k = 0; k < 3; ++k) { o0[pos] = jp[k]; o1[pos] = jp[3 + k]; oc[pos] = base + k; ++pos; }
    }
  }
}

// The matrix-free Jacobian products of the lsmr mode, serially, with the device functions the kernels use (k_lsmr_jv / k_lsmr_jtu /
// k_lsmr_fused*): per view w = That (v restricted to the view's pose blocks) | v_K, per observation  J v = row . w + jp . v_point,
// and the adjoint  J^T u = That^T sum_p E_p^T u_p | sum_p K_p^T u_p | jp^T u  scattered through local_to_x.  Unscaled columns.
template <int ND, int FISH, bool ROLL, bool OPTK>
void lsmr_products_t(Host& h, const double* vin, const double* uin, double* jv, double* jtu) {
  const Dims& d = h.hp.d;
  constexpr int DE = ROLL ? 12 : 6, KI = OPTK ? 4 + ND : 0, NV = DE + KI + 1, NS = DE + KI;
  const int NPC = 6 * d.NPB;
  for (int i = 0; i < d.n; ++i) jtu[i] = 0.0;
  Dims dl = d;
  dl.loss = 0;
  for (int v = 0; v < d.views(); ++v) {
    const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
    double That[12][24], wl[NS], sums[NS];
    for (int j = 0; j < NPC; ++j) {
      double col[12];
      view_column(d, h.t, f, c, b, j, col);
      for (int a = 0; a < DE; ++a) That[a][j] = col[a];
    }
    for (int a = 0; a < DE; ++a) {
      double s = 0.0;
      for (int j = 0; j < NPC; ++j) {
        const int xi = local_to_x(d, f, c, b, j);
        if (xi >= 0) s += That[a][j] * vin[xi];
      }
      wl[a] = s;
    }
    for (int q = 0; q < KI; ++q) {
      const int xi = local_to_x(d, f, c, b, NPC + q);
      wl[DE + q] = xi >= 0 ? vin[xi] : 0.0;
    }
    for (int k = 0; k < NS; ++k) sums[k] = 0.0;
    for (int p = 0; p < d.P; ++p) {
      const size_t s = (size_t)v * d.P + p;
      const int idx = h.t.obs_index[s];
      if (idx < 0) continue;
      PointState<ND, ROLL> ps;
      point_state<ND, FISH, ROLL, false>(dl, h.t, v, c, b, p, h.t.obs[s], ps);
      double bterm[2] = {0.0, 0.0};
      const int gq = d.off_boards >= 0 ? d.off_boards + 3 * (h.t.board_off[b] + p) : -1;
      if (gq >= 0) {
        double w3[3];
        board_point_direction<ROLL>(h.t, v, ps.tr, vin[gq], vin[gq + 1], vin[gq + 2], w3);
        for (int a = 0; a < 2; ++a) bterm[a] = ps.rs[a] * (ps.A[3 * a] * w3[0] + ps.A[3 * a + 1] * w3[1] + ps.A[3 * a + 2] * w3[2]);
      }
      double uu[2] = {uin[2 * idx], uin[2 * idx + 1]};
      for (int a = 0; a < 2; ++a) {
        double row[NV];
        point_row<ND, ROLL, OPTK>(ps, a, row);
        double val = 0.0;
        for (int k = 0; k < NS; ++k) val += row[k] * wl[k];
        jv[2 * idx + a] = val + bterm[a];
        for (int k = 0; k < NS; ++k) sums[k] += row[k] * uu[a];
      }
      if (gq >= 0) {
        double q3[3], w3[3];
        for (int k = 0; k < 3; ++k) q3[k] = ps.rs[0] * uu[0] * ps.A[k] + ps.rs[1] * uu[1] * ps.A[3 + k];
        board_point_adjoint<ROLL>(h.t, v, ps.tr, q3, w3);
        for (int k = 0; k < 3; ++k) jtu[gq + k] += w3[k];
      }
    }
    for (int j = 0; j < NPC; ++j) {
      const int xi = local_to_x(d, f, c, b, j);
      if (xi < 0) continue;
      double s = 0.0;
      for (int a = 0; a < DE; ++a) s += That[a][j] * sums[a];
      jtu[xi] += s;
    }
    for (int q = 0; q < KI; ++q) {
      const int xi = local_to_x(d, f, c, b, NPC + q);
      if (xi >= 0) jtu[xi] += sums[DE + q];
    }
  }
}
template <int ND, int FISH, bool ROLL>
void lsmr_products_k(Host& h, const double* vin, const double* uin, double* jv, double* jtu) {
  if (h.hp.d.KI > 0) lsmr_products_t<ND, FISH, ROLL, true>(h, vin, uin, jv, jtu);
  else lsmr_products_t<ND, FISH, ROLL, false>(h, vin, uin, jv, jtu);
}

// per-view S = V^T V, M = That^T S That, scattered into dense H / g  (the kernels' algebra, serial)
template <int ND, int FISH, bool ROLL, bool OPTK>
void normal_t(Host& h, double* H, double* g, double* cost_out) {
  const Dims& d = h.hp.d;
  constexpr int DE = ROLL ? 12 : 6, KI = OPTK ? 4 + ND : 0, NV = DE + KI + 1;
  const int NPC = 6 * d.NPB, NL = NPC + KI, N1 = NL + 1, n = d.n;
  double cost = 0.0;
  std::vector<double> S(NV * NV), That(NV * N1), Y(NV * N1), M(N1 * N1);
  for (int v = 0; v < d.views(); ++v) {
    if (h.t.view_count[v] == 0) continue;
    const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
    std::fill(S.begin(), S.end(), 0.0);
    for (int p = 0; p < d.P; ++p) {
      const size_t s = (size_t)v * d.P + p;
      if (!h.t.inlier[s]) continue;
      double vr[2 * NV];
      cost += point_rows<ND, FISH, ROLL, OPTK>(d, h.t, v, c, b, p, h.t.obs[s], vr);
      for (int a = 0; a < 2; ++a)
        for (int i = 0; i < NV; ++i)
          for (int j = 0; j < NV; ++j) S[i * NV + j] += vr[a * NV + i] * vr[a * NV + j];
    }
    std::fill(That.begin(), That.end(), 0.0);
    for (int j = 0; j < NPC; ++j)   // the lane-uniform form used by k_tmat (the Jacobian above goes through view_column)
      view_that_column(d, global_pose_src(d, h.t), h.t.bwg, f, c, b, j, That.data(), N1);
    for (int q = 0; q < KI + 1; ++q) That[(DE + q) * N1 + NPC + q] = 1.0;
    for (int a = 0; a < NV; ++a)
      for (int j = 0; j < N1; ++j) {
        double sum = 0;
        for (int bb = 0; bb < NV; ++bb) sum += S[a * NV + bb] * That[bb * N1 + j];
        Y[a * N1 + j] = sum;
      }
    for (int i = 0; i < N1; ++i)
      for (int j = 0; j < N1; ++j) {
        double sum = 0;
        for (int a = 0; a < NV; ++a) sum += That[a * N1 + i] * Y[a * N1 + j];
        M[i * N1 + j] = sum;
      }
    for (int i = 0; i < NL; ++i) {
      const int gi = local_to_x(d, f, c, b, i);
      if (gi < 0) continue;
      g[gi] += M[i * N1 + NL];
      for (int j = 0; j < NL; ++j) {
        const int gj = local_to_x(d, f, c, b, j);
        if (gj >= 0) H[(size_t)gi * n + gj] += M[i * N1 + j];
      }
    }
  }
  *cost_out = 0.5 * cost;
}

#define DISPATCH_CAM(FN, ...)                                                                  \
  do {                                                                                         \
    const Dims& dd_ = h.hp.d;                                                                  \
    const bool roll_ = dd_.motion == MOTION_ROLLING;                                           \
    if (dd_.fisheye == 2) { if (roll_) FN<14, 2, true>(__VA_ARGS__); else FN<14, 2, false>(__VA_ARGS__); } \
    else if (dd_.fisheye) { if (roll_) FN<4, 1, true>(__VA_ARGS__); else FN<4, 1, false>(__VA_ARGS__); } \
    else switch (dd_.ND) {                                                                     \
      case 4: if (roll_) FN<4, 0, true>(__VA_ARGS__); else FN<4, 0, false>(__VA_ARGS__); break;   \
      case 5: if (roll_) FN<5, 0, true>(__VA_ARGS__); else FN<5, 0, false>(__VA_ARGS__); break;   \
      case 8: if (roll_) FN<8, 0, true>(__VA_ARGS__); else FN<8, 0, false>(__VA_ARGS__); break;   \
      case 12: if (roll_) FN<12, 0, true>(__VA_ARGS__); else FN<12, 0, false>(__VA_ARGS__); break; \
      default: if (roll_) FN<14, 0, true>(__VA_ARGS__); else FN<14, 0, false>(__VA_ARGS__); break; \
    }                                                                                          \
  } while (0)

template <int ND, int FISH, bool ROLL>
void normal_k(Host& h, double* H, double* g, double* cost) {
  if (h.hp.d.KI > 0) normal_t<ND, FISH, ROLL, true>(h, H, g, cost);
  else normal_t<ND, FISH, ROLL, false>(h, H, g, cost);
}

}  // namespace

#define HM_BEGIN try {
#define HM_END return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }

extern "C" {

const char* hm_last_error() { return g_err.c_str(); }

int32_t hm_sizes(const mcba_problem* p, int64_t* n_params, int64_t* n_residuals, int32_t* row_nnz) {
  HM_BEGIN
  Host h; h.init(p);
  const Dims& d = h.hp.d;
  *n_params = h.hp.ext2int.empty() ? d.n : h.hp.n_ext;   // length of the CALLER's vector (ragged camera blocks are padded inside)
  *n_residuals = 2 * h.hp.n_inliers;
  int nnz = 0;
  if (d.off_campose >= 0) nnz += 6;
  if (d.off_boardpose >= 0) nnz += 6;
  if (d.off_motion >= 0) nnz += d.motion == MOTION_STATIC ? 6 : 12;
  if (d.off_cameras >= 0) nnz += 5 + d.ND;
  if (d.off_boards >= 0) nnz += 3;
  *row_nnz = nnz;
  HM_END
}

int32_t hm_residuals(const mcba_problem* p, const double* x, double* r, double* err, uint8_t* valid) {
  HM_BEGIN
  Host h; h.init(p); h.eval_tables(x);
  DISPATCH_CAM(residuals_t, h, r, err, valid);
  HM_END
}

int32_t hm_jacobian(const mcba_problem* p, const double* x, int32_t row_nnz, double* vals, int32_t* cols) {
  HM_BEGIN
  Host h; h.init(p); h.eval_tables(x);
  DISPATCH_CAM(jacobian_t, h, row_nnz, vals, cols);
  if (!h.hp.int2ext.empty()) {   // internal (padded) column -> the caller's column, -1 for a coefficient the camera does not have
    const size_t nc = (size_t)h.hp.n_inliers * row_nnz;
    for (size_t i = 0; i < nc; ++i) cols[i] = cols[i] >= 0 ? h.hp.int2ext[cols[i]] : -1;
  }
  HM_END
}

int32_t hm_normal_equations(const mcba_problem* p, const double* x, int32_t loss, double f_scale, double* H, double* g,
                            double* cost) {
  HM_BEGIN
  Host h; h.init(p);
  h.hp.d.loss = loss; h.hp.d.f_scale = f_scale;
  h.eval_tables(x);
  const size_t n = h.hp.d.n;
  if (h.hp.ext2int.empty()) {
    std::memset(H, 0, n * n * sizeof(double));
    std::memset(g, 0, n * sizeof(double));
    DISPATCH_CAM(normal_k, h, H, g, cost);
  } else {   // ragged camera blocks: accumulate in the padded layout, hand out the caller's rows / columns
    std::vector<double> Hi(n * n, 0.0), gi(n, 0.0);
    double* Hp = Hi.data();
    double* gp = gi.data();
    DISPATCH_CAM(normal_k, h, Hp, gp, cost);
    const size_t ne = (size_t)h.hp.n_ext;
    for (size_t i = 0; i < ne; ++i) {
      g[i] = gi[h.hp.ext2int[i]];
      for (size_t j = 0; j < ne; ++j) H[i * ne + j] = Hi[(size_t)h.hp.ext2int[i] * n + h.hp.ext2int[j]];
    }
  }
  HM_END
}

// J(x) v and J(x)^T u through the matrix-free factorisation of the lsmr mode (uniform rigs: caller's layout = internal layout)
int32_t hm_lsmr_products(const mcba_problem* p, const double* x, const double* v, const double* u, double* jv, double* jtu) {
  HM_BEGIN
  Host h; h.init(p); h.eval_tables(x);
  if (!h.hp.ext2int.empty()) throw std::runtime_error("hm_lsmr_products: uniform camera blocks only");
  DISPATCH_CAM(lsmr_products_k, h, v, u, jv, jtu);
  HM_END
}

// the scalar trust-region algebra of the drivers (csrc/mcba_trmath.h): S is the TR_* block (hm_tr_nslots() doubles)
int32_t hm_tr_nslots() { return TR_NSLOTS; }
double hm_tr_reg_term(double q00, double gh2, double Delta, double floor) { return tr_reg_term(q00, gh2, Delta, floor); }
void hm_tr_subspace(double* S, int32_t explicit_forms, double Q01, double Q11) { tr_subspace(S, explicit_forms != 0, Q01, Q11); }
void hm_tr_trial(double* S, double Delta) { tr_trial(S, Delta); }
double hm_tr_update_radius(double Delta, double actual, double predicted, double step_norm, int32_t bound_hit, double* ratio) {
  tr_update_radius(Delta, actual, predicted, step_norm, bound_hit != 0, *ratio);
  return Delta;
}
int32_t hm_tr_check_termination(double dF, double F, double dx_norm, double x_norm, double ratio, double ftol, double xtol) {
  return tr_check_termination(dF, F, dx_norm, x_norm, ratio, ftol, xtol);
}
int32_t hm_tr_slot(const char* name) {
  struct { const char* n; int i; } const t[] = {{"gnorm", TR_GNORM}, {"gh2", TR_GH2}, {"xs2", TR_XS2}, {"q00", TR_Q00}, {"reg", TR_REG}, {"delta", TR_DELTA},
    {"d00", TR_D00}, {"d01", TR_D01}, {"d11", TR_D11}, {"alpha", TR_ALPHA}, {"beta", TR_BETA}, {"pred", TR_PRED}};
  for (const auto& e : t) if (std::strcmp(e.n, name) == 0) return e.i;
  return -1;
}

// the scalar recurrences of the device-resident LSMR solve (csrc/mcba_lsmr.h), one call each: the state block L has
// hm_lsmr_nslots() doubles
int32_t hm_lsmr_nslots() { return LS_NSLOTS; }
void hm_lsmr_state_init(double* L, double alpha, double beta, double damp, double normb, double maxiter) {
  lsmr_state_init(L, alpha, beta, damp, normb, maxiter);
}
void hm_lsmr_state_beta(double* L, double u2) { lsmr_state_beta(L, u2); }
void hm_lsmr_state_rotate(double* L, double v2) { lsmr_state_rotate(L, v2); }
int32_t hm_lsmr_state_test(const double* L, double x2) { return lsmr_state_test(L, x2); }
int64_t hm_lsmr_chunk_allowed(int32_t have_word, int32_t istop, int64_t done, int64_t chunk, int64_t cap) {
  return lsmr_chunk_allowed(have_word != 0, istop, done, chunk, cap);
}
int32_t hm_lsmr_slot(const char* name) {
  static const char* names[] = {"alpha", "beta", "inv_beta", "inv_alpha", "c_hbar", "c_x", "c_h", "skipv", "istop", "itn", "maxiter", "damp",
                                "normb", "zetabar", "alphabar", "rho", "rhobar", "cbar", "sbar", "betadd", "betad", "rhodold", "tautildeold",
                                "thetatilde", "zeta", "dd", "norma2", "maxrbar", "minrbar", "normr", "norma", "conda", "normar", "pending", "x2"};
  static_assert(sizeof(names) / sizeof(names[0]) == LS_NSLOTS, "slot names out of date");
  for (int i = 0; i < LS_NSLOTS; ++i)
    if (std::strcmp(names[i], name) == 0) return i;
  return -1;
}

}  // extern "C"
