// mcba_view.h -- per-slot / per-view device functions of the hot path (no thread indexing, no LDS):
//   prep_item / view_item    x -> pose, camera, board-point and per-view chain tables
//   slot_forward             forward model of one observation slot
//   view_column / local_to_x pose-block structure of a view (column of That, x index of a local parameter)
//   point_rows               the 2 x NV row pair V = [E | K | r] of one observation
// They are __host__ __device__ so that tests/hostmath can run the SAME source serially with g++ on the CPU-only build
// box and compare it with the oracle; the product only ever calls them from kernels.
#pragma once
#include "mcba_device.h"

namespace mcba {

MCBA_HD double param_value(const Tables& t, const double* x, int j) {
  const int a = t.full2act[j];
  return a >= 0 ? x[a] : t.xfull[j];
}

// the same value without the dependent read of the full -> active index map: the caller names the block (offsets inside
// the active vector x and inside the full vector come from Dims) -- one memory round trip less on the latency-bound
// table kernels
// The active vector may also be one that is not stored anywhere yet: StepX evaluates entry i of the trial point
// x + D (alpha u0 + beta u1) exactly as k_vec_step stores it (same two fused operations), so that the table entries of a
// trial point can be formed by the kernel that produces the point.
struct PlainX {
  const double* x;
  MCBA_HD double operator()(int i) const { return x[i]; }
};
MCBA_HD double step_direction(double alpha, double u0, double beta, double u1) { return fma(alpha, u0, beta * u1); }
MCBA_HD double step_point(double x, double dsc, double p) { return fma(dsc, p, x); }
struct StepX {
  const double* x; const double* dsc; const double* u0; const double* u1;
  double alpha, beta;
  MCBA_HD double operator()(int i) const { return step_point(x[i], dsc[i], step_direction(alpha, u0[i], beta, u1[i])); }
};
template <class XV>
MCBA_HD double block_value(const Tables& t, const XV& xv, int off_active, int off_full, int i) {
  return off_active >= 0 ? xv(off_active + i) : t.xfull[off_full + i];
}
MCBA_HD double block_value(const Tables& t, const double* x, int off_active, int off_full, int i) {
  return off_active >= 0 ? x[off_active + i] : t.xfull[off_full + i];
}

MCBA_HD int tri_index(int i, int j, int N1) {   // packed upper triangle, i <= j
  return i * N1 - (i * (i - 1)) / 2 + (j - i);
}

// ---------------------------------------------------------------------------------------------------------------
// forward model of one table slot (motion/static_frames.py:16-25, motion/rolling_frames.py:15-41)
// ---------------------------------------------------------------------------------------------------------------
template <int ND, int FISH, bool ROLL, bool JAC>
MCBA_HD void slot_forward(const Dims& d, const Tables& t, int v, int c, int b, int p, double2 ob,
                                             double* uv, double* A, double* Kc, double* Xs, double* Xe, double& tr,
                                             const double* Xpre = nullptr /* prefetched board point */,
                                             const double* Vpre = nullptr /* chain matrices of the view (registers / LDS) */,
                                             const double* camp = nullptr /* the camera's parameter block (x / registers) */,
                                             const double* extp = nullptr /* tail of the camera entry, CAM_TILT onwards */,
                                             const double* trpre = nullptr /* rolling shutter: the observation's scan time, precomputed */) {
  const double* X = Xpre != nullptr ? Xpre : t.board_points + 3 * (size_t)(b * d.P + p);
  const double* V = Vpre != nullptr ? Vpre : t.view + (size_t)v * (VIEW_STRIDE * (ROLL ? 2 : 1));
  const double* cam = t.cam + (size_t)c * CAM_STRIDE;
  const double* ext = extp != nullptr ? extp : cam + CAM_TILT;
  if (camp != nullptr) cam = camp;
  const double bx = X[0], by = X[1], bz = X[2];
  double Xc[3];
  for (int i = 0; i < 3; ++i) Xs[i] = V[3 * i] * bx + V[3 * i + 1] * by + V[3 * i + 2] * bz + V[9 + i];
  if constexpr (ROLL) {
    const double* W = V + VIEW_STRIDE;
    for (int i = 0; i < 3; ++i) Xe[i] = W[3 * i] * bx + W[3 * i + 1] * by + W[3 * i + 2] * bz + W[9 + i];
    tr = trpre != nullptr ? *trpre : ob.y / ext[CAM_HEIGHT - CAM_TILT];   // rolling_frames.py:15-19 (observed row)
    for (int i = 0; i < 3; ++i) Xc[i] = Xs[i] * (1.0 - tr) + Xe[i] * tr;   // interpolate.py:6-8
  } else {
    tr = 0.0;
    for (int i = 0; i < 3; ++i) Xc[i] = Xs[i];
  }
  project_point<ND, FISH, JAC>(cam, ext, Xc, uv, A, Kc);
}

// ---------------------------------------------------------------------------------------------------------------
// pose-block structure of a view: column j of That (DE rows) and the x index of every local parameter
// ---------------------------------------------------------------------------------------------------------------
// column j of That from explicit pose entries: Pc camera, Pb board, Pm0 / Pm1 the motion entries (static: Pm0 = frame;
// rolling: start / end pose of the frame; hand-eye: world_wrt_base / gripper_wrt_camera), Bf = base_wrt_gripper[f] (R | t)
MCBA_HD void view_column_p(const Dims& d, const double* Pc, const double* Pb, const double* Pm0, const double* Pm1,
                           const double* Bf, int j, double* col, int stride = 1) {
  const int k = j / 6, jj = j % 6;
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double* Rc = Pc + POSE_R;
  const double* tc = Pc + POSE_T;
  if (d.motion == MOTION_STATIC) {
    const double* Pf = Pm0;
    if (k == 0) {
      view_pose_column(I3, Pc + POSE_L, tc, jj, col, stride);
    } else {
      double R1[9], t1[3];
      se3_mul(Rc, tc, Pf + POSE_R, Pf + POSE_T, R1, t1);          // camera . frame
      if (k == 1) {
        view_pose_column(Rc, Pf + POSE_L, t1, jj, col, stride);
      } else {
        double o[3], v3[3];
        mat3_vec(R1, Pb + POSE_T, v3);
        for (int i = 0; i < 3; ++i) o[i] = t1[i] + v3[i];
        view_pose_column(R1, Pb + POSE_L, o, jj, col, stride);
      }
    }
  } else if (d.motion == MOTION_ROLLING) {
    for (int i = 0; i < 12; ++i) col[i * stride] = 0.0;
    if (k == 0) {
      view_pose_column(I3, Pc + POSE_L, tc, jj, col, stride);
      for (int i = 0; i < 6; ++i) col[(6 + i) * stride] = col[i * stride];
    } else {
      for (int ch = 0; ch < 2; ++ch) {
        if ((k == 1 && ch == 1) || (k == 2 && ch == 0)) continue;
        const double* Pf = ch == 0 ? Pm0 : Pm1;
        double R1[9], t1[3];
        se3_mul(Rc, tc, Pf + POSE_R, Pf + POSE_T, R1, t1);
        if (k == 3) {
          double o[3], v3[3];
          mat3_vec(R1, Pb + POSE_T, v3);
          for (int i = 0; i < 3; ++i) o[i] = t1[i] + v3[i];
          view_pose_column(R1, Pb + POSE_L, o, jj, col + 6 * ch * stride, stride);
        } else {
          view_pose_column(Rc, Pf + POSE_L, t1, jj, col + 6 * ch * stride, stride);
        }
      }
    }
  } else {  // hand-eye: chain camera . G . B_f . Wb . board ; local blocks: cam | wb | gc | board
    const double* Wb = Pm0;
    const double* G = Pm1;
    if (k == 0) {
      view_pose_column(I3, Pc + POSE_L, tc, jj, col, stride);
    } else {
      double R1[9], t1[3];
      se3_mul(Rc, tc, G + POSE_R, G + POSE_T, R1, t1);            // camera . G
      if (k == 2) {
        view_pose_column(Rc, G + POSE_L, t1, jj, col, stride);
      } else {
        double R2[9], t2[3], R3[9], t3[3];
        se3_mul(R1, t1, Bf, Bf + 9, R2, t2);                       // . B_f
        se3_mul(R2, t2, Wb + POSE_R, Wb + POSE_T, R3, t3);         // . Wb
        if (k == 1) {
          view_pose_column(R2, Wb + POSE_L, t3, jj, col, stride);
        } else {
          double o[3], v3[3];
          mat3_vec(R3, Pb + POSE_T, v3);
          for (int i = 0; i < 3; ++i) o[i] = t3[i] + v3[i];
          view_pose_column(R3, Pb + POSE_L, o, jj, col, stride);
        }
      }
    }
  }
}
MCBA_HD void view_column(const Dims& d, const Tables& t, int f, int c, int b, int j, double* col, int stride = 1) {
  const double* Pc = t.pose + (size_t)(d.pose_cam + c) * POSE_STRIDE;
  const double* Pb = t.pose + (size_t)(d.pose_board + b) * POSE_STRIDE;
  const double* Pm0;
  const double* Pm1;
  if (d.motion == MOTION_STATIC) {
    Pm0 = Pm1 = t.pose + (size_t)(d.pose_motion + f) * POSE_STRIDE;
  } else if (d.motion == MOTION_ROLLING) {
    Pm0 = t.pose + (size_t)(d.pose_motion + f) * POSE_STRIDE;
    Pm1 = t.pose + (size_t)(d.pose_motion + d.F + f) * POSE_STRIDE;
  } else {
    Pm0 = t.pose + (size_t)(d.pose_motion + 0) * POSE_STRIDE;
    Pm1 = t.pose + (size_t)(d.pose_motion + 1) * POSE_STRIDE;
  }
  view_column_p(d, Pc, Pb, Pm0, Pm1, t.bwg + 12 * (size_t)f, j, col, stride);
}

// Lane-uniform construction of the That columns and the chain matrices of one view from its pose entries (static and
// rolling-shutter motion): the chain products camera . frame (. board) are formed ONCE by every lane (same instruction
// stream, no divergence), then lane j < 6 NPB picks the prefix rotation / left Jacobian / origin of ITS pose block with
// selects and evaluates one column (view_pose_column written branch-free).  Equivalent to view_column_p, whose four block
// cases a wavefront had to execute one after the other.  Tm[a * NPC + j] receives column j, Vm the chain matrices.
template <bool ROLL, bool CHAINS = true>
MCBA_HD void fused_view_tables(const double* Pc, const double* Pm0, const double* Pm1, const double* Pb, int j,
                               double* Tm, double* Vm, int stride = ROLL ? 24 : 18) {
  constexpr int NCH = ROLL ? 2 : 1, NPB = ROLL ? 4 : 3, NPC = 6 * NPB;
  const double* Rc = Pc + POSE_R;
  const double* tc = Pc + POSE_T;
  const double* Lc = Pc + POSE_L;
  const double* Rb = Pb + POSE_R;
  const double* tb = Pb + POSE_T;
  const double* Lb = Pb + POSE_L;
  const int k = j / 6, jj = j % 6;
  const bool rotcol = jj < 3;
  const int ju = rotcol ? jj : jj - 3;
  const double e0 = ju == 0 ? 1.0 : 0.0, e1 = ju == 1 ? 1.0 : 0.0, e2 = ju == 2 ? 1.0 : 0.0;
  for (int ch = 0; ch < NCH; ++ch) {
    const double* Pf = ch == 0 ? Pm0 : Pm1;
    double R1[9], t1[3], o[3], v3[3], R2[9];
    se3_mul(Rc, tc, Pf + POSE_R, Pf + POSE_T, R1, t1);            // camera . frame
    mat3_vec(R1, tb, v3);
    for (int i = 0; i < 3; ++i) o[i] = t1[i] + v3[i];             // origin of camera . frame . board
    mat3_mul(R1, Rb, R2);
    if constexpr (CHAINS) {
      if (j == NPC + ch) {                                        // the chain matrix board -> camera of this chain
        for (int i = 0; i < 9; ++i) Vm[ch * VIEW_STRIDE + i] = R2[i];
        for (int i = 0; i < 3; ++i) Vm[ch * VIEW_STRIDE + 9 + i] = o[i];
      }
    }
    // pose block of this lane: 0 camera (identity prefix), NPB - 1 board (prefix camera . frame), between: the frame pose
    // of chain `ch` (prefix camera); rolling shutter: block 1 touches chain 0 only, block 2 chain 1 only
    const bool is_cam = k == 0, is_board = k == NPB - 1;
    const bool active = is_cam || is_board || !ROLL || k == 1 + ch;
    double u[3];                                                  // column ju of L_k (rotation columns) or a unit vector
    for (int i = 0; i < 3; ++i) {
      const double lc = Lc[3 * i] * e0 + Lc[3 * i + 1] * e1 + Lc[3 * i + 2] * e2;
      const double lf = Pf[POSE_L + 3 * i] * e0 + Pf[POSE_L + 3 * i + 1] * e1 + Pf[POSE_L + 3 * i + 2] * e2;
      const double lb = Lb[3 * i] * e0 + Lb[3 * i + 1] * e1 + Lb[3 * i + 2] * e2;
      const double unit = i == ju ? 1.0 : 0.0;
      u[i] = rotcol ? (is_cam ? lc : (is_board ? lb : lf)) : unit;
    }
    double tv[3], ov[3];
    for (int i = 0; i < 3; ++i) {
      const double r0 = is_cam ? (i == 0 ? 1.0 : 0.0) : (is_board ? R1[3 * i] : Rc[3 * i]);
      const double r1 = is_cam ? (i == 1 ? 1.0 : 0.0) : (is_board ? R1[3 * i + 1] : Rc[3 * i + 1]);
      const double r2 = is_cam ? (i == 2 ? 1.0 : 0.0) : (is_board ? R1[3 * i + 2] : Rc[3 * i + 2]);
      tv[i] = r0 * u[0] + r1 * u[1] + r2 * u[2];                  // R_pre u
      ov[i] = is_cam ? tc[i] : (is_board ? o[i] : t1[i]);         // o_k
    }
    const double c0 = ov[1] * tv[2] - ov[2] * tv[1], c1 = ov[2] * tv[0] - ov[0] * tv[2], c2 = ov[0] * tv[1] - ov[1] * tv[0];
    const double w = active ? 1.0 : 0.0;
    if (j < NPC) {
      double* col = Tm + (size_t)(6 * ch) * stride + j;
      col[0 * stride] = rotcol ? w * tv[0] : 0.0;
      col[1 * stride] = rotcol ? w * tv[1] : 0.0;
      col[2 * stride] = rotcol ? w * tv[2] : 0.0;
      col[3 * stride] = w * (rotcol ? c0 : tv[0]);
      col[4 * stride] = w * (rotcol ? c1 : tv[1]);
      col[5 * stride] = w * (rotcol ? c2 : tv[2]);
    }
  }
}

// chain matrix board -> camera from explicit pose entries (Pm = the motion entry of the wanted chain; hand-eye: Pm0 =
// world_wrt_base, Pm1 = gripper_wrt_camera): out[12] = R | t
MCBA_HD void view_chain_p(const Dims& d, const double* Pc, const double* Pb, const double* Pm0, const double* Pm1,
                          const double* Bf, double* out) {
  double R1[9], t1[3], R2[9], t2[3];
  if (d.motion == MOTION_HAND_EYE) {
    const double* Wb = Pm0;
    const double* G = Pm1;
    se3_mul(Pc + POSE_R, Pc + POSE_T, G + POSE_R, G + POSE_T, R1, t1);
    se3_mul(R1, t1, Bf, Bf + 9, R2, t2);
    se3_mul(R2, t2, Wb + POSE_R, Wb + POSE_T, R1, t1);
    se3_mul(R1, t1, Pb + POSE_R, Pb + POSE_T, R2, t2);
  } else {
    se3_mul(Pc + POSE_R, Pc + POSE_T, Pm0 + POSE_R, Pm0 + POSE_T, R1, t1);
    se3_mul(R1, t1, Pb + POSE_R, Pb + POSE_T, R2, t2);
  }
  for (int k = 0; k < 9; ++k) out[k] = R2[k];
  for (int k = 0; k < 3; ++k) out[9 + k] = t2[k];
}

// Where the pose entries (R, t, L) of a view come from: the global pose table written by k_prep, or a workgroup-local
// table in LDS that k_tmat fills straight from x (the same pose_entry function: identical bits).  Motion entries are
// addressed as mot + (ch * chain + (f - f0)) * POSE_STRIDE (rolling shutter: ch = 0 start, 1 end); hand-eye: mot[0] =
// world_wrt_base, mot[1] = gripper_wrt_camera.
struct PoseSrc {
  const double* cam;     // entry of camera c at cam + c * POSE_STRIDE
  const double* board;   // entry of board b
  const double* mot;     // motion entries
  int chain;             // entries per chain
  int f0;                // global frame of motion entry 0
};
MCBA_HD PoseSrc global_pose_src(const Dims& d, const Tables& t) {
  PoseSrc s;
  s.cam = t.pose + (size_t)d.pose_cam * POSE_STRIDE;
  s.board = t.pose + (size_t)d.pose_board * POSE_STRIDE;
  s.mot = t.pose + (size_t)d.pose_motion * POSE_STRIDE;
  s.chain = d.F;
  s.f0 = 0;
  return s;
}

// The same construction in two steps, for k_tmat: the chain prefix of a view is formed ONCE per (view, chain) --
//   pre[24] = R1 (camera . frame) | t1 | R2 (camera . frame . board) | o      (R2 | o IS the view-table chain matrix)
// -- and the columns of That pick their block's prefix rotation / left Jacobian / origin from it (static and rolling).
constexpr int PRE_STRIDE = 24;
MCBA_HD void view_prefix(const double* Pc, const double* Pf, const double* Pb, double* pre) {
  se3_mul(Pc + POSE_R, Pc + POSE_T, Pf + POSE_R, Pf + POSE_T, pre, pre + 9);
  se3_mul(pre, pre + 9, Pb + POSE_R, Pb + POSE_T, pre + 12, pre + 21);
}
template <bool ROLL>
MCBA_HD void that_column_from_prefix(const double* Pc, const double* Pm0, const double* Pm1, const double* Pb,
                                     const double* pre /*[NCH][PRE_STRIDE]*/, int j, double* Tm, int stride) {
  constexpr int NCH = ROLL ? 2 : 1, NPB = ROLL ? 4 : 3;
  const double* Rc = Pc + POSE_R;
  const double* tc = Pc + POSE_T;
  const int k = j / 6, jj = j % 6;
  const bool rotcol = jj < 3;
  const int ju = rotcol ? jj : jj - 3;
  const bool is_cam = k == 0, is_board = k == NPB - 1;
  for (int ch = 0; ch < NCH; ++ch) {
    const double* Pf = ch == 0 ? Pm0 : Pm1;
    const double* R1 = pre + ch * PRE_STRIDE;
    const double* t1 = R1 + 9;
    const double* o = R1 + 21;
    const bool active = is_cam || is_board || !ROLL || k == 1 + ch;
    const double* L = is_cam ? Pc + POSE_L : (is_board ? Pb + POSE_L : Pf + POSE_L);   // left Jacobian of the lane's block
    double u[3];                                                  // column ju of L (rotation columns) or a unit vector
    for (int i = 0; i < 3; ++i) u[i] = rotcol ? L[3 * i + ju] : (i == ju ? 1.0 : 0.0);
    double tv[3], ov[3];
    for (int i = 0; i < 3; ++i) {
      const double r0 = is_cam ? (i == 0 ? 1.0 : 0.0) : (is_board ? R1[3 * i] : Rc[3 * i]);
      const double r1 = is_cam ? (i == 1 ? 1.0 : 0.0) : (is_board ? R1[3 * i + 1] : Rc[3 * i + 1]);
      const double r2 = is_cam ? (i == 2 ? 1.0 : 0.0) : (is_board ? R1[3 * i + 2] : Rc[3 * i + 2]);
      tv[i] = r0 * u[0] + r1 * u[1] + r2 * u[2];                  // R_pre u
      ov[i] = is_cam ? tc[i] : (is_board ? o[i] : t1[i]);         // o_k
    }
    const double c0 = ov[1] * tv[2] - ov[2] * tv[1], c1 = ov[2] * tv[0] - ov[0] * tv[2], c2 = ov[0] * tv[1] - ov[1] * tv[0];
    const double w = active ? 1.0 : 0.0;
    double* col = Tm + (size_t)(6 * ch) * stride + j;
    col[0 * stride] = rotcol ? w * tv[0] : 0.0;
    col[1 * stride] = rotcol ? w * tv[1] : 0.0;
    col[2 * stride] = rotcol ? w * tv[2] : 0.0;
    col[3 * stride] = w * (rotcol ? c0 : tv[0]);
    col[4 * stride] = w * (rotcol ? c1 : tv[1]);
    col[5 * stride] = w * (rotcol ? c2 : tv[2]);
  }
}

// column j < 6 NPB of That of view (f, c, b) exactly as k_tmat forms it (which keeps the chain prefixes of its views in
// LDS and calls the two steps itself): Tm[a * stride + j], a < DE
MCBA_HD void view_that_column(const Dims& d, const PoseSrc& ps, const double* bwg, int f, int c, int b, int j, double* Tm,
                              int stride) {
  const double* Pc = ps.cam + (size_t)c * POSE_STRIDE;
  const double* Pb = ps.board + (size_t)b * POSE_STRIDE;
  if (d.motion == MOTION_HAND_EYE) {
    view_column_p(d, Pc, Pb, ps.mot, ps.mot + POSE_STRIDE, bwg + 12 * (size_t)f, j, Tm + j, stride);
    return;
  }
  const double* Pm0 = ps.mot + (size_t)(f - ps.f0) * POSE_STRIDE;
  const double* Pm1 = Pm0 + (size_t)ps.chain * POSE_STRIDE;
  double pre[2 * PRE_STRIDE];
  view_prefix(Pc, Pm0, Pb, pre);
  if (d.motion == MOTION_ROLLING) view_prefix(Pc, Pm1, Pb, pre + PRE_STRIDE);
  if (d.motion == MOTION_ROLLING) that_column_from_prefix<true>(Pc, Pm0, Pm1, Pb, pre, j, Tm, stride);
  else that_column_from_prefix<false>(Pc, Pm0, Pm0, Pb, pre, j, Tm, stride);
}

// x index of local parameter i of view (f, c, b); -1 when its block is not optimised (or i is the residual column)
MCBA_HD int local_to_x(const Dims& d, int f, int c, int b, int i) {
  const int npose = 6 * d.NPB;
  if (i < npose) {
    const int k = i / 6, jj = i % 6;
    if (k == 0) return d.off_campose < 0 ? -1 : d.off_campose + 6 * c + jj;
    if (k == d.NPB - 1) return d.off_boardpose < 0 ? -1 : d.off_boardpose + 6 * b + jj;
    if (d.off_motion < 0) return -1;
    if (d.motion == MOTION_STATIC) return d.off_motion + 6 * f + jj;
    if (d.motion == MOTION_ROLLING) return d.off_motion + (k == 2 ? 6 * d.F : 0) + 6 * f + jj;
    return d.off_motion + 6 * (k - 1) + jj;
  }
  const int q = i - npose;
  if (q >= d.KI || d.off_cameras < 0) return -1;
  if (d.cam_kmask != nullptr && ((d.cam_kmask[c] >> q) & 1u)) return -1;   // coefficient this camera's model does not have
  return d.off_cameras + c * (5 + d.ND) + (q < 4 ? q : q + 1);   // skip the skew slot (camera.py:153)
}

// Layout of the per-view partials of J_h^T uhat in the two-launch LSMR iteration (k_lsmr_fused2 writes, k_lsmr_gather3 sums): TRANSPOSED
// so that the views an x entry sums over are ONE contiguous run (the [view][local] layout of k_lsmr_jtu made every addend a cache
// line of its own: 4 000 lines for a board-pose entry at the north-star rig).  Segments, v = (fl C + c) B + b frame-major:
//   camera c, entry l_c in [0, 6 + KI) (pose | intrinsics):  [(c (6 + KI) + l_c) Fl B + fl B + b]
//   board b, entry l_b in [0, 6):                            [(b 6 + l_b) Fl C + fl C + c]
//   frame fl, entry e in [0, DFm) (static 6, rolling 12):    [(fl DFm + e) C B + c B + b]      hand-eye entry q in [0, 12): [q views + v]
// local = index in the view's parameter vector: [0, 6) camera pose | [6, NPC - 6) motion | [NPC - 6, NPC) board pose | [NPC, NPC + KI)
MCBA_HD size_t lsmr_part_cam(const Dims& d, int c, int lc) { return (size_t)(c * (6 + d.KI) + lc) * ((size_t)d.Fl * d.B); }
MCBA_HD size_t lsmr_part_board(const Dims& d, int b, int lb) {
  return (size_t)d.C * (6 + d.KI) * d.Fl * d.B + (size_t)(b * 6 + lb) * ((size_t)d.Fl * d.C);
}
MCBA_HD size_t lsmr_part_motion(const Dims& d) { return (size_t)d.views() * (6 + d.KI) + (size_t)d.views() * 6; }
MCBA_HD size_t lsmr_part_index(const Dims& d, int v, int local) {
  const int npc = 6 * d.NPB, b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C);
  if (local < 6) return lsmr_part_cam(d, c, local) + (size_t)fl * d.B + b;
  if (local >= npc) return lsmr_part_cam(d, c, 6 + local - npc) + (size_t)fl * d.B + b;
  if (local >= npc - 6) return lsmr_part_board(d, b, local - (npc - 6)) + (size_t)fl * d.C + c;
  const int e = local - 6, DFm = npc - 12;
  if (d.motion == MOTION_HAND_EYE) return lsmr_part_motion(d) + (size_t)e * d.views() + v;
  return lsmr_part_motion(d) + ((size_t)fl * DFm + e) * ((size_t)d.C * d.B) + (size_t)c * d.B + b;
}

// is local parameter i an ELIMINATED per-frame parameter?
MCBA_HD bool local_is_frame(const Dims& d, int i) {
  if (d.DF == 0) return false;
  return i >= 6 && i < 6 + d.DF;
}

// ---------------------------------------------------------------------------------------------------------------
// per-point row pair V = [E | K | r] (scaled for the robust loss), in two phases:
//   point_state   forward model + analytic derivatives of one observation (everything both rows share)
//   point_row     row a (0 = u, 1 = v) of V from the state
// k_linearize builds, consumes and stages one row at a time (the registers of the u-row are free before the v-row is
// formed); point_rows = state + both rows is what every other caller uses -- the arithmetic is the same.
// ---------------------------------------------------------------------------------------------------------------
template <int ND, bool ROLL>
struct PointState {
  double A[6];              // d(u,v)/d X_cam
  double Kc[2 * (4 + ND)];  // d(u,v)/d intrinsics (skew column omitted)
  double Xs[3], Xe[3];      // camera-frame point of the start / end chain
  double tr;                // scan time of the observed row (rolling shutter)
  double e[2];              // reprojection residual
  double rs[2], fs[2];      // robust-loss row scale / residual scale
};

// returns rho0_u + rho0_v
// ROBUST = false compiles the linear loss in (no loss switch, no row scaling: the hot kernel's default instantiation)
template <int ND, int FISH, bool ROLL, bool ROBUST = true>
MCBA_HD double point_state(const Dims& d, const Tables& t, int v, int c, int b, int p, double2 ob,
                           PointState<ND, ROLL>& st, const double* Xpre = nullptr, const double* Vpre = nullptr,
                           const double* camp = nullptr, const double* extp = nullptr, const double* trpre = nullptr) {
  double uv[2];
  slot_forward<ND, FISH, ROLL, true>(d, t, v, c, b, p, ob, uv, st.A, st.Kc, st.Xs, st.Xe, st.tr, Xpre, Vpre, camp, extp, trpre);
  st.e[0] = uv[0] - ob.x;
  st.e[1] = uv[1] - ob.y;
  double rho = 0.0;
  const int loss = ROBUST ? d.loss : 0;
  rho += robust_loss(loss, d.f_scale, st.e[0], &st.rs[0], &st.fs[0]);
  rho += robust_loss(loss, d.f_scale, st.e[1], &st.rs[1], &st.fs[1]);
  return rho;
}

template <int ND, bool ROLL, bool OPTK>
MCBA_HD void point_row(const PointState<ND, ROLL>& st, int a, double* row /*[NV]*/) {
  constexpr int DE = ROLL ? 12 : 6, KI = OPTK ? 4 + ND : 0, NV = DE + KI + 1, KIA = 4 + ND;
  const double* ar = st.A + 3 * a;
  if constexpr (ROLL) {
    double Es[6], Ee[6];
    base_row(ar, st.Xs, Es);
    base_row(ar, st.Xe, Ee);
    for (int i = 0; i < 6; ++i) {
      row[i] = st.rs[a] * (1.0 - st.tr) * Es[i];
      row[6 + i] = st.rs[a] * st.tr * Ee[i];
    }
  } else {
    double E[6];
    base_row(ar, st.Xs, E);
    for (int i = 0; i < 6; ++i) row[i] = st.rs[a] * E[i];
  }
  if constexpr (OPTK) {
    for (int i = 0; i < KI; ++i) row[DE + i] = st.rs[a] * st.Kc[a * KIA + i];
  }
  row[NV - 1] = st.e[a] * st.fs[a];
}

template <int ND, int FISH, bool ROLL, bool OPTK>
MCBA_HD double point_rows(const Dims& d, const Tables& t, int v, int c, int b, int p, double2 ob,
                                             double* vr /*[2][NV]*/, double* jp = nullptr /*[2][3]: d r / d X_board*/,
                                             const double* Xpre = nullptr) {
  constexpr int DE = ROLL ? 12 : 6, KI = OPTK ? 4 + ND : 0, NV = DE + KI + 1;
  PointState<ND, ROLL> st;
  const double rho = point_state<ND, FISH, ROLL>(d, t, v, c, b, p, ob, st, Xpre);
  point_row<ND, ROLL, OPTK>(st, 0, vr);
  point_row<ND, ROLL, OPTK>(st, 1, vr + NV);
  if (jp != nullptr) {
    // d r / d X_board = A . R_view  (rolling: A ((1-t) R_start + t R_end)); board/charuco.py:112-117 `adjusted_points`
    const double* V = t.view + (size_t)v * (VIEW_STRIDE * (ROLL ? 2 : 1));
    for (int a = 0; a < 2; ++a)
      for (int k = 0; k < 3; ++k) {
        double sum = 0.0;
        for (int i = 0; i < 3; ++i) {
          double rik = V[3 * i + k];
          if constexpr (ROLL) rik = (1.0 - st.tr) * rik + st.tr * V[VIEW_STRIDE + 3 * i + k];
          sum += st.A[3 * a + i] * rik;
        }
        jp[3 * a + k] = st.rs[a] * sum;
      }
  }
  return rho;
}

// The board-point block of a row pair (boards=True) is jp = rs A R with R = the view's rotation at the observation's scan time
// (rolling shutter: (1 - t) R_start + t R_end), see point_rows.  The matrix-free products of the LSMR mode need
//   R w   (J_h v:   jp . w = rs A . (R w))      and      R^T q   (J_h^T u:  jp^T u = R^T (sum_a rs_a u_a A_a)).
template <bool ROLL>
MCBA_HD void board_point_direction(const Tables& t, int v, double tr, double w0, double w1, double w2, double* out /*[3]*/) {
  const double* V = t.view + (size_t)v * (VIEW_STRIDE * (ROLL ? 2 : 1));
  for (int i = 0; i < 3; ++i) {
    double r[3];
    for (int k = 0; k < 3; ++k) {
      r[k] = V[3 * i + k];
      if constexpr (ROLL) r[k] = (1.0 - tr) * r[k] + tr * V[VIEW_STRIDE + 3 * i + k];
    }
    out[i] = r[0] * w0 + r[1] * w1 + r[2] * w2;
  }
}
template <bool ROLL>
MCBA_HD void board_point_adjoint(const Tables& t, int v, double tr, const double* q /*[3]*/, double* out /*[3]*/) {
  const double* V = t.view + (size_t)v * (VIEW_STRIDE * (ROLL ? 2 : 1));
  for (int k = 0; k < 3; ++k) {
    double sum = 0.0;
    for (int i = 0; i < 3; ++i) {
      double rik = V[3 * i + k];
      if constexpr (ROLL) rik = (1.0 - tr) * rik + tr * V[VIEW_STRIDE + 3 * i + k];
      sum += q[i] * rik;
    }
    out[k] = sum;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// table preparation (bodies of k_prep / k_views)
// ---------------------------------------------------------------------------------------------------------------
template <class XV>
MCBA_HD void prep_item(const Dims& d, const Tables& t, const XV& x, int i) {
  if (i < d.n_pose) {
    int oa, of, r;
    if (i < d.pose_board) { oa = d.off_campose; of = d.foff_campose; r = 6 * i; }
    else if (i < d.pose_motion) { oa = d.off_boardpose; of = d.foff_boardpose; r = 6 * (i - d.pose_board); }
    else { oa = d.off_motion; of = d.foff_motion; r = 6 * (i - d.pose_motion); }
    double rt[6];
    for (int k = 0; k < 6; ++k) rt[k] = block_value(t, x, oa, of, r + k);
    double e[POSE_STRIDE];
    pose_entry(rt, e);
    for (int k = 0; k < POSE_STRIDE; ++k) t.pose[(size_t)i * POSE_STRIDE + k] = e[k];
    return;
  }
  i -= d.n_pose;
  if (i < d.C) {
    double p[5 + MAX_DIST];
    const int kc = 5 + d.ND;
#pragma unroll
    for (int k = 0; k < 5 + MAX_DIST; ++k) p[k] = k < kc ? block_value(t, x, d.off_cameras, d.foff_cameras, i * kc + k) : 0.0;
    double e[CAM_STRIDE];
    camera_entry(p, d.ND, t.img_h[i], (t.fix_aspect[i] & 1) != 0, e, (t.fix_aspect[i] & 2) != 0);   // (bit 1: fisheye camera of a mixed rig)
#pragma unroll
    for (int k = 0; k < CAM_STRIDE; ++k) t.cam[(size_t)i * CAM_STRIDE + k] = e[k];
    return;
  }
  i -= d.C;
  if (i < d.B * d.P) {
    const int b = i / d.P, p = i % d.P;
    const int nb = t.board_off[b + 1] - t.board_off[b];
    for (int k = 0; k < 3; ++k)
      t.board_points[3 * i + k] = (p < nb) ? block_value(t, x, d.off_boards, d.foff_boards, 3 * (t.board_off[b] + p) + k) : 0.0;
  }
}
MCBA_HD void prep_item(const Dims& d, const Tables& t, const double* x, int i) { prep_item(d, t, PlainX{x}, i); }

// chain matrix board -> camera of view (f, c, b), chain ch (rolling shutter: 0 = start pose, 1 = end pose): out[12] = R | t
MCBA_HD void view_chain(const Dims& d, const PoseSrc& ps, const double* bwg, int f, int c, int b, int ch, double* out) {
  const double* Pc = ps.cam + (size_t)c * POSE_STRIDE;
  const double* Pb = ps.board + (size_t)b * POSE_STRIDE;
  double R1[9], t1[3], R2[9], t2[3];
  if (d.motion == MOTION_HAND_EYE) {
    const double* Wb = ps.mot;
    const double* G = ps.mot + POSE_STRIDE;
    const double* Bf = bwg + 12 * (size_t)f;
    se3_mul(Pc + POSE_R, Pc + POSE_T, G + POSE_R, G + POSE_T, R1, t1);
    se3_mul(R1, t1, Bf, Bf + 9, R2, t2);
    se3_mul(R2, t2, Wb + POSE_R, Wb + POSE_T, R1, t1);
    se3_mul(R1, t1, Pb + POSE_R, Pb + POSE_T, R2, t2);
  } else {
    const double* Pf = ps.mot + (size_t)(ch * ps.chain + (f - ps.f0)) * POSE_STRIDE;
    se3_mul(Pc + POSE_R, Pc + POSE_T, Pf + POSE_R, Pf + POSE_T, R1, t1);
    se3_mul(R1, t1, Pb + POSE_R, Pb + POSE_T, R2, t2);
  }
  for (int k = 0; k < 9; ++k) out[k] = R2[k];
  for (int k = 0; k < 3; ++k) out[9 + k] = t2[k];
}
MCBA_HD void view_chain(const Dims& d, const Tables& t, int f, int c, int b, int ch, double* out) {
  view_chain(d, global_pose_src(d, t), t.bwg, f, c, b, ch, out);
}

MCBA_HD void view_item(const Dims& d, const Tables& t, int i) {
  const int nch = d.motion == MOTION_ROLLING ? 2 : 1;
  if (i >= d.views() * nch) return;
  const int v = i / nch, ch = i % nch;
  const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
  view_chain(d, t, f, c, b, ch, t.view + (size_t)v * d.view_stride() + ch * VIEW_STRIDE);
}

}  // namespace mcba
