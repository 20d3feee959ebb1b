// mcba_math.h -- per-point / per-pose mathematics of the bundle-adjustment hot path (FP64 throughout).
//
// Pure functions shared by every kernel in mcba_kernels.hip.  They compile as __host__ __device__ so that
// tests/hostmath/ can build them with g++ and check the formulas against the oracle on the CPU-only build box;
// the product never runs them on the host.
//
// Reference behaviour restated here (paths relative to /root/reference/multical/):
//   rotation vector -> matrix          transform/rtvec.py:24-27 (scipy Rotation.from_rotvec().as_matrix())
//   pose chain                         X_cam = camera_pose[c] . rig_pose[f] . board_pose[b] . X
//                                      optimization/calibration.py:87-90, motion/static_frames.py:16-25, tables.py:303-304
//   rolling shutter                    X_cam = (1-t) X_start + t X_end, t = y_observed / image_height
//                                      motion/rolling_frames.py:15-41, transform/interpolate.py:6-8
//   hand-eye                           rig[f] = gripper_wrt_camera . base_wrt_gripper[f] . world_wrt_base
//                                      motion/hand_eye.py:43-46
//   pinhole + Brown-Conrady            camera.py:124-128 -> cv2.projectPoints (cvProjectPoints2Internal)
//   Kannala-Brandt fisheye             camera_fisheye.py:113-117 -> cv2.fisheye.projectPoints
// The analytic derivatives have no counterpart in the reference (it uses scipy's finite differences,
// optimization/calibration.py:209-210); they are checked against those finite differences in tests/.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MCBA_HD __host__ __device__ __forceinline__
#else
#define MCBA_HD inline
#endif

namespace mcba {

// ---------------------------------------------------------------------------------------------------------
// layout constants of the per-evaluation device tables
// ---------------------------------------------------------------------------------------------------------
constexpr int POSE_STRIDE = 24;   // R[9] t[3] L[9] pad[3]      (L = left Jacobian of SO(3) at the rotation vector)
constexpr int POSE_R = 0, POSE_T = 9, POSE_L = 12;
constexpr int MAX_DIST = 14;
constexpr int CAM_STRIDE = 56;    // fx fy cx cy skew | k[14] | T[9] dTx[9] dTy[9] | image_height fix_aspect is_fisheye pad
constexpr int CAM_FX = 0, CAM_FY = 1, CAM_CX = 2, CAM_CY = 3, CAM_SKEW = 4, CAM_K = 5, CAM_TILT = 19,
              CAM_DTX = 28, CAM_DTY = 37, CAM_HEIGHT = 46, CAM_FIXASPECT = 47, CAM_ISFISH = 48;
constexpr int VIEW_STRIDE = 12;   // R[9] t[3] of the full chain board -> camera (x2 for rolling shutter)

// ---------------------------------------------------------------------------------------------------------
// small dense helpers (row-major 3x3)
// ---------------------------------------------------------------------------------------------------------
MCBA_HD void mat3_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
MCBA_HD void mat3_vec(const double* A, const double* v, double* out) {
  for (int i = 0; i < 3; ++i) out[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
MCBA_HD void cross3(const double* a, const double* b, double* out) {
  out[0] = a[1] * b[2] - a[2] * b[1];
  out[1] = a[2] * b[0] - a[0] * b[2];
  out[2] = a[0] * b[1] - a[1] * b[0];
}
// (R1,t1) . (R2,t2)  ->  (R1 R2, R1 t2 + t1)
MCBA_HD void se3_mul(const double* R1, const double* t1, const double* R2, const double* t2, double* R, double* t) {
  mat3_mul(R1, R2, R);
  double v[3];
  mat3_vec(R1, t2, v);
  t[0] = v[0] + t1[0];
  t[1] = v[1] + t1[1];
  t[2] = v[2] + t1[2];
}

// ---------------------------------------------------------------------------------------------------------
// Rodrigues: R = exp([w]x) and the left Jacobian L(w) with  d(R(w) y)/dw = -[R y]x L(w).
//   R = I + a [w]x + b [w]x^2,  L = I + b [w]x + c [w]x^2,
//   a = sin(t)/t, b = (1-cos t)/t^2 = 0.5 (sin(t/2)/(t/2))^2, c = (t - sin t)/t^3   (series for small t)
// Parameters are GLOBAL rotation vectors (optimization/pose_set.py:51-57), so the derivative is taken with respect
// to w itself, including at w = 0 where identity poses start (tables.py:218).
// ---------------------------------------------------------------------------------------------------------
MCBA_HD void rodrigues(const double* w, double* R, double* L) {
  const double x = w[0], y = w[1], z = w[2];
  const double xx = x * x, yy = y * y, zz = z * z;
  const double t2 = xx + yy + zz;
  double a, b, c;
  if (t2 < 1e-4) {
    // |t| < 1e-2: Taylor series, next neglected terms < 1e-22 relative
    a = 1.0 + t2 * (-1.0 / 6 + t2 * (1.0 / 120 + t2 * (-1.0 / 5040 + t2 * (1.0 / 362880))));
    b = 0.5 + t2 * (-1.0 / 24 + t2 * (1.0 / 720 + t2 * (-1.0 / 40320 + t2 * (1.0 / 3628800))));
    c = 1.0 / 6 + t2 * (-1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800))));
  } else {
    const double t = sqrt(t2);
    const double s = sin(t);
    const double sh = sin(0.5 * t);
    a = s / t;
    b = 2.0 * sh * sh / t2;
    if (t2 < 0.25) {
      // avoid the cancellation in t - sin t
      c = 1.0 / 6 + t2 * (-1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800 +
          t2 * (-1.0 / 6227020800.0 + t2 * (1.0 / 1307674368000.0 + t2 * (-1.0 / 355687428096000.0)))))));
    } else {
      c = (t - s) / (t2 * t);
    }
  }
  const double xy = x * y, xz = x * z, yz = y * z;
  R[0] = 1.0 - b * (yy + zz); R[1] = b * xy - a * z;       R[2] = b * xz + a * y;
  R[3] = b * xy + a * z;       R[4] = 1.0 - b * (xx + zz); R[5] = b * yz - a * x;
  R[6] = b * xz - a * y;       R[7] = b * yz + a * x;       R[8] = 1.0 - b * (xx + yy);
  L[0] = 1.0 - c * (yy + zz); L[1] = c * xy - b * z;       L[2] = c * xz + b * y;
  L[3] = c * xy + b * z;       L[4] = 1.0 - c * (xx + zz); L[5] = c * yz - b * x;
  L[6] = c * xz - b * y;       L[7] = c * yz + b * x;       L[8] = 1.0 - c * (xx + yy);
}

// pose table entry from a 6-vector (rx ry rz tx ty tz)
MCBA_HD void pose_entry(const double* rt, double* entry) {
  rodrigues(rt, entry + POSE_R, entry + POSE_L);
  entry[POSE_T + 0] = rt[3];
  entry[POSE_T + 1] = rt[4];
  entry[POSE_T + 2] = rt[5];
  entry[21] = entry[22] = entry[23] = 0.0;
}

// ---------------------------------------------------------------------------------------------------------
// tilt-sensor model of OpenCV (distortion_model.hpp: computeTiltProjectionMatrix) and its derivatives
// ---------------------------------------------------------------------------------------------------------
MCBA_HD void tilt_matrices(double tx, double ty, double* T, double* dTx, double* dTy) {
  const double cx = cos(tx), sx = sin(tx), cy = cos(ty), sy = sin(ty);
  // Rxy = Ry Rx,  Rx = [1 0 0; 0 cx sx; 0 -sx cx],  Ry = [cy 0 -sy; 0 1 0; sy 0 cy]
  const double Rm[9] = {cy, sy * sx, -sy * cx, 0, cx, sx, sy, -cy * sx, cy * cx};
  const double dRx[9] = {0, sy * cx, sy * sx, 0, -sx, cx, 0, -cy * cx, -cy * sx};
  const double dRy[9] = {-sy, cy * sx, -cy * cx, 0, 0, 0, cy, sy * sx, -sy * cx};
  // Pz(R) = [R22 0 -R02; 0 R22 -R12; 0 0 1],  T = Pz R
  auto build = [](const double* Rv, const double* dR, double* out, bool deriv, const double* Rbase) {
    // out = Pz(Rbase) Rv                 (deriv == false, Rv == Rbase)
    // out = dPz Rbase + Pz(Rbase) dR     (deriv == true)
    const double p22 = Rbase[8], p02 = Rbase[2], p12 = Rbase[5];
    if (!deriv) {
      for (int j = 0; j < 3; ++j) {
        out[j] = p22 * Rv[j] - p02 * Rv[6 + j];
        out[3 + j] = p22 * Rv[3 + j] - p12 * Rv[6 + j];
        out[6 + j] = Rv[6 + j];
      }
    } else {
      const double d22 = dR[8], d02 = dR[2], d12 = dR[5];
      for (int j = 0; j < 3; ++j) {
        out[j] = d22 * Rbase[j] - d02 * Rbase[6 + j] + p22 * dR[j] - p02 * dR[6 + j];
        out[3 + j] = d22 * Rbase[3 + j] - d12 * Rbase[6 + j] + p22 * dR[3 + j] - p12 * dR[6 + j];
        out[6 + j] = dR[6 + j];
      }
    }
  };
  build(Rm, nullptr, T, false, Rm);
  build(nullptr, dRx, dTx, true, Rm);
  build(nullptr, dRy, dTy, true, Rm);
}

// camera table entry from the reference's per-camera parameter block [fx fy cx cy skew dist...] (camera.py:144-171)
MCBA_HD void camera_entry(const double* p, int n_dist, double image_height, bool fix_aspect, double* e,
                          bool is_fisheye = false) {
  for (int i = 0; i < CAM_STRIDE; ++i) e[i] = 0.0;
  e[CAM_FX] = p[0];
  e[CAM_FY] = fix_aspect ? p[0] : p[1];      // camera.py:159-160: fx, fy = (f[0], f[0]) under fix_aspect
  e[CAM_CX] = p[2];
  e[CAM_CY] = p[3];
  e[CAM_SKEW] = p[4];                        // carried, never read by either OpenCV projection
#pragma unroll
  for (int i = 0; i < MAX_DIST; ++i)           // (fixed trip count: p and e stay in registers, no scratch)
    if (i < n_dist) e[CAM_K + i] = p[5 + i];
  if (n_dist == 14) {
    tilt_matrices(p[5 + 12], p[5 + 13], e + CAM_TILT, e + CAM_DTX, e + CAM_DTY);
  } else {
    e[CAM_TILT + 0] = e[CAM_TILT + 4] = e[CAM_TILT + 8] = 1.0;
  }
  e[CAM_HEIGHT] = image_height;
  e[CAM_FIXASPECT] = fix_aspect ? 1.0 : 0.0;
  e[CAM_ISFISH] = is_fisheye ? 1.0 : 0.0;      // (read by the mixed-rig instantiation only)
}

// ---------------------------------------------------------------------------------------------------------
// projection of a camera-frame point; optional derivatives.
//   uv[2]            pixel
//   A[6]             d(u,v)/d(X,Y,Z)                     (row-major 2x3)
//   Kc[2*(4+ND)]     d(u,v)/d(fx, fy, cx, cy, k_0..k_{ND-1}), row-major 2 x (4+ND); skew column omitted (== 0).
//                    Under fix_aspect column 0 is d/df of the single focal parameter and column 1 is zero.
// ---------------------------------------------------------------------------------------------------------
// `cam` points at the parameter block [fx fy cx cy skew k...] -- the head of a camera-table entry, or the camera's block
// inside x itself (same order, camera.py:150-155) -- and `ext` at the table entry's tail [T dTx dTy height fix_aspect]
// (CAM_TILT onwards).  Under fix_aspect the second focal entry is ignored (camera.py:159-160).
// distorted normalised coordinates (xd, yd) of the normalised point (x, y), d(xd, yd)/d(x, y) and d(xd, yd)/d k (dk[ND + i]
// = the y row), for the two projection families.  k = the camera's coefficient block (ND slots).
template <int ND, bool JAC>
MCBA_HD void distort_fisheye(const double* k, double x, double y, double& xd, double& yd, double& dxx, double& dxy, double& dyx,
                             double& dyy, double* dk) {
  const double r2 = x * x + y * y;
  const double r = sqrt(r2);
  const double th = atan(r);
  const double th2 = th * th, th4 = th2 * th2, th6 = th4 * th2, th8 = th4 * th4;
  const double poly = 1.0 + k[0] * th2 + k[1] * th4 + k[2] * th6 + k[3] * th8;
  const double thd = th * poly;
  const bool big = r > 1e-8;
  const double inv_r = big ? 1.0 / r : 1.0;
  const double s = big ? thd * inv_r : 1.0;
  xd = x * s;
  yd = y * s;
  if constexpr (JAC) {
    if (big) {
      const double dthd = 1.0 + 3.0 * k[0] * th2 + 5.0 * k[1] * th4 + 7.0 * k[2] * th6 + 9.0 * k[3] * th8;
      const double ds = (dthd / (1.0 + r2) * r - thd) * inv_r * inv_r;   // ds/dr
      const double gx = ds * x * inv_r, gy = ds * y * inv_r;             // ds/dx, ds/dy
      dxx = s + x * gx; dxy = x * gy; dyx = y * gx; dyy = s + y * gy;
      const double t3 = th * th2 * inv_r;
      dk[0] = x * t3;        dk[ND + 0] = y * t3;
      dk[1] = x * t3 * th2;  dk[ND + 1] = y * t3 * th2;
      dk[2] = x * t3 * th4;  dk[ND + 2] = y * t3 * th4;
      dk[3] = x * t3 * th6;  dk[ND + 3] = y * t3 * th6;
    } else {
      dxx = 1.0; dxy = 0.0; dyx = 0.0; dyy = 1.0;
      for (int i = 0; i < 2 * ND; ++i) dk[i] = 0.0;
    }
  }
  if constexpr (JAC && ND > 4) {   // (a fisheye camera inside a rig whose coefficient blocks are wider: the rest is not its own)
    for (int i = 4; i < ND; ++i) dk[i] = dk[ND + i] = 0.0;
  }
}

template <int ND, bool JAC>
MCBA_HD void distort_pinhole(const double* k, const double* ext, double x, double y, double& xd, double& yd, double& dxx,
                             double& dxy, double& dyx, double& dyy, double* dk) {
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double a1 = 2.0 * x * y, a2 = r2 + 2.0 * x * x, a3 = r2 + 2.0 * y * y;
  const double k4 = (ND >= 5) ? k[4] : 0.0;
  const double cdist = 1.0 + k[0] * r2 + k[1] * r4 + k4 * r6;
  double icd = 1.0, den_d = 0.0;
  if constexpr (ND >= 8) {
    icd = 1.0 / (1.0 + k[5] * r2 + k[6] * r4 + k[7] * r6);
    den_d = k[5] + 2.0 * k[6] * r2 + 3.0 * k[7] * r4;
  }
  const double radial = cdist * icd;
  double xd0 = x * radial + k[2] * a1 + k[3] * a2;
  double yd0 = y * radial + k[2] * a3 + k[3] * a1;
  double sx = 0.0, sy = 0.0;   // d(thin prism)/dr2
  if constexpr (ND >= 12) {
    xd0 += k[8] * r2 + k[9] * r4;
    yd0 += k[10] * r2 + k[11] * r4;
    sx = k[8] + 2.0 * k[9] * r2;
    sy = k[10] + 2.0 * k[11] * r2;
  }
  double j00 = 0, j01 = 0, j10 = 0, j11 = 0;
  if constexpr (JAC) {
    const double cd_d = k[0] + 2.0 * k[1] * r2 + 3.0 * k4 * r4;
    const double rad_d = cd_d * icd - cdist * icd * icd * den_d;   // d radial / d r2
    j00 = radial + 2.0 * x * x * rad_d + 2.0 * k[2] * y + 6.0 * k[3] * x + 2.0 * x * sx;
    j01 = 2.0 * x * y * rad_d + 2.0 * k[2] * x + 2.0 * k[3] * y + 2.0 * y * sx;
    j10 = 2.0 * x * y * rad_d + 2.0 * k[2] * x + 2.0 * k[3] * y + 2.0 * x * sy;
    j11 = radial + 2.0 * y * y * rad_d + 6.0 * k[2] * y + 2.0 * k[3] * x + 2.0 * y * sy;
    dk[0] = x * r2 * icd;  dk[ND + 0] = y * r2 * icd;
    dk[1] = x * r4 * icd;  dk[ND + 1] = y * r4 * icd;
    dk[2] = a1;            dk[ND + 2] = a3;
    dk[3] = a2;            dk[ND + 3] = a1;
    if constexpr (ND >= 5) { dk[4] = x * r6 * icd; dk[ND + 4] = y * r6 * icd; }
    if constexpr (ND >= 8) {
      const double q = -cdist * icd * icd;
      dk[5] = x * q * r2;  dk[ND + 5] = y * q * r2;
      dk[6] = x * q * r4;  dk[ND + 6] = y * q * r4;
      dk[7] = x * q * r6;  dk[ND + 7] = y * q * r6;
    }
    if constexpr (ND >= 12) {
      dk[8] = r2;   dk[ND + 8] = 0.0;
      dk[9] = r4;   dk[ND + 9] = 0.0;
      dk[10] = 0.0; dk[ND + 10] = r2;
      dk[11] = 0.0; dk[ND + 11] = r4;
    }
  }
  if constexpr (ND >= 14) {
    const double* T = ext;
    const double vx = T[0] * xd0 + T[1] * yd0 + T[2];
    const double vy = T[3] * xd0 + T[4] * yd0 + T[5];
    const double vz = T[6] * xd0 + T[7] * yd0 + T[8];
    const double inv = (vz != 0.0) ? 1.0 / vz : 1.0;
    xd = vx * inv;
    yd = vy * inv;
    if constexpr (JAC) {
      // d(xd,yd)/d(xd0,yd0)
      const double t00 = (T[0] - xd * T[6]) * inv, t01 = (T[1] - xd * T[7]) * inv;
      const double t10 = (T[3] - yd * T[6]) * inv, t11 = (T[4] - yd * T[7]) * inv;
      dxx = t00 * j00 + t01 * j10; dxy = t00 * j01 + t01 * j11;
      dyx = t10 * j00 + t11 * j10; dyy = t10 * j01 + t11 * j11;
      for (int i = 0; i < 12; ++i) {
        const double ax = dk[i], ay = dk[ND + i];
        dk[i] = t00 * ax + t01 * ay;
        dk[ND + i] = t10 * ax + t11 * ay;
      }
      const double* D[2] = {ext + (CAM_DTX - CAM_TILT), ext + (CAM_DTY - CAM_TILT)};
      for (int q = 0; q < 2; ++q) {
        const double* d = D[q];
        const double wx = d[0] * xd0 + d[1] * yd0 + d[2];
        const double wy = d[3] * xd0 + d[4] * yd0 + d[5];
        const double wz = d[6] * xd0 + d[7] * yd0 + d[8];
        dk[12 + q] = (wx - xd * wz) * inv;
        dk[ND + 12 + q] = (wy - yd * wz) * inv;
      }
    }
  } else {
    xd = xd0;
    yd = yd0;
    if constexpr (JAC) { dxx = j00; dxy = j01; dyx = j10; dyy = j11; }
  }
}

// FISHEYE: 0 = Brown-Conrady pinhole (cv2.projectPoints), 1 = Kannala-Brandt fisheye (cv2.fisheye.projectPoints), 2 = decided
// per camera at run time by the camera entry's CAM_ISFISH flag (rigs that MIX the two families; wave-uniform in the kernels:
// a view has one camera)
template <int ND, int FISHEYE, bool JAC>
MCBA_HD void project_point(const double* cam, const double* ext, const double* X, double* uv, double* A, double* Kc) {
  constexpr int KI = 4 + ND;
  const bool fa = ext[CAM_FIXASPECT - CAM_TILT] != 0.0;
  const double fx = cam[CAM_FX], fy = fa ? cam[CAM_FX] : cam[CAM_FY], cx = cam[CAM_CX], cy = cam[CAM_CY];
  const double* k = cam + CAM_K;
  const double Z = X[2];
  const double iz = (Z != 0.0) ? 1.0 / Z : 1.0;   // cvProjectPoints2Internal: z = z ? 1/z : 1
  const double x = X[0] * iz, y = X[1] * iz;
  double xd, yd;              // distorted normalised coordinates
  double dxx = 0, dxy = 0, dyx = 0, dyy = 0;  // d(xd,yd)/d(x,y)
  double dk[2 * (ND > 0 ? ND : 1)];
  if constexpr (FISHEYE == 1) {
    distort_fisheye<ND, JAC>(k, x, y, xd, yd, dxx, dxy, dyx, dyy, dk);
  } else if constexpr (FISHEYE == 0) {
    distort_pinhole<ND, JAC>(k, ext, x, y, xd, yd, dxx, dxy, dyx, dyy, dk);
  } else {
    if (ext[CAM_ISFISH - CAM_TILT] != 0.0) distort_fisheye<ND, JAC>(k, x, y, xd, yd, dxx, dxy, dyx, dyy, dk);
    else distort_pinhole<ND, JAC>(k, ext, x, y, xd, yd, dxx, dxy, dyx, dyy, dk);
  }

  uv[0] = fx * xd + cx;
  uv[1] = fy * yd + cy;

  if constexpr (JAC) {
    // d(x,y)/d(X,Y,Z) = [iz 0 -x iz; 0 iz -y iz]
    const double ux = fx * dxx, uy = fx * dxy, vx_ = fy * dyx, vy_ = fy * dyy;
    A[0] = ux * iz; A[1] = uy * iz; A[2] = -(ux * x + uy * y) * iz;
    A[3] = vx_ * iz; A[4] = vy_ * iz; A[5] = -(vx_ * x + vy_ * y) * iz;
    // row 0 (u)                           row 1 (v)
    Kc[0] = xd;                            Kc[KI + 0] = fa ? yd : 0.0;
    Kc[1] = 0.0;                           Kc[KI + 1] = fa ? 0.0 : yd;
    Kc[2] = 1.0;                           Kc[KI + 2] = 0.0;
    Kc[3] = 0.0;                           Kc[KI + 3] = 1.0;
    for (int i = 0; i < ND; ++i) {
      Kc[4 + i] = fx * dk[i];
      Kc[KI + 4 + i] = fy * dk[ND + i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// "base" row pair of the pose Jacobian in the camera frame:  E = A [ -[X]x | I ]   (2 x 6)
// Every pose block k of the chain has d r / d(pose k) = E . T_k with a view-constant 6x6 matrix T_k
// (see view_pose_column below), so only E is accumulated per point.
// ---------------------------------------------------------------------------------------------------------
MCBA_HD void base_row(const double* a, const double* X, double* E /*[6]*/) {
  // a^T (-[X]x) = (X x a)^T
  E[0] = X[1] * a[2] - X[2] * a[1];
  E[1] = X[2] * a[0] - X[0] * a[2];
  E[2] = X[0] * a[1] - X[1] * a[0];
  E[3] = a[0];
  E[4] = a[1];
  E[5] = a[2];
}
MCBA_HD void base_rows(const double* A, const double* X, double* E /*[2][6]*/) {
  base_row(A, X, E);
  base_row(A + 3, X, E + 6);
}

// Column j (0..5) of T_k = [[Rpre L_k, 0], [[o_k]x Rpre L_k, Rpre]] for a pose with prefix rotation Rpre (product of
// everything left of it in the chain), left Jacobian L_k and o_k = translation of (prefix . pose_k).
// Written without dynamic register indexing (unit-vector products, strided output) so that nothing spills to scratch.
MCBA_HD void view_pose_column(const double* Rpre, const double* Lk, const double* ok, int j, double* col, int stride = 1) {
  const int jj = j < 3 ? j : j - 3;
  const double e0 = jj == 0 ? 1.0 : 0.0, e1 = jj == 1 ? 1.0 : 0.0, e2 = jj == 2 ? 1.0 : 0.0;
  if (j < 3) {
    const double l0 = Lk[0] * e0 + Lk[1] * e1 + Lk[2] * e2;   // column jj of L_k
    const double l1 = Lk[3] * e0 + Lk[4] * e1 + Lk[5] * e2;
    const double l2 = Lk[6] * e0 + Lk[7] * e1 + Lk[8] * e2;
    const double t0 = Rpre[0] * l0 + Rpre[1] * l1 + Rpre[2] * l2;
    const double t1 = Rpre[3] * l0 + Rpre[4] * l1 + Rpre[5] * l2;
    const double t2 = Rpre[6] * l0 + Rpre[7] * l1 + Rpre[8] * l2;
    col[0] = t0; col[stride] = t1; col[2 * stride] = t2;
    col[3 * stride] = ok[1] * t2 - ok[2] * t1;
    col[4 * stride] = ok[2] * t0 - ok[0] * t2;
    col[5 * stride] = ok[0] * t1 - ok[1] * t0;
  } else {
    col[0] = 0.0; col[stride] = 0.0; col[2 * stride] = 0.0;
    col[3 * stride] = Rpre[0] * e0 + Rpre[1] * e1 + Rpre[2] * e2;   // column jj of R_pre
    col[4 * stride] = Rpre[3] * e0 + Rpre[4] * e1 + Rpre[5] * e2;
    col[5 * stride] = Rpre[6] * e0 + Rpre[7] * e1 + Rpre[8] * e2;
  }
}

// ---------------------------------------------------------------------------------------------------------
// robust loss, per scalar residual, exactly scipy's construct_loss_function + scale_for_robust_loss_function
// (scipy/optimize/_lsq/least_squares.py:169-238, common.py:720-731):
//   z = (f/C)^2, rho0 = C^2 rho(z), rho1 = rho'(z), rho2 = rho''(z)/C^2,
//   J_scale = sqrt(max(rho1 + 2 rho2 f^2, EPS)),  f <- f rho1 / J_scale,  J row <- J row * J_scale.
// Returns rho0 (its half-sum is the cost); *row_scale and *res_scale are the two factors.
// ---------------------------------------------------------------------------------------------------------
MCBA_HD double robust_loss(int loss, double f_scale, double f, double* row_scale, double* res_scale) {
  if (loss == 0) {
    *row_scale = 1.0;
    *res_scale = 1.0;
    return f * f;
  }
  const double C2 = f_scale * f_scale;
  const double z = f * f / C2;
  double r0, r1, r2;
  switch (loss) {
    case 1: {  // soft_l1
      const double t = 1.0 + z, st = sqrt(t);
      r0 = 2.0 * (st - 1.0); r1 = 1.0 / st; r2 = -0.5 / (t * st);
    } break;
    case 2: {  // huber
      if (z <= 1.0) { r0 = z; r1 = 1.0; r2 = 0.0; }
      else { const double sz = sqrt(z); r0 = 2.0 * sz - 1.0; r1 = 1.0 / sz; r2 = -0.5 / (z * sz); }
    } break;
    case 3: {  // cauchy
      const double t = 1.0 + z;
      r0 = log1p(z); r1 = 1.0 / t; r2 = -1.0 / (t * t);
    } break;
    default: {  // arctan
      const double t = 1.0 + z * z;
      r0 = atan(z); r1 = 1.0 / t; r2 = -2.0 * z / (t * t);
    } break;
  }
  r2 /= C2;
  double js = r1 + 2.0 * r2 * f * f;
  const double EPS = 2.220446049250313e-16;
  if (js < EPS) js = EPS;
  js = sqrt(js);
  *row_scale = js;
  *res_scale = r1 / js;
  return C2 * r0;
}

}  // namespace mcba
