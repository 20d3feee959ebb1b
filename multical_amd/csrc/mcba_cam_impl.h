// mcba_cam_impl.h -- body of one camera-model translation unit: define MCBA_ND, MCBA_FISH, MCBA_CAM_FN, include.
#include <algorithm>
#include "mcba_kernels.h"
#include "mcba_camops.h"

namespace mcba {
namespace {

constexpr int ND_ = MCBA_ND;
constexpr int FISH_ = MCBA_FISH;   // 0 pinhole, 1 fisheye, 2 per camera (mixed rig)

inline int slot_grid(const Dims& d) {
  const int n = d.slots();
  int g = (n + 255) / 256;
  return g < 1 ? 1 : (g > 4096 ? 4096 : g);
}

void residual(const Dims& d, const Tables& t, hipStream_t s, const int32_t* first, double* r, double* proj, double* err,
              uint8_t* valid) {
  if (d.views() == 0) return;
  const dim3 grid(std::min(8192, (d.views() + 3) / 4)), block(256);   // one wavefront per view, four views per block
  if (d.motion == MOTION_ROLLING)
    hipLaunchKernelGGL((k_residual<ND_, FISH_, true>), grid, block, 0, s, d, t, first, r, proj, err, valid);
  else
    hipLaunchKernelGGL((k_residual<ND_, FISH_, false>), grid, block, 0, s, d, t, first, r, proj, err, valid);
}

void project_model(const Dims& d, const Tables& t, hipStream_t s, int iterations, double* proj) {
  if (d.motion == MOTION_ROLLING)
    hipLaunchKernelGGL((k_project_model<ND_, FISH_, true>), dim3(slot_grid(d)), dim3(256), 0, s, d, t, iterations, proj);
  else
    hipLaunchKernelGGL((k_project_model<ND_, FISH_, false>), dim3(slot_grid(d)), dim3(256), 0, s, d, t, iterations, proj);
}

void cost(const Dims& d, const Tables& t, hipStream_t s, double* partial, int nblk) {
  const bool roll = d.motion == MOTION_ROLLING, robust = d.loss != 0;
  if (roll && robust) hipLaunchKernelGGL((k_cost<ND_, FISH_, true, true>), dim3(nblk), dim3(64), 0, s, d, t, partial);
  else if (roll) hipLaunchKernelGGL((k_cost<ND_, FISH_, true, false>), dim3(nblk), dim3(64), 0, s, d, t, partial);
  else if (robust) hipLaunchKernelGGL((k_cost<ND_, FISH_, false, true>), dim3(nblk), dim3(64), 0, s, d, t, partial);
  else hipLaunchKernelGGL((k_cost<ND_, FISH_, false, false>), dim3(nblk), dim3(64), 0, s, d, t, partial);
}

void jacobian(const Dims& d, const Tables& t, hipStream_t s, int row_nnz, double* vals, int32_t* cols) {
  if (d.motion == MOTION_ROLLING)
    hipLaunchKernelGGL((k_jacobian<ND_, FISH_, true>), dim3(slot_grid(d)), dim3(128), 0, s, d, t, row_nnz, vals, cols);
  else
    hipLaunchKernelGGL((k_jacobian<ND_, FISH_, false>), dim3(slot_grid(d)), dim3(128), 0, s, d, t, row_nnz, vals, cols);
}

template <int MOTION, bool OPTK>
void lin2(const Dims& d, const Tables& t, hipStream_t s, double* rec, const uint16_t* tri, bool mfma, int epoch,
          const double* x, double* za, int na, double* zb, int nb, const LsmrCompact* cpp) {
  // table-fed fused form: za != nullptr (pose entries / intrinsics from the pose and camera tables; zeroes the assembly
  // targets za[na], zb[nb]); za == nullptr: table form (That / chains per view from k_tmat)
  const bool fused = za != nullptr;
  // cpp != nullptr (fused form only): the compacted observation tables of the current inlier set (form 3)
  const LsmrCompact cp = cpp != nullptr ? *cpp : LsmrCompact{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (d.views() == 0 && !fused) return;   // empty frame shard (the fused form still zeroes the assembly targets)
  // persistent wavefronts: 8 single-wave workgroups per CU (2 per SIMD: 256-VGPR budget, 20 KB LDS each) x 256 CUs
  // (rigs with tens of thousands of views: four times as many workgroups for the dispatcher to hand out -- 16 x 1000 x 5, 80 000 views,
  //  4 waves per SIMD: 4096 -> 57.8 us, 8192 -> 54.5, 16384 -> 53.5; 8 x 500 x 2 stays at 4096: 41.6 against 42.3 us -- profiles/r06_lin_compact.txt)
  const int want = epoch > 0 ? epoch : (d.views() >= 32768 ? 4 * LIN_GRID_MAX : LIN_GRID_MAX);   // the last argument carries the debug grid override
  const dim3 grid(std::max(1, d.views() < want ? d.views() : want)), block(64);
  // the linear loss (the reference's default, calibration.py:199) has its own instantiation of the MFMA kernel: no loss
  // switch and no robust-scale constants in the hot loop; the plain-FMA validation build keeps the generic form
  if (t.dbg != nullptr) {   // per-phase cycle stamps (debug API): the table form of the MFMA kernel
    hipLaunchKernelGGL((k_linearize<ND_, FISH_, MOTION, OPTK, true, true, 0, true>), grid, block, 0, s, d, t, rec, tri, epoch, x, za, na, zb, nb, cp);
    return;
  }
  const bool robust = d.loss != 0;
#define MCBA_LIN(ROB, FM) hipLaunchKernelGGL((k_linearize<ND_, FISH_, MOTION, OPTK, true, ROB, FM>), grid, block, 0, s, d, t, rec, tri, epoch, x, za, na, zb, nb, cp)
  if (mfma && fused && cpp != nullptr) { if (robust) MCBA_LIN(true, 3); else MCBA_LIN(false, 3); }
  else if (mfma && fused) { if (robust) MCBA_LIN(true, 2); else MCBA_LIN(false, 2); }
  else if (mfma) { if (robust) MCBA_LIN(true, 0); else MCBA_LIN(false, 0); }
  else
    hipLaunchKernelGGL((k_linearize<ND_, FISH_, MOTION, OPTK, false, true, 0>), grid, block, 0, s, d, t, rec, tri, epoch, x, za, na, zb, nb, cp);
#undef MCBA_LIN
}

template <int MOTION>
void lin1(const Dims& d, const Tables& t, hipStream_t s, double* rec, const uint16_t* tri, bool mfma, int epoch,
          const double* x, double* za, int na, double* zb, int nb, const LsmrCompact* cpp) {
  if (d.KI > 0) lin2<MOTION, true>(d, t, s, rec, tri, mfma, epoch, x, za, na, zb, nb, cpp);
  else lin2<MOTION, false>(d, t, s, rec, tri, mfma, epoch, x, za, na, zb, nb, cpp);
}

void linearize(const Dims& d, const Tables& t, hipStream_t s, double* rec, const uint16_t* tri, bool mfma, int epoch,
               const double* x, double* za, int na, double* zb, int nb, const LsmrCompact* cpp) {
  if (d.motion == MOTION_STATIC) lin1<MOTION_STATIC>(d, t, s, rec, tri, mfma, epoch, x, za, na, zb, nb, cpp);
  else if (d.motion == MOTION_ROLLING) lin1<MOTION_ROLLING>(d, t, s, rec, tri, mfma, epoch, x, za, na, zb, nb, cpp);
  else lin1<MOTION_HAND_EYE>(d, t, s, rec, tri, mfma, epoch, x, za, na, zb, nb, cpp);
}

template <int MOTION>
void pts1(const Dims& d, const Tables& t, hipStream_t s, int nq, double* Hss, double* Hfs, double* g) {
  if (d.KI > 0)
    hipLaunchKernelGGL((k_points<ND_, FISH_, MOTION, true>), dim3(nq), dim3(256), 0, s, d, t, Hss, Hfs, g);
  else
    hipLaunchKernelGGL((k_points<ND_, FISH_, MOTION, false>), dim3(nq), dim3(256), 0, s, d, t, Hss, Hfs, g);
}

void points(const Dims& d, const Tables& t, hipStream_t s, int nq, double* Hss, double* Hfs, double* g) {
  if (nq <= 0) return;
  if (d.motion == MOTION_STATIC) pts1<MOTION_STATIC>(d, t, s, nq, Hss, Hfs, g);
  else if (d.motion == MOTION_ROLLING) pts1<MOTION_ROLLING>(d, t, s, nq, Hss, Hfs, g);
  else pts1<MOTION_HAND_EYE>(d, t, s, nq, Hss, Hfs, g);
}

template <int MOTION, bool OPTK>
void jv2(const Dims& d, const Tables& t, hipStream_t s, const int32_t* first, int mode, const double* dscale, const double* v,
         double alpha, double* u, double* partial, int nblk, const double* ls) {
  if (d.loss != 0)
    hipLaunchKernelGGL((k_lsmr_jv<ND_, FISH_, MOTION, OPTK, true>), dim3(nblk), dim3(64), 0, s, d, t, first, mode, dscale, v, alpha, u, partial, ls);
  else
    hipLaunchKernelGGL((k_lsmr_jv<ND_, FISH_, MOTION, OPTK, false>), dim3(nblk), dim3(64), 0, s, d, t, first, mode, dscale, v, alpha, u, partial, ls);
}
template <int MOTION>
void jv1(const Dims& d, const Tables& t, hipStream_t s, const int32_t* first, int mode, const double* dscale, const double* v,
         double alpha, double* u, double* partial, int nblk, const double* ls) {
  if (d.KI > 0) jv2<MOTION, true>(d, t, s, first, mode, dscale, v, alpha, u, partial, nblk, ls);
  else jv2<MOTION, false>(d, t, s, first, mode, dscale, v, alpha, u, partial, nblk, ls);
}
void lsmr_jv(const Dims& d, const Tables& t, hipStream_t s, const int32_t* first, int mode, const double* dscale, const double* v,
             double alpha, double* u, double* partial, int nblk, const double* ls) {
  if (d.motion == MOTION_STATIC) jv1<MOTION_STATIC>(d, t, s, first, mode, dscale, v, alpha, u, partial, nblk, ls);
  else if (d.motion == MOTION_ROLLING) jv1<MOTION_ROLLING>(d, t, s, first, mode, dscale, v, alpha, u, partial, nblk, ls);
  else jv1<MOTION_HAND_EYE>(d, t, s, first, mode, dscale, v, alpha, u, partial, nblk, ls);
}

template <int MOTION, bool OPTK>
void jtu2(const Dims& d, const Tables& t, hipStream_t s, const int32_t* first, double inv_beta, double* u, double* part,
          int part_stride, double* bpart, int nblk, const double* ls) {
  if (d.loss != 0)
    hipLaunchKernelGGL((k_lsmr_jtu<ND_, FISH_, MOTION, OPTK, true>), dim3(nblk), dim3(64), 0, s, d, t, first, inv_beta, u, part, part_stride, bpart, ls);
  else
    hipLaunchKernelGGL((k_lsmr_jtu<ND_, FISH_, MOTION, OPTK, false>), dim3(nblk), dim3(64), 0, s, d, t, first, inv_beta, u, part, part_stride, bpart, ls);
}
template <int MOTION>
void jtu1(const Dims& d, const Tables& t, hipStream_t s, const int32_t* first, double inv_beta, double* u, double* part,
          int part_stride, double* bpart, int nblk, const double* ls) {
  if (d.KI > 0) jtu2<MOTION, true>(d, t, s, first, inv_beta, u, part, part_stride, bpart, nblk, ls);
  else jtu2<MOTION, false>(d, t, s, first, inv_beta, u, part, part_stride, bpart, nblk, ls);
}
void lsmr_jtu(const Dims& d, const Tables& t, hipStream_t s, const int32_t* first, double inv_beta, double* u, double* part,
              int part_stride, double* bpart, int nblk, const double* ls) {
  if (d.motion == MOTION_STATIC) jtu1<MOTION_STATIC>(d, t, s, first, inv_beta, u, part, part_stride, bpart, nblk, ls);
  else if (d.motion == MOTION_ROLLING) jtu1<MOTION_ROLLING>(d, t, s, first, inv_beta, u, part, part_stride, bpart, nblk, ls);
  else jtu1<MOTION_HAND_EYE>(d, t, s, first, inv_beta, u, part, part_stride, bpart, nblk, ls);
}

template <int MOTION, bool OPTK>
void fus2(const Dims& d, const Tables& t, hipStream_t s, const int32_t* first, const double* dscale, const double* v, double* u,
          double* partial, double* part, int part_stride, double* bpart, int nblk, const double* ls) {
  if (d.loss != 0)
    hipLaunchKernelGGL((k_lsmr_fused<ND_, FISH_, MOTION, OPTK, true>), dim3(nblk), dim3(64), 0, s, d, t, first, dscale, v, u, partial, part, part_stride, bpart, ls);
  else
    hipLaunchKernelGGL((k_lsmr_fused<ND_, FISH_, MOTION, OPTK, false>), dim3(nblk), dim3(64), 0, s, d, t, first, dscale, v, u, partial, part, part_stride, bpart, ls);
}
template <int MOTION>
void fus1(const Dims& d, const Tables& t, hipStream_t s, const int32_t* first, const double* dscale, const double* v, double* u,
          double* partial, double* part, int part_stride, double* bpart, int nblk, const double* ls) {
  if (d.KI > 0) fus2<MOTION, true>(d, t, s, first, dscale, v, u, partial, part, part_stride, bpart, nblk, ls);
  else fus2<MOTION, false>(d, t, s, first, dscale, v, u, partial, part, part_stride, bpart, nblk, ls);
}
void lsmr_fused(const Dims& d, const Tables& t, hipStream_t s, const int32_t* first, const double* dscale, const double* v, double* u,
                double* partial, double* part, int part_stride, double* bpart, int nblk, const double* ls) {
  if (d.motion == MOTION_STATIC) fus1<MOTION_STATIC>(d, t, s, first, dscale, v, u, partial, part, part_stride, bpart, nblk, ls);
  else if (d.motion == MOTION_ROLLING) fus1<MOTION_ROLLING>(d, t, s, first, dscale, v, u, partial, part, part_stride, bpart, nblk, ls);
  else fus1<MOTION_HAND_EYE>(d, t, s, first, dscale, v, u, partial, part, part_stride, bpart, nblk, ls);
}

#define MCBA_F2_ARGS const Dims& d, const Tables& t, hipStream_t s, const int32_t* first, const double* dscale, const double* v, double* u, \
                     double* partial, double* xpart, double* part, int part_stride, double* bpart, int nblk, const double* lsIn, \
                     double* lsOut, const double* vpart, int nv, double* hbar, double* x, double* h, double* cache, int mode, LsmrCompact cp
#define MCBA_F2_PASS d, t, s, first, dscale, v, u, partial, xpart, part, part_stride, bpart, nblk, lsIn, lsOut, vpart, nv, hbar, x, h, cache, mode, cp
#define MCBA_F2_LAUNCH(ROB, MO) \
    hipLaunchKernelGGL((k_lsmr_fused2<ND_, FISH_, MOTION, OPTK, ROB, MO>), dim3(nblk), dim3(64), 0, s, d, t, first, dscale, v, u, partial, xpart, part, \
                       part_stride, bpart, lsIn, lsOut, vpart, nv, hbar, x, h, cache, cp)
// mode (k_lsmr_fused2's MODE): 0 = masks (boards=True), 3 = compact tables (default), 4 = compact + store the per-observation state,
// 2 = stream the state back
template <int MOTION, bool OPTK>
void fus22(MCBA_F2_ARGS) {
  if (d.loss != 0) {
    if (mode == 2) MCBA_F2_LAUNCH(true, 2);
    else if (mode == 3) MCBA_F2_LAUNCH(true, 3);
    else if (mode == 4) MCBA_F2_LAUNCH(true, 4);
    else MCBA_F2_LAUNCH(true, 0);
  } else {
    if (mode == 2) MCBA_F2_LAUNCH(false, 2);
    else if (mode == 3) MCBA_F2_LAUNCH(false, 3);
    else if (mode == 4) MCBA_F2_LAUNCH(false, 4);
    else MCBA_F2_LAUNCH(false, 0);
  }
}
template <int MOTION>
void fus21(MCBA_F2_ARGS) {
  if (d.KI > 0) fus22<MOTION, true>(MCBA_F2_PASS);
  else fus22<MOTION, false>(MCBA_F2_PASS);
}
void lsmr_fused2(MCBA_F2_ARGS) {
  if (d.motion == MOTION_STATIC) fus21<MOTION_STATIC>(MCBA_F2_PASS);
  else if (d.motion == MOTION_ROLLING) fus21<MOTION_ROLLING>(MCBA_F2_PASS);
  else fus21<MOTION_HAND_EYE>(MCBA_F2_PASS);
}
#undef MCBA_F2_LAUNCH
#undef MCBA_F2_ARGS
#undef MCBA_F2_PASS

const CamOps OPS = {residual, project_model, cost, jacobian, linearize, points, lsmr_jv, lsmr_jtu, lsmr_fused, lsmr_fused2};

}  // namespace

const CamOps* MCBA_CAM_FN() { return &OPS; }

}  // namespace mcba
