// mcba_camops.h -- launch table of the kernels that are specialised on the camera model.
//
// The per-slot kernels are templates over <number of distortion coefficients, fisheye?> (x motion model x
// intrinsics-on/off x MFMA/plain accumulate).  Each camera model is instantiated in its own translation unit
// (mcba_cam_*.hip) so that the variants compile in parallel; the driver picks a table at mcba_create().
#pragma once
#include <hip/hip_runtime.h>
#include "mcba_device.h"

namespace mcba {

struct CamOps {
  // first: per-view first residual index (k_view_scan), needed when r is written
  void (*residual)(const Dims&, const Tables&, hipStream_t, const int32_t* first, double* r, double* proj, double* err,
                   uint8_t* valid);
  void (*project_model)(const Dims&, const Tables&, hipStream_t, int iterations, double* proj);
  void (*cost)(const Dims&, const Tables&, hipStream_t, double* partial, int nblk);
  void (*jacobian)(const Dims&, const Tables&, hipStream_t, int row_nnz, double* vals, int32_t* cols);
  // za != nullptr: the table-fed fused form (k_linearize builds That / the chains from the pose table itself and zeroes za[na], zb[nb]);
  // compact != nullptr: ... over the compacted observation tables of the current inlier set (form 3)
  void (*linearize)(const Dims&, const Tables&, hipStream_t, double* rec, const uint16_t* tri, bool mfma, int epoch,
                    const double* x, double* za, int na, double* zb, int nb, const LsmrCompact* compact);
  void (*points)(const Dims&, const Tables&, hipStream_t, int n_points, double* Hss, double* Hfs, double* g);
  // solver = "lsmr": u <- J_h v - alpha u (mode 0) / u <- f (mode 1), and the per-view partials of J_h^T (u inv_beta);
  // ls != nullptr: alpha / inv_beta and the stop flag come from the device-resident state of the solve (mcba_lsmr.h)
  void (*lsmr_jv)(const Dims&, const Tables&, hipStream_t, const int32_t* first, int mode, const double* dscale, const double* v,
                  double alpha, double* u, double* partial, int nblk, const double* ls);
  // bpart != nullptr (boards=True): + jp^T u of every observation, 3 doubles in the residual order (k_lsmr_gather sums them per point)
  void (*lsmr_jtu)(const Dims&, const Tables&, hipStream_t, const int32_t* first, double inv_beta, double* u, double* part,
                   int part_stride, double* bpart, int nblk, const double* ls);
  // both products of an LSMR iteration in one pass (k_lsmr_fused): uhat <- J_h v - alpha uhat_old / beta_old, per-view J_h^T uhat
  void (*lsmr_fused)(const Dims&, const Tables&, hipStream_t, const int32_t* first, const double* dscale, const double* v, double* u,
                     double* partial, double* part, int part_stride, double* bpart, int nblk, const double* ls);
  // k_lsmr_fused + the rotation / vector update of the previous step in its head (two-launch iteration)
  void (*lsmr_fused2)(const Dims&, const Tables&, hipStream_t, const int32_t* first, const double* dscale, const double* v, double* u,
                      double* partial, double* xpart, double* part, int part_stride, double* bpart, int nblk, const double* lsIn,
                      double* lsOut, const double* vpart, int nv, double* hbar, double* x, double* h, double* cache, int mode, LsmrCompact cp);
};

const CamOps* cam_ops_pin4();
const CamOps* cam_ops_pin5();
const CamOps* cam_ops_pin8();
const CamOps* cam_ops_pin12();
const CamOps* cam_ops_pin14();
const CamOps* cam_ops_fish4();
const CamOps* cam_ops_mix14();   // pinhole and fisheye cameras in one rig (family per camera, 14-wide coefficient blocks)

}  // namespace mcba
