// mcba_device.h -- plain-old-data descriptors shared by the host driver (mcba_api.hip) and the kernels.
#pragma once
#include <stdint.h>
#include "mcba_math.h"

#if !defined(__HIPCC__)
struct double2 { double x, y; };   // host-only builds (tests/hostmath); hipcc provides the vector types
struct int4 { int x, y, z, w; };
#endif

namespace mcba {

// persistent k_linearize: 2048 single-wave workgroups are resident (8 per CU x 256 CUs); launching twice that many lets
// the dispatcher hand the rest out as slots free up.  Together with the largest-first order of the active-view list
// (k_active_views) this is longest-processing-time-first list scheduling: the short views fill the tail of the launch.
// Measured at the north-star rig (4401 active views): ascending order 2048 -> 95.7 us, 3072 -> 81.5 us; largest-first
// 2048 -> 74.9, 3072 -> 72.4, 4096 -> 65.7, 4401 (one view each) -> 67.6, exactly 3 views each (1467, ascending) -> 112 us.
constexpr int LIN_GRID_MAX = 4096;
constexpr int MOTION_STATIC = 0, MOTION_ROLLING = 1, MOTION_HAND_EYE = 2;

// Problem shape + index maps, passed BY VALUE to every kernel (fits the kernarg segment).
struct Dims {
  int C, F, B, P;        // cameras, frames (global), boards, padded points per board
  int f0, Fl;            // frame shard owned by this handle: global frames [f0, f0 + Fl)
  int motion;            // MOTION_*
  int ND;                // distortion coefficients per camera
  int fisheye;           // 0/1
  int n;                 // active parameters (length of x)
  int nfull;             // all five blocks
  // offsets of the blocks inside the ACTIVE vector x (-1 = block disabled)
  int off_campose, off_boardpose, off_motion, off_cameras, off_boards;
  // offsets inside the FULL vector
  int foff_campose, foff_boardpose, foff_motion, foff_cameras, foff_boards;
  int n_motion;          // length of the motion block (6F, 12F or 12)
  int KI;                // intrinsic columns carried per observation (4 + ND, skew omitted) or 0 when cameras are fixed
  int NPB;               // pose blocks per view: 3 (static) or 4 (rolling: cam|start|end|board, hand-eye: cam|wb|gc|board)
  int DE;                // base row width: 6, or 12 for rolling shutter
  int NV;                // DE + KI + 1   columns of the per-point row pair  [E | K | r]
  int NL;                // 6*NPB + KI    local parameters of one view
  int N1;                // NL + 1        (+ residual column)
  int rec_size;          // N1 (N1+1)/2   packed upper triangle of the local normal equations
  int rec_stride;        // rec_size + 2  (+ cost, count), even
  int DF;                // eliminated parameters per frame: 6 static, 12 rolling, 0 hand-eye / motion disabled
  int ns;                // shared (reduced) parameters = n - Fl_total*DF ... see shared_index()
  int loss;              // MCBA_LOSS_*
  double f_scale;
  // pose table layout
  int pose_cam, pose_board, pose_motion, n_pose;
  // Rigs whose cameras carry DIFFERENT numbers of distortion coefficients (mcba_problem.camera_n_dist): internally every
  // camera has ND = the largest size, a camera with fewer keeps the rest at zero.  Bit q of cam_kmask[c] marks intrinsic
  // column q (0..3 = fx fy cx cy, 4 + k = distortion coefficient k) of camera c as FROZEN: local_to_x reports it as "not
  // a parameter", so the assembly never writes its rows / columns of H and g (zero column: unit scale, zero step).
  // nullptr for uniform rigs.  A device pointer inside the kernels; host code substitutes a host copy.
  const uint32_t* cam_kmask;
  // Frame sharding: rank / number of ranks of the problem this handle owns a frame shard of (shard_world = 0:
  // not sharded).  The n-vectors of a sharded handle are complete in the SHARED entries and in the entries of its OWN frames
  // only -- nothing of length n ever crosses the ranks; sums over all parameters are formed as all-reduced per-rank partial
  // sums in which every entry is counted exactly once (entry_weight).
  int shard_rank, shard_world;

  // global x index of a shared-parameter index (x order with the eliminated motion block removed)
  MCBA_HD int shared_to_x(int s) const {
    if (DF == 0 || off_motion < 0) return s;
    return s < off_motion ? s : s + n_motion;
  }
  MCBA_HD int x_to_shared(int i) const {   // -1 when i is an eliminated frame parameter
    if (DF == 0 || off_motion < 0) return i;
    if (i < off_motion) return i;
    if (i < off_motion + n_motion) return -1;
    return i - n_motion;
  }
  // x index of eliminated parameter d (0..DF-1) of GLOBAL frame f
  MCBA_HD int frame_to_x(int f, int d) const {
    return (d < 6) ? off_motion + 6 * f + d : off_motion + 6 * F + 6 * f + (d - 6);
  }
  // weight of entry i of an n-vector in this rank's partial of a global sum: eliminated frame parameters count on the rank
  // that owns the frame, shared parameters on rank 0 (the shared entries are replicated: identical on every rank)
  MCBA_HD double entry_weight(int i) const {
    if (shard_world == 0) return 1.0;
    if (DF == 0 || off_motion < 0 || i < off_motion || i >= off_motion + n_motion) return shard_rank == 0 ? 1.0 : 0.0;
    const int q = i - off_motion, f = (q >= 6 * F ? q - 6 * F : q) / 6;
    return (f >= f0 && f < f0 + Fl) ? 1.0 : 0.0;
  }
  MCBA_HD int views() const { return Fl * C * B; }
  MCBA_HD int slots() const { return Fl * C * B * P; }
  MCBA_HD int view_stride() const { return VIEW_STRIDE * (motion == MOTION_ROLLING ? 2 : 1); }
};

// Device tables owned by the handle.  Observation tables are FRAME-MAJOR ([Fl][C][B][P]) so that a frame shard is
// contiguous and the per-frame reductions read consecutive views.
struct Tables {
  const double2* obs;          // observed points
  const uint8_t* inlier;       // Calibration.inliers
  const uint8_t* evalid;       // proj.valid & obs.valid  (mask of tables.reprojection_error)
  const int32_t* obs_index;    // index of the observation in the reference's residual ordering, -1 if not an inlier
  const int32_t* view_count;   // [Fl][C][B] inliers per view
  const int32_t* active_views; // [1 + views]: count, then the indices of the non-empty views (ascending)
  const int32_t* board_off;    // [B+1] prefix of board sizes (points)
  const int32_t* full2act;     // [nfull] full index -> active index or -1
  const double* xfull;         // [nfull] constants for disabled blocks
  const double* bwg;           // [F][12] base_wrt_gripper (R,t), hand-eye
  const double* img_h;         // [C]
  const uint8_t* fix_aspect;   // [C]
  double* board_points;        // [B][P][3] current board geometry
  double* pose;                // [n_pose][POSE_STRIDE]
  double* cam;                 // [C][CAM_STRIDE]
  double* view;                // [Fl][C][B][view_stride]
  int32_t* work_counter;       // [2] dynamic view hand-out of k_linearize (alternating between launches)
  double* tmat;                // [Fl][C][B][DE*NPC] That columns per view (k_tmat), consumed by k_linearize
  long long* dbg;              // optional [views][8] cycle stamps of k_linearize phases (profiling aid), else null
  const int32_t* int2ext;      // [n] index in the CALLER's (ragged) parameter vector, -1 = padded coefficient; null = identity
};

// Compacted observation tables of the lsmr route (built once per inlier set: k_compact_views): the observations of every view in
// the reference's RESIDUAL order -- observed point, board point (x, y) and z -- so that a product kernel streams them without mask
// bytes, compaction or point-index gathers, and one descriptor {view, first residual pair, inliers, 0} per ACTIVE view in the
// largest-first order of Tables::active_views.
struct LsmrCompact {
  const double2* obs;      // [n_inliers] observed (u, v)
  const double2* bxy;      // [n_inliers] board point x, y
  const double* bz;        // [n_inliers] board point z
  const double* tr;        // [n_inliers] rolling shutter: scan time of the observation = observed row / image height (a CONSTANT of the
                           //             observation: the division leaves the head of every evaluation's dependency chain); else null
  const int4* desc;        // [active views] {v, first, count, 0}
};

}  // namespace mcba
