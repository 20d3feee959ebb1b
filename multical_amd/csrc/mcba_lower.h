// mcba_lower.h -- host-side lowering of an mcba_problem (include/mcba.h) to the frame-major tables and index maps the
// kernels consume.  Pure C++ (no HIP calls): shared by the driver (mcba_api.hip) and by tests/hostmath.
//
// Reference behaviour captured here:
//   Calibration.valid / inliers                      optimization/calibration.py:69-81
//   mask of tables.reprojection_error                 tables.py:244-249 (reprojected.valid & point_table.valid)
//   residual ordering of `evaluate`                   calibration.py:204-206 (C-order over the inlier mask)
//   parameter block order / enable flags              calibration.py:146-161
#pragma once
#include <algorithm>
#include <stdexcept>
#include <string>
#include <memory>
#include <thread>
#include <utility>
#include <vector>
#include "../../include/mcba.h"
#include "mcba_device.h"

namespace mcba {

// The lowering walks every table slot three times (masks, frame-major re-layout, inlier tables): 2.6 M slots at the
// north-star rig, 6.5 M at BASELINE configs[3].  The outer loops are independent, so they are spread over a few host
// threads (the handle is created once per Calibration; this is the PCIe-inclusive part of the boundary).
template <typename F>
inline void lower_parallel_for(int n, F&& fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int nt = std::max(1, std::min({n, 16, (int)(hw ? hw : 1)}));
  if (nt == 1) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve(nt);
  for (int t = 0; t < nt; ++t)
    pool.emplace_back([=, &fn]() {
      for (int i = t; i < n; i += nt) fn(i);
    });
  for (auto& th : pool) th.join();
}


struct LowerError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
#define MCBA_REQUIRE(cond, msg)                  \
  do {                                           \
    if (!(cond)) throw ::mcba::LowerError(msg);  \
  } while (0)

// vector whose resize() leaves new elements uninitialised: the slot tables are tens of MB and every element is written by
// the (threaded) lowering loops anyway -- value-initialising them first cost a serial memset plus all the page faults
template <typename T>
struct NoInitAlloc : std::allocator<T> {
  template <typename U> struct rebind { using other = NoInitAlloc<U>; };
  template <typename U, typename... A>
  void construct(U* ptr, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)ptr) U;
    else ::new ((void*)ptr) U(std::forward<A>(a)...);
  }
};
template <typename T> using SlotVec = std::vector<T, NoInitAlloc<T>>;

struct HostProblem {
  Dims d{};
  SlotVec<uint8_t> valid_ref;        // Calibration.valid, [C,F,B,P] reference order, all frames
  SlotVec<uint8_t> evalid_ref;       // proj.valid & obs.valid
  SlotVec<double2> obs;              // frame-major shard tables ...
  SlotVec<uint8_t> evalid, inlier;
  SlotVec<int32_t> obs_index;
  std::vector<uint8_t> fix_aspect;
  std::vector<int32_t> view_count, board_off, full2act;
  std::vector<double> xfull, bwg, img_h;
  std::vector<uint16_t> tri;
  int64_t n_inliers = 0;
  // ragged camera blocks (mcba_problem.camera_n_dist): the caller's vectors use the reference's layout, camera c carries
  // 5 + cam_nd[c] entries; the library pads every camera to 5 + ND.  ext2int[i] = internal index of entry i of the caller's
  // active vector; int2ext = the inverse (-1 for the padded coefficients).  Empty for uniform rigs (identity).
  std::vector<int32_t> cam_nd, ext2int, int2ext;
  std::vector<uint32_t> cam_kmask;
  int n_ext = 0;                     // length of the caller's active vector (== d.n for uniform rigs)
};

inline int64_t full_size_of(const mcba_problem* p) {
  int64_t nb = 0;
  for (int b = 0; b < p->n_boards; ++b) nb += p->board_sizes[b];
  const int64_t nm = p->motion == MCBA_MOTION_STATIC ? 6LL * p->n_frames
                     : p->motion == MCBA_MOTION_ROLLING ? 12LL * p->n_frames : 12LL;
  int64_t ncam = (int64_t)p->n_cameras * (5 + p->n_dist);
  if (p->camera_n_dist != nullptr) {   // ragged camera blocks (the reference's own layout)
    ncam = 0;
    for (int c = 0; c < p->n_cameras; ++c) ncam += 5 + p->camera_n_dist[c];
  }
  return 6LL * p->n_cameras + 6LL * p->n_boards + nm + ncam + 3 * nb;
}

// (re)build inlier table, residual ordering and per-view counts of the shard; mask in reference order or null
inline void lower_inliers(HostProblem& hp, const uint8_t* mask_ref) {
  const Dims& d = hp.d;
  const size_t nslot = (size_t)d.slots();
  hp.inlier.resize(nslot);       // every element is written below
  hp.obs_index.resize(nslot);
  hp.view_count.assign((size_t)d.views(), 0);
  // residual order = reference C-order over (c, f, b, p) restricted to the shard's frames.  Pass 1 (parallel over
  // (c, f)): inlier bytes and per-view counts; the exclusive prefix of the counts in C-order gives every view its first
  // residual index; pass 2 (parallel) numbers the observations.
  const int nv = d.C * d.Fl;
  std::vector<int64_t> first((size_t)nv * d.B + 1, 0);     // [c][fl][b] view offsets in reference order
  lower_parallel_for(nv, [&](int cf) {
    const int c = cf / d.Fl, fl = cf % d.Fl, f = d.f0 + fl;
    for (int b = 0; b < d.B; ++b) {
      const size_t ref0 = (((size_t)c * d.F + f) * d.B + b) * d.P;
      const size_t v = ((size_t)fl * d.C + c) * d.B + b;
      int vc = 0;
      for (int p = 0; p < d.P; ++p) {
        const uint8_t in = mask_ref ? mask_ref[ref0 + p] : hp.valid_ref[ref0 + p];
        hp.inlier[v * d.P + p] = in ? 1 : 0;
        vc += in ? 1 : 0;
      }
      hp.view_count[v] = vc;
      first[(size_t)cf * d.B + b + 1] = vc;
    }
  });
  for (size_t i = 1; i < first.size(); ++i) first[i] += first[i - 1];
  const int64_t count = first.back();
  lower_parallel_for(nv, [&](int cf) {
    const int c = cf / d.Fl, fl = cf % d.Fl;
    for (int b = 0; b < d.B; ++b) {
      const size_t v = ((size_t)fl * d.C + c) * d.B + b;
      int64_t idx = first[(size_t)cf * d.B + b];
      for (int p = 0; p < d.P; ++p) hp.obs_index[v * d.P + p] = hp.inlier[v * d.P + p] ? (int32_t)idx++ : -1;
    }
  });
  MCBA_REQUIRE(count < (1LL << 30), "too many observations for 32-bit residual indices");
  hp.n_inliers = count;
}

// shape, index maps and the small parameter tables (everything but the per-slot tables): what mcba_create needs on the
// host -- the slot tables are built on the device (k_lower_view)
inline void lower_dims(const mcba_problem* p, HostProblem& hp) {
  MCBA_REQUIRE(p != nullptr, "null problem");
  MCBA_REQUIRE(p->version == MCBA_VERSION, "mcba_problem.version mismatch");
  MCBA_REQUIRE(p->n_cameras > 0 && p->n_frames > 0 && p->n_boards > 0 && p->n_points > 0, "empty problem");
  MCBA_REQUIRE((p->points != nullptr) != (p->points_f32 != nullptr), "exactly one of points / points_f32 must be given");
  MCBA_REQUIRE(p->point_valid && p->board_sizes && p->camera_valid && p->frame_valid && p->board_valid &&
                   p->x_full && p->image_heights && p->fix_aspect,
               "null array in mcba_problem");
  MCBA_REQUIRE(p->motion >= 0 && p->motion <= 2, "unknown motion model");
  MCBA_REQUIRE(p->motion != MCBA_MOTION_HAND_EYE || p->base_wrt_gripper, "hand-eye motion needs base_wrt_gripper");
  MCBA_REQUIRE(p->optimize != 0, "no parameter block enabled");
  MCBA_REQUIRE(p->n_points <= 65535, "boards with more than 65535 points are not supported (16-bit point lists)");
  MCBA_REQUIRE((int64_t)p->n_cameras * p->n_boards < 65535, "more than 65534 (camera, board) pairs are not supported (16-bit view ranks)");
  // projection family of every camera: uniform (camera_model) or per camera (camera_fisheye: the reference's ParamList holds
  // independent Camera / CameraFisheye objects, optimization/parameters.py:54-85)
  bool any_fish = p->camera_model == MCBA_CAMERA_FISHEYE, any_pin = !any_fish;
  if (p->camera_fisheye != nullptr) {
    any_fish = any_pin = false;
    for (int c = 0; c < p->n_cameras; ++c) (p->camera_fisheye[c] ? any_fish : any_pin) = true;
  }
  const bool mixed = any_fish && any_pin;
  auto cam_is_fish = [&](int c) { return p->camera_fisheye != nullptr ? p->camera_fisheye[c] != 0 : any_fish; };
  if (!mixed && any_fish)
    MCBA_REQUIRE(p->n_dist == 4, "fisheye cameras carry 4 distortion coefficients (camera_fisheye.py:113-117)");
  else
    MCBA_REQUIRE(p->n_dist == 4 || p->n_dist == 5 || p->n_dist == 8 || p->n_dist == 12 || p->n_dist == 14,
                 "pinhole cameras carry 4, 5, 8, 12 or 14 distortion coefficients (cv2.projectPoints)");

  Dims& d = hp.d;
  d.cam_kmask = nullptr;
  d.shard_rank = d.shard_world = 0;   // (mcba_set_shard_rank / mcba_rccl_init)
  // per-camera distortion sizes: a ParamList of independent Camera objects (optimization/parameters.py:54-85)
  bool ragged = false;
  hp.cam_nd.assign((size_t)p->n_cameras, p->n_dist);
  if (p->camera_n_dist != nullptr) {
    int mx = 0;
    for (int c = 0; c < p->n_cameras; ++c) {
      const int nd = p->camera_n_dist[c];
      if (cam_is_fish(c))
        MCBA_REQUIRE(nd == 4, "fisheye cameras carry 4 distortion coefficients (camera_fisheye.py:113-117)");
      else
        MCBA_REQUIRE(nd == 4 || nd == 5 || nd == 8 || nd == 12 || nd == 14,
                     "pinhole cameras carry 4, 5, 8, 12 or 14 distortion coefficients (cv2.projectPoints)");
      hp.cam_nd[c] = nd;
      mx = std::max(mx, nd);
      ragged = ragged || nd != p->n_dist;
    }
    MCBA_REQUIRE(mx == p->n_dist, "n_dist must be the largest entry of camera_n_dist");
    MCBA_REQUIRE(!ragged || !any_fish || mixed,
                 "cameras of different distortion sizes must all be pinhole cameras (fisheye cameras carry exactly 4)");
  } else if (mixed) {
    MCBA_REQUIRE(p->n_dist == 4, "a rig that mixes pinhole and fisheye cameras needs camera_n_dist unless every camera carries 4 coefficients");
  }
  // A MIXED rig runs the one instantiation that decides the family per camera at run time; it is compiled for the widest
  // coefficient block (14) and every camera is padded to it, exactly like pinhole cameras of different sizes are padded to
  // the largest (the coefficients a camera does not have stay zero and are frozen: cam_kmask)
  const int nd_internal = mixed ? MAX_DIST : p->n_dist;
  if (mixed) ragged = true;
  d.C = p->n_cameras; d.F = p->n_frames; d.B = p->n_boards; d.P = p->n_points;
  d.f0 = 0; d.Fl = d.F;
  if (p->frame_begin >= 0) {
    MCBA_REQUIRE(0 <= p->frame_begin && p->frame_begin <= p->frame_end && p->frame_end <= d.F, "bad frame shard");
    d.f0 = p->frame_begin;
    d.Fl = p->frame_end - p->frame_begin;
  }
  d.motion = p->motion; d.ND = nd_internal; d.fisheye = mixed ? 2 : (any_fish ? 1 : 0);
  int64_t nboardpts = 0;
  std::vector<int32_t> board_off(d.B + 1, 0);
  for (int b = 0; b < d.B; ++b) {
    MCBA_REQUIRE(p->board_sizes[b] >= 0 && p->board_sizes[b] <= d.P, "board size exceeds n_points");
    nboardpts += p->board_sizes[b];
    board_off[b + 1] = (int32_t)nboardpts;
  }
  d.n_motion = d.motion == MOTION_STATIC ? 6 * d.F : d.motion == MOTION_ROLLING ? 12 * d.F : 12;
  const int sizes[5] = {6 * d.C, 6 * d.B, d.n_motion, d.C * (5 + d.ND), (int)(3 * nboardpts)};
  const uint32_t bits[5] = {MCBA_OPT_CAMERA_POSES, MCBA_OPT_BOARD_POSES, MCBA_OPT_MOTION, MCBA_OPT_CAMERAS,
                            MCBA_OPT_BOARDS};
  int foff[5], aoff[5], nf = 0, na = 0;
  for (int k = 0; k < 5; ++k) {
    foff[k] = nf;
    nf += sizes[k];
    if (p->optimize & bits[k]) { aoff[k] = na; na += sizes[k]; } else aoff[k] = -1;
  }
  d.nfull = nf; d.n = na;
  d.foff_campose = foff[0]; d.foff_boardpose = foff[1]; d.foff_motion = foff[2]; d.foff_cameras = foff[3];
  d.foff_boards = foff[4];
  d.off_campose = aoff[0]; d.off_boardpose = aoff[1]; d.off_motion = aoff[2]; d.off_cameras = aoff[3];
  d.off_boards = aoff[4];
  d.KI = (p->optimize & MCBA_OPT_CAMERAS) ? 4 + d.ND : 0;
  d.NPB = d.motion == MOTION_STATIC ? 3 : 4;
  d.DE = d.motion == MOTION_ROLLING ? 12 : 6;
  d.NV = d.DE + d.KI + 1;
  d.NL = 6 * d.NPB + d.KI;
  d.N1 = d.NL + 1;
  d.rec_size = d.N1 * (d.N1 + 1) / 2;
  d.rec_stride = (d.rec_size + 2 + 1) / 2 * 2;
  d.DF = ((p->optimize & MCBA_OPT_MOTION) && d.motion != MOTION_HAND_EYE) ? (d.motion == MOTION_ROLLING ? 12 : 6) : 0;
  d.ns = d.DF > 0 ? d.n - d.n_motion : d.n;
  d.loss = 0; d.f_scale = 1.0;
  d.pose_cam = 0; d.pose_board = d.C; d.pose_motion = d.C + d.B;
  d.n_pose = d.C + d.B + d.n_motion / 6;
  MCBA_REQUIRE((int64_t)d.slots() < (1LL << 31), "observation table too large for 32-bit slot indices");

  // ---- parameter tables -------------------------------------------------------------------------------------
  {
    std::vector<int32_t> f2a((size_t)d.nfull, -1);
    for (int k = 0; k < 5; ++k)
      if (aoff[k] >= 0)
        for (int i = 0; i < sizes[k]; ++i) f2a[foff[k] + i] = aoff[k] + i;
    hp.full2act = f2a;
    hp.n_ext = d.n;
    if (!ragged) {
      hp.xfull.assign(p->x_full, p->x_full + d.nfull);
    } else {
      // the caller's vectors are ragged in the cameras block: scatter x_full into the padded layout (absent coefficients
      // are zero), and build the maps of the ACTIVE vector
      hp.xfull.assign((size_t)d.nfull, 0.0);
      hp.cam_kmask.assign((size_t)d.C, 0u);
      std::vector<int32_t> fe2i;   // caller's full index -> internal full index
      for (int j = 0; j < d.foff_cameras; ++j) fe2i.push_back(j);
      for (int c = 0; c < d.C; ++c) {
        for (int q = 0; q < 5 + hp.cam_nd[c]; ++q) fe2i.push_back(d.foff_cameras + c * (5 + d.ND) + q);
        for (int k = hp.cam_nd[c]; k < d.ND; ++k) hp.cam_kmask[c] |= 1u << (4 + k);
      }
      for (int j = d.foff_boards; j < d.nfull; ++j) fe2i.push_back(j);
      for (size_t j = 0; j < fe2i.size(); ++j) hp.xfull[fe2i[j]] = p->x_full[j];
      hp.int2ext.assign((size_t)d.n, -1);
      for (size_t j = 0; j < fe2i.size(); ++j) {
        const int a = f2a[fe2i[j]];
        if (a >= 0) {
          hp.int2ext[a] = (int32_t)hp.ext2int.size();
          hp.ext2int.push_back(a);
        }
      }
      hp.n_ext = (int)hp.ext2int.size();
    }
    hp.board_off = board_off;
    hp.img_h.assign(p->image_heights, p->image_heights + d.C);
    hp.fix_aspect.resize((size_t)d.C);   // bit 0: Camera.fix_aspect, bit 1: fisheye camera (read by the mixed-rig kernels)
    for (int c = 0; c < d.C; ++c) hp.fix_aspect[c] = (uint8_t)((p->fix_aspect[c] ? 1 : 0) | (cam_is_fish(c) ? 2 : 0));
    std::vector<double> bwg;
    if (d.motion == MOTION_HAND_EYE) {
      bwg.resize((size_t)12 * d.F);
      for (int f = 0; f < d.F; ++f) {
        const double* m = p->base_wrt_gripper + 16 * (size_t)f;
        for (int i = 0; i < 3; ++i) {
          for (int j = 0; j < 3; ++j) bwg[12 * f + 3 * i + j] = m[4 * i + j];
          bwg[12 * f + 9 + i] = m[4 * i + 3];
        }
      }
    }
    hp.bwg = bwg;
    std::vector<uint16_t> tri((size_t)d.rec_size);
    size_t e = 0;
    for (int i = 0; i < d.N1; ++i)
      for (int j = i; j < d.N1; ++j) tri[e++] = (uint16_t)((i << 8) | j);
    hp.tri = tri;
  }
}

// the complete host lowering incl. the per-slot tables: used by tests/hostmath (the CPU build of the device functions);
// the product builds the slot tables on the device
inline void lower_problem(const mcba_problem* p, HostProblem& hp) {
  lower_dims(p, hp);
  const Dims& d = hp.d;

  // ---- masks (Calibration.valid, calibration.py:69-76; tables.reprojection_error mask, tables.py:244-249) ----
  const size_t nref = (size_t)d.C * d.F * d.B * d.P;
  hp.valid_ref.resize(nref);
  hp.evalid_ref.resize(nref);
  lower_parallel_for(d.C * d.F, [&](int cf) {
    const int c = cf / d.F, f = cf % d.F;
    for (int b = 0; b < d.B; ++b) {
      const bool pv = p->camera_valid[c] && p->frame_valid[f] && p->board_valid[b];
      const size_t r0 = (((size_t)c * d.F + f) * d.B + b) * d.P;
      for (int q = 0; q < d.P; ++q) {
        const bool v = pv && p->point_valid[r0 + q];
        hp.valid_ref[r0 + q] = v;
        hp.evalid_ref[r0 + q] = v && q < p->board_sizes[b];
      }
    }
  });

  // ---- frame-major observation tables of the shard ----------------------------------------------------------
  const size_t nslot = (size_t)d.slots();
  {
    SlotVec<double2>& obs = hp.obs;
    SlotVec<uint8_t>& ev = hp.evalid;
    obs.resize(nslot);
    ev.resize(nslot);
    lower_parallel_for(d.Fl, [&](int fl) {
      for (int c = 0; c < d.C; ++c)
        for (int b = 0; b < d.B; ++b) {
          const size_t r0 = (((size_t)c * d.F + d.f0 + fl) * d.B + b) * d.P;
          const size_t s0 = (((size_t)fl * d.C + c) * d.B + b) * d.P;
          for (int q = 0; q < d.P; ++q) {
            obs[s0 + q].x = p->points ? p->points[2 * (r0 + q)] : (double)p->points_f32[2 * (r0 + q)];
            obs[s0 + q].y = p->points ? p->points[2 * (r0 + q) + 1] : (double)p->points_f32[2 * (r0 + q) + 1];
            ev[s0 + q] = hp.evalid_ref[r0 + q];
          }
        }
    });
  }
  lower_inliers(hp, p->inlier_mask);
}

}  // namespace mcba
