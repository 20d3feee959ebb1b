// mcba_solver_kernels.h -- kernels that do not depend on the camera model: table preparation, assembly of the
// per-view records into the block normal equations, and the damped normal-equation solve of the trust-region
// driver (Schur elimination of the per-frame blocks, dense Cholesky, vector updates).  Included only by mcba_api.hip.
#pragma once
#include <type_traits>
#include <vector>
#include <utility>
#include "mcba_kernels.h"
#include "mcba_trmath.h"

namespace mcba {

// ---------------------------------------------------------------------------------------------------------------
// k_prep: x -> device tables (replaces the object re-construction of Calibration.with_param_vec,
//         optimization/calibration.py:164-171 -> pose_set.py:55-57, camera.py:157-171, board/charuco.py:116-117)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_prep(Dims d, Tables t, const double* __restrict__ x) {
  prep_item(d, t, x, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// k_views: chain matrix board -> camera per view (tables.expand_views + transform_points, tables.py:284-304,400-405;
//          hand-eye: motion/hand_eye.py:43-46).  Rolling shutter stores two chains (start, end).
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_views(Dims d, Tables t) {
  view_item(d, t, blockIdx.x * blockDim.x + threadIdx.x);
}

// That of every non-empty view: tmat[v][a][j], a < DE, j < 6 NPB.  A workgroup owns TMV consecutive views and gives each
// 32 lanes: lane j < 6 NPB evaluates column j with the lane-uniform construction the fused k_linearize uses
// (fused_view_tables: the chain products are formed by every lane, the pose block of the lane is picked with selects --
// the earlier thread-per-(view, pose block) form ran its four block cases one after the other on one wavefront: 12.9 k of
// the workgroup's 26 k cycles, measured with s_memtime stamps), lanes 6 NPB and 6 NPB + 1 the chain matrices of the view
// table.  The columns go through LDS so that the table is written with coalesced stores.  The kernel opens every
// linearisation, so it also zeroes the two accumulation targets of the assembly that follows ([g | diag | cost] and H_ss).
//
// x != nullptr: the kernel ALSO replaces k_prep.  Every workgroup forms the pose entries its views need (all cameras,
// all boards, the frames it touches) in LDS straight from x -- a handful of Rodrigues evaluations instead of a kernel
// boundary and a dependent read of the pose table -- and the blocks behind the view blocks (blockIdx >= nb_views) write
// the global pose / camera / board-point tables for the kernels that follow (k_linearize reads the camera and board-point
// tables, k_cost / k_points the pose table).  TM_LOCAL_POSES bounds the local table (the host checks the shape).
constexpr int TMV = 8;               // views per workgroup
constexpr int TM_THREADS = 32 * TMV;
constexpr int TM_LOCAL_POSES = 64;   // pose entries of the workgroup-local table
__host__ __device__ inline int tmat_local_poses(const Dims& d) {   // upper bound of the entries one workgroup needs
  const int CB = d.C * d.B;
  const int nfl = (TMV - 1) / (CB > 0 ? CB : 1) + 2;
  return d.C + d.B + (d.motion == MOTION_HAND_EYE ? 2 : (d.motion == MOTION_ROLLING ? 2 : 1) * nfl);
}
// LOCAL: the pose entries of the workgroup's views live in LDS -- formed from x, or (x == nullptr) copied from the pose
// table -- so that the chain products read them with LDS instructions; a pointer that may be either global or LDS made
// every read a flat load (9.3 k -> cycles for the two steps).  LOCAL = false (a rig whose cameras + boards exceed the local
// table; x must be nullptr): entries are read from the global pose table.
template <bool LOCAL>
__global__ __launch_bounds__(TM_THREADS, 4) void k_tmat(Dims d, Tables t, double* __restrict__ zero_a, int na,
                                                     double* __restrict__ zero_b, int nb, const double* __restrict__ x,
                                                     int nb_views) {
  __shared__ double tile[TMV * 12 * 24];
  __shared__ double lpose[TM_LOCAL_POSES * POSE_STRIDE];
  __shared__ double pre[TMV * 2 * PRE_STRIDE];
  __shared__ uint8_t vlive[TMV];
  const int NPB = d.NPB, npc = 6 * NPB, DE = d.DE, vsz = DE * npc;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(MCBA_EXP_TMAT_PROF)
  long long st[6] = {0, 0, 0, 0, 0, 0};
  st[0] = clock64();
#endif
  for (int e = i; e < na; e += gridDim.x * blockDim.x) zero_a[e] = 0.0;
  for (int e = i; e < nb; e += gridDim.x * blockDim.x) zero_b[e] = 0.0;
  if ((int)blockIdx.x >= nb_views) {   // table blocks (only launched with x): the body of k_prep
    prep_item(d, t, x, ((int)blockIdx.x - nb_views) * blockDim.x + threadIdx.x);
    return;
  }
  const int v0 = blockIdx.x * TMV, nv = min(TMV, d.views() - v0);
  if (nv <= 0) return;
  const int vl = threadIdx.x >> 5, j = threadIdx.x & 31, v = v0 + vl;
  const bool live = vl < nv && t.view_count[v] != 0;   // (the flag load overlaps the pose entries)
  if (j == 0) vlive[vl] = live;
  PoseSrc ps = global_pose_src(d, t);
  if constexpr (LOCAL) {
    const int CB = d.C * d.B, nch = d.motion == MOTION_ROLLING ? 2 : 1;
    const int f_lo = d.f0 + v0 / CB, nfl = (d.f0 + (v0 + nv - 1) / CB) - f_lo + 1;
    const int nmot = d.motion == MOTION_HAND_EYE ? 2 : nch * nfl, np = d.C + d.B + nmot;
    if (x != nullptr) {
      for (int e = threadIdx.x; e < np; e += blockDim.x) {
        int oa, of, r;
        if (e < d.C) { oa = d.off_campose; of = d.foff_campose; r = 6 * e; }
        else if (e < d.C + d.B) { oa = d.off_boardpose; of = d.foff_boardpose; r = 6 * (e - d.C); }
        else {
          const int li = e - d.C - d.B;
          oa = d.off_motion;
          of = d.foff_motion;
          r = d.motion == MOTION_HAND_EYE ? 6 * li : 6 * ((li / nfl) * d.F + f_lo + li % nfl);
        }
        double rt[6];
        for (int k = 0; k < 6; ++k) rt[k] = block_value(t, x, oa, of, r + k);
        pose_entry(rt, lpose + (size_t)e * POSE_STRIDE);
      }
    } else {   // copy the entries from the pose table (same local order)
      for (int q = threadIdx.x; q < np * POSE_STRIDE; q += blockDim.x) {
        const int e = q / POSE_STRIDE, k = q - e * POSE_STRIDE;
        int gi;
        if (e < d.C) gi = d.pose_cam + e;
        else if (e < d.C + d.B) gi = d.pose_board + (e - d.C);
        else {
          const int li = e - d.C - d.B;
          gi = d.pose_motion + (d.motion == MOTION_HAND_EYE ? li : (li / nfl) * d.F + f_lo + li % nfl);
        }
        lpose[q] = t.pose[(size_t)gi * POSE_STRIDE + k];
      }
    }
    ps.chain = nfl;
    ps.f0 = f_lo;
  }
  // (LOCAL: the three tables are addressed as lpose + ... below, so that the compiler sees LDS pointers)
  const double* cam_tab = LOCAL ? lpose : ps.cam;
  const double* board_tab = LOCAL ? lpose + (size_t)d.C * POSE_STRIDE : ps.board;
  const double* mot_tab = LOCAL ? lpose + (size_t)(d.C + d.B) * POSE_STRIDE : ps.mot;
  __syncthreads();   // (local pose entries and the liveness flags written)
#if defined(MCBA_EXP_TMAT_PROF)
  st[1] = clock64();
#endif
  // step 1: the chain prefixes of the workgroup's views, one thread per (view, chain); R2 | o is the view table's chain
  // matrix (the trial step that led here only ran k_prep; k_cost forms its own chains with the same two products)
  const int nch = d.motion == MOTION_ROLLING ? 2 : 1;
  const bool hand_eye = d.motion == MOTION_HAND_EYE;
  if ((int)threadIdx.x < TMV * nch) {
    const int pv = threadIdx.x / nch, ch = threadIdx.x % nch, vp = v0 + pv;
    if (vlive[pv]) {
      const int b = vp % d.B, c = (vp / d.B) % d.C, f = d.f0 + vp / (d.B * d.C);
      double* out = t.view + (size_t)vp * d.view_stride() + ch * VIEW_STRIDE;
      if (hand_eye) {
        view_chain_p(d, cam_tab + (size_t)c * POSE_STRIDE, board_tab + (size_t)b * POSE_STRIDE, mot_tab,
                     mot_tab + POSE_STRIDE, t.bwg + 12 * (size_t)f, out);
      } else {
        double* pr = pre + (pv * 2 + ch) * PRE_STRIDE;
        view_prefix(cam_tab + (size_t)c * POSE_STRIDE, mot_tab + (size_t)(ch * ps.chain + (f - ps.f0)) * POSE_STRIDE,
                    board_tab + (size_t)b * POSE_STRIDE, pr);
        for (int i = 0; i < 12; ++i) out[i] = pr[12 + i];
      }
    }
  }
  __syncthreads();
  // step 2: lane j < 6 NPB of a view's 32 lanes forms column j of That
  if (live && j < npc) {
    const int b = v % d.B, c = (v / d.B) % d.C, f = d.f0 + v / (d.B * d.C);
    const double* Pc = cam_tab + (size_t)c * POSE_STRIDE;
    const double* Pb = board_tab + (size_t)b * POSE_STRIDE;
    const double* Pm0 = mot_tab + (size_t)(f - ps.f0) * POSE_STRIDE;
    const double* pv = pre + vl * 2 * PRE_STRIDE;
    double* Tm = tile + vl * vsz;
    if (hand_eye) view_column_p(d, Pc, Pb, mot_tab, mot_tab + POSE_STRIDE, t.bwg + 12 * (size_t)f, j, Tm + j, npc);
    else if (nch == 2) that_column_from_prefix<true>(Pc, Pm0, Pm0 + (size_t)ps.chain * POSE_STRIDE, Pb, pv, j, Tm, npc);
    else that_column_from_prefix<false>(Pc, Pm0, Pm0, Pb, pv, j, Tm, npc);
  }
  __syncthreads();
#if defined(MCBA_EXP_TMAT_PROF)
  st[2] = clock64();
#endif
  // views of a workgroup are contiguous: coalesced stores; empty views (45 % of the north-star rig) are skipped -- nobody
  // reads their That, and the table is the largest thing this kernel writes (18 MB for all views)
  double* tg = t.tmat + (size_t)v0 * vsz;
  const float inv_vsz = 1.0f / (float)vsz;
  for (int e = threadIdx.x; e < nv * vsz; e += blockDim.x) {
    int vv = (int)((float)e * inv_vsz);   // view of element e (float quotient, corrected)
    if (vv * vsz > e) --vv; else if ((vv + 1) * vsz <= e) ++vv;
    if (vlive[vv]) tg[e] = tile[e];
  }
#if defined(MCBA_EXP_TMAT_PROF)
  st[3] = clock64();
  if (t.dbg != nullptr && threadIdx.x == 0) {
    long long* o = t.dbg + ((size_t)d.views() + blockIdx.x) * 8;
    o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3];
    o[4] = wall_clock64();
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// assembly of the records (deterministic gathers; no atomics)
// ---------------------------------------------------------------------------------------------------------------
// per-frame blocks: H_ff [Fl][DF][DF], H_fs [Fl][DF][ns], g / diag entries of the frame's parameters.
//
// Every sum runs over records of the frame's C B views.  Gathering them entry by entry from memory made each output
// element a chain of dependent round trips, and working out which entry goes where (local_to_x, the packed-triangle
// index, a dozen integer divisions per element) cost more ALU time than the sums.  Both are gone:
//   * the entries a frame block needs from a record are two short ranges of the packed triangle,
//       part 1   (l, 6 + dd), l < 6 (camera pose x frame):  6 runs of DF entries
//       part 2   rows 6 .. 6 + DF - 1 in full (frame x frame | later pose blocks | intrinsics | residual): ONE contiguous run
//     the workgroup copies those NE entries of `gviews` records at a time into LDS -- independent loads, eleven in flight
//     per thread, nothing read for an empty view (45 % of the rig's views);
//   * what to sum and where it goes is a table built once by mcba_create (frame_table below): per output element the
//     staged entry k, the arithmetic sequence of views (first, stride, count) and the destination.  The same table serves
//     every frame (the frame only shifts the destinations of its own parameters).
MCBA_HD int frame_entries(const Dims& d) {   // NE: entries of one record a frame block reads
  return 6 * d.DF + (tri_index(6 + d.DF, 6 + d.DF, d.N1) - tri_index(6, 6, d.N1));
}
enum { FT_HFS = 0, FT_HFF = 1, FT_GRAD = 2 };
// element: x = staged entry k, y = view sequence (ft_seq: first view, stride 1 or B, count B, C or C B -- no limit on the
// number of (camera, board) pairs), z = destination, w = kind | (dd + 1) << 8 for a diagonal element of H_ff (whose value
// is also diag[x index of the frame parameter dd])
enum { FT_CNT_B = 0, FT_CNT_C = 1, FT_CNT_CB = 2 };
MCBA_HD int ft_seq(int first, bool stride_b, int count_sel) { return first | ((stride_b ? 1 : 0) << 24) | (count_sel << 25); }
inline std::vector<int4> frame_table(const Dims& d) {
  std::vector<int4> tab;
  if (d.DF == 0) return tab;
  const int DF = d.DF, N1 = d.N1, NL = d.NL, ns = d.ns, CW = 6 + d.KI;
  const int base2 = tri_index(6, 6, N1), P1 = 6 * DF;
  auto kidx = [&](int a, int b) {
    if (a > b) std::swap(a, b);
    return a < 6 ? a * DF + (b - 6) : P1 + tri_index(a, b, N1) - base2;
  };
  for (int c = 0; c < d.C; ++c)            // frame x camera(c): sum over boards
    for (int dd = 0; dd < DF; ++dd)
      for (int q = 0; q < CW; ++q) {
        const int li = q < 6 ? q : 6 * d.NPB + (q - 6);
        const int gi = local_to_x(d, 0, c, 0, li);
        if (gi < 0) continue;
        tab.push_back(make_int4(kidx(li, 6 + dd), ft_seq(c * d.B, false, FT_CNT_B), dd * ns + d.x_to_shared(gi), FT_HFS));
      }
  for (int b = 0; b < d.B; ++b)            // frame x board(b): sum over cameras
    for (int dd = 0; dd < DF; ++dd)
      for (int q = 0; q < 6; ++q) {
        const int li = 6 * (d.NPB - 1) + q;
        const int gi = local_to_x(d, 0, 0, b, li);
        if (gi < 0) continue;
        tab.push_back(make_int4(kidx(6 + dd, li), ft_seq(b, true, FT_CNT_C), dd * ns + d.x_to_shared(gi), FT_HFS));
      }
  for (int dd = 0; dd < DF; ++dd)          // frame x frame and the gradient: sum over all views
    for (int d2 = 0; d2 <= DF; ++d2) {
      const int y = ft_seq(0, false, FT_CNT_CB);
      if (d2 < DF) tab.push_back(make_int4(kidx(6 + dd, 6 + d2), y, dd * DF + d2, FT_HFF | (d2 == dd ? (dd + 1) << 8 : 0)));
      else tab.push_back(make_int4(kidx(6 + dd, NL), y, dd, FT_GRAD));
    }
  return tab;
}

__device__ __forceinline__ void assemble_frame_block(const Dims& d, const Tables& t, int fl, int gviews,
                                                     const int4* __restrict__ tab, int ntab,
                                                     const double* __restrict__ rec, double* __restrict__ Hff,
                                                     double* __restrict__ Hfs, double* __restrict__ g,
                                                     double* __restrict__ diag, double* __restrict__ stage) {
  // `gviews` = staging SLOTS.  Only the frame's NON-EMPTY views are staged, compactly: slot[view] = rank of the view among
  // the frame's active views (0xFFFF = empty).  A rig with many views per frame stages one group as long as the active views
  // of the frame fit the slots (16 x 1000 x 5: 80 views per frame, about a third of them active -- staging a slot for every
  // view took three barrier-separated groups per frame: 57 us); a frame with more active views than slots takes several
  // passes over slot ranges.
  const int f = d.f0 + fl;
  const int DF = d.DF, ns = d.ns, N1 = d.N1, CB = d.C * d.B;
  const int base2 = tri_index(6, 6, N1), NE = frame_entries(d), P1 = 6 * DF;
  int* soff = reinterpret_cast<int*>(stage + (size_t)gviews * NE);   // record offset of staged entry k
  uint16_t* slot = reinterpret_cast<uint16_t*>(soff + NE);            // [C B] staging rank of a view, 0xFFFF = empty
  uint16_t* vlist = slot + ((CB + 3) & ~3);                           // [C B] active views in rank order
  __shared__ int n_act_s;
  double* hfs = Hfs + (size_t)fl * DF * ns;
  double* hff = Hff + (size_t)fl * DF * DF;
  if (threadIdx.x < 64) {   // ranks of the active views: ballot prefix over chunks of 64 views (first wavefront)
    int basec = 0;
    for (int e0 = 0; e0 < CB; e0 += 64) {
      const int e = e0 + (int)threadIdx.x;
      const bool on = e < CB && t.view_count[fl * CB + e] != 0;
      const unsigned long long m = __ballot(on);
      const int rk = basec + __popcll(m & ((1ull << threadIdx.x) - 1ull));
      if (e < CB) slot[e] = on ? (uint16_t)rk : (uint16_t)0xFFFF;
      if (on) vlist[rk] = (uint16_t)e;
      basec += __popcll(m);
    }
    if (threadIdx.x == 0) n_act_s = basec;
  }
  for (int k = threadIdx.x; k < NE; k += blockDim.x) soff[k] = k < P1 ? tri_index(k / DF, 6 + k % DF, N1) : base2 + (k - P1);
  // H_fs of the frame is cleared only where the table below does not store every element anyway: adjusted board points (their
  // columns belong to k_points) and rigs with frozen distortion coefficients (columns nobody writes).  Otherwise every
  // (frame parameter, shared parameter) element is the `*dst = sum` of exactly one thread -- the blanket clear wrote the
  // 6.7 MB of H_fs twice per launch at the north-star rig.
  if (d.off_boards >= 0 || d.cam_kmask != nullptr)
    for (int e = threadIdx.x; e < DF * ns; e += blockDim.x) hfs[e] = 0.0;
  const double* rf = rec + (size_t)fl * CB * d.rec_stride;   // records of this frame: view (c, b) at (c B + b) rec_stride
  const float inv_ne = 1.0f / (float)NE;
  constexpr int LB = 11, TB = 7;   // loads / table elements in flight per thread
  int4 q0[TB];                     // first batch of table elements: in flight together with the activity flags
#pragma unroll
  for (int u = 0; u < TB; ++u) {
    const int e = threadIdx.x + u * blockDim.x;
    q0[u] = e < ntab ? tab[e] : make_int4(0, 0, 0, -1);
  }
  __syncthreads();   // (slot, vlist, n_act_s, soff, the zeroes of H_fs written)
  const int n_act = n_act_s;
  for (int s0 = 0; s0 == 0 || s0 < n_act; s0 += gviews) {   // (one pass unless the frame has more active views than slots)
    const int ng = min(gviews, n_act - s0), tot = max(ng, 0) * NE;
    if (s0 > 0) __syncthreads();   // (previous pass consumed)
    for (int i0 = threadIdx.x; i0 < tot; i0 += LB * blockDim.x) {
      double val[LB];
#pragma unroll
      for (int u = 0; u < LB; ++u) {
        const int idx = i0 + u * blockDim.x;
        val[u] = 0.0;
        if (idx < tot) {
          int gv = (int)((float)idx * inv_ne), k = idx - gv * NE;   // idx = gv NE + k (float quotient, corrected)
          if (k < 0) { --gv; k += NE; } else if (k >= NE) { ++gv; k -= NE; }
          val[u] = rf[(size_t)vlist[s0 + gv] * d.rec_stride + soff[k]];
        }
      }
#pragma unroll
      for (int u = 0; u < LB; ++u) {
        const int idx = i0 + u * blockDim.x;
        if (idx < tot) stage[idx] = val[u];
      }
    }
    __syncthreads();
    // every output element belongs to one thread for all passes: later passes add to what the thread stored before
    for (int e0 = threadIdx.x; e0 < ntab; e0 += TB * blockDim.x) {
      int4 q[TB];
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        const int e = e0 + u * blockDim.x;
        q[u] = e0 == (int)threadIdx.x ? q0[u] : (e < ntab ? tab[e] : make_int4(0, 0, 0, -1));
      }
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        if (q[u].w < 0) continue;
        const int k = q[u].x, gv0 = q[u].y & 0xFFFFFF, gst = ((q[u].y >> 24) & 1) ? d.B : 1;
        const int csel = q[u].y >> 25, cnt = csel == FT_CNT_B ? d.B : (csel == FT_CNT_C ? d.C : CB);
        double s0a = 0.0, s1a = 0.0;
        int i = 0;
        for (; i + 2 <= cnt; i += 2) {
          const int ra = (int)slot[gv0 + i * gst] - s0, rb = (int)slot[gv0 + (i + 1) * gst] - s0;   // (0xFFFF - s0 >= ng)
          if (ra >= 0 && ra < ng) s0a += stage[ra * NE + k];
          if (rb >= 0 && rb < ng) s1a += stage[rb * NE + k];
        }
        if (i < cnt) {
          const int ra = (int)slot[gv0 + i * gst] - s0;
          if (ra >= 0 && ra < ng) s0a += stage[ra * NE + k];
        }
        double sum = s0a + s1a;
        const int kind = q[u].w & 255;
        double* dst = kind == FT_HFS ? hfs + q[u].z : (kind == FT_HFF ? hff + q[u].z : g + d.frame_to_x(f, q[u].z));
        if (s0 > 0) sum += *dst;
        *dst = sum;
        if ((q[u].w >> 8) != 0) diag[d.frame_to_x(f, (q[u].w >> 8) - 1)] = sum;
      }
    }
  }
}

// shared part, stage 1: partial[pair=(c,b)][chunk][rec_stride] = sum of the records over a chunk of frames.  The first
// wavefront compacts the chunk's non-empty views (records of empty views are never written); every thread then owns up to
// two entries of the record and keeps the loads of eight views in flight for each.
__device__ __forceinline__ void shared_partial_block(const Dims& d, const Tables& t, int pair, int ch,
                                                     const double* __restrict__ rec, int nchunk,
                                                     double* __restrict__ partial) {
  __shared__ int vsel[64];
  __shared__ int nsel;
  const int c = pair / d.B, b = pair % d.B;
  const int per = (d.Fl + nchunk - 1) / nchunk;
  const int fa = ch * per, fb = min(d.Fl, fa + per);
  const int rs = d.rec_stride, nt = blockDim.x;
  double* out = partial + ((size_t)pair * nchunk + ch) * rs;
  if (fa >= fb) {   // (more chunks than frames)
    for (int e = threadIdx.x; e < rs; e += nt) out[e] = 0.0;
    return;
  }
  for (int f0 = fa; f0 < fb; f0 += 64) {        // (chunks hold at most a few dozen frames: one pass)
    const int nf = min(64, fb - f0);
    __syncthreads();
    if (threadIdx.x < 64) {
      const int v = ((f0 + (int)threadIdx.x) * d.C + c) * d.B + b;
      const bool on = (int)threadIdx.x < nf && t.view_count[v] != 0;
      const unsigned long long m = __ballot(on);
      if (on) vsel[__popcll(m & ((1ull << threadIdx.x) - 1ull))] = v;
      if (threadIdx.x == 0) nsel = __popcll(m);
    }
    __syncthreads();
    const int n = nsel;
    // The rows of the eliminated frame parameters -- a contiguous block of the packed triangle, 270 of 597 entries at the
    // north-star rig -- are no business of the shared part (k_shared_final drops them): walked over, not read and summed.
    const int skip0 = d.DF > 0 ? tri_index(6, 6, d.N1) : rs, nskip = d.DF > 0 ? d.DF * d.N1 - (6 * d.DF + d.DF * (d.DF - 1) / 2) : 0;
    for (int q0 = threadIdx.x; q0 < rs - nskip; q0 += 2 * nt) {
      const int q1 = q0 + nt;
      const int e0 = q0 < skip0 ? q0 : q0 + nskip, e1 = q1 < skip0 ? q1 : q1 + nskip;
      const bool h1 = q1 < rs - nskip;
      double s0 = 0.0, s1 = 0.0;
      if (f0 != fa) {
        s0 = out[e0];
        if (h1) s1 = out[e1];
      }
      int k = 0;
      for (; k + 8 <= n; k += 8) {
        double a0[8], a1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double* r = rec + (size_t)vsel[k + u] * rs;
          a0[u] = r[e0];
          a1[u] = h1 ? r[e1] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          s0 += a0[u];
          s1 += a1[u];
        }
      }
      if (k < n) {   // remainder: up to seven views, loads still issued together
        double a0[7], a1[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
          const bool on = k + u < n;
          const double* r = rec + (size_t)vsel[on ? k + u : k] * rs;
          a0[u] = on ? r[e0] : 0.0;
          a1[u] = on && h1 ? r[e1] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) {
          s0 += a0[u];
          s1 += a1[u];
        }
      }
      out[e0] = s0;
      if (h1) out[e1] = s1;
    }
  }
}

// ONE launch for both gathers (they are independent and each is too small to fill the chip):
//   blocks [0, nfb)                 frame blocks (nfb = Fl when per-frame parameters are eliminated, else 0)
//   blocks [nfb, nfb + C B nchunk)  chunk sums of the shared part
constexpr int ASM_THREADS = 512;
__global__ __launch_bounds__(ASM_THREADS) void k_assemble(Dims d, Tables t, const double* __restrict__ rec, int nfb, int nchunk,
                                                  int gviews, const int4* __restrict__ ftab, int nftab,
                                                  double* __restrict__ Hff, double* __restrict__ Hfs,
                                                  double* __restrict__ g, double* __restrict__ diag,
                                                  double* __restrict__ partial) {
  extern __shared__ double asm_stage[];   // [slots][NE] staged record entries of a frame block + [NE] record offsets + view ranks
  if ((int)blockIdx.x < nfb) {
    assemble_frame_block(d, t, blockIdx.x, gviews, ftab, nftab, rec, Hff, Hfs, g, diag, asm_stage);
  } else {
    const int q = blockIdx.x - nfb;
    shared_partial_block(d, t, q / nchunk, q % nchunk, rec, nchunk, partial);
  }
}

// shared part, stage 2: H_ss (dense ns x ns, zeroed by the caller), g, diag and the total cost from the chunk sums.
// A block covers SF_ENT (16) packed local entries e x ALL pairs (c, b): the wavefront of a pair first adds the pair's chunk sums
// (entry = lane & 15, chunks split over the four rows of the wavefront, 8 independent loads in flight per lane) into LDS.  A local entry maps to an element of H_ss that depends on the camera
// only (camera pose / intrinsics columns), on the board only (board pose), on both, or on neither (hand-eye blocks):
// the thread of the FIRST pair of each equivalence class owns the element, adds the pair sums of its class from LDS in
// a fixed order and stores -- no atomics, no read-modify-write chains, every element written once.
// The cost of a linearisation written straight into the host's pinned memory by the ONE thread that forms it, followed by a
// sequence number with system-scope release (the scalars that earlier kernels of the iteration published the same way landed
// at their kernel boundaries): the driver of an LM iteration spins on that number -- no copy, no event, no extra launch.
__device__ __forceinline__ void publish_cost(double val, double* host_cost, unsigned long long* host_seq,
                                             unsigned long long seq) {
  host_cost[0] = val;
  __threadfence_system();
  __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr int SHARED_FINAL_MAX_PAIRS = 128;   // pair sums of k_shared_final: [pairs][SF_ENT] doubles of LDS
// Round 6: 16 packed entries per workgroup instead of 64 (38 workgroups instead of 10 at the north-star rig: the 2.4 MB of chunk sums are
// pulled by four times as many CUs) and the chunks of a pair split over the four 16-lane rows of its wavefront (8 loads per lane in one
// batch instead of 32), folded with two lane-xor exchanges.
// (written as ONE kernel body on purpose: the same body as a __forceinline__ device function shared with a second kernel -- round 6's merged
//  k_assemble + final-stage launch, measured and dropped, profiles/r06_lsmr_experiments.txt item 9 -- compiled to 46 instead of 98 VGPRs: the
//  chunk-sum loads per pair were no longer kept in flight together and the launch took 8.6 instead of 6.2 us)
constexpr int SF_ENT = 16;   // packed entries per workgroup
__global__ __launch_bounds__(1024) void k_shared_final(Dims d, const double* __restrict__ partial, int nchunk,
                                                       const uint16_t* __restrict__ tri, double* __restrict__ Hss,
                                                       double* __restrict__ g, double* __restrict__ diag,
                                                       double* __restrict__ cost_count, double* host_cost = nullptr,
                                                       unsigned long long* host_seq = nullptr, unsigned long long seq = 0) {
  extern __shared__ double pair_sum[];   // [C B][SF_ENT]
  const int ns = d.ns, NL = d.NL, npose = 6 * d.NPB, npair = d.C * d.B;
  const int el = threadIdx.x & (SF_ENT - 1), cq = (threadIdx.x & 63) >> 4, wv = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int e = blockIdx.x * SF_ENT + el;
  // (the rows of the eliminated frame parameters hold no chunk sums: shared_partial_block walks over them)
  const int skip0 = d.DF > 0 ? tri_index(6, 6, d.N1) : 0, nskip = d.DF > 0 ? d.DF * d.N1 - (6 * d.DF + d.DF * (d.DF - 1) / 2) : 0;
  const bool in = e < d.rec_size + 2 && !(e >= skip0 && e < skip0 + nskip);
  const size_t rs = d.rec_stride;
  const int ij = e < d.rec_size ? tri[e] : 0;   // (issued with the chunk sums, used after them)
  for (int pair = wv; pair < npair; pair += NW) {
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.0;
    if (in) {
      // row cq of the wavefront takes chunks cq, cq + 4, ...: unconditional loads (clamped index, 0 / 1 weight), eight in flight
      const double* base = partial + (size_t)pair * nchunk * rs + e;
      for (int c0 = 0; c0 < nchunk; c0 += 32)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = c0 + 4 * u + cq;
          a[u] += (c < nchunk ? 1.0 : 0.0) * base[(size_t)min(c, nchunk - 1) * rs];
        }
    }
    double sum = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);            // (commutative pairings: the four rows hold the same bits)
    if (cq == 0) pair_sum[pair * SF_ENT + el] = sum;
  }
  __syncthreads();
  if (!in) return;
  bool depc = false, depb = false;
  int i = 0, j = 0;
  if (e < d.rec_size) {
    i = ij >> 8;
    j = ij & 255;
    if (i == NL) return;                                   // (r, r) = sum f^2: the cost is carried separately
    if (local_is_frame(d, i) || local_is_frame(d, j)) return;
    auto dep_c = [&](int l) { return l < 6 || l >= npose; };
    auto dep_b = [&](int l) { return l >= npose - 6 && l < npose; };
    depc = dep_c(i) || (j < NL && dep_c(j));
    depb = dep_b(i) || (j < NL && dep_b(j));
  }
  for (int pair = threadIdx.x / SF_ENT; pair < npair; pair += blockDim.x / SF_ENT) {
    const int c = pair / d.B, b = pair % d.B;
    if ((!depc && c != 0) || (!depb && b != 0)) continue;    // not the first pair of its class
    int gi = -1, gj = -1;
    if (e < d.rec_size) {
      // the frame index is irrelevant for shared parameters (hand-eye blocks do not depend on f either)
      gi = local_to_x(d, 0, c, b, i);
      if (gi < 0) continue;
      if (j < NL) {
        gj = local_to_x(d, 0, c, b, j);
        if (gj < 0) continue;
      }
    }
    const int c0 = depc ? c : 0, c1 = depc ? c + 1 : d.C, b0 = depb ? b : 0, b1 = depb ? b + 1 : d.B;
    double val = 0.0;
    for (int cc = c0; cc < c1; ++cc)
      for (int bb = b0; bb < b1; ++bb) val += pair_sum[(cc * d.B + bb) * SF_ENT + el];
    if (e >= d.rec_size) {
      cost_count[e - d.rec_size] = val;
      if (host_cost != nullptr && e == d.rec_size) publish_cost(val, host_cost, host_seq, seq);
    } else if (j == NL) {
      g[gi] = val;
    } else {
      const int si = d.x_to_shared(gi), sj = d.x_to_shared(gj);
      Hss[(size_t)si * ns + sj] = val;
      if (si != sj) Hss[(size_t)sj * ns + si] = val;
      else diag[gi] = val;
    }
  }
}

// The same reduction for rigs with MORE (camera, board) pairs than the LDS pair sums of k_shared_final hold (C B > 128,
// e.g. 16 cameras x 10 boards): the owner thread of an element adds the chunk sums of all pairs of its class straight from
// memory, class by class in the same fixed order.  grid = (entry blocks, pair groups of 16), block = 64 x 16.
__global__ __launch_bounds__(1024) void k_shared_final_big(Dims d, const double* __restrict__ partial, int nchunk,
                                                           const uint16_t* __restrict__ tri, double* __restrict__ Hss,
                                                           double* __restrict__ g, double* __restrict__ diag,
                                                           double* __restrict__ cost_count, double* host_cost = nullptr,
                                                           unsigned long long* host_seq = nullptr, unsigned long long seq = 0) {
  const int ns = d.ns, NL = d.NL, npose = 6 * d.NPB, npair = d.C * d.B;
  const int el = threadIdx.x & 63, pair = blockIdx.y * 16 + (threadIdx.x >> 6);
  const int e = blockIdx.x * 64 + el;
  if (e >= d.rec_size + 2 || pair >= npair) return;
  const size_t rs = d.rec_stride;
  bool depc = false, depb = false;
  int i = 0, j = 0;
  if (e < d.rec_size) {
    const int ij = tri[e];
    i = ij >> 8;
    j = ij & 255;
    if (i == NL) return;
    if (local_is_frame(d, i) || local_is_frame(d, j)) return;
    auto dep_c = [&](int l) { return l < 6 || l >= npose; };
    auto dep_b = [&](int l) { return l >= npose - 6 && l < npose; };
    depc = dep_c(i) || (j < NL && dep_c(j));
    depb = dep_b(i) || (j < NL && dep_b(j));
  }
  const int c = pair / d.B, b = pair % d.B;
  if ((!depc && c != 0) || (!depb && b != 0)) return;        // not the first pair of its class
  int gi = -1, gj = -1;
  if (e < d.rec_size) {
    gi = local_to_x(d, 0, c, b, i);
    if (gi < 0) return;
    if (j < NL) {
      gj = local_to_x(d, 0, c, b, j);
      if (gj < 0) return;
    }
  }
  const int c0 = depc ? c : 0, c1 = depc ? c + 1 : d.C, b0 = depb ? b : 0, b1 = depb ? b + 1 : d.B;
  double val = 0.0;
  for (int cc = c0; cc < c1; ++cc)
    for (int bb = b0; bb < b1; ++bb) {
      const double* base = partial + (size_t)(cc * d.B + bb) * nchunk * rs + e;
      double s0 = 0.0, s1 = 0.0;
      int ch = 0;
      for (; ch + 2 <= nchunk; ch += 2) {
        s0 += base[(size_t)ch * rs];
        s1 += base[(size_t)(ch + 1) * rs];
      }
      if (ch < nchunk) s0 += base[(size_t)ch * rs];
      val += s0 + s1;
    }
  if (e >= d.rec_size) {
    cost_count[e - d.rec_size] = val;
    if (host_cost != nullptr && e == d.rec_size) publish_cost(val, host_cost, host_seq, seq);
  } else if (j == NL) {
    g[gi] = val;
  } else {
    const int si = d.x_to_shared(gi), sj = d.x_to_shared(gj);
    Hss[(size_t)si * ns + sj] = val;
    if (si != sj) Hss[(size_t)sj * ns + si] = val;
    else diag[gi] = val;
  }
}

// diag(H) of the shared parameters from H_ss (only needed after k_points added the board-point blocks)
__global__ void k_shared_diag(Dims d, const double* __restrict__ Hss, double* __restrict__ diag) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < d.ns; s += gridDim.x * blockDim.x)
    diag[d.shared_to_x(s)] = Hss[(size_t)s * d.ns + s];
}

// dense J^T J in x order from the block form (debug / parity tests)
__global__ void k_dense_hessian(Dims d, const double* __restrict__ Hss, const double* __restrict__ Hfs,
                                const double* __restrict__ Hff, double* __restrict__ H) {
  const size_t n = d.n;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n * n; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / n), j = (int)(e % n);
    const int si = d.x_to_shared(i), sj = d.x_to_shared(j);
    double val = 0.0;
    auto frame_of = [&](int xi, int& fl, int& dd) {
      int o = xi - d.off_motion;
      int half = 0;
      if (o >= 6 * d.F) { o -= 6 * d.F; half = 1; }
      fl = o / 6 - d.f0;
      dd = o % 6 + 6 * half;
    };
    if (si >= 0 && sj >= 0) {
      val = Hss[(size_t)si * d.ns + sj];
    } else if (si < 0 && sj < 0) {
      int fi, di, fj, dj;
      frame_of(i, fi, di);
      frame_of(j, fj, dj);
      if (fi == fj && fi >= 0 && fi < d.Fl) val = Hff[((size_t)fi * d.DF + di) * d.DF + dj];
    } else {
      int fl, dd;
      frame_of(si < 0 ? i : j, fl, dd);
      const int s = si < 0 ? sj : si;
      if (fl >= 0 && fl < d.Fl) val = Hfs[((size_t)fl * d.DF + dd) * d.ns + s];
    }
    H[e] = val;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// trust-region driver kernels.  Vectors of length n live on the device and are complete (replicated) on every
// rank of a frame-sharded problem; only H-dependent partial sums are reduced across ranks.
// ---------------------------------------------------------------------------------------------------------------
// scipy compute_jac_scale (common.py:598-611) + g_h = d * g, one element per thread.
// part[3 blk + {0,1,2}] = {|g|_inf, |g_h|^2, |x * scale_inv|^2} of the block (the host folds the blocks: the vectors are
// complete on every rank, so there is nothing to all-reduce); block 0 forwards {cost, count} of the linearisation.
__global__ __launch_bounds__(256) void k_vec_scale(Dims d, const double* __restrict__ x, const double* __restrict__ g,
                                                   const double* __restrict__ diag, double* __restrict__ scale_inv,
                                                   double* __restrict__ dsc, double* __restrict__ gh, int first,
                                                   double* __restrict__ part, const double* __restrict__ cost_count,
                                                   double* __restrict__ cost_out) {
  __shared__ double scratch[16];
  double mx = 0, gg = 0, xs = 0;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.n) {
    double si = sqrt(diag[i]);
    if (first) { if (si == 0.0) si = 1.0; }
    else si = fmax(si, scale_inv[i]);
    scale_inv[i] = si;
    const double di = 1.0 / si;
    dsc[i] = di;
    const double gi = g[i];
    gh[i] = di * gi;
    const double w = d.entry_weight(i);   // (1 unless frame-sharded: every entry counts once in the sum over the ranks)
    mx = fabs(gi);
    gg = w * (di * gi * di * gi);
    xs = w * (x[i] * si * x[i] * si);
  }
  const double a = block_reduce<true>(mx, scratch);
  const double b = block_reduce<false>(gg, scratch);
  const double c = block_reduce<false>(xs, scratch);
  if (threadIdx.x == 0) {
    part[3 * blockIdx.x + 0] = a;
    part[3 * blockIdx.x + 1] = b;
    part[3 * blockIdx.x + 2] = c;
    if (blockIdx.x == 0 && cost_count) { cost_out[0] = cost_count[0]; cost_out[1] = cost_count[1]; }
  }
}

// q = u^T (D H D) u for ONE vector, as a plain streaming weighted sum over the stored blocks (grid-stride, coalesced):
//   q = sum_{f,dd,s} 2 a_f[dd] H_fs[f][dd][s] a_s[s] + sum_{f,dd,d2} a_f[dd] H_ff[f][dd][d2] a_f[d2]
//     + sum_{i,j} a_s[i] H_ss[i][j] a_s[j],            a = D u.
// The trust-region driver only needs this for u = g_h (Cauchy curvature): the forms involving the Gauss-Newton step
// follow from (D H D + reg I) gn = g_h without touching H again (mcba_solve).  partial[blockIdx.x] = block sum
// (folded by k_tr_reg; a frame-sharded handle all-reduces the partial array element-wise first).
// Cauchy curvature q = w^T H w over the blocks of H, w_i = g_i / s_i^2: per-block partial sums.  H_fs and H_ff are walked a
// wavefront per ROW (row = one eliminated frame parameter): w of the row once per wavefront, lanes over the shared columns
// (coalesced, no integer division per element -- the element-per-thread form spent more on e / ns and on the two w per element
// than on the loads: 7.4 us standalone, 13.4 us inside the merged launch).  w_x(x index) / w_s(shared index) are supplied by the
// caller: k_q00 reads the scaled gradient k_vec_scale left behind, k_vec_scale_q00 forms it on the fly.
template <class WX, class WS>
__device__ __forceinline__ double q00_partial(const Dims& d, const double* __restrict__ Hss, const double* __restrict__ Hfs,
                                              const double* __restrict__ Hff, WX w_x, WS w_s, int qb, int nqb) {
  const int ns = d.ns, DF = d.DF, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int rows = DF > 0 ? d.Fl * DF : 0;
  double q = 0.0;
  const int T = nqb * nwave;
  for (int row0 = qb * nwave + wave; row0 < rows; row0 += 64 * T) {
    // w of up to 64 rows of this wavefront at once, one per lane (three dependent loads, a square root and a division when it
    // is formed on the fly: paid once, not per row), then handed out with a shuffle
    const int myrow = row0 + lane * T;
    const double wmine = myrow < rows ? w_x(d.frame_to_x(d.f0 + myrow / DF, myrow % DF)) : 0.0;
    for (int k = 0; k < 64; ++k) {
      const int row = row0 + k * T;
      if (row >= rows) break;
      const int f = d.f0 + row / DF;
      const double wr = __shfl(wmine, k, 64);
      const double* hr = Hfs + (size_t)row * ns;
      double acc = 0.0;
      for (int sidx = lane; sidx < ns; sidx += 64) acc += hr[sidx] * w_s(sidx);
      q += 2.0 * wr * acc;
      if (lane < DF) q += wr * Hff[(size_t)row * DF + lane] * w_x(d.frame_to_x(f, lane));
    }
  }
  for (int e = qb * blockDim.x + threadIdx.x; e < ns * ns; e += nqb * blockDim.x) {
    const int i = e / ns, j = e - i * ns;
    q += w_s(i) * Hss[e] * w_s(j);
  }
  return q;
}

__global__ __launch_bounds__(256) void k_q00(Dims d, const double* __restrict__ Hss, const double* __restrict__ Hfs,
                                             const double* __restrict__ Hff, const double* __restrict__ dsc,
                                             const double* __restrict__ u, double* __restrict__ partial) {
  __shared__ double scratch[16];
  auto w_x = [&](int xi) { return dsc[xi] * u[xi]; };
  auto w_s = [&](int sidx) { const int xi = d.shared_to_x(sidx); return dsc[xi] * u[xi]; };
  const double q = q00_partial(d, Hss, Hfs, Hff, w_x, w_s, (int)blockIdx.x, (int)gridDim.x);
  const double r = block_reduce<false>(q, scratch);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// k_vec_scale and k_q00 in ONE launch (round 3): the Cauchy curvature g_h^T H_h g_h needs w_i = g_i / s_i^2 of every parameter,
// and s_i = max(sqrt(diag_i), previous scale_i) is three loads and a square root away from the linearisation's [g | diag] --
// nothing k_vec_scale produces.  Blocks [0, nvb) are k_vec_scale, the rest k_q00 with w formed on the fly (the same operations
// in the same order: bit-identical sums).  scale_out may be scale_inv itself (in place): that race is benign -- the vec_scale
// blocks replace scale_i by max(sqrt(diag_i), scale_i), and max(sqrt(diag_i), .) of the old and of the new value coincide --
// or another buffer (the speculative scaling of a trial point, adopted by a pointer swap when the step is accepted).  One launch and one dependent
// kernel boundary less in every LM iteration (k_vec_scale 5.9 us + gap at the north-star rig).
__global__ __launch_bounds__(256) void k_vec_scale_q00(Dims d, const double* __restrict__ x, const double* __restrict__ g,
                                                       const double* __restrict__ diag, const double* scale_inv,
                                                       double* scale_out, double* __restrict__ dsc, double* __restrict__ gh, int first,
                                                       double* __restrict__ part, const double* __restrict__ cost_count,
                                                       double* __restrict__ cost_out, int nvb, const double* __restrict__ Hss,
                                                       const double* __restrict__ Hfs, const double* __restrict__ Hff,
                                                       double* __restrict__ partial) {
  __shared__ double scratch[16];
  if ((int)blockIdx.x < nvb) {
    double mx = 0, gg = 0, xs = 0;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.n) {
      double si = sqrt(diag[i]);
      if (first) { if (si == 0.0) si = 1.0; }
      else si = fmax(si, scale_inv[i]);
      scale_out[i] = si;
      const double di = 1.0 / si;
      dsc[i] = di;
      const double gi = g[i];
      gh[i] = di * gi;
      const double w = d.entry_weight(i);
      mx = fabs(gi);
      gg = w * (di * gi * di * gi);
      xs = w * (x[i] * si * x[i] * si);
    }
    const double a = block_reduce<true>(mx, scratch);
    const double b = block_reduce<false>(gg, scratch);
    const double c = block_reduce<false>(xs, scratch);
    if (threadIdx.x == 0) {
      part[3 * blockIdx.x + 0] = a;
      part[3 * blockIdx.x + 1] = b;
      part[3 * blockIdx.x + 2] = c;
      if (blockIdx.x == 0 && cost_count) { cost_out[0] = cost_count[0]; cost_out[1] = cost_count[1]; }
    }
    return;
  }
  auto w_of = [&](int xi) {
    double si = sqrt(diag[xi]);
    if (first) { if (si == 0.0) si = 1.0; }
    else si = fmax(si, scale_inv[xi]);
    const double di = 1.0 / si;
    return di * (di * g[xi]);
  };
  const int qb = (int)blockIdx.x - nvb, nqb = (int)gridDim.x - nvb;
  const int ns = d.ns;
  // w of the shared parameters once per block
  constexpr int WS_MAX = 2048;
  __shared__ double ws[WS_MAX];
  const bool cached = ns <= WS_MAX;
  if (cached) {
    for (int sidx = threadIdx.x; sidx < ns; sidx += blockDim.x) ws[sidx] = w_of(d.shared_to_x(sidx));
    __syncthreads();
  }
  auto w_shared = [&](int sidx) { return cached ? ws[sidx] : w_of(d.shared_to_x(sidx)); };
  const double q = q00_partial(d, Hss, Hfs, Hff, w_of, w_shared, qb, nqb);
  const double r = block_reduce<false>(q, scratch);
  if (threadIdx.x == 0) partial[qb] = r;
}

// out[0..2] = {u0.u0, u0.u1, u1.u1} over the full (replicated) vectors
__global__ void k_dots3(int n, const double* __restrict__ u0, const double* __restrict__ u1, double* __restrict__ out,
                        const int* __restrict__ info = nullptr) {
  __shared__ double scratch[16];
  double dt[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double a = u0[i], b = u1[i];
    dt[0] += a * a;
    dt[1] += a * b;
    dt[2] += b * b;
  }
  for (int k = 0; k < 3; ++k) {
    const double ds = block_reduce<false>(dt[k], scratch);
    if (threadIdx.x == 0) out[k] = ds;
  }
  if (info != nullptr && threadIdx.x == 0) out[3] = (double)info[0];   // pivot report of the reduced Cholesky
}

// One wavefront: folds the k_vec_scale / k_q00 partials, fixes the trust radius of the first iteration and computes the
// damping of the Gauss-Newton solve; returns it in every lane.  S (may be null): the scalar block that receives the folded
// values for the kernels and the host that read them later.
__device__ __forceinline__ double tr_reg_wave(double* S, const double* __restrict__ vs_part, int nvb,
                                              const double* __restrict__ q_part, int nq, int first, double Delta_in,
                                              int lane) {
  double mx = 0, gg = 0, xs = 0, q = 0;
  for (int b = lane; b < nvb; b += 64) {
    mx = fmax(mx, vs_part[3 * b]);
    gg += vs_part[3 * b + 1];
    xs += vs_part[3 * b + 2];
  }
  for (int b = lane; b < nq; b += 64) q += q_part[b];
  mx = wave_max(mx);
  gg = wave_sum(gg);
  xs = wave_sum(xs);
  q = wave_sum(q);
  double reg = 0.0;
  if (lane == 0) {
    double Delta = Delta_in;
    if (first) {   // trf.py:428-430
      Delta = sqrt(xs);
      if (Delta == 0) Delta = 1.0;
    }
    reg = gg > 0 ? tr_reg_term(q, gg, Delta) : TR_REG_FLOOR;
    if (S != nullptr) {
      S[TR_GNORM] = mx;
      S[TR_GH2] = gg;
      S[TR_XS2] = xs;
      S[TR_Q00] = q;
      S[TR_DELTA] = Delta;
      S[TR_REG] = reg;
    }
  }
  return __shfl(reg, 0, 64);
}

// (Schur step 1 -- the per-frame factor and W = L^-1 D_f H_fs D_s -- is k_schur_frame below, behind the Cholesky tile helpers)

// Schur step 2: partial SYRK  P[split][tile] = sum_{k in split} W'[k][ti*16..]^T W'[k][tj*16..]  over the stacked rows
// k = (frame, dd) of W' = [W | y]  [K x (ns+1)].  One wavefront per (upper tile, K split); MFMA f64 16x16x4 reads its
// operands straight from global memory (row-major W': 128-byte coalesced segments per 16 lanes).
template <bool MFMA>
__global__ __launch_bounds__(64, 2) void k_schur_syrk(int K, int ncol, int ntile, int ksplit, const double* __restrict__ W,
                                                   double* __restrict__ P) {
  // XCD-aware launch: workgroups go to the eight XCDs round-robin by their linear index, so with the SPLIT as the fast grid
  // dimension (ksplit is a multiple of 8) every tile pair of the splits s, s + 8, .. runs on XCD s and that XCD's L2 holds only
  // its eighth of W' -- with the tile as the fast dimension every XCD pulled all of W' from memory (8 x 6.8 MB per launch)
  const int tile = blockIdx.y, split = blockIdx.x, lane = threadIdx.x;
  int ti = 0, rem = tile;
  while (rem >= ntile - ti) { rem -= ntile - ti; ++ti; }
  const int tj = ti + rem;
  const int per = ((K + ksplit - 1) / ksplit + 3) / 4 * 4;
  const int k0 = split * per, k1 = min(K, k0 + per);
  double* out = P + ((size_t)split * (ntile * (ntile + 1) / 2) + tile) * 256;
  const int rsub = lane >> 4, csub = lane & 15;
  const int ci = ti * 16 + csub, cj = tj * 16 + csub;
  if constexpr (MFMA) {
    // Operand loads of 8 MFMA steps per batch, the NEXT batch requested before the MFMAs of the current one are issued (a
    // wavefront that loads, waits, multiplies and loads again spends three quarters of its time in memory round trips: two
    // accumulators so that the eight MFMAs of a batch are not one dependent chain either).  Raw loads of clamped addresses;
    // the 0 / 1 masks are applied when the batch is consumed (see k_schur_syrk3).
    double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    constexpr int UB = 8;
    const double mi = ci < ncol ? 1.0 : 0.0, mj = cj < ncol ? 1.0 : 0.0;
    const int cic = min(ci, ncol - 1), cjc = min(cj, ncol - 1);
    double av[2][UB], bv[2][UB];
    auto request = [&](int buf, int k) {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const size_t row = (size_t)min(k + 4 * u + rsub, K - 1) * ncol;
        av[buf][u] = W[row + cic];
        bv[buf][u] = W[row + cjc];
      }
    };
    auto multiply = [&](int buf, int k) {
#pragma unroll
      for (int u = 0; u < UB; u += 2) {
        const double m0 = (k + 4 * u + rsub < k1) ? mi : 0.0, m1 = (k + 4 * u + 4 + rsub < k1) ? mi : 0.0;
        // (operands swapped: the tile holds S[16 tj + row][16 ti + col], the LOWER-triangle block of the pair -- the Cholesky
        //  kernels read the lower triangle, and k_schur_reduce then walks each tile row by row, coalesced)
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[buf][u] * mj, av[buf][u] * m0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[buf][u + 1] * mj, av[buf][u + 1] * m1, acc1, 0, 0, 0);
      }
    };
    request(0, k0);
    for (int k = k0; k < k1; k += 8 * UB) {
      request(1, k + 4 * UB);
      __builtin_amdgcn_sched_barrier(0);
      multiply(0, k);
      __builtin_amdgcn_sched_barrier(0);
      if (k + 4 * UB >= k1) break;
      request(0, k + 8 * UB);
      __builtin_amdgcn_sched_barrier(0);
      multiply(1, k + 4 * UB);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int r = 0; r < 4; ++r) out[(rsub + 4 * r) * 16 + csub] = acc0[r] + acc1[r];
  } else {
    double acc[4] = {0, 0, 0, 0};
    for (int k = k0; k < k1; ++k) {   // (tile = S[16 tj + row][16 ti + col], as on the matrix pipe)
      const double b = ci < ncol ? W[(size_t)k * ncol + ci] : 0.0;
      for (int r = 0; r < 4; ++r) {
        const int ri = tj * 16 + rsub + 4 * r;
        acc[r] += (ri < ncol ? W[(size_t)k * ncol + ri] : 0.0) * b;
      }
    }
    for (int r = 0; r < 4; ++r) out[(rsub + 4 * r) * 16 + csub] = acc[r];
  }
}

// The same partial SYRK with 3 x 3 REGISTER BLOCKING (round 3), used for reduced systems of nine tile columns or more.
// k_schur_syrk gives every (tile pair, K split) its own wavefront, which loads two 16-column strips of W' per MFMA: 1 KB of
// operands per 64-cycle matrix instruction, 69 MB of L2 reads per launch for a 6.8 MB matrix at the north-star rig.  Here a
// wavefront owns a 48 x 48 block of S (nine tiles, 72 accumulator registers) and loads three + three strips per nine MFMAs:
// 21 MB.  Four wavefronts of a workgroup split the rows of one K split among themselves and fold their accumulators through
// LDS in a fixed order.  P keeps the layout of k_schur_syrk: [split][upper tile][16 x 16].
// What it took (profiles/r03_syrk_experiments.txt): 8 wavefronts x 96 workgroups: 18.9 us (384 SIMDs carry all MFMAs); a
// wave-uniform `if (!diag)` around the B loads: eight serial round trips per batch (16.8); masked_load's multiply at request
// time: vmcnt(0) at the end of every trip; no scheduling barriers: both requests hoisted, all MFMAs behind one wait (12.6);
// the diagonal choice inside multiply(): 503 v_accvgpr_mov per trip (11.3); a `break` in the middle of the trip or a register
// budget above 256: AGPR-form MFMAs whose 72 accumulators are copied to VGPRs and back in every trip (10.9; 31 us at 16
// cameras x 5 boards) -- __launch_bounds__(256, 2) keeps the budget at 256 and the MFMAs in VGPR form: 8.7 / 20.8 us against
// 9.4 / 27.2 us of the one-tile kernel (which gained its own double buffering and the XCD-aware grid order on the way).
constexpr int SYRK3_WAVES = 4, SYRK3_THREADS = 64 * SYRK3_WAVES;
__global__ __launch_bounds__(SYRK3_THREADS, 2) void k_schur_syrk3(int K, int ncol, int ntile, int ksplit,
                                                               const double* __restrict__ W, double* __restrict__ P) {
  __shared__ double red[4][9][256];
  const int nt3 = (ntile + 2) / 3;
  int si = 0, rem = blockIdx.y;                      // (split = fast grid dimension: see k_schur_syrk)
  while (rem >= nt3 - si) { rem -= nt3 - si; ++si; }
  const int sj = si + rem, split = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rsub = lane >> 4, csub = lane & 15;
  // rows of this split, a multiple of 4 per wavefront
  const int per = ((K + ksplit - 1) / ksplit + 4 * SYRK3_WAVES - 1) / (4 * SYRK3_WAVES) * (4 * SYRK3_WAVES);
  const int pw = per / SYRK3_WAVES, k0 = split * per + wave * pw, k1 = min(K, k0 + pw);
  int ca[3], cb[3];
#pragma unroll
  for (int x = 0; x < 3; ++x) {
    ca[x] = (3 * si + x) * 16 + csub;
    cb[x] = (3 * sj + x) * 16 + csub;
  }
  double4_t acc[3][3];
#pragma unroll
  for (int x = 0; x < 3; ++x)
#pragma unroll
    for (int y = 0; y < 3; ++y) acc[x][y] = double4_t{0.0, 0.0, 0.0, 0.0};
  // Operand loads of four K steps (24 loads) per batch, and the NEXT batch is requested before the 36 MFMAs of the current one
  // are issued (the wavefront has its SIMD to itself: nothing else would cover the round trip).  The B strips are loaded on
  // the diagonal too -- same addresses as the A strips, so they hit L1 -- because a wave-uniform `if (!diag)` inside the
  // unrolled batch made hipcc branch around every group of three loads and drain vmcnt behind each (eight serial round trips
  // per batch: 16.8 us).
  // (The loads are RAW reads of clamped addresses; the 0 / 1 masks of rows behind the split and columns behind the matrix are
  //  applied when a batch is consumed: masked_load's multiply at request time made hipcc wait for the next batch at the end of
  //  every trip, vmcnt(0).)
  constexpr int UB = 2;                              // (3: 9.9 instead of 8.7 us at the north-star rig; 4: spills)
  double av[2][UB][3], bv[2][UB][3];
  double ma[3], mb[3];
#pragma unroll
  for (int x = 0; x < 3; ++x) {
    ma[x] = ca[x] < ncol ? 1.0 : 0.0;
    mb[x] = cb[x] < ncol ? 1.0 : 0.0;
    ca[x] = min(ca[x], ncol - 1);
    cb[x] = min(cb[x], ncol - 1);
  }
  auto request = [&](int buf, int k) {
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const size_t row = (size_t)min(k + 4 * u + rsub, K - 1) * ncol;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        av[buf][u][x] = W[row + ca[x]];
        bv[buf][u][x] = W[row + cb[x]];
      }
    }
  };
  // (The diagonal / off-diagonal choice is made ONCE around the whole loop: as a wave-uniform branch inside multiply() it made
  //  the 72 accumulator registers phi nodes of every batch -- 503 v_accvgpr_mov per trip, each waiting for the MFMA that wrote
  //  its source: 31 us at 16 cameras x 5 boards.)
  auto run = [&](auto diag_tag) {
    constexpr bool DIAGB = decltype(diag_tag)::value;
    auto multiply = [&](int buf, int k) {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const double mk = (k + 4 * u + rsub < k1) ? 1.0 : 0.0;
        double a[3], b[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          a[x] = av[buf][u][x] * (mk * ma[x]);
          b[x] = bv[buf][u][x] * mb[x];
        }
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
          for (int y = DIAGB ? x : 0; y < 3; ++y)   // (the three tiles below the diagonal of a diagonal block are never stored)
            acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[y], a[x], acc[x][y], 0, 0, 0);   // (lower block: see k_schur_syrk)
      }
    };
    request(0, k0);
    int k = k0;
    // two batches per trip (the buffer index stays a compile-time constant), an odd last batch behind the loop: a `break` in
    // the middle of the trip made hipcc keep the accumulators in VGPRs and copy all 72 to the AGPRs and back in every trip
    for (; k + 4 * UB < k1; k += 8 * UB) {
      // (scheduling barriers: left alone, hipcc moves both requests of a trip to its top and all MFMAs behind one vmcnt(0))
      request(1, k + 4 * UB);
      __builtin_amdgcn_sched_barrier(0);
      multiply(0, k);
      __builtin_amdgcn_sched_barrier(0);
      request(0, k + 8 * UB);
      __builtin_amdgcn_sched_barrier(0);
      multiply(1, k + 4 * UB);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (k < k1) multiply(0, k);
  };
  if (si == sj) run(std::true_type{});
  else run(std::false_type{});
  // fold the four wavefronts through LDS: every thread sums the four partials of its elements in a fixed order
#pragma unroll
  for (int x = 0; x < 3; ++x)
#pragma unroll
    for (int y = 0; y < 3; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][3 * x + y][(rsub + 4 * r) * 16 + csub] = acc[x][y][r];
  __syncthreads();
  const int nt2 = ntile * (ntile + 1) / 2;
  for (int e = tid; e < 9 * 256; e += SYRK3_THREADS) {
    const int t9 = e >> 8, el = e & 255, ti = 3 * si + t9 / 3, tj = 3 * sj + t9 % 3;
    if (ti > tj || tj >= ntile) continue;            // (lower tile of a diagonal block / padding behind the last strip)
    const int tile = ti * ntile - (ti * (ti - 1)) / 2 + (tj - ti);
    P[((size_t)split * nt2 + tile) * 256 + el] = (red[0][t9][el] + red[1][t9][el]) + (red[2][t9][el] + red[3][t9][el]);
  }
}

// Schur step 3:  S = D_s H_ss D_s - W^T W  (reg I is added after the cross-rank reduction),
//                rhs = [own shared gradient] - W^T y.   buf = [S (ns*ns) | rhs (ns)]: rhs is "row ns" of the matrix.
// FOUR threads per element: thread q of a quad sums the splits q, q + 4, ... (four loads in flight each), the quad is
// folded with two shuffles in a fixed order.  One thread per element walked its 32 splits as eight dependent round trips.
__global__ __launch_bounds__(256) void k_schur_reduce(Dims d, const double* __restrict__ Hss, const double* __restrict__ dsc,
                                                      const double* __restrict__ gh, const double* __restrict__ P, int ntile,
                                                      int ksplit, int K, double g_weight, double* __restrict__ buf,
                                                      const double* __restrict__ tr = nullptr) {
  const int ns = d.ns;
  const int nt2 = ntile * (ntile + 1) / 2;
  const int total = ns * ns + ns;
  const int q = threadIdx.x & 3;
  for (int e0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 2; e0 < ((total + 63) & ~63); e0 += (gridDim.x * blockDim.x) >> 2) {
    const bool live = e0 < total;
    const int e = live ? e0 : 0;
    const int i = e / ns, j = e % ns;            // i == ns: the right-hand-side row
    const int a = min(i, j), b = max(i, j);
    const int ti = a / 16, tj = b / 16;
    const int tile = ti * ntile - (ti * (ti - 1)) / 2 + (tj - ti);
    // Only the LOWER triangle (and the right-hand-side row) is formed: nothing reads the rest of buf (it keeps the zeroes of
    // its allocation), and a lower element sits at (row b, column a) of its tile -- consecutive threads read consecutive
    // doubles.  Forming both triangles from upper-triangle tiles read 11.3 MB per launch for 2.9 MB of partial sums.
    const bool wanted = live && j <= i;
    double sum = 0.0;
    if (K > 0 && wanted) {
      const double* pp = P + (size_t)tile * 256 + (b % 16) * 16 + (a % 16);
      const size_t st = (size_t)nt2 * 256;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int sp = q;
      for (; sp + 12 < ksplit; sp += 16) {         // four independent loads in flight, fixed summation order
        s0 += pp[(size_t)sp * st];
        s1 += pp[(size_t)(sp + 4) * st];
        s2 += pp[(size_t)(sp + 8) * st];
        s3 += pp[(size_t)(sp + 12) * st];
      }
      for (; sp < ksplit; sp += 4) s0 += pp[(size_t)sp * st];
      sum = (s0 + s1) + (s2 + s3);
    }
    sum += __shfl_down(sum, 2, 4);                 // (all 64 lanes take part: the element loop is padded to whole waves)
    sum += __shfl_down(sum, 1, 4);
    if (q != 0 || !wanted) continue;
    // (the damping lives on the device, tr[TR_REG]; only the root rank adds it: the buffer is summed over the ranks next)
    if (i < ns) buf[e] = dsc[d.shared_to_x(i)] * Hss[e] * dsc[d.shared_to_x(j)] - sum + ((tr != nullptr && i == j) ? g_weight * tr[TR_REG] : 0.0);
    else buf[e] = g_weight * gh[d.shared_to_x(j)] - sum;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_chol_blk: the reduced system of the usual calibration (ns <= 159: cameras x (pose + intrinsics) + boards) solved by
// ONE workgroup of 8 wavefronts with the whole lower triangle resident in LDS as 16 x 16 tiles (row stride 17).
// (The column-by-column kernel of round 1 spent two 1024-thread barriers per column: 234 us at ns = 140.) here a block
// column costs two barriers:
//   (a) wave 0 factors the diagonal tile entirely in REGISTERS: lane i owns row i, the pivot and the scaled column
//       entries travel through v_readlane (an LDS round trip costs ~250 cycles on this chip and there are ns of them
//       on the critical path), then inverts it the same way, one column of L^-1 per lane,
//   (b) all waves form the panel below as X = A L_kk^-T with v_mfma_f64_16x16x4_f64 (four steps per tile),
//   (c) the symmetric rank-16 update of the trailing tiles, again on the matrix pipe -- with look-ahead: wave 0 updates
//       the next diagonal tile first and runs (a) for it while waves 1-7 update the remaining tiles.
// The right-hand side is row ns of the matrix, so (b)/(c) are also the forward substitution; the backward substitution
// runs on wave 0 alone with the inverted diagonal tiles: p_k = L_kk^-T z_k, z_j -= L_kj^T p_k, no workgroup barrier.
// ---------------------------------------------------------------------------------------------------------------
constexpr int CT = 16, CTL = 17, CTS = CT * CTL;
constexpr int CHOL_BLK_MAX_N1 = 160, CHOL_BLK_THREADS = 512;
__host__ __device__ inline size_t chol_blk_lds_bytes(int ns) {
  const int nb = (ns + 1 + CT - 1) / CT;
  return ((size_t)(nb * (nb + 1) / 2 + nb) * CTS + 3 * nb * CT) * sizeof(double) + 16;
}

__device__ __forceinline__ double lane_bcast(double v, int src) {   // value of lane src (wave-uniform) in every lane
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Cholesky factor of a 16 x 16 tile by one wavefront, in 4 x 4 BLOCKS with the trailing update on the matrix pipe (round 3;
// the column-by-column form it replaced -- lane i owns row i, 16 x (2 + 2 (15 - j)) readlanes and 120 FMAs in one dependent chain --
// took ~310 cycles per pivot, 5 k cycles per tile).  Lane (li, lg) keeps four entries of
// row li -- the columns lg, lg + 4, lg + 8, lg + 12 -- which is exactly the accumulator layout of v_mfma_f64_16x16x4 for the
// symmetric update C -= A A^T (lane (li, lg) receives C[lg + 4 r][li] = C[li][lg + 4 r]) and, for block b, exactly its operand
// layout (lane (li, lg) supplies L[li][4 b + lg]): no data movement between the steps.  Per block: the four columns of the block
// are gathered into every lane (four cross-lane reads), factored redundantly by all lanes (pivot j touches only the columns left
// in the block: 3 + 2 + 1 broadcasts instead of 15 + ... + 12), written back, and ONE rank-4 MFMA updates the rest of the tile.
template <bool FULL>
__device__ __forceinline__ void chol_tile_factor_b4(double* __restrict__ D, double* __restrict__ dinv, int ncol, int col0,
                                                    int lane, int& badcol) {
  const int li = lane & 15, lg = lane >> 4;
  double4_t a4;
#pragma unroll
  for (int r = 0; r < 4; ++r) a4[r] = D[li * CTL + lg + 4 * r];
  double dv[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) dv[j] = 1.0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (FULL || 4 * b < ncol) {                      // wave-uniform
      double blk[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) blk[g] = __shfl(a4[b], li + 16 * g, 64);   // row li, columns 4 b .. 4 b + 3
      // Pivot chain: d_j -> 1 / sqrt(d_j) -> l = a_j / sqrt(d_j) -> d_(j+1) = a_(j+1) - l_(j+1)^2, the last step on the two
      // broadcast values directly (lane j + 1 forms the same fma on the same operands in the update below).  v_rsq_f64 + one
      // third-order correction, no clamp / class selects on the chain: a non-positive pivot is reported through badcol and
      // poisons the factor with NaN (every caller discards the solve then).
      double dn = lane_bcast(blk[0], 4 * b);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * b + jj;
        if (FULL || j < ncol) {
          const double dj = dn;
          badcol = (dj > 0.0 || badcol != 0) ? badcol : col0 + j + 1;
          const double y = __builtin_amdgcn_rsq(dj), e = fma(-y * dj, y, 1.0);
          const double inv = fma(y * e, fma(e, 0.375, 0.5), y);
          dv[j] = inv;
          const double l = blk[jj] * inv;
          blk[jj] = l;
          if (jj < 3) {
            const double apre = lane_bcast(blk[jj + 1], j + 1), ln = lane_bcast(l, j + 1);
            dn = fma(-ln, ln, apre);
          }
          double lk[4];
#pragma unroll
          for (int kk = jj + 1; kk < 4; ++kk) lk[kk] = lane_bcast(l, 4 * b + kk);
#pragma unroll
          for (int kk = jj + 1; kk < 4; ++kk) blk[kk] = fma(-l, lk[kk], blk[kk]);
        }
      }
      // own column of the block back into the accumulator layout
      a4[b] = lg == 0 ? blk[0] : (lg == 1 ? blk[1] : (lg == 2 ? blk[2] : blk[3]));
      if (b < 3) {
        // rest of the tile: C[i][c] -= sum_k L[i][4 b + k] L[c][4 b + k] for i, c >= 4 b + 4 (columns of pivots that do not exist
        // -- the padding behind the matrix -- contribute nothing)
        const bool on = li >= 4 * b + 4 && (FULL || 4 * b + lg < ncol);
        const double op = on ? a4[b] : 0.0;
        a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-op, op, a4, 0, 0, 0);
      }
    }
  }
  // stored SYMMETRICALLY (L below the diagonal, L^T above): the panel solve reads column j of L as the adjacent entries of row j
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = lg + 4 * r;
    if (c <= li) D[li * CTL + c] = a4[r];
    if (c < li) D[c * CTL + li] = a4[r];
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < CT; ++j) dinv[j] = dv[j];
  }
}

// factor only; dinv[0 .. 16) = 1 / L_jj
__device__ __forceinline__ void chol_tile_factor_noinv(double* __restrict__ D, double* __restrict__ dinv, int ncol, int col0,
                                                       int lane, int& badcol) {
  if (ncol >= CT) chol_tile_factor_b4<true>(D, dinv, CT, col0, lane, badcol);
  else chol_tile_factor_b4<false>(D, dinv, ncol, col0, lane, badcol);
}

// X = A L^-T for FOUR 16 x 16 tiles at once by forward substitution: lane (lg, li) owns row li of tile lg (A4[lg], may be
// null).  Nothing crosses lanes; L and 1 / L_jj are read from LDS with wave-uniform addresses (broadcast reads).  Replaces
// "multiply by the inverted diagonal tile", which needed the 16 dependent columns of the inverse on the critical path of every
// block column.  RIGHT-looking: x_j = a_j / L_jj, then a_m -= x_j L_mj for the columns m > j -- the 15 - j updates of a step are
// independent, so the dependent chain is mul -> fma per column (32 operations) where the left-looking form (round 2) summed
// j products in two chains per column (92 operations, 2.0 k cycles per block column).  Column j of L is row j of the factored
// tile above the diagonal (chol_tile_factor_b4 stores L^T there): adjacent entries, requested three steps ahead into rotating
// registers (left to the compiler every broadcast read is followed by its own s_waitcnt).
__device__ __forceinline__ void chol_panel_solve4(const double* __restrict__ L, const double* __restrict__ dinv,
                                                  double* __restrict__ A, int li) {
  if (A == nullptr) return;
  double a[CT], dv[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) a[c] = A[li * CTL + c];
#pragma unroll
  for (int c = 0; c < CT; ++c) dv[c] = dinv[c];
  double lr[4][CT];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int m = r + 1; m < CT; ++m) lr[r][m] = L[r * CTL + m];
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    if (j + 3 < CT) {
#pragma unroll
      for (int m = j + 4; m < CT; ++m) lr[(j + 3) & 3][m] = L[(j + 3) * CTL + m];
    }
    const double x = a[j] * dv[j];
    a[j] = x;
#pragma unroll
    for (int m = j + 1; m < CT; ++m) a[m] = fma(-x, lr[j & 3][m], a[m]);
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) A[li * CTL + c] = a[c];
}

// inverse of a factored 16 x 16 tile (L in LDS with L^T above the diagonal, dinv = 1 / L_jj): lane c < 16 solves L x = e_c by
// the right-looking substitution of chol_panel_solve4 (chain: mul -> fma per row); Xi[i][c] = x_i.  Runs on a wavefront that is
// off the critical path; every L entry is a broadcast read.  Only the leading ncol x ncol part is inverted (the rest of Xi is
// zero): a last tile carries the right-hand-side row and identity padding behind the matrix.
__device__ __forceinline__ void chol_tile_invert(const double* __restrict__ L, const double* __restrict__ dinv,
                                                 double* __restrict__ Xi, int lane, int ncol = CT) {
  const int c = lane & 15;
  double a[CT], dv[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    a[i] = (i == c) ? 1.0 : 0.0;
    dv[i] = (i < ncol) ? dinv[i] : 0.0;
  }
  double lr[4][CT];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int m = r + 1; m < CT; ++m) lr[r][m] = L[r * CTL + m];
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    if (j + 3 < CT) {
#pragma unroll
      for (int m = j + 4; m < CT; ++m) lr[(j + 3) & 3][m] = L[(j + 3) * CTL + m];
    }
    const double x = a[j] * dv[j];
    a[j] = x;
#pragma unroll
    for (int m = j + 1; m < CT; ++m) a[m] = fma(-x, lr[j & 3][m], a[m]);
  }
  if (lane < CT) {
#pragma unroll
    for (int i = 0; i < CT; ++i) Xi[i * CTL + c] = a[i];
  }
}

// C -= Xa Xb^T for one 16 x 16 tile (four MFMA steps), one wavefront
__device__ __forceinline__ void chol_tile_syrk(const double* __restrict__ Xa, const double* __restrict__ Xb,
                                               double* __restrict__ Cm, int li, int lg) {
  double4_t acc;
  double av[4], bv[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    av[s4] = -Xa[li * CTL + 4 * s4 + lg];
    bv[s4] = Xb[li * CTL + 4 * s4 + lg];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = Cm[(lg + 4 * r) * CTL + li];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) Cm[(lg + 4 * r) * CTL + li] = acc[r];
}

// forward-substituted right-hand side = row ns of the factored matrix; then the blocked back substitution on ONE wavefront with
// the inverted diagonal tiles: p_k = L_kk^-T z_k, z_j -= L_kj^T p_k for the blocks above
__device__ __forceinline__ void chol_blk_backsub(int ns, int nb, const double* __restrict__ Lb, const double* __restrict__ Li,
                                                 double* __restrict__ yv, double* __restrict__ ps, int lane) {
  const int li = lane & 15, lg = lane >> 4;
  const int by = ns / CT, ry = ns % CT;
  for (int e = lane; e < nb * CT; e += 64) {
    const int bj = e / CT, c = e % CT;
    yv[e] = (e < ns && bj <= by) ? Lb[(size_t)(by * (by + 1) / 2 + bj) * CTS + ry * CTL + c] : 0.0;
  }
  lds_fence();
  const int nbc = (ns + CT - 1) / CT;
  for (int kb = nbc - 1; kb >= 0; --kb) {
    double p;
    {   // p_k = L_kk^-T z_k as a mat-vec with the inverted tile: lane (c, q) sums the rows i = q, q + 4, .. of column c
      const double* Xk = Li + (size_t)kb * CTS;
      double s4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) s4[r] = Xk[(lg + 4 * r) * CTL + li] * yv[CT * kb + lg + 4 * r];
      p = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      p = fold32(p, p);
      p = fold16(p, p);                        // every lane (c, *) holds p_c
      if (lane < CT && CT * kb + li < ns) ps[CT * kb + li] = p;
    }
    if (kb > 0) {   // z_j -= L_kj^T p_k for the blocks above; p_r as scalars
      double pr[CT];
#pragma unroll
      for (int r = 0; r < CT; ++r) pr[r] = lane_bcast(p, r);
      for (int e = lane; e < CT * kb; e += 64) {
        const int bj = e / CT, c = e % CT;
        const double* Lt = Lb + (size_t)(kb * (kb + 1) / 2 + bj) * CTS;
        double sp[4] = {yv[e], 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < CT; ++r) sp[r & 3] -= Lt[r * CTL + c] * pr[r];
        yv[e] = (sp[0] + sp[1]) + (sp[2] + sp[3]);
      }
      lds_fence();
    }
  }
}

__global__ __launch_bounds__(CHOL_BLK_THREADS) void k_chol_blk(int ns, double reg, const double* __restrict__ buf,
                                                               double* __restrict__ ps, int* __restrict__ info,
                                                               long long* __restrict__ prof) {
  extern __shared__ __attribute__((aligned(16))) double chol_b[];
  constexpr int NTHR = CHOL_BLK_THREADS, NW = NTHR / 64;
  const int n1 = ns + 1, nb = (n1 + CT - 1) / CT, ntile = nb * (nb + 1) / 2;
  double* Lb = chol_b;                         // tiles (bi, bj), bj <= bi, at (bi (bi + 1) / 2 + bj) * CTS
  double* Li = Lb + (size_t)ntile * CTS;       // inverted diagonal tiles
  double* yv = Li + (size_t)nb * CTS;          // forward-substituted right-hand side, updated by the back substitution
  double* pv = yv + nb * CT;                   // solution
  double* dvv = pv + nb * CT;                  // 1 / L_jj
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int badcol = 0;                              // wave 0: first non-positive pivot (1-based)
  long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = 0;   // phase stamps (prof != nullptr): -, a0 || load, b, c + a, back, load of D_0
  if (prof) tc = clock64();
#define CHOL_STAMP(i) if (prof) { const long long now = clock64(); tp[i] += now - tc; tc = now; }
  const int li = lane & 15, lg = lane >> 4;
  {   // load: whole tiles per wavefront, lane (row lg + 4 q, column li); padding rows / columns continue the matrix with the
      // identity.  Wave 0 fetches only the first diagonal tile and factors it while the other seven wavefronts fetch the rest
      // (round 4: the first pivot chain, 4.2 k cycles, used to start behind a barrier after the whole load, ~8 k cycles of
      // mostly latency).
    constexpr int MAXT = (CHOL_BLK_MAX_N1 / CT) * (CHOL_BLK_MAX_N1 / CT + 1) / 2;   // 55 tiles
    constexpr int PER = (MAXT - 1 + NW - 2) / (NW - 1);                             // <= 8 tiles per loading wavefront
    double v[PER][4];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int tile = wave == 0 ? (u == 0 ? 0 : ntile) : 1 + (wave - 1) + (NW - 1) * u;
      int bi = 0;
      while ((bi + 1) * (bi + 2) / 2 <= tile) ++bi;
      const int bj = tile - bi * (bi + 1) / 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int gi = CT * bi + lg + 4 * q, gj = CT * bj + li;
        const bool in = tile < ntile && gi < n1 && gj < ns && gj <= gi;
        v[u][q] = masked_load(buf, (size_t)gi * ns + gj, in) + (in ? ((gi == gj) ? reg : 0.0) : ((gi == gj) ? 1.0 : 0.0));
      }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int tile = wave == 0 ? (u == 0 ? 0 : ntile) : 1 + (wave - 1) + (NW - 1) * u;
      if (tile < ntile) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Lb[(size_t)tile * CTS + (lg + 4 * q) * CTL + li] = v[u][q];
      }
    }
    CHOL_STAMP(5)
  }
  // (the panels are solved with L itself and 1 / L_jj; the INVERSES of the diagonal tiles, which only the back substitution
  //  uses, are formed one block column late by a wavefront with slack: off the critical path)
  if (wave == 0 && ns > 0) {
    lds_fence();
    CHOL_STAMP(0)
    chol_tile_factor_noinv(Lb, dvv, min(CT, ns), 0, lane, badcol);
  }
  __syncthreads();
  CHOL_STAMP(1)
  for (int k = 0; k < nb; ++k) {
    if (ns - CT * k <= 0) break;               // no column of S in this block (right-hand-side row / padding only)
    {
      // (b) panel tiles below the diagonal:  X = A L_kk^-T by forward substitution, four tiles per wavefront
      const int bi = k + 1 + 4 * wave + lg;
      chol_panel_solve4(Lb + (size_t)(k * (k + 1) / 2 + k) * CTS, dvv + k * CT,
                        bi < nb ? Lb + (size_t)(bi * (bi + 1) / 2 + k) * CTS : nullptr, li);
    }
    __syncthreads();
    CHOL_STAMP(2)
    {
      // (c) trailing tiles (bi, bj), k < bj <= bi:  C -= X_bi X_bj^T.  Tile 0 of the enumeration is the next diagonal
      // tile: wave 0 updates it and factors it at once (look-ahead) while waves 1 .. 7 share the other tiles.
      const int m = nb - k - 1, nt = m * (m + 1) / 2;
      if (wave == 0) {
        if (nt > 0) {
          const int b1 = k + 1;
          const double* X1 = Lb + (size_t)(b1 * (b1 + 1) / 2 + k) * CTS;
          double* D1 = Lb + (size_t)(b1 * (b1 + 1) / 2 + b1) * CTS;
          chol_tile_syrk(X1, X1, D1, li, lg);
          lds_fence();
          const int ncol1 = min(CT, ns - CT * b1);
          if (ncol1 > 0) chol_tile_factor_noinv(D1, dvv + b1 * CT, ncol1, CT * b1, lane, badcol);
        }
      } else {
        // the last wavefront inverts the diagonal tile of this block column (about as long as wave 0's look-ahead factor) and
        // takes no trailing tiles
        if (wave == NW - 1)
          chol_tile_invert(Lb + (size_t)(k * (k + 1) / 2 + k) * CTS, dvv + k * CT, Li + (size_t)k * CTS, lane, min(CT, ns - CT * k));
        else for (int tt = wave; tt < nt; tt += NW - 2) {
          int a = 0, rem = tt;
          while (rem > a) { rem -= a + 1; ++a; }           // tt -> (a, rem), rem <= a
          const int bi = k + 1 + a, bj = k + 1 + rem;
          chol_tile_syrk(Lb + (size_t)(bi * (bi + 1) / 2 + k) * CTS, Lb + (size_t)(bj * (bj + 1) / 2 + k) * CTS,
                         Lb + (size_t)(bi * (bi + 1) / 2 + bj) * CTS, li, lg);
        }
      }
    }
    __syncthreads();
    CHOL_STAMP(3)
  }
  if (wave == 0) {
    chol_blk_backsub(ns, nb, Lb, Li, yv, ps, lane);
    CHOL_STAMP(4)
    if (lane == 0) {
      info[0] = badcol;
      if (prof)
        for (int i = 0; i < 8; ++i) prof[i] = tp[i];
    }
  }
#undef CHOL_STAMP
}

constexpr int CHOLP_MAX_N1 = 1024;       // largest reduced system (ns + 1) of the panel kernels below

// ---------------------------------------------------------------------------------------------------------------
// k_cholp_*: PANEL Cholesky for reduced systems that do not fit one workgroup's LDS (ns + 1 > 160: many cameras --
// 16 cameras x 5 boards give ns = 286 -- or adjust_board).  Round 2's single-workgroup kernel kept the matrix in L2 and paid a global round trip
// for every 16-column step on ONE compute unit (265 us at ns = 286: the trailing update of up to 153 tiles per step is the
// critical path).  Here a block column of up to 96 columns (as many 16-column tiles as fit 150 KB of LDS, at most six: the
// panels get wider as the remaining matrix gets shorter -- 286: 3 + 4 + 6 + 5 tile columns, eight launches) lives in LDS while it is factored with the
// register / LDS tile kernels of k_chol_blk, and the trailing update of the rest of the matrix runs on the WHOLE chip:
//   k_cholp_panel   ONE workgroup: block column [c0, c0 + 16 wt) x rows [c0, ns] from L2 into LDS; per tile column the
//                   diagonal tile is factored in registers (wave 0), the tiles below are solved by forward substitution
//                   (four per wavefront), the remaining tile columns of the panel are updated on the matrix pipe (the next
//                   diagonal tile first: wave 0 factors it at once), the diagonal tiles are inverted for the back
//                   substitution at the end; L goes back to memory
//   k_cholp_trail   one wavefront per 16 x 16 tile of the trailing lower triangle:  C -= X_i X_j^T  with K = 16 wt, X from
//                   L2, twelve MFMAs, every load issued before the first one
//   k_cholp_back    one workgroup: p = L^-T y with the stored inverses of the diagonal tiles (a mat-vec per tile, no serial
//                   pivot chain) and the row block of L of the next step requested before the current tile is solved
// Same in-place layout as the other kernels: buf = [S (ns x ns, lower triangle used) | rhs (ns)] = an (ns + 1) x ns matrix
// whose last row becomes the forward-substituted right-hand side.
// ---------------------------------------------------------------------------------------------------------------
constexpr int CHOLP_THREADS = 512, CHOLP_WT = 6, CHOLP_LDS_MAX = 150 * 1024;
__host__ __device__ inline int cholp_panel_tiles(int ns, int kt0) {      // tile columns of the panel that starts at kt0
  const int nb = (ns + 1 + CT - 1) / CT, nbc = (ns + CT - 1) / CT, nbr = nb - kt0;
  int wt = CHOLP_LDS_MAX / (int)(nbr * CTS * sizeof(double));
  wt = wt < 1 ? 1 : (wt > CHOLP_WT ? CHOLP_WT : wt);
  return wt < nbc - kt0 ? wt : nbc - kt0;
}
__host__ __device__ inline size_t cholp_lds_bytes(int ns, int kt0, int wt) {
  const int nb = (ns + 1 + CT - 1) / CT;
  return ((size_t)(nb - kt0) * wt * CTS + (size_t)wt * CTS + wt * CT) * sizeof(double);
}

__global__ __launch_bounds__(CHOLP_THREADS) void k_cholp_panel(int ns, int kt0, int wt, double reg, double* __restrict__ buf,
                                                               double* __restrict__ Linv, int* __restrict__ info,
                                                               long long* __restrict__ prof = nullptr) {
  extern __shared__ __attribute__((aligned(16))) double cholp[];
  constexpr int NW = CHOLP_THREADS / 64;
  const int n1 = ns + 1, nb = (n1 + CT - 1) / CT, nbr = nb - kt0, c0 = CT * kt0;
  double* P = cholp;                              // tile (bl, kk) of the panel at (bl * wt + kk) * CTS, bl = block row - kt0
  double* Xi = P + (size_t)nbr * wt * CTS;        // inverted diagonal tiles [wt]
  double* dinv = Xi + (size_t)wt * CTS;           // 1 / L_jj [wt][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  int badcol = 0;
  long long tp[6] = {0, 0, 0, 0, 0, 0}, tc = 0;   // phase stamps (prof != nullptr): load, first factor, solve, trailing, invert, store
  if (prof) tc = clock64();
#define CHOLP_STAMP(i) if (prof) { const long long now = clock64(); tp[i] += now - tc; tc = now; }
  // ---- load: a wavefront per tile (tile indices are wave-uniform: no per-element index arithmetic -- the first version
  // spent 15-19 k cycles per panel here, three times what the memory system needs for 117 KB, on integer divisions), lane
  // (r = lane / 4, c4 = lane % 4) reads four consecutive entries of row r; the loads of up to eight tiles are in flight
  // per wavefront.  Outside the matrix the identity continues it; the upper triangle is not read.
  {
    const int ntl = nbr * wt, r = lane >> 2, c4 = (lane & 3) * 4;
    constexpr int TB = 8;
    for (int t0 = wave; t0 < ntl; t0 += TB * NW) {
      double v[TB][4];
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        const int tile = t0 + u * NW, tc = tile < ntl ? tile : 0, bl = tc / wt, kk = tc - bl * wt;
        const int gi = c0 + CT * bl + r, gj0 = c0 + CT * kk + c4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gj = gj0 + q;
          const bool in = tile < ntl && gi < n1 && gj < ns && gj <= gi;
          v[u][q] = masked_load(buf, (size_t)gi * ns + gj, in) + (in ? (gi == gj ? reg : 0.0) : (gi == gj ? 1.0 : 0.0));
        }
      }
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        const int tile = t0 + u * NW;
        if (tile < ntl) {
#pragma unroll
          for (int q = 0; q < 4; ++q) P[(size_t)tile * CTS + r * CTL + c4 + q] = v[u][q];
        }
      }
    }
  }
  __syncthreads();
  CHOLP_STAMP(0)
  if (wave == 0) chol_tile_factor_noinv(P, dinv, min(CT, ns - c0), c0, lane, badcol);
  __syncthreads();
  CHOLP_STAMP(1)
  for (int kk = 0; kk < wt; ++kk) {
    double* Dk = P + (size_t)(kk * wt + kk) * CTS;       // (factored: by the line above or by the look-ahead of step kk - 1)
    // tiles below the diagonal: X = A L_kk^-T (four per wavefront)
    for (int b0 = kk + 1 + 4 * wave; b0 < nbr; b0 += 4 * NW) {
      const int bl = b0 + lg;
      chol_panel_solve4(Dk, dinv + kk * CT, bl < nbr ? P + (size_t)(bl * wt + kk) * CTS : nullptr, li);
    }
    __syncthreads();
    CHOLP_STAMP(2)
    // remaining tile columns of the panel: C(bl, bj) -= X(bl, kk) X(bj, kk)^T for kk < bj < wt, bl >= bj.  Tile 0 of the
    // enumeration is the next diagonal tile: wave 0 updates it and factors it at once (look-ahead) while the other waves
    // share the rest.
    {
      const int ncol = wt - kk - 1;
      int nt = 0;
      for (int q = 0; q < ncol; ++q) nt += nbr - (kk + 1 + q);
      if (wave == 0) {
        if (nt > 0) {
          const int b1 = kk + 1;
          double* D1 = P + (size_t)(b1 * wt + b1) * CTS;
          const double* X1 = P + (size_t)(b1 * wt + kk) * CTS;
          chol_tile_syrk(X1, X1, D1, li, lg);
          lds_fence();
          const int ncol1 = min(CT, ns - (c0 + CT * b1));
          if (ncol1 > 0) chol_tile_factor_noinv(D1, dinv + b1 * CT, ncol1, c0 + CT * b1, lane, badcol);
        }
      } else {
        for (int tt = wave; tt < nt; tt += NW - 1) {
          int q = 0, rem = tt;
          while (rem >= nbr - (kk + 1 + q)) { rem -= nbr - (kk + 1 + q); ++q; }
          const int bj = kk + 1 + q, bl = bj + rem;
          chol_tile_syrk(P + (size_t)(bl * wt + kk) * CTS, P + (size_t)(bj * wt + kk) * CTS, P + (size_t)(bl * wt + bj) * CTS, li, lg);
        }
      }
    }
    __syncthreads();
    CHOLP_STAMP(3)
  }
  // ---- store the factor (lower triangle incl. the right-hand-side row); while those stores are in flight the diagonal tiles
  // are inverted for the back substitution, one wavefront per tile (inverting tile kk inside step kk put 3 k cycles of one
  // wave in front of every panel solve; inverting before the store added 4.6 k cycles per panel to the critical path)
  {
    const int ntl = nbr * wt, r = lane >> 2, c4 = (lane & 3) * 4;
    for (int tile = wave; tile < ntl; tile += NW) {
      const int bl = tile / wt, kk = tile - bl * wt;
      if (bl < kk) continue;                                   // (above the diagonal)
      const int gi = c0 + CT * bl + r, gj0 = c0 + CT * kk + c4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int gj = gj0 + q;
        if (gi < n1 && gj < ns && gj <= gi) buf[(size_t)gi * ns + gj] = P[(size_t)tile * CTS + r * CTL + c4 + q];
      }
    }
  }
  CHOLP_STAMP(5)
  if (wave < wt) chol_tile_invert(P + (size_t)(wave * wt + wave) * CTS, dinv + wave * CT, Xi + (size_t)wave * CTS, lane,
                                    min(CT, ns - c0 - CT * wave));
  __syncthreads();
  CHOLP_STAMP(4)
  for (int e = tid; e < wt * CT * CT; e += CHOLP_THREADS) {
    const int kk = e >> 8, r = (e >> 4) & 15, c = e & 15;
    Linv[(size_t)(kt0 + kk) * CT * CT + r * CT + c] = Xi[(size_t)kk * CTS + r * CTL + c];
  }
  // first non-positive pivot (1-based; 0 = none): the first panel initialises the report, later ones only add to it
  if (tid == 0 && (kt0 == 0 || (badcol != 0 && info[0] == 0))) info[0] = badcol;
  if (prof) {
    __syncthreads();
    if (tid == 0)
      for (int i = 0; i < 6; ++i) prof[i] += tp[i];   // (panels run one after the other: plain accumulation)
  }
#undef CHOLP_STAMP
}

// trailing update behind the panel [kt0, kt0 + wt): one wavefront per tile (bi, bj), bj <= bi, both >= kt0 + wt
__global__ __launch_bounds__(256) void k_cholp_trail(int ns, int kt0, int wt, double* __restrict__ buf) {
  const int n1 = ns + 1, nb = (n1 + CT - 1) / CT, nbc = (ns + CT - 1) / CT, k1 = kt0 + wt;
  const int m = nb - k1, nt = m * (m + 1) / 2;
  const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
  const int tt = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tt >= nt) return;
  int a = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);   // tt -> (a, rem), rem <= a
  a += ((a + 1) * (a + 2) / 2 <= tt) ? 1 : 0;
  a -= (a * (a + 1) / 2 > tt) ? 1 : 0;
  const int rem = tt - a * (a + 1) / 2, bi = k1 + a, bj = k1 + rem;
  if (bj >= nbc) return;                                            // (a column block behind the matrix: rhs-row tile only)
  double4_t acc;
  double av[CHOLP_WT * 4], bv[CHOLP_WT * 4];
  const int gia = CT * bi + li, gib = CT * bj + li, gjc = CT * bj + li;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = CT * bi + lg + 4 * r;
    acc[r] = masked_load(buf, (size_t)gi * ns + gjc, gi < n1 && gjc < ns);
  }
#pragma unroll
  for (int q = 0; q < CHOLP_WT * 4; ++q) {
    const int gk = CT * kt0 + 4 * q + lg;                          // column of L inside the panel
    const bool on = q < 4 * wt;
    av[q] = -masked_load(buf, (size_t)gia * ns + gk, on && gia < n1);
    bv[q] = masked_load(buf, (size_t)gib * ns + gk, on && gib < n1);
  }
#pragma unroll
  for (int q = 0; q < CHOLP_WT * 4; ++q)
    if (q < 4 * wt) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = CT * bi + lg + 4 * r;
    if (gi < n1 && gjc < ns && gjc <= gi) buf[(size_t)gi * ns + gjc] = acc[r];
  }
}

// p = L^-T y: y = row ns of the factored matrix; Linv = inverted diagonal tiles.  ONE workgroup of 512 threads, thread t owns
// columns t (and t + 512: NCOL = 2 for ns > 512).  A step of the blocked back substitution is a 16 x 16 mat-vec with the
// stored inverse (wave 0) and a rank-16 update of the entries above by all threads; everything it reads from memory -- the
// inverse of the NEXT tile and the next row block of L -- is requested a step ahead into the other half of a register
// ping-pong (the step loop is unrolled by two, no copies), so that no global round trip sits between two steps.  (First
// version: both waited for in every step, 40 us for the 18 steps of ns = 286; a 1024-thread version with the same
// prefetch spilled its 128-register budget and took 143 us.)
constexpr int CHOLP_BACK_THREADS = 512;
// DEPTH: steps whose inverse tile / row block of L are in flight at any time (a ring of register sets; the step loop is
// unrolled by DEPTH so that the ring index is a compile-time constant).  One step ahead was not enough: the factor was
// written by other compute units (other XCDs), so every request is a ~2 us round trip to memory, ten times the ~0.2 us of
// a step -- 41.8 us at ns = 286 with DEPTH = 2, whatever the order of the instructions.
// Wave 0 owns the tile solves (its ring slots hold the inverse tiles), waves 1 .. 7 own the columns (their slots hold the row
// blocks of L): ONE register ring serves both roles -- two rings side by side did not fit the register file.
constexpr int CHOLP_BACK_COLS = CHOLP_BACK_THREADS - 64;   // columns per pass of the column waves
template <int NCOL, int DEPTH>
__global__ __launch_bounds__(CHOLP_BACK_THREADS) void k_cholp_back(int ns, const double* __restrict__ buf,
                                                                    const double* __restrict__ Linv, double* __restrict__ ps) {
  __shared__ double yv[NCOL * CHOLP_BACK_COLS + 2 * CT], pv[CT], pall[NCOL * CHOLP_BACK_COLS + 2 * CT];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15;
  const bool solver = tid < 64;
  const int col0 = tid - 64;                                 // first column of a column thread
  const int nbc = (ns + CT - 1) / CT;
  for (int e = tid; e < nbc * CT; e += CHOLP_BACK_THREADS) yv[e] = e < ns ? buf[(size_t)ns * ns + e] : 0.0;
  double ring[DEPTH][NCOL][CT];
  auto fetch = [&](int kb, auto slot) {
    constexpr int S = decltype(slot)::value;
    // RAW loads from clamped (always valid) addresses: whatever does not belong to the step is masked when the slot is
    // USED.  (masked_load multiplies the loaded value by its 0 / 1 mask at once -- an ALU instruction that has to wait for
    // the data: the "prefetch" of the first version was a synchronous load.)
    const int kc = max(kb, 0);
    if (solver) {   // inverse of tile kb, column li
#pragma unroll
      for (int i = 0; i < CT; ++i) ring[S][0][i] = Linv[(size_t)kc * CT * CT + i * CT + li];
    } else {        // row block kb of L, the thread's columns
#pragma unroll
      for (int q = 0; q < NCOL; ++q) {
        const int col = min(col0 + q * CHOLP_BACK_COLS, ns - 1);
#pragma unroll
        for (int r = 0; r < CT; ++r) ring[S][q][r] = buf[(size_t)min(CT * kc + r, ns - 1) * ns + col];
      }
    }
  };
  auto step = [&](int kb, auto slot) {
    constexpr int S = decltype(slot)::value;
    if (solver) {          // p_k = L_kk^-T z_k (the strict upper part of the stored inverse is zero)
      double sp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < CT; ++i) sp[i & 3] += ring[S][0][i] * yv[CT * kb + i];
      const double sum = (sp[0] + sp[1]) + (sp[2] + sp[3]);
      if (lane < CT) {
        pv[li] = sum;
        pall[CT * kb + li] = sum;   // (written to memory once, after the last step: a global store in front of the barrier
      }                             //  made the solver wave wait for its ~2 us write round trip in EVERY step)
    }
    __syncthreads();
    if (!solver) {
#pragma unroll
      for (int q = 0; q < NCOL; ++q) {
        const int col = col0 + q * CHOLP_BACK_COLS;
        if (col < CT * kb) {
          double sp[4] = {yv[col], 0.0, 0.0, 0.0};
#pragma unroll
          for (int r = 0; r < CT; ++r) sp[r & 3] -= (CT * kb + r < ns ? ring[S][q][r] : 0.0) * pv[r];   // (rows behind the matrix)
          yv[col] = (sp[0] + sp[1]) + (sp[2] + sp[3]);
        }
      }
    }
    fetch(kb - DEPTH, slot);   // the slot is free again: request the step DEPTH behind this one
    __syncthreads();
  };
  // fill the ring: steps nbc - 1 .. nbc - DEPTH
  if constexpr (DEPTH >= 1) fetch(nbc - 1, std::integral_constant<int, 0>());
  if constexpr (DEPTH >= 2) fetch(nbc - 2, std::integral_constant<int, 1 % DEPTH>());
  if constexpr (DEPTH >= 3) fetch(nbc - 3, std::integral_constant<int, 2 % DEPTH>());
  if constexpr (DEPTH >= 4) fetch(nbc - 4, std::integral_constant<int, 3 % DEPTH>());
  static_assert(DEPTH >= 1 && DEPTH <= 4, "ring depth");
  __syncthreads();
  for (int kb = nbc - 1; kb >= 0; kb -= DEPTH) {
    step(kb, std::integral_constant<int, 0>());
    if constexpr (DEPTH >= 2) { if (kb - 1 >= 0) step(kb - 1, std::integral_constant<int, 1 % DEPTH>()); }
    if constexpr (DEPTH >= 3) { if (kb - 2 >= 0) step(kb - 2, std::integral_constant<int, 2 % DEPTH>()); }
    if constexpr (DEPTH >= 4) { if (kb - 3 >= 0) step(kb - 3, std::integral_constant<int, 3 % DEPTH>()); }
  }
  for (int e = tid; e < ns; e += CHOLP_BACK_THREADS) ps[e] = pall[e];
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-workgroup blocked Cholesky for large reduced systems (adjust_board: ns = shared + 3 x #board points, up to a
// few thousand).  Data layout: (ns+1) x ns, row ns = rhs, 64-column panels, three launches per
// panel: diagonal block (one workgroup), panel solve (one row per thread), symmetric rank-64 trailing update with
// v_mfma_f64_16x16x4_f64 on 64 x 64 tiles staged in LDS.  Then a blocked backward substitution (two launches / panel).
// ---------------------------------------------------------------------------------------------------------------
constexpr int CB = 64, CBL = CB + 1;

__global__ __launch_bounds__(64) void k_cholb_diag(int ns, int k0, double reg, double* __restrict__ A, int* __restrict__ info) {
  __shared__ double D[CB * CBL];
  const int tid = threadIdx.x, nb = min(CB, ns - k0);
  for (int e = tid; e < CB * CB; e += 64) {
    const int i = e / CB, j = e % CB;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < nb && j < nb) {
      v = A[(size_t)(k0 + i) * ns + k0 + j];
      if (i == j) v += reg;
    }
    D[i * CBL + j] = v;
  }
  lds_fence();
  for (int j = 0; j < nb; ++j) {
    double dj = D[j * CBL + j];
    if (!(dj > 0.0)) { if (tid == 0 && info[0] == 0) info[0] = k0 + j + 1; dj = 1e-300; }
    dj = sqrt(dj);
    lds_fence();
    if (tid == 0) D[j * CBL + j] = dj;
    for (int i = j + 1 + tid; i < nb; i += 64) D[i * CBL + j] /= dj;
    lds_fence();
    const int mm = nb - j - 1;
    for (int e = tid; e < mm * mm; e += 64) {
      const int i = j + 1 + e / mm, k = j + 1 + e % mm;
      if (k <= i) D[i * CBL + k] -= D[i * CBL + j] * D[k * CBL + j];
    }
    lds_fence();
  }
  for (int e = tid; e < nb * nb; e += 64) {
    const int i = e / nb, j = e % nb;
    if (j <= i) A[(size_t)(k0 + i) * ns + k0 + j] = D[i * CBL + j];
  }
}

// rows k0+nb .. ns (the last one is the right-hand side):  row <- row L11^-T, one row per thread, row kept in LDS
__global__ __launch_bounds__(64) void k_cholb_trsm(int ns, int k0, double* __restrict__ A) {
  __shared__ double D[CB * CBL];
  __shared__ double X[64 * CBL];
  const int tid = threadIdx.x, nb = min(CB, ns - k0);
  for (int e = tid; e < CB * CB; e += 64) {
    const int i = e / CB, j = e % CB;
    D[i * CBL + j] = (i < nb && j < nb && j <= i) ? A[(size_t)(k0 + i) * ns + k0 + j] : (i == j ? 1.0 : 0.0);
  }
  lds_fence();
  const int r = k0 + nb + blockIdx.x * 64 + tid;
  if (r > ns) return;
  double* arow = A + (size_t)r * ns + k0;
  double* xr = X + tid * CBL;
  for (int j = 0; j < nb; ++j) xr[j] = arow[j];
  for (int j = 0; j < nb; ++j) {
    double v = xr[j];
    const double* dj = D + j * CBL;
    for (int k = 0; k < j; ++k) v -= xr[k] * dj[k];
    v /= dj[j];
    xr[j] = v;
    arow[j] = v;
  }
}

// trailing update: tile (br, bc), bc <= br, of rows/cols k0+nb+64*b .. ;  A[r][c] -= L[r][k0..k0+nb) . L[c][k0..k0+nb)
__global__ __launch_bounds__(256) void k_cholb_syrk(int ns, int k0, double* __restrict__ A) {
  __shared__ double Lr[CB * CBL];
  __shared__ double Lc[CB * CBL];
  const int br = blockIdx.x, bc = blockIdx.y;
  if (bc > br) return;
  const int nb = min(CB, ns - k0);
  const int base = k0 + nb;
  const int r0 = base + br * CB, c0 = base + bc * CB;
  for (int e = threadIdx.x; e < CB * CB; e += 256) {
    const int i = e / CB, k = e % CB;
    const int r = r0 + i, c = c0 + i;
    Lr[i * CBL + k] = (r <= ns && k < nb) ? A[(size_t)r * ns + k0 + k] : 0.0;
    Lc[i * CBL + k] = (c < ns && k < nb) ? A[(size_t)c * ns + k0 + k] : 0.0;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, rsub = lane >> 4, csub = lane & 15;
  double4_t acc[4];
  for (int ct = 0; ct < 4; ++ct) acc[ct] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int st = 0; st < CB / 4; ++st) {
    const double a = Lr[(16 * wave + csub) * CBL + 4 * st + rsub];
    for (int ct = 0; ct < 4; ++ct) {
      const double b = Lc[(16 * ct + csub) * CBL + 4 * st + rsub];
      acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[ct], 0, 0, 0);
    }
  }
  for (int ct = 0; ct < 4; ++ct)
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + 16 * wave + rsub + 4 * q, c = c0 + 16 * ct + csub;
      if (r <= ns && c < ns && c <= r) A[(size_t)r * ns + c] -= acc[ct][q];
    }
}

// backward substitution, panel k0: solve L11^T p = y_panel (one wavefront)
__global__ __launch_bounds__(64) void k_cholb_back_diag(int ns, int k0, double* __restrict__ A, double* __restrict__ ps) {
  __shared__ double D[CB * CBL];
  const int tid = threadIdx.x, nb = min(CB, ns - k0);
  double* y = A + (size_t)ns * ns;
  for (int e = tid; e < nb * nb; e += 64) {
    const int i = e / nb, j = e % nb;
    D[i * CBL + j] = A[(size_t)(k0 + i) * ns + k0 + j];
  }
  lds_fence();
  double yv = tid < nb ? y[k0 + tid] : 0.0;
  for (int i = nb - 1; i >= 0; --i) {
    const double pi = __shfl(yv, i, 64) / D[i * CBL + i];
    if (tid == i) yv = pi;
    else if (tid < i) yv -= D[i * CBL + tid] * pi;
  }
  if (tid < nb) {
    y[k0 + tid] = yv;
    ps[k0 + tid] = yv;
  }
}
// y[j] -= sum_i L[k0+i][j] p[k0+i]  for j < k0
__global__ void k_cholb_back_update(int ns, int k0, double* __restrict__ A) {
  const int nb = min(CB, ns - k0);
  double* y = A + (size_t)ns * ns;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k0) return;
  double sum = 0.0;
  for (int i = 0; i < nb; ++i) sum += A[(size_t)(k0 + i) * ns + j] * y[k0 + i];
  y[j] -= sum;
}

// back-substitution: gn_s = p_s ; gn_f = L^-T (y_f - W p_s).  One wavefront per frame: W p_s (DF x ns, rows padded to
// 16) runs on the matrix pipe with p_s broadcast over the 16 output columns -- the MFMA does the cross-lane reduction
// that a lane-per-column dot product would need 12 shuffles trees for; z goes through LDS and lane dd < DF applies
// column dd of L^-1.  The last block copies the shared part.  dots (may be null): partial dots over the entries the
// block has written, dots[3 blk + {0,1,2}] = {g_h.g_h, g_h.gn, gn.gn}, and the Cholesky pivot report in
// dots[3 gridDim.x] -- summed on the host by the single-GPU driver (sharded handles need the all-reduced gn: k_dots3).
// ---------------------------------------------------------------------------------------------------------------
// Schur step 1, ONE workgroup per frame (round 2; replaces k_tr_reg + k_frame_factor + k_schur_w):
//   * the first wavefront folds the k_vec_scale / k_q00 partials and computes the damping (tr_reg_wave; workgroup 0
//     publishes the scalar block), when the device-side trust-region algebra is on;
//   * A_ff = D_f H_ff D_f + reg I goes into a 16 x 16 LDS tile (identity padding) and is factored by the first wavefront in
//     registers (chol_tile_factor_noinv: lane = row, pivots through v_readlane) -- one thread per frame did the same
//     12 x 12 factor AND its inverse as a serial chain of ~700 dependent operations (8.8 us);
//   * thread s takes column s of [D_f H_fs D_s | g_f] and solves L w = b by forward substitution (L is a broadcast read of
//     LDS): W[:, s], and y = W[:, ns].
// L (strict lower part) with 1 / L_ii on the diagonal is kept for the back substitution (Lf [Fl][DF][DF]).
// ---------------------------------------------------------------------------------------------------------------
template <int DF>
__global__ __launch_bounds__(256, 2) void k_schur_frame(Dims d, const double* __restrict__ Hff, const double* __restrict__ Hfs,
                                                    const double* __restrict__ dsc, const double* __restrict__ gh, double reg,
                                                    double* __restrict__ Lf, double* __restrict__ W, double* __restrict__ yf,
                                                    double* tr, const double* __restrict__ vs_part, int nvb,
                                                    const double* __restrict__ q_part, int nq, int first, double Delta_in) {
  __shared__ double Dt[CTS];
  __shared__ double dinv_s[CT], df_s[CT];
  __shared__ double reg_s;
  const int fl = blockIdx.x, f = d.f0 + fl, ns = d.ns, ldw = ns + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    double r = reg;
    if (tr != nullptr)
      r = vs_part != nullptr ? tr_reg_wave(blockIdx.x == 0 ? tr : nullptr, vs_part, nvb, q_part, nq, first, Delta_in, lane)
                             : tr[TR_REG];
    if (lane == 0) reg_s = r;
  }
  if (tid >= 64 && tid < 64 + CT) df_s[tid - 64] = (tid - 64) < DF ? dsc[d.frame_to_x(f, tid - 64)] : 0.0;
  __syncthreads();
  const double* hff = Hff + (size_t)fl * DF * DF;
  {
    const int i = tid >> 4, j = tid & 15;   // 256 threads = the 16 x 16 tile
    double v = (i == j) ? 1.0 : 0.0;
    if (i < DF && j < DF) v = df_s[i] * hff[i * DF + j] * df_s[j] + (i == j ? reg_s : 0.0);
    Dt[i * CTL + j] = v;
  }
  __syncthreads();
  if (wave == 0) {
    int badcol = 0;   // (a non-positive pivot is clamped, as in the reduced system's factor: the step is then rejected by its cost)
    chol_tile_factor_noinv(Dt, dinv_s, DF, 0, lane, badcol);
  }
  __syncthreads();
  const double* hfs = Hfs + (size_t)fl * DF * ns;
  double* w = W + (size_t)fl * DF * ldw;
  for (int s = tid; s <= ns; s += blockDim.x) {
    double b[DF];
    if (s < ns) {
      const double dsv = dsc[d.shared_to_x(s)];
#pragma unroll
      for (int i = 0; i < DF; ++i) b[i] = df_s[i] * hfs[i * ns + s] * dsv;
    } else {
#pragma unroll
      for (int i = 0; i < DF; ++i) b[i] = gh[d.frame_to_x(f, i)];
    }
#pragma unroll
    for (int i = 0; i < DF; ++i) {
      double s0 = b[i], s1 = 0.0;
#pragma unroll
      for (int m = 0; m < i; ++m) {
        if (m & 1) s1 -= Dt[i * CTL + m] * b[m]; else s0 -= Dt[i * CTL + m] * b[m];
      }
      b[i] = (s0 + s1) * dinv_s[i];
    }
#pragma unroll
    for (int i = 0; i < DF; ++i) w[i * ldw + s] = b[i];
    if (s == ns) {
#pragma unroll
      for (int i = 0; i < DF; ++i) yf[fl * DF + i] = b[i];
    }
  }
  if (tid < DF * DF) {
    const int i = tid / DF, j = tid % DF;
    Lf[(size_t)fl * DF * DF + tid] = (i == j) ? dinv_s[i] : (j < i ? Dt[i * CTL + j] : 0.0);
  }
}

template <int DF>
__global__ __launch_bounds__(64, 8) void k_schur_backsub(Dims d, const double* __restrict__ Linv, const double* __restrict__ W,
                                                      const double* __restrict__ yf, const double* __restrict__ ps,
                                                      double* __restrict__ gn, const double* __restrict__ gh,
                                                      const int* __restrict__ info, double* __restrict__ dots) {
  __shared__ double zs[16];
  __shared__ double dl[3][16];
  const int ns = d.ns, lane = threadIdx.x;
  if (blockIdx.x == gridDim.x - 1) {   // last block: shared part (frame entries of other shards stay 0: memset)
    double dt[3] = {0, 0, 0};
    for (int s = lane; s < ns; s += 64) {
      const int xi = d.shared_to_x(s);
      const double a = gh[xi], b = ps[s];
      gn[xi] = b;
      dt[0] += a * a; dt[1] += a * b; dt[2] += b * b;
    }
    if (dots == nullptr) return;
    if (d.shard_world > 0 && d.shard_rank != 0) dt[0] = dt[1] = dt[2] = 0.0;   // shared entries count on rank 0 only
    for (int k = 0; k < 3; ++k) {
      const double ds = wave_sum(dt[k]);
      if (lane == 0) dots[3 * blockIdx.x + k] = ds;
    }
    if (lane == 0) dots[3 * gridDim.x] = (double)info[0];
    return;
  }
  const int fl = blockIdx.x, ldw = ns + 1;
  const double* w = W + (size_t)fl * DF * ldw;
  const int ri = lane & 15, kq = lane >> 4;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  constexpr int UB = 8;
  for (int k0 = 0; k0 < ns; k0 += 4 * UB) {
    double av[UB], bv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int kk = k0 + 4 * u + kq;
      av[u] = masked_load(w, (size_t)(ri * ldw + kk), ri < DF && kk < ns);
      bv[u] = masked_load(ps, (size_t)kk, kk < ns);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
  if (ri == 0) {   // output column 0 (all 16 columns are equal): rows kq + 4 r
#pragma unroll
    for (int r = 0; r < 4; ++r) zs[kq + 4 * r] = acc[r];
  }
  lds_fence();
  // gn_f = L^-T (y - W p_s) by back substitution with the factor k_schur_frame left (strict lower part, 1 / L_ii on the
  // diagonal): lane m < DF carries r_m; v_i travels through v_readlane.  The column of L is read first: only
  // mul -> readlane -> fma stay on the chain.
  double a = 0.0, v = 0.0;
  {
    const double* Lg = Linv + (size_t)fl * DF * DF;
    const int lm = lane < DF ? lane : 0;
    double lc[DF], dv[DF];
#pragma unroll
    for (int i = 0; i < DF; ++i) {
      lc[i] = (lane < i) ? Lg[i * DF + lm] : 0.0;
      dv[i] = Lg[i * DF + i];
    }
    double r = lane < DF ? yf[fl * DF + lm] - zs[lm] : 0.0;
#pragma unroll
    for (int i = DF - 1; i >= 0; --i) {
      const double vi = lane_bcast(r * dv[i], i);
      v = (lane == i) ? vi : v;
      r -= lc[i] * vi;
    }
  }
  if (lane < DF) {
    const int xi = d.frame_to_x(d.f0 + fl, lane);
    gn[xi] = v;
    a = gh[xi];
  }
  if (dots == nullptr) return;
  if (lane < 16) {
    dl[0][lane] = a * a;
    dl[1][lane] = a * v;
    dl[2][lane] = v * v;
  }
  lds_fence();
  if (lane < 3) {
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < DF; ++k) sum += dl[lane][k];
    dots[3 * blockIdx.x + lane] = sum;
  }
}

// folds the partial dots of the back-substitution, builds the 2-D subspace model and solves it for the current radius.
// One wavefront; S may be the global scalar block or a copy of it.
__device__ __forceinline__ void tr_step_wave(double* S, const double* __restrict__ dot_part, int nblk, int lane) {
  double dt[3] = {0, 0, 0};
  for (int b = lane; b < nblk; b += 64)
    for (int k = 0; k < 3; ++k) dt[k] += dot_part[3 * b + k];
  for (int k = 0; k < 3; ++k) dt[k] = wave_sum(dt[k]);
  if (lane == 0) {
    S[TR_D00] = dt[0];
    S[TR_D01] = dt[1];
    S[TR_D11] = dt[2];
    S[TR_INFO] = dot_part[3 * nblk];
    tr_subspace(S);
    tr_trial(S, S[TR_DELTA]);
  }
}

// step: p_h = alpha u0 + beta u1; step = d * p_h; x_new = x + step, one element per thread.
// part[3 blk + {0,1,2}] = {|p_h|^2, |step|^2, |x|^2} of the block (folded by the host)
//
// The kernel is also the END of the device-side trust-region algebra and the START of the trial evaluation (two launches
// less per iteration, ~5 us of timeline each):
//   * S != nullptr: the coefficients come from the 2-D subspace step.  The first wavefront of EVERY workgroup runs that
//     scalar algebra (tr_step_wave: fold of the back-substitution's partial dots, subspace model, trial step) on a copy of
//     the scalar block in LDS -- redundant, identical arithmetic -- and workgroup 0 publishes the block for the host.
//   * workgroups behind the nvb vector blocks (prep_blocks of them, 0 = none) form the pose / camera / board-point table
//     entries of the trial point for k_cost; they evaluate the entries of x_new they need with the same two fused
//     operations as the vector blocks (StepX), so they do not wait for x_new to be stored.
__global__ __launch_bounds__(256) void k_vec_step(Dims d, Tables t, const double* __restrict__ x,
                                                  const double* __restrict__ dsc, const double* __restrict__ u0,
                                                  const double* __restrict__ u1, double alpha, double beta,
                                                  double* __restrict__ xnew, double* __restrict__ part, double* S,
                                                  const double* __restrict__ dot_part, int nblk, int nvb,
                                                  double* host_S = nullptr, double* host_part = nullptr) {
  __shared__ double scratch[16];
  __shared__ double Sl[TR_NSLOTS];
  if (S != nullptr) {
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      if (lane < TR_NSLOTS) Sl[lane] = S[lane];
      lds_fence();
      tr_step_wave(Sl, dot_part, nblk, lane);
    }
    __syncthreads();
    alpha = Sl[TR_ALPHA];
    beta = Sl[TR_BETA];
    if (blockIdx.x == 0 && threadIdx.x < TR_NSLOTS) {
      S[threadIdx.x] = Sl[threadIdx.x];
      if (host_S != nullptr) host_S[threadIdx.x] = Sl[threadIdx.x];   // (pinned host memory: see publish_cost)
    }
  }
  if ((int)blockIdx.x >= nvb) {
    prep_item(d, t, StepX{x, dsc, u0, u1, alpha, beta}, ((int)blockIdx.x - nvb) * blockDim.x + threadIdx.x);
    return;
  }
  double ph = 0, st = 0, xx = 0;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.n) {
    const double p = step_direction(alpha, u0[i], beta, u1[i]);
    const double s = dsc[i] * p;
    const double xi = x[i];
    xnew[i] = step_point(xi, dsc[i], p);
    const double w = d.entry_weight(i);
    ph = w * (p * p);
    st = w * (s * s);
    xx = w * (xi * xi);
  }
  const double a = block_reduce<false>(ph, scratch);
  const double b = block_reduce<false>(st, scratch);
  const double c = block_reduce<false>(xx, scratch);
  if (threadIdx.x == 0) {
    part[3 * blockIdx.x + 0] = a;
    part[3 * blockIdx.x + 1] = b;
    part[3 * blockIdx.x + 2] = c;
    if (host_part != nullptr) {
      host_part[3 * blockIdx.x + 0] = a;
      host_part[3 * blockIdx.x + 1] = b;
      host_part[3 * blockIdx.x + 2] = c;
    }
  }
}

// The scalars the HOST waits for in every LM iteration (trust-region results, step norms, trial-cost partials) written straight
// into its pinned memory, followed by a sequence number with system-scope release: the driver spins on that word instead of
// waiting for a device-to-host copy + event (which delivered the trial cost ~75 us after k_vec_step although k_cost was done
// after ~45: the host then enqueued the next iteration a few microseconds too late, 10 us of dispatch gaps per iteration).
__global__ __launch_bounds__(256) void k_publish(const double* __restrict__ scal, double* __restrict__ host_dst, int n,
                                                 unsigned long long* host_seq, unsigned long long seq) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) host_dst[i] = scal[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The partial sums that the head of every k_schur_frame workgroup folds (3 nvb of k_vec_scale, nq of the curvature), folded ONCE
// into four scalars -- in exactly the order tr_reg_wave uses, so the totals are bit-identical.  Enqueued behind the speculative
// k_vec_scale_q00 of a trial point, where it runs in the host's decision latency: the 500 frame workgroups of the next
// iteration then read 4 values instead of 584 (k_schur_frame 10.0 -> 8.5 us with the partials of 128 curvature blocks).
__global__ __launch_bounds__(64) void k_fold_tr(const double* __restrict__ vs_part, int nvb, const double* __restrict__ q_part,
                                                int nq, double* __restrict__ out /*[4]: mx, gg, xs, q*/) {
  const int lane = threadIdx.x;
  double mx = 0, gg = 0, xs = 0, q = 0;
  for (int b = lane; b < nvb; b += 64) {
    mx = fmax(mx, vs_part[3 * b]);
    gg += vs_part[3 * b + 1];
    xs += vs_part[3 * b + 2];
  }
  for (int b = lane; b < nq; b += 64) q += q_part[b];
  mx = wave_max(mx);
  gg = wave_sum(gg);
  xs = wave_sum(xs);
  q = wave_sum(q);
  if (lane == 0) { out[0] = mx; out[1] = gg; out[2] = xs; out[3] = q; }
}

__global__ __launch_bounds__(64) void k_fold_partials(double* __restrict__ part, int n) {
  double s = 0.0;
  for (int b = threadIdx.x; b < n; b += 64) s += part[b];
  s = wave_sum(s);
  if (threadIdx.x == 0) part[0] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// solver = "lsmr" (scipy's TRF with its LSMR trust-region solver on the device): the reduction of the per-view partials of
// J_h^T u (k_lsmr_jtu) to the n entries of the vector, and the vector updates of the LSMR iteration.
// ---------------------------------------------------------------------------------------------------------------
// vout[i] = dscale[i] * (sum over the views that contain parameter i of part[view][local index]) - beta * vold[i];
// nrm[i] = vout[i]^2 (folded afterwards).  Fixed summation order: "lane L adds the views L, L + 64, ..; wave_sum".
//   workgroups [0, n - nfe):       one wavefront per entry outside the per-frame pose block (camera poses, board poses, intrinsics,
//                                  the hand-eye pair: sums over up to all views)
//   workgroups [n - nfe, .. + Fl): one wavefront per FRAME for its 6 / 12 pose entries (sums over the C x B views of the frame):
//                                  16 lanes per entry -- as one wavefront per entry these were 6 000 of the launch's 6 140
//                                  workgroups at the north-star rig (26.6 us; lsmr_gather_frame_entries gives the count).  The
//                                  four accumulators of a lane are the lanes L, L + 16, L + 32, L + 48 of the order above.
__host__ __device__ inline int lsmr_gather_frame_entries(const Dims& d) {
  return (d.off_motion >= 0 && d.motion != MOTION_HAND_EYE) ? d.n_motion : 0;
}
//   boards=True: entry (point q, coordinate k) of the board-point block sums bpart[3 idx + k] over the inlier observations of the
//                point (all frames x cameras of its board; idx = the observation's residual index, obs_index)
//   frame-sharded handles (raw_shared): the SHARED entries are left as the rank's raw sum -- they are all-reduced and finished
//                by k_lsmr_shard_finish; the entries of the own frames are final
struct LsmrGatherExtra {
  const int32_t* obs_index;
  const int32_t* board_off;
  const double* bpart;
  int raw_shared;
};
__global__ __launch_bounds__(64) void k_lsmr_gather(Dims d, const double* __restrict__ part, int part_stride,
                                                    const double* __restrict__ dscale, double beta, const double* __restrict__ vold,
                                                    double* __restrict__ vout, double* __restrict__ nrm,
                                                    const double* __restrict__ ls, LsmrGatherExtra ex) {
  const int lane = threadIdx.x;
  const int nfe = lsmr_gather_frame_entries(d), ngen = d.n - nfe;
  bool skip = false;
  if (ls != nullptr) {   // device-resident solve: beta from the state; beta == 0 leaves v as it is
    if (ls[LS_ISTOP] != 0.0) return;
    skip = ls[LS_SKIPV] != 0.0;
    beta = ls[LS_BETA];
  }
  const int CB = d.C * d.B, npc = 6 * d.NPB;
  if ((int)blockIdx.x >= ngen) {
    const int fl = (int)blockIdx.x - ngen;
    if (fl >= d.Fl) return;
    const int DFm = d.motion == MOTION_ROLLING ? 12 : 6, g = lane >> 4, l16 = lane & 15;
    for (int e = g; e < DFm; e += 4) {                  // entry e of the frame: chain e / 6, component e % 6
      const int chain = e / 6, i = d.off_motion + chain * 6 * d.F + 6 * (d.f0 + fl) + e % 6;
      if (skip) {
        if (l16 == 0) vout[i] = vold[i];
        continue;
      }
      double s4[4] = {0.0, 0.0, 0.0, 0.0};
      const double* src = part + (size_t)fl * CB * part_stride + 6 + e;
      for (int w0 = 0; w0 < CB; w0 += 64)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int vw = w0 + 16 * k + l16;
          if (vw < CB) s4[k] += src[(size_t)vw * part_stride];
        }
      double sum = (s4[0] + s4[2]) + (s4[1] + s4[3]);
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) sum += __shfl_down(sum, off, 16);
      if (l16 == 0) {
        const double val = dscale[i] * sum - beta * vold[i];
        vout[i] = val;
        nrm[i] = val * val;
      }
    }
    return;
  }
  const int i = nfe > 0 && (int)blockIdx.x >= d.off_motion ? (int)blockIdx.x + nfe : (int)blockIdx.x;
  if (i >= d.n) return;
  if (skip && !ex.raw_shared) {
    if (lane == 0) vout[i] = vold[i];
    return;
  }
  if (d.off_boards >= 0 && i >= d.off_boards) {   // adjusted board point: sum over the observations of the point
    const int q = (i - d.off_boards) / 3, k = (i - d.off_boards) % 3;
    int b = 0;
    while (q >= ex.board_off[b + 1]) ++b;
    const int p = q - ex.board_off[b], total = d.Fl * d.C;
    double sum = 0.0;
    for (int e = lane; e < total; e += 64) {   // e = fl C + c: the frame-major views of board b
      const int idx = ex.obs_index[((size_t)e * d.B + b) * d.P + p];
      if (idx >= 0) sum += ex.bpart[3 * (size_t)idx + k];
    }
    sum = wave_sum(sum);
    if (lane == 0) {
      if (ex.raw_shared) { vout[i] = sum; return; }
      const double val = dscale[i] * sum - beta * vold[i];
      vout[i] = val;
      nrm[i] = val * val;
    }
    return;
  }
  int base = 0, na = 0, sa = 0, nb = 1, sb = 0, local = -1;
  if (d.off_campose >= 0 && i >= d.off_campose && i < d.off_campose + 6 * d.C) {
    const int q = i - d.off_campose, c = q / 6;
    local = q % 6; base = c * d.B; na = d.Fl; sa = CB; nb = d.B; sb = 1;
  } else if (d.off_boardpose >= 0 && i >= d.off_boardpose && i < d.off_boardpose + 6 * d.B) {
    const int q = i - d.off_boardpose, b = q / 6;
    local = 6 * (d.NPB - 1) + q % 6; base = b; na = d.Fl; sa = CB; nb = d.C; sb = d.B;
  } else if (d.off_motion >= 0 && i >= d.off_motion && i < d.off_motion + d.n_motion) {
    const int q = i - d.off_motion;
    if (d.motion == MOTION_HAND_EYE) { local = 6 + q; base = 0; na = d.views(); sa = 1; }
  } else if (d.off_cameras >= 0 && i >= d.off_cameras && i < d.off_cameras + d.C * (5 + d.ND)) {
    const int q = i - d.off_cameras, c = q / (5 + d.ND), qq = q % (5 + d.ND);
    const int lq = qq < 4 ? qq : qq - 1;
    const bool masked = d.cam_kmask != nullptr && ((d.cam_kmask[c] >> lq) & 1u);
    if (qq != 4 && !masked && d.KI > 0) { local = npc + lq; base = c * d.B; na = d.Fl; sa = CB; nb = d.B; sb = 1; }
  }
  double sum = 0.0;
  if (local >= 0) {
    // lane L adds the parts L, L + 64, .. IN THAT ORDER, but eight loads are in flight at a time: the parts were written by other
    // XCDs a kernel ago, and one dependent round trip per part made this kernel 26 us at 1 000 parts per camera entry
    const int total = na * nb;
    constexpr int UNR = 8;
    for (int e0 = lane; e0 < total; e0 += 64 * UNR) {
      double v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int e = e0 + 64 * u, a = e / nb, b_ = e - a * nb;
        v[u] = e < total ? part[(size_t)(base + a * sa + b_ * sb) * part_stride + local] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (e0 + 64 * u < total) sum += v[u];
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) {
    if (ex.raw_shared) { vout[i] = sum; return; }
    const double val = dscale[i] * sum - beta * vold[i];
    vout[i] = val;
    nrm[i] = val * val;
  }
}

// Frame-sharded LSMR: J_h^T u is a sum over views, so its SHARED entries are a sum over the ranks -- one all-reduce
// of ns doubles per product; the entries of a rank's own frames are complete.  comm[s] = raw sum of shared entry s.
__global__ __launch_bounds__(256) void k_lsmr_shard_pack(Dims d, const double* __restrict__ vraw, double* __restrict__ comm) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < d.ns) comm[s] = vraw[d.shared_to_x(s)];
}
// after the all-reduce: the shared entries are finished exactly as k_lsmr_gather finishes them on a single GPU
// (val = dscale sum - beta vold), and nrm[i] = weight_i val_i^2 for EVERY entry (own frames 1, foreign frames 0, shared entries
// on rank 0 only), so that the sum of the ranks' folds is |v|^2
__global__ __launch_bounds__(256) void k_lsmr_shard_finish(Dims d, const double* __restrict__ comm, const double* __restrict__ dscale,
                                                           double beta, const double* __restrict__ vold, double* __restrict__ vout,
                                                           double* __restrict__ nrm, const double* __restrict__ ls,
                                                           int unnormalised = 0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.n) return;
  bool skip = false;
  double inv_beta = 1.0, inv_alpha = 1.0;
  if (ls != nullptr) {
    if (ls[LS_ISTOP] != 0.0) return;
    skip = ls[LS_SKIPV] != 0.0;
    beta = ls[LS_BETA];
    if (unnormalised) inv_beta = ls[LS_INV_BETA];   // (k_lsmr_fused leaves sums of the un-normalised uhat)
    if (unnormalised == 2) inv_alpha = ls[LS_INV_ALPHA];   // (two-launch iteration: v_old is v_raw of the last step, v = v_raw / alpha)
  }
  const int s = d.x_to_shared(i);
  double val = vout[i];
  if (s >= 0) {
    const double vn = unnormalised == 2 ? vold[i] * inv_alpha : vold[i];
    val = skip ? vn : dscale[i] * (unnormalised ? comm[s] * inv_beta : comm[s]) - beta * vn;
    vout[i] = val;
  }
  nrm[i] = d.entry_weight(i) * (val * val);
}
// one double per rank for a sum over all ranks: out[0] = sum_i weight_i a[i] (b == nullptr) or sum_i weight_i a[i] b[i]
__global__ __launch_bounds__(1024) void k_dot_weighted(Dims d, const double* __restrict__ a, const double* __restrict__ b,
                                                       double* __restrict__ out, int three) {
  __shared__ double scratch[16];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < d.n; i += blockDim.x) {
    const double w = d.entry_weight(i), x = a[i], y = b ? b[i] : 1.0;
    s0 += w * (x * y);
    s1 += w * (x * x);
    s2 += w * (y * y);
  }
  const double t0 = block_reduce<false>(s0, scratch);
  const double t1 = block_reduce<false>(s1, scratch);
  const double t2 = block_reduce<false>(s2, scratch);
  if (threadIdx.x == 0) {
    out[0] = t0;
    if (three) { out[1] = t1; out[2] = t2; }
  }
}
// [ |u|^2 partial of this rank, |x|^2 partial of this rank ] in front of k_lsmr_scal_a (folded in k_dot's order)
__global__ __launch_bounds__(1024) void k_lsmr_shard_fold_a(Dims d, const double* __restrict__ upart, int nblk,
                                                            const double* __restrict__ xsq, double* __restrict__ out) {
  __shared__ double scratch[16];
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) s0 += upart[i];
  for (int i = threadIdx.x; i < d.n; i += blockDim.x) s1 += d.entry_weight(i) * xsq[i];
  const double t0 = block_reduce<false>(s0, scratch);
  const double t1 = block_reduce<false>(s1, scratch);
  if (threadIdx.x == 0) { out[0] = t0; out[1] = t1; }
}

// v <- v_raw * inv_alpha;  hbar <- c_hbar hbar + h;  x <- x + c_x hbar;  h <- c_h h + v;  nrm[i] = x[i]^2
// (lsmr.py: "Update h, h_hat, x" with the normalisation of v folded in)
__global__ __launch_bounds__(256) void k_lsmr_update(int n, double inv_alpha, double c_hbar, double c_x, double c_h,
                                                     double* __restrict__ v, double* __restrict__ hbar, double* __restrict__ x,
                                                     double* __restrict__ h, double* __restrict__ nrm,
                                                     const double* __restrict__ ls = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (ls != nullptr) {
    if (ls[LS_ISTOP] != 0.0) return;
    inv_alpha = ls[LS_INV_ALPHA]; c_hbar = ls[LS_C_HBAR]; c_x = ls[LS_C_X]; c_h = ls[LS_C_H];
  }
  const double vi = v[i] * inv_alpha;
  v[i] = vi;
  const double hb = c_hbar * hbar[i] + h[i];
  hbar[i] = hb;
  const double xi = x[i] + c_x * hb;
  x[i] = xi;
  h[i] = c_h * h[i] + vi;
  nrm[i] = xi * xi;
}

// ---------------------------------------------------------------------------------------------------------------
// Round 5: the LSMR iteration in THREE launches (k_lsmr_fused -> k_lsmr_gather2 -> k_lsmr_update2) instead of six.  The two
// single-workgroup scalar kernels are gone: every workgroup of the consuming kernel folds the (few) partial sums itself, in
// the same fixed order, and runs the scalar recurrence redundantly -- no atomics, no "last block" tickets (rejected on this
// multi-XCD part, see mcba_kernels.h), bit-identical scalars in every workgroup.  Workgroup 0 publishes the state.  The state
// is double-buffered so that a workgroup that is late never reads what workgroup 0 has already advanced:
//     k_lsmr_fused    reads A (alpha, 1 / beta_old, stop flag)
//     k_lsmr_gather2  reads A, |uhat|^2 partials, |x|^2 partials -> stopping tests of the previous iteration, beta; writes B
//     k_lsmr_update2  reads B, |v|^2 values -> alpha, plane rotations, update coefficients; writes A
// ---------------------------------------------------------------------------------------------------------------
constexpr int LSG_THREADS = 256;
__device__ __forceinline__ double lsmr_fold256(const double* __restrict__ a, int n, double* scratch) {
  double s = 0.0;
  constexpr int FB = 8;   // loads in flight per thread (see wave_fold_batched)
  for (int i0 = threadIdx.x; i0 < n; i0 += LSG_THREADS * FB) {
    double v[FB];
#pragma unroll
    for (int k = 0; k < FB; ++k) v[k] = a[min(i0 + LSG_THREADS * k, n - 1)];
#pragma unroll
    for (int k = 0; k < FB; ++k)
      if (i0 + LSG_THREADS * k < n) s += v[k];
  }
  const double t = block_reduce<false>(s, scratch);
  __shared__ double bc;
  if (threadIdx.x == 0) bc = t;
  __syncthreads();
  const double r = bc;
  __syncthreads();
  return r;   // the same value in every thread
}

// vout[i] = dscale[i] * (inv_beta * sum over the views that contain parameter i of part[view][local]) - beta * vold[i];
// nrm[i] = vout[i]^2.  Head: beta and the stopping tests (see above).  Tasks, one wavefront each (4 per workgroup): the
// entries outside the per-frame pose block, then one per frame (k_lsmr_gather's two kinds of workgroups).
__global__ __launch_bounds__(LSG_THREADS) void k_lsmr_gather2(Dims d, const double* __restrict__ part, int part_stride,
                                                              const double* __restrict__ dscale, const double* __restrict__ vold,
                                                              double* __restrict__ vout, double* __restrict__ nrm,
                                                              const double* __restrict__ lsA, double* __restrict__ lsB,
                                                              const double* __restrict__ upart, int nu,
                                                              const double* __restrict__ xpart, int nx, unsigned long long call,
                                                              unsigned long long* host_word, LsmrGatherExtra ex) {
  __shared__ double scratch[16];
  __shared__ double head[4];
  if (lsA[LS_ISTOP] != 0.0) return;
  const double u2 = lsmr_fold256(upart, nu, scratch);
  const double x2 = lsmr_fold256(xpart, nx, scratch);
  if (threadIdx.x == 0) {
    double L[LS_NSLOTS];
    for (int k = 0; k < LS_NSLOTS; ++k) L[k] = lsA[k];
    const int istop = L[LS_ITN] > 0.0 ? lsmr_state_test(L, x2) : 0;
    L[LS_X2] = x2;
    if (istop != 0) L[LS_ISTOP] = (double)istop;
    else lsmr_state_beta(L, u2);
    head[0] = L[LS_BETA]; head[1] = L[LS_INV_BETA]; head[2] = L[LS_SKIPV]; head[3] = L[LS_ISTOP];
    if (blockIdx.x == 0) {
      for (int k = 0; k < LS_NSLOTS; ++k) lsB[k] = L[k];
      __hip_atomic_store(host_word, lsmr_progress_word(call, istop, (long long)L[LS_ITN]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (head[3] != 0.0) return;
  const double beta = head[0], inv_beta = head[1];
  const bool skip = head[2] != 0.0;
  const int lane = threadIdx.x & 63;
  const int nfe = lsmr_gather_frame_entries(d), ngen = d.n - nfe;
  const int task = (int)blockIdx.x * (LSG_THREADS / 64) + (threadIdx.x >> 6);
  const int CB = d.C * d.B, npc = 6 * d.NPB;
  if (task >= ngen) {
    const int fl = task - ngen;
    if (fl >= d.Fl || nfe == 0) return;
    const int DFm = d.motion == MOTION_ROLLING ? 12 : 6, g = lane >> 4, l16 = lane & 15;
    for (int e = g; e < DFm; e += 4) {                  // entry e of the frame: chain e / 6, component e % 6
      const int chain = e / 6, i = d.off_motion + chain * 6 * d.F + 6 * (d.f0 + fl) + e % 6;
      if (skip) {
        if (l16 == 0) vout[i] = vold[i];
        continue;
      }
      double s4[4] = {0.0, 0.0, 0.0, 0.0};
      const double* src = part + (size_t)fl * CB * part_stride + 6 + e;
      for (int w0 = 0; w0 < CB; w0 += 64)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int vw = w0 + 16 * k + l16;
          if (vw < CB) s4[k] += src[(size_t)vw * part_stride];
        }
      double sum = (s4[0] + s4[2]) + (s4[1] + s4[3]);
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) sum += __shfl_down(sum, off, 16);
      if (l16 == 0) {
        const double val = dscale[i] * (sum * inv_beta) - beta * vold[i];
        vout[i] = val;
        nrm[i] = val * val;
      }
    }
    return;
  }
  const int i = nfe > 0 && task >= d.off_motion ? task + nfe : task;
  if (i >= d.n) return;
  if (skip && !ex.raw_shared) {
    if (lane == 0) vout[i] = vold[i];
    return;
  }
  double sum = 0.0;
  if (d.off_boards >= 0 && i >= d.off_boards) {   // adjusted board point: sum over the observations of the point
    const int q = (i - d.off_boards) / 3, k = (i - d.off_boards) % 3;
    int b = 0;
    while (q >= ex.board_off[b + 1]) ++b;
    const int p = q - ex.board_off[b], total = d.Fl * d.C;
    for (int e = lane; e < total; e += 64) {
      const int idx = ex.obs_index[((size_t)e * d.B + b) * d.P + p];
      if (idx >= 0) sum += ex.bpart[3 * (size_t)idx + k];
    }
  } else {
    int base = 0, na = 0, sa = 0, nb = 1, sb = 0, local = -1;
    if (d.off_campose >= 0 && i >= d.off_campose && i < d.off_campose + 6 * d.C) {
      const int q = i - d.off_campose, c = q / 6;
      local = q % 6; base = c * d.B; na = d.Fl; sa = CB; nb = d.B; sb = 1;
    } else if (d.off_boardpose >= 0 && i >= d.off_boardpose && i < d.off_boardpose + 6 * d.B) {
      const int q = i - d.off_boardpose, b = q / 6;
      local = 6 * (d.NPB - 1) + q % 6; base = b; na = d.Fl; sa = CB; nb = d.C; sb = d.B;
    } else if (d.off_motion >= 0 && i >= d.off_motion && i < d.off_motion + d.n_motion) {
      const int q = i - d.off_motion;
      if (d.motion == MOTION_HAND_EYE) { local = 6 + q; base = 0; na = d.views(); sa = 1; }
    } else if (d.off_cameras >= 0 && i >= d.off_cameras && i < d.off_cameras + d.C * (5 + d.ND)) {
      const int q = i - d.off_cameras, c = q / (5 + d.ND), qq = q % (5 + d.ND);
      const int lq = qq < 4 ? qq : qq - 1;
      const bool masked = d.cam_kmask != nullptr && ((d.cam_kmask[c] >> lq) & 1u);
      if (qq != 4 && !masked && d.KI > 0) { local = npc + lq; base = c * d.B; na = d.Fl; sa = CB; nb = d.B; sb = 1; }
    }
    if (local >= 0) {
      const int total = na * nb;
      constexpr int UNR = 8;
      for (int e0 = lane; e0 < total; e0 += 64 * UNR) {
        double v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int e = e0 + 64 * u, a = e / nb, b_ = e - a * nb;
          v[u] = e < total ? part[(size_t)(base + a * sa + b_ * sb) * part_stride + local] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (e0 + 64 * u < total) sum += v[u];
      }
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) {
    if (ex.raw_shared) { vout[i] = sum; return; }       // (frame-sharded: summed over the ranks, finished by k_lsmr_shard_finish)
    const double val = dscale[i] * (sum * inv_beta) - beta * vold[i];
    vout[i] = val;
    nrm[i] = val * val;
  }
}

// Two-launch iteration (k_lsmr_fused2 -> k_lsmr_gather3): the gather of k_lsmr_gather2 with
//   * one workgroup per entry outside the frame block, its run split over the four task wavefronts (round 6; frames: four per workgroup);
//   * a FIFTH wavefront per workgroup that folds the |uhat|^2 partials into beta WHILE the four task wavefronts form their sums
//     over the per-view partials -- the sums do not depend on beta; one barrier, then the finish
//     v_raw[i] = D_i (sum / beta) - beta (v_old[i] / alpha)   (v is kept un-normalised: 1 / alpha comes from the state);
//   * the stopping tests of the previous iteration (|x|^2 partials, lsmr_state_test) and the state only in ONE extra workgroup
//     that has no tasks (the last one): nobody waits for them inside the kernel;
//   * per-workgroup partials of |v_raw|^2 (vpart[workgroup]) for the head of the next k_lsmr_fused2 instead of n squares.
// State: in = lsIn (written by k_lsmr_fused2), out = lsOut.
constexpr int LSG3_THREADS = 320;
__global__ __launch_bounds__(LSG3_THREADS) void k_lsmr_gather3(Dims d, const double* __restrict__ part, int part_stride,
                                                               const double* __restrict__ dscale, const double* __restrict__ vold,
                                                               double* __restrict__ vout, double* __restrict__ nrm,
                                                               double* __restrict__ vpart,
                                                               const double* __restrict__ lsIn, double* __restrict__ lsOut,
                                                               const double* __restrict__ upart, int nu,
                                                               const double* __restrict__ xpart, int nx, unsigned long long call,
                                                               unsigned long long* host_word, LsmrGatherExtra ex) {
  __shared__ double head[5];
  __shared__ double wsq[4], qs[4];
  if (lsIn[LS_ISTOP] != 0.0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nfe = lsmr_gather_frame_entries(d), ngen = d.n - nfe;
  const int CB = d.C * d.B;
  (void)part_stride;   // (transposed layout: lsmr_part_index)
  // Workgroups [0, ngen): ONE entry outside the frame block each -- its run of hundreds to tens of thousands of per-view partials is
  // split over the four task wavefronts (quarters, combined through LDS: a board-pose entry of 16 x 1000 x 5 sums 16 000 of them);
  // workgroups behind them: four frames each, one per wavefront; the last workgroup: the publisher (no tasks).
  // (Round 5 gave every wavefront a whole entry, dealt round-robin: the launch then waited for the wavefront with the longest run.)
  const bool publisher = blockIdx.x == gridDim.x - 1;   // one extra workgroup without tasks: stopping tests + state (off the others' path)
  const bool gen_wg = !publisher && (int)blockIdx.x < ngen;
  const int task = publisher ? 0x3fffffff : (gen_wg ? (int)blockIdx.x : ngen + ((int)blockIdx.x - ngen) * 4 + wave);
  // ---- phase 1: sums (task wavefronts) || head (fifth wavefront) ------------------------------------------------------------
  double sum = 0.0, fsum[3] = {0.0, 0.0, 0.0};
  int kind = 0, gi = -1;   // kind 1: general entry gi (lane 0 finishes it); kind 2: frame task (lanes with l16 == 0, entries g, g + 4, g + 8)
  const int DFm = d.motion == MOTION_ROLLING ? 12 : 6, g16 = lane >> 4, l16 = lane & 15;
  int fl = -1;
  // frame-sharded, ONE collective per iteration (ex.raw_shared == 2): this kernel only forms the raw sums of EVERY entry -- beta needs
  // |uhat|^2 over all ranks, which travels in the same message as the shared sums (k_lsmr_shard_pack2 / k_lsmr_shard_finish2)
  const bool raw_all = ex.raw_shared == 2;
  if (wave == 4) {
    if (raw_all) {
      if (lane == 0) { head[0] = 0.0; head[1] = 1.0; head[2] = 0.0; head[3] = 0.0; head[4] = 1.0; }
    } else {
    // every workgroup: beta = |uhat| and 1 / beta (lsmr_state_beta), nothing else -- the tasks do not wait for the stopping tests
    // (what a stopped iteration writes is never read: the next k_lsmr_fused2 returns on the flag)
    const double inv_alpha_cur = lsIn[LS_INV_ALPHA];
    const double u2 = wave_fold_batched<16>(upart, nu, lane);
    const double beta = sqrt(u2);
    if (lane == 0) {
      head[0] = beta; head[1] = beta > 0 ? 1.0 / beta : 1.0; head[2] = beta > 0 ? 0.0 : 1.0; head[3] = 0.0; head[4] = inv_alpha_cur;
    }
    if (publisher) {
      double L[LS_NSLOTS];
#pragma unroll
      for (int k = 0; k < LS_NSLOTS; ++k) L[k] = lsIn[k];
      const double x2 = wave_fold_batched<4>(xpart, nx, lane);
      const int istop = L[LS_ITN] > 0.0 ? lsmr_state_test(L, x2) : 0;
      L[LS_X2] = x2;
      if (istop != 0) L[LS_ISTOP] = (double)istop;
      else lsmr_state_beta(L, u2);
      L[LS_PENDING] = 1.0;
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < LS_NSLOTS; ++k) lsOut[k] = L[k];
        __hip_atomic_store(host_word, lsmr_progress_word(call, istop, (long long)L[LS_ITN]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    }
  } else if (task >= ngen) {
    fl = task - ngen;
    if (fl < d.Fl && nfe > 0) {
      kind = 2;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int e = g16 + 4 * q;
        if (e < DFm) {
          double s4[4] = {0.0, 0.0, 0.0, 0.0};
          const double* src = part + lsmr_part_motion(d) + ((size_t)fl * DFm + e) * CB;      // (contiguous: lsmr_part_index)
          for (int w0 = 0; w0 < CB; w0 += 64)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int vw = w0 + 16 * k + l16;
              if (vw < CB) s4[k] += src[vw];
            }
          double sq = (s4[0] + s4[2]) + (s4[1] + s4[3]);
#pragma unroll
          for (int off = 8; off > 0; off >>= 1) sq += __shfl_down(sq, off, 16);
          fsum[q] = sq;
        }
      }
    }
  } else {
    const int i = nfe > 0 && task >= d.off_motion ? task + nfe : task;
    if (i < d.n) {
      kind = 1;
      gi = i;
      if (d.off_boards >= 0 && i >= d.off_boards) {   // adjusted board point: sum over the observations of the point
        const int q = (i - d.off_boards) / 3, k = (i - d.off_boards) % 3;
        int b = 0;
        while (q >= ex.board_off[b + 1]) ++b;
        const int p = q - ex.board_off[b], total = d.Fl * d.C;
        for (int e = wave * 64 + lane; e < total; e += 256) {
          const int idx = ex.obs_index[((size_t)e * d.B + b) * d.P + p];
          if (idx >= 0) sum += ex.bpart[3 * (size_t)idx + k];
        }
      } else {
        // the contiguous run of the entry in the transposed layout (lsmr_part_index)
        const double* run = nullptr;
        int total = 0;
        if (d.off_campose >= 0 && i >= d.off_campose && i < d.off_campose + 6 * d.C) {
          const int q = i - d.off_campose;
          run = part + lsmr_part_cam(d, q / 6, q % 6); total = d.Fl * d.B;
        } else if (d.off_boardpose >= 0 && i >= d.off_boardpose && i < d.off_boardpose + 6 * d.B) {
          const int q = i - d.off_boardpose;
          run = part + lsmr_part_board(d, q / 6, q % 6); total = d.Fl * d.C;
        } else if (d.off_motion >= 0 && i >= d.off_motion && i < d.off_motion + d.n_motion) {
          const int q = i - d.off_motion;
          if (d.motion == MOTION_HAND_EYE) { run = part + lsmr_part_motion(d) + (size_t)q * d.views(); total = d.views(); }
        } else if (d.off_cameras >= 0 && i >= d.off_cameras && i < d.off_cameras + d.C * (5 + d.ND)) {
          const int q = i - d.off_cameras, c = q / (5 + d.ND), qq = q % (5 + d.ND);
          const int lq = qq < 4 ? qq : qq - 1;
          const bool masked = d.cam_kmask != nullptr && ((d.cam_kmask[c] >> lq) & 1u);
          if (qq != 4 && !masked && d.KI > 0) { run = part + lsmr_part_cam(d, c, 6 + lq); total = d.Fl * d.B; }
        }
        if (run != nullptr) {
          constexpr int UNR = 8;
          const int q0 = (int)(((long long)total * wave) / 4), q1 = (int)(((long long)total * (wave + 1)) / 4);   // this wavefront's quarter
          for (int e0 = q0 + lane; e0 < q1; e0 += 64 * UNR) {
            double v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = run[min(e0 + 64 * u, q1 - 1)];
#pragma unroll
            for (int u = 0; u < UNR; ++u)
              if (e0 + 64 * u < q1) sum += v[u];
          }
        }
      }
      sum = wave_sum(sum);
      if (lane == 0) qs[wave] = sum;
    }
  }
  // (what the finish reads besides beta is requested in front of the barrier: one round trip less behind it)
  double pvo[3] = {0.0, 0.0, 0.0}, pds[3] = {0.0, 0.0, 0.0};
  if (kind == 2) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int e = min(g16 + 4 * q, DFm - 1);
      const int chain = e / 6, i = d.off_motion + chain * 6 * d.F + 6 * (d.f0 + fl) + e % 6;
      pvo[q] = vold[i];
      pds[q] = dscale[i];
    }
  } else if (kind == 1 && wave == 0) {
    pvo[0] = vold[gi];
    pds[0] = dscale[gi];
  }
  __syncthreads();
  if (kind == 1) sum = (qs[0] + qs[1]) + (qs[2] + qs[3]);   // (the four quarters of the entry's run, fixed order)
  // ---- phase 2: finish with beta ------------------------------------------------------------------------------------------------
  const double beta = head[0], inv_beta = head[1], inv_alpha = head[4];
  const bool skip = head[2] != 0.0;
  double vsq = 0.0;
  if (raw_all) {
    if (kind == 2) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int e = g16 + 4 * q;
        if (e < DFm && l16 == 0) vout[d.off_motion + (e / 6) * 6 * d.F + 6 * (d.f0 + fl) + e % 6] = fsum[q];
      }
    } else if (kind == 1 && wave == 0 && lane == 0) {
      vout[gi] = sum;
    }
    return;     // (no barrier behind this point)
  }
  if (kind == 2) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int e = g16 + 4 * q;
      if (e < DFm && l16 == 0) {
        const int chain = e / 6, i = d.off_motion + chain * 6 * d.F + 6 * (d.f0 + fl) + e % 6;
        const double vn = pvo[q] * inv_alpha;
        const double val = skip ? vn : pds[q] * (fsum[q] * inv_beta) - beta * vn;
        vout[i] = val;
        nrm[i] = val * val;
        vsq += val * val;
      }
    }
  } else if (kind == 1 && wave == 0 && lane == 0) {
    if (ex.raw_shared) {
      vout[gi] = sum;          // (frame-sharded: summed over the ranks, finished by k_lsmr_shard_finish)
    } else {
      const double vn = pvo[0] * inv_alpha;
      const double val = skip ? vn : pds[0] * (sum * inv_beta) - beta * vn;
      vout[gi] = val;
      nrm[gi] = val * val;
      vsq = val * val;
    }
  }
  if (wave < 4) {
    vsq = wave_sum(vsq);
    if (lane == 0) wsq[wave] = vsq;
  }
  __syncthreads();
  if (threadIdx.x == 0) vpart[blockIdx.x] = (wsq[0] + wsq[1]) + (wsq[2] + wsq[3]);
}

// head: |v|^2 (vpart[0 .. nv)) -> alpha, the plane rotations and the coefficients of the update (lsmr_state_rotate); state B -> A.
// body: v <- v_raw / alpha;  hbar <- c_hbar hbar + h;  x <- x + c_x hbar;  h <- c_h h + v;  xpart[workgroup] = sum_i w_i x_i^2
// (w = 1; frame-sharded: entry_weight, so that the ranks' partials add up to |x|^2)
__global__ __launch_bounds__(LSG_THREADS) void k_lsmr_update2(Dims d, const double* __restrict__ lsB, double* __restrict__ lsA,
                                                              const double* __restrict__ vpart, int nv, double* __restrict__ v,
                                                              double* __restrict__ hbar, double* __restrict__ x, double* __restrict__ h,
                                                              double* __restrict__ xpart) {
  __shared__ double scratch[16];
  __shared__ double head[5];
  if (lsB[LS_ISTOP] != 0.0) {   // stopped by the tests in front of this update: hand the flag on, leave x as it is
    if (blockIdx.x == 0 && threadIdx.x == 0) lsA[LS_ISTOP] = lsB[LS_ISTOP];
    return;
  }
  const double v2 = lsmr_fold256(vpart, nv, scratch);
  if (threadIdx.x == 0) {
    double L[LS_NSLOTS];
    for (int k = 0; k < LS_NSLOTS; ++k) L[k] = lsB[k];
    lsmr_state_rotate(L, v2);
    head[0] = L[LS_INV_ALPHA]; head[1] = L[LS_C_HBAR]; head[2] = L[LS_C_X]; head[3] = L[LS_C_H];
    if (blockIdx.x == 0)
      for (int k = 0; k < LS_NSLOTS; ++k) lsA[k] = L[k];
  }
  __syncthreads();
  const double inv_alpha = head[0], c_hbar = head[1], c_x = head[2], c_h = head[3];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double sq = 0.0;
  if (i < d.n) {
    const double vi = v[i] * inv_alpha;
    v[i] = vi;
    const double hb = c_hbar * hbar[i] + h[i];
    hbar[i] = hb;
    const double xi = x[i] + c_x * hb;
    x[i] = xi;
    h[i] = c_h * h[i] + vi;
    sq = d.entry_weight(i) * (xi * xi);
  }
  const double tot = block_reduce<false>(sq, scratch);
  if (threadIdx.x == 0) xpart[blockIdx.x] = tot;
}

// frame-sharded form of the fused iteration: [ |uhat|^2 partial of this rank, |x|^2 partial of this rank ] (plain sums of partials)
__global__ __launch_bounds__(LSG_THREADS) void k_lsmr_shard_fold_a2(const double* __restrict__ upart, int nu,
                                                                   const double* __restrict__ xpart, int nx, double* __restrict__ out) {
  __shared__ double scratch[16];
  const double a = lsmr_fold256(upart, nu, scratch), b = lsmr_fold256(xpart, nx, scratch);
  if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
}

// ---------------------------------------------------------------------------------------------------------------
// Frame-sharded two-launch iteration with ONE collective per LSMR iteration (round 6).  Everything an iteration needs from the other
// ranks is known BEFORE beta is: the raw shared sums s = sum_views J^T uhat (ns doubles), |uhat|^2 and |x|^2 (per-rank partials) and
// three sums over the rank's OWN frame entries that give the frame part of |v_raw|^2 for any beta:
//     v_raw_i = D_i s_i / beta - beta vn_i   =>   sum_i v_raw_i^2 = a / beta^2 - 2 b + beta^2 c,
//     a = sum (D_i s_i)^2,  b = sum D_i s_i vn_i,  c = sum vn_i^2          (vn = v_old / alpha: the normalised v of the step)
// message = [s (ns) | |uhat|^2 | |x|^2 | a | b | c]; after the all-reduce EVERY rank forms beta, runs the stopping tests of the previous
// step, finishes v_raw (shared entries from the message, own frame entries from its own raw sums) and |v_raw|^2 = (shared part, summed
// in a fixed order: bit-identical on all ranks) + (a / beta^2 - 2 b + beta^2 c) -- so the state stays bit-identical across the ranks.
// Both kernels are ONE workgroup: n is a few thousand entries, and a single fold order is what makes the scalars rank-independent.
// ---------------------------------------------------------------------------------------------------------------
constexpr int LSP_THREADS = 1024;
__global__ __launch_bounds__(LSP_THREADS) void k_lsmr_shard_pack2(Dims d, const double* __restrict__ vraw, const double* __restrict__ vold,
                                                                   const double* __restrict__ dscale, const double* __restrict__ ls,
                                                                   const double* __restrict__ upart, int nu,
                                                                   const double* __restrict__ xpart, int nx, double* __restrict__ comm) {
  __shared__ double scratch[16];
  const int ns = d.ns;
  if (ls[LS_ISTOP] != 0.0) {   // stopped: the message of an iteration enqueued behind the stop is never read (zeros keep it finite)
    for (int s = threadIdx.x; s < ns + 5; s += blockDim.x) comm[s] = 0.0;
    return;
  }
  const double inv_alpha = ls[LS_INV_ALPHA];
  for (int s = threadIdx.x; s < ns; s += blockDim.x) comm[s] = vraw[d.shared_to_x(s)];
  double u2 = 0.0, x2 = 0.0, a = 0.0, b = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nu; i += blockDim.x) u2 += upart[i];
  for (int i = threadIdx.x; i < nx; i += blockDim.x) x2 += xpart[i];
  if (d.DF > 0 && d.off_motion >= 0) {
    const int nfe = d.Fl * d.DF;     // own frame entries: local frame fl, eliminated parameter q
    for (int k = threadIdx.x; k < nfe; k += blockDim.x) {
      const int i = d.frame_to_x(d.f0 + k / d.DF, k % d.DF);
      const double t = dscale[i] * vraw[i], vn = vold[i] * inv_alpha;
      a += t * t;
      b += t * vn;
      c += vn * vn;
    }
  }
  const double U = block_reduce<false>(u2, scratch), X = block_reduce<false>(x2, scratch);
  const double A = block_reduce<false>(a, scratch), B = block_reduce<false>(b, scratch), Cc = block_reduce<false>(c, scratch);
  if (threadIdx.x == 0) { comm[ns] = U; comm[ns + 1] = X; comm[ns + 2] = A; comm[ns + 3] = B; comm[ns + 4] = Cc; }
}

// state: in = lsIn (written by k_lsmr_fused2), out = lsOut; vsq[0] = |v_raw|^2 for the head of the next k_lsmr_fused2
__global__ __launch_bounds__(LSP_THREADS) void k_lsmr_shard_finish2(Dims d, const double* __restrict__ comm, const double* __restrict__ dscale,
                                                                     const double* __restrict__ vold, double* __restrict__ vout,
                                                                     const double* __restrict__ lsIn, double* __restrict__ lsOut,
                                                                     double* __restrict__ vsq, unsigned long long call,
                                                                     unsigned long long* host_word) {
  __shared__ double scratch[16];
  __shared__ double head[6];
  if (lsIn[LS_ISTOP] != 0.0) return;   // (lsOut keeps the flag the stopping iteration left there)
  const int ns = d.ns;
  if (threadIdx.x == 0) {
    double L[LS_NSLOTS];
    for (int k = 0; k < LS_NSLOTS; ++k) L[k] = lsIn[k];
    const double u2 = comm[ns], x2 = comm[ns + 1];
    const int istop = L[LS_ITN] > 0.0 ? lsmr_state_test(L, x2) : 0;
    L[LS_X2] = x2;
    if (istop != 0) L[LS_ISTOP] = (double)istop;
    else lsmr_state_beta(L, u2);
    L[LS_PENDING] = 1.0;
    for (int k = 0; k < LS_NSLOTS; ++k) lsOut[k] = L[k];
    head[0] = L[LS_BETA]; head[1] = L[LS_INV_BETA]; head[2] = L[LS_SKIPV]; head[3] = (double)istop; head[4] = L[LS_INV_ALPHA];
    __hip_atomic_store(host_word, lsmr_progress_word(call, istop, (long long)L[LS_ITN]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  if (head[3] != 0.0) return;
  const double beta = head[0], inv_beta = head[1], inv_alpha = head[4];
  const bool skip = head[2] != 0.0;
  double sq = 0.0;
  for (int i = threadIdx.x; i < d.n; i += blockDim.x) {
    const int s = d.x_to_shared(i);
    double sum;
    if (s >= 0) sum = comm[s];
    else if (d.entry_weight(i) != 0.0) sum = vout[i];     // own frame entry: the raw sum k_lsmr_gather3 left there
    else continue;                                         // a frame of another rank
    const double vn = vold[i] * inv_alpha;
    const double val = skip ? vn : dscale[i] * (sum * inv_beta) - beta * vn;
    vout[i] = val;
    if (s >= 0) sq += val * val;
  }
  const double shared = block_reduce<false>(sq, scratch);
  if (threadIdx.x == 0) {
    const double a = comm[ns + 2], b = comm[ns + 3], c = comm[ns + 4];
    const double frames = skip ? c : (a * inv_beta) * inv_beta - 2.0 * b + (beta * beta) * c;
    vsq[0] = shared + (frames > 0.0 ? frames : 0.0);
  }
}

// out[0] = sum a[i] b[i] (one workgroup, fixed order); out[1] = sum a[i]^2, out[2] = sum b[i]^2 when three != 0
__global__ __launch_bounds__(1024) void k_dot(size_t n, const double* __restrict__ a, const double* __restrict__ b,
                                               double* __restrict__ out, int three) {
  __shared__ double scratch[16];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
    const double x = a[i], y = b ? b[i] : 1.0;
    s0 += x * y;
    s1 += x * x;
    s2 += y * y;
  }
  const double t0 = block_reduce<false>(s0, scratch);
  const double t1 = block_reduce<false>(s1, scratch);
  const double t2 = block_reduce<false>(s2, scratch);
  if (threadIdx.x == 0) {
    out[0] = t0;
    if (three) { out[1] = t1; out[2] = t2; }
  }
}

// The same three sums over a LONG vector (the m residual rows of the subspace products J g_h, J gn) -- in k_dot's OWN order, so that not
// a bit of a solve changes: k_dot's single workgroup walks 1.5 M entries with one load round trip per step (0.49 ms per call at the
// north-star rig, 4 % of a default solve).  Here each of its 16 wavefronts is a workgroup of its own (16 CUs pull the vectors instead of
// one), lane t of wavefront w still adds the entries 64 w + t, + 1024, + 2048, ... one after the other -- eight loads in flight per
// operand instead of one -- and k_dot3_fin adds the 16 wavefront totals in block_reduce's order.
constexpr int DOT_WAVES = 16;   // = k_dot's 1024 threads / 64
__global__ __launch_bounds__(64) void k_dot3_part(size_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                  double* __restrict__ part /*[3][DOT_WAVES]*/) {
  constexpr size_t STRIDE = 64 * DOT_WAVES;
  constexpr int NB = 8;    // loads in flight per operand and lane (106 us per call at m = 1.5 M; 16 in flight: 151 us)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n; i += NB * STRIDE) {   // (lanes past the end: no iteration, as in k_dot)
    double x[NB], y[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const size_t j = i + u * STRIDE, jc = j < n ? j : i;   // (clamped: the load is unconditional, the sum below is not)
      x[u] = a[jc];
      y[u] = b[jc];
    }
#pragma unroll
    for (int u = 0; u < NB; ++u)
      if (i + u * STRIDE < n) {
        s0 += x[u] * y[u];
        s1 += x[u] * x[u];
        s2 += y[u] * y[u];
      }
  }
  const double t0 = wave_sum(s0), t1 = wave_sum(s1), t2 = wave_sum(s2);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = t0;
    part[DOT_WAVES + blockIdx.x] = t1;
    part[2 * DOT_WAVES + blockIdx.x] = t2;
  }
}
// out[k] = the 16 wavefront totals of value k, added in the order of block_reduce (scratch[0] + scratch[1] + ...)
__global__ void k_dot3_fin(const double* __restrict__ part, double* __restrict__ out) {
  if (threadIdx.x < 3) {
    double r = part[threadIdx.x * DOT_WAVES];
    for (int i = 1; i < DOT_WAVES; ++i) r += part[threadIdx.x * DOT_WAVES + i];
    out[threadIdx.x] = r;
  }
}

// The scalar side of a device-resident LSMR solve (mcba_lsmr.h), two launches of ONE workgroup per iteration:
//   k_lsmr_scal_a behind k_lsmr_jv:     |x|^2 of the PREVIOUS iteration's update -> its stopping tests; |u|^2 -> beta
//   k_lsmr_scal_b behind k_lsmr_gather: |v|^2 -> alpha, the plane rotations, the coefficients of k_lsmr_update
// Sums in the order of k_dot (thread t adds the entries t, t + 1024, ..; waves, then the wave totals in order).  The stopping
// tests of an iteration run after the NEXT k_lsmr_jv: that product is wasted once per solve and every kernel behind a stop is
// an empty launch, in exchange for one scalar launch less per iteration.  Progress goes to a word in pinned host memory.
__device__ __forceinline__ double lsmr_fold(const double* __restrict__ a, int n, double* scratch) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += a[i] * 1.0;
  return block_reduce<false>(s, scratch);
}
__global__ __launch_bounds__(64) void k_lsmr_init(double* __restrict__ ls, double alpha, double beta, double damp, double normb,
                                                  double maxiter) {
  if (threadIdx.x == 0) lsmr_state_init(ls, alpha, beta, damp, normb, maxiter);
}
__global__ __launch_bounds__(1024) void k_lsmr_scal_a(double* __restrict__ ls, const double* __restrict__ upart, int nblk,
                                                      const double* __restrict__ xsq, int n, unsigned long long call,
                                                      unsigned long long* host_word) {
  __shared__ double scratch[16];
  if (ls[LS_ISTOP] != 0.0) return;
  const bool test = ls[LS_ITN] > 0.0;
  const double u2 = lsmr_fold(upart, nblk, scratch);
  const double x2 = test ? lsmr_fold(xsq, n, scratch) : 0.0;
  if (threadIdx.x == 0) {
    const int istop = test ? lsmr_state_test(ls, x2) : 0;
    ls[LS_X2] = x2;
    if (istop != 0) ls[LS_ISTOP] = (double)istop;
    else lsmr_state_beta(ls, u2);
    // (relaxed: the host reads nothing but the word itself -- a system-scope RELEASE here would first write back the 12 MB of u that
    //  k_lsmr_jv has just left in L2)
    __hip_atomic_store(host_word, lsmr_progress_word(call, istop, (long long)ls[LS_ITN]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(1024) void k_lsmr_scal_b(double* __restrict__ ls, const double* __restrict__ vsq, int n) {
  __shared__ double scratch[16];
  if (ls[LS_ISTOP] != 0.0) return;
  const double v2 = lsmr_fold(vsq, n, scratch);
  if (threadIdx.x == 0) lsmr_state_rotate(ls, v2);
}

// y[i] = a * x[i] (+ y0 copy helpers of the driver)
__global__ __launch_bounds__(256) void k_scale_to(int n, double a, const double* x, double* y) {   // (x may be y)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i];
}

// ---------------------------------------------------------------------------------------------------------------
// Frame-sharded handles: the messages of the trust-region iteration.  No n-vector crosses the ranks: every
// message is a handful of per-rank partial sums or the SHARED entries of [g | diag], independent of the number of frames.
// Per-rank partials travel as "gather by sum": rank r writes its values into block r of a zeroed buffer, the all-reduce adds
// the buffers, and every rank then folds the blocks in rank order with the very kernels that fold the per-workgroup partials
// of a single GPU (tr_reg_wave / tr_step_wave) -- the same arithmetic on every rank, and a max (|g|_inf) needs no second
// collective.
// ---------------------------------------------------------------------------------------------------------------
// message 1 (after the assembly): comm = [g_s (ns) | diag_s (ns) | cost, count | |p_h|^2, |step|^2, |x|^2 partial sums of the
// step that led to this point (k_vec_step partials, own entries only) | trial cost partial (retries)]
constexpr int SHARD_TAIL = 6;
__global__ __launch_bounds__(256) void k_shard_pack1(Dims d, const double* __restrict__ g, const double* __restrict__ diag,
                                                     const double* __restrict__ cost_count, const double* __restrict__ step_part,
                                                     int nvb, double* __restrict__ comm) {
  const int ns = d.ns;
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < ns; s += gridDim.x * blockDim.x) {
    const int xi = d.shared_to_x(s);
    comm[s] = g[xi];
    comm[ns + s] = diag[xi];
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const int lane = threadIdx.x;
    double s3[3] = {0, 0, 0};
    if (step_part != nullptr)
      for (int b = lane; b < nvb; b += 64)
        for (int k = 0; k < 3; ++k) s3[k] += step_part[3 * b + k];
    for (int k = 0; k < 3; ++k) s3[k] = wave_sum(s3[k]);
    if (lane == 0) {
      comm[2 * ns] = cost_count[0];
      comm[2 * ns + 1] = cost_count[1];
      for (int k = 0; k < 3; ++k) comm[2 * ns + 2 + k] = s3[k];
      comm[2 * ns + 5] = 0.0;
    }
  }
}
// ... and back: the shared entries of g / diag, {cost, count}, and the step norms as block 0 of the k_vec_step partials (the
// other blocks zero), where the host's fold expects them
__global__ __launch_bounds__(256) void k_shard_unpack1(Dims d, const double* __restrict__ comm, double* __restrict__ g,
                                                       double* __restrict__ diag, double* __restrict__ cost_count,
                                                       double* __restrict__ step_part, int nvb) {
  const int ns = d.ns;
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < ns; s += gridDim.x * blockDim.x) {
    const int xi = d.shared_to_x(s);
    g[xi] = comm[s];
    diag[xi] = comm[ns + s];
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x < 2) cost_count[threadIdx.x] = comm[2 * ns + threadIdx.x];
    if (step_part != nullptr)
      for (int e = threadIdx.x; e < 3 * nvb; e += blockDim.x) step_part[e] = e < 3 ? comm[2 * ns + 2 + e] : 0.0;
  }
}
// message 2 (after the gradient scaling + Cauchy curvature): block r of `vs` = [|g|_inf, |g_h|^2, |x scale|^2] of rank r,
// q[r] = its curvature partial; out = [vs (3 W) | q (W)], everything but this rank's entries zero
__global__ __launch_bounds__(64) void k_shard_fold2(const double* __restrict__ vs_part, int nvb, const double* __restrict__ q_part,
                                                    int nq, int rank, int world, double* __restrict__ out) {
  const int lane = threadIdx.x;
  double mx = 0, gg = 0, xs = 0, q = 0;
  for (int b = lane; b < nvb; b += 64) {
    mx = fmax(mx, vs_part[3 * b]);
    gg += vs_part[3 * b + 1];
    xs += vs_part[3 * b + 2];
  }
  for (int b = lane; b < nq; b += 64) q += q_part[b];
  mx = wave_max(mx);
  gg = wave_sum(gg);
  xs = wave_sum(xs);
  q = wave_sum(q);
  mx = __shfl(mx, 0, 64); gg = __shfl(gg, 0, 64); xs = __shfl(xs, 0, 64); q = __shfl(q, 0, 64);
  for (int e = lane; e < 4 * world; e += 64)   // (every entry written once, by one lane)
    out[e] = e == 3 * rank ? mx : (e == 3 * rank + 1 ? gg : (e == 3 * rank + 2 ? xs : (e == 3 * world + rank ? q : 0.0)));
}
// message 4 (after the back substitution): the partial dots {g_h.g_h, g_h.gn, gn.gn} of this rank (own frames; rank 0: +
// the shared entries) as block `rank` of [3 W | info]; the pivot report is replicated: rank 0 contributes it
__global__ __launch_bounds__(64) void k_shard_fold_dots(const double* __restrict__ dot_part, int nblk, int rank, int world,
                                                        double* __restrict__ out) {
  const int lane = threadIdx.x;
  double dt[3] = {0, 0, 0};
  for (int b = lane; b < nblk; b += 64)
    for (int k = 0; k < 3; ++k) dt[k] += dot_part[3 * b + k];
  for (int k = 0; k < 3; ++k) dt[k] = wave_sum(dt[k]);
  for (int k = 0; k < 3; ++k) dt[k] = __shfl(dt[k], 0, 64);
  const double info = rank == 0 ? dot_part[3 * nblk] : 0.0;
  for (int e = lane; e < 3 * world + 1; e += 64)
    out[e] = e == 3 * world ? info : (e == 3 * rank ? dt[0] : (e == 3 * rank + 1 ? dt[1] : (e == 3 * rank + 2 ? dt[2] : 0.0)));
}
// retry after a rejected step: [trial cost | step norms] of this rank -> comm[0 .. 4) (summed over the ranks), and back
__global__ __launch_bounds__(64) void k_shard_trial_pack(const double* __restrict__ cost_part, int ncost,
                                                         const double* __restrict__ step_part, int nvb, double* __restrict__ comm) {
  const int lane = threadIdx.x;
  double c = 0, s3[3] = {0, 0, 0};
  for (int b = lane; b < ncost; b += 64) c += cost_part[b];
  for (int b = lane; b < nvb; b += 64)
    for (int k = 0; k < 3; ++k) s3[k] += step_part[3 * b + k];
  c = wave_sum(c);
  for (int k = 0; k < 3; ++k) s3[k] = wave_sum(s3[k]);
  if (lane == 0) {
    comm[0] = c;
    for (int k = 0; k < 3; ++k) comm[1 + k] = s3[k];
  }
}
__global__ __launch_bounds__(256) void k_shard_trial_unpack(const double* __restrict__ comm, double* __restrict__ cost_part,
                                                            double* __restrict__ step_part, int nvb) {
  if (threadIdx.x == 0) cost_part[0] = comm[0];
  for (int e = threadIdx.x; e < 3 * nvb; e += blockDim.x) step_part[e] = e < 3 ? comm[1 + e] : 0.0;
}
// the eliminated frame parameters of this rank's frames, zeros elsewhere (the all-reduce that follows is an all-gather: the
// complete x a solve returns, and the complete g / diag of the host-boundary evaluation)
__global__ __launch_bounds__(256) void k_shard_own_frames(Dims d, const double* __restrict__ v, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.n_motion) {
    const int xi = d.off_motion + i;
    out[i] = d.entry_weight(xi) != 0.0 ? v[xi] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// one-wave kernels that keep the scalar trust-region algebra on the device between the vector kernels (single-GPU
// driver: one host synchronisation per iteration).  S = scal[0 .. TR_NSLOTS), see mcba_trmath.h.
// ---------------------------------------------------------------------------------------------------------------
// (stand-alone form of tr_reg_wave: problems without eliminated frame blocks, where k_schur_frame is not launched)
__global__ __launch_bounds__(64) void k_tr_reg(double* __restrict__ S, const double* __restrict__ vs_part, int nvb,
                                               const double* __restrict__ q_part, int nq, int first, double Delta_in) {
  tr_reg_wave(S, vs_part, nvb, q_part, nq, first, Delta_in, threadIdx.x);
}


// ---------------------------------------------------------------------------------------------------------------
// outlier loop on the device (Calibration.reject_outliers / report, calibration.py:240-252,290-310):
// exact order statistics of the per-point reprojection errors by radix select (6 passes over 11/9-bit digits of the
// IEEE bit pattern -- errors are >= 0, so the unsigned bit pattern is monotone), error sums, threshold + mask update.
// Integer atomics only: results are deterministic.  mask2 may be null (mask = evalid) or the inlier table.
// ---------------------------------------------------------------------------------------------------------------
struct SelState {
  unsigned long long prefix;   // bits fixed so far (high bits), rest zero
  long long rank;              // remaining rank inside the current prefix bucket
};

// ---- several order statistics in the same six data passes (report: five quantiles = five selections) ----------------
constexpr int SEL_MAX = 6;   // selections per batch: SEL_MAX x 2048 privatised bins = 48 KB of LDS

__global__ void k_selm_init(SelState* st, int nsel, const long long* __restrict__ ranks, unsigned int* hist) {
  if (threadIdx.x < nsel) { st[threadIdx.x].prefix = 0ull; st[threadIdx.x].rank = ranks[threadIdx.x]; }
  for (int i = threadIdx.x; i < SEL_MAX * 2048; i += blockDim.x) hist[i] = 0u;
}

// pass `first`: nothing is fixed yet, one histogram (hist[0]) serves every selection; later passes: selection r counts the
// elements that share its prefix into hist[r]
__global__ __launch_bounds__(256) void k_selm_hist(const double* __restrict__ err, const uint8_t* __restrict__ m1,
                                                   const uint8_t* __restrict__ m2, int n, const SelState* __restrict__ st,
                                                   int nsel, int shift, int bits, int first,
                                                   unsigned int* __restrict__ hist) {
  __shared__ unsigned int lh[SEL_MAX * 2048];
  const int nh = first ? 1 : nsel;
  for (int i = threadIdx.x; i < nh * 2048; i += blockDim.x) lh[i] = 0u;
  __syncthreads();
  unsigned long long pre[SEL_MAX];
#pragma unroll
  for (int r = 0; r < SEL_MAX; ++r) pre[r] = r < nsel ? st[r].prefix >> (shift + bits) : ~0ull;
  const unsigned int mask = (1u << bits) - 1u;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!m1[i] || (m2 != nullptr && !m2[i])) continue;
    const unsigned long long key = (unsigned long long)__double_as_longlong(err[i]);
    const unsigned int digit = (unsigned int)(key >> shift) & mask;
    if (first) {
      atomicAdd(&lh[digit], 1u);
    } else {
      const unsigned long long top = key >> (shift + bits);
#pragma unroll
      for (int r = 0; r < SEL_MAX; ++r)
        if (r < nsel && top == pre[r]) atomicAdd(&lh[r * 2048 + digit], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nh * 2048; i += blockDim.x)
    if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// one wavefront per selection: lane l owns bins [32 l, 32 l + 32); wave-level exclusive scan of the lane totals finds the
// lane whose range holds the rank, that lane walks its 32 bins (a single thread walking 2048 bins in global memory with
// an early exit cost 134 us per pass)
__global__ __launch_bounds__(64 * SEL_MAX) void k_selm_pick(SelState* st, int nsel, unsigned int* hist, int shift, int bits,
                                                            int first) {
  __shared__ unsigned int lh[SEL_MAX * 2048];
  const int nh = first ? 1 : nsel;
  for (int i = threadIdx.x; i < nh * 2048; i += blockDim.x) lh[i] = hist[i];
  __syncthreads();
  const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (r < nsel) {
    const unsigned int* hr = lh + (first ? 0 : r * 2048);
    const int nb = 1 << bits, per = 32;
    long long tot = 0;
    for (int k = 0; k < per; ++k) {
      const int bin = lane * per + k;
      tot += bin < nb ? hr[bin] : 0u;
    }
    long long incl = tot;                                  // inclusive scan over the lanes
    for (int off = 1; off < 64; off <<= 1) {
      const long long o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    const long long excl = incl - tot, rank = st[r].rank;
    // the owner: first lane whose inclusive count exceeds the rank (the last non-empty range if the rank is past the end)
    const bool mine = rank >= excl && rank < incl;
    const unsigned long long m = __ballot(mine);
    const int owner = m ? __ffsll((long long)m) - 1 : min(63, (nb - 1) / per);
    if (lane == owner) {
      long long rem = rank - excl;
      int bb = lane * per;
      const int last = min(nb - 1, lane * per + per - 1);
      for (; bb < last; ++bb) {
        const long long c = hr[bb];
        if (rem < c) break;
        rem -= c;
      }
      st[r].prefix |= ((unsigned long long)bb) << shift;
      st[r].rank = rem;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SEL_MAX * 2048; i += blockDim.x) hist[i] = 0u;
}

// part[(blk * SEL_MAX + r) * 2 + {0, 1}] = {#elements <= v_r, bit pattern of the smallest element > v_r (+inf if none)}
// of the block; folded by the host (atomics on 2 nsel addresses from 4096 waves serialise at ~75 ns each: 200 us)
__global__ __launch_bounds__(256) void k_selm_next(const double* __restrict__ err, const uint8_t* __restrict__ m1,
                                                   const uint8_t* __restrict__ m2, int n, const SelState* __restrict__ st,
                                                   int nsel, unsigned long long* __restrict__ part) {
  __shared__ unsigned long long red[4][2 * SEL_MAX];
  unsigned long long v[SEL_MAX], cnt[SEL_MAX], mn[SEL_MAX];
#pragma unroll
  for (int r = 0; r < SEL_MAX; ++r) {
    v[r] = r < nsel ? st[r].prefix : 0ull;
    cnt[r] = 0;
    mn[r] = 0x7FF0000000000000ull;   // +inf
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!m1[i] || (m2 != nullptr && !m2[i])) continue;
    const unsigned long long key = (unsigned long long)__double_as_longlong(err[i]);
#pragma unroll
    for (int r = 0; r < SEL_MAX; ++r) {
      if (key <= v[r]) ++cnt[r];
      else if (key < mn[r]) mn[r] = key;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < SEL_MAX; ++r) {
    for (int off = 32; off > 0; off >>= 1) {
      cnt[r] += __shfl_down(cnt[r], off, 64);
      const unsigned long long o = __shfl_down(mn[r], off, 64);
      mn[r] = o < mn[r] ? o : mn[r];
    }
    if (lane == 0) { red[wave][2 * r] = cnt[r]; red[wave][2 * r + 1] = mn[r]; }
  }
  __syncthreads();
  if (threadIdx.x < SEL_MAX) {
    const int r = threadIdx.x;
    unsigned long long c = 0, m = 0x7FF0000000000000ull;
    for (int w = 0; w < 4; ++w) {
      c += red[w][2 * r];
      m = red[w][2 * r + 1] < m ? red[w][2 * r + 1] : m;
    }
    part[((size_t)blockIdx.x * SEL_MAX + r) * 2] = c;
    part[((size_t)blockIdx.x * SEL_MAX + r) * 2 + 1] = m;
  }
}

__global__ void k_u32_to_f64(const unsigned int* __restrict__ in, double* __restrict__ out, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = (double)in[i];
}
__global__ void k_f64_to_u32(const double* __restrict__ in, unsigned int* __restrict__ out, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = (unsigned int)in[i];
}

// partial[blk*2 + {0,1}] = {sum err^2, count} over the mask
__global__ void k_err_sums(const double* __restrict__ err, const uint8_t* __restrict__ m1, const uint8_t* __restrict__ m2,
                           int n, double* __restrict__ partial) {
  __shared__ double scratch[16];
  double sq = 0.0, cnt = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!m1[i] || (m2 != nullptr && !m2[i])) continue;
    sq += err[i] * err[i];
    cnt += 1.0;
  }
  const double a = block_reduce<false>(sq, scratch);
  const double b = block_reduce<false>(cnt, scratch);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = a; partial[2 * blockIdx.x + 1] = b; }
}
__global__ void k_sum2(const double* __restrict__ partial, int nblk, double* __restrict__ out) {
  __shared__ double scratch[16];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  const double ra = block_reduce<false>(a, scratch);
  const double rb = block_reduce<false>(b, scratch);
  if (threadIdx.x == 0) { out[0] = ra; out[1] = rb; }
}

// inliers = (err < threshold) & valid  (calibration.py:243-244); one block per view also refreshes view_count
__global__ void k_reject(Dims d, const double* __restrict__ err, const uint8_t* __restrict__ evalid, double threshold,
                         uint8_t* __restrict__ inlier, int32_t* __restrict__ view_count) {
  __shared__ double scratch[16];
  const int v = blockIdx.x;
  double cnt = 0.0;
  for (int p = threadIdx.x; p < d.P; p += blockDim.x) {
    const size_t s = (size_t)v * d.P + p;
    const bool in = evalid[s] && err[s] < threshold;
    inlier[s] = in ? 1 : 0;
    cnt += in ? 1.0 : 0.0;
  }
  const double tot = block_reduce<false>(cnt, scratch);
  if (threadIdx.x == 0) view_count[v] = (int32_t)tot;
}

// compact list of the non-empty views: out[0] = count, out[1..] = view indices, LARGEST views first (by the number of
// 64-observation chunks; ascending index within a class, so the list is deterministic).  The persistent kernels hand
// the list out front to back: the long views start first and the short ones fill the tail of the launch (longest-
// processing-time-first list scheduling).  Runs only when the inlier set changes.
// Two launches of ceil(nviews / 1024) workgroups (a single workgroup walking all views class by class took 0.82 ms for
// the 80 000 views of the 16 x 1000 x 5 rig): k_active_count leaves per-workgroup class counts, k_active_scatter turns
// them into the workgroup's write offsets (classes descending, workgroups ascending) and places its views.
constexpr int AV_CLASSES = LIN_MAX_POINTS / 64, AV_THREADS = 1024;
__device__ __forceinline__ int active_class(int cnt) { return cnt == 0 ? 0 : min((cnt + 63) / 64, AV_CLASSES); }
__global__ __launch_bounds__(AV_THREADS) void k_active_count(int nviews, const int32_t* __restrict__ view_count,
                                                             int32_t* __restrict__ counts /*[blocks][AV_CLASSES]*/) {
  __shared__ int tot[AV_CLASSES];
  if (threadIdx.x < AV_CLASSES) tot[threadIdx.x] = 0;
  __syncthreads();
  const int v = blockIdx.x * AV_THREADS + threadIdx.x;
  const int cls = v < nviews ? active_class(view_count[v]) : 0;
  for (int k = 1; k <= AV_CLASSES; ++k) {
    const unsigned long long m = __ballot(cls == k);
    if ((threadIdx.x & 63) == 0 && m != 0) atomicAdd(&tot[k - 1], (int)__popcll(m));   // (integer sums: order-free)
  }
  __syncthreads();
  if (threadIdx.x < AV_CLASSES) counts[blockIdx.x * AV_CLASSES + threadIdx.x] = tot[threadIdx.x];
}
__global__ __launch_bounds__(AV_THREADS) void k_active_scatter(int nviews, const int32_t* __restrict__ view_count,
                                                               const int32_t* __restrict__ counts, int32_t* __restrict__ out) {
  __shared__ int base[AV_CLASSES], wave_tot[AV_THREADS / 64], total;
  const int nblk = gridDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < AV_CLASSES) {   // thread k: offset of class k + 1 for this workgroup
    const int k = threadIdx.x;
    int before = 0, all = 0;
    for (int b = 0; b < nblk; ++b) {
      for (int kk = k + 1; kk < AV_CLASSES; ++kk) before += counts[b * AV_CLASSES + kk];   // larger classes come first
      if (b < (int)blockIdx.x) before += counts[b * AV_CLASSES + k];
      if (k == 0)
        for (int kk = 0; kk < AV_CLASSES; ++kk) all += counts[b * AV_CLASSES + kk];
    }
    base[k] = before;
    if (k == 0) total = all;
  }
  __syncthreads();
  const int v = blockIdx.x * AV_THREADS + threadIdx.x;
  const int cls = v < nviews ? active_class(view_count[v]) : 0;
  for (int k = AV_CLASSES; k >= 1; --k) {
    const bool on = cls == k;
    const unsigned long long m = __ballot(on);
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = base[k - 1];
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (on) out[1 + off + __popcll(m & ((1ull << lane) - 1ull))] = v;
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = total;
}

// ---------------------------------------------------------------------------------------------------------------
// lowering ON THE DEVICE (mcba_create / mcba_set_inliers): the caller's arrays are uploaded as they are -- the shard's
// frames of point_table.points / .valid / inlier_mask in the reference's [C][F][B][P] order, one contiguous slab per
// camera -- and these kernels build the frame-major tables from them.  (The host loops they replace cost 7 ms per
// Calibration at the north-star rig: three passes over 2.6 M slots.)
//   Calibration.valid / inliers                      optimization/calibration.py:69-81
//   mask of tables.reprojection_error                 tables.py:244-249 (reprojected.valid & point_table.valid)
//   residual ordering of `evaluate`                   calibration.py:204-206 (C-order over the inlier mask)
// ---------------------------------------------------------------------------------------------------------------
// one wavefront per view v (frame-major) of the shard; raw slabs are [C][Fl][B][P]
__global__ __launch_bounds__(64) void k_lower_view(Dims d, const double2* __restrict__ pts_raw,
                                                   const float2* __restrict__ pts_raw32 /* the table as float32, or null */,
                                                   const uint8_t* __restrict__ pvalid_raw,
                                                   const uint8_t* __restrict__ mask_raw /* or null: inliers = valid */,
                                                   const uint8_t* __restrict__ cam_valid, const uint8_t* __restrict__ frame_valid,
                                                   const uint8_t* __restrict__ board_valid, const int32_t* __restrict__ board_off,
                                                   double2* __restrict__ obs, uint8_t* __restrict__ valid_fm,
                                                   uint8_t* __restrict__ evalid, uint8_t* __restrict__ inlier,
                                                   int32_t* __restrict__ view_count, int32_t* __restrict__ view_ecount) {
  const int v = blockIdx.x, lane = threadIdx.x;
  const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
  const bool pv = cam_valid[c] && frame_valid[f] && board_valid[b];
  const int nb = board_off[b + 1] - board_off[b];
  const size_t r0 = (((size_t)c * d.Fl + fl) * d.B + b) * d.P, s0 = (size_t)v * d.P;
  int cnt = 0, ecnt = 0;
  for (int q0 = 0; q0 < d.P; q0 += 64) {
    const int q = q0 + lane;
    bool in = false, ev = false;
    if (q < d.P) {
      const bool vv = pv && pvalid_raw[r0 + q] != 0;
      ev = vv && q < nb;
      in = mask_raw != nullptr ? mask_raw[r0 + q] != 0 : vv;
      if (pts_raw != nullptr) obs[s0 + q] = pts_raw[r0 + q];
      else if (pts_raw32 != nullptr) {
        const float2 w = pts_raw32[r0 + q];
        obs[s0 + q] = make_double2((double)w.x, (double)w.y);
      }
      if (valid_fm != nullptr) { valid_fm[s0 + q] = vv ? 1 : 0; evalid[s0 + q] = ev ? 1 : 0; }
      inlier[s0 + q] = in ? 1 : 0;
    }
    cnt += __popcll(__ballot(in));
    ecnt += __popcll(__ballot(ev));
  }
  if (lane == 0) {
    view_count[v] = cnt;
    if (view_ecount != nullptr) view_ecount[v] = ecnt;
  }
}

// inliers = valid (mcba_set_inliers(NULL)): from the frame-major validity table kept on the device
__global__ __launch_bounds__(64) void k_inliers_from_valid(Dims d, const uint8_t* __restrict__ valid_fm,
                                                           uint8_t* __restrict__ inlier, int32_t* __restrict__ view_count) {
  const int v = blockIdx.x, lane = threadIdx.x;
  const size_t s0 = (size_t)v * d.P;
  int cnt = 0;
  for (int q0 = 0; q0 < d.P; q0 += 64) {
    const int q = q0 + lane;
    const bool in = q < d.P && valid_fm[s0 + q] != 0;
    if (q < d.P) inlier[s0 + q] = in ? 1 : 0;
    cnt += __popcll(__ballot(in));
  }
  if (lane == 0) view_count[v] = cnt;
}

// exclusive prefix of the per-view inlier counts in the REFERENCE order of the views (camera, frame, board) -- the order
// of the residual vector -- written back per frame-major view: first[v] = index of the view's first residual pair;
// totals[0] = number of inliers, totals[1] = sum of the second count array (may be null).  One workgroup.
__global__ __launch_bounds__(1024) void k_view_scan(Dims d, const int32_t* __restrict__ view_count,
                                                    const int32_t* __restrict__ view_ecount, int32_t* __restrict__ first,
                                                    long long* __restrict__ totals) {
  __shared__ long long wsum[16], wsum_e[16];
  const int nv = d.views(), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (nv + 1023) / 1024, r0 = tid * per, r1 = min(nv, r0 + per);
  auto fm_of = [&](int r) {   // reference-order view index r = (c Fl + fl) B + b  ->  frame-major index
    const int b = r % d.B, fl = (r / d.B) % d.Fl, c = r / (d.B * d.Fl);
    return (fl * d.C + c) * d.B + b;
  };
  long long mine = 0, mine_e = 0;
  for (int r = r0; r < r1; ++r) {
    const int v = fm_of(r);
    mine += view_count[v];
    if (view_ecount != nullptr) mine_e += view_ecount[v];
  }
  long long incl = mine, incl_e = mine_e;
  for (int off = 1; off < 64; off <<= 1) {
    const long long o = __shfl_up(incl, off, 64), oe = __shfl_up(incl_e, off, 64);
    if (lane >= off) { incl += o; incl_e += oe; }
  }
  if (lane == 63) { wsum[wave] = incl; wsum_e[wave] = incl_e; }
  __syncthreads();
  long long base = 0, tot = 0, tot_e = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
    tot_e += wsum_e[w];
  }
  long long run = base + incl - mine;
  for (int r = r0; r < r1; ++r) {
    const int v = fm_of(r);
    first[v] = (int32_t)run;
    run += view_count[v];
  }
  if (tid == 0) { totals[0] = tot; totals[1] = tot_e; }
}

// residual index of every slot: obs_index[s] = first[v] + (number of inliers before p in the view), -1 if not an inlier
__global__ __launch_bounds__(64) void k_obs_index(Dims d, const uint8_t* __restrict__ inlier, const int32_t* __restrict__ first,
                                                  int32_t* __restrict__ obs_index) {
  const int v = blockIdx.x, lane = threadIdx.x;
  const size_t s0 = (size_t)v * d.P;
  int base = first[v];
  for (int q0 = 0; q0 < d.P; q0 += 64) {
    const int q = q0 + lane;
    const bool in = q < d.P && inlier[s0 + q] != 0;
    const unsigned long long m = __ballot(in);
    if (q < d.P) obs_index[s0 + q] = in ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
    base += __popcll(m);
  }
}

// the compacted observation tables of the lsmr route (LsmrCompact): one wavefront per ACTIVE view, in the order of the active list
__global__ __launch_bounds__(64) void k_compact_views(Dims d, Tables t, const int32_t* __restrict__ first, double2* __restrict__ obsC,
                                                      double2* __restrict__ bxy, double* __restrict__ bz, double* __restrict__ trC,
                                                      int4* __restrict__ desc) {
  const int vi = blockIdx.x, lane = threadIdx.x;
  if (vi >= t.active_views[0]) return;
  const int v = t.active_views[1 + vi];
  if (v < 0) { if (lane == 0) desc[vi] = make_int4(-1, 0, 0, 0); return; }
  const int b = v % d.B, c = (v / d.B) % d.C;
  const size_t s0 = (size_t)v * d.P;
  const int f0 = first[v];
  const double height = trC != nullptr ? t.img_h[c] : 1.0;   // (camera_entry: CAM_HEIGHT = image_height)
  int base = f0;
  for (int q0 = 0; q0 < d.P; q0 += 64) {
    const int q = q0 + lane;
    const bool in = q < d.P && t.inlier[s0 + q] != 0;
    const unsigned long long m = __ballot(in);
    if (in) {
      const int gi = base + __popcll(m & ((1ull << lane) - 1ull));
      const double2 ob = t.obs[s0 + q];
      obsC[gi] = ob;
      if (trC != nullptr) trC[gi] = ob.y / height;   // (the division slot_forward performs: same operands, same bits)
      const double* X = t.board_points + 3 * (size_t)(b * d.P + q);
      bxy[gi] = make_double2(X[0], X[1]);
      bz[gi] = X[2];
    }
    base += __popcll(m);
  }
  if (lane == 0) desc[vi] = make_int4(v, f0, base - f0, 0);
}

// frame-major inlier table -> reference [C,F,B,P] order (only this shard's frames are written)
__global__ void k_inliers_to_ref(Dims d, const uint8_t* __restrict__ inlier, uint8_t* __restrict__ out) {
  const int n = d.slots();
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const int p = s % d.P, v = s / d.P;
    const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
    out[(((size_t)c * d.F + f) * d.B + b) * d.P + p] = inlier[s];
  }
}

// MFMA layout self-test: D = A^T-style product with an asymmetric operand pair, checked on the host
__global__ void k_mfma_probe(const double* __restrict__ V /*[4][32]*/, double* __restrict__ out /*[16][16]*/) {
  const int lane = threadIdx.x, rsub = lane >> 4, csub = lane & 15;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const double a = V[rsub * 32 + csub], b = V[rsub * 32 + 16 + csub];
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(rsub + 4 * r) * 16 + csub] = acc[r];
}


// Pipe probe: does a wave's FP64 VALU work overlap with another wave's FP64 MFMAs on the same SIMD?
// One workgroup of 512 threads per CU = two waves per SIMD.  mode 0: every wave runs `iters` rounds of 16 independent FP64
// FMAs (64 instructions of 4 issue cycles each = 256 pipe cycles per round... x 4 unrolled); mode 1: every wave runs `iters`
// rounds of 4 independent MFMA f64 16x16x4 (4 x 64 cycles); mode 2: waves 0-3 (one per SIMD) do the FMA loop, waves 4-7 the
// MFMA loop.  If the two kinds of work shared nothing, mode 2 would take as long as one wave alone per SIMD (half of
// modes 0 / 1); if FP64 MFMA and FP64 VALU share the pipe, mode 2 takes the sum.
// workgroup dispatch probe: every workgroup records the 100 MHz wall clock when it starts and after `spin` dependent FMAs
// debug: does data written by workgroup w of one kernel stay in the L2 of w's XCD for the next kernel?  k_xcd_write: workgroup
// b fills region b (n doubles); k_xcd_read: ONE workgroup sums region `region` with 16 loads in flight per thread and reports
// its shader-clock cycles.  Workgroups are handed to the XCDs round-robin, so workgroup 0 of both kernels shares an XCD.
__global__ void k_xcd_write(double* __restrict__ buf, int n) {
  double* p = buf + (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = (double)(i + blockIdx.x);
}
__global__ __launch_bounds__(512) void k_xcd_read(const double* __restrict__ buf, int n, int region, double* __restrict__ sink,
                                                  long long* __restrict__ cycles) {
  const double* p = buf + (size_t)region * n;
  const long long t0 = clock64();
  double acc = 0.0;
  for (int i0 = threadIdx.x; i0 < n; i0 += 16 * 512) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[min(i0 + u * 512, n - 1)];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  sink[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) cycles[0] = clock64() - t0;
}

__global__ void k_dispatch_probe(int spin, long long* __restrict__ out) {
  extern __shared__ double probe_lds[];
  const long long t0 = wall_clock64();
  double a = (double)threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;
  if (a == 12345.678) probe_lds[threadIdx.x] = a;   // (keeps the loop and the LDS allocation alive)
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = t0;
    out[2 * blockIdx.x + 1] = wall_clock64();
  }
}

__global__ __launch_bounds__(512) void k_pipe_probe(int mode, int iters, double* __restrict__ sink) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 1 || (mode == 2 && wave >= 4);
  double acc = 0.0;
  if (do_mfma) {
    double4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    acc = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    double f[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) f[k] = 1.0 + k * 1e-3 + threadIdx.x * 1e-9;
    const double m = 1.0 - 1e-12, c = 1e-13;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int k = 0; k < 16; ++k) f[k] = f[k] * m + c;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += f[k];
  }
  if (acc == 12345.678) sink[0] = acc;   // keep the work alive
}

}  // namespace mcba
