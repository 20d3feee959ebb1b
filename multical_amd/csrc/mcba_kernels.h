// mcba_kernels.h -- HIP kernels of the bundle-adjustment hot path, written for gfx950 (CDNA4, wave64).
//
// Evaluation pipeline at a parameter vector x (all FP64):
//   k_prep        x -> pose table (R, t, left Jacobian per pose), camera table, board points (the non-linearising entry
//                 points; a linearisation gets these from k_tmat, a trial step from the tail of k_vec_step)
//   k_views       pose table -> one board->camera chain matrix per view (camera, frame, board)
//   k_tmat        x -> That (pose-block structure) and chain matrix of every non-empty view + the tables; opens a linearisation
//   k_residual    a wavefront per non-empty view over its compacted inliers: residuals (evaluate()); a loop over the table
//                 slots for projections / reprojection errors
//   k_cost        persistent wavefronts over the non-empty views (largest first): robust cost of a trial step
//   k_jacobian    analytic Jacobian rows in the reference's sparsity pattern                        (parity / scipy-driven mode)
//   k_linearize   persistent wavefronts, ONE WAVEFRONT PER VIEW at a time: per-point row pairs V = [E | K | r] are staged
//                 through LDS and accumulated into S = V^T V with v_mfma_f64_16x16x4_f64 (the MFMA does the cross-lane
//                 reduction); the view's local normal equations M = That^T S That are written as one record.
//   k_assemble, k_shared_final   deterministic reductions of the records into H_ss (dense, shared parameters),
//                 H_fs / H_ff (per-frame blocks), g and diag(H)                                    (mcba_solver_kernels.h)
//   k_schur_* / k_chol_* / k_vec_* / k_tr_*   the damped normal-equation solve and the scalar algebra of
//                 the trust-region driver                                                           (mcba_solver_kernels.h)
//
// There is no reference counterpart for the normal-equation kernels (the reference hands a finite-difference
// sparse Jacobian to scipy's LSMR, optimization/calibration.py:209-210); their specification is J^T J, J^T f of the
// residual function `evaluate` (calibration.py:204-206) and is tested as such.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "mcba_view.h"
#include "mcba_lsmr.h"

namespace mcba {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// small device utilities
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_fence() {
  // LDS operations of one wavefront complete in issue order; the fence only stops the compiler from moving
  // accesses across it and drains lgkmcnt (single-wave workgroups: no s_barrier needed).
#if defined(MCBA_EXP_LIGHT_FENCE)
  // experiment: scheduling barrier only.  In-order LDS + the compiler's own may-alias ordering of LDS accesses make the
  // drain (s_waitcnt lgkmcnt(0)) unnecessary for a single wavefront; data dependences get their own waits.
  __builtin_amdgcn_wave_barrier();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}

// p[i] where ok, else 0 -- as an UNCONDITIONAL load from a clamped address whose value always enters the arithmetic.
// Written as `ok ? p[i] : 0`, hipcc branches around every such load and waits for it (s_waitcnt vmcnt(0)) before the
// join, which turns a batch of independent loads into one serial memory round trip each (found in the ISA of the
// k_linearize prologue and of the k_chol_blk load loop: 8 and 12 dependent round trips).  No fast-math: the multiply by
// 0 / 1 cannot be folded back into a select.  (An empty asm on the mask is worse: hipcc drains vmcnt before inline asm.)
__device__ __forceinline__ double masked_load(const double* __restrict__ p, size_t i, bool ok) {
  return p[ok ? i : 0] * (ok ? 1.0 : 0.0);
}
// row[i] for i < n, else 0; row is wave-uniform and the lane offset stays a 32-bit value (no 64-bit per-lane index that
// the register allocator would hoist out of the view loop and spill)
__device__ __forceinline__ double masked_load_row(const double* __restrict__ row, int i, int n) {
  return row[min(i, n - 1)] * (i < n ? 1.0 : 0.0);
}
__device__ __forceinline__ uint8_t masked_load_row(const uint8_t* __restrict__ row, int i, int n) {
  return (uint8_t)(row[min(i, n - 1)] & (i < n ? 0xFFu : 0u));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;   // valid in lane 0
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// A wave-uniform double as a SCALAR value: every lane holds the same number (it was loaded from one address); reading
// lane 0 back through v_readfirstlane puts it into an SGPR pair, so that it costs no vector register and serves as the
// scalar operand of the FP64 instructions that use it.  (hipcc does not turn the uniform table reads of k_linearize into
// s_load by itself: the LDS fences of the view loop count as clobbers of global memory in its analysis.)
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Pairwise in-register folding of per-lane partial sums with the gfx950 lane-swap instructions (VALU, no LDS traffic):
//   fold32(a, b): lanes  0..31 return a[l] + a[l + 32],  lanes 32..63 return b[l - 32] + b[l]
//                 (v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second)
//   fold16(a, b): with rows of 16 lanes a = [a0 a1 a2 a3], b = [b0 b1 b2 b3]:  [a0 + a1, b0 + b1, a2 + a3, b2 + b3]
//                 (v_permlane16_swap: odd rows of the first operand <-> even rows of the second)
__device__ __forceinline__ double fold32(double a, double b) {
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double fold16(double a, double b) {
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// deterministic block reduction (blockDim.x multiple of 64, <= 1024); result valid in thread 0
template <bool MAX>
__device__ __forceinline__ double block_reduce(double v, double* scratch /*[16]*/) {
  v = MAX ? wave_max(v) : wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    r = scratch[0];
    for (int i = 1; i < nw; ++i) r = MAX ? fmax(r, scratch[i]) : r + scratch[i];
  }
  return r;
}

// N per-lane partial sums -> N wave totals in ~N/2 + N/4 + N/8 .. lane exchanges instead of 6 N (wave_sum per value): a butterfly
// that halves the number of live registers while it halves the lanes left to add.  fold32 / fold16 pair the values up over
// the lane halves / the 16-lane rows, the remaining four steps (lane distance 8, 4, 2, 1) pair the registers with a select +
// __shfl_xor until one is left.  Lane L then holds the total of value
//     many_index(L) = b5 + 2 b4 + 4 b3 + 8 b2 + 16 b1        (b_k = bit k of L)
// provided many_writer<N>(L): the lane bits that took part in no register pairing are zero (those lanes hold copies).
// (index map verified by symbolic simulation of exactly these steps: profiles/scripts/sim_wave_reduce_many.py)
template <int NREG, int S>
struct ManyStep {
  static __device__ __forceinline__ void run(double* q, int lane) {
    if constexpr (S >= 1) {
      if constexpr (NREG == 1) {
        q[0] += __shfl_xor(q[0], S, 64);
        ManyStep<1, S / 2>::run(q, lane);
      } else {
        constexpr int H = (NREG + 1) / 2;
        const bool up = (lane & S) != 0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
          const double a = q[2 * i], b = (2 * i + 1 < NREG) ? q[2 * i + 1] : 0.0;
          const double keep = up ? b : a, send = up ? a : b;
          q[i] = keep + __shfl_xor(send, S, 64);
        }
        ManyStep<H, S / 2>::run(q, lane);
      }
    }
  }
};
template <int N>
__device__ __forceinline__ double wave_reduce_many(const double (&v)[N], int lane) {
  static_assert(N >= 1 && N <= 32, "at most 32 values");
  constexpr int NP = (N + 3) / 4 * 4;
  double r[NP / 2], q[NP / 4];
#pragma unroll
  for (int i = 0; i < NP / 2; ++i) r[i] = fold32(2 * i < N ? v[2 * i] : 0.0, 2 * i + 1 < N ? v[2 * i + 1] : 0.0);
#pragma unroll
  for (int i = 0; i < NP / 4; ++i) q[i] = fold16(r[2 * i], r[2 * i + 1]);
  ManyStep<NP / 4, 8>::run(q, lane);
  return q[0];
}
__device__ __forceinline__ int many_index(int lane) {
  return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4);
}
template <int N>
__device__ __forceinline__ bool many_writer(int lane) {
  constexpr int R = ((N + 3) / 4 * 4) / 4;   // registers after fold16; pairing steps: distance 8 if R > 1, 4 if R > 2, 2 if R > 4
  constexpr int unused = 1 | (R > 4 ? 0 : 2) | (R > 2 ? 0 : 4) | (R > 1 ? 0 : 8);
  return (lane & unused) == 0 && many_index(lane) < N;
}

// sum of a[0 .. n) over the lanes of ONE wavefront (lane L adds L, L + 64, ..), B loads in flight per lane: a plain
// "for (i = lane; i < n; i += 64) s += a[i]" waits for every load before it issues the next -- 32 dependent round trips of
// ~0.7 us for 2048 partials, which is what made the heads of the first fused LSMR kernels cost 6 - 10 us (round 5 trace).
// n >= 1.  Result in every lane.
template <int B>
__device__ __forceinline__ double wave_fold_batched(const double* __restrict__ a, int n, int lane) {
  double s = 0.0;
  for (int i0 = lane; i0 < n; i0 += 64 * B) {
    double v[B];
#pragma unroll
    for (int k = 0; k < B; ++k) v[k] = a[min(i0 + 64 * k, n - 1)];   // (clamped index: unconditional, independent loads)
#pragma unroll
    for (int k = 0; k < B; ++k)
      if (i0 + 64 * k < n) s += v[k];
  }
  s = wave_sum(s);
  return __shfl(s, 0, 64);
}

// (Fusing the "sum the per-block partials" launch into the producing kernel with the threadfence + ticket-counter idiom
//  was measured and rejected on this chip: an agent-scope release is an L2 write-back on a multi-XCD part and ~75 ns per
//  same-address atomic serialises 500-1000 tickets into 40-90 us.  Single-GPU solves copy the partials to the host with
//  the scalars they already fetch instead; sharded solves keep the small final-sum kernels.)

// Points of a view that the compaction lists hold at a time.  Boards with more points (the reference has no cap:
// tables.stack_boards, tables.py:385-394 -- a 25 x 35 charuco has 816 corners) are walked in SEGMENTS of this many table
// slots: masks of a segment are compacted, its dense chunks are processed, and the accumulators / the residual run of the
// view carry on into the next segment.  Point indices are kept as uint16 (mcba_create checks n_points <= 65535).
constexpr int LIN_MAX_POINTS = 512;

// ---------------------------------------------------------------------------------------------------------------
// k_residual: evaluate() of optimization/calibration.py:204-206 (+ projections and per-slot errors of
//             tables.reprojection_error, tables.py:244-249).  ONE WAVEFRONT PER VIEW (four views per 256-thread block):
//             camera, board and chain matrices are wave-uniform (scalar loads), the frame-major 16-byte observations and
//             the mask bytes stream in coalesced, and the residuals of a view leave as ONE contiguous run: the view's
//             first residual index comes from k_view_scan (prefix of the inlier counts in the reference's view order) and
//             the position inside the view from a ballot prefix -- no per-slot index table is read (the round-1 kernel
//             gathered a 4-byte obs_index per slot and scattered 8-byte stores: 36 % of the HBM peak).
//             Algorithmic traffic: 17 B per slot read + 16 B per observation written.
// ---------------------------------------------------------------------------------------------------------------
template <int ND, int FISH, bool ROLL>
__global__ __launch_bounds__(256) void k_residual(Dims d, Tables t, const int32_t* __restrict__ first,
                                                  double* __restrict__ r, double* __restrict__ proj,
                                                  double* __restrict__ err, uint8_t* __restrict__ valid) {
  const int lane = threadIdx.x & 63;
  const int nv = d.views();
  const bool all_slots = proj != nullptr || err != nullptr;
  if (!all_slots) {
    // residuals only (evaluate(), the hot entry): like k_cost, the inlier bytes of a view are ballot-compacted into a point
    // list first and the projections run on dense 64-lane chunks -- a loop over the table slots evaluated the forward
    // model for every 64-slot group although 29 % of the slots of the north-star rig hold an inlier (VALU-bound on idle
    // lanes).  Residual i of the view's list goes to first[v] + i: the list order is the slot order.
    __shared__ uint16_t plist[4][LIN_MAX_POINTS];
    uint16_t* pidx = plist[threadIdx.x >> 6];
    const int n_active = t.active_views[0];
    constexpr int NPB64 = LIN_MAX_POINTS / 64;
    for (int vi = blockIdx.x * 4 + (threadIdx.x >> 6); vi < n_active; vi += gridDim.x * 4) {
      const int v = __builtin_amdgcn_readfirstlane(t.active_views[1 + vi]);
      const int b = v % d.B, c = (v / d.B) % d.C;
      size_t out0 = (size_t)first[v];
      for (int seg0 = 0; seg0 < d.P; seg0 += LIN_MAX_POINTS) {   // (one segment unless a board has > LIN_MAX_POINTS points)
      uint8_t inb[NPB64];
#pragma unroll
      for (int k = 0; k < NPB64; ++k) inb[k] = masked_load_row(t.inlier + (size_t)v * d.P, seg0 + k * 64 + lane, d.P);
      int count = 0;
#pragma unroll
      for (int k = 0; k < NPB64; ++k) {
        const bool in = inb[k] != 0;
        const unsigned long long m = __ballot(in);
        if (in) pidx[count + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(seg0 + k * 64 + lane);
        count += __popcll(m);
      }
      lds_fence();
      int p_cur = lane < count ? pidx[lane] : 0;
      double2 ob_cur = t.obs[(size_t)v * d.P + p_cur];
      double X_cur[3], X_nxt[3];
      for (int k = 0; k < 3; ++k) X_cur[k] = t.board_points[3 * (size_t)(b * d.P + p_cur) + k];
      for (int base = 0; base < count; base += 64) {
        const int i = base + lane, inx = i + 64;
        const int p_nxt = inx < count ? pidx[inx] : p_cur;
        const double2 ob_nxt = t.obs[(size_t)v * d.P + p_nxt];
        for (int k = 0; k < 3; ++k) X_nxt[k] = t.board_points[3 * (size_t)(b * d.P + p_nxt) + k];
        if (i < count) {
          double uv[2], Xs[3], Xe[3], tr;
          slot_forward<ND, FISH, ROLL, false>(d, t, v, c, b, p_cur, ob_cur, uv, nullptr, nullptr, Xs, Xe, tr, X_cur);
          double2 e2;
          e2.x = uv[0] - ob_cur.x;
          e2.y = uv[1] - ob_cur.y;
          reinterpret_cast<double2*>(r)[out0 + i] = e2;
        }
        p_cur = p_nxt;
        ob_cur = ob_nxt;
        for (int k = 0; k < 3; ++k) X_cur[k] = X_nxt[k];
      }
      out0 += (size_t)count;
      lds_fence();   // the next segment / view rewrites the list
      }
    }
    return;
  }
  for (int vw = blockIdx.x * 4 + (threadIdx.x >> 6); vw < nv; vw += gridDim.x * 4) {
    const int v = __builtin_amdgcn_readfirstlane(vw);             // wave-uniform: everything derived from it is scalar
    if (!all_slots && t.view_count[v] == 0) continue;
    const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
    int base = (r != nullptr && first != nullptr) ? first[v] : 0;
    for (int q0 = 0; q0 < d.P; q0 += 64) {
      const int p = q0 + lane;
      const bool ok = p < d.P;
      const size_t s = (size_t)v * d.P + (ok ? p : 0);
      const bool in = ok && t.inlier[s] != 0;
      const unsigned long long m = __ballot(in);
      if (ok && (in || all_slots)) {
        const double2 ob = t.obs[s];
        double uv[2], Xs[3], Xe[3], tr;
        slot_forward<ND, FISH, ROLL, false>(d, t, v, c, b, p, ob, uv, nullptr, nullptr, Xs, Xe, tr);
        const double ex = uv[0] - ob.x, ey = uv[1] - ob.y;
        if (r != nullptr && in) {
          const size_t idx = (size_t)base + __popcll(m & ((1ull << lane) - 1ull));
          double2 e2;
          e2.x = ex;
          e2.y = ey;
          reinterpret_cast<double2*>(r)[idx] = e2;
        }
        const size_t ri = (((size_t)c * d.F + f) * d.B + b) * d.P + p;
        if (proj != nullptr) {
          proj[2 * ri] = uv[0];
          proj[2 * ri + 1] = uv[1];
        }
        if (err != nullptr) {
          const bool ev = t.evalid[s] != 0;
          const double e = ev ? sqrt(ex * ex + ey * ey) : 0.0;
          if (valid != nullptr) {          // reference [C,F,B,P] order (host-facing)
            err[ri] = e;
            valid[ri] = ev ? 1 : 0;
          } else {                         // frame-major, device-internal (outlier loop)
            err[s] = e;
          }
        }
      }
      base += __popcll(m);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_project_model: Calibration.projected (optimization/calibration.py:113-119): the projection of every table slot WITHOUT
// the measured points.  Rolling shutter (motion/rolling_frames.py:115-133): scan time 0.5 in the first pass, then
// `iterations` fixed-point passes with the scan time taken from the projected row of the previous pass; the other motion
// models project once.  Output in the reference's [C,F,B,P,2] order (host-facing: GUI / reprojection tables).
// ---------------------------------------------------------------------------------------------------------------
template <int ND, int FISH, bool ROLL>
__global__ void k_project_model(Dims d, Tables t, int iterations, double* __restrict__ proj) {
  const int n = d.slots();
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const int p = s % d.P, v = s / d.P;
    const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
    const double height = t.cam[(size_t)c * CAM_STRIDE + CAM_HEIGHT];
    double uv[2], Xs[3], Xe[3], tr;
    double2 ob;
    ob.x = 0.0;
    ob.y = 0.5 * height;             // scan time 0.5 exactly ((0.5 h) / h)
    slot_forward<ND, FISH, ROLL, false>(d, t, v, c, b, p, ob, uv, nullptr, nullptr, Xs, Xe, tr);
    if constexpr (ROLL) {
      for (int it = 0; it < iterations; ++it) {
        ob.y = uv[1];                // rolling_times of the projected points: t = y / image height
        slot_forward<ND, FISH, ROLL, false>(d, t, v, c, b, p, ob, uv, nullptr, nullptr, Xs, Xe, tr);
      }
    }
    const size_t ri = (((size_t)c * d.F + f) * d.B + b) * d.P + p;
    proj[2 * ri] = uv[0];
    proj[2 * ri + 1] = uv[1];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_cost: 0.5 * sum rho(f^2) over the inliers of the shard; partial sums per workgroup (fixed grid -> deterministic).
// Persistent single-wave workgroups walk the compact list of non-empty views like k_linearize does: the inlier bytes of
// a view are ballot-compacted into a point list first, so the projections run on dense 64-lane chunks (about a quarter
// of the table slots of a real rig hold an inlier; a thread-per-slot loop idles three lanes out of four).
// ---------------------------------------------------------------------------------------------------------------
// entry q of the rigid product (R1 | t1) . (R2 | t2) = (R1 R2 | R1 t2 + t1); A, B: R[9] t[3] (the expressions of se3_mul)
__device__ __forceinline__ double se3_mul_entry(const double* A, const double* B, int q) {
  if (q < 9) {
    const int i = q / 3, j = q % 3;
    return A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  }
  const int i = q - 9;
  return (A[3 * i] * B[9] + A[3 * i + 1] * B[10] + A[3 * i + 2] * B[11]) + A[9 + i];
}

// Chain matrices board -> camera of view (f, c, b) straight from the pose table, by the lanes of ONE wavefront: lane
// (chain, entry) forms one entry of each product, the intermediate goes through LDS (tmp [2][12], out [NCH][VIEW_STRIDE]).
// The same matrices as view_chain; formed by every lane in registers they kept 36 doubles live (k_cost: 191 VGPRs).
// Column j of That from the chain prefixes, for the lane-per-column prologue of k_linearize: the same construction as
// that_column_from_prefix (mcba_view.h), but what depends on the lane's pose block -- prefix rotation (identity for the camera,
// R1 for the board, the camera rotation for a frame pose), origin, left Jacobian or unit vector -- is chosen by selecting the
// LDS ADDRESS once (a handful of 32-bit selects) instead of selecting every loaded double (~120 v_cndmask per view, a fifth
// of the prologue's VALU instructions).  eye = a 3 x 3 identity in LDS.
template <bool ROLL>
__device__ __forceinline__ void that_column_sel(const double* Pc, const double* Pm0, const double* Pm1, const double* Pb,
                                                const double* pre /*[NCH][PRE_STRIDE]*/, const double* eye, int j, double* Tm,
                                                int stride) {
  constexpr int NCH = ROLL ? 2 : 1, NPB = ROLL ? 4 : 3;
  const int k = j / 6, jj = j - 6 * k;
  const bool rotcol = jj < 3;
  const int ju = rotcol ? jj : jj - 3;
  const bool is_cam = k == 0, is_board = k == NPB - 1;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const double* Pf = ch == 0 ? Pm0 : Pm1;
    const double* R1 = pre + ch * PRE_STRIDE;
    const bool active = is_cam || is_board || !ROLL || k == 1 + ch;
    const double* Lp = !rotcol ? eye : (is_cam ? Pc + POSE_L : (is_board ? Pb + POSE_L : Pf + POSE_L));
    const double* Rp = is_cam ? eye : (is_board ? R1 : Pc + POSE_R);
    const double* op = is_cam ? Pc + POSE_T : (is_board ? R1 + 21 : R1 + 9);
    const double u0 = Lp[ju], u1 = Lp[3 + ju], u2 = Lp[6 + ju];   // column ju of L (rotation columns) or a unit vector
    double tv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) tv[i] = Rp[3 * i] * u0 + Rp[3 * i + 1] * u1 + Rp[3 * i + 2] * u2;   // R_pre u
    const double o0 = op[0], o1 = op[1], o2 = op[2];
    const double c0 = o1 * tv[2] - o2 * tv[1], c1 = o2 * tv[0] - o0 * tv[2], c2 = o0 * tv[1] - o1 * tv[0];
    const double w = active ? 1.0 : 0.0, wr = rotcol ? w : 0.0, wt = rotcol ? 0.0 : w;
    double* col = Tm + (size_t)(6 * ch) * stride + j;
    col[0 * stride] = wr * tv[0];
    col[1 * stride] = wr * tv[1];
    col[2 * stride] = wr * tv[2];
    col[3 * stride] = wr * c0 + wt * tv[0];
    col[4 * stride] = wr * c1 + wt * tv[1];
    col[5 * stride] = wr * c2 + wt * tv[2];
  }
}

template <bool ROLL>
__device__ __forceinline__ void view_chain_wave(const Dims& d, const Tables& t, int f, int c, int b, int lane,
                                                double* tmp, double* out) {
  const double* Pc = t.pose + (size_t)(d.pose_cam + c) * POSE_STRIDE;
  const double* Pb = t.pose + (size_t)(d.pose_board + b) * POSE_STRIDE;
  const double* Pm = t.pose + (size_t)d.pose_motion * POSE_STRIDE;
  constexpr int NCH = ROLL ? 2 : 1;
  const int ch = lane / 12, q = lane % 12;
  const bool on = lane < 12 * NCH;
  if (d.motion == MOTION_HAND_EYE) {   // camera . G . B_f . Wb . board  (mot[0] = world_wrt_base, mot[1] = gripper_wrt_camera)
    if (on) tmp[q] = se3_mul_entry(Pc, Pm + POSE_STRIDE, q);
    lds_fence();
    if (on) tmp[12 + q] = se3_mul_entry(tmp, t.bwg + 12 * (size_t)f, q);
    lds_fence();
    if (on) tmp[q] = se3_mul_entry(tmp + 12, Pm, q);
    lds_fence();
    if (on) out[q] = se3_mul_entry(tmp, Pb, q);
  } else {
    const double* Pf = Pm + (size_t)((on ? ch : 0) * d.F + f) * POSE_STRIDE;
    if (on) tmp[12 * ch + q] = se3_mul_entry(Pc, Pf, q);
    lds_fence();
    if (on) out[VIEW_STRIDE * ch + q] = se3_mul_entry(tmp + 12 * ch, Pb, q);
  }
  lds_fence();
}

// ROBUST = false: the linear loss compiled in (the loss switch pulls log1p / atan into the kernel: 182 VGPRs = 2 waves per
// SIMD, i.e. two rounds of workgroups at the north-star rig)
template <int ND, int FISH, bool ROLL, bool ROBUST>
__global__ __launch_bounds__(64) void k_cost(Dims d, Tables t, double* __restrict__ partial) {
  __shared__ uint16_t pidx[LIN_MAX_POINTS];
  __shared__ double Vc[2 * VIEW_STRIDE], Vtmp[24];   // chain matrices of the view, intermediate products
  const int lane = threadIdx.x;
  const int n_active = t.active_views[0];
  double acc = 0.0;
  for (int vi = blockIdx.x; vi < n_active; vi += gridDim.x) {
    const int v = t.active_views[1 + vi];
    const int b = v % d.B, c = (v / d.B) % d.C, f = d.f0 + v / (d.B * d.C);
    // the chain matrices of the view straight from the pose table (wave-uniform work, done by every lane): a trial step
    // then needs k_prep only, not the per-view table pass k_views
    view_chain_wave<ROLL>(d, t, f, c, b, lane, Vtmp, Vc);
    constexpr int NPB64 = LIN_MAX_POINTS / 64;
    for (int seg0 = 0; seg0 < d.P; seg0 += LIN_MAX_POINTS) {   // (one segment unless a board has > LIN_MAX_POINTS points)
    uint8_t inb[NPB64];
#pragma unroll
    for (int k = 0; k < NPB64; ++k) {
      const int p = seg0 + k * 64 + lane;
      inb[k] = masked_load_row(t.inlier + (size_t)v * d.P, p, d.P);
    }
    int count = 0;
#pragma unroll
    for (int k = 0; k < NPB64; ++k) {
      const bool in = inb[k] != 0;
      const unsigned long long m = __ballot(in);
      if (in) pidx[count + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(seg0 + k * 64 + lane);
      count += __popcll(m);
    }
    lds_fence();
    // observation + board point of the NEXT chunk are requested before the current one is evaluated (as in k_linearize)
    int p_cur = lane < count ? pidx[lane] : 0;
    double2 ob_cur = t.obs[(size_t)v * d.P + p_cur];
    double X_cur[3], X_nxt[3];
    for (int k = 0; k < 3; ++k) X_cur[k] = t.board_points[3 * (size_t)(b * d.P + p_cur) + k];
    for (int base = 0; base < count; base += 64) {
      const int i = base + lane, inx = i + 64;
      const int p_nxt = inx < count ? pidx[inx] : p_cur;
      const double2 ob_nxt = t.obs[(size_t)v * d.P + p_nxt];
      for (int k = 0; k < 3; ++k) X_nxt[k] = t.board_points[3 * (size_t)(b * d.P + p_nxt) + k];
      if (i < count) {
        double uv[2], Xs[3], Xe[3], tr;
        slot_forward<ND, FISH, ROLL, false>(d, t, v, c, b, p_cur, ob_cur, uv, nullptr, nullptr, Xs, Xe, tr, X_cur, Vc);
        double rs, fs;
        const int loss = ROBUST ? d.loss : 0;
        acc += robust_loss(loss, d.f_scale, uv[0] - ob_cur.x, &rs, &fs);
        acc += robust_loss(loss, d.f_scale, uv[1] - ob_cur.y, &rs, &fs);
      }
      p_cur = p_nxt;
      ob_cur = ob_nxt;
      for (int k = 0; k < 3; ++k) X_cur[k] = X_nxt[k];
    }
    lds_fence();   // the next segment / view rewrites the list
    }
  }
  const double tot = wave_sum(acc);
  if (lane == 0) partial[blockIdx.x] = 0.5 * tot;
}

// ---------------------------------------------------------------------------------------------------------------
// k_lsmr_jv / k_lsmr_jtu: the two products of the Golub-Kahan bidiagonalisation with the column-scaled Jacobian
// J_h = J diag(dscale), matrix-free (solver = "lsmr": scipy's TRF with its LSMR trust-region solver, trf.py:474-489, run on the
// device so that the reference's own steps -- not exact normal-equation steps -- are taken).  Both walk the compact list of
// non-empty views with persistent single-wave workgroups like k_cost: the inlier bytes of a view are ballot-compacted, every
// lane re-evaluates the forward model and the analytic row pair V = [E | K | r] of its observation (point_state / point_row:
// the rows k_linearize accumulates), and the pose part goes through the view's That (k_tmat's table):
//   J_h v  (row a of observation p)  =  E_a . (That vp) + K_a . vK,     vp / vK = the view's scaled local parameters
//   J_h^T u (per view)               =  That^T (sum_p E_p^T u_p)  |  sum_p K_p^T u_p,   reduced over the views by k_lsmr_gather
// Vectors of length m live in the residual order (first[v] + position in the view's list).
//   k_lsmr_jv mode 0:  u <- J_h v - alpha u      mode 1:  u <- f  (the loss-scaled residual vector, LSMR's b)      mode 2:  u <- J_h v
//   partial[blockIdx.x] = sum of squares of what this workgroup wrote
// ---------------------------------------------------------------------------------------------------------------
template <int ND, int FISH, int MOTION, bool OPTK, bool ROBUST>
__global__ __launch_bounds__(64) void k_lsmr_jv(Dims d, Tables t, const int32_t* __restrict__ first, int mode,
                                                const double* __restrict__ dscale, const double* __restrict__ vin, double alpha,
                                                double* __restrict__ u, double* __restrict__ partial,
                                                const double* __restrict__ ls = nullptr) {
  constexpr bool ROLL = MOTION == MOTION_ROLLING;
  constexpr int DE = ROLL ? 12 : 6, NPB = MOTION == MOTION_STATIC ? 3 : 4, NPC = 6 * NPB, KI = OPTK ? 4 + ND : 0;
  constexpr int NV = DE + KI + 1;
  __shared__ uint16_t pidx[LIN_MAX_POINTS];
  __shared__ double vp[NPC], wl[DE + KI + 1];
  const int lane = threadIdx.x;
  if (ls != nullptr) {   // iteration of a device-resident solve (mcba_lsmr.h): alpha from the state, nothing to do once it stopped
    if (ls[LS_ISTOP] != 0.0) return;
    alpha = ls[LS_ALPHA];
  }
  const int n_active = t.active_views[0];
  double acc = 0.0;
  for (int vi = blockIdx.x; vi < n_active; vi += gridDim.x) {
    const int v = t.active_views[1 + vi];
    if (v < 0) continue;
    const int b = v % d.B, c = (v / d.B) % d.C, f = d.f0 + v / (d.B * d.C);
    if (mode != 1) {   // the view's scaled local parameter vector, then w = That vp
      if (lane < NPC + KI) {
        const int xi = local_to_x(d, f, c, b, lane);
        const double val = xi >= 0 ? dscale[xi] * vin[xi] : 0.0;
        if (lane < NPC) vp[lane] = val; else wl[DE + lane - NPC] = val;
      }
      lds_fence();
      if (lane < DE) {
        const double* Tm = t.tmat + (size_t)v * (DE * NPC) + lane * NPC;
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < NPC; ++j) sum += Tm[j] * vp[j];
        wl[lane] = sum;
      }
      lds_fence();
    }
    size_t out0 = (size_t)first[v];
    constexpr int NPB64 = LIN_MAX_POINTS / 64;
    for (int seg0 = 0; seg0 < d.P; seg0 += LIN_MAX_POINTS) {
      uint8_t inb[NPB64];
#pragma unroll
      for (int k = 0; k < NPB64; ++k) inb[k] = masked_load_row(t.inlier + (size_t)v * d.P, seg0 + k * 64 + lane, d.P);
      int count = 0;
#pragma unroll
      for (int k = 0; k < NPB64; ++k) {
        const bool in = inb[k] != 0;
        const unsigned long long m = __ballot(in);
        if (in) pidx[count + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(seg0 + k * 64 + lane);
        count += __popcll(m);
      }
      lds_fence();
      for (int base = 0; base < count; base += 64) {
        const int i = base + lane;
        if (i < count) {
          const int p = pidx[i];
          PointState<ND, ROLL> ps;
          point_state<ND, FISH, ROLL, ROBUST>(d, t, v, c, b, p, t.obs[(size_t)v * d.P + p], ps);
          double2 o;
          double2 old;
          old.x = old.y = 0.0;
          if (mode == 0) old = reinterpret_cast<const double2*>(u)[out0 + i];
          double bterm[2] = {0.0, 0.0};
          if (mode != 1 && d.off_boards >= 0) {   // adjusted board points (boards=True): + jp . (D v)[point], jp = rs A R_view
            const int gq = d.off_boards + 3 * (t.board_off[b] + p);
            double w3[3];
            board_point_direction<ROLL>(t, v, ps.tr, dscale[gq] * vin[gq], dscale[gq + 1] * vin[gq + 1], dscale[gq + 2] * vin[gq + 2], w3);
#pragma unroll
            for (int a = 0; a < 2; ++a)
              bterm[a] = ps.rs[a] * (ps.A[3 * a] * w3[0] + ps.A[3 * a + 1] * w3[1] + ps.A[3 * a + 2] * w3[2]);
          }
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            double row[NV];
            point_row<ND, ROLL, OPTK>(ps, a, row);
            double val;
            if (mode != 1) {
              val = 0.0;
#pragma unroll
              for (int k = 0; k < DE + KI; ++k) val += row[k] * wl[k];
              val += bterm[a];
              val -= alpha * (a == 0 ? old.x : old.y);
            } else {
              val = row[NV - 1];
            }
            if (a == 0) o.x = val; else o.y = val;
          }
          reinterpret_cast<double2*>(u)[out0 + i] = o;
          acc += o.x * o.x + o.y * o.y;
        }
      }
      out0 += (size_t)count;
      lds_fence();
    }
  }
  const double tot = wave_sum(acc);
  if (lane == 0) partial[blockIdx.x] = tot;
}

// per view: part[v][0 .. NPC + KI) = [That^T sum_p E_p^T u_p | sum_p K_p^T u_p] with u <- u * inv_beta (normalised in place)
template <int ND, int FISH, int MOTION, bool OPTK, bool ROBUST>
__global__ __launch_bounds__(64) void k_lsmr_jtu(Dims d, Tables t, const int32_t* __restrict__ first, double inv_beta,
                                                 double* __restrict__ u, double* __restrict__ part, int part_stride,
                                                 double* __restrict__ bpart, const double* __restrict__ ls = nullptr) {
  constexpr bool ROLL = MOTION == MOTION_ROLLING;
  constexpr int DE = ROLL ? 12 : 6, NPB = MOTION == MOTION_STATIC ? 3 : 4, NPC = 6 * NPB, KI = OPTK ? 4 + ND : 0;
  constexpr int NV = DE + KI + 1, NS = DE + KI;
  __shared__ uint16_t pidx[LIN_MAX_POINTS];
  __shared__ double sl[NS];
  const int lane = threadIdx.x;
  if (ls != nullptr) {
    if (ls[LS_ISTOP] != 0.0 || ls[LS_SKIPV] != 0.0) return;
    inv_beta = ls[LS_INV_BETA];
  }
  const int n_active = t.active_views[0];
  for (int vi = blockIdx.x; vi < n_active; vi += gridDim.x) {
    const int v = t.active_views[1 + vi];
    if (v < 0) continue;
    const int b = v % d.B, c = (v / d.B) % d.C;
    double sums[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) sums[k] = 0.0;
    size_t out0 = (size_t)first[v];
    constexpr int NPB64 = LIN_MAX_POINTS / 64;
    for (int seg0 = 0; seg0 < d.P; seg0 += LIN_MAX_POINTS) {
      uint8_t inb[NPB64];
#pragma unroll
      for (int k = 0; k < NPB64; ++k) inb[k] = masked_load_row(t.inlier + (size_t)v * d.P, seg0 + k * 64 + lane, d.P);
      int count = 0;
#pragma unroll
      for (int k = 0; k < NPB64; ++k) {
        const bool in = inb[k] != 0;
        const unsigned long long m = __ballot(in);
        if (in) pidx[count + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(seg0 + k * 64 + lane);
        count += __popcll(m);
      }
      lds_fence();
      for (int base = 0; base < count; base += 64) {
        const int i = base + lane;
        if (i < count) {
          const int p = pidx[i];
          PointState<ND, ROLL> ps;
          point_state<ND, FISH, ROLL, ROBUST>(d, t, v, c, b, p, t.obs[(size_t)v * d.P + p], ps);
          double2 uu = reinterpret_cast<const double2*>(u)[out0 + i];
          uu.x *= inv_beta;
          uu.y *= inv_beta;
          reinterpret_cast<double2*>(u)[out0 + i] = uu;
          if (bpart != nullptr) {   // boards=True: jp^T u of this observation (3 doubles, residual order), summed per point by k_lsmr_gather
            double q3[3], w3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) q3[k] = ps.rs[0] * uu.x * ps.A[k] + ps.rs[1] * uu.y * ps.A[3 + k];
            board_point_adjoint<ROLL>(t, v, ps.tr, q3, w3);
#pragma unroll
            for (int k = 0; k < 3; ++k) bpart[3 * (out0 + i) + k] = w3[k];
          }
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            double row[NV];
            point_row<ND, ROLL, OPTK>(ps, a, row);
            const double w = a == 0 ? uu.x : uu.y;
#pragma unroll
            for (int k = 0; k < NS; ++k) sums[k] += row[k] * w;
          }
        }
      }
      out0 += (size_t)count;
      lds_fence();
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const double tot = wave_sum(sums[k]);
      if (lane == 0) sl[k] = tot;
    }
    lds_fence();
    double* out = part + (size_t)v * part_stride;
    if (lane < NPC) {
      const double* Tm = t.tmat + (size_t)v * (DE * NPC);
      double sum = 0.0;
#pragma unroll
      for (int a = 0; a < DE; ++a) sum += Tm[a * NPC + lane] * sl[a];
      out[lane] = sum;
    } else if (lane < NPC + KI) {
      out[lane] = sl[DE + lane - NPC];
    }
    lds_fence();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_lsmr_fused: BOTH products of a Golub-Kahan step in ONE pass over the observations (round 5; k_lsmr_jv + k_lsmr_jtu
// evaluated the analytic row pair of every observation twice per LSMR iteration):
//     uhat <- J_h v - alpha (uhat_old inv_beta_old)          stored UN-normalised; beta = |uhat| is known after the pass
//     part[view] <- [That^T sum_p E_p^T uhat_p | sum_p K_p^T uhat_p]      (k_lsmr_gather2 applies 1 / beta to the sums)
// alpha, inv_beta_old and the stop flag come from the device-resident state `ls` (mcba_lsmr.h).  Element for element the
// arithmetic of uhat is that of k_lsmr_jv on the normalised u that k_lsmr_jtu used to store (u = uhat * inv_beta is formed
// on the fly with the same rounding).  The NS per-lane sums of a view are reduced with wave_reduce_many (one butterfly for
// all of them instead of NS wave_sums).  partial[blockIdx.x] = sum of squares of what this workgroup wrote.
// ---------------------------------------------------------------------------------------------------------------
template <int ND, int FISH, int MOTION, bool OPTK, bool ROBUST>
__global__ __launch_bounds__(64) void k_lsmr_fused(Dims d, Tables t, const int32_t* __restrict__ first,
                                                   const double* __restrict__ dscale, const double* __restrict__ vin,
                                                   double* __restrict__ u, double* __restrict__ partial,
                                                   double* __restrict__ part, int part_stride, double* __restrict__ bpart,
                                                   const double* __restrict__ ls) {
  constexpr bool ROLL = MOTION == MOTION_ROLLING;
  constexpr int DE = ROLL ? 12 : 6, NPB = MOTION == MOTION_STATIC ? 3 : 4, NPC = 6 * NPB, KI = OPTK ? 4 + ND : 0;
  constexpr int NV = DE + KI + 1, NS = DE + KI;
  __shared__ uint16_t pidx[LIN_MAX_POINTS];
  __shared__ double vp[NPC], wl[NS], sl[NS];
  const int lane = threadIdx.x;
  if (ls[LS_ISTOP] != 0.0) return;
  const double alpha = ls[LS_ALPHA], inv_beta_old = ls[LS_INV_BETA];
  const int n_active = t.active_views[0];
  double acc = 0.0;
  for (int vi = blockIdx.x; vi < n_active; vi += gridDim.x) {
    const int v = t.active_views[1 + vi];
    if (v < 0) continue;
    const int b = v % d.B, c = (v / d.B) % d.C, f = d.f0 + v / (d.B * d.C);
    // the view's scaled local parameter vector, then w = That vp
    if (lane < NPC + KI) {
      const int xi = local_to_x(d, f, c, b, lane);
      const double val = xi >= 0 ? dscale[xi] * vin[xi] : 0.0;
      if (lane < NPC) vp[lane] = val; else wl[DE + lane - NPC] = val;
    }
    lds_fence();
    if (lane < DE) {
      const double* Tm = t.tmat + (size_t)v * (DE * NPC) + lane * NPC;
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < NPC; ++j) sum += Tm[j] * vp[j];
      wl[lane] = sum;
    }
    lds_fence();
    double sums[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) sums[k] = 0.0;
    size_t out0 = (size_t)first[v];
    constexpr int NPB64 = LIN_MAX_POINTS / 64;
    for (int seg0 = 0; seg0 < d.P; seg0 += LIN_MAX_POINTS) {
      uint8_t inb[NPB64];
#pragma unroll
      for (int k = 0; k < NPB64; ++k) inb[k] = masked_load_row(t.inlier + (size_t)v * d.P, seg0 + k * 64 + lane, d.P);
      int count = 0;
#pragma unroll
      for (int k = 0; k < NPB64; ++k) {
        const bool in = inb[k] != 0;
        const unsigned long long m = __ballot(in);
        if (in) pidx[count + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(seg0 + k * 64 + lane);
        count += __popcll(m);
      }
      lds_fence();
      for (int base = 0; base < count; base += 64) {
        const int i = base + lane;
        if (i < count) {
          const int p = pidx[i];
          PointState<ND, ROLL> ps;
          point_state<ND, FISH, ROLL, ROBUST>(d, t, v, c, b, p, t.obs[(size_t)v * d.P + p], ps);
          double2 old = reinterpret_cast<const double2*>(u)[out0 + i];
          old.x *= inv_beta_old;      // (the normalised u of the previous step, rounded as k_lsmr_jtu stored it)
          old.y *= inv_beta_old;
          double bterm[2] = {0.0, 0.0};
          if (d.off_boards >= 0) {   // adjusted board points: + jp . (D v)[point]
            const int gq = d.off_boards + 3 * (t.board_off[b] + p);
            double w3[3];
            board_point_direction<ROLL>(t, v, ps.tr, dscale[gq] * vin[gq], dscale[gq + 1] * vin[gq + 1], dscale[gq + 2] * vin[gq + 2], w3);
#pragma unroll
            for (int a = 0; a < 2; ++a)
              bterm[a] = ps.rs[a] * (ps.A[3 * a] * w3[0] + ps.A[3 * a + 1] * w3[1] + ps.A[3 * a + 2] * w3[2]);
          }
          double2 o;
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            double row[NV];
            point_row<ND, ROLL, OPTK>(ps, a, row);
            double val = 0.0;
#pragma unroll
            for (int k = 0; k < NS; ++k) val += row[k] * wl[k];
            val += bterm[a];
            val -= alpha * (a == 0 ? old.x : old.y);
#pragma unroll
            for (int k = 0; k < NS; ++k) sums[k] += row[k] * val;
            if (a == 0) o.x = val; else o.y = val;
          }
          reinterpret_cast<double2*>(u)[out0 + i] = o;
          acc += o.x * o.x + o.y * o.y;
          if (bpart != nullptr) {   // boards=True: jp^T uhat of this observation (summed per point by the gather)
            double q3[3], w3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) q3[k] = ps.rs[0] * o.x * ps.A[k] + ps.rs[1] * o.y * ps.A[3 + k];
            board_point_adjoint<ROLL>(t, v, ps.tr, q3, w3);
#pragma unroll
            for (int k = 0; k < 3; ++k) bpart[3 * (out0 + i) + k] = w3[k];
          }
        }
      }
      out0 += (size_t)count;
      lds_fence();
    }
#if defined(MCBA_EXP_F2_NO_REDUCE)  // what-if (variant builds only): no butterfly over the wave, no That^T product, one store per view
    if (lane == 0) part[lsmr_part_index(d, v, 0)] = sums[0] + sums[NS - 1];
    continue;
#endif
    const double tot = wave_reduce_many<NS>(sums, lane);
    if (many_writer<NS>(lane)) sl[many_index(lane)] = tot;
    lds_fence();
    double* out = part + (size_t)v * part_stride;
    if (lane < NPC) {
      const double* Tm = t.tmat + (size_t)v * (DE * NPC);
      double sum = 0.0;
#pragma unroll
      for (int a = 0; a < DE; ++a) sum += Tm[a * NPC + lane] * sl[a];
      out[lane] = sum;
    } else if (lane < NPC + KI) {
      out[lane] = sl[DE + lane - NPC];
    }
    lds_fence();
  }
  const double tot = wave_sum(acc);
  if (lane == 0) partial[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------------------------------------------
// k_lsmr_fused2: k_lsmr_fused + the scalar recurrence and the vector update of the PREVIOUS Golub-Kahan step in its head, so that
// an LSMR iteration is TWO launches (k_lsmr_fused2 -> k_lsmr_gather3).  Every workgroup (one wavefront) folds the |v_raw|^2
// partials of the last gather and forms alpha and 1 / alpha (redundantly, bit-identical); the plane rotations, the update
// coefficients (lsmr_state_rotate) and the vector update h_bar, x, h run in the TAIL of the workgroups that own a 64-entry slice
// of the vectors, xpart[slice] = its part of |x|^2, and the last workgroup publishes the state (in: lsIn, written by the gather;
// out: lsOut -- double buffer, see k_lsmr_gather3).  v is never stored normalised: v = v_raw / alpha is formed where it is read.
// ---------------------------------------------------------------------------------------------------------------
// Source of an observation's data in k_lsmr_fused2 (template MODE):
//   0  masks: the frame-major tables -- mask bytes compacted per view in LDS, observation and board point gathered by point index
//      (the only form with boards=True: the board-point block needs the point index, and the board points move)
//   3  compact (default): the observations of a view streamed in residual order from the compacted tables LsmrCompact (built once
//      per inlier set) -- no mask bytes, no compaction, no gathers, and ONE round trip per view instead of three: the view's
//      descriptor {view, first, count} is a scalar load prefetched one view ahead, and the first chunk's loads go out together with
//      the That / parameter staging loads.  What-if builds of the masks form (profiles/r06_f2_whatif.txt) showed the kernel
//      latency-bound on its per-view chain of dependent loads: without ANY arithmetic 22.2 of its 26.7 us remained.
//   4  compact + store the per-observation state (first iteration of a solve under the cached form)
//   2  stream the state back.  The Jacobian is FIXED during an LSMR solve: what an observation contributes to both products
//      is determined by its PointState --
// A = d(u, v) / d X_cam, the camera-frame points of the start / end chain, the scan time (+ the robust row scales) -- 13 doubles
// (rolling shutter; 9 static; + 2 robust): MODE 2 reads them instead of observations / board points and re-deriving the state: the
// intrinsic columns K_c are rebuilt from X_cam by the same project_point (its A / uv halves are dead code there), everything behind the
// state is the same source.  Cache layout: blocks of 64 observations in residual order, component-major inside a block (one 512-byte
// run per component and wavefront).  mcba_debug_set_lsmr_fused(h, 3); measured in profiles/r06_lsmr_iteration.txt.
template <bool ROLL, bool ROBUST>
struct LsmrCacheLayout {
  static constexpr int NC = 6 + 3 + (ROLL ? 4 : 0) + (ROBUST ? 2 : 0);
  __host__ __device__ static size_t index(size_t gi, int k) { return (gi >> 6) * (size_t)(NC * 64) + (size_t)k * 64 + (gi & 63); }
};
__host__ __device__ inline int lsmr_cache_components(int motion, int loss) {
  return 6 + 3 + (motion == MOTION_ROLLING ? 4 : 0) + (loss != 0 ? 2 : 0);
}

#if defined(MCBA_EXP_F2_WAVES)       // what-if (variant builds only): force N waves per SIMD (the register allocator spills to fit)
#define MCBA_F2_OCCUPANCY __attribute__((amdgpu_waves_per_eu(MCBA_EXP_F2_WAVES, MCBA_EXP_F2_WAVES)))
#else
// two waves per SIMD, pinned: the instantiations need 166 - 256 registers; without the attribute a small change of the source lets the
// allocator drift above 256 (one wave per SIMD) silently.  Three waves cost the rolling-shutter instantiation 276 B of scratch per lane
// (35 spilled doubles, measured 60 us instead of 24), the static ones 0 - 68 B with mixed results (profiles/r06_lsmr_experiments.txt).
#define MCBA_F2_OCCUPANCY __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
template <int ND, int FISH, int MOTION, bool OPTK, bool ROBUST, int MODE>
__global__ __launch_bounds__(64) MCBA_F2_OCCUPANCY void k_lsmr_fused2(Dims d, Tables t, const int32_t* __restrict__ first,
                                                    const double* __restrict__ dscale, const double* __restrict__ vin,
                                                    double* __restrict__ u, double* __restrict__ partial, double* __restrict__ xpart,
                                                    double* __restrict__ part, int part_stride, double* __restrict__ bpart,
                                                    const double* __restrict__ lsIn, double* __restrict__ lsOut,
                                                    const double* __restrict__ vpart, int nv, double* __restrict__ hbar,
                                                    double* __restrict__ xv, double* __restrict__ hv, double* __restrict__ cache,
                                                    LsmrCompact cp) {
  constexpr bool ROLL = MOTION == MOTION_ROLLING;
  constexpr bool DESC = MODE >= 2;          // the view list comes as descriptors {view, first, count}
  constexpr int DE = ROLL ? 12 : 6, NPB = MOTION == MOTION_STATIC ? 3 : 4, NPC = 6 * NPB, KI = OPTK ? 4 + ND : 0;
  constexpr int NS = DE + KI;
  using CL = LsmrCacheLayout<ROLL, ROBUST>;
  __shared__ uint16_t pidx[LIN_MAX_POINTS];
  __shared__ double vp[NPC], wl[NS], sl[NS];
  __shared__ double TmS[DE * NPC];   // That of the view: requested with the masks and parameters (ONE round trip), used by both products
  constexpr int NTL = (DE * NPC + 63) / 64;
  const int lane = threadIdx.x;
  const int last = (int)gridDim.x - 1;
  if (lsIn[LS_ISTOP] != 0.0) {   // stopped by the tests of the last gather: hand the flag on, leave x as it is
    if ((int)blockIdx.x == last && lane == 0) lsOut[LS_ISTOP] = lsIn[LS_ISTOP];
    return;
  }
  // ---- head: ONLY what the product needs -- alpha and 1 / alpha, the first two operations of lsmr_state_rotate -- in every
  // workgroup; the rest of the recurrence, the vector update and the state go to the TAIL of the workgroups that own a slice of
  // the vectors (the last ones of the grid: the active-view list is largest-first, so they hold the lightest views).  The full
  // recurrence in the head of every workgroup was 6 us on the critical path of the launch (round 5 trace).
  const bool pending = lsIn[LS_PENDING] != 0.0;
  double alpha = lsIn[LS_ALPHA], inv_alpha = lsIn[LS_INV_ALPHA], v2 = 0.0;
  const double inv_beta_old = lsIn[LS_INV_BETA];
  if (pending) {
    v2 = wave_fold_batched<4>(vpart, nv, lane);
    inv_alpha = 1.0;
    if (lsIn[LS_SKIPV] == 0.0) {   // (lsmr_state_rotate: the same two operations, the same rounding)
      alpha = sqrt(v2);
      if (alpha > 0) inv_alpha = 1.0 / alpha;
    }
  }
  const int n_active = t.active_views[0];
  double acc = 0.0;
  double sums[NS];
  // one observation: uhat, its part of |uhat|^2 and of the per-view sums of row^T uhat, from the state ps (shared by all three forms)
  auto observe = [&](const PointState<ND, ROLL>& ps, double2 old, int v, int b, int p, size_t pair) {
#if defined(MCBA_EXP_F2_NO_MATH)    // what-if (variant builds only): loads, stores and per-view work stay, the two products go
    reinterpret_cast<double2*>(u)[pair] = old;
    acc += old.x + ps.A[0] + ps.Xs[0];
    sums[0] += old.y;
    return;
#endif
    old.x *= inv_beta_old;
    old.y *= inv_beta_old;
    double bterm[2] = {0.0, 0.0};
    if (d.off_boards >= 0) {
      const int gq = d.off_boards + 3 * (t.board_off[b] + p);
      double w3[3];
      board_point_direction<ROLL>(t, v, ps.tr, dscale[gq] * (vin[gq] * inv_alpha), dscale[gq + 1] * (vin[gq + 1] * inv_alpha),
                                  dscale[gq + 2] * (vin[gq + 2] * inv_alpha), w3);
#pragma unroll
      for (int a = 0; a < 2; ++a)
        bterm[a] = ps.rs[a] * (ps.A[3 * a] * w3[0] + ps.A[3 * a + 1] * w3[1] + ps.A[3 * a + 2] * w3[2]);
    }
    // The row pair is never formed: with E_a = [X x a_a | a_a] (base_row) the product is  E_a . w = a_a . (w_t + w_r x X)  and
    // the adjoint  sum_a E_a^T c_a = [X x q | q],  q = sum_a c_a a_a  -- two cross products each way instead of 2 x 2 x 12
    // multiply-adds with 24 row entries (rolling shutter: y and q are blended with the scan time).  K_a stays a plain dot.
    constexpr int KIA = 4 + ND;
    double y[3];
    {
      const double w0 = wl[0], w1 = wl[1], w2 = wl[2];
      const double ys0 = wl[3] + (w1 * ps.Xs[2] - w2 * ps.Xs[1]);
      const double ys1 = wl[4] + (w2 * ps.Xs[0] - w0 * ps.Xs[2]);
      const double ys2 = wl[5] + (w0 * ps.Xs[1] - w1 * ps.Xs[0]);
      if constexpr (ROLL) {
        const double w6 = wl[6], w7 = wl[7], w8 = wl[8];
        const double ye0 = wl[9] + (w7 * ps.Xe[2] - w8 * ps.Xe[1]);
        const double ye1 = wl[10] + (w8 * ps.Xe[0] - w6 * ps.Xe[2]);
        const double ye2 = wl[11] + (w6 * ps.Xe[1] - w7 * ps.Xe[0]);
        const double ts = 1.0 - ps.tr;
        y[0] = ts * ys0 + ps.tr * ye0; y[1] = ts * ys1 + ps.tr * ye1; y[2] = ts * ys2 + ps.tr * ye2;
      } else {
        y[0] = ys0; y[1] = ys1; y[2] = ys2;
      }
    }
    double cv[2];   // c_a = rs_a uhat_a
    double2 o;
    double dots[2] = {ps.A[0] * y[0] + ps.A[1] * y[1] + ps.A[2] * y[2], ps.A[3] * y[0] + ps.A[4] * y[1] + ps.A[5] * y[2]};
    if constexpr (OPTK) {
#pragma unroll
      for (int k = 0; k < KI; ++k) {   // (row by row the same additions in the same order as before)
        const double wk = wl[DE + k];
        dots[0] += ps.Kc[k] * wk;
        dots[1] += ps.Kc[KIA + k] * wk;
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const double dot = dots[a];
      double val = ps.rs[a] * dot;
      val += bterm[a];
      val -= alpha * (a == 0 ? old.x : old.y);
      cv[a] = ps.rs[a] * val;
      if (a == 0) o.x = val; else o.y = val;
    }
    {
      const double q0 = cv[0] * ps.A[0] + cv[1] * ps.A[3], q1 = cv[0] * ps.A[1] + cv[1] * ps.A[4], q2 = cv[0] * ps.A[2] + cv[1] * ps.A[5];
      const double ws = ROLL ? 1.0 - ps.tr : 1.0;
      sums[0] += ws * (ps.Xs[1] * q2 - ps.Xs[2] * q1);
      sums[1] += ws * (ps.Xs[2] * q0 - ps.Xs[0] * q2);
      sums[2] += ws * (ps.Xs[0] * q1 - ps.Xs[1] * q0);
      sums[3] += ws * q0; sums[4] += ws * q1; sums[5] += ws * q2;
      if constexpr (ROLL) {
        sums[6] += ps.tr * (ps.Xe[1] * q2 - ps.Xe[2] * q1);
        sums[7] += ps.tr * (ps.Xe[2] * q0 - ps.Xe[0] * q2);
        sums[8] += ps.tr * (ps.Xe[0] * q1 - ps.Xe[1] * q0);
        sums[9] += ps.tr * q0; sums[10] += ps.tr * q1; sums[11] += ps.tr * q2;
      }
      if constexpr (OPTK) {
#pragma unroll
        for (int k = 0; k < KI; ++k) sums[DE + k] += cv[0] * ps.Kc[k] + cv[1] * ps.Kc[KIA + k];
      }
    }
    reinterpret_cast<double2*>(u)[pair] = o;
    acc += o.x * o.x + o.y * o.y;
    if (bpart != nullptr) {
      double q3[3], w3[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) q3[k] = ps.rs[0] * o.x * ps.A[k] + ps.rs[1] * o.y * ps.A[3 + k];
      board_point_adjoint<ROLL>(t, v, ps.tr, q3, w3);
#pragma unroll
      for (int k = 0; k < 3; ++k) bpart[3 * pair + k] = w3[k];
    }
  };
  // (a boustrophedon order of the largest-first list -- odd rounds backwards, pairing large with small views -- was measured:
  //  40.3 against 39.0 us per iteration; what a workgroup spends is dominated by the per-view round trips, not by its observations)
  int4 dnext = make_int4(0, 0, 0, 0);
  if constexpr (DESC) {
    if ((int)blockIdx.x < n_active) dnext = cp.desc[blockIdx.x];
  }
#if defined(MCBA_EXP_F2_PROF)   // profiling build: per-workgroup stamps (shader clock) written to `cache` as long long [grid][8]
  const long long pf_t0 = clock64();
  long long pf_stage = 0, pf_chunks = 0, pf_epi = 0, pf_views = 0, pf_nch = 0;
#define PF_STAMP(var) const long long var = clock64()
#else
#define PF_STAMP(var)
#endif
  for (int vi = blockIdx.x; vi < n_active; vi += gridDim.x) {
    int v, desc_first = 0, desc_count = 0;
    if constexpr (DESC) {
      v = dnext.x; desc_first = dnext.y; desc_count = dnext.z;
      if (vi + (int)gridDim.x < n_active) dnext = cp.desc[vi + gridDim.x];   // (scalar load, one view ahead)
    } else {
      v = t.active_views[1 + vi];
    }
    if (v < 0) continue;
    PF_STAMP(pf_a);
    const int b = v % d.B, c = (v / d.B) % d.C, f = d.f0 + v / (d.B * d.C);
    constexpr int NPB64 = LIN_MAX_POINTS / 64;
    // the mask bytes of the first segment are requested in front of the parameter staging: one round trip for both
    uint8_t inb[NPB64];
    if constexpr (!DESC) {
#pragma unroll
      for (int k = 0; k < NPB64; ++k) inb[k] = masked_load_row(t.inlier + (size_t)v * d.P, k * 64 + lane, d.P);
    }
    double tl[NTL];
    // compact form: the first chunk of the view is requested with the staging loads (same round trip)
    double2 ob_cur = make_double2(0.0, 0.0), xy_cur = make_double2(0.0, 0.0), old_cur = make_double2(0.0, 0.0);
    double z_cur = 0.0;
    double2 ob_nxt = make_double2(0.0, 0.0), xy_nxt = make_double2(0.0, 0.0), old_nxt = make_double2(0.0, 0.0);
    double z_nxt = 0.0;
    // the observed point is only needed for the residual (robust row scaling); the products of the linear loss never read it, and the
    // scan time of a rolling-shutter observation (observed row / image height) comes precomputed with the tables
    constexpr bool NEED_OB = ROBUST;
    double tr_cur = 0.0, tr_nxt = 0.0;
    if constexpr (MODE >= 3) {
      const size_t g0 = (size_t)desc_first + (size_t)(lane < desc_count ? lane : 0);
      if constexpr (NEED_OB) ob_cur = cp.obs[g0];
      if constexpr (ROLL) tr_cur = cp.tr[g0];
      xy_cur = cp.bxy[g0];
      z_cur = cp.bz[g0];
      old_cur = reinterpret_cast<const double2*>(u)[g0];
    }
    {
      const double* tg = t.tmat + (size_t)v * (DE * NPC);
#pragma unroll
      for (int k = 0; k < NTL; ++k) tl[k] = masked_load_row(tg, k * 64 + lane, DE * NPC);
#if defined(MCBA_EXP_F2_NO_TMAT)   // what-if (variant builds only): the 2.3 KB of That per view are not streamed
#pragma unroll
      for (int k = 0; k < NTL; ++k) tl[k] = 1e-3 * (double)(k * 64 + lane);
#endif
    }
    if (lane < NPC + KI) {
      const int xi = local_to_x(d, f, c, b, lane);
      const double val = xi >= 0 ? dscale[xi] * (vin[xi] * inv_alpha) : 0.0;
      if (lane < NPC) vp[lane] = val; else wl[DE + lane - NPC] = val;
    }
#pragma unroll
    for (int k = 0; k < NTL; ++k)
      if (k * 64 + lane < DE * NPC) TmS[k * 64 + lane] = tl[k];
    lds_fence();
    if (lane < DE) {
      const double* Tm = TmS + lane * NPC;
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < NPC; ++j) sum += Tm[j] * vp[j];
      wl[lane] = sum;
    }
    lds_fence();
#pragma unroll
    for (int k = 0; k < NS; ++k) sums[k] = 0.0;
    PF_STAMP(pf_b);
    size_t out0 = DESC ? (size_t)desc_first : (size_t)first[v];
    if constexpr (MODE >= 3) {
      // ---- compact form: observation, board point and old uhat stream in residual order; the NEXT chunk is requested before the
      // current one is evaluated
      const int count = desc_count;
      // (two chunks in flight instead of one were measured: 26.4 against 25.3 us at the north-star rig -- profiles/r06_lsmr_experiments.txt)
      for (int base = 0; base < count; base += 64) {
        const int i = base + lane, inx = i + 64;
        const size_t gn = out0 + (size_t)(inx < count ? inx : 0);
        if constexpr (NEED_OB) ob_nxt = cp.obs[gn];
        if constexpr (ROLL) tr_nxt = cp.tr[gn];
        xy_nxt = cp.bxy[gn];
        z_nxt = cp.bz[gn];
        old_nxt = reinterpret_cast<const double2*>(u)[gn];
        if (i < count) {
          const double X_cur[3] = {xy_cur.x, xy_cur.y, z_cur};
          PointState<ND, ROLL> ps;
          point_state<ND, FISH, ROLL, ROBUST>(d, t, v, c, b, 0, ob_cur, ps, X_cur, nullptr, nullptr, nullptr, ROLL ? &tr_cur : nullptr);
          if constexpr (MODE == 4) {   // the state of the observation for the cached iterations that follow
            const size_t gi = out0 + (size_t)i;
#pragma unroll
            for (int k = 0; k < 6; ++k) cache[CL::index(gi, k)] = ps.A[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) cache[CL::index(gi, 6 + k)] = ps.Xs[k];
            if constexpr (ROLL) {
#pragma unroll
              for (int k = 0; k < 3; ++k) cache[CL::index(gi, 9 + k)] = ps.Xe[k];
              cache[CL::index(gi, 12)] = ps.tr;
            }
            if constexpr (ROBUST) { cache[CL::index(gi, CL::NC - 2)] = ps.rs[0]; cache[CL::index(gi, CL::NC - 1)] = ps.rs[1]; }
          }
          observe(ps, old_cur, v, b, 0, out0 + (size_t)i);
        }
        ob_cur = ob_nxt; xy_cur = xy_nxt; z_cur = z_nxt; old_cur = old_nxt; tr_cur = tr_nxt;
      }
    } else if constexpr (MODE == 2) {
      // ---- the state of every observation comes back from the cache: no masks, no compaction, no observation / board-point reads
      const double* cam = t.cam + (size_t)c * CAM_STRIDE;
      const int cached_count = desc_count;
      for (int base = 0; base < cached_count; base += 64) {
        const int i = base + lane;
        const size_t gi = out0 + (size_t)(i < cached_count ? i : 0);
        double cv_[CL::NC];
#pragma unroll
        for (int k = 0; k < CL::NC; ++k) cv_[k] = cache[CL::index(gi, k)];
        const double2 old = reinterpret_cast<const double2*>(u)[gi];
        if (i < cached_count) {
          PointState<ND, ROLL> ps;
#pragma unroll
          for (int k = 0; k < 6; ++k) ps.A[k] = cv_[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) ps.Xs[k] = cv_[6 + k];
          double Xc[3];
          if constexpr (ROLL) {
#pragma unroll
            for (int k = 0; k < 3; ++k) ps.Xe[k] = cv_[9 + k];
            ps.tr = cv_[12];
#pragma unroll
            for (int k = 0; k < 3; ++k) Xc[k] = ps.Xs[k] * (1.0 - ps.tr) + ps.Xe[k] * ps.tr;   // (slot_forward's blend)
          } else {
            ps.tr = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { ps.Xe[k] = 0.0; Xc[k] = ps.Xs[k]; }
          }
          if constexpr (ROBUST) { ps.rs[0] = cv_[CL::NC - 2]; ps.rs[1] = cv_[CL::NC - 1]; }
          else { ps.rs[0] = 1.0; ps.rs[1] = 1.0; }
          if constexpr (OPTK) {
            double uv_[2], A_[6];
            project_point<ND, FISH, true>(cam, cam + CAM_TILT, Xc, uv_, A_, ps.Kc);   // (only the K_c half survives)
          }
          observe(ps, old, v, b, /*p: the point index is only needed with boards=True, which the host keeps on form 2*/ 0, gi);
        }
      }
    } else {
    for (int seg0 = 0; seg0 < d.P; seg0 += LIN_MAX_POINTS) {
      if (seg0 > 0) {
#pragma unroll
        for (int k = 0; k < NPB64; ++k) inb[k] = masked_load_row(t.inlier + (size_t)v * d.P, seg0 + k * 64 + lane, d.P);
      }
      int count = 0;
#pragma unroll
      for (int k = 0; k < NPB64; ++k) {
        const bool in = inb[k] != 0;
        const unsigned long long m = __ballot(in);
        if (in) pidx[count + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(seg0 + k * 64 + lane);
        count += __popcll(m);
      }
      lds_fence();
      // observation, board point and old uhat of the NEXT chunk are requested before the current one is evaluated (as in k_cost)
      int p_cur = lane < count ? pidx[lane] : 0;
      ob_cur = t.obs[(size_t)v * d.P + p_cur];
      double X_cur[3], X_nxt[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) X_cur[k] = t.board_points[3 * (size_t)(b * d.P + p_cur) + k];
      const size_t o0 = count > 0 ? out0 : 0;   // (a segment without inliers behind the last residual must not read past the vector)
      old_cur = reinterpret_cast<const double2*>(u)[o0 + (lane < count ? lane : 0)];
      for (int base = 0; base < count; base += 64) {
        const int i = base + lane, inx = i + 64;
        const int p_nxt = inx < count ? pidx[inx] : p_cur;
        const double2 ob_nxt = t.obs[(size_t)v * d.P + p_nxt];
#pragma unroll
        for (int k = 0; k < 3; ++k) X_nxt[k] = t.board_points[3 * (size_t)(b * d.P + p_nxt) + k];
        const double2 old_nxt = reinterpret_cast<const double2*>(u)[o0 + (inx < count ? inx : 0)];
        if (i < count) {
          const int p = p_cur;
          PointState<ND, ROLL> ps;
#if defined(MCBA_EXP_F2_NO_STATE)   // what-if (variant builds only): the forward model + derivatives of the observation are not evaluated
          for (int k = 0; k < 6; ++k) ps.A[k] = ob_cur.x + k;
          for (int k = 0; k < 3; ++k) { ps.Xs[k] = X_cur[k]; ps.Xe[k] = X_cur[k] + 1.0; }
          for (int k = 0; k < 2 * (4 + ND); ++k) ps.Kc[k] = ob_cur.y + k;
          ps.tr = 0.5; ps.rs[0] = ps.rs[1] = 1.0;
#else
          point_state<ND, FISH, ROLL, ROBUST>(d, t, v, c, b, p, ob_cur, ps, X_cur);
#endif
          observe(ps, old_cur, v, b, p, out0 + (size_t)i);
        }
        p_cur = p_nxt;
        ob_cur = ob_nxt;
        old_cur = old_nxt;
#pragma unroll
        for (int k = 0; k < 3; ++k) X_cur[k] = X_nxt[k];
      }
      out0 += (size_t)count;
      lds_fence();
    }
    }
    PF_STAMP(pf_c);
#if defined(MCBA_EXP_F2_NO_REDUCE)  // what-if (variant builds only): no butterfly over the wave, no That^T product, one store per view
    if (lane == 0) part[lsmr_part_index(d, v, 0)] = sums[0] + sums[NS - 1];
    continue;
#endif
    const double tot = wave_reduce_many<NS>(sums, lane);
    if (many_writer<NS>(lane)) sl[many_index(lane)] = tot;
    lds_fence();
    // (transposed layout, lsmr_part_index: the gather then sums contiguous runs)
    if (lane < NPC) {
      double sum = 0.0;
#pragma unroll
      for (int a = 0; a < DE; ++a) sum += TmS[a * NPC + lane] * sl[a];
      part[lsmr_part_index(d, v, lane)] = sum;
    } else if (lane < NPC + KI) {
      part[lsmr_part_index(d, v, lane)] = sl[DE + lane - NPC];
    }
    lds_fence();
#if defined(MCBA_EXP_F2_PROF)
    { const long long pf_d = clock64(); pf_stage += pf_b - pf_a; pf_chunks += pf_c - pf_b; pf_epi += pf_d - pf_c; ++pf_views; pf_nch += (desc_count + 63) / 64; }
#endif
  }
#if defined(MCBA_EXP_F2_PROF)
  if (lane == 0 && cache != nullptr) {
    long long* pf = reinterpret_cast<long long*>(cache) + 8 * (size_t)blockIdx.x;
    pf[0] = pf_t0; pf[1] = clock64(); pf[2] = pf_stage; pf[3] = pf_chunks; pf[4] = pf_epi; pf[5] = pf_views; pf[6] = pf_nch;
    unsigned hw;   // XCC / SE / CU / SIMD / wave slot of this wavefront
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    pf[7] = hw;
  }
#endif
  const double tot = wave_sum(acc);
  if (lane == 0) partial[blockIdx.x] = tot;
  // ---- tail (slice owners): rotation, vector update h_bar, x, h of slice j = entries j 64 + lane (+ multiples of 64 gridDim), the
  // slice's part of |x|^2; the last workgroup publishes the state
  const int j = last - (int)blockIdx.x;
  if (j * 64 < d.n || j == 0) {
    double L[LS_NSLOTS];
#pragma unroll
    for (int k = 0; k < LS_NSLOTS; ++k) L[k] = lsIn[k];
    double xsq = 0.0;
    if (pending) {
      lsmr_state_rotate(L, v2);
      L[LS_PENDING] = 0.0;
      const double ia = L[LS_INV_ALPHA], c_hbar = L[LS_C_HBAR], c_x = L[LS_C_X], c_h = L[LS_C_H];
      for (int i = j * 64 + lane; i < d.n; i += gridDim.x * 64) {
        const double vi = vin[i] * ia;
        const double hb = c_hbar * hbar[i] + hv[i];
        hbar[i] = hb;
        const double xi = xv[i] + c_x * hb;
        xv[i] = xi;
        hv[i] = c_h * hv[i] + vi;
        xsq += d.entry_weight(i) * (xi * xi);
      }
    }
    xsq = wave_sum(xsq);
    if (lane == 0) xpart[j] = xsq;
    if (j == 0 && lane == 0) {
#pragma unroll
      for (int k = 0; k < LS_NSLOTS; ++k) lsOut[k] = L[k];
    }
  }
}

#undef PF_STAMP

// ---------------------------------------------------------------------------------------------------------------
// k_jacobian: analytic Jacobian rows in the column order of Calibration.sparsity_matrix
//             (optimization/calibration.py:173-196).  One thread per inlier observation (not a hot path).
// ---------------------------------------------------------------------------------------------------------------
template <int ND, int FISH, bool ROLL>
__global__ void k_jacobian(Dims d, Tables t, int row_nnz, double* __restrict__ vals, int32_t* __restrict__ cols) {
  constexpr int DE = ROLL ? 12 : 6, KIA = 4 + ND, NV = DE + KIA + 1;
  const int n = d.slots();
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const int idx = t.obs_index[s];
    if (idx < 0) continue;
    const int p = s % d.P, v = s / d.P;
    const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
    Dims dl = d;
    dl.loss = 0;   // the Jacobian of evaluate() itself: no robust scaling
    double vr[2 * NV], jp[6];
    point_rows<ND, FISH, ROLL, true>(dl, t, v, c, b, p, t.obs[s], vr, jp);
    auto xcol = [&](int col) { return t.int2ext != nullptr ? t.int2ext[col] : col; };   // caller's (ragged) column index
    double* o0 = vals + (size_t)(2 * idx) * row_nnz;
    double* o1 = o0 + row_nnz;
    int32_t* oc = cols + (size_t)idx * row_nnz;
    int pos = 0;
    // ascending x order: camera pose | board pose | motion block(s) | intrinsics (with the structurally-present skew)
    const int order[4] = {0, d.NPB - 1, 1, 2};
    for (int oi = 0; oi < d.NPB; ++oi) {
      const int k = order[oi];
      if (local_to_x(d, f, c, b, 6 * k) < 0) continue;
      for (int jj = 0; jj < 6; ++jj) {
        double col[12];
        view_column(d, t, f, c, b, 6 * k + jj, col);
        double a0 = 0.0, a1 = 0.0;
        for (int a = 0; a < DE; ++a) {
          a0 += vr[a] * col[a];
          a1 += vr[NV + a] * col[a];
        }
        o0[pos] = a0;
        o1[pos] = a1;
        oc[pos] = xcol(local_to_x(d, f, c, b, 6 * k + jj));
        ++pos;
      }
    }
    if (d.off_cameras >= 0) {
      const int base = d.off_cameras + c * (5 + ND);
      for (int q = 0; q < 5 + ND; ++q) {
        const int lq = q < 4 ? q : q - 1;
        // the skew slot is structurally present with a zero derivative; a coefficient this camera's model does not have
        // (ragged rigs) is no column at all: value 0, column -1
        const bool skew = q == 4;
        const bool absent = !skew && d.cam_kmask != nullptr && ((d.cam_kmask[c] >> lq) & 1u);
        o0[pos] = (skew || absent) ? 0.0 : vr[DE + lq];
        o1[pos] = (skew || absent) ? 0.0 : vr[NV + DE + lq];
        oc[pos] = absent ? -1 : xcol(base + q);
        ++pos;
      }
    }
    if (d.off_boards >= 0) {   // adjusted board points (board/charuco.py:112-117): last block of x
      const int base = d.off_boards + 3 * (t.board_off[b] + p);
      for (int k = 0; k < 3; ++k) {
        o0[pos] = jp[k];
        o1[pos] = jp[3 + k];
        oc[pos] = xcol(base + k);
        ++pos;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_points: normal-equation blocks of the adjusted board points (optimize.boards, `adjust_board`).
//   Board point q = (b, p) carries 3 shared parameters; its Jacobian per observation is jp = A R_view (2 x 3).  The
//   blocks  H[q, q] (3x3),  H[q, camera c | intrinsics c] (sum over frames),  H[q, board b] / H[q, hand-eye] (sum over
//   all views),  H[q, frame f] (sum over cameras -> H_fs columns)  and g[q] are gathered by ONE WORKGROUP PER POINT:
//   threads own frames, cameras are walked in an outer loop, so every output element is produced by a fixed thread /
//   a fixed reduction order (no atomics).  The local Jacobian columns come from the same view_column / point_rows
//   functions as k_jacobian.  The remaining blocks come from k_linearize.
// ---------------------------------------------------------------------------------------------------------------
template <int ND, int FISH, int MOTION, bool OPTK>
__global__ __launch_bounds__(256) void k_points(Dims d, Tables t, double* __restrict__ Hss, double* __restrict__ Hfs,
                                                double* __restrict__ g) {
  constexpr bool ROLL = MOTION == MOTION_ROLLING;
  constexpr int DE = ROLL ? 12 : 6, NPB = MOTION == MOTION_STATIC ? 3 : 4, KI = OPTK ? 4 + ND : 0;
  constexpr int NV = DE + KI + 1, NPC = 6 * NPB, CW = 6 + KI;
  constexpr int NMISC = 18 + 6 + 3 + (MOTION == MOTION_HAND_EYE ? 36 : 0);   // board | pt-pt | g | hand-eye
  __shared__ double red[4][3 * CW > NMISC ? 3 * CW : NMISC];

  const int q = blockIdx.x;                       // global point index over all boards
  int b = 0;
  while (q >= t.board_off[b + 1]) ++b;
  const int p = q - t.board_off[b];
  const int ns = d.ns, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gq = d.off_boards + 3 * q;            // x index of the point's first coordinate
  const int sq = d.x_to_shared(gq);

  double misc[NMISC];
  for (int i = 0; i < NMISC; ++i) misc[i] = 0.0;

  for (int c = 0; c < d.C; ++c) {
    double camacc[3 * CW];
    for (int i = 0; i < 3 * CW; ++i) camacc[i] = 0.0;
    for (int fl = threadIdx.x; fl < d.Fl; fl += blockDim.x) {
      const int v = (fl * d.C + c) * d.B + b, f = d.f0 + fl;
      const size_t s = (size_t)v * d.P + p;
      if (!t.inlier[s]) continue;
      double vr[2 * NV], jp[6];
      point_rows<ND, FISH, ROLL, OPTK>(d, t, v, c, b, p, t.obs[s], vr, jp);
      // pose columns of the local Jacobian, one at a time
      for (int j = 0; j < NPC; ++j) {
        const int k = j / 6, jj = j % 6;
        const int gx = local_to_x(d, f, c, b, j);
        if (gx < 0) continue;
        double col[12];
        view_column(d, t, f, c, b, j, col);
        double a0 = 0.0, a1 = 0.0;
        for (int a = 0; a < DE; ++a) {
          a0 += vr[a] * col[a];
          a1 += vr[NV + a] * col[a];
        }
        double h3[3];
        for (int kk = 0; kk < 3; ++kk) h3[kk] = jp[kk] * a0 + jp[3 + kk] * a1;
        if (k == 0) {
          for (int kk = 0; kk < 3; ++kk) camacc[kk * CW + jj] += h3[kk];
        } else if (k == NPB - 1) {
          for (int kk = 0; kk < 3; ++kk) misc[kk * 6 + jj] += h3[kk];
        } else if (local_is_frame(d, j)) {
          const int dd = j - 6;
          for (int kk = 0; kk < 3; ++kk) Hfs[((size_t)fl * d.DF + dd) * ns + sq + kk] += h3[kk];   // owned by this thread
        } else if (MOTION == MOTION_HAND_EYE) {
          for (int kk = 0; kk < 3; ++kk) misc[27 + kk * 12 + (j - 6)] += h3[kk];
        }
      }
      if constexpr (OPTK) {
        for (int qq = 0; qq < KI; ++qq)
          for (int kk = 0; kk < 3; ++kk) camacc[kk * CW + 6 + qq] += jp[kk] * vr[DE + qq] + jp[3 + kk] * vr[NV + DE + qq];
      }
      // point x point (upper triangle) and gradient
      int e = 18;
      for (int k0 = 0; k0 < 3; ++k0)
        for (int k1 = k0; k1 < 3; ++k1) misc[e++] += jp[k0] * jp[k1] + jp[3 + k0] * jp[3 + k1];
      for (int kk = 0; kk < 3; ++kk) misc[24 + kk] += jp[kk] * vr[NV - 1] + jp[3 + kk] * vr[2 * NV - 1];
    }
    // reduce the camera block over the workgroup (fixed order: lanes by shuffle, then waves 0..3)
    for (int i = 0; i < 3 * CW; ++i) {
      const double w = wave_sum(camacc[i]);
      if (lane == 0) red[wave][i] = w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * CW; i += blockDim.x) {
      const double val = red[0][i] + red[1][i] + red[2][i] + red[3][i];
      const int kk = i / CW, qq = i % CW;
      const int li = qq < 6 ? qq : NPC + (qq - 6);
      const int gx = local_to_x(d, 0, c, b, li);
      if (gx >= 0) {
        const int sx = d.x_to_shared(gx);
        Hss[(size_t)(sq + kk) * ns + sx] = val;
        Hss[(size_t)sx * ns + sq + kk] = val;
      }
    }
    __syncthreads();
  }
  for (int i = 0; i < NMISC; ++i) {
    const double w = wave_sum(misc[i]);
    if (lane == 0) red[wave][i] = w;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NMISC; i += blockDim.x) {
    const double val = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (i < 18) {                                  // board pose block
      const int kk = i / 6, jj = i % 6;
      const int gx = local_to_x(d, 0, 0, b, 6 * (NPB - 1) + jj);
      if (gx >= 0) {
        const int sx = d.x_to_shared(gx);
        Hss[(size_t)(sq + kk) * ns + sx] = val;
        Hss[(size_t)sx * ns + sq + kk] = val;
      }
    } else if (i < 24) {                           // point x point
      const int e = i - 18;
      const int k0 = e < 3 ? 0 : (e < 5 ? 1 : 2), k1 = e < 3 ? e : (e < 5 ? e - 2 : 2);
      Hss[(size_t)(sq + k0) * ns + sq + k1] = val;
      Hss[(size_t)(sq + k1) * ns + sq + k0] = val;
    } else if (i < 27) {
      g[gq + (i - 24)] = val;
    } else {                                       // hand-eye blocks (shared)
      const int kk = (i - 27) / 12, jj = (i - 27) % 12;
      const int gx = local_to_x(d, 0, 0, b, 6 + jj);
      if (gx >= 0) {
        const int sx = d.x_to_shared(gx);
        Hss[(size_t)(sq + kk) * ns + sx] = val;
        Hss[(size_t)sx * ns + sq + kk] = val;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_linearize: fused residual + Jacobian -> per-view local normal equations.
//
//   grid  = one 64-thread workgroup (one wavefront) per view (frame-major), empty views exit at once.
//   compaction: the inlier slots of the view are gathered into an index list in LDS (ballot + popcount), so that
//           lanes and MFMA rows are only spent on real observations (about half of the slots of a visible view).
//   loop  = chunks of 64 observations: lane = observation.  Each lane evaluates the forward model and the 2 x NV row
//           pair V = [E | K | r] in registers, the rows are transposed through an LDS staging buffer
//           (row stride NVP+1 doubles: conflict-free 128-byte row reads, 2-way write conflicts), and
//           S += V^T V is accumulated by v_mfma_f64_16x16x4_f64: operand lane l holds V[4 s + (l>>4)][16 t + (l&15)],
//           A and B operands are the SAME register for diagonal tiles.  NV <= 16 -> one tile, NV <= 32 -> three.
//           Only the MFMA steps that contain observations are issued.
//   epilogue: Y = S That, M = That^T Y (That = [T_cam | T_frame.. | T_board] (+) I) -> packed upper triangle record.
//   LDS: one buffer is time-shared between the row staging and the epilogue matrices (S, Y) to keep >= 2 waves/SIMD.
//   MFMA=false keeps the identical data flow with plain FMAs over the staged rows (validation / fallback build).
// ---------------------------------------------------------------------------------------------------------------

// __launch_bounds__(64, 2): two waves per SIMD = a 256-register budget.  Besides fixing the occupancy the kernel is
// designed for, this makes hipcc select the VGPR form of the MFMA: with the default 512-register budget it keeps the
// loop-carried accumulators in VGPRs, issues AGPR-form MFMAs and brackets EVERY step with 24 v_accvgpr_write +
// 24 v_accvgpr_read and a full-latency s_nop (196 instead of 64 cycles per MFMA, measured with s_memtime stamps).
// FUSED (table-fed fused form, the default): the kernel copies its view's pose entries from the pose table (written by k_prep,
// or by the tail of the k_vec_step that produced the point), forms the chain prefixes and the That columns ITSELF and zeroes
// the accumulation targets of the assembly: no k_tmat launch, no That table (10 MB written + read per evaluation at the
// north-star rig).  The table form (k_tmat writes That / the chains per view) serves adjusted board points and the profiling
// instantiation.  (A third form that evaluated the pose entries from x inside this kernel -- trigonometry on 4 of 64 lanes --
// was measured in rounds 2 and 3, never won and is gone.)
// PROF: the instantiation with s_memtime stamps per phase (mcba_debug_linearize_profile).  The production instantiations
// contain no global store besides the record (a __restrict__ argument): hipcc can then prove that the wave-uniform reads
// of the view / camera tables are never clobbered and issues them as scalar loads (s_load, operands in SGPRs) instead of
// 17 vector loads of one address per 64-observation chunk.
template <int ND, int FISH, int MOTION, bool OPTK, bool MFMA, bool ROBUST, int FUSED_MODE, bool PROF = false>
// (static pinhole kernels with the linear loss fit 128 registers: they ask for four waves per SIMD explicitly, so that the
//  table-fed fused form -- 132 registers under the two-wave budget -- is allocated into 128 as well)
__global__ __launch_bounds__(64, (MOTION == MOTION_STATIC && !FISH && MFMA && !ROBUST && !PROF && OPTK && ND <= 5) ? 4 : 2)
void k_linearize(Dims d, Tables t, double* __restrict__ rec,
                                                     const uint16_t* __restrict__ tri, int epoch,
                                                     const double* __restrict__ x, double* __restrict__ zero_a, int na,
                                                     double* __restrict__ zero_b, int nb, LsmrCompact cp) {
  // FUSED_MODE: 0 = table form (That / chains from k_tmat), 2 = table-fed fused form (pose entries copied from the pose table,
  // intrinsics from the camera table), 3 = the table-fed fused form over the COMPACTED observation tables (LsmrCompact: observed point
  // and board point of every inlier in residual order + one descriptor per active view, built once per inlier set): no mask bytes, no
  // compaction list, no point-index gathers -- the first chunk of a view is requested with its pose entries (one round trip per view
  // instead of two); same lane for every observation, so the records are bit-identical to form 2
  static_assert(FUSED_MODE == 0 || FUSED_MODE == 2 || FUSED_MODE == 3, "linearisation forms: 0 = table form, 2 / 3 = table-fed fused forms");
  constexpr bool FUSED = FUSED_MODE != 0;
  constexpr bool COMPACT = FUSED_MODE == 3;
  (void)x; (void)cp;
  constexpr bool ROLL = MOTION == MOTION_ROLLING;
  constexpr int DE = ROLL ? 12 : 6, NPB = MOTION == MOTION_STATIC ? 3 : 4, KI = OPTK ? 4 + ND : 0;
  constexpr int NV = DE + KI + 1, NT = (NV + 15) / 16, NVP = 16 * NT;
  // shifted-tile kernels (see TAILV below) never read a column >= NV: their staging rows and their epilogue copy of S are
  // NV wide instead of 32 (LDS per workgroup 20.4 -> 16.1 KB at NV = 22)
  constexpr bool NARROW = MFMA && NT == 2 && NV - 16 <= 7;
  constexpr int LDV = (NARROW ? NV : NVP) + 1;
  constexpr int SLD = NARROW ? ((NV + 1) & ~1) : NVP, SROWS = NARROW ? NV : NVP;   // epilogue S: SROWS x SLD
  constexpr int NPC = 6 * NPB, NL = NPC + KI, N1 = NL + 1;
  // staging: one LDS row per lane; a chunk of 64 observations is accumulated in TWO rounds, first the u-rows of all 64
  // lanes, then the v-rows (S = sum of the outer products of all rows: the order is free).  Compared with staging the
  // row pairs of 32 lanes per round this issues half as many ds_write instructions (every lane is active in every store)
  // and the registers of a row die as soon as it is staged.
  constexpr int ROWS = 64;
  constexpr int REC = N1 * (N1 + 1) / 2;
  // NV in 17..23 (rolling shutter + intrinsics: 22): ONE shifted MFMA tile per step instead of three.  With A = columns
  // [0, 16) and B = columns [TW, TW + 16), TW = NV - 16, the 16 x 16 product covers S[0:16][TW:NV], i.e. every pair
  // {i, j} except the pairs inside the first TW and inside the last TW columns.  Those 2 * TW (TW + 1) / 2 (= 42) products
  // per row are accumulated per lane on the VALU (same FP64 pipe and rate as the MFMA on gfx950, but only the unique
  // products) and reduced across lanes once per view.  211 of the 256 products of the tile are distinct entries of the
  // symmetric S (the tiles (0,0) + (0,1) of the straightforward blocking delivered 232 distinct entries for 512 products).
  constexpr int TW = NV - 16;
  constexpr bool TAILV = MFMA && NT == 2 && TW <= 7;
  constexpr int NTAIL = TAILV ? TW * (TW + 1) / 2 : 1;
  // epilogue footprint: S (NVP^2), Y (DE x 16 ceil(NPC/16)) and the packed record
  constexpr int STAGE = ROWS * LDV;
  constexpr int EPI = SLD * SROWS + (MFMA ? DE * 16 * ((NPC + 15) / 16) + (REC + 2 + 1) / 2 * 2 + 64 : 0);
  constexpr int BUF = STAGE > EPI ? STAGE : EPI;

  __shared__ __attribute__((aligned(16))) double Buf[BUF];   // staging rows in the main loop; [S | Y | M] in the epilogue
  // PIPE: the "front" of the NEXT view (mask bytes, That, chain matrices and camera parameters: two dependent memory round
  // trips and the compaction) is loaded behind the main loop of the current view and finished in the middle of its
  // epilogue, into the second copy of the small per-view LDS tables, so that the next main loop starts on resident data.
  // Measured and NOT adopted (the switch stays for the record): 50.2 us against 48.3 us at cfg3, 92 against 68 us at cfg4.
  // The front registers are live across the epilogue (static kernels: 124 -> 160 VGPRs, one wave per SIMD less), and the
  // second wavefront of the SIMD already covers the prologue's round trips.
#if defined(MCBA_EXP_PIPE)
  constexpr bool PIPE = MFMA && !FUSED && !PROF;
#else
  constexpr bool PIPE = false;
#endif
  constexpr int NBUFS = PIPE ? 2 : 1;
  constexpr int NVS = (ROLL ? 2 : 1) * VIEW_STRIDE;
  __shared__ double TmBuf[NBUFS * DE * NPC];
  __shared__ uint16_t PidxBuf[NBUFS * LIN_MAX_POINTS];
  // chain matrices of the view in LDS: written by the fused prologue; the rolling-shutter kernels (two chains = 48 scalar
  // registers, more than the SGPR file has left) also read them from here in the main loop
  constexpr bool VLDS = FUSED || ROLL;
  __shared__ double VmBuf[VLDS ? NBUFS * NVS : 1];
  __shared__ double Eye3[9];                 // 3 x 3 identity for the address selects of that_column_sel
  if (threadIdx.x < 9) Eye3[threadIdx.x] = (threadIdx.x & 3) == 0 ? 1.0 : 0.0;
  double* Vbuf = Buf;
  if constexpr (FUSED) {   // the assembly that follows accumulates into [g | diag | cost] and H_ss
    for (int e = blockIdx.x * 64 + threadIdx.x; e < na; e += gridDim.x * 64) zero_a[e] = 0.0;
    for (int e = blockIdx.x * 64 + threadIdx.x; e < nb; e += gridDim.x * 64) zero_b[e] = 0.0;
  }

  // persistent wavefronts: the grid holds about as many workgroups as the chip keeps resident and each one walks the
  // compact list of non-empty views (k_active_views).  One workgroup per view was DISPATCH-bound: ~42 cycles per
  // launched workgroup x 8000 views = the whole kernel time, with the SIMD slots idle more than half of it.
  const int lane = threadIdx.x;
  // (a dynamic hand-out of the views through an atomic counter was measured slower at every size: static stride it is)
  const int n_active = t.active_views[0];
  (void)epoch;   // (de-phasing the workgroups with a start-up delay per blockIdx & 3 was measured: only slower)
  constexpr int NPB64 = LIN_MAX_POINTS / 64;
  constexpr int NTL = (DE * NPC + 63) / 64;
  // ---- front of a view: registers filled by front_issue, consumed by front_finish -----------------------------------
  uint8_t inb[NPB64];
  double tl[NTL], vm_f = 0.0, cam_f[5 + ND], ext_f[2];
  double camr[5 + ND], extr[CAM_STRIDE - CAM_TILT], Vr[ROLL ? 1 : NVS];
#pragma unroll
  for (int k = 0; k < CAM_STRIDE - CAM_TILT; ++k) extr[k] = 0.0;
  // all global loads of the front are issued back to back (mask bytes, That, chain matrices, camera): one round trip
  auto front_issue = [&](int vv, int pl) {
    const int cc = (vv / d.B) % d.C;
    if constexpr (!COMPACT) {
      const uint8_t* mrow = t.inlier + (size_t)vv * d.P;
#pragma unroll
      for (int k = 0; k < NPB64; ++k) inb[k] = masked_load_row(mrow, k * 64 + pl, d.P);
    }
    if constexpr (!FUSED) {
      const double* tg = t.tmat + (size_t)vv * (DE * NPC);   // That of this view, precomputed by k_tmat
#pragma unroll
      for (int k = 0; k < NTL; ++k) tl[k] = masked_load_row(tg, k * 64 + pl, DE * NPC);
      const double* vsrc = t.view + (size_t)vv * NVS;
      if constexpr (ROLL) {
        vm_f = vsrc[min(pl, NVS - 1)];
      } else {
#pragma unroll
        for (int k = 0; k < NVS; ++k) Vr[k] = vsrc[k];         // (made scalar in front_finish)
      }
    }
    {   // camera entry from the table: same round trip
      const double* csrc = t.cam + (size_t)cc * CAM_STRIDE;
#pragma unroll
      for (int k = 0; k < 5 + ND; ++k) cam_f[k] = csrc[k];
      ext_f[0] = csrc[CAM_HEIGHT];
      ext_f[1] = csrc[CAM_FIXASPECT];
    }
  };
  // That / chain matrices / compaction list into LDS copy `buf`; camera parameters and (static) chain matrices become
  // scalars.  Returns the number of inliers of the view.
  auto front_finish = [&](int buf, int pl) {
    double* Tm = TmBuf + buf * (DE * NPC);
    uint16_t* pidx = PidxBuf + buf * LIN_MAX_POINTS;
    if constexpr (!FUSED) {
#pragma unroll
      for (int k = 0; k < NTL; ++k)
        if (k * 64 + pl < DE * NPC) Tm[k * 64 + pl] = tl[k];
      if constexpr (ROLL) {
        if (pl < NVS) VmBuf[buf * NVS + pl] = vm_f;
      } else {
#pragma unroll
        for (int k = 0; k < NVS; ++k) Vr[k] = uniform_f64(Vr[k]);
      }
    }
    {
#pragma unroll
      for (int k = 0; k < 5 + ND; ++k) camr[k] = uniform_f64(cam_f[k]);
      extr[CAM_HEIGHT - CAM_TILT] = uniform_f64(ext_f[0]);
      extr[CAM_FIXASPECT - CAM_TILT] = uniform_f64(ext_f[1]);
    }
    int cnt = 0;
    if constexpr (!COMPACT) {
#pragma unroll
      for (int k = 0; k < NPB64; ++k) {
        const bool in = inb[k] != 0;
        const unsigned long long m = __ballot(in);
        if (in) pidx[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(k * 64 + lane);
        cnt += __popcll(m);
      }
    }
    return cnt;
  };

  int cur = 0;                 // LDS copy that holds the front of the current view
  int count = 0, p_cur = 0;
  double2 ob_cur;
  ob_cur.x = ob_cur.y = 0.0;
  double X_cur[3] = {0.0, 0.0, 0.0}, X_nxt[3];
  bool front_ready = false;    // the front of the view about to be processed is already in LDS copy `cur` (PIPE)
  int4 dnext = make_int4(0, 0, 0, 0);   // (COMPACT) descriptor {view, first observation, inliers} of the next view of this workgroup
  if constexpr (COMPACT) {
    if ((int)blockIdx.x < n_active) dnext = cp.desc[blockIdx.x];
  }
  for (int vi = blockIdx.x; vi < n_active; vi += gridDim.x) {
  int v_, desc_first = 0, desc_count = 0;
  if constexpr (COMPACT) {
    v_ = dnext.x; desc_first = dnext.y; desc_count = dnext.z;
    if (vi + (int)gridDim.x < n_active) dnext = cp.desc[vi + gridDim.x];   // (scalar load, one view ahead)
  } else {
    v_ = t.active_views[1 + vi];
  }
  const int v = v_;
  (void)desc_first; (void)desc_count;
  if (v < 0) continue;         // (padding of a hand-made list: mcba_debug_set_frame_groups)
  const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
  (void)f;
  // (PIPE) the view after this one: its index is needed early, it heads the next front's dependent chain of loads
  const int vi_n = vi + (int)gridDim.x;
  const bool has_next = PIPE && vi_n < n_active;
  const int v_n = has_next ? t.active_views[1 + vi_n] : v;
  long long stamp[6] = {0, 0, 0, 0, 0, 0};
  constexpr bool prof = PROF;
  if (prof) stamp[0] = clock64();
  // Opaque copy of the lane id: the prologue offsets depend on the lane only, so hipcc hoists their 64-bit forms out of
  // the view loop and SPILLS them (256 registers are taken) -- every load was then preceded by a scratch reload and a
  // vmcnt(0) wait.  Recomputing a 32-bit offset per view is two integer instructions.
  int pl;
  asm volatile("v_mov_b32 %0, %1" : "=v"(pl) : "v"(lane));
  double* Tm = TmBuf + cur * (DE * NPC);
  uint16_t* pidx = PidxBuf + cur * LIN_MAX_POINTS;
  double* Vm = VmBuf + (VLDS ? cur * NVS : 0);

  if (!front_ready) front_issue(v, pl);
  const double* camp = nullptr;   // FUSED: the camera's parameter block [fx fy cx cy skew k...] inside x (or the constants)
  if constexpr (FUSED) {
    // x == nullptr: the TABLE-FED fused form -- the pose / camera tables already hold the point (k_prep, or the tail of the
    // k_vec_step that produced it): the view's pose entries are copied from the pose table (no trigonometry here) and the
    // intrinsics come from the camera table; the chain products and the That columns are formed below all the same.
    camp = t.cam + (size_t)c * CAM_STRIDE;
    (void)camp;
    // pose entries of the view: entry 0 camera, entry NPB - 1 board, between them the motion poses
    static_assert(NPB * POSE_STRIDE <= BUF, "pose entries do not fit the staging buffer");
    double* Pl = Buf;                      // (the staging buffer is free until the first chunk)
    {
      // the pose entries are requested with the front loads (same round trip as the mask bytes); then the masks are
      // compacted and the FIRST CHUNK's observations / board points are requested, and only then the chain products and That
      // columns are formed -- under that round trip instead of in front of it
      constexpr int NPE = (NPB * POSE_STRIDE + 63) / 64;
      double pe_f[NPE];
#pragma unroll
      for (int u = 0; u < NPE; ++u) {
        const int e = min(pl + 64 * u, NPB * POSE_STRIDE - 1), k = e / POSE_STRIDE, q = e - k * POSE_STRIDE;
        const int gi = k == 0 ? d.pose_cam + c
                     : (k == NPB - 1 ? d.pose_board + b
                                     : d.pose_motion + (MOTION == MOTION_STATIC ? f : (MOTION == MOTION_ROLLING ? (k - 1) * d.F + f : k - 1)));
        pe_f[u] = t.pose[(size_t)gi * POSE_STRIDE + q];
      }
      if constexpr (COMPACT) {   // the first chunk rides in the same round trip as the pose entries and the camera
        const size_t g0 = (size_t)desc_first + (size_t)(lane < desc_count ? lane : 0);
        ob_cur = cp.obs[g0];
        const double2 xy = cp.bxy[g0];
        X_cur[0] = xy.x; X_cur[1] = xy.y; X_cur[2] = cp.bz[g0];
        front_finish(cur, pl);
        count = desc_count;
      } else {
        count = front_finish(cur, pl);
        lds_fence();
        p_cur = lane < count ? pidx[lane] : 0;
        ob_cur = t.obs[(size_t)v * d.P + p_cur];
        for (int k = 0; k < 3; ++k) X_cur[k] = t.board_points[3 * (size_t)(b * d.P + p_cur) + k];
      }
#pragma unroll
      for (int u = 0; u < NPE; ++u) Pl[min(pl + 64 * u, NPB * POSE_STRIDE - 1)] = pe_f[u];
    }
    lds_fence();
    const double* Pc = Pl;
    const double* Pb = Pl + (NPB - 1) * POSE_STRIDE;
    const double* Pm0 = Pl + POSE_STRIDE;
    const double* Pm1 = Pl + (NPB > 3 ? 2 : 1) * POSE_STRIDE;
    const double* Bf = t.bwg + 12 * (size_t)f;
    if constexpr (MOTION != MOTION_HAND_EYE) {
      // static / rolling shutter: the chain prefixes  R1 | t1 = camera . frame  and  R2 | o = camera . frame . board  are formed
      // ONCE per view, one entry per lane in two dependent steps through LDS (48 lanes x 3 FMAs each for the two chains of a
      // rolling-shutter view), then lane j < 6 NPB forms column j of That from the prefixes -- the construction k_tmat uses.
      // (fused_view_tables let EVERY lane form every product: ~400 FP64 instructions per view against ~60 here; the
      //  table-fed fused kernel took 50.5 us against 44.3 us of the table form at the north-star rig.)
      static_assert(NPB * POSE_STRIDE + 2 * PRE_STRIDE <= BUF, "chain prefixes do not fit the staging buffer");
      constexpr int NCH = ROLL ? 2 : 1;
      double* pre = Buf + NPB * POSE_STRIDE;
      const int pch = pl / 12, pq = pl - 12 * pch;
      const bool pon = pl < 12 * NCH;
      const int pcc = pon ? pch : 0;
      if (pon) pre[pcc * PRE_STRIDE + pq] = se3_mul_entry(Pc, pcc == 0 ? Pm0 : Pm1, pq);
      lds_fence();
      if (pon) {
        const double val = se3_mul_entry(pre + pcc * PRE_STRIDE, Pb, pq);
        pre[pcc * PRE_STRIDE + 12 + pq] = val;
        Vm[pcc * VIEW_STRIDE + pq] = val;   // the chain matrix board -> camera of this chain
      }
      lds_fence();
      if (pl < NPC) that_column_sel<ROLL>(Pc, Pm0, Pm1, Pb, pre, Eye3, pl, Tm, NPC);
    } else if (pl < NPC) {                 // hand-eye (five-pose chain): one column of That per lane
      double col[DE];
      view_column_p(d, Pc, Pb, Pm0, Pm1, Bf, pl, col);
#pragma unroll
      for (int a = 0; a < DE; ++a) Tm[a * NPC + pl] = col[a];
    } else if (pl < NPC + (ROLL ? 2 : 1)) {   // the chain matrices board -> camera (start / end pose)
      const int ch = pl - NPC;
      double Vc[VIEW_STRIDE];
      view_chain_p(d, Pc, Pb, (ROLL && ch == 1) ? Pm1 : Pm0, Pm1, Bf, Vc);
#pragma unroll
      for (int k = 0; k < VIEW_STRIDE; ++k) Vm[ch * VIEW_STRIDE + k] = Vc[k];
    }
    lds_fence();                           // Pl is dead: the staging buffer may be written again
  }
  if (prof) stamp[4] = clock64();
  // only the pad columns need clearing: every staged row is fully rewritten (columns < NV) in every round
  // (the shifted tile reads columns < NV only: nothing to clear)
  if constexpr (!TAILV)
    for (int e = lane; e < ROWS * (LDV - NV); e += 64) Vbuf[(e / (LDV - NV)) * LDV + NV + e % (LDV - NV)] = 0.0;

  if (prof) stamp[5] = clock64();
  if constexpr (!FUSED) {   // (the table-fed fused form has compacted its masks above)
    if (!front_ready) count = front_finish(cur, pl);
  }

  constexpr int NACC_V = NVP * NVP / 64;                  // plain-FMA variant: NVP*NVP entries over 64 lanes
  constexpr int NTILE = TAILV ? 2 : NT * (NT + 1) / 2;   // shifted tile: two accumulators (even / odd steps), no dependent MFMAs
  double accv[MFMA ? 1 : NACC_V];
  double4_t accm[MFMA ? NTILE : 1];
  if constexpr (MFMA) {
    for (int i = 0; i < NTILE; ++i) accm[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  } else {
    for (int i = 0; i < NACC_V; ++i) accv[i] = 0.0;
  }
  double cost = 0.0;
  double tail[NTAIL], head[NTAIL];
  for (int i = 0; i < NTAIL; ++i) tail[i] = head[i] = 0.0;
  lds_fence();
  if (prof) stamp[1] = clock64();

  // The view's chain matrices and the camera's parameters are read ONCE per view (front) and kept as scalars (SGPRs;
  // rolling shutter: the two chains stay in LDS).  Left to the compiler they were 17 vector loads of one address each in
  // EVERY chunk, with 68 vector registers to hold them.
  if constexpr (FUSED) {
    if constexpr (!ROLL) {
#pragma unroll
      for (int k = 0; k < NVS; ++k) Vr[k] = uniform_f64(Vm[k]);
    }
  }
  // (the tilted model reads its 27 tilt-matrix entries from the table: rare, not worth the scalar registers)
  const double* extp = ND >= 14 ? t.cam + (size_t)c * CAM_STRIDE + CAM_TILT : extr;

  // software prefetch: observation + board point of the NEXT chunk are requested before the current one is processed
  // (the first chunk of a pipelined view was requested from the middle of the previous epilogue)
  if (!FUSED && !front_ready) {   // (table-fed fused form: requested before the chain arithmetic)
    p_cur = lane < count ? pidx[lane] : 0;
    ob_cur = t.obs[(size_t)v * d.P + p_cur];
    for (int k = 0; k < 3; ++k) X_cur[k] = t.board_points[3 * (size_t)(b * d.P + p_cur) + k];
  }
  int count_n = 0;             // (PIPE) inliers of the next view, known once its front is finished
  int count_total = count;     // inliers of the whole view (a board with > LIN_MAX_POINTS points is walked in segments)
  for (int seg0 = 0;;) {
  // one chunk of 64 observations.  FULL = every lane holds an observation: no predicate, no zero state for idle lanes (the
  // merge of `in ? state : 0` cost 29 moves + a dozen selects per chunk); only the last chunk of a view takes the general form
  auto do_chunk = [&](int base, auto full_tag) {
    constexpr bool FULLC = decltype(full_tag)::value;
    const int i = base + lane;
    const bool in = FULLC || i < count;
    const int inx = i + 64;
    int p_nxt = 0;
    double2 ob_nxt;
    if constexpr (COMPACT) {
      const size_t gn = (size_t)desc_first + (size_t)(inx < count ? inx : 0);
      ob_nxt = cp.obs[gn];
      const double2 xy = cp.bxy[gn];
      X_nxt[0] = xy.x; X_nxt[1] = xy.y; X_nxt[2] = cp.bz[gn];
    } else {
      p_nxt = inx < count ? pidx[inx] : p_cur;
      ob_nxt = t.obs[(size_t)v * d.P + p_nxt];
      for (int k = 0; k < 3; ++k) X_nxt[k] = t.board_points[3 * (size_t)(b * d.P + p_nxt) + k];
    }
    PointState<ND, ROLL> ps;
    long long t0 = 0;
    if (prof) t0 = clock64();
    // rolling shutter: the two chain matrices of the view are read from LDS into registers HERE, as LDS reads (handed to
    // point_state as a pointer that may also be the global view table they became twelve flat loads per chunk)
    double Vl[ROLL ? NVS : 1];
    if constexpr (ROLL) {
#pragma unroll
      for (int k = 0; k < NVS; ++k) Vl[k] = Vm[k];
    }
    if (in) {
#if defined(MCBA_EXP_NO_SCALAR_TABLES)   // A/B switch of the profiling builds: table reads left to the compiler
      cost += point_state<ND, FISH, ROLL, ROBUST>(d, t, v, c, b, p_cur, ob_cur, ps, X_cur, FUSED ? Vm : nullptr, camp);
#else
      // (the precomputed scan time of the compacted tables -- LsmrCompact::tr, read by the LSMR product kernel -- was measured here too:
      //  no change of the step, 6 MB more traffic per launch: the division stays)
      cost += point_state<ND, FISH, ROLL, ROBUST>(d, t, v, c, b, p_cur, ob_cur, ps, X_cur, ROLL ? Vl : Vr, camr, extp);
#endif
    } else {   // lanes past the end of the list stage zero rows
      ps = PointState<ND, ROLL>{};
    }
    p_cur = p_nxt;
    ob_cur = ob_nxt;
    for (int k = 0; k < 3; ++k) X_cur[k] = X_nxt[k];
    if (prof) stamp[2] += clock64() - t0;
    const int nchunk = min(64, count - base);              // observations in this chunk
    const int nsteps = (nchunk + 3) / 4;                   // 4 rows (4 observations, one image axis) per MFMA step
#pragma unroll
    for (int q = 0; q < 2; ++q) {                          // q = 0: u-rows, q = 1: v-rows
      double row[NV];
      point_row<ND, ROLL, OPTK>(ps, q, row);
      if constexpr (TAILV) {                               // the two VALU blocks of this row (zero rows add zeros)
        int e = 0;
#pragma unroll
        for (int i0 = 0; i0 < TW; ++i0)
#pragma unroll
          for (int i1 = i0; i1 < TW; ++i1, ++e) {
            head[e] += row[i0] * row[i1];
            tail[e] += row[16 + i0] * row[16 + i1];
          }
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) Vbuf[lane * LDV + k] = row[k];
      lds_fence();
      if constexpr (MFMA) {
        const int rsub = lane >> 4, csub = lane & 15;
        const double* vp = Vbuf + rsub * LDV + csub;
        // two steps per iteration with ping-pong operand registers: the LDS reads of the next step are in flight
        // while the MFMAs of the current one issue, and no register copy forces an early s_waitcnt
        // (every lane stages its row, a zero row for lanes without an observation, so the step count can be rounded
        //  up to an even number and the loop body stays branch-free)
        // operand column offsets: plain blocking = tile t at column 16 t; shifted tile = A at column 0, B at column TW
        constexpr int COFF1 = TAILV ? TW : 16;
        double a0[NT], a1[NT];
        constexpr int MAXS = ROWS / 4;
        const int nsteps2 = (nsteps + 1) & ~1;
        for (int tt = 0; tt < NT; ++tt) a0[tt] = vp[COFF1 * tt];
        for (int st = 0; st < nsteps2; st += 2) {
          for (int tt = 0; tt < NT; ++tt) a1[tt] = vp[(4 * (st + 1)) * LDV + COFF1 * tt];
          if constexpr (TAILV) {
            accm[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], a0[1], accm[0], 0, 0, 0);
          } else {
            int ti = 0;
            for (int t0 = 0; t0 < NT; ++t0)
              for (int t1 = t0; t1 < NT; ++t1, ++ti)
                accm[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t0], a0[t1], accm[ti], 0, 0, 0);
          }
          const int nx = min(st + 2, MAXS - 1);
          for (int tt = 0; tt < NT; ++tt) a0[tt] = vp[(4 * nx) * LDV + COFF1 * tt];
          if constexpr (TAILV) {
            accm[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], a1[1], accm[1], 0, 0, 0);
          } else {
            int ti = 0;
            for (int t0 = 0; t0 < NT; ++t0)
              for (int t1 = t0; t1 < NT; ++t1, ++ti)
                accm[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t0], a1[t1], accm[ti], 0, 0, 0);
          }
        }
      } else {
        constexpr int IW = NVP, JW = NACC_V;
        const int ii = lane % IW, j0 = (lane / IW) * JW;
        for (int row = 0; row < 4 * nsteps; ++row) {
          const double vi = Vbuf[row * LDV + ii];
          for (int jj = 0; jj < JW; ++jj) accv[jj] += vi * Vbuf[row * LDV + j0 + jj];
        }
      }
      lds_fence();
    }
  };
  {
    int base = 0;
    for (; base + 64 <= count; base += 64) do_chunk(base, std::true_type{});
    if (base < count) do_chunk(base, std::false_type{});
  }
  if constexpr (COMPACT) break;   // (the compacted run of a view is walked whole: no segments)
  seg0 += LIN_MAX_POINTS;
  if (seg0 >= d.P) break;
  {   // next segment of a large board: its mask bytes are compacted into the (now free) list, the accumulators carry on
    const uint8_t* mrow = t.inlier + (size_t)v * d.P;
#pragma unroll
    for (int k = 0; k < NPB64; ++k) inb[k] = masked_load_row(mrow, seg0 + k * 64 + pl, d.P);
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < NPB64; ++k) {
      const bool in = inb[k] != 0;
      const unsigned long long m = __ballot(in);
      if (in) pidx[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(seg0 + k * 64 + lane);
      cnt += __popcll(m);
    }
    lds_fence();
    count = cnt;
    count_total += cnt;
    p_cur = lane < count ? pidx[lane] : 0;
    ob_cur = t.obs[(size_t)v * d.P + p_cur];
    for (int k = 0; k < 3; ++k) X_cur[k] = t.board_points[3 * (size_t)(b * d.P + p_cur) + k];
  }
  }

  if (prof) stamp[3] = clock64();
  if (has_next) front_issue(v_n, pl);   // (PIPE) the next view's front is in flight during the epilogue
#if defined(MCBA_EXP_PRIO)
  __builtin_amdgcn_s_setprio(MCBA_EXP_PRIO);   // latency-bound epilogue: ask for the issue slots, the partner wave keeps the pipe busy
#endif
  // The epilogue indices only depend on the lane, so the compiler hoists them out of the view loop and then SPILLS them
  // (the row evaluation needs the whole register budget): every store was preceded by a scratch reload and an
  // s_waitcnt vmcnt(0), i.e. it waited for the previous global store to retire.  An opaque copy of the lane id keeps
  // the index arithmetic (a handful of integer ops) inside the epilogue.
  int el;
  asm volatile("v_mov_b32 %0, %1" : "=v"(el) : "v"(lane));
  // cross-lane reduction of the two VALU blocks (first TW and last TW columns): the 2 NTAIL values per lane are folded in
  // registers, two values per fold32 (64 -> 32 partials each) and two of those per fold16 (-> 16 partials), so that a
  // quarter of the vectors goes through LDS: row g of vector j carries 16 partials of value 4 j + {0, 2, 1, 3}[g].  Lane
  // e < 2 NTAIL then adds the 16 partials of value e (all reads issued before the first add).  Value e < NTAIL belongs to
  // the first block, the others to the last block.
  double corner_red = 0.0;
  if constexpr (TAILV) {
    constexpr int NC = 2 * NTAIL, NZ = (NC + 1) / 2, NW = (NZ + 1) / 2;
    static_assert(NC <= 64 && 4 * NW * 17 <= BUF, "corner transpose does not fit");
    double cv[2 * NZ], z[2 * NW];
#pragma unroll
    for (int e = 0; e < 2 * NZ; ++e) cv[e] = e < NTAIL ? head[e < NTAIL ? e : 0] : (e < NC ? tail[e < NC ? e - NTAIL : 0] : 0.0);
#pragma unroll
    for (int i = 0; i < 2 * NW; ++i) z[i] = i < NZ ? fold32(cv[2 * (i < NZ ? i : 0)], cv[2 * (i < NZ ? i : 0) + 1]) : 0.0;
    const int grow = (el >> 4) & 3, gperm = ((grow & 1) << 1) | (grow >> 1);   // row -> value offset {0, 2, 1, 3}
#pragma unroll
    for (int j = 0; j < NW; ++j) Buf[(4 * j + gperm) * 17 + (el & 15)] = fold16(z[2 * j], z[2 * j + 1]);
    lds_fence();
    {
      const int e = el < NC ? el : 0;
      double vals[16];
      const double* row = Buf + e * 17;
#pragma unroll
      for (int k = 0; k < 16; ++k) vals[k] = row[k];
#pragma unroll
      for (int w = 1; w < 16; w *= 2)
#pragma unroll
        for (int k = 0; k + w < 16; k += 2 * w) vals[k] += vals[k + w];
      corner_red = vals[0];
    }
    lds_fence();
  }
  // S -> LDS (full symmetric matrix), time-sharing the staging buffer
  double* Sbuf = Buf;
  if constexpr (MFMA) {
    const int rsub = el >> 4, csub = el & 15;
    if constexpr (TAILV) {
      // shifted tile: accumulator entry (row, csub) is S[row][TW + csub]; entries below the diagonal duplicate their
      // mirror image bit for bit (same products, same order), so both lanes store the same value
      for (int r = 0; r < 4; ++r) {
        const int row = rsub + 4 * r, col = TW + csub;
        const double val = accm[0][r] + accm[1][r];
        Sbuf[row * SLD + col] = val;
        Sbuf[col * SLD + row] = val;
      }
    } else {
      int ti = 0;
      for (int t0 = 0; t0 < NT; ++t0)
        for (int t1 = t0; t1 < NT; ++t1, ++ti) {
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * t0 + rsub + 4 * r, col = 16 * t1 + csub;
            Sbuf[row * SLD + col] = accm[ti][r];
            if (t0 != t1) Sbuf[col * SLD + row] = accm[ti][r];
          }
        }
    }
  } else {
    constexpr int IW = NVP, JW = NACC_V;
    const int ii = el % IW, j0 = (el / IW) * JW;
    for (int jj = 0; jj < JW; ++jj) Sbuf[ii * SLD + j0 + jj] = accv[jj];
  }
  if constexpr (TAILV) {   // lane e carries packed entry e = (i0 <= i1) of the first (e < NTAIL) or of the last block
    if (el < 2 * NTAIL) {
      const int blk = el < NTAIL ? 0 : 1, e = el - blk * NTAIL;
      int i0 = 0, rem = e;
      while (rem >= TW - i0) { rem -= TW - i0; ++i0; }
      const int i1 = i0 + rem, o = 16 * blk;
      Sbuf[(o + i0) * SLD + o + i1] = corner_red;
      Sbuf[(o + i1) * SLD + o + i0] = corner_red;
    }
  }
  lds_fence();

  double* out = rec + (size_t)v * d.rec_stride;
  long long ep0 = 0, ep1 = 0;
  if (prof) ep0 = clock64();
  if constexpr (MFMA) {
    // epilogue on the matrix pipe:  Y = S_EE That,  M_pp = That^T Y,  M_pK = That^T S_E,[K r]   (16x16x4 tiles, K = DE)
    // M is assembled in LDS in the packed record order and leaves the CU with fully coalesced 16-byte stores.
    constexpr int NPCT = (NPC + 15) / 16, NPCP = 16 * NPCT, KR = KI + 1, KRT = (KR + 15) / 16, KS = (DE + 3) / 4;
    constexpr int RECP = (REC + 2 + 1) / 2 * 2;                      // == d.rec_stride
    static_assert(SLD * SROWS + DE * NPCP + RECP + 64 <= BUF, "Y, the packed record and the cost slots do not fit behind S");
    double* Yb = Buf + SLD * SROWS;                                   // [DE][NPCP]
    double* Mp = Yb + DE * NPCP;                                    // packed upper triangle + cost, count
    double* Cb = Mp + RECP;                                         // [64] per-lane costs
    // (& 3: el is opaque to the optimiser, the mask tells it that 4 ks + rsub < 4 KS)
    const int rsub = (el >> 4) & 3, csub = el & 15;
    // LDS operand loads are written as unconditional reads of a clamped index times a 0/1 mask and issued in batches
    // ahead of the MFMAs that use them: `cond ? lds[i] : 0` compiles to a branch + ds_read + lgkmcnt(0) per MFMA, i.e. one
    // LDS round trip (~130 cycles) for each of the 21 matrix instructions.
    auto lmask = [](const double* p, int i, bool ok) { return p[ok ? i : 0] * (ok ? 1.0 : 0.0); };
    Cb[el] = cost;
    {                                                               // step 1: Y = S_EE That
      double a1[KS], b1[NPCT][KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a1[ks] = Sbuf[csub * SLD + 4 * ks + rsub];
#pragma unroll
      for (int tj = 0; tj < NPCT; ++tj)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int k = 4 * ks + rsub, jc = 16 * tj + csub;
          b1[tj][ks] = lmask(Tm, k * NPC + jc, k < DE && jc < NPC);
        }
#pragma unroll
      for (int tj = 0; tj < NPCT; ++tj) {
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ks], b1[tj][ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rsub + 4 * r;
          if (row < DE) Yb[row * NPCP + 16 * tj + csub] = acc[r];
        }
      }
    }
    lds_fence();
    if (prof) ep1 = clock64();
    if (has_next) {
      // (PIPE) the front of the next view has landed: finish it into the other LDS copy and request its first chunk, so
      // that the remaining half of this epilogue hides that round trip too
      count_n = front_finish(cur ^ 1, el);
      const uint16_t* pn = PidxBuf + (cur ^ 1) * LIN_MAX_POINTS;
      const int b_n = v_n % d.B;
      p_cur = el < count_n ? pn[el] : 0;
      ob_cur = t.obs[(size_t)v_n * d.P + p_cur];
      for (int k = 0; k < 3; ++k) X_cur[k] = t.board_points[3 * (size_t)(b_n * d.P + p_cur) + k];
    }
#pragma unroll
    for (int ti = 0; ti < NPCT; ++ti) {
      double av[KS], b2[NPCT][KS], b3[KRT][KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + rsub, ic = 16 * ti + csub;
        av[ks] = lmask(Tm, k * NPC + ic, k < DE && ic < NPC);
#pragma unroll
        for (int tj = ti; tj < NPCT; ++tj) b2[tj][ks] = lmask(Yb, k * NPCP + 16 * tj + csub, k < DE);
#pragma unroll
        for (int tk = 0; tk < KRT; ++tk) {
          const int jc = 16 * tk + csub;
          b3[tk][ks] = lmask(Sbuf, k * SLD + DE + jc, k < DE && jc < KR);
        }
      }
      int rowoff[4];                                                // packed offset of row i:  i (2 N1 - 1 - i) / 2
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + rsub + 4 * r;
        rowoff[r] = (i * (2 * N1 - 1 - i)) / 2;
      }
#pragma unroll
      for (int tj = ti; tj < NPCT; ++tj) {                          // step 2: pose x pose
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], b2[tj][ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * ti + rsub + 4 * r, j = 16 * tj + csub;
          if (i <= j && j < NPC) Mp[rowoff[r] + j] = acc[r];
        }
      }
#pragma unroll
      for (int tk = 0; tk < KRT; ++tk) {                            // step 3: pose x (intrinsics | residual)
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], b3[tk][ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * ti + rsub + 4 * r, jc = 16 * tk + csub;
          if (i < NPC && jc < KR) Mp[rowoff[r] + NPC + jc] = acc[r];
        }
      }
    }
    for (int e = el; e < KR * KR; e += 64) {                        // step 4: (intrinsics | residual)^2 block = copy of S
      const int i = e / KR, j = e % KR;
      if (i <= j) Mp[((NPC + i) * (2 * N1 - 1 - (NPC + i))) / 2 + NPC + j] = Sbuf[(DE + i) * SLD + DE + j];
    }
    if (el == 0) {   // cost of the view: the 64 per-lane sums in one LDS round trip (a shuffle tree is six of them)
      double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        c0 += Cb[4 * q];
        c1 += Cb[4 * q + 1];
        c2 += Cb[4 * q + 2];
        c3 += Cb[4 * q + 3];
      }
      Mp[REC] = 0.5 * ((c0 + c1) + (c2 + c3));
      Mp[REC + 1] = (double)count_total;
      if (RECP > REC + 2) Mp[REC + 2] = 0.0;
    }
    lds_fence();
    const double2* mp2 = reinterpret_cast<const double2*>(Mp);
    double2* out2 = reinterpret_cast<double2*>(out);
    for (int e = el; e < RECP / 2; e += 64) out2[e] = mp2[e];
  } else {
    // epilogue.  lane j < N1 owns column j of the local system:
    //   y[a] = (S That)[a][j]  for the DE base rows  (That column j in registers, S rows are LDS broadcasts)
    //   M[i][j] = sum_a That[a][i] y[a]  (i < NPC, i <= j),   M[i][j] = S[DE + i - NPC][.]-row entries otherwise
    const int j = lane;
    if (j < N1) {
      double tcol[DE], y[DE];
      const bool pose_col = j < NPC;
  #pragma unroll
      for (int a = 0; a < DE; ++a) tcol[a] = pose_col ? Tm[a * NPC + j] : 0.0;
  #pragma unroll
      for (int a = 0; a < DE; ++a) {
        double sum;
        if (pose_col) {
          sum = 0.0;
  #pragma unroll
          for (int bb = 0; bb < DE; ++bb) sum += Sbuf[a * SLD + bb] * tcol[bb];
        } else {
          sum = Sbuf[a * SLD + DE + (j - NPC)];
        }
        y[a] = sum;
      }
      // rows i < NPC of column j
      const int imax = pose_col ? j : NPC - 1;
      for (int i = 0; i <= imax; ++i) {
        double m = 0.0;
  #pragma unroll
        for (int a = 0; a < DE; ++a) m += Tm[a * NPC + i] * y[a];
        out[tri_index(i, j, N1)] = m;
      }
      // rows i >= NPC (intrinsics / residual rows): M[i][j] = S[DE + i - NPC][DE + j - NPC]
      if (!pose_col)
        for (int i = NPC; i <= j; ++i) out[tri_index(i, j, N1)] = Sbuf[(DE + i - NPC) * SLD + DE + (j - NPC)];
    }
  }
  if constexpr (!MFMA) {
    cost = wave_sum(cost);
    if (lane == 0) {
      out[REC] = 0.5 * cost;
      out[REC + 1] = (double)count_total;
    }
  }
  if (lane == 0) {
    if (prof) {
      long long* o = t.dbg + (size_t)v * 8;
      o[0] = stamp[1] - stamp[0];                 // setup: That columns, staging clear, compaction
      o[1] = stamp[2];                            // forward model + Jacobian rows (all chunks)
      o[2] = stamp[3] - stamp[1] - stamp[2];      // LDS staging + MFMA
      o[3] = clock64() - stamp[3];                // epilogue
      o[4] = count_total;
      o[5] = ep0 - stamp[3];                      // epilogue part 1: corner reduction + S -> LDS
      o[6] = ep1 - ep0;                           // epilogue part 2: Y = S_EE That
      o[7] = clock64() - stamp[0];                // lifetime
    }
  }
  lds_fence();   // the next view reuses the LDS buffers
#if defined(MCBA_EXP_PRIO)
  __builtin_amdgcn_s_setprio(0);
#endif
  front_ready = has_next;
  if (has_next) {
    cur ^= 1;
    count = count_n;
  }
  }
}

}  // namespace mcba
