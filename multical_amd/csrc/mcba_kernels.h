// mcba_kernels.h -- HIP kernels of the bundle-adjustment hot path, written for gfx950 (CDNA4, wave64).
//
// Evaluation pipeline at a parameter vector x (all FP64):
//   k_prep        x -> pose table (R, t, left Jacobian per pose), camera table, board points
//   k_views       pose table -> one board->camera chain matrix per view (camera, frame, board)
//   k_residual    one thread per table slot: residuals / projections / reprojection errors         (evaluate())
//   k_cost        one thread per slot + block reduction: robust cost of a trial step
//   k_jacobian    analytic Jacobian rows in the reference's sparsity pattern                        (parity / scipy-driven mode)
//   k_linearize   ONE WAVEFRONT PER VIEW: per-point row pairs V = [E | K | r] are staged through LDS and
//                 accumulated into S = V^T V with v_mfma_f64_16x16x4_f64 (the MFMA does the cross-lane
//                 reduction); the view's local normal equations M = That^T S That are written as one record.
//   k_assemble_*  deterministic reductions of the records into H_ss (dense, shared parameters),
//                 H_fs / H_ff (per-frame blocks), g and diag(H)
//   k_schur_* / k_chol_* / k_vec_*   the damped normal-equation solve of the trust-region driver
//
// There is no reference counterpart for the normal-equation kernels (the reference hands a finite-difference
// sparse Jacobian to scipy's LSMR, optimization/calibration.py:209-210); their specification is J^T J, J^T f of the
// residual function `evaluate` (calibration.py:204-206) and is tested as such.
#pragma once
#include <hip/hip_runtime.h>
#include "mcba_view.h"

namespace mcba {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// small device utilities
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_fence() {
  // LDS operations of one wavefront complete in issue order; the fence only stops the compiler from moving
  // accesses across it and drains lgkmcnt (single-wave workgroups: no s_barrier needed).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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

// ---------------------------------------------------------------------------------------------------------------
// k_residual: evaluate() of optimization/calibration.py:204-206 (+ projections and per-slot errors of
//             tables.reprojection_error, tables.py:244-249).  One thread per slot, frame-major coalesced loads of the
//             16-byte observations; outputs are scattered into the reference's [C,F,B,P] order / compacted residual
//             order through the precomputed obs_index.
// ---------------------------------------------------------------------------------------------------------------
template <int ND, bool FISH, bool ROLL>
__global__ void k_residual(Dims d, Tables t, double* __restrict__ r, double* __restrict__ proj,
                           double* __restrict__ err, uint8_t* __restrict__ valid) {
  const int n = d.slots();
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const int idx = t.obs_index[s];
    if (proj == nullptr && err == nullptr && idx < 0) continue;
    const int p = s % d.P, v = s / d.P;
    const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;
    const double2 ob = t.obs[s];
    double uv[2], Xs[3], Xe[3], tr;
    slot_forward<ND, FISH, ROLL, false>(d, t, v, c, b, p, ob, uv, nullptr, nullptr, Xs, Xe, tr);
    const double ex = uv[0] - ob.x, ey = uv[1] - ob.y;
    if (r != nullptr && idx >= 0) {
      r[2 * (size_t)idx] = ex;
      r[2 * (size_t)idx + 1] = ey;
    }
    const size_t ri = (((size_t)c * d.F + f) * d.B + b) * d.P + p;
    if (proj != nullptr) {
      proj[2 * ri] = uv[0];
      proj[2 * ri + 1] = uv[1];
    }
    if (err != nullptr) {
      const bool ok = t.evalid[s] != 0;
      err[ri] = ok ? sqrt(ex * ex + ey * ey) : 0.0;
      valid[ri] = ok ? 1 : 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_cost: 0.5 * sum rho(f^2) over the inliers of the shard; partial sums per block (fixed grid -> deterministic)
// ---------------------------------------------------------------------------------------------------------------
template <int ND, bool FISH, bool ROLL>
__global__ void k_cost(Dims d, Tables t, double* __restrict__ partial) {
  __shared__ double scratch[16];
  const int n = d.slots();
  double acc = 0.0;
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    if (!t.inlier[s]) continue;
    const int p = s % d.P, v = s / d.P;
    const int b = v % d.B, c = (v / d.B) % d.C;
    const double2 ob = t.obs[s];
    double uv[2], Xs[3], Xe[3], tr;
    slot_forward<ND, FISH, ROLL, false>(d, t, v, c, b, p, ob, uv, nullptr, nullptr, Xs, Xe, tr);
    double rs, fs;
    acc += robust_loss(d.loss, d.f_scale, uv[0] - ob.x, &rs, &fs);
    acc += robust_loss(d.loss, d.f_scale, uv[1] - ob.y, &rs, &fs);
  }
  const double tot = block_reduce<false>(acc, scratch);
  if (threadIdx.x == 0) partial[blockIdx.x] = 0.5 * tot;
}

// ---------------------------------------------------------------------------------------------------------------
// k_jacobian: analytic Jacobian rows in the column order of Calibration.sparsity_matrix
//             (optimization/calibration.py:173-196).  One thread per inlier observation (not a hot path).
// ---------------------------------------------------------------------------------------------------------------
template <int ND, bool FISH, bool ROLL>
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
    double vr[2 * NV];
    point_rows<ND, FISH, ROLL, true>(dl, t, v, c, b, p, t.obs[s], vr);
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
        oc[pos] = local_to_x(d, f, c, b, 6 * k + jj);
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
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_linearize: fused residual + Jacobian -> per-view local normal equations.
//
//   grid  = one 64-thread workgroup (one wavefront) per view (frame-major), empty views exit at once.
//   loop  = chunks of 64 board points: lane = point.  Each lane evaluates the forward model and the 2 x NV row pair
//           V = [E | K | r] in registers, the rows are transposed through an LDS staging buffer
//           (row stride NVP+1 doubles: conflict-free 128-byte row reads, 2-way write conflicts), and
//           S += V^T V is accumulated by v_mfma_f64_16x16x4_f64: operand lane l holds V[4 s + (l>>4)][16 t + (l&15)],
//           A and B operands are the SAME register for diagonal tiles.  NV <= 16 -> one tile, NV <= 32 -> three.
//   epilogue: Y = S That, M = That^T Y (That = [T_cam | T_frame.. | T_board] (+) I) -> packed upper triangle record.
//   MFMA=false keeps the identical data flow with plain FMAs over the staged rows (validation / fallback build).
// ---------------------------------------------------------------------------------------------------------------
template <int ND, bool FISH, int MOTION, bool OPTK, bool MFMA>
__global__ __launch_bounds__(64) void k_linearize(Dims d, Tables t, double* __restrict__ rec,
                                                  const uint16_t* __restrict__ tri) {
  constexpr bool ROLL = MOTION == MOTION_ROLLING;
  constexpr int DE = ROLL ? 12 : 6, NPB = MOTION == MOTION_STATIC ? 3 : 4, KI = OPTK ? 4 + ND : 0;
  constexpr int NV = DE + KI + 1, NT = (NV + 15) / 16, NVP = 16 * NT, LDV = NVP + 1;
  constexpr int NPC = 6 * NPB, NL = NPC + KI, N1 = NL + 1;
  constexpr int PTS = NT == 1 ? 64 : 32, RND = 64 / PTS, ROWS = 2 * PTS;
  constexpr int REC = N1 * (N1 + 1) / 2;
  static_assert(NV * N1 <= ROWS * LDV, "Y does not fit in the staging buffer");

  __shared__ double Vbuf[ROWS * LDV];
  __shared__ double Sbuf[NVP * NVP];
  __shared__ double Tm[DE * NPC];

  const int v = blockIdx.x, lane = threadIdx.x;
  const int count = t.view_count[v];
  if (count == 0) return;
  const int b = v % d.B, c = (v / d.B) % d.C, fl = v / (d.B * d.C), f = d.f0 + fl;

  if (lane < NPC) {
    double col[12];
    view_column(d, t, f, c, b, lane, col);
    for (int a = 0; a < DE; ++a) Tm[a * NPC + lane] = col[a];
  }
  for (int e = lane; e < ROWS * LDV; e += 64) Vbuf[e] = 0.0;   // pad columns stay zero for the whole kernel

  // accumulators
  constexpr int NACC_V = NVP * NVP / 64;                  // plain-FMA variant: NVP*NVP entries over 64 lanes
  constexpr int NTILE = NT * (NT + 1) / 2;
  double accv[MFMA ? 1 : NACC_V];
  double4_t accm[MFMA ? NTILE : 1];
  if constexpr (MFMA) {
    for (int i = 0; i < NTILE; ++i) accm[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  } else {
    for (int i = 0; i < NACC_V; ++i) accv[i] = 0.0;
  }
  double cost = 0.0;
  lds_fence();

  const int nchunks = (d.P + 63) / 64;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int p = chunk * 64 + lane;
    const size_t s = (size_t)v * d.P + p;
    const bool in = p < d.P && t.inlier[s] != 0;
    double vr[2 * NV];
    if (in) {
      cost += point_rows<ND, FISH, ROLL, OPTK>(d, t, v, c, b, p, t.obs[s], vr);
    } else {
      for (int i = 0; i < 2 * NV; ++i) vr[i] = 0.0;
    }
    if (__ballot(in) == 0ull) continue;   // wave-uniform: nothing to add

    for (int q = 0; q < RND; ++q) {
      if (RND == 1 || (lane / PTS) == q) {
        const int row0 = 2 * (lane % PTS);
        for (int i = 0; i < NV; ++i) {
          Vbuf[row0 * LDV + i] = vr[i];
          Vbuf[(row0 + 1) * LDV + i] = vr[NV + i];
        }
      }
      lds_fence();
      if constexpr (MFMA) {
        const int rsub = lane >> 4, csub = lane & 15;
        for (int st = 0; st < ROWS / 4; ++st) {
          double a[NT];
          for (int tt = 0; tt < NT; ++tt) a[tt] = Vbuf[(4 * st + rsub) * LDV + 16 * tt + csub];
          int ti = 0;
          for (int t0 = 0; t0 < NT; ++t0)
            for (int t1 = t0; t1 < NT; ++t1, ++ti)
              accm[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t0], a[t1], accm[ti], 0, 0, 0);
        }
      } else {
        constexpr int IW = NVP;                  // i index width
        constexpr int JW = NACC_V;               // j entries per lane
        const int i = lane % IW, j0 = (lane / IW) * JW;
        for (int row = 0; row < ROWS; ++row) {
          const double vi = Vbuf[row * LDV + i];
          for (int jj = 0; jj < JW; ++jj) accv[jj] += vi * Vbuf[row * LDV + j0 + jj];
        }
      }
      lds_fence();
    }
  }

  // S -> LDS (full symmetric matrix)
  if constexpr (MFMA) {
    const int rsub = lane >> 4, csub = lane & 15;
    int ti = 0;
    for (int t0 = 0; t0 < NT; ++t0)
      for (int t1 = t0; t1 < NT; ++t1, ++ti)
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * t0 + rsub + 4 * r, col = 16 * t1 + csub;
          Sbuf[row * NVP + col] = accm[ti][r];
          if (t0 != t1) Sbuf[col * NVP + row] = accm[ti][r];
        }
  } else {
    constexpr int IW = NVP, JW = NACC_V;
    const int i = lane % IW, j0 = (lane / IW) * JW;
    for (int jj = 0; jj < JW; ++jj) Sbuf[i * NVP + j0 + jj] = accv[jj];
  }
  lds_fence();

  // epilogue: Y = S That  (NV x N1), staged in the (now free) row buffer
  double* Y = Vbuf;
  for (int e = lane; e < NV * N1; e += 64) {
    const int a = e / N1, j = e % N1;
    double sum;
    if (j < NPC) {
      sum = 0.0;
      for (int bb = 0; bb < DE; ++bb) sum += Sbuf[a * NVP + bb] * Tm[bb * NPC + j];
    } else {
      sum = Sbuf[a * NVP + DE + (j - NPC)];
    }
    Y[e] = sum;
  }
  lds_fence();
  double* out = rec + (size_t)v * d.rec_stride;
  for (int e = lane; e < REC; e += 64) {
    const int ij = tri[e];
    const int i = ij >> 8, j = ij & 255;
    double m;
    if (i < NPC) {
      m = 0.0;
      for (int a = 0; a < DE; ++a) m += Tm[a * NPC + i] * Y[a * N1 + j];
    } else {
      m = Y[(DE + i - NPC) * N1 + j];
    }
    out[e] = m;
  }
  cost = wave_sum(cost);
  if (lane == 0) {
    out[REC] = 0.5 * cost;
    out[REC + 1] = (double)count;
  }
}

}  // namespace mcba
