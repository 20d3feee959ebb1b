// mcba_init_kernels.h -- initialisation tables on the device: the robust alignment of two collections of
// poses that multical's pose-graph initialisation is made of (paths relative to /root/reference/multical/):
//     matrix.align_transforms_robust      transform/matrix.py:140-153
//       = mean_robust(relative_to(m1[mask], m2[mask]))   -> errors of ALL entries -> upper-quartile outlier test -> mean_robust
//     matrix.mean_robust                  transform/matrix.py:109-113: poses -> (rotation vector | translation) 6-vectors,
//                                         common.mean_robust (transform/common.py:6-21): scipy linkage(whiten(v), 'ward'),
//                                         fcluster(maxclust = max(n / 10, 3)), mean of the most common cluster
// used by tables.estimate_transform (tables.py:153-176: camera and board pairs of the overlap spanning tree, hundreds to
// thousands of entries each) and tables.relative_between_n (tables.py:334-345: one small problem per frame).
//
// ONE WORKGROUP PER PROBLEM, all stages inside the kernel.  The Ward clustering is the nearest-neighbour-chain algorithm
// scipy runs (scipy/cluster/_hierarchy.pyx: nn_chain), on cluster centroids and sizes instead of a condensed distance
// matrix (for Ward the Lance-Williams recurrence equals d(A,B) = sqrt(2 nA nB / (nA + nB)) |cA - cB|): every nearest-
// neighbour search is a parallel scan over the live clusters with scipy's tie rules (lowest index; the previous chain
// element wins ties).  Cutting the dendrogram at `maxclust` = applying the n - t merges of smallest height (union-find on
// representatives), the most common cluster is the largest component (ties: the one whose first member comes first).
// Floating point is not bit-identical to scipy (different summation orders): the result agrees to ~1e-12 unless two
// merge heights tie to the last bit.
#pragma once
#include <hip/hip_runtime.h>
#include "mcba_math.h"

namespace mcba {

constexpr int ALIGN_THREADS = 1024;   // the nearest-neighbour scans of the big pair problems (thousands of entries) set the pace

struct AlignScratch {        // per-problem device scratch, sized for the largest problem (n entries)
  double* vec;       // [n][6]  relative poses as rotation vector | translation (compacted)
  double* cen;       // [n][6]  whitened cluster centroids
  double* err;       // [n]     alignment errors of all entries
  double* hgt;       // [n]     merge heights
  int* size;         // [n]     cluster sizes (0 = dead)
  int* chain;        // [n]
  int* rep_a;        // [n]     representative members of the two merged clusters
  int* rep_b;        // [n]
  int* parent;       // [n]     union-find / labels
  int* list;         // [n]     compacted entry indices
};

// ---- SE(3) helpers on row-major 4x4 ---------------------------------------------------------------------------------
__device__ __forceinline__ void se3_load(const double* m, double* R, double* t) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R[3 * i + j] = m[4 * i + j];
    t[i] = m[4 * i + 3];
  }
}
__device__ __forceinline__ void se3_inv(const double* R, const double* t, double* Ri, double* ti) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Ri[3 * i + j] = R[3 * j + i];
  for (int i = 0; i < 3; ++i) ti[i] = -(Ri[3 * i] * t[0] + Ri[3 * i + 1] * t[1] + Ri[3 * i + 2] * t[2]);
}

// rotation matrix -> rotation vector like scipy's Rotation.from_matrix(...).as_rotvec() (transform/rtvec.py:29-32)
__device__ __forceinline__ void rotvec_from_matrix(const double* R, double* w) {
  const double m00 = R[0], m11 = R[4], m22 = R[8], tr = m00 + m11 + m22;
  double q[4];
  int choice;
  // (numpy argmax over [m00, m11, m22, trace] takes the FIRST maximum: strict comparisons against the running best)
  {
    const double dec[4] = {m00, m11, m22, tr};
    choice = 0;
    for (int i = 1; i < 4; ++i)
      if (dec[i] > dec[choice]) choice = i;
  }
  if (choice != 3) {
    const int i = choice, j = (i + 1) % 3, k = (j + 1) % 3;
    q[i] = 1.0 - tr + 2.0 * R[3 * i + i];
    q[j] = R[3 * j + i] + R[3 * i + j];
    q[k] = R[3 * k + i] + R[3 * i + k];
    q[3] = R[3 * k + j] - R[3 * j + k];
  } else {
    q[0] = R[7] - R[5];
    q[1] = R[2] - R[6];
    q[2] = R[3] - R[1];
    q[3] = 1.0 + tr;
  }
  const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= nrm;
  if (q[3] < 0.0)
    for (int i = 0; i < 4; ++i) q[i] = -q[i];
  const double sn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  const double angle = 2.0 * atan2(sn, q[3]);
  double scale;
  if (angle <= 1e-3) {
    const double a2 = angle * angle;
    scale = 2.0 + a2 / 12.0 + 7.0 * a2 * a2 / 2880.0;
  } else {
    scale = angle / sin(angle / 2.0);
  }
  w[0] = scale * q[0];
  w[1] = scale * q[1];
  w[2] = scale * q[2];
}

// block-wide argmin of (value, index) with the LOWEST index among equal values; all threads get the result
__device__ __forceinline__ void block_argmin(double& v, int& idx, double* sv, int* si) {
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_down(v, off, 64);
    const int oi = __shfl_down(idx, off, 64);
    if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) { sv[wave] = v; si[wave] = idx; }
  __syncthreads();
  v = sv[0];
  idx = si[0];
  for (int w = 1; w < nw; ++w)
    if (sv[w] < v || (sv[w] == v && si[w] < idx)) { v = sv[w]; idx = si[w]; }
}

// robust mean of the n 6-vectors vec[0..n) (transform/common.py:6-21) -> out[6]; every thread of the block takes part
__device__ void robust_mean_block(int n, const AlignScratch& s, double* out /* shared [6] */, double* sv, int* si,
                                  int* s_int /* shared [4] */) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (n == 1) {
    if (tid < 6) out[tid] = s.vec[tid];
    __syncthreads();
    return;
  }
  __shared__ double stdv[6];
  if (tid < 6) {     // scipy.cluster.vq.whiten: divide by the population standard deviation (zero -> 1)
    double mean = 0.0;
    for (int i = 0; i < n; ++i) mean += s.vec[6 * i + tid];
    mean /= n;
    double var = 0.0;
    for (int i = 0; i < n; ++i) {
      const double dlt = s.vec[6 * i + tid] - mean;
      var += dlt * dlt;
    }
    const double sd = sqrt(var / n);
    stdv[tid] = sd == 0.0 ? 1.0 : sd;
  }
  __syncthreads();
  for (int i = tid; i < n; i += nthr) {
    for (int j = 0; j < 6; ++j) s.cen[6 * i + j] = s.vec[6 * i + j] / stdv[j];
    s.size[i] = 1;
    s.parent[i] = i;
  }
  __syncthreads();
  const int t_clust = max((int)fmax((double)n / 10.0, 3.0), 1);   // fcluster(..., t = max(n / 10, 3)): int(t)
  if (t_clust < n) {
    // ---- nearest-neighbour chain (scipy _hierarchy.nn_chain, method 'ward') ----------------------------------------
    int chain_len = 0;
    for (int k = 0; k < n - 1; ++k) {
      if (chain_len == 0) {
        // first live cluster
        double fv = 0.0;
        int fi = 0x7fffffff;
        for (int i = tid; i < n; i += nthr)
          if (s.size[i] > 0 && i < fi) fi = i;
        fv = (double)fi;
        block_argmin(fv, fi, sv, si);
        if (tid == 0) s.chain[0] = fi;
        chain_len = 1;
        __syncthreads();
      }
      int x, y;
      double dmin;
      while (true) {
        x = s.chain[chain_len - 1];
        const int prev = chain_len > 1 ? s.chain[chain_len - 2] : -1;
        double cx[6];
        for (int j = 0; j < 6; ++j) cx[j] = s.cen[6 * x + j];
        const double nx = (double)s.size[x];
        auto ward = [&](int i) {
          const double ni = (double)s.size[i];
          double d2 = 0.0;
          for (int j = 0; j < 6; ++j) {
            const double dl = cx[j] - s.cen[6 * i + j];
            d2 += dl * dl;
          }
          return sqrt(2.0 * nx * ni / (nx + ni) * d2);
        };
        const double dprev = prev >= 0 ? ward(prev) : INFINITY;
        double bv = INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < n; i += nthr) {
          if (s.size[i] == 0 || i == x) continue;
          const double dd = ward(i);
          if (dd < bv) { bv = dd; bi = i; }      // (ascending i per thread: the first minimum)
        }
        block_argmin(bv, bi, sv, si);
        if (prev >= 0 && !(bv < dprev)) { y = prev; dmin = dprev; } else { y = bi; dmin = bv; }   // previous element wins ties
        if (prev >= 0 && y == prev) break;
        if (tid == 0) s.chain[chain_len] = y;
        ++chain_len;
        __syncthreads();
      }
      chain_len -= 2;
      if (x > y) { const int tmp = x; x = y; y = tmp; }
      __syncthreads();
      if (tid == 0) {
        const int nx = s.size[x], ny = s.size[y];
        s.hgt[k] = dmin;
        s.rep_a[k] = x;       // slot indices double as representatives: slot y keeps holding the merged cluster, x dies;
        s.rep_b[k] = y;       // a slot index is always a member of the cluster it holds (it is one of the original points)
        for (int j = 0; j < 6; ++j)
          s.cen[6 * y + j] = ((double)nx * s.cen[6 * x + j] + (double)ny * s.cen[6 * y + j]) / (double)(nx + ny);
        s.size[y] = nx + ny;
        s.size[x] = 0;
      }
      __syncthreads();
    }
    // ---- cut: scipy's fcluster(criterion='maxclust') finds the smallest merge height thr that leaves at most t_clust
    // clusters and then applies EVERY merge of height <= thr (cluster_maxclust_monocrit + cluster_monocrit: for a monotone
    // Ward dendrogram the criterion is the merge height itself).  thr is the (n - t_clust)-th smallest height; merges that
    // tie with it are applied too -- exact duplicates among the relative poses (noise-free or repeated detections) give
    // zero-height merges, and the flat clusters then hold fewer than t_clust groups, exactly as in the reference.
    const int nm = n - 1, keep = n - t_clust;
    for (int k = tid; k < nm; k += nthr) {
      const double hk = s.hgt[k];
      int rank = 0;
      for (int q = 0; q < nm; ++q) {
        const double hq = s.hgt[q];
        rank += (hq < hk || (hq == hk && q < k)) ? 1 : 0;
      }
      if (rank == keep - 1) sv[0] = hk;      // (ranks are a permutation: exactly one writer)
    }
    __syncthreads();
    if (tid == 0) {
      const double thr = sv[0];
      for (int k = 0; k < nm; ++k) {
        if (!(s.hgt[k] <= thr)) continue;
        int a = s.rep_a[k], b = s.rep_b[k];
        while (s.parent[a] != a) a = s.parent[a];
        while (s.parent[b] != b) b = s.parent[b];
        if (a != b) s.parent[max(a, b)] = min(a, b);
      }
    }
    __syncthreads();
  }
  // labels = root of every point; component sizes; the most common cluster (ties: first encountered in index order)
  for (int i = tid; i < n; i += nthr) {
    int r = i;
    while (s.parent[r] != r) r = s.parent[r];
    s.list[i] = r;            // label = smallest member (roots are minima)
  }
  for (int i = tid; i < n; i += nthr) s.size[i] = 0;
  __syncthreads();
  if (tid == 0) {
    for (int i = 0; i < n; ++i) s.size[s.list[i]] += 1;
    int best = -1, bestc = 0;
    for (int i = 0; i < n; ++i) {          // first-encountered label in index order = its root (the smallest member)
      const int c = s.size[i];
      if (s.list[i] == i && c > bestc) { bestc = c; best = i; }
    }
    s_int[0] = best;
    s_int[1] = bestc;
  }
  __syncthreads();
  if (tid < 6) {      // numpy mean over axis 0: rows are added in index order
    const int best = s_int[0];
    double acc = 0.0;
    for (int i = 0; i < n; ++i)
      if (s.list[i] == best) acc += s.vec[6 * i + tid];
    out[tid] = acc / (double)s_int[1];
  }
  __syncthreads();
}

// rtvec -> 4x4 (rtvec.py:24-27)
__device__ __forceinline__ void rtvec_to_matrix4(const double* v, double* R, double* t) {
  double L[9];
  rodrigues(v, R, L);
  t[0] = v[3]; t[1] = v[4]; t[2] = v[5];
}

// One workgroup per problem p: entries [off[p], off[p+1]) of A / B (row-major 4x4), mask (or null = all).
//   invert != 0: relative_between_inv (tables.py:334-335): the inputs are inverted and so is the result.
//   out[p] = the aligned transform (identity when the problem has no masked entry -> out_valid[p] = 0);
//   inliers (or null) = the entries that passed the outlier test.
__global__ __launch_bounds__(ALIGN_THREADS) void k_align_robust(const long long* __restrict__ off, const double* __restrict__ A,
                                                                const double* __restrict__ B, const uint8_t* __restrict__ mask,
                                                                double threshold, int invert, long long scratch_stride,
                                                                AlignScratch base, double* __restrict__ out,
                                                                uint8_t* __restrict__ out_valid, uint8_t* __restrict__ inliers) {
  __shared__ double sv[ALIGN_THREADS / 64], mean6[6], Rm[9], tm[3];
  __shared__ int si[ALIGN_THREADS / 64], s_int[4], s_cnt;
  const int p = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const long long e0 = off[p];
  const int n = (int)(off[p + 1] - e0);
  AlignScratch s = base;
  {
    const long long o = (long long)p * scratch_stride;
    s.vec += 6 * o; s.cen += 6 * o; s.err += o; s.hgt += o; s.size += o; s.chain += o; s.rep_a += o; s.rep_b += o;
    s.parent += o; s.list += o;
  }
  const double* Ap = A + 16 * e0;
  const double* Bp = B + 16 * e0;
  const uint8_t* mp = mask ? mask + e0 : nullptr;
  auto load_pair = [&](int k, double* Ra, double* ta, double* Rb, double* tb) {
    se3_load(Ap + 16 * (size_t)k, Ra, ta);
    se3_load(Bp + 16 * (size_t)k, Rb, tb);
    if (invert) {
      double Ri[9], ti[3];
      se3_inv(Ra, ta, Ri, ti);
      for (int i = 0; i < 9; ++i) Ra[i] = Ri[i];
      for (int i = 0; i < 3; ++i) ta[i] = ti[i];
      se3_inv(Rb, tb, Ri, ti);
      for (int i = 0; i < 9; ++i) Rb[i] = Ri[i];
      for (int i = 0; i < 3; ++i) tb[i] = ti[i];
    }
  };
  // pass = 0: entries of the mask; pass = 1: inliers of the outlier test
  for (int pass = 0; pass < 2; ++pass) {
    // ---- stable compaction of the selected entries (serial prefix by one thread: n is at most a few thousand) ------
    if (tid == 0) {
      int c = 0;
      for (int k = 0; k < n; ++k) {
        const bool sel = pass == 0 ? (mp == nullptr || mp[k] != 0) : (s.parent[k] != 0);   // (pass 1 reads the inlier flags)
        if (sel) s.list[c++] = k;
      }
      s_cnt = c;
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt == 0) {     // tables.relative_between: no common entry -> invalid pose (identity)
      if (tid < 16) out[16 * (size_t)p + tid] = (tid % 5 == 0) ? 1.0 : 0.0;
      if (tid == 0) out_valid[p] = 0;
      if (inliers != nullptr && pass == 0)
        for (int k = tid; k < n; k += nthr) inliers[e0 + k] = 0;
      return;
    }
    // ---- relative poses dest . source^-1 (matrix.relative_to) as 6-vectors ---------------------------------------
    for (int i = tid; i < cnt; i += nthr) {
      double Ra[9], ta[3], Rb[9], tb[3], Rai[9], tai[3], Rr[9], tr3[3];
      load_pair(s.list[i], Ra, ta, Rb, tb);
      se3_inv(Ra, ta, Rai, tai);
      se3_mul(Rb, tb, Rai, tai, Rr, tr3);
      double w[3];
      rotvec_from_matrix(Rr, w);
      for (int j = 0; j < 3; ++j) { s.vec[6 * i + j] = w[j]; s.vec[6 * i + 3 + j] = tr3[j]; }
    }
    __syncthreads();
    robust_mean_block(cnt, s, mean6, sv, si, s_int);
    if (tid == 0) rtvec_to_matrix4(mean6, Rm, tm);
    __syncthreads();
    if (pass == 1) break;
    // ---- errors of ALL entries |m A_k - B_k|_F, upper quartile, outlier test (matrix.py:135-153) -------------------
    for (int k = tid; k < n; k += nthr) {
      double Ra[9], ta[3], Rb[9], tb[3], Rr[9], tr3[3];
      load_pair(k, Ra, ta, Rb, tb);
      se3_mul(Rm, tm, Ra, ta, Rr, tr3);
      double e2 = 0.0;
      for (int i = 0; i < 9; ++i) e2 += (Rr[i] - Rb[i]) * (Rr[i] - Rb[i]);
      for (int i = 0; i < 3; ++i) e2 += (tr3[i] - tb[i]) * (tr3[i] - tb[i]);
      s.err[k] = sqrt(e2);
    }
    __syncthreads();
    {   // numpy quantile 0.75, method 'linear': virtual index (n - 1) * 0.75 between two order statistics
      const double virt = (double)(n - 1) * 0.75;
      const int lo = (int)floor(virt), hi = min(lo + 1, n - 1);
      const double gamma = virt - floor(virt);
      for (int k = tid; k < n; k += nthr) {
        const double ek = s.err[k];
        int rank = 0;
        for (int q = 0; q < n; ++q) {
          const double eq = s.err[q];
          rank += (eq < ek || (eq == ek && q < k)) ? 1 : 0;
        }
        if (rank == lo) sv[0] = ek;            // (sv doubles as the hand-over of the two order statistics)
        if (rank == hi) sv[1] = ek;
      }
      __syncthreads();
      const double a = sv[0], b = sv[1], diff = b - a;
      const double uq = gamma >= 0.5 ? b - diff * (1.0 - gamma) : a + diff * gamma;   // numpy _lerp
      __syncthreads();
      for (int k = tid; k < n; k += nthr) {
        const bool in = s.err[k] < uq * threshold && (mp == nullptr || mp[k] != 0);
        s.parent[k] = in ? 1 : 0;              // inlier flags (parent is re-initialised by the next robust mean)
        if (inliers != nullptr) inliers[e0 + k] = in ? 1 : 0;
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    double Ro[9], to[3];
    if (invert) se3_inv(Rm, tm, Ro, to);
    else { for (int i = 0; i < 9; ++i) Ro[i] = Rm[i]; for (int i = 0; i < 3; ++i) to[i] = tm[i]; }
    double* o = out + 16 * (size_t)p;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) o[4 * i + j] = Ro[3 * i + j];
      o[4 * i + 3] = to[i];
    }
    o[12] = o[13] = o[14] = 0.0;
    o[15] = 1.0;
    out_valid[p] = 1;
  }
}

}  // namespace mcba
