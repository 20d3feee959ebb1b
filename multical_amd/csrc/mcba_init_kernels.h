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
// ONE WORKGROUP PER PROBLEM, all stages inside the kernel.  The Ward clustering produces the dendrogram of the nearest-
// neighbour-chain algorithm scipy runs (scipy/cluster/_hierarchy.pyx: nn_chain), on cluster centroids and sizes instead of a
// condensed distance matrix (for Ward the Lance-Williams recurrence equals d(A,B) = sqrt(2 nA nB / (nA + nB)) |cA - cB|), by
// rounds of PARALLEL reciprocal-nearest-neighbour merges (Ward is reducible: see robust_mean_block); the clustering state of
// problems with up to ALIGN_LDS_CAP selected entries lives in LDS.  Cutting the dendrogram at `maxclust` = applying every merge
// up to the threshold height (union-find on representatives), the most common cluster is the largest component (ties: the
// one whose first member comes first).  Floating point is not bit-identical to scipy (different summation orders): the
// result agrees to ~1e-12 unless two merge heights tie to the last bit.
#pragma once
#include <hip/hip_runtime.h>
#include "mcba_math.h"

namespace mcba {

constexpr int ALIGN_THREADS = 1024;   // the nearest-neighbour scans of the big pair problems (thousands of entries) set the pace
constexpr int ALIGN_LDS_CAP = 1600;   // entries whose clustering state (8 doubles + 7 ints = 92 B) fits 147 KB of dynamic LDS
__host__ __device__ inline size_t align_lds_bytes(int cap) { return (size_t)cap * (8 * sizeof(double) + 7 * sizeof(int)) + 16; }

struct AlignScratch {        // per-problem device scratch, sized for the largest problem (n entries)
  double* vec;       // [n][6]  relative poses as rotation vector | translation (compacted)
  double* cen;       // [n][6]  whitened cluster centroids
  double* err;       // [n]     alignment errors of all entries
  double* hgt;       // [n]     merge heights
  double* nd;        // [n]     distance to the nearest neighbour (clustering rounds); shares storage with err in memory
  int* size;         // [n]     cluster sizes (0 = dead)
  int* chain;        // [n]
  int* rep_a;        // [n]     representative members of the two merged clusters
  int* rep_b;        // [n]
  int* parent;       // [n]     union-find / labels
  int* live;         // [n]     slots of the live clusters (clustering rounds)
  int* list;         // [n]     compacted entry indices
  long long* prof;   // optional [problems][16] phase cycles (debug), else null
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

// k-th smallest (0-based) of the n NON-NEGATIVE doubles v[0..n) (merge heights, errors: their IEEE bit patterns order like the
// values); every thread of the block takes part and gets the value.  Small sets are RANKED (n^2 / threads comparisons, two
// barriers); large ones go through a radix select on the bit patterns, eight 8-bit passes with a 256-bin histogram in LDS
// (ranking 16 000 values is 2.6e8 comparisons on one compute unit -- it was 82 M cycles of the quartile in round 2 and
// 0.8 M cycles of every dendrogram cut until round 4; the eight passes cost a 26-entry problem more than its clustering).
constexpr int KTH_RANK_MAX = 2048;
__device__ double block_kth_smallest(const double* v, int n, int k, double* sv /* shared, [>= 1] */) {
  __shared__ unsigned int r_hist[256];
  __shared__ unsigned long long r_prefix;
  __shared__ int r_rank;
  const int tid = threadIdx.x, nthr = blockDim.x;
  __syncthreads();                                       // (v may have been written by other threads just now)
  if (n <= KTH_RANK_MAX) {
    for (int i = tid; i < n; i += nthr) {
      const double vi = v[i];
      int rank = 0;
      for (int q = 0; q < n; ++q) {
        const double vq = v[q];
        rank += (vq < vi || (vq == vi && q < i)) ? 1 : 0;
      }
      if (rank == k) sv[0] = vi;                         // (ranks are a permutation: exactly one writer)
    }
    __syncthreads();
    const double r = sv[0];
    __syncthreads();
    return r;
  }
  if (tid == 0) { r_prefix = 0ull; r_rank = k; }
  for (int shift = 56; shift >= 0; shift -= 8) {
    if (tid < 256) r_hist[tid] = 0u;
    __syncthreads();
    const unsigned long long pre = r_prefix;
    for (int i = tid; i < n; i += nthr) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(v[i]);
      const bool match = shift == 56 || (key >> (shift + 8)) == (pre >> (shift + 8));
      if (match) atomicAdd(&r_hist[(unsigned)(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      int r = r_rank, dgt = 0;
      while (dgt < 255 && r >= (int)r_hist[dgt]) { r -= (int)r_hist[dgt]; ++dgt; }
      r_rank = r;
      r_prefix = pre | ((unsigned long long)dgt << shift);
    }
    __syncthreads();
  }
  const double r = __longlong_as_double((long long)r_prefix);
  __syncthreads();
  return r;
}

// the k-th and the k2-th smallest (k2 = k or k + 1) in one go: the order statistics on both sides of a quantile's virtual index.
// Large sets: ONE radix descent for the k-th, then one pass for the number of values <= it and the smallest value above it
// (sixteen histogram passes for the two statistics were most of k_align_stage_post at 16 000 entries).
__device__ void block_kth_pair(const double* v, int n, int k, int k2, double* sv /* shared, [>= 2] */, double& a, double& b) {
  a = block_kth_smallest(v, n, k, sv);
  if (k2 == k) { b = a; return; }
  if (n <= KTH_RANK_MAX) { b = block_kth_smallest(v, n, k2, sv); return; }
  __shared__ int p_le[ALIGN_THREADS / 64];
  __shared__ double p_min[ALIGN_THREADS / 64];
  const int tid = threadIdx.x, nthr = blockDim.x;
  int le = 0;
  double mn = INFINITY;
  for (int i = tid; i < n; i += nthr) {
    const double x = v[i];
    le += x <= a ? 1 : 0;
    mn = x > a ? fmin(mn, x) : mn;
  }
  for (int off = 32; off > 0; off >>= 1) {
    le += __shfl_down(le, off, 64);
    mn = fmin(mn, __shfl_down(mn, off, 64));
  }
  __syncthreads();
  if ((tid & 63) == 0) { p_le[tid >> 6] = le; p_min[tid >> 6] = mn; }
  __syncthreads();
  int tle = 0;
  double tmn = INFINITY;
  for (int w = 0; w < (nthr >> 6); ++w) { tle += p_le[w]; tmn = fmin(tmn, p_min[w]); }
  b = tle > k2 ? a : tmn;        // (the k2-th is still the value a when at least k2 + 1 values are <= a)
  __syncthreads();
}

// robust mean of the n 6-vectors vec[0..n) (transform/common.py:6-21) -> out[6]; every thread of the block takes part
constexpr int ALIGN_STAGED_MIN = 2048;  // batches with a larger problem run the staged kernels (k_align_stage_*)
constexpr int ALIGN_TILE = 1024;      // live centroids per LDS tile of the nearest-neighbour scans (56 B each)

// ---- the robust mean in PIECES (round 4): one workgroup runs them back to back (robust_mean_block, problems of up to
// ALIGN_STAGED_MIN entries), or the large pair problems run them as separate launches with the nearest-neighbour scans of a round
// spread over many compute units (k_align_stage_*).  Every piece is called by all threads of a workgroup.

// whitening + initial state of the clustering: cen = vec / std, every point its own cluster
__device__ void rm_whiten_init(int n, const AlignScratch& s, double* sv) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  // scipy.cluster.vq.whiten: divide by the population standard deviation (zero -> 1).  Column sums by all threads: thread t
  // adds the rows t, t + nthr, ... of its column set in order, the per-thread partials are folded in a fixed tree (six
  // threads walking n rows of memory one after the other took ~1.5 ms per thousand rows)
  __shared__ double stdv[6], red6[6];
  auto column_sums = [&](auto f) {   // red6[j] = sum over rows of f(row value of column j, j)
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = tid; i < n; i += nthr)
      for (int j = 0; j < 6; ++j) acc[j] += f(s.vec[6 * i + j], j);
    for (int j = 0; j < 6; ++j) {
      double v = acc[j];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      __syncthreads();
      if ((tid & 63) == 0) sv[tid >> 6] = v;
      __syncthreads();
      if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < (nthr >> 6); ++w) t += sv[w];
        red6[j] = t;
      }
    }
    __syncthreads();
  };
  column_sums([](double v, int) { return v; });
  double mean_j[6];
  for (int j = 0; j < 6; ++j) mean_j[j] = red6[j] / n;
  __syncthreads();
  column_sums([&](double v, int j) { const double dlt = v - mean_j[j]; return dlt * dlt; });
  if (tid < 6) {
    const double sd = sqrt(red6[tid] / n);
    stdv[tid] = sd == 0.0 ? 1.0 : sd;
  }
  __syncthreads();
  for (int i = tid; i < n; i += nthr) {
    for (int j = 0; j < 6; ++j) s.cen[6 * i + j] = s.vec[6 * i + j] / stdv[j];
    s.size[i] = 1;
    s.parent[i] = i;
  }
  __syncthreads();
}

// stable compaction by the whole workgroup: out[0 .. count) = the k < n with pred(k), ascending; returns count to every thread
// (ranks inside a wavefront from a ballot, wave offsets from the counts of the wavefronts in front: two barriers per blockDim
//  entries -- one wavefront stepping through the entries 64 at a time waited for a memory round trip per step)
template <typename Pred>
__device__ int block_compact(int n, int* out, Pred pred) {
  __shared__ int bc_cnt[ALIGN_THREADS / 64];
  __shared__ int bc_total;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  int total = 0;
  for (int base = 0; base < n; base += nthr) {
    const int k = base + tid;
    const bool on = k < n && pred(k);
    const unsigned long long m = __ballot(on);
    __syncthreads();                            // (bc_cnt of the previous chunk has been read)
    if (lane == 0) bc_cnt[wave] = __popcll(m);
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < nw; ++w) {
      const int c = bc_cnt[w];
      before += w < wave ? c : 0;
      all += c;
    }
    if (on) out[total + before + __popcll(m & ((1ull << lane) - 1ull))] = k;
    total += all;
  }
  if (tid == 0) bc_total = total;
  __syncthreads();
  return bc_total;
}

// the live clusters in ascending slot order -> s.live; returns their number to every thread
// (the scans only visit live clusters, so a round costs live^2 distance evaluations, not live x n)
__device__ int rm_compact_live(int n, const AlignScratch& s) {
  __syncthreads();
  const int* size = s.size;
  if (n > 1024) return block_compact(n, s.live, [&](int k) { return size[k] > 0; });
  // small problems (their state is in LDS): the first wavefront alone, without the barriers of block_compact
  __shared__ int s_nlive;
  const int tid = threadIdx.x;
  if (tid < 64) {
    int* live = s.live;
    int c = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
      const int k = k0 + tid;
      const bool on = k < n && size[k] > 0;
      const unsigned long long m = __ballot(on);
      if (on) live[c + __popcll(m & ((1ull << tid) - 1ull))] = k;
      c += __popcll(m);
    }
    if (tid == 0) s_nlive = c;
  }
  __syncthreads();
  return s_nlive;
}

// nearest neighbour (s.chain) and merge height (s.nd) of the live clusters li = lbase0 + tid, + li_stride, ..; tile: LDS for
// ALIGN_TILE candidates (56 B each) when the state lives in memory, else null
__device__ void rm_scan(const AlignScratch& s, int nlive, int lbase0, int li_stride, double* tile) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  int* nn = s.chain;
  double* nd = s.nd;
  int* live = s.live;
  // The candidates of a scan are the SAME for every thread.  With the clustering state in memory (more selected entries
  // than the dynamic LDS holds) every candidate was three dependent round trips per wavefront -- live[lc] -> size / centroid
  // -- per wavefront.  Round 4: the live centroids stream through the idle dynamic LDS in tiles of ALIGN_TILE, loaded by all
  // threads at once and read as LDS broadcasts (-7 % at the 5 000- and 16 000-entry pair problems of a 16 x 1000 x 5 table).
  // What bounds a scan is its arithmetic, ~45 FP64-pipe instructions per candidate with every lane busy: keeping the weights
  // n_c / (n_x + n_c) of the sizes 1 .. 4 in registers instead of dividing per candidate changed nothing (measured).
  for (int lbase = lbase0; lbase < nlive; lbase += li_stride) {
    const int li = lbase + tid;
    const bool active = li < nlive;
    const int i = active ? live[li] : -1;
    double cx[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (active)
      for (int j = 0; j < 6; ++j) cx[j] = s.cen[6 * i + j];
    const double nx = active ? (double)s.size[i] : 1.0;
    // the candidates are compared on  d^2 n_c / (n_x + n_c)  -- the Ward distance squared without the factor 2 n_x that
    // is common to the scan: the same order as the distances (the square root and the factor are monotone), one division
    // and no square root per candidate; the height sqrt(2 n_x n_c / (n_x + n_c) d^2) is formed once, for the winner,
    // with the expression of the chain algorithm (bit-symmetric in the pair)
    double bkey = INFINITY, bd2 = 0.0;
    int bi = -1, bn = 1;
    auto candidate = [&](int c_any, int n_any, const double* cc) {      // (every lane looks at the SAME candidate)
      const int c = __builtin_amdgcn_readfirstlane(c_any), nci = __builtin_amdgcn_readfirstlane(n_any);
      double d2 = 0.0;
      for (int j = 0; j < 6; ++j) {
        const double dl = cx[j] - cc[j];
        d2 += dl * dl;
      }
      const double ni = (double)nci, key = d2 * ni / (nx + ni);
      if (c != i && key < bkey) { bkey = key; bi = c; bd2 = d2; bn = nci; }
    };
    if (tile != nullptr) {
      int* tile_c = reinterpret_cast<int*>(tile + 6 * (size_t)ALIGN_TILE);     // [ALIGN_TILE] slot | [ALIGN_TILE] size
      for (int t0 = 0; t0 < nlive; t0 += ALIGN_TILE) {
        const int nt = min(ALIGN_TILE, nlive - t0);
        __syncthreads();                       // (the previous tile has been read by everyone)
        for (int k = tid; k < nt; k += nthr) {
          const int c = live[t0 + k];
          tile_c[k] = c;
          tile_c[ALIGN_TILE + k] = s.size[c];
          for (int j = 0; j < 6; ++j) tile[6 * k + j] = s.cen[6 * c + j];
        }
        __syncthreads();
        for (int k = 0; k < nt; ++k) candidate(tile_c[k], tile_c[ALIGN_TILE + k], tile + 6 * k);   // ascending slots
      }
    } else if (__builtin_amdgcn_readfirstlane(lbase + (tid & ~63)) < nlive) {   // (wave-uniform: the wave owns a cluster)
      for (int lc = 0; lc < nlive; ++lc) {     // ascending slots: the first minimum is the lowest index
        const int c = __builtin_amdgcn_readfirstlane(live[lc]);
        candidate(c, s.size[c], s.cen + 6 * (size_t)c);
      }
    }
    if (active) {
      nn[i] = bi;
      const double ni = (double)bn;
      nd[i] = sqrt(2.0 * nx * ni / (nx + ni) * bd2);
    }
  }
}

// all reciprocal nearest-neighbour pairs merge; *nm counts the merges of the clustering (shared or global);
// changed (or null): changed[j] = round for every slot j that holds a merged cluster of this round
__device__ void rm_merge(int n, const AlignScratch& s, int* nm, int* changed = nullptr, int round = 0) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  int* nn = s.chain;
  double* nd = s.nd;
  // reciprocal pairs owned by this thread (cluster i < its partner), found for 64 of its entries at a time: a merge only
  // writes the two slots of its own pair, and the partner j > i of a pair never owns one (nn[j] = i < j), so the tests of a
  // later batch see exactly what they would have seen before the merges of an earlier one
  for (int base = 0; base < n; base += 64 * nthr) {
    unsigned long long mrg = 0ull;
    {
      int q = 0;
      for (int i = base + tid; i < n && q < 64; i += nthr, ++q)
        if (s.size[i] > 0) {
          const int j = nn[i];
          if (j > i && nn[j] == i) mrg |= 1ull << q;
        }
    }
    __syncthreads();
    {
      int q = 0;
      for (int i = base + tid; i < n && q < 64; i += nthr, ++q)
        if ((mrg >> q) & 1ull) {
          const int j = nn[i], m = atomicAdd(nm, 1);
          const int nx = s.size[i], ny = s.size[j];
          s.hgt[m] = nd[i];
          s.rep_a[m] = i;      // slot indices double as representatives: slot j keeps holding the merged cluster, i dies;
          s.rep_b[m] = j;      // a slot index is always a member of the cluster it holds (it is one of the original points)
          for (int k = 0; k < 6; ++k)
            s.cen[6 * j + k] = ((double)nx * s.cen[6 * i + k] + (double)ny * s.cen[6 * j + k]) / (double)(nx + ny);
          s.size[j] = nx + ny;
          s.size[i] = 0;
          if (changed != nullptr) changed[j] = round;
        }
    }
    __syncthreads();
  }
}

// cut of the dendrogram (clustered: heights / representatives of n - 1 merges are in place), labels, the most common cluster,
// its mean -> out[6] (shared)
__device__ void rm_cut_labels_mean(int n, const AlignScratch& s, int t_clust, bool clustered, double* out, double* sv, int* s_int) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (clustered) {
    // ---- cut: scipy's fcluster(criterion='maxclust') finds the smallest merge height thr that leaves at most t_clust
    // clusters and then applies EVERY merge of height <= thr (cluster_maxclust_monocrit + cluster_monocrit: for a monotone
    // Ward dendrogram the criterion is the merge height itself).  thr is the (n - t_clust)-th smallest height; merges that
    // tie with it are applied too -- exact duplicates among the relative poses (noise-free or repeated detections) give
    // zero-height merges, and the flat clusters then hold fewer than t_clust groups, exactly as in the reference.
    const int nm = n - 1, keep = n - t_clust;
    const double thr = block_kth_smallest(s.hgt, nm, keep - 1, sv);
    {
      // Every slot dies at most once (a merge moves the cluster of the smaller slot into the larger one), so the applied
      // merges form a forest of "dies into" links that can be written in parallel; the flat cluster of a point is the slot
      // its links end in.  (The first version ran a union-find without path compression on ONE thread: 37 M cycles for
      // 1300 points, more than the clustering itself.)
      for (int k = tid; k < nm; k += nthr)
        if (s.hgt[k] <= thr) s.parent[s.rep_a[k]] = s.rep_b[k];
    }
    __syncthreads();
    // pointer jumping: parent <- parent of parent, ceil(log2 n) + 1 rounds.  In place: a concurrent reader sees the old or the
    // new link of another point -- both are ancestors on the same path, so the distance to the root still at least halves
    // per round and the fixed point (every point linked to its final slot) does not depend on the interleaving.
    for (int span = 1; span < 2 * n; span *= 2) {
      for (int i = tid; i < n; i += nthr) s.parent[i] = s.parent[s.parent[i]];
      __syncthreads();
    }
  }
  // labels = final slot of every point; component sizes and smallest members; the most common cluster (ties: the cluster
  // that is met first in index order, i.e. the one with the smallest member -- collections.Counter.most_common)
  for (int i = tid; i < n; i += nthr) {
    s.size[i] = 0;
    s.chain[i] = 0x7fffffff;
  }
  __syncthreads();
  for (int i = tid; i < n; i += nthr) {
    const int r = s.parent[i];
    s.list[i] = r;
    atomicAdd(&s.size[r], 1);
    atomicMin(&s.chain[r], i);
  }
  __syncthreads();
  {   // (one thread walking n slots of memory was 0.7 ms of a 16 000-entry problem: block-wide maximum of (count, -smallest member))
    unsigned long long bestk = 0ull;
    for (int i = tid; i < n; i += nthr) {
      const int c = s.size[i];
      if (c > 0) {
        const unsigned long long key = ((unsigned long long)(unsigned)c << 32) | (unsigned long long)(0x7fffffffu - (unsigned)s.chain[i]);
        bestk = key > bestk ? key : bestk;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_down(bestk, off, 64);
      bestk = o > bestk ? o : bestk;
    }
    __shared__ unsigned long long s_best[ALIGN_THREADS / 64];
    __syncthreads();
    if ((tid & 63) == 0) s_best[tid >> 6] = bestk;
    __syncthreads();
    if (tid == 0) {
      unsigned long long b = 0ull;
      for (int w = 0; w < (nthr >> 6); ++w) b = s_best[w] > b ? s_best[w] : b;
      const int bestc = (int)(b >> 32), bestm = (int)(0x7fffffffu - (unsigned)(b & 0xffffffffull));
      s_int[0] = bestc > 0 ? s.list[bestm] : -1;      // the slot that holds the cluster of its smallest member
      s_int[1] = bestc;
    }
  }
  __syncthreads();
  {                   // mean of the most common cluster (numpy adds the rows in index order; here: fixed tree, ~1e-16 apart)
    const int best = s_int[0];
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = tid; i < n; i += nthr)
      if (s.list[i] == best)
        for (int j = 0; j < 6; ++j) acc[j] += s.vec[6 * i + j];
    for (int j = 0; j < 6; ++j) {
      double v = acc[j];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      __syncthreads();
      if ((tid & 63) == 0) sv[tid >> 6] = v;
      __syncthreads();
      if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < (nthr >> 6); ++w) t += sv[w];
        out[j] = t / (double)s_int[1];
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ int rm_num_clusters(int n) { return max((int)fmax((double)n / 10.0, 3.0), 1); }   // fcluster(t = max(n / 10, 3))

// robust mean of the n 6-vectors vec[0..n) (transform/common.py:6-21) -> out[6]; every thread of the block takes part
__device__ void robust_mean_block(int n, const AlignScratch& s, double* out /* shared [6] */, double* sv, int* si,
                                  int* s_int /* shared [4] */, long long* prof = nullptr /* [16] phase cycles of this pass */,
                                  double* tile = nullptr /* LDS, 56 B x ALIGN_TILE, when the state lives in memory */) {
  long long tprof = prof ? clock64() : 0;
#define RM_STAMP(k) if (prof != nullptr && threadIdx.x == 0) { const long long now = clock64(); prof[k] += now - tprof; tprof = now; }
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (n == 1) {
    if (tid < 6) out[tid] = s.vec[tid];
    __syncthreads();
    return;
  }
  rm_whiten_init(n, s, sv);
  RM_STAMP(5)
  const int t_clust = rm_num_clusters(n);
  if (t_clust < n) {
    // ---- Ward linkage by PARALLEL RECIPROCAL-NEAREST-NEIGHBOUR rounds -------------------------------------------------
    // scipy builds the dendrogram with the nearest-neighbour-chain algorithm (scipy/cluster/_hierarchy.pyx: nn_chain): ~3 n
    // SEQUENTIAL nearest-neighbour searches, each a block-wide scan + argmin (round 2 ran exactly that: ~10 k cycles per
    // search, 40 M cycles = 17 ms per clustering of 1300 relative poses, two clusterings per alignment -- profiled with
    // -DMCBA_EXP_ALIGN_PROF).  Ward linkage is REDUCIBLE: merging a reciprocal nearest-neighbour pair never changes the
    // nearest neighbours of the other clusters into something closer, so every RNN pair that exists at a time is a merge
    // of the final dendrogram and all of them can be applied AT ONCE.  A round = every live cluster finds its nearest
    // neighbour (thread per cluster, all threads scan the live centroids: n^2 / threads distance evaluations, throughput
    // instead of latency), the reciprocal pairs merge in parallel (they are disjoint).  A constant fraction of the clusters
    // merges per round on real data, so the work is ~2 n^2 distance evaluations in a few dozen barriers.  The tree, the slot
    // that holds a merged cluster (the larger index), the centroid arithmetic and therefore every merge height are exactly
    // those of the chain algorithm; only the ORDER in which the merges are recorded differs (atomic counter), which neither
    // the threshold cut nor the union-find below depends on.  Distances are bit-symmetric (d(i, j) == d(j, i)), ties go to
    // the lowest index: the closest pair with the smallest indices is always reciprocal, so every round merges something.
    __shared__ int s_nm;
    if (tid == 0) s_nm = 0;
    __syncthreads();
    int nm_prev = 0, rounds = 0;
    while (true) {
      ++rounds;
      const int nlive = rm_compact_live(n, s);
      rm_scan(s, nlive, 0, nthr, tile);
      __syncthreads();
      rm_merge(n, s, &s_nm);
      const int nm_now = s_nm;
      // (no merge in a round can only happen with non-finite poses -- every comparison false: stop instead of spinning;
      //  the heights left unset make the cut keep the remaining clusters apart)
      if (nm_now >= n - 1 || nm_now == nm_prev) {
        for (int m = nm_now + tid; m < n - 1; m += nthr) { s.hgt[m] = INFINITY; s.rep_a[m] = 0; s.rep_b[m] = 0; }
        break;
      }
      nm_prev = nm_now;
    }
    __syncthreads();
    RM_STAMP(6)
    if (prof != nullptr && tid == 0) prof[8] += rounds;
  }
  rm_cut_labels_mean(n, s, t_clust, t_clust < n, out, sv, s_int);
  RM_STAMP(7)
#undef RM_STAMP
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
//   ia / ib (or null): entry k of the problem is the pair (A[ia[e0 + k]], B[ib[e0 + k]]) of two pose TABLES instead of
//   (A[e0 + k], B[e0 + k]) -- mcba_align_poses_indexed: the pose table goes up once, the pair lists as 32-bit indices.
__global__ __launch_bounds__(ALIGN_THREADS) void k_align_robust(const long long* __restrict__ off, const double* __restrict__ A,
                                                                const double* __restrict__ B, const int32_t* __restrict__ ia,
                                                                const int32_t* __restrict__ ib, const uint8_t* __restrict__ mask,
                                                                double threshold, int invert, long long scratch_stride,
                                                                AlignScratch base, double* __restrict__ out,
                                                                uint8_t* __restrict__ out_valid, uint8_t* __restrict__ inliers,
                                                                int lds_cap) {
  extern __shared__ __attribute__((aligned(16))) double align_lds[];   // [lds_cap][6] centroids | [lds_cap] sizes
  __shared__ double sv[ALIGN_THREADS / 64], mean6[6], Rm[9], tm[3];
  __shared__ int si[ALIGN_THREADS / 64], s_int[4], s_cnt;
  const int p = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const long long e0 = off[p];
  const int n = (int)(off[p + 1] - e0);
  AlignScratch s = base;
  {
    const long long o = (long long)p * scratch_stride;
    s.vec += 6 * o; s.cen += 6 * o; s.err += o; s.hgt += o; s.nd += o; s.size += o; s.chain += o; s.rep_a += o; s.rep_b += o;
    s.parent += o; s.list += o; s.live += o;
  }
  const uint8_t* mp = mask ? mask + e0 : nullptr;
  auto load_pair = [&](int k, double* Ra, double* ta, double* Rb, double* tb) {
    se3_load(A + 16 * (size_t)(ia != nullptr ? (long long)ia[e0 + k] : e0 + k), Ra, ta);
    se3_load(B + 16 * (size_t)(ib != nullptr ? (long long)ib[e0 + k] : e0 + k), Rb, tb);
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
  // phase stamps (s.prof != nullptr: MCBA_ALIGN_PROF=1): prof[p][16 * pass + k], k = 0 compaction, 1 relative poses, 2 robust
  // mean (whitening + clustering + cut + mean), 3 errors, 4 quantile + outlier test; robust_mean_block adds 5 whitening,
  // 6 clustering rounds, 7 cut + labels + mean, 8 number of rounds
  long long tprev = clock64();
  int stamp_pass = 0;
#define ALIGN_STAMP(k) if (s.prof != nullptr && tid == 0) { const long long now = clock64(); s.prof[(size_t)p * 32 + 16 * stamp_pass + (k)] += now - tprev; tprev = now; }
  // pass = 0: entries of the mask; pass = 1: inliers of the outlier test
  for (int pass = 0; pass < 2; ++pass) {
    // ---- stable compaction of the selected entries ---------------------------------------------------------------
    if (tid < 64) {   // first wavefront: 64 entries per step, ranks from a ballot prefix (a one-thread loop took n dependent steps)
      int c = 0;
      for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + tid;
        const bool sel = k < n && (pass == 0 ? (mp == nullptr || mp[k] != 0) : (s.parent[k] != 0));   // (pass 1: inlier flags)
        const unsigned long long m = __ballot(sel);
        if (sel) s.list[c + __popcll(m & ((1ull << tid) - 1ull))] = k;
        c += __popcll(m);
      }
      if (tid == 0) s_cnt = c;
    }
    __syncthreads();
    stamp_pass = pass;
    ALIGN_STAMP(0)
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
    {
      // The nearest-neighbour chain of the clustering reads centroids and sizes ~3 n times each, one dependent round trip
      // per search when they live in memory (the chain is sequential: ~2.5 us per search, 30 ms for a 1400-entry pair
      // problem).  Problems whose SELECTED entries fit the workgroup's dynamic LDS keep both there.
      AlignScratch sl = s;
      if (cnt <= lds_cap) {   // centroids, heights and every integer array of the clustering (the union-find and the label
        sl.cen = align_lds;   // count are single-thread walks: at LDS latency instead of one memory round trip per step)
        sl.hgt = align_lds + 6 * (size_t)lds_cap;
        sl.nd = align_lds + 7 * (size_t)lds_cap;
        int* ib = reinterpret_cast<int*>(align_lds + 8 * (size_t)lds_cap);
        sl.size = ib;
        sl.parent = ib + lds_cap;
        sl.chain = ib + 2 * (size_t)lds_cap;
        sl.rep_a = ib + 3 * (size_t)lds_cap;
        sl.rep_b = ib + 4 * (size_t)lds_cap;
        sl.list = ib + 5 * (size_t)lds_cap;     // (labels; the caller's entry list is rebuilt by the next pass)
        sl.live = ib + 6 * (size_t)lds_cap;
      }
      ALIGN_STAMP(1)
      robust_mean_block(cnt, sl, mean6, sv, si, s_int, s.prof ? s.prof + (size_t)p * 32 + 16 * pass : nullptr,
                        cnt > lds_cap && lds_cap * 92 >= ALIGN_TILE * 56 ? align_lds : nullptr);
      ALIGN_STAMP(2)
    }
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
    ALIGN_STAMP(3)
    {   // numpy quantile 0.75, method 'linear': virtual index (n - 1) * 0.75 between two order statistics
      const double virt = (double)(n - 1) * 0.75;
      const int lo = (int)floor(virt), hi = min(lo + 1, n - 1);
      const double gamma = virt - floor(virt);
      // the two order statistics (block_kth_smallest: ranking for small problems, radix select on the bit patterns for large ones)
      double a, b;
      block_kth_pair(s.err, n, lo, hi, sv, a, b);
      const double diff = b - a;
      const double uq = gamma >= 0.5 ? b - diff * (1.0 - gamma) : a + diff * gamma;   // numpy _lerp
      __syncthreads();
      for (int k = tid; k < n; k += nthr) {
        const bool in = s.err[k] < uq * threshold && (mp == nullptr || mp[k] != 0);
        s.parent[k] = in ? 1 : 0;              // inlier flags (parent is re-initialised by the next robust mean)
        if (inliers != nullptr) inliers[e0 + k] = in ? 1 : 0;
      }
      __syncthreads();
      ALIGN_STAMP(4)
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


// ---------------------------------------------------------------------------------------------------------------
// The STAGED form of k_align_robust for batches of large problems (round 4).  The clustering of a 16 000-entry board pair is
// ~5 n^2 distance evaluations spread over ~50 rounds: one workgroup on one compute unit needs 8 M cycles for it -- 83 % of the
// 8 ms the four board pairs of a 16 x 1000 x 5 pose table take, with 252 compute units idle.  Here the pieces of robust_mean_block
// are separate launches per round -- k_align_stage_scan spreads the nearest-neighbour scans of every problem over
// gridDim.y workgroups, k_align_stage_merge applies the reciprocal pairs and compacts the live list with one workgroup per problem
// -- around k_align_stage_pre / _post (everything else of a pass).  State lives in memory (AlignScratch + AlignStage); the
// kernel boundaries are the only synchronisation (no spinning between workgroups).  The host enqueues rounds a few ahead of a
// progress word in pinned memory (as the LSMR driver does) until every problem has reported the end of its clustering.
// Same arithmetic per cluster, same merges: the results are those of k_align_robust.
// ---------------------------------------------------------------------------------------------------------------
struct AlignStage {      // per problem
  int cnt, nlive, nmerged, nm_prev, done /* clustering finished */, finished /* result written */, rounds, nwork;
  long long sum_live, sum_work;   // (statistics: clusters alive / scanning, summed over the rounds of both passes)
};
struct AlignArgs {
  const long long* off; const double* A; const double* B; const int32_t* ia; const int32_t* ib; const uint8_t* mask;
  double threshold; int invert; long long scratch_stride; AlignScratch base;
  double* out; uint8_t* out_valid; uint8_t* inliers;
  AlignStage* st; int* done_count; unsigned long long* host_progress;
  double* pkey; double* pd2; int* pidx; int* pn;   // [problem][ALIGN_SCAN_Z][scratch_stride] partial winners of a split scan
  int* changed; int* work;                         // [problem][scratch_stride] round of the last merge into a slot | clusters to scan
};
constexpr int ALIGN_SCAN_Z = 16;       // a scan is split over up to this many workgroups by CANDIDATE range (tiles of ALIGN_SCAN_TILE)
constexpr int ALIGN_SCAN_TILE = 64;    // (the selected entries of a 16 000-entry board pair are ~700: small tiles spread even those)
__device__ __forceinline__ int align_scan_splits(int nlive) {
  return max(1, min(ALIGN_SCAN_Z, (nlive + ALIGN_SCAN_TILE - 1) / ALIGN_SCAN_TILE));
}
__device__ __forceinline__ AlignScratch align_scratch_of(const AlignArgs& a, int p) {
  AlignScratch s = a.base;
  const long long o = (long long)p * a.scratch_stride;
  s.vec += 6 * o; s.cen += 6 * o; s.err += o; s.hgt += o; s.nd += o; s.size += o; s.chain += o; s.rep_a += o; s.rep_b += o;
  s.parent += o; s.list += o; s.live += o;
  return s;
}
__device__ __forceinline__ void align_load_pair(const AlignArgs& a, long long e0, int k, double* Ra, double* ta, double* Rb, double* tb) {
  se3_load(a.A + 16 * (size_t)(a.ia != nullptr ? (long long)a.ia[e0 + k] : e0 + k), Ra, ta);
  se3_load(a.B + 16 * (size_t)(a.ib != nullptr ? (long long)a.ib[e0 + k] : e0 + k), Rb, tb);
  if (a.invert) {
    double Ri[9], ti[3];
    se3_inv(Ra, ta, Ri, ti);
    for (int i = 0; i < 9; ++i) Ra[i] = Ri[i];
    for (int i = 0; i < 3; ++i) ta[i] = ti[i];
    se3_inv(Rb, tb, Ri, ti);
    for (int i = 0; i < 9; ++i) Rb[i] = Ri[i];
    for (int i = 0; i < 3; ++i) tb[i] = ti[i];
  }
}
__host__ __device__ inline unsigned long long align_progress_word(unsigned long long call, int done, int round) {
  return ((call & 0xffffull) << 48) | ((unsigned long long)(done & 0xffffff) << 24) | (unsigned long long)(round & 0xffffff);
}

// selection of the pass (0: the mask, 1: the inliers of the outlier test), relative poses, whitening, first live list
__global__ __launch_bounds__(ALIGN_THREADS) void k_align_stage_pre(AlignArgs a, int pass) {
  __shared__ double sv[ALIGN_THREADS / 64];
  __shared__ int s_cnt;
  const int p = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  AlignStage* st = a.st + p;
  if (pass == 1 && st->finished) {
    if (tid == 0) atomicAdd(a.done_count, 1);
    return;
  }
  const long long e0 = a.off[p];
  const int n = (int)(a.off[p + 1] - e0);
  const AlignScratch s = align_scratch_of(a, p);
  const uint8_t* mp = a.mask ? a.mask + e0 : nullptr;
  if (tid < 64) {
    int c = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
      const int k = k0 + tid;
      const bool sel = k < n && (pass == 0 ? (mp == nullptr || mp[k] != 0) : (s.parent[k] != 0));
      const unsigned long long m = __ballot(sel);
      if (sel) s.list[c + __popcll(m & ((1ull << tid) - 1ull))] = k;
      c += __popcll(m);
    }
    if (tid == 0) s_cnt = c;
  }
  __syncthreads();
  const int cnt = s_cnt;
  if (cnt == 0) {     // tables.relative_between: no common entry -> invalid pose (identity)
    if (tid < 16) a.out[16 * (size_t)p + tid] = (tid % 5 == 0) ? 1.0 : 0.0;
    if (tid == 0) {
      a.out_valid[p] = 0;
      st->cnt = 0; st->done = 1; st->finished = 1;
      atomicAdd(a.done_count, 1);
    }
    if (a.inliers != nullptr && pass == 0)
      for (int k = tid; k < n; k += nthr) a.inliers[e0 + k] = 0;
    return;
  }
  for (int i = tid; i < cnt; i += nthr) {   // relative poses dest . source^-1 (matrix.relative_to) as 6-vectors
    double Ra[9], ta[3], Rb[9], tb[3], Rai[9], tai[3], Rr[9], tr3[3];
    align_load_pair(a, e0, s.list[i], Ra, ta, Rb, tb);
    se3_inv(Ra, ta, Rai, tai);
    se3_mul(Rb, tb, Rai, tai, Rr, tr3);
    double w[3];
    rotvec_from_matrix(Rr, w);
    for (int j = 0; j < 3; ++j) { s.vec[6 * i + j] = w[j]; s.vec[6 * i + 3 + j] = tr3[j]; }
  }
  __syncthreads();
  int nlive = 0;
  bool cluster = false;
  if (cnt > 1) {
    rm_whiten_init(cnt, s, sv);
    cluster = rm_num_clusters(cnt) < cnt;
    if (cluster) {
      nlive = rm_compact_live(cnt, s);
      int* changed = a.changed + (size_t)p * a.scratch_stride;
      int* work = a.work + (size_t)p * a.scratch_stride;
      for (int i = tid; i < cnt; i += nthr) { changed[i] = 0; work[i] = s.live[i]; }     // (first round: every cluster scans)
    }
  }
  if (tid == 0) {
    st->cnt = cnt; st->nlive = nlive; st->nwork = nlive; st->nmerged = 0; st->nm_prev = 0; st->rounds = 0; st->finished = 0;
    st->done = cluster ? 0 : 1;
    if (!cluster) atomicAdd(a.done_count, 1);
  }
}

// nearest neighbours of the clusters on the WORK list of every problem whose clustering runs (first round: all; later: those
// whose nearest neighbour merged or that merged themselves -- Ward linkage is reducible, a merged cluster is never closer to a
// third one than its parts were, so every other cluster keeps its nearest neighbour and the distance to it: ~20 % of the live
// clusters per round).  Block (p, y, z): the clusters y * 256 + tid (+ 256 gridDim.y ..) of the list of problem p against the
// candidate tiles z, z + Z, .. of ALL live clusters (Z = align_scan_splits(nlive) <= gridDim.z): a
// first round with 3 000 live clusters is 3 000 candidates per wavefront when only the clusters are spread (208 us), and
// 140 k wavefront-candidate steps for 1 024 SIMDs when the candidates are spread too.  The partial winners (key, d^2, slot,
// size) of the splits are combined by k_align_stage_merge -- smallest key, lowest slot among equal keys: the candidate a single
// ascending scan finds.
__global__ __launch_bounds__(256) void k_align_stage_scan(AlignArgs a) {
  __shared__ double tile[6 * ALIGN_SCAN_TILE];
  __shared__ int tile_c[2 * ALIGN_SCAN_TILE];
  const int p = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const AlignStage* st = a.st + p;
  if (st->done) return;
  const int nlive = st->nlive, nwork = st->nwork, Z = align_scan_splits(nlive), z = blockIdx.z;
  if (z >= Z) return;
  const AlignScratch s = align_scratch_of(a, p);
  const size_t po = ((size_t)p * ALIGN_SCAN_Z + z) * (size_t)a.scratch_stride;
  const int* live = s.live;
  const int* work = a.work + (size_t)p * a.scratch_stride;
  for (int lbase = blockIdx.y * nthr; lbase < nwork; lbase += gridDim.y * nthr) {
    const int li = lbase + tid;
    const bool active = li < nwork;
    const int i = active ? work[li] : -1;
    double cx[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (active)
      for (int j = 0; j < 6; ++j) cx[j] = s.cen[6 * i + j];
    const double nx = active ? (double)s.size[i] : 1.0;
    double bkey = INFINITY, bd2 = 0.0;
    int bi = -1, bn = 1;
    for (int t0 = z * ALIGN_SCAN_TILE; t0 < nlive; t0 += Z * ALIGN_SCAN_TILE) {
      const int nt = min(ALIGN_SCAN_TILE, nlive - t0);
      __syncthreads();                       // (the previous tile has been read by everyone)
      for (int k = tid; k < nt; k += nthr) {
        const int c = live[t0 + k];
        tile_c[k] = c;
        tile_c[ALIGN_SCAN_TILE + k] = s.size[c];
        for (int j = 0; j < 6; ++j) tile[6 * k + j] = s.cen[6 * c + j];
      }
      __syncthreads();
      for (int k = 0; k < nt; ++k) {         // ascending slots; the expressions of rm_scan
        const int c = __builtin_amdgcn_readfirstlane(tile_c[k]), nci = __builtin_amdgcn_readfirstlane(tile_c[ALIGN_SCAN_TILE + k]);
        const double* cc = tile + 6 * k;
        double d2 = 0.0;
        for (int j = 0; j < 6; ++j) {
          const double dl = cx[j] - cc[j];
          d2 += dl * dl;
        }
        const double ni = (double)nci, key = d2 * ni / (nx + ni);
        if (c != i && key < bkey) { bkey = key; bi = c; bd2 = d2; bn = nci; }
      }
    }
    if (active) {
      a.pkey[po + li] = bkey;
      a.pd2[po + li] = bd2;
      a.pidx[po + li] = bi;
      a.pn[po + li] = bn;
    }
  }
}

// reciprocal pairs merge, end test, next live list; block 0 reports the progress of the batch to the host
__global__ __launch_bounds__(ALIGN_THREADS) void k_align_stage_merge(AlignArgs a, unsigned long long call, int round) {
  __shared__ int s_now;
  const int p = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  AlignStage* st = a.st + p;
  if (p == 0 && tid == 0)
    __hip_atomic_store(a.host_progress, align_progress_word(call, atomicAdd(a.done_count, 0), round), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  if (st->done) return;
  const AlignScratch s = align_scratch_of(a, p);
  const int cnt = st->cnt;
  {   // nearest neighbour of every live cluster from the partial winners of the split scan
    const int nlive = st->nlive, nwork = st->nwork, Z = align_scan_splits(nlive);
    const int* work = a.work + (size_t)p * a.scratch_stride;
    for (int li = tid; li < nwork; li += nthr) {
      const int i = work[li];
      double bkey = INFINITY, bd2 = 0.0;
      int bi = -1, bn = 1;
      for (int z = 0; z < Z; ++z) {
        const size_t o = ((size_t)p * ALIGN_SCAN_Z + z) * (size_t)a.scratch_stride + li;
        const double key = a.pkey[o];
        const int c = a.pidx[o];
        if (c >= 0 && (key < bkey || (key == bkey && c < bi))) { bkey = key; bi = c; bd2 = a.pd2[o]; bn = a.pn[o]; }
      }
      const double nx = (double)s.size[i], ni = (double)bn;
      s.chain[i] = bi;
      s.nd[i] = sqrt(2.0 * nx * ni / (nx + ni) * bd2);
    }
    __syncthreads();
  }
  int* changed = a.changed + (size_t)p * a.scratch_stride;
  rm_merge(cnt, s, &st->nmerged, changed, round);
  if (tid == 0) s_now = atomicAdd(&st->nmerged, 0);
  __syncthreads();
  const int nm_now = s_now;
  // (no merge in a round can only happen with non-finite poses: stop instead of spinning; see robust_mean_block)
  if (nm_now >= cnt - 1 || nm_now == st->nm_prev) {
    for (int m = nm_now + tid; m < cnt - 1; m += nthr) { s.hgt[m] = INFINITY; s.rep_a[m] = 0; s.rep_b[m] = 0; }
    __syncthreads();
    if (tid == 0) {
      st->done = 1;
      atomicAdd(a.done_count, 1);
    }
    return;
  }
  const int nlive = rm_compact_live(cnt, s);
  // the clusters that scan in the next round: merged in this one, or their nearest neighbour died / merged
  int* work = a.work + (size_t)p * a.scratch_stride;
  const int* live = s.live;
  // (block_compact returns positions k of the live list: translate them to slots afterwards)
  const int s_nwork = block_compact(nlive, work, [&](int k) {
    const int x = live[k], y = s.chain[x];
    return changed[x] == round || y < 0 || s.size[y] == 0 || changed[y] == round;
  });
  for (int k = tid; k < s_nwork; k += nthr) work[k] = live[work[k]];
  __syncthreads();
  if (tid == 0) {
    st->nm_prev = nm_now; st->nlive = nlive; st->nwork = s_nwork; st->rounds += 1;
    st->sum_live += nlive; st->sum_work += s_nwork;
  }
}

// cut, labels, mean of the most common cluster -> transform; pass 0: errors, upper quartile, outlier test; pass 1: the result
__global__ __launch_bounds__(ALIGN_THREADS) void k_align_stage_post(AlignArgs a, int pass) {
  __shared__ double sv[ALIGN_THREADS / 64], mean6[6], Rm[9], tm[3];
  __shared__ int s_int[4];
  const int p = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  AlignStage* st = a.st + p;
  if (st->finished) return;
  const long long e0 = a.off[p];
  const int n = (int)(a.off[p + 1] - e0);
  const AlignScratch s = align_scratch_of(a, p);
  const uint8_t* mp = a.mask ? a.mask + e0 : nullptr;
  const int cnt = st->cnt;
  if (cnt == 1) {
    if (tid < 6) mean6[tid] = s.vec[tid];
    __syncthreads();
  } else {
    const int t_clust = rm_num_clusters(cnt);
    rm_cut_labels_mean(cnt, s, t_clust, t_clust < cnt, mean6, sv, s_int);
  }
  if (tid == 0) rtvec_to_matrix4(mean6, Rm, tm);
  __syncthreads();
  if (pass == 0) {
    for (int k = tid; k < n; k += nthr) {   // errors of ALL entries |m A_k - B_k|_F (matrix.py:135-153)
      double Ra[9], ta[3], Rb[9], tb[3], Rr[9], tr3[3];
      align_load_pair(a, e0, k, Ra, ta, Rb, tb);
      se3_mul(Rm, tm, Ra, ta, Rr, tr3);
      double e2 = 0.0;
      for (int i = 0; i < 9; ++i) e2 += (Rr[i] - Rb[i]) * (Rr[i] - Rb[i]);
      for (int i = 0; i < 3; ++i) e2 += (tr3[i] - tb[i]) * (tr3[i] - tb[i]);
      s.err[k] = sqrt(e2);
    }
    __syncthreads();
    const double virt = (double)(n - 1) * 0.75;      // numpy quantile 0.75, method 'linear'
    const int lo = (int)floor(virt), hi = min(lo + 1, n - 1);
    const double gamma = virt - floor(virt);
    double qa, qb;
    block_kth_pair(s.err, n, lo, hi, sv, qa, qb);
    const double diff = qb - qa;
    const double uq = gamma >= 0.5 ? qb - diff * (1.0 - gamma) : qa + diff * gamma;   // numpy _lerp
    for (int k = tid; k < n; k += nthr) {
      const bool in = s.err[k] < uq * a.threshold && (mp == nullptr || mp[k] != 0);
      s.parent[k] = in ? 1 : 0;              // inlier flags (parent is re-initialised by the next robust mean)
      if (a.inliers != nullptr) a.inliers[e0 + k] = in ? 1 : 0;
    }
    return;
  }
  if (tid == 0) {
    double Ro[9], to[3];
    if (a.invert) se3_inv(Rm, tm, Ro, to);
    else { for (int i = 0; i < 9; ++i) Ro[i] = Rm[i]; for (int i = 0; i < 3; ++i) to[i] = tm[i]; }
    double* o = a.out + 16 * (size_t)p;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) o[4 * i + j] = Ro[3 * i + j];
      o[4 * i + 3] = to[i];
    }
    o[12] = o[13] = o[14] = 0.0;
    o[15] = 1.0;
    a.out_valid[p] = 1;
    st->finished = 1;
  }
}

}  // namespace mcba
