// mcba_api.hip -- C ABI (include/mcba.h) + host driver of the MI355X bundle-adjustment back-end.
//
// Host side of the path that replaces Calibration.bundle_adjust (multical/optimization/calibration.py:199-212):
//   * lowering of the flat problem description to frame-major device tables (mcba_create)
//   * launch sequences for evaluate() / Jacobian / fused normal equations
//   * the trust-region loop.  It mirrors scipy's `trf_no_bounds` (scipy/optimize/_lsq/trf.py:401-560), the solver
//     the reference calls at calibration.py:209-210, step for step -- x_scale='jac' column scaling, Cauchy-step
//     regularisation, 2-D subspace {g, regularised Gauss-Newton step}, radius update, ftol/xtol/gtol tests -- with
//     ONE substitution: the regularised Gauss-Newton step that scipy gets from LSMR on a finite-difference sparse
//     Jacobian is computed exactly from the analytic normal equations (Schur elimination of the per-frame pose blocks
//     + dense Cholesky of the reduced system), all on the GPU.  Only 2x2 algebra and control flow run on the host.
#include <hip/hip_runtime.h>
#include <atomic>
#include <algorithm>
#include <array>
#include <dlfcn.h>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mcba.h"
#include "mcba_debug.h"
#include "mcba_camops.h"
#include "mcba_lower.h"
#include "mcba_solver_kernels.h"
#include "mcba_init_kernels.h"

using namespace mcba;

namespace {

thread_local std::string g_error;

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define HIP_OK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) throw Error(std::string(#expr) + " failed: " + hipGetErrorString(e_));        \
  } while (0)

#define REQUIRE(cond, msg)                 \
  do {                                     \
    if (!(cond)) throw Error(msg);         \
  } while (0)

thread_local hipStream_t g_fill_stream = nullptr;   // stream of the running API call (see DevBuf::alloc)

// Experiment / path-forcing switches (A/B variants of kernels, forcing the code paths of very large rigs on small fixtures,
// solver traces).  The PRODUCT library does not read them from the environment: it behaves the same in every process.  Tests
// and the profiling scripts set them through mcba_debug_set_switch (mcba_debug.h) BEFORE the first handle is created (most are
// latched on first use); a library built with -DMCBA_ENV_SWITCHES (python -m multical_amd.build with MCBA_BUILD_VARIANT set)
// reads the environment as well.  Environment variables the product does honour: MCBA_NO_MFMA (validation accumulate),
// MCBA_CACHE_MB (parked device memory), MCBA_TIMING (host-phase timing lines), and MCBA_NO_NATIVE_RCCL on the Python side.
static std::map<std::string, std::string>& dbg_switch_table() {
  static std::map<std::string, std::string> t;
  return t;
}
static std::mutex g_dbg_switch_mutex;
static const char* dbg_switch(const char* name) {
  {
    std::lock_guard<std::mutex> lock(g_dbg_switch_mutex);
    auto& t = dbg_switch_table();
    auto it = t.find(name);
    if (it != t.end()) return it->second.c_str();   // (entries are never erased: the pointer stays valid)
  }
#if defined(MCBA_ENV_SWITCHES)
  return getenv(name);
#else
  return nullptr;
#endif
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (function, device) pair, not to a handle: keep a process-wide
// high-water mark per pair and only ever RAISE it, so that a second live handle with a smaller requirement cannot lower the
// limit under the first one (two handles stay alive in the Python-side cache).
static void raise_dynamic_lds(const void* fn, int device, size_t bytes) {
  static std::mutex m;
  static std::map<std::pair<const void*, int>, size_t> set;
  if (bytes <= 48 * 1024) return;   // the default limit
  std::lock_guard<std::mutex> lock(m);
  size_t& cur = set[{fn, device}];
  if (bytes <= cur) return;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) throw std::runtime_error(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: ") + hipGetErrorString(e));
  cur = bytes;
}

// ---------------------------------------------------------------------------------------------------------------
// Process-wide cache of the resources a handle owns.  A Workspace.calibrate creates one handle per Calibration and
// destroys it afterwards: ~50 hipMalloc / hipFree pairs, four pinned host buffers and a stream cost more than the three
// bundle adjustments they serve (4 ms of a 15 ms calibrate at the north-star rig).  mcba_destroy parks them here (its
// stream is idle by then) and the next mcba_create of a problem of the same shape takes them back, exact size match only.
// Bounded (MCBA_CACHE_MB of device memory, default 512 MB; 0 = off), emptied by the allocator before it reports
// out-of-memory; mcba_release_cached_memory() returns everything to the runtime.
// ---------------------------------------------------------------------------------------------------------------
struct ResourceCache {
  // Device memory parked for the next handle of the same shape: MCBA_CACHE_MB (default 512; 0 disables parking -- every
  // mcba_destroy then returns its memory to the runtime).  A process that shares the GPU with torch keeps at most this
  // much idle, and an allocation that fails with out-of-memory empties the cache and is retried once (DevBuf::alloc).
  static size_t cache_bytes_max() {
    static const size_t cap = [] {
      const char* e = getenv("MCBA_CACHE_MB");
      const long mb = e ? atol(e) : 512;
      return (size_t)(mb > 0 ? mb : 0) << 20;
    }();
    return cap;
  }
  static constexpr size_t HOST_BYTES_MAX = 64ull << 20;
  std::mutex m;
  std::multimap<std::pair<int, size_t>, void*> dev, host;   // (device, bytes) -> pointer
  std::vector<std::pair<int, hipStream_t>> streams, side_streams;
  size_t dev_bytes = 0, host_bytes = 0;
  static int device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
  }
  void* take(std::multimap<std::pair<int, size_t>, void*>& pool, size_t& total, size_t bytes) {
    std::lock_guard<std::mutex> lock(m);
    auto it = pool.find({device(), bytes});
    if (it == pool.end()) return nullptr;
    void* p = it->second;
    pool.erase(it);
    total -= bytes;
    return p;
  }
  // (resources are parked under the device of the handle that owned them, which need not be the caller's current device)
  static int& park_device() {
    static thread_local int d = -1;
    return d;
  }
  bool park(std::multimap<std::pair<int, size_t>, void*>& pool, size_t& total, size_t cap, void* p, size_t bytes) {
    std::lock_guard<std::mutex> lock(m);
    if (total + bytes > cap) return false;
    pool.insert({{park_device() >= 0 ? park_device() : device(), bytes}, p});
    total += bytes;
    return true;
  }
  void* take_dev(size_t bytes) { return take(dev, dev_bytes, bytes); }
  void* take_host(size_t bytes) { return take(host, host_bytes, bytes); }
  bool park_dev(void* p, size_t bytes) { return park(dev, dev_bytes, cache_bytes_max(), p, bytes); }
  bool park_host(void* p, size_t bytes) { return cache_bytes_max() > 0 && park(host, host_bytes, HOST_BYTES_MAX, p, bytes); }
  // (side = the non-blocking second stream of a handle, which carries the trial cost beside the speculative linearisation:
  //  its own pool -- the main stream must stay a blocking stream, the synchronous copies of the API rely on that.  Creating
  //  it anew cost every mcba_create 2-3 ms.)
  hipStream_t take_stream(bool side = false) {
    std::lock_guard<std::mutex> lock(m);
    auto& pool = side ? side_streams : streams;
    const int d = device();
    for (size_t i = 0; i < pool.size(); ++i)
      if (pool[i].first == d) {
        hipStream_t s = pool[i].second;
        pool.erase(pool.begin() + i);
        return s;
      }
    return nullptr;
  }
  bool park_stream(hipStream_t s, bool side = false) {
    std::lock_guard<std::mutex> lock(m);
    auto& pool = side ? side_streams : streams;
    if (pool.size() >= 4 || cache_bytes_max() == 0) return false;
    pool.push_back({park_device() >= 0 ? park_device() : device(), s});
    return true;
  }
  void clear() {
    std::lock_guard<std::mutex> lock(m);
    for (auto& e : dev) (void)hipFree(e.second);
    for (auto& e : host) (void)hipHostFree(e.second);
    for (auto& e : streams) (void)hipStreamDestroy(e.second);
    for (auto& e : side_streams) (void)hipStreamDestroy(e.second);
    dev.clear(); host.clear(); streams.clear(); side_streams.clear();
    dev_bytes = host_bytes = 0;
  }
};
ResourceCache& resource_cache() {
  static ResourceCache* c = new ResourceCache();   // (never destroyed: the HIP runtime may be gone at exit)
  return *c;
}
thread_local bool g_park_on_release = false;   // set by mcba_destroy while the handle's members go away

// pinned host memory through the cache (hipHostMalloc is ~0.2 ms a call)
void* pinned_alloc(size_t bytes) {
  bytes = std::max<size_t>(bytes, 8);
  if (void* p = resource_cache().take_host(bytes)) return p;
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes);
  if (e == hipErrorOutOfMemory) {   // parked buffers are the first thing to give back
    (void)hipGetLastError();
    resource_cache().clear();
    e = hipHostMalloc(&p, bytes);
  }
  if (e != hipSuccess) throw std::runtime_error(std::string("hipHostMalloc failed: ") + hipGetErrorString(e));
  return p;
}
void pinned_free(void* p, size_t bytes) {
  bytes = std::max<size_t>(bytes, 8);
  if (p == nullptr) return;
  if (!(g_park_on_release && resource_cache().park_host(p, bytes))) (void)hipHostFree(p);
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  static size_t bytes_of(size_t count) { return (std::max<size_t>(count, 1) * sizeof(T) + 511) / 512 * 512; }
  void release() {
    if (p && !(g_park_on_release && resource_cache().park_dev(p, bytes_of(n)))) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count, bool zero = true) {
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (p == nullptr || n < count) {   // (a buffer that is large enough is reused: hipFree / hipMalloc cost ~100 us each)
      release();
      n = count;
      p = (T*)resource_cache().take_dev(bytes_of(count));
      if (p == nullptr) {
        hipError_t e = hipMalloc((void**)&p, bytes_of(count));
        if (e == hipErrorOutOfMemory) {   // memory parked by destroyed handles must never cause an out-of-memory failure
          (void)hipGetLastError();
          resource_cache().clear();
          e = hipMalloc((void**)&p, bytes_of(count));
        }
        if (e != hipSuccess) {
          p = nullptr;
          n = 0;
          throw Error(std::string("hipMalloc of ") + std::to_string(bytes_of(count)) + " bytes failed: " + hipGetErrorString(e));
        }
      }
    }
    if (zero) {
      // zero-fill ON THE HANDLE'S STREAM (g_fill_stream is set by every API entry that allocates): ordered against the
      // kernels that use the buffer, no host synchronisation per buffer (mcba_create makes ~40 of them).  Without a
      // stream: hipMemset on the NULL stream is asynchronous w.r.t. the host and not ordered against a non-blocking
      // stream (torch.cuda.Stream) -> wait.
      if (g_fill_stream != nullptr) {
        HIP_OK(hipMemsetAsync(p, 0, bytes, g_fill_stream));
      } else {
        HIP_OK(hipMemset(p, 0, bytes));
        HIP_OK(hipStreamSynchronize(nullptr));
      }
    }
  }
  template <typename V>
  void upload(const V& h) {   // std::vector<T> or SlotVec<T>
    alloc(h.size(), h.empty());
    if (!h.empty()) HIP_OK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  }
};

constexpr int SEL_NEXT_BLOCKS = 256;    // workgroups (= per-block partials) of k_selm_next
constexpr int COST_BLOCKS_MAX = 4608;   // persistent single-wave workgroups of k_cost (see LIN_GRID_MAX; one view each at cfg3)
// Scalars and per-block partial sums fetched by the trust-region driver: ONE contiguous device-to-host copy per
// iteration.  Partials are folded in a fixed order by the one-wave driver kernels or by the host, which saves the tiny
// final-sum kernels and device-to-device copies of a latency-bound loop (every launch costs ~5 us of GPU timeline).
// Frame-sharded handles all-reduce the H-dependent partial arrays element-wise before they are folded.
constexpr int N_SCALARS = 32;
constexpr int Q00_BLOCKS = 512;
constexpr int SHARD_MAX_WORLD = 64;   // ranks of a frame-sharded problem (one node has 8)
struct ScalLayout {
  int nvb = 0;        // blocks of the element-wise vector kernels, ceil(n / 256)
  int vs = 0;         // [3 nvb]           k_vec_scale partials
  int q00p = 0;       // [Q00_BLOCKS]      k_q00 partials (all-reduced element-wise when sharded)
  int step = 0;       // [3 nvb]           k_vec_step partials
  int costp = 0;      // [COST_BLOCKS_MAX] k_cost partials (all-reduced element-wise when sharded)
  int dotp = 0;       // [3 nblk + 1]      partial dots + pivot report (folded by k_vec_step)
  int fold = 0;       // [4]               totals of the vs / q00p partials of a point scaled ahead (k_fold_tr)
  // frame-sharded handles: the per-rank partials of a message, gathered by summation (block r = rank r, k_shard_fold*)
  int shard2 = 0;     // [4 W]             [|g|_inf, |g_h|^2, |x scale|^2] x W | curvature x W
  int shard4 = 0;     // [3 W + 1]         {g_h.g_h, g_h.gn, gn.gn} x W | pivot report
  int total = 0;
  void init(int n, int dot_blocks) {
    nvb = (n + 255) / 256;
    vs = N_SCALARS;
    q00p = vs + (3 * nvb + 1) / 2 * 2;
    step = q00p + Q00_BLOCKS;
    costp = step + (3 * nvb + 1) / 2 * 2;
    dotp = costp + COST_BLOCKS_MAX;
    fold = dotp + (3 * dot_blocks + 8 + 1) / 2 * 2;
    shard2 = fold + 4;
    shard4 = shard2 + 4 * SHARD_MAX_WORLD;
    total = shard4 + 3 * SHARD_MAX_WORLD + 2;
  }
};

}  // namespace

namespace { void destroy_rccl_comm(void* comm); }

struct mcba_handle_s {
  Dims d{};
  Tables t{};
  const CamOps* ops = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool use_mfma = true;
  bool shard_root = true;
  DevBuf<double> comm;   // frame-sharded handles: [g_s | diag_s | cost, count | step norms] of the linearisation's message
  // solver "lsmr": m-vectors u (bidiagonalisation), J_h g_h, J_h gn; per-view partials of J_h^T u; n-vectors v, v_raw, h, hbar, x
  DevBuf<double> ls_u, ls_ua, ls_ub, ls_part, ls_v, ls_vraw, ls_h, ls_hbar, ls_x, ls_nrm, ls_partial, ls_out, ls_bpart, ls_comm, ls_xpart, ls_vpart, ls_part2;
  // compacted observation tables of the lsmr route (LsmrCompact; rebuilt when the inlier set changes: ensure_compact)
  DevBuf<double2> cp_obs, cp_bxy;
  DevBuf<double> cp_bz, cp_tr;
  DevBuf<double> dot_part;   // wavefront totals of k_dot3_part
  DevBuf<int4> cp_desc;
  bool compact_dirty = true;
  DevBuf<double> ls_cache;                // lsmr_fused == 3: the per-observation state A, X_start, X_end, t (+ robust scales) of the current linearisation
  // LSMR iteration: -1 (default) = automatic: 3 for static / hand-eye rigs, 2 for rolling shutter (measured, profiles/r06_lsmr_iteration.txt:
  // streaming the 9-double state back beats re-deriving it from the compacted tables by 2 - 6 % per product launch, the 13 doubles of a
  // rolling-shutter observation lose 17 %: 136 B per observation and iteration at 3.5 TB/s); 3 = two launches with the per-observation state cached,
  // 2 = two launches (k_lsmr_fused2 / k_lsmr_gather3), 1 = three (k_lsmr_fused), 0 = the six-launch form of round 4 (A/B, tests)
  int lsmr_fused_setting = -1;
  bool lsmr_masks_form = false;           // debug (mcba_debug_set_lsmr_masks_form): k_lsmr_fused2 reads the frame-major tables on every rig (A/B, tests)
  int lsmr_fused = 2;                     // the form in force (resolved from lsmr_fused_setting by lsmr_setup)
  ScalLayout sl;
  DevBuf<double> chol_linv;   // inverted diagonal tiles of the panel kernels (k_cholp_back)
  int lin_grid = 0;          // 0 = automatic (see lin2), > 0 = forced number of persistent workgroups (debug)

  int64_t n_inliers = 0;               // inliers of this shard
  int64_t n_evalid = 0;                // points in the mask of tables.reprojection_error (this shard)

  // observation tables
  DevBuf<double2> obs;
  DevBuf<uint8_t> inlier, evalid, fix_aspect;
  DevBuf<uint8_t> valid_fm, raw_mask, cam_valid, frame_valid, board_valid;   // Calibration.valid (frame-major), mask slabs
  DevBuf<int32_t> view_first;            // first residual pair of every view (k_view_scan)
  DevBuf<long long> totals;              // {inliers, evalid points} of the shard
  long long* h_totals = nullptr;         // pinned
  DevBuf<int32_t> obs_index, view_count, active_views, av_counts, work_counter, board_off, full2act;
  DevBuf<double> xfull, bwg, img_h, board_points, pose, cam, view, tmat;
  DevBuf<uint16_t> tri;
  DevBuf<int4> ftab;   // frame_table(d): what a frame block of k_assemble sums and where it goes
  int nftab = 0;
  DevBuf<long long> dbg;
  DevBuf<double> err_fm, sel_f64;
  std::vector<double> err_x;           // parameter vector the frame-major error table err_fm was evaluated at
  bool err_valid = false;
  DevBuf<unsigned int> sel_hist;
  DevBuf<unsigned long long> sel_state;   // SelState x SEL_MAX | ranks | per-block (count, next) of k_selm_next
  bool obs_index_dirty = false;
  bool view_first_dirty = false;         // view_first (first residual index per view) is stale w.r.t. the inlier table

  // linearisation
  DevBuf<double> rec, partial, Hss, Hfs, Hff, gbuf;   // gbuf = [g (n) | diag (n) | cost, count]
  int nchunk = 1;
  // solver state
  DevBuf<double> x, xnew, scale_inv, dsc, gh, gn, scal, costpart, Lf, W, yf, P, sbuf, ps;
  long long lsmr_iterations_last = 0;     // LSMR iterations of the last solve_lsmr (mcba_debug_lsmr_info)
  int lsmr_grid = 2048;                   // persistent single-wave workgroups of the LSMR product kernels (mcba_debug_set_lsmr_grid: grid experiments)
  // one row per LSMR call of the last solve_lsmr (mcba_debug_lsmr_trace): scipy's return tuple of `lsmr` beside the trust-region
  // quantities the call was made with
  struct LsmrCall { double tr_iteration, damp, Delta, istop, itn, normr, normar, normA, condA, normx; };
  std::vector<LsmrCall> lsmr_trace;
  bool lsmr_trace_scalars = false;        // also fetch normr .. normx of every call from the state block (one small copy + wait per call)
  DevBuf<double> ls_state;                // solver = "lsmr": scalar state of the running LSMR solve (mcba_lsmr.h)
  unsigned long long ls_call = 0;         // ... and the number of the solve (tag of its progress word, h_pub_seq[1])
  DevBuf<double> scale_inv2, dsc2, gh2;   // scaling of a trial point, computed speculatively (mcba_solve) and swapped in on acceptance
  DevBuf<int32_t> info;
  double* h_scal = nullptr;   // pinned
  unsigned long long* h_pub_seq = nullptr;   // pinned: sequence number of the last k_publish (see there)
  unsigned long long pub_seq = 0;
  double* h_x = nullptr;      // pinned staging of x uploads [n]
  double* h_gbuf = nullptr;   // pinned landing zone of [g | diag | cost, count]
  int ntile = 0, ksplit = 1, cost_blocks = 1;

  // outputs staging
  DevBuf<double> out_r, out_big;
  DevBuf<uint8_t> out_valid;
  DevBuf<int32_t> out_cols;

  // ragged camera blocks (mcba_problem.camera_n_dist): maps between the caller's parameter vector and the padded one
  std::vector<int32_t> ext2int;      // empty = identity
  int n_ext = 0;                     // length of the caller's vector (== d.n without a map)
  DevBuf<uint32_t> cam_kmask;
  DevBuf<int32_t> int2ext;
  std::vector<double> xmap_tmp;      // staging of mapped outputs

  mcba_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  // collectives issued through call_allreduce since the last mcba_allreduce_stats(reset): count, doubles moved and the
  // sizes of the first calls in issue order (the sequence the world-2 tests assert)
  int64_t ar_calls = 0, ar_doubles = 0;
  std::vector<int64_t> ar_trace;
  size_t ar_trace_cap = 4096;   // sizes kept for mcba_allreduce_stats (a long-lived handle must not grow; tests raise it: mcba_debug_set_allreduce_trace)
  void* rccl_comm = nullptr;   // ncclComm_t of the native all-reduce path (mcba_rccl_init)
  mcba_log_fn log = nullptr;
  void* log_ctx = nullptr;

  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fetch = nullptr;
  // side stream of the trust-region driver: the trial cost (k_cost + the scalar copy) of a single-GPU solve runs beside the
  // speculative linearisation of the same point instead of in front of it
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_side = nullptr;

  size_t h_scal_bytes = 0, h_x_bytes = 0, h_gbuf_bytes = 0, h_totals_bytes = 0;   // sizes of the pinned buffers
  ~mcba_handle_s() {
    pinned_free(h_scal, h_scal_bytes);
    pinned_free(h_pub_seq, 64);
    pinned_free(h_x, h_x_bytes);
    pinned_free(h_gbuf, h_gbuf_bytes);
    pinned_free(h_totals, h_totals_bytes);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev_fetch) (void)hipEventDestroy(ev_fetch);
    if (ev_side) (void)hipEventDestroy(ev_side);
    if (stream2 && !(g_park_on_release && resource_cache().park_stream(stream2, true))) (void)hipStreamDestroy(stream2);
    if (rccl_comm) destroy_rccl_comm(rccl_comm);
    if (own_stream && stream && !(g_park_on_release && resource_cache().park_stream(stream))) (void)hipStreamDestroy(stream);
  }
  double* g() { return gbuf.p; }
  double* diag() { return gbuf.p + d.n; }
  double* costcount() { return gbuf.p + 2 * (size_t)d.n; }
};

namespace {

const CamOps* pick_ops(int model, int nd) {   // model: Dims.fisheye (0 pinhole, 1 fisheye, 2 mixed)
  if (model == 2) return cam_ops_mix14();
  if (model == MCBA_CAMERA_FISHEYE) {
    REQUIRE(nd == 4, "fisheye cameras carry 4 distortion coefficients (camera_fisheye.py:113-117)");
    return cam_ops_fish4();
  }
  switch (nd) {
    case 4: return cam_ops_pin4();
    case 5: return cam_ops_pin5();
    case 8: return cam_ops_pin8();
    case 12: return cam_ops_pin12();
    case 14: return cam_ops_pin14();
  }
  throw Error("pinhole cameras carry 4, 5, 8, 12 or 14 distortion coefficients (cv2.projectPoints)");
}

void refresh_active_views(mcba_handle_s* h) {
  const int nv = h->d.views(), nblk = std::max(1, (nv + AV_THREADS - 1) / AV_THREADS);
  if (h->av_counts.n < (size_t)nblk * AV_CLASSES) h->av_counts.alloc((size_t)nblk * AV_CLASSES, false);
  hipLaunchKernelGGL(k_active_count, dim3(nblk), dim3(AV_THREADS), 0, h->stream, nv, h->view_count.p, h->av_counts.p);
  hipLaunchKernelGGL(k_active_scatter, dim3(nblk), dim3(AV_THREADS), 0, h->stream, nv, h->view_count.p, h->av_counts.p,
                     h->active_views.p);
}

// upload the shard's frames of a [C,F,B,P] host array (element size esz bytes) as C contiguous slabs [C][Fl][B][P]
void upload_shard_slabs(mcba_handle_s* h, void* dst, const void* src, size_t esz) {
  const Dims& d = h->d;
  const size_t slab = (size_t)d.Fl * d.B * d.P * esz, cam = (size_t)d.F * d.B * d.P * esz, off = (size_t)d.f0 * d.B * d.P * esz;
  if (slab == 0) return;
  if (d.Fl == d.F) {
    HIP_OK(hipMemcpyAsync(dst, src, slab * d.C, hipMemcpyHostToDevice, h->stream));
    return;
  }
  for (int c = 0; c < d.C; ++c)
    HIP_OK(hipMemcpyAsync((char*)dst + c * slab, (const char*)src + c * cam + off, slab, hipMemcpyHostToDevice, h->stream));
}

// per-view first residual index + shard totals (inliers, evalid points) -> host copies; one synchronisation
void scan_views(mcba_handle_s* h) {
  const Dims& d = h->d;
  hipLaunchKernelGGL(k_view_scan, dim3(1), dim3(1024), 0, h->stream, d, h->view_count.p, (const int32_t*)nullptr,
                     h->view_first.p, h->totals.p);
  HIP_OK(hipMemcpyAsync(h->h_totals, h->totals.p, sizeof(long long), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  REQUIRE(h->h_totals[0] < (1LL << 30), "too many observations for 32-bit residual indices");
  h->n_inliers = h->h_totals[0];
  h->view_first_dirty = false;
}

// (re)build inlier table and per-view counts of the shard ON THE DEVICE; mask in reference order or null (= valid).
// The residual ordering (obs_index) is rebuilt lazily by ensure_obs_index.
void build_inliers(mcba_handle_s* h, const uint8_t* mask_ref) {
  const Dims& d = h->d;
  if (d.views() > 0) {
    if (mask_ref != nullptr) {
      if (h->raw_mask.n < (size_t)d.slots()) h->raw_mask.alloc((size_t)d.slots(), false);
      upload_shard_slabs(h, h->raw_mask.p, mask_ref, 1);
      hipLaunchKernelGGL(k_lower_view, dim3(d.views()), dim3(64), 0, h->stream, d, (const double2*)nullptr, (const float2*)nullptr,
                         (const uint8_t*)h->raw_mask.p, (const uint8_t*)h->raw_mask.p, h->cam_valid.p, h->frame_valid.p,
                         h->board_valid.p, h->board_off.p, (double2*)nullptr, (uint8_t*)nullptr, (uint8_t*)nullptr,
                         h->inlier.p, h->view_count.p, (int32_t*)nullptr);
    } else {
      hipLaunchKernelGGL(k_inliers_from_valid, dim3(d.views()), dim3(64), 0, h->stream, d, h->valid_fm.p, h->inlier.p,
                         h->view_count.p);
    }
  }
  refresh_active_views(h);
  scan_views(h);
  h->out_r.alloc((size_t)std::max<int64_t>(2 * h->n_inliers, 1), false);
  h->obs_index_dirty = true; h->compact_dirty = true;
}

void set_loss(mcba_handle_s* h, const mcba_options* opt) {
  h->d.loss = opt ? opt->loss : MCBA_LOSS_LINEAR;
  h->d.f_scale = opt ? opt->f_scale : 1.0;
  REQUIRE(h->d.loss >= 0 && h->d.loss <= 4, "unknown loss");
  REQUIRE(h->d.loss == 0 || h->d.f_scale > 0, "f_scale must be positive");
}

// x goes up through a pinned staging buffer (a pageable source makes the runtime stage and synchronise internally).
// The previous upload from the buffer has completed: every API entry synchronises the stream before it returns.
void upload_x(mcba_handle_s* h, const double* x, double* dst) {
  if (h->ext2int.empty()) {
    memcpy(h->h_x, x, (size_t)h->d.n * sizeof(double));
  } else {   // the caller's cameras block is ragged: scatter into the padded layout (absent coefficients are zero)
    memset(h->h_x, 0, (size_t)h->d.n * sizeof(double));
    for (int i = 0; i < h->n_ext; ++i) h->h_x[h->ext2int[i]] = x[i];
  }
  HIP_OK(hipMemcpyAsync(dst, h->h_x, (size_t)h->d.n * sizeof(double), hipMemcpyHostToDevice, h->stream));
}

// x (device) -> pose / camera / view tables
// pose / camera / board-point tables only (enough for k_cost and for launch_linearize, whose k_tmat rebuilds the view table)
void eval_pose_tables(mcba_handle_s* h, const double* dx) {
  const Dims& d = h->d;
  const int items = d.n_pose + d.C + d.B * d.P;
  hipLaunchKernelGGL(k_prep, dim3((items + 127) / 128), dim3(128), 0, h->stream, d, h->t, dx);
}

void eval_tables(mcba_handle_s* h, const double* dx) {
  const Dims& d = h->d;
  eval_pose_tables(h, dx);
  const int nv = d.views() * (d.motion == MOTION_ROLLING ? 2 : 1);
  if (nv > 0) hipLaunchKernelGGL(k_views, dim3((nv + 127) / 128), dim3(128), 0, h->stream, d, h->t);
}

// ---- native RCCL all-reduce (one process per GPU over xGMI): librccl is loaded at run time, so the library has no link-
// time dependency on it and single-GPU users never touch it.  Enumerators from rccl.h (ABI-stable): ncclSum = 0,
// ncclMax = 2, ncclFloat64 = 8, ncclSuccess = 0; the unique id is 128 opaque bytes.
struct RcclApi {
  typedef struct { char internal[128]; } UniqueId;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int version = 0;
  bool ok = false;
};
const RcclApi& rccl_api() {
  static const RcclApi api = [] {
    RcclApi a;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return a;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    a.GetVersion = (decltype(a.GetVersion))dlsym(lib, "ncclGetVersion");
    // The enumerator values hard-coded above (ncclSum = 0, ncclMax = 2, ncclFloat64 = 8) and the by-value 128-byte
    // ncclUniqueId are those of the NCCL 2.x ABI that RCCL ships (2.26 in this image); refuse anything else rather than
    // reduce with the wrong operator.  ncclGetVersion reports major * 10000 + minor * 100 + patch.
    if (a.GetVersion && a.GetVersion(&a.version) != 0) a.version = 0;
    const bool abi_ok = a.version >= 20000 && a.version < 30000;
    a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy && abi_ok;
    return a;
  }();
  return api;
}
void destroy_rccl_comm(void* comm) {
  if (comm && rccl_api().ok) (void)rccl_api().CommDestroy(comm);
}
int32_t rccl_allreduce_native(void* ctx, void* device_buf, size_t count, int32_t op, void* stream) {
  mcba_handle_s* h = static_cast<mcba_handle_s*>(ctx);
  const int rc = rccl_api().AllReduce(device_buf, device_buf, count, /*ncclFloat64*/ 8, op == 0 ? /*ncclSum*/ 0 : /*ncclMax*/ 2,
                                      h->rccl_comm, (hipStream_t)stream);
  return rc;
}

int call_allreduce(mcba_handle_s* h, double* buf, size_t count, int op) {
  if (!h->allreduce) return 0;
  ++h->ar_calls;
  h->ar_doubles += (int64_t)count;
  if (h->ar_trace.size() < h->ar_trace_cap) h->ar_trace.push_back(op == 0 ? (int64_t)count : -(int64_t)count);
  const int rc = h->allreduce(h->allreduce_ctx, buf, count, op, (void*)h->stream);
  if (rc != 0) throw Error("all-reduce hook failed with code " + std::to_string(rc));
  return 0;
}


// A kernel launch that the runtime rejects (too much dynamic LDS, a bad grid) does not throw by itself: it leaves an
// error behind and the stream simply lacks that kernel -- the solver would go on with stale H, g or tables.  Every
// synchronisation point of the API therefore also collects the launch status, and the launches whose LDS size / grid are
// computed at run time check right away.
void check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw Error(std::string("kernel launch failed (") + what + "): " + hipGetErrorString(e));
}

// MCBA_FUSED=0 forces the table form (k_tmat + k_linearize); default: the table-fed fused form wherever it applies
void ensure_compact(mcba_handle_s* h);
LsmrCompact compact_tables(const mcba_handle_s* h);
bool linearize_table_form() {
  static const bool table = dbg_switch("MCBA_FUSED") != nullptr && atoi(dbg_switch("MCBA_FUSED")) == 0;
  return table;
}

// the table-fed fused k_linearize at the pose / camera tables already prepared.  Round 6: its observations come from the compacted
// tables of the lsmr route (LsmrCompact, built once per inlier set; the board points are constants here: d.off_boards < 0) -- one round
// trip per view instead of two, bit-identical records.  MCBA_LIN_COMPACT=0 keeps the masks form (A/B runs, tests).
void launch_fused_linearize_kernel(mcba_handle_s* h) {
  const Dims& d = h->d;
  static const bool masks = dbg_switch("MCBA_LIN_COMPACT") != nullptr && dbg_switch("MCBA_LIN_COMPACT")[0] == '0';
  LsmrCompact cp{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (!masks) {
    ensure_compact(h);
    cp = compact_tables(h);
  }
  h->ops->linearize(d, h->t, h->stream, h->rec.p, h->tri.p, true, h->lin_grid, nullptr, h->gbuf.p, 2 * d.n + 2, h->Hss.p,
                    d.ns * d.ns, masks ? nullptr : &cp);
}

// fused residual+Jacobian -> block normal equations.  dx != nullptr: at the parameter vector dx (device); k_tmat then also
// prepares the pose / camera / board-point tables (no k_prep launch).  dx == nullptr: at the tables already prepared
// (the trial step that was just accepted ran k_prep for its cost evaluation).
void launch_linearize(mcba_handle_s* h, const double* dx) {
  const Dims& d = h->d;
  // k_tmat also zeroes [g | diag | cost] and H_ss for the assembly that follows (entries of frames owned by other ranks
  // must be zero before the cross-rank sum: they hold the previous global values after an all-reduce)
  // Table-fed fused form: no k_tmat, no That table.  The pose / camera tables of the point come from k_prep
  // (dx given) or from the tail of the k_vec_step that produced the point (dx == nullptr); k_linearize copies its view's
  // four pose entries from the table and forms the chain products / That columns itself.  Not with adjusted board points
  // (k_points reads the per-view chain table that only k_tmat / k_views write).
  // DEFAULT (MCBA_FUSED unset).  Measured on MI355X, evaluation step / LM iteration in us, table form -> table-fed fused
  // form: 8 x 500 x 2 rolling 73.6 -> 71.4 / 233 -> 225, 16 x 1000 x 5 181 -> 147 / 529 -> 481, 4 x 200 x 1 33.7 -> 31.9 /
  // 123 -> 117 -- although k_linearize itself grows (44.5 -> 47.3 us at the north-star rig: the chain products and That
  // columns are now inside it), the k_tmat launch, its That table (10 MB written + read per evaluation) and, behind a trial
  // step, any table kernel at all are gone.  MCBA_FUSED=0 forces the table form (k_tmat).
  if (!linearize_table_form() && h->use_mfma && d.off_boards < 0 && h->t.dbg == nullptr) {
    if (dx != nullptr) eval_pose_tables(h, dx);
    launch_fused_linearize_kernel(h);
    return;
  }
  const int nb_views = std::max((d.views() + TMV - 1) / TMV, 1);
  // (MCBA_TMAT_GLOBAL=1 forces the path of rigs too large for the workgroup-local pose table: tests)
  static const bool force_global = dbg_switch("MCBA_TMAT_GLOBAL") != nullptr && dbg_switch("MCBA_TMAT_GLOBAL")[0] == '1';
  const bool local = tmat_local_poses(d) <= TM_LOCAL_POSES && !force_global;
  if (dx != nullptr && !local) {   // unusual shape (cameras + boards exceed the workgroup-local pose table): separate table pass
    eval_pose_tables(h, dx);
    dx = nullptr;
  }
  const int nb_prep = dx ? (d.n_pose + d.C + d.B * d.P + TM_THREADS - 1) / TM_THREADS : 0;
  if (local)
    hipLaunchKernelGGL(k_tmat<true>, dim3(nb_views + nb_prep), dim3(TM_THREADS), 0, h->stream, d, h->t, h->gbuf.p,
                       2 * d.n + 2, h->Hss.p, d.ns * d.ns, dx, nb_views);
  else
    hipLaunchKernelGGL(k_tmat<false>, dim3(nb_views), dim3(TM_THREADS), 0, h->stream, d, h->t, h->gbuf.p, 2 * d.n + 2,
                       h->Hss.p, d.ns * d.ns, (const double*)nullptr, nb_views);
  h->ops->linearize(d, h->t, h->stream, h->rec.p, h->tri.p, h->use_mfma, h->lin_grid, nullptr, nullptr, 0, nullptr, 0, nullptr);
}

// publish_seq != 0: the kernel that forms the cost of the linearisation also writes it to h_scal[cost_slot] (pinned) and then
// the sequence number to h_pub_seq (publish_cost)
// dynamic LDS of a k_assemble frame block besides its staging slots: record offsets + the two view-rank tables
constexpr size_t ASM_LDS_MAX = 150 * 1024;
inline size_t assemble_lds_fixed(const Dims& d) {
  return (size_t)frame_entries(d) * sizeof(int) + 2 * (size_t)((d.C * d.B + 3) & ~3) * sizeof(uint16_t) + 16;
}

// step_valid (frame-sharded solver): the k_vec_step partials of the step that led to this point ride with the message
void launch_assemble(mcba_handle_s* h, unsigned long long publish_seq = 0, int cost_slot = 0, bool step_valid = false) {
  const Dims& d = h->d;
  // ([g | diag | cost] and H_ss were zeroed by k_tmat at the start of this linearisation)
  const int nfb = (d.DF > 0) ? d.Fl : 0;
  // frame blocks stage the record entries they sum in LDS, `gviews` records at a time (all C B views of the frame when they
  // fit the budget: 43.8 KB at the north-star rig)
  // (clamped to what a workgroup may ask for by default: 64 KB of dynamic LDS minus the kernel's static tables)
  static const int stage_kb = dbg_switch("MCBA_ASM_STAGE_KB") ? std::min(60, std::max(4, atoi(dbg_switch("MCBA_ASM_STAGE_KB")))) : 44;
  const int ne = frame_entries(d), cb = d.C * d.B;
  // staging slots: as many non-empty views of a frame as the budget holds (all C B when they fit); a frame with more active
  // views than slots takes several passes
  // rigs with thousands of (camera, board) pairs: the view-rank tables (4 B per pair) take LDS next to the staging slots;
  // mcba_create has checked that one slot + the tables fit what a workgroup can ask for (assemble_lds_fixed)
  const size_t fixed = assemble_lds_fixed(d);
  const size_t stage_bytes = std::min<size_t>((size_t)stage_kb * 1024, ASM_LDS_MAX - fixed);
  const int gviews = nfb ? std::max(1, std::min(cb, (int)(stage_bytes / ((size_t)ne * 8)))) : 1;
  const size_t lds = nfb ? (size_t)gviews * ne * sizeof(double) + fixed : 0;
  raise_dynamic_lds((const void*)k_assemble, h->device, lds);
  const int npair = d.C * d.B;
  // MCBA_SHARED_FINAL_BIG=1 forces the many-pairs kernel (tests).  (Both stages in ONE launch with in-kernel completion counters were
  // measured in round 6 and dropped: evaluation step + 12 us, trial step of the exact solver + 10 us at the north-star rig --
  // profiles/r06_lsmr_experiments.txt item 9; the code lived in commit 3964829.)
  static const bool force_big = dbg_switch("MCBA_SHARED_FINAL_BIG") != nullptr && dbg_switch("MCBA_SHARED_FINAL_BIG")[0] == '1';
  // (timed apart at cfg3: frame blocks alone 10.0 us, chunk sums alone 6.9 us, together 12.4 us)
  hipLaunchKernelGGL(k_assemble, dim3(nfb + d.C * d.B * h->nchunk), dim3(ASM_THREADS), lds, h->stream, d, h->t, h->rec.p, nfb, h->nchunk, gviews,
                     h->ftab.p, h->nftab, h->Hff.p, h->Hfs.p, h->g(), h->diag(), h->partial.p);
  check_launch("k_assemble");
  {
    const int pg = std::min(npair, 16);
    if (npair <= SHARED_FINAL_MAX_PAIRS && !force_big)   // pair sums in LDS
      hipLaunchKernelGGL(k_shared_final, dim3((d.rec_size + 2 + SF_ENT - 1) / SF_ENT), dim3(64 * pg), (size_t)npair * SF_ENT * sizeof(double),
                         h->stream, d, h->partial.p, h->nchunk, h->tri.p, h->Hss.p, h->g(), h->diag(), h->costcount(),
                         publish_seq ? h->h_scal + cost_slot : nullptr, h->h_pub_seq, publish_seq);
    else                                                 // more (camera, board) pairs than the LDS table holds
      hipLaunchKernelGGL(k_shared_final_big, dim3((d.rec_size + 2 + 63) / 64, (npair + 15) / 16), dim3(1024), 0, h->stream, d,
                         h->partial.p, h->nchunk, h->tri.p, h->Hss.p, h->g(), h->diag(), h->costcount(),
                         publish_seq ? h->h_scal + cost_slot : nullptr, h->h_pub_seq, publish_seq);
    check_launch("k_shared_final");
  }
  if (d.off_boards >= 0) {   // adjusted board points: their blocks of H_ss / H_fs / g (unique entries, plain stores)
    h->ops->points(d, h->t, h->stream, (d.n - d.off_boards) / 3, h->Hss.p, h->Hfs.p, h->g());
    hipLaunchKernelGGL(k_shared_diag, dim3((d.ns + 255) / 256), dim3(256), 0, h->stream, d, h->Hss.p, h->diag());
  }
  if (h->allreduce) {
    // Frame-sharded: ONE message of 2 ns + 6 doubles -- the SHARED entries of [g | diag], {cost, count} and the step norms.
    // The frame entries of g / diag are sums over the owner's observations alone: they never cross the ranks (round 3: the
    // whole [g | diag | cost], 2 n + 2 doubles with n growing in the TOTAL number of frames).
    if (d.shard_world <= 0) throw Error("frame-sharded handle without a rank: call mcba_set_shard_rank (or mcba_rccl_init)");
    const size_t nc = 2 * (size_t)d.ns + SHARD_TAIL;
    if (h->comm.n < nc) h->comm.alloc(nc, false);
    const int grid = std::max(1, std::min(64, (d.ns + 255) / 256));
    double* stepp = step_valid ? h->scal.p + h->sl.step : nullptr;
    hipLaunchKernelGGL(k_shard_pack1, dim3(grid), dim3(256), 0, h->stream, d, h->g(), h->diag(), h->costcount(), stepp, h->sl.nvb,
                       h->comm.p);
    call_allreduce(h, h->comm.p, nc, 0);
    hipLaunchKernelGGL(k_shard_unpack1, dim3(grid), dim3(256), 0, h->stream, d, h->comm.p, h->g(), h->diag(), h->costcount(),
                       stepp, h->sl.nvb);
  }
}

// frame-sharded handles: v[motion block] <- the owners' entries on every rank (an all-gather written as a sum of vectors that
// are zero outside the own frames): the complete x a solve returns, the complete g / diag of the host-boundary evaluation
void gather_frame_entries(mcba_handle_s* h, double* v) {
  const Dims& d = h->d;
  if (!h->allreduce || d.DF == 0 || d.off_motion < 0 || d.n_motion == 0) return;
  if (h->sbuf.n < (size_t)d.n_motion) h->sbuf.alloc((size_t)d.n_motion, false);
  hipLaunchKernelGGL(k_shard_own_frames, dim3((d.n_motion + 255) / 256), dim3(256), 0, h->stream, d, v, h->sbuf.p);
  call_allreduce(h, h->sbuf.p, (size_t)d.n_motion, 0);
  HIP_OK(hipMemcpyAsync(v + d.off_motion, h->sbuf.p, (size_t)d.n_motion * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
}

// internal (padded) n-vector -> the caller's vector
void to_caller(const mcba_handle_s* h, double* dst, const double* src_internal) {
  if (h->ext2int.empty()) memcpy(dst, src_internal, (size_t)h->d.n * sizeof(double));
  else
    for (int i = 0; i < h->n_ext; ++i) dst[i] = src_internal[h->ext2int[i]];
}

void sync(mcba_handle_s* h) {
  check_launch("before synchronisation");
  HIP_OK(hipStreamSynchronize(h->stream));
  check_launch("after synchronisation");
}

void fetch_scalars(mcba_handle_s* h, int count, int first = 0) {   // scal[first, first + count) -> h_scal (same offsets)
  HIP_OK(hipMemcpyAsync(h->h_scal + first, h->scal.p + first, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  sync(h);
}
// split form: the copy is followed by an event, more work may be enqueued behind it, the host waits for the event only
void fetch_scalars_begin(mcba_handle_s* h, int count, int first = 0) {
  HIP_OK(hipMemcpyAsync(h->h_scal + first, h->scal.p + first, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipEventRecord(h->ev_fetch, h->stream));
}
void fetch_scalars_end(mcba_handle_s* h) {
  check_launch("iteration enqueue");
  HIP_OK(hipEventSynchronize(h->ev_fetch));
}
double now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// k_publish form of the fetch: scal[0, count) and a sequence number go to pinned host memory from a kernel on `st`; the host
// spins on the number (falls back to a stream synchronisation after 20 ms: a failed launch must not hang the caller)
void publish_scalars_begin(mcba_handle_s* h, int count, hipStream_t st) {
  ++h->pub_seq;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, st, h->scal.p, h->h_scal, count, h->h_pub_seq, h->pub_seq);
}
void publish_scalars_end(mcba_handle_s* h, hipStream_t st) {
  check_launch("iteration enqueue");
  const double t0 = now_seconds();
  int spins = 0;
  while (__atomic_load_n(h->h_pub_seq, __ATOMIC_ACQUIRE) != h->pub_seq) {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if ((++spins & 1023) == 0 && now_seconds() - t0 > 0.02) {
      HIP_OK(hipStreamSynchronize(st));
      REQUIRE(__atomic_load_n(h->h_pub_seq, __ATOMIC_ACQUIRE) == h->pub_seq, "the published scalars did not arrive");
      break;
    }
  }
}
double host_sum(const double* p, int n) {   // fixed order: the result does not depend on block scheduling
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int i = 0;
  for (; i + 4 <= n; i += 4) { s0 += p[i]; s1 += p[i + 1]; s2 += p[i + 2]; s3 += p[i + 3]; }
  for (; i < n; ++i) s0 += p[i];
  return (s0 + s1) + (s2 + s3);
}




bool g_force_blocked_chol = false;   // test hooks: the multi-workgroup kernels / the panel kernels at any size
bool g_force_panel2_chol = false;
long long* g_chol_prof = nullptr;    // device buffer of 8 phase stamps (mcba_debug_chol, blocked == 4 / 7 / 9)

// (S + reg I) p = rhs for buf = [S (ns x ns) | rhs (ns)]; S is overwritten by its Cholesky factor.  Three paths by size:
//   ns + 1 <= 160    k_chol_blk     one workgroup, the lower triangle resident in LDS as 16 x 16 tiles
//   ns + 1 <= 1024   k_cholp_*      block columns factored in LDS by one workgroup, trailing update on the whole chip
//   larger           k_cholb_*      multi-workgroup 64-column panels (adjust_board with thousands of board points)
// (The column-by-column LDS kernel of round 1, the 32-column single-workgroup kernel and the matrix-in-L2 tile kernel of
//  round 2 lost to these at every size and are gone: profiles/r03_cholesky_paths.txt.)
void launch_chol(mcba_handle_s* h, int ns, double reg, double* buf, double* ps) {
  if (ns + 1 <= CHOL_BLK_MAX_N1 && !g_force_blocked_chol && !g_force_panel2_chol) {
    const size_t lds_blk = chol_blk_lds_bytes(ns);
    raise_dynamic_lds((const void*)k_chol_blk, h->device, lds_blk);
    hipLaunchKernelGGL(k_chol_blk, dim3(1), dim3(CHOL_BLK_THREADS), lds_blk, h->stream, ns, reg, buf, ps, h->info.p, g_chol_prof);
    return;
  }
  if (!g_force_blocked_chol && ns + 1 <= CHOLP_MAX_N1) {
    const int nb = (ns + 1 + CT - 1) / CT, nbc = (ns + CT - 1) / CT;
    const size_t nlinv = (size_t)nb * CT * CT;
    if (h->chol_linv.n < nlinv) h->chol_linv.alloc(nlinv, false);
    for (int kt0 = 0; kt0 < nbc;) {
      const int wt = cholp_panel_tiles(ns, kt0);
      const size_t lds = cholp_lds_bytes(ns, kt0, wt);
      raise_dynamic_lds((const void*)k_cholp_panel, h->device, lds);
      hipLaunchKernelGGL(k_cholp_panel, dim3(1), dim3(CHOLP_THREADS), lds, h->stream, ns, kt0, wt, reg, buf, h->chol_linv.p,
                         h->info.p, g_chol_prof);
      const int k1 = kt0 + wt, m = nb - k1, nt = m * (m + 1) / 2;
      if (k1 < nbc && nt > 0)
        hipLaunchKernelGGL(k_cholp_trail, dim3((nt + 3) / 4), dim3(256), 0, h->stream, ns, kt0, wt, buf);
      kt0 = k1;
    }
    if (ns <= CHOLP_BACK_COLS)
      hipLaunchKernelGGL((k_cholp_back<1, 4>), dim3(1), dim3(CHOLP_BACK_THREADS), 0, h->stream, ns, (const double*)buf,
                         (const double*)h->chol_linv.p, ps);
    else if (ns <= 2 * CHOLP_BACK_COLS)
      hipLaunchKernelGGL((k_cholp_back<2, 2>), dim3(1), dim3(CHOLP_BACK_THREADS), 0, h->stream, ns, (const double*)buf,
                         (const double*)h->chol_linv.p, ps);
    else
      hipLaunchKernelGGL((k_cholp_back<3, 1>), dim3(1), dim3(CHOLP_BACK_THREADS), 0, h->stream, ns, (const double*)buf,
                         (const double*)h->chol_linv.p, ps);
    return;
  }
  // large reduced system: multi-workgroup blocked factorisation
  HIP_OK(hipMemsetAsync(h->info.p, 0, sizeof(int32_t), h->stream));
  for (int k0 = 0; k0 < ns; k0 += CB) {
    const int nb = std::min(CB, ns - k0);
    hipLaunchKernelGGL(k_cholb_diag, dim3(1), dim3(64), 0, h->stream, ns, k0, reg, buf, h->info.p);
    const int m = ns - k0 - nb + 1;   // rows below the panel incl. the rhs row
    hipLaunchKernelGGL(k_cholb_trsm, dim3((m + 63) / 64), dim3(64), 0, h->stream, ns, k0, buf);
    const int nblk = (m + CB - 1) / CB;
    if (m > 1) hipLaunchKernelGGL(k_cholb_syrk, dim3(nblk, nblk), dim3(256), 0, h->stream, ns, k0, buf);
  }
  for (int k0 = ((ns - 1) / CB) * CB; k0 >= 0; k0 -= CB) {
    hipLaunchKernelGGL(k_cholb_back_diag, dim3(1), dim3(64), 0, h->stream, ns, k0, buf, ps);
    if (k0 > 0) hipLaunchKernelGGL(k_cholb_back_update, dim3((k0 + 255) / 256), dim3(256), 0, h->stream, ns, k0, buf);
  }
}

// gn = (H_h + reg I)^-1 g_h  through the Schur complement of the per-frame blocks
// dots_out (device, may be null): a single-GPU handle receives the per-block partial dots {g_h.g_h, g_h.gn, gn.gn} of
// the back-substitution blocks, dots_out[3 blk + k], followed by the Cholesky pivot report (see gn_dot_blocks); a
// frame-sharded handle receives the three complete dots (k_dots3 after the all-reduce of gn).
int gn_dot_blocks(const Dims& d) {
  const int K = d.DF * d.Fl;
  return (K > 0 ? d.Fl : 0) + 1;
}
// tr_dev (device, may be null): the scalar block of the single-GPU driver; the damping is then read from tr_dev[TR_REG]
// on the device (`reg` is ignored) and added to the reduced system by k_schur_reduce.
// tr_dev: the damping is S[TR_REG] of that scalar block; with `trp` (the partial sums k_tr_reg would fold) the first kernel
// of the chain computes and publishes it itself
struct TrRegPartials { const double* vs; int nvb; const double* q; int nq; int first; double Delta; };
// 3 x 3 register-blocked Schur SYRK (MCBA_SYRK3=0 keeps one wavefront per tile pair)
// (reduced systems of nine tile columns or more: below that the blocks are mostly padding -- 6.9 against 4.5 us at 4 x 200 x 1;
//  MCBA_SYRK3=1 forces it for every size)
static bool syrk3_enabled(const mcba_handle_s* h) {
  static const char* env = dbg_switch("MCBA_SYRK3");
  if (env != nullptr && env[0] == '0') return false;
  return h->use_mfma && (h->ntile >= 9 || (env != nullptr && env[0] == '1'));
}

void launch_gn_solve(mcba_handle_s* h, double reg, bool root_rank, double* dots_out = nullptr,
                     double* tr_dev = nullptr, const TrRegPartials* trp = nullptr) {
  const Dims& d = h->d;
  const int K = d.DF * d.Fl;
  // Whether the frame part of the step is reduced across ranks must not depend on THIS rank's shard size (an empty shard,
  // K == 0, is legal: frame_shards produces them when there are fewer frames than ranks): every rank of a problem with
  // eliminated frame parameters (DF > 0) joins the same sequence of collectives and contributes zeros where it owns nothing.
  // (frame-sharded: the frame entries of gn belong to the owner of the frame and stay zero everywhere else -- the buffer is
  //  zero-initialised and the back substitution writes own frames + shared entries only; the dots of the 2-D subspace cross
  //  the ranks as per-rank partial sums, see k_shard_fold_dots)
  double* fused_dots = dots_out;
  if (trp != nullptr && K == 0)   // no frame blocks on this rank: the stand-alone fold
    hipLaunchKernelGGL(k_tr_reg, dim3(1), dim3(64), 0, h->stream, tr_dev, trp->vs, trp->nvb, trp->q, trp->nq, trp->first,
                       trp->Delta);
  if (K > 0) {
    const TrRegPartials z = trp ? *trp : TrRegPartials{nullptr, 0, nullptr, 0, 0, 0.0};
    if (d.DF == 12)
      hipLaunchKernelGGL((k_schur_frame<12>), dim3(d.Fl), dim3(256), 0, h->stream, d, h->Hff.p, h->Hfs.p, h->dsc.p, h->gh.p, reg,
                         h->Lf.p, h->W.p, h->yf.p, tr_dev, z.vs, z.nvb, z.q, z.nq, z.first, z.Delta);
    else
      hipLaunchKernelGGL((k_schur_frame<6>), dim3(d.Fl), dim3(256), 0, h->stream, d, h->Hff.p, h->Hfs.p, h->dsc.p, h->gh.p, reg,
                         h->Lf.p, h->W.p, h->yf.p, tr_dev, z.vs, z.nvb, z.q, z.nq, z.first, z.Delta);
    const int nt2 = h->ntile * (h->ntile + 1) / 2;
    if (syrk3_enabled(h)) {
      const int nt3 = (h->ntile + 2) / 3;
      hipLaunchKernelGGL(k_schur_syrk3, dim3(h->ksplit, nt3 * (nt3 + 1) / 2), dim3(SYRK3_THREADS), 0, h->stream, K, d.ns + 1,
                         h->ntile, h->ksplit, h->W.p, h->P.p);
    } else if (h->use_mfma)
      hipLaunchKernelGGL((k_schur_syrk<true>), dim3(h->ksplit, nt2), dim3(64), 0, h->stream, K, d.ns + 1, h->ntile,
                         h->ksplit, h->W.p, h->P.p);
    else
      hipLaunchKernelGGL((k_schur_syrk<false>), dim3(h->ksplit, nt2), dim3(64), 0, h->stream, K, d.ns + 1, h->ntile,
                         h->ksplit, h->W.p, h->P.p);
  }
  const int total = d.ns * d.ns + d.ns;
  hipLaunchKernelGGL(k_schur_reduce, dim3(std::min(2048, (4 * total + 255) / 256)), dim3(256), 0, h->stream, d, h->Hss.p,
                     h->dsc.p, h->gh.p, h->P.p, h->ntile, h->ksplit, K, root_rank ? 1.0 : 0.0, h->sbuf.p, tr_dev);
  call_allreduce(h, h->sbuf.p, (size_t)total, 0);
  launch_chol(h, d.ns, tr_dev ? 0.0 : reg, h->sbuf.p, h->ps.p);
  check_launch("reduced Cholesky");
  if (d.DF == 12) {
    const int nblk = gn_dot_blocks(d);
    hipLaunchKernelGGL((k_schur_backsub<12>), dim3(nblk), dim3(64), 0, h->stream, d, h->Lf.p, h->W.p, h->yf.p, h->ps.p,
                       h->gn.p, h->gh.p, h->info.p, fused_dots);
  } else {
    const int nblk = gn_dot_blocks(d);
    hipLaunchKernelGGL((k_schur_backsub<6>), dim3(nblk), dim3(64), 0, h->stream, d, h->Lf.p, h->W.p, h->yf.p, h->ps.p,
                       h->gn.p, h->gh.p, h->info.p, fused_dots);
  }
  if (h->allreduce && dots_out) {   // message 4: 3 doubles per rank (+ the pivot report), gathered by summation
    const int W = d.shard_world;
    hipLaunchKernelGGL(k_shard_fold_dots, dim3(1), dim3(64), 0, h->stream, dots_out, gn_dot_blocks(d), d.shard_rank, W,
                       h->scal.p + h->sl.shard4);
    call_allreduce(h, h->scal.p + h->sl.shard4, (size_t)3 * W + 1, 0);
  }
}

// ---- device outlier loop -------------------------------------------------------------------------------------------
// first residual index of every view for the current inlier table (k_residual writes each view's residuals as one run)
void ensure_view_first(mcba_handle_s* h) {
  if (!h->view_first_dirty) return;
  hipLaunchKernelGGL(k_view_scan, dim3(1), dim3(1024), 0, h->stream, h->d, h->view_count.p, (const int32_t*)nullptr,
                     h->view_first.p, h->totals.p);
  h->view_first_dirty = false;
}

void ensure_obs_index(mcba_handle_s* h) {
  if (!h->obs_index_dirty) return;
  // residual ordering of the current inlier table (needed by residuals / jacobian only): prefix sums on the device
  const Dims& d = h->d;
  if (d.views() > 0) {
    hipLaunchKernelGGL(k_view_scan, dim3(1), dim3(1024), 0, h->stream, d, h->view_count.p, (const int32_t*)nullptr,
                       h->view_first.p, h->totals.p);
    hipLaunchKernelGGL(k_obs_index, dim3(d.views()), dim3(64), 0, h->stream, d, h->inlier.p, h->view_first.p, h->obs_index.p);
  }
  h->obs_index_dirty = false;
}

// compacted observation tables of the lsmr route for the current inlier table (the board points are constants here: boards=True keeps
// the masks form).  Needs the board-point table, i.e. a preceding table preparation (lsmr_linearize).
void ensure_compact(mcba_handle_s* h) {
  if (!h->compact_dirty) return;
  const Dims& d = h->d;
  ensure_view_first(h);
  const size_t n = (size_t)std::max<int64_t>(h->n_inliers, 1) + 64;
  if (h->cp_obs.n < n) { h->cp_obs.alloc(n, false); h->cp_bxy.alloc(n, false); h->cp_bz.alloc(n, false); }
  if (d.motion == MOTION_ROLLING && h->cp_tr.n < n) h->cp_tr.alloc(n, false);
  // (one descriptor per entry of the active list: a hand-made list -- mcba_debug_set_frame_groups -- may be longer than the views, padded with -1)
  const size_t nv = std::max<size_t>(std::max(d.views(), 1), h->active_views.n > 0 ? h->active_views.n - 1 : 0);
  if (h->cp_desc.n < nv) h->cp_desc.alloc(nv, false);
  if (d.views() > 0)
    hipLaunchKernelGGL(k_compact_views, dim3((unsigned)nv), dim3(64), 0, h->stream, d, h->t, (const int32_t*)h->view_first.p, h->cp_obs.p, h->cp_bxy.p,
                       h->cp_bz.p, d.motion == MOTION_ROLLING ? h->cp_tr.p : nullptr, h->cp_desc.p);
  check_launch("k_compact_views");
  h->compact_dirty = false;
}
LsmrCompact compact_tables(const mcba_handle_s* h) {
  return LsmrCompact{h->cp_obs.p, h->cp_bxy.p, h->cp_bz.p, h->d.motion == MOTION_ROLLING ? h->cp_tr.p : nullptr, h->cp_desc.p};
}

void compute_errors(mcba_handle_s* h, const double* x) {
  const Dims& d = h->d;
  if (h->err_fm.n < (size_t)std::max(d.slots(), 1)) h->err_fm.alloc((size_t)std::max(d.slots(), 1));
  if (!h->sel_hist.p) {
    h->sel_hist.alloc((size_t)SEL_MAX * 2048);
    h->sel_state.alloc(3 * SEL_MAX + (size_t)SEL_NEXT_BLOCKS * SEL_MAX * 2);   // SelState | ranks | per-block (count, next)
    h->sel_f64.alloc((size_t)SEL_MAX * 2048);
  }
  // the per-slot errors depend on x only (the observation tables of a handle are fixed): report() asks for the statistics
  // of the same point several times (all valid points, inliers, the rejection threshold) -- evaluate once
  const size_t nx = h->ext2int.empty() ? (size_t)d.n : (size_t)h->n_ext;   // length of the caller's vector
  if (h->err_valid && h->err_x.size() == nx && memcmp(h->err_x.data(), x, nx * sizeof(double)) == 0) return;
  upload_x(h, x, h->x.p);
  eval_tables(h, h->x.p);
  h->ops->residual(d, h->t, h->stream, nullptr, nullptr, nullptr, h->err_fm.p, nullptr);   // frame-major errors
  h->err_x.assign(x, x + nx);
  h->err_valid = true;
}

int sel_grid(const Dims& d) { return std::max(1, std::min(1024, (d.slots() + 255) / 256)); }

void reduce_hist(mcba_handle_s* h, int nh = 1) {
  if (!h->allreduce) return;
  hipLaunchKernelGGL(k_u32_to_f64, dim3(8 * nh), dim3(256), 0, h->stream, h->sel_hist.p, h->sel_f64.p, 2048 * nh);
  call_allreduce(h, h->sel_f64.p, (size_t)2048 * nh, 0);
  hipLaunchKernelGGL(k_f64_to_u32, dim3(8 * nh), dim3(256), 0, h->stream, h->sel_f64.p, h->sel_hist.p, 2048 * nh);
}

// exact k-th smallest (0-based, over all ranks of a sharded problem) of the masked errors, and the (k+1)-th, for
// several order statistics at once (nsel <= SEL_MAX): the same six passes over the errors serve all of them
void select_ranks_multi(mcba_handle_s* h, const uint8_t* m2, int nsel, const long long* ranks, double* v_k, double* v_k1) {
  const Dims& d = h->d;
  SelState* st = reinterpret_cast<SelState*>(h->sel_state.p);
  long long* dranks = reinterpret_cast<long long*>(h->sel_state.p + 2 * SEL_MAX);
  unsigned long long* dout = h->sel_state.p + 3 * SEL_MAX;
  const int n = d.slots(), grid = sel_grid(d);
  HIP_OK(hipMemcpyAsync(dranks, ranks, nsel * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_selm_init, dim3(1), dim3(256), 0, h->stream, st, nsel, dranks, h->sel_hist.p);
  const int shifts[6] = {53, 42, 31, 20, 9, 0};
  const int bits[6] = {11, 11, 11, 11, 11, 9};
  for (int p = 0; p < 6; ++p) {
    hipLaunchKernelGGL(k_selm_hist, dim3(grid), dim3(256), 0, h->stream, h->err_fm.p, h->evalid.p, m2, n, st, nsel, shifts[p],
                       bits[p], p == 0 ? 1 : 0, h->sel_hist.p);
    reduce_hist(h, p == 0 ? 1 : nsel);
    hipLaunchKernelGGL(k_selm_pick, dim3(1), dim3(64 * SEL_MAX), 0, h->stream, st, nsel, h->sel_hist.p, shifts[p], bits[p],
                       p == 0 ? 1 : 0);
  }
  hipLaunchKernelGGL(k_selm_next, dim3(SEL_NEXT_BLOCKS), dim3(256), 0, h->stream, h->err_fm.p, h->evalid.p, m2, n, st, nsel,
                     dout);
  std::vector<unsigned long long> host(3 * SEL_MAX + (size_t)SEL_NEXT_BLOCKS * SEL_MAX * 2);
  HIP_OK(hipMemcpyAsync(host.data(), h->sel_state.p, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                        h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  for (int r = 0; r < nsel; ++r) {
    double vk, next;
    memcpy(&vk, &host[2 * r], 8);                       // SelState r: prefix = bit pattern of the order statistic
    unsigned long long cnt = 0, mn = 0x7FF0000000000000ull;
    for (int blk = 0; blk < SEL_NEXT_BLOCKS; ++blk) {
      const unsigned long long* pp = &host[3 * SEL_MAX + ((size_t)blk * SEL_MAX + r) * 2];
      cnt += pp[0];
      mn = pp[1] < mn ? pp[1] : mn;
    }
    memcpy(&next, &mn, 8);
    double cnt_le = (double)cnt;
    if (h->allreduce) {   // combine (count, min) across ranks: sum and -max(-x)
      double buf[2] = {cnt_le, -next};
      HIP_OK(hipMemcpyAsync(h->sel_f64.p, buf, sizeof(buf), hipMemcpyHostToDevice, h->stream));
      call_allreduce(h, h->sel_f64.p, 1, 0);
      call_allreduce(h, h->sel_f64.p + 1, 1, 1);
      HIP_OK(hipMemcpyAsync(buf, h->sel_f64.p, sizeof(buf), hipMemcpyDeviceToHost, h->stream));
      HIP_OK(hipStreamSynchronize(h->stream));
      cnt_le = buf[0];
      next = -buf[1];
    }
    v_k[r] = vk;
    v_k1[r] = (cnt_le > (double)(ranks[r] + 1)) ? vk : next;
  }
}

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
// (g_fill_stream is valid for the duration of ONE API call: the entry points that allocate set it, every exit clears it)
#define API_BEGIN try {
#define API_END                                 \
  g_fill_stream = nullptr;                      \
  return 0;                                     \
  }                                             \
  catch (const std::exception& e) {             \
    g_fill_stream = nullptr;                    \
    g_error = e.what();                         \
    return 1;                                   \
  }                                             \
  catch (...) {                                 \
    g_fill_stream = nullptr;                    \
    g_error = "unknown error";                  \
    return 1;                                   \
  }

extern "C" {

const char* mcba_last_error(void) { return g_error.c_str(); }

int32_t mcba_full_size(const mcba_problem* p, int64_t* out) {
  API_BEGIN
  REQUIRE(p && out, "null argument");
  *out = full_size_of(p);
  API_END
}

int32_t mcba_create(const mcba_problem* p, void* hip_stream, mcba_handle* out) {
  API_BEGIN
  REQUIRE(p && out, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    throw Error("no HIP device: the mcba back-end is GPU-only (there is no CPU fallback)");
  auto h = std::make_unique<mcba_handle_s>();
  HIP_OK(hipGetDevice(&h->device));
  {   // the architecture check costs a hipGetDeviceProperties (milliseconds): once per device and process
    static std::string arch_of[64];
    static bool arch_known[64] = {false};
    static std::mutex arch_mutex;             // handles may be created from several threads at once
    std::lock_guard<std::mutex> arch_lock(arch_mutex);
    const int di = h->device;
    if (di < 0 || di >= 64 || !arch_known[di]) {
      hipDeviceProp_t prop;
      HIP_OK(hipGetDeviceProperties(&prop, h->device));
      if (di >= 0 && di < 64) { arch_of[di] = prop.gcnArchName; arch_known[di] = true; }
      REQUIRE(std::string(prop.gcnArchName).rfind("gfx950", 0) == 0,
              std::string("mcba kernels are built for gfx950 only, found ") + prop.gcnArchName);
    } else {
      REQUIRE(arch_of[di].rfind("gfx950", 0) == 0, std::string("mcba kernels are built for gfx950 only, found ") + arch_of[di]);
    }
  }
  if (hip_stream) {
    h->stream = (hipStream_t)hip_stream;
  } else {
    h->stream = resource_cache().take_stream();
    if (h->stream == nullptr) HIP_OK(hipStreamCreate(&h->stream));
    h->own_stream = true;
  }
  if (const char* env = getenv("MCBA_NO_MFMA")) h->use_mfma = !(env[0] == '1');

  const bool timing = getenv("MCBA_TIMING") != nullptr;
  const double tc0 = now_seconds();
  g_fill_stream = h->stream;      // buffers are zero-filled on the handle's stream (no synchronisation per buffer)
  HostProblem hp;
  lower_dims(p, hp);              // shape, index maps, small parameter tables; the slot tables are built on the device
  h->ops = pick_ops(hp.d.fisheye, hp.d.ND);   // (a rig that mixes the projection families runs the per-camera instantiation)
  const double tc1 = now_seconds();
  h->d = hp.d;
  Dims& d = h->d;
  h->ext2int = hp.ext2int;
  h->n_ext = hp.n_ext;
  if (!hp.cam_kmask.empty()) {
    h->cam_kmask.upload(hp.cam_kmask);
    d.cam_kmask = h->cam_kmask.p;      // (device pointer: kernels only; host code below uses hp.cam_kmask)
    if (!hp.int2ext.empty()) h->int2ext.upload(hp.int2ext);
  }
  // ---- raw upload + device lowering (k_lower_view): the caller's [C,F,B,P] arrays go up as they are ------------------
  const size_t nslot = (size_t)d.slots();
  h->obs.alloc(nslot, false);
  h->evalid.alloc(nslot, false);
  h->valid_fm.alloc(nslot, false);
  h->inlier.alloc(nslot, false);
  h->obs_index.alloc(nslot, false);
  h->view_count.alloc((size_t)d.views(), false);
  h->view_first.alloc((size_t)d.views(), false);
  h->totals.alloc(2);
  h->h_totals_bytes = 2 * sizeof(long long);
  h->h_totals = (long long*)pinned_alloc(h->h_totals_bytes);
  h->board_off.upload(hp.board_off);
  h->cam_valid.upload(std::vector<uint8_t>(p->camera_valid, p->camera_valid + d.C));
  h->frame_valid.upload(std::vector<uint8_t>(p->frame_valid, p->frame_valid + d.F));
  h->board_valid.upload(std::vector<uint8_t>(p->board_valid, p->board_valid + d.B));
  {
    DevBuf<double2> raw_pts;
    DevBuf<float2> raw_pts32;        // (a float32 table goes up as it is: half the bytes of the largest upload)
    DevBuf<uint8_t> raw_valid, view_e_unused;
    DevBuf<int32_t> view_ecount;
    raw_valid.alloc(nslot, false);
    view_ecount.alloc((size_t)d.views(), false);
    if (p->points) {
      raw_pts.alloc(nslot, false);
      upload_shard_slabs(h.get(), raw_pts.p, p->points, sizeof(double2));
    } else {
      raw_pts32.alloc(nslot, false);
      upload_shard_slabs(h.get(), raw_pts32.p, p->points_f32, sizeof(float2));
    }
    upload_shard_slabs(h.get(), raw_valid.p, p->point_valid, 1);
    if (p->inlier_mask) {
      h->raw_mask.alloc(nslot, false);
      upload_shard_slabs(h.get(), h->raw_mask.p, p->inlier_mask, 1);
    }
    if (d.views() > 0) {
      hipLaunchKernelGGL(k_lower_view, dim3(d.views()), dim3(64), 0, h->stream, d, (const double2*)raw_pts.p,
                         (const float2*)raw_pts32.p, (const uint8_t*)raw_valid.p, (const uint8_t*)(p->inlier_mask ? h->raw_mask.p : nullptr),
                         h->cam_valid.p, h->frame_valid.p, h->board_valid.p, h->board_off.p, h->obs.p, h->valid_fm.p,
                         h->evalid.p, h->inlier.p, h->view_count.p, view_ecount.p);
    }
    hipLaunchKernelGGL(k_view_scan, dim3(1), dim3(1024), 0, h->stream, d, h->view_count.p, (const int32_t*)view_ecount.p,
                       h->view_first.p, h->totals.p);
    HIP_OK(hipMemcpyAsync(h->h_totals, h->totals.p, 2 * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));   // (the raw slabs are freed at the end of this scope)
    REQUIRE(h->h_totals[0] < (1LL << 30), "too many observations for 32-bit residual indices");
    h->n_inliers = h->h_totals[0];
    h->n_evalid = h->h_totals[1];
  }
  h->obs_index_dirty = true; h->compact_dirty = true;      // the residual ordering is built on first use (mcba_residuals / mcba_jacobian)
  const double tc2 = now_seconds();
  h->active_views.alloc((size_t)d.views() + 1);
  h->work_counter.alloc(2);
  {
    const int32_t g0[2] = {LIN_GRID_MAX < d.views() ? LIN_GRID_MAX : d.views(), LIN_GRID_MAX < d.views() ? LIN_GRID_MAX : d.views()};
    HIP_OK(hipMemcpyAsync(h->work_counter.p, g0, sizeof(g0), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));   // (g0 lives on the stack)
  }
  h->out_r.alloc((size_t)std::max<int64_t>(2 * h->n_inliers, 1), false);
  h->full2act.upload(hp.full2act);
  h->xfull.upload(hp.xfull);
  h->img_h.upload(hp.img_h);
  h->fix_aspect.upload(hp.fix_aspect);
  h->bwg.upload(hp.bwg);
  h->tri.upload(hp.tri);
  if (d.DF > 0 && assemble_lds_fixed(d) + (size_t)frame_entries(d) * sizeof(double) > ASM_LDS_MAX)
    throw Error("too many (camera, board) pairs for the per-frame assembly (view-rank tables of 4 B per pair exceed the 150 KB of "
                "LDS a workgroup can have: about 36 000 pairs)");
  {
    Dims dh = h->d;
    dh.cam_kmask = hp.cam_kmask.empty() ? nullptr : hp.cam_kmask.data();   // host copy for the host-side index arithmetic
    const std::vector<int4> ft = frame_table(dh);
    h->ftab.upload(ft);
    h->nftab = (int)ft.size();
  }
  h->board_points.alloc((size_t)d.B * d.P * 3);
  h->pose.alloc((size_t)d.n_pose * POSE_STRIDE);
  h->cam.alloc((size_t)d.C * CAM_STRIDE);
  h->view.alloc((size_t)d.views() * d.view_stride());
  h->tmat.alloc((size_t)d.views() * d.DE * 6 * d.NPB);

  Tables& t = h->t;
  t.obs = h->obs.p; t.inlier = h->inlier.p; t.evalid = h->evalid.p; t.obs_index = h->obs_index.p;
  t.view_count = h->view_count.p; t.active_views = h->active_views.p; t.work_counter = h->work_counter.p; t.board_off = h->board_off.p; t.full2act = h->full2act.p; t.xfull = h->xfull.p;
  t.bwg = h->bwg.p; t.img_h = h->img_h.p; t.fix_aspect = h->fix_aspect.p; t.board_points = h->board_points.p;
  t.pose = h->pose.p; t.cam = h->cam.p; t.view = h->view.p; t.tmat = h->tmat.p; t.dbg = nullptr;
  t.int2ext = h->int2ext.p;
  refresh_active_views(h.get());

  // ---- work buffers -----------------------------------------------------------------------------------------
  h->rec.alloc((size_t)d.views() * d.rec_stride);
  // chunk sums of the shared part: about 512 (pair, chunk) workgroups in k_assemble; k_shared_final reads C B nchunk
  // partial records per entry, so many pairs get fewer chunks
  {
    const char* e = dbg_switch("MCBA_NCHUNK_TARGET");   // tuning knob: (pair, chunk) workgroups aimed at
    const int target = e ? std::max(1, atoi(e)) : 256;   // measured at cfg3: 256 -> 114.4 us / step, 512 -> 113.1, 1024 -> 119.3
    h->nchunk = std::max(1, std::min(std::min(64, (d.Fl + 7) / 8), std::max(4, target / std::max(1, d.C * d.B))));
    // Rigs with many (camera, board) pairs AND many frames got 4 chunks of hundreds of frames: a chunk-sum block walks its frames eight
    // views at a time, one round trip each -- 16 x 1000 x 5 (80 pairs, 250 frames per chunk): k_assemble 40.8 us.  No chunk longer than
    // 64 frames; then about 40 (round 6, profiles/scripts/prof_nchunk.py: step 106.6 -> 97.0 us there; 8 x 500 x 2 and 6 x 400 x 5 unchanged,
    // they lose 2 - 4 us with more chunks than they have).
    if (!e && d.Fl > 64 * h->nchunk) h->nchunk = std::min(64, (d.Fl + 39) / 40);
  }
  h->partial.alloc((size_t)d.C * d.B * h->nchunk * d.rec_stride);
  h->Hss.alloc((size_t)d.ns * d.ns);
  h->Hfs.alloc((size_t)d.Fl * d.DF * d.ns);
  h->Hff.alloc((size_t)d.Fl * d.DF * d.DF);
  h->gbuf.alloc(2 * (size_t)d.n + 2);
  for (DevBuf<double>* b : {&h->x, &h->xnew, &h->scale_inv, &h->dsc, &h->gh, &h->gn, &h->scale_inv2, &h->dsc2, &h->gh2})
    b->alloc((size_t)d.n);
  h->sl.init(d.n, d.Fl + 2);
  h->scal.alloc((size_t)h->sl.total);
  h->cost_blocks = std::max(1, std::min(COST_BLOCKS_MAX, d.views()));
  h->costpart.alloc((size_t)h->cost_blocks);
  h->Lf.alloc((size_t)d.Fl * d.DF * d.DF);
  h->W.alloc((size_t)d.Fl * d.DF * (d.ns + 1));
  h->yf.alloc((size_t)d.Fl * d.DF);
  h->ntile = (d.ns + 1 + 15) / 16;   // tiles of W' = [W | y]
  {
    const int K = d.DF * d.Fl;
    const int nt2 = h->ntile * (h->ntile + 1) / 2;
    int ks = 1;
    if (syrk3_enabled(h.get())) {   // k_schur_syrk3: four wavefronts per (48 x 48 block, split); >= 192 workgroups, >= 16 rows each
      const int nt3 = (h->ntile + 2) / 3, np3 = nt3 * (nt3 + 1) / 2;
      while (ks < 64 && np3 * ks < 192 && K / (ks * 2 * SYRK3_WAVES) >= 16) ks *= 2;
    } else {
      while (ks < 64 && nt2 * ks < 1024 && K / (ks * 2) >= 32) ks *= 2;
    }
    if (const char* e = dbg_switch("MCBA_KSPLIT")) ks = std::max(1, std::min(64, atoi(e)));   // (experiments)
    h->ksplit = ks;
    h->P.alloc((size_t)ks * nt2 * 256);
  }
  h->sbuf.alloc((size_t)d.ns * d.ns + d.ns);
  h->ps.alloc((size_t)d.ns);
  h->info.alloc(4);
  h->h_scal_bytes = h->scal.n * sizeof(double);
  h->h_x_bytes = std::max<size_t>(d.n, 1) * sizeof(double);
  h->h_gbuf_bytes = (2 * (size_t)d.n + 2) * sizeof(double);
  h->h_scal = (double*)pinned_alloc(h->h_scal_bytes);
  h->h_pub_seq = (unsigned long long*)pinned_alloc(64);
  h->h_pub_seq[0] = 0;
  h->h_pub_seq[1] = 0;   // progress word of LSMR solves (tagged with the call number: lsmr_solve)
  h->h_x = (double*)pinned_alloc(h->h_x_bytes);
  h->h_gbuf = (double*)pinned_alloc(h->h_gbuf_bytes);
  HIP_OK(hipEventCreate(&h->ev0));
  HIP_OK(hipEventCreate(&h->ev1));
  HIP_OK(hipEventCreateWithFlags(&h->ev_fetch, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming));
  h->stream2 = resource_cache().take_stream(true);
  if (h->stream2 == nullptr) HIP_OK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  {   // the tables of the initial point: the camera table's constant part (image height, fix_aspect) and the board points
      // when they are not optimised are only ever written here (the fused k_linearize reads them, nothing refreshes them)
    for (int j = 0; j < d.nfull; ++j)
      if (hp.full2act[j] >= 0) h->h_x[hp.full2act[j]] = hp.xfull[j];
    HIP_OK(hipMemcpyAsync(h->x.p, h->h_x, (size_t)d.n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    eval_pose_tables(h.get(), h->x.p);
  }
  HIP_OK(hipDeviceSynchronize());
  if (timing)
    fprintf(stderr, "[mcba_create] lowering %.2f ms, observation tables up %.2f ms, remaining buffers %.2f ms\n",
            (tc1 - tc0) * 1e3, (tc2 - tc1) * 1e3, (now_seconds() - tc2) * 1e3);
  *out = h.release();
  API_END
}

int32_t mcba_destroy(mcba_handle h) {
  API_BEGIN
  if (h) {
    (void)hipStreamSynchronize(h->stream);
    g_park_on_release = true;    // the handle's stream is idle: its buffers / stream go to the resource cache
    ResourceCache::park_device() = h->device;
    delete h;
    ResourceCache::park_device() = -1;
    g_park_on_release = false;
  }
  API_END
}

/* returns the device buffers, pinned buffers and streams that destroyed handles left in the process-wide cache          */
int32_t mcba_release_cached_memory(void) {
  API_BEGIN
  resource_cache().clear();
  API_END
}

int32_t mcba_device_info(mcba_handle h, char* buf, size_t buf_len) {
  API_BEGIN
  REQUIRE(h && buf && buf_len > 0, "null argument");
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, h->device));
  snprintf(buf, buf_len, "%s:%s:cus=%d:mfma=%d", prop.gcnArchName, prop.name, prop.multiProcessorCount,
           h->use_mfma ? 1 : 0);
  API_END
}

int32_t mcba_num_params(mcba_handle h, int64_t* n) {
  API_BEGIN
  REQUIRE(h && n, "null argument");
  *n = h->ext2int.empty() ? h->d.n : h->n_ext;   // length of the CALLER's vector (Calibration.param_vec)
  API_END
}

int32_t mcba_num_residuals(mcba_handle h, int64_t* n) {
  API_BEGIN
  REQUIRE(h && n, "null argument");
  *n = 2 * h->n_inliers;
  API_END
}

int32_t mcba_set_inliers(mcba_handle h, const uint8_t* mask) {
  API_BEGIN
  REQUIRE(h, "null handle");
  g_fill_stream = h->stream;
  build_inliers(h, mask);
  API_END
}

int32_t mcba_set_allreduce(mcba_handle h, mcba_allreduce_fn fn, void* ctx) {
  API_BEGIN
  REQUIRE(h, "null handle");
  h->allreduce = fn;
  h->allreduce_ctx = ctx;
  if (fn == nullptr) h->d.shard_world = 0;   // (a hook needs mcba_set_shard_rank as well)
  API_END
}

// native path: 128-byte RCCL unique id, created by ONE rank and distributed by the caller
int32_t mcba_rccl_unique_id(uint8_t* id_out) {
  API_BEGIN
  REQUIRE(id_out, "null argument");
  const RcclApi& api = rccl_api();
  REQUIRE(api.ok, "librccl could not be loaded");
  RcclApi::UniqueId id;
  const int rc = api.GetUniqueId(&id);
  REQUIRE(rc == 0, std::string("ncclGetUniqueId failed: ") + (api.GetErrorString ? api.GetErrorString(rc) : "?"));
  memcpy(id_out, id.internal, 128);
  API_END
}

// collective over all `world` ranks: creates the communicator of this handle; from then on every reduction of the handle
// is an in-place ncclAllReduce on the handle's stream -- no Python, no host synchronisation
int32_t mcba_rccl_init(mcba_handle h, const uint8_t* id_in, int32_t rank, int32_t world) {
  API_BEGIN
  REQUIRE(h && id_in && world >= 1 && rank >= 0 && rank < world, "bad argument");
  REQUIRE(world <= SHARD_MAX_WORLD, "at most 64 ranks");   // (before anything is installed on the handle)
  const RcclApi& api = rccl_api();
  REQUIRE(api.ok, "librccl could not be loaded");
  REQUIRE(h->rccl_comm == nullptr, "communicator already initialised");
  HIP_OK(hipSetDevice(h->device));
  RcclApi::UniqueId id;
  memcpy(id.internal, id_in, 128);
  void* comm = nullptr;
  const int rc = api.CommInitRank(&comm, world, id, rank);
  REQUIRE(rc == 0 && comm, std::string("ncclCommInitRank failed: ") + (api.GetErrorString ? api.GetErrorString(rc) : "?"));
  h->rccl_comm = comm;
  // Self-test of the hard-coded enumerators (ncclFloat64 = 8, ncclSum = 0, ncclMax = 2: librccl is bound through dlopen, no
  // header): a sum and a max over known doubles must come back exact, otherwise the native path is refused on every rank
  // (the caller's MIN over the success flags) and the torch.distributed hook takes over.
  {
    DevBuf<double> probe;
    probe.alloc(8);
    const double in[8] = {1.5, -2.25, (double)(rank + 1), 0.0, (double)rank, 0.0, 0.0, 0.0};
    HIP_OK(hipMemcpyAsync(probe.p, in, sizeof(in), hipMemcpyHostToDevice, h->stream));
    const int r0 = rccl_allreduce_native(h, probe.p, 3, 0, (void*)h->stream);
    const int r1 = rccl_allreduce_native(h, probe.p + 4, 1, 1, (void*)h->stream);
    double out[8];
    HIP_OK(hipMemcpyAsync(out, probe.p, sizeof(out), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    const double w = (double)world;
    const bool good = r0 == 0 && r1 == 0 && out[0] == 1.5 * w && out[1] == -2.25 * w && out[2] == 0.5 * w * (w + 1.0) &&
                      out[4] == w - 1.0;
    if (!good) {
      destroy_rccl_comm(comm);
      h->rccl_comm = nullptr;
      throw Error("librccl answered the all-reduce self-test wrongly (enumerator or ABI mismatch)");
    }
  }
  h->allreduce = rccl_allreduce_native;
  h->allreduce_ctx = h;
  h->d.shard_rank = rank;
  h->d.shard_world = world;
  h->shard_root = rank == 0;
  API_END
}

int32_t mcba_rccl_version(int32_t* version_out) {
  API_BEGIN
  REQUIRE(version_out, "null argument");
  const RcclApi& api = rccl_api();
  *version_out = api.ok ? api.version : 0;
  API_END
}

// leaves the native path again (used when the caller's consistency check across ranks fails)
int32_t mcba_rccl_shutdown(mcba_handle h) {
  API_BEGIN
  REQUIRE(h, "null handle");
  if (h->rccl_comm) {
    destroy_rccl_comm(h->rccl_comm);
    h->rccl_comm = nullptr;
    if (h->allreduce == rccl_allreduce_native) { h->allreduce = nullptr; h->allreduce_ctx = nullptr; h->d.shard_world = 0; }
  }
  API_END
}

/* collectives this handle issued since the last reset: number of all-reduce calls, doubles moved, and the sizes of the
 * first `cap` calls in issue order (negative = max reduction).  The sharded solver's chain per accepted iteration is
 * [g | diag | cost] -> Cauchy curvature (1) -> Schur system (ns^2 + ns) -> frame part of the step (n_motion).           */
int32_t mcba_allreduce_stats(mcba_handle h, int32_t reset, int64_t* calls, int64_t* doubles, int64_t* sizes, int32_t cap,
                             int32_t* n_sizes) {
  API_BEGIN
  REQUIRE(h, "null handle");
  if (calls) *calls = h->ar_calls;
  if (doubles) *doubles = h->ar_doubles;
  const int n = (int)std::min<size_t>(h->ar_trace.size(), (size_t)std::max(cap, 0));
  if (sizes) for (int i = 0; i < n; ++i) sizes[i] = h->ar_trace[i];
  if (n_sizes) *n_sizes = n;
  if (reset) { h->ar_calls = 0; h->ar_doubles = 0; h->ar_trace.clear(); }
  API_END
}

int32_t mcba_set_shard_root(mcba_handle h, int32_t is_root) {
  API_BEGIN
  REQUIRE(h, "null handle");
  h->shard_root = is_root != 0;
  API_END
}

/* rank of this handle among the `world` handles that share one frame-sharded problem (rank 0 is the root) */
int32_t mcba_set_shard_rank(mcba_handle h, int32_t rank, int32_t world) {
  API_BEGIN
  REQUIRE(h && world >= 1 && world <= SHARD_MAX_WORLD && rank >= 0 && rank < world, "bad rank / world (at most 64 ranks)");
  h->d.shard_rank = rank;
  h->d.shard_world = world;
  h->shard_root = rank == 0;
  API_END
}

int32_t mcba_set_log(mcba_handle h, mcba_log_fn fn, void* ctx) {
  API_BEGIN
  REQUIRE(h, "null handle");
  h->log = fn;
  h->log_ctx = ctx;
  API_END
}

// test / debug knob: 1 = MFMA accumulate (default), 0 = plain-FMA accumulate with the identical data flow
int32_t mcba_set_mfma(mcba_handle h, int32_t on) {
  API_BEGIN
  REQUIRE(h, "null handle");
  h->use_mfma = on != 0;
  API_END
}

// debug knob: number of persistent k_linearize workgroups (0 = automatic)
int32_t mcba_debug_set_lin_grid(mcba_handle h, int32_t grid) {
  API_BEGIN
  REQUIRE(h && grid >= 0, "bad argument");
  h->lin_grid = grid;
  API_END
}

// Experiment: what would a FRAME-level linearisation cost in load balance alone?  Replaces the
// largest-first list of views by a list in which wave (frame rank r, w), w < nw, walks only views of frame r -- the frames
// largest-first, the views of a frame dealt to its nw waves by greedy longest-processing-time -- and sets the grid to
// frames x nw single-wave workgroups.  The kernel itself is unchanged (per-view records, no LDS reduction), so the time
// measured is a LOWER bound for a kernel whose workgroups own whole frames.  nw = 0 restores the product's list.
int32_t mcba_debug_set_frame_groups(mcba_handle h, int32_t nw) {
  API_BEGIN
  REQUIRE(h && nw >= 0 && nw <= 16, "bad argument");
  const Dims& d = h->d;
  if (nw == 0) {
    h->lin_grid = 0;
    refresh_active_views(h);
    sync(h);
    h->compact_dirty = true;
    return 0;
  }
  const int cb = d.C * d.B, nv = d.views();
  std::vector<int32_t> cnt((size_t)nv);
  HIP_OK(hipMemcpyAsync(cnt.data(), h->view_count.p, (size_t)nv * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  std::vector<std::pair<long long, int>> frames;   // (observations, local frame)
  for (int fl = 0; fl < d.Fl; ++fl) {
    long long tot = 0;
    for (int k = 0; k < cb; ++k) tot += cnt[(size_t)fl * cb + k];
    if (tot > 0) frames.push_back({tot, fl});
  }
  std::sort(frames.begin(), frames.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
  const int nf = (int)frames.size(), grid = nf * nw;
  std::vector<std::vector<int>> lists((size_t)grid);
  size_t rounds = 0;
  for (int r = 0; r < nf; ++r) {
    const int fl = frames[r].second;
    std::vector<std::pair<int, int>> vs;
    for (int k = 0; k < cb; ++k)
      if (cnt[(size_t)fl * cb + k] > 0) vs.push_back({cnt[(size_t)fl * cb + k], fl * cb + k});
    std::sort(vs.begin(), vs.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    std::vector<long long> load((size_t)nw, 0);
    for (const auto& e : vs) {
      int w = 0;
      for (int q = 1; q < nw; ++q)
        if (load[q] < load[w]) w = q;
      load[w] += e.first + 100;     // (+ the fixed cost of a view, in observations)
      lists[(size_t)r * nw + w].push_back(e.second);
      rounds = std::max(rounds, lists[(size_t)r * nw + w].size());
    }
  }
  std::vector<int32_t> host(1 + rounds * (size_t)grid, -1);
  host[0] = (int32_t)(rounds * (size_t)grid);
  for (int g = 0; g < grid; ++g)
    for (size_t q = 0; q < lists[g].size(); ++q) host[1 + q * (size_t)grid + g] = lists[g][q];
  if (h->active_views.n < host.size()) h->active_views.alloc(host.size(), false);
  h->t.active_views = h->active_views.p;
  HIP_OK(hipMemcpyAsync(h->active_views.p, host.data(), host.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  sync(h);
  h->compact_dirty = true;   // (the descriptors of the compacted tables follow the active list)
  h->lin_grid = grid;
  API_END
}

int32_t mcba_residuals(mcba_handle h, const double* x, double* r) {
  API_BEGIN
  REQUIRE(h && x && r, "null argument");
  ensure_view_first(h);
  upload_x(h, x, h->x.p);
  eval_tables(h, h->x.p);
  h->ops->residual(h->d, h->t, h->stream, h->view_first.p, h->out_r.p, nullptr, nullptr, nullptr);
  HIP_OK(hipMemcpyAsync(r, h->out_r.p, 2 * (size_t)h->n_inliers * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  API_END
}

int32_t mcba_jacobian(mcba_handle h, const double* x, int32_t* row_nnz, double* vals, int32_t* cols) {
  API_BEGIN
  REQUIRE(h && row_nnz, "null argument");
  const Dims& d = h->d;
  int nnz = 0;
  if (d.off_campose >= 0) nnz += 6;
  if (d.off_boardpose >= 0) nnz += 6;
  if (d.off_motion >= 0) nnz += d.motion == MOTION_STATIC ? 6 : 12;
  if (d.off_cameras >= 0) nnz += 5 + d.ND;
  if (d.off_boards >= 0) nnz += 3;
  *row_nnz = nnz;
  if (!vals && !cols) return 0;
  REQUIRE(x && vals, "null argument");   // (cols == NULL: values only -- the pattern does not change between two evaluations)
  ensure_obs_index(h);
  const size_t nv = 2 * (size_t)h->n_inliers * nnz, nc = (size_t)h->n_inliers * nnz;
  h->out_big.alloc(std::max<size_t>(nv, 1), false);
  h->out_cols.alloc(std::max<size_t>(nc, 1), false);
  upload_x(h, x, h->x.p);
  eval_tables(h, h->x.p);
  h->ops->jacobian(d, h->t, h->stream, nnz, h->out_big.p, h->out_cols.p);
  HIP_OK(hipMemcpyAsync(vals, h->out_big.p, nv * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (cols) HIP_OK(hipMemcpyAsync(cols, h->out_cols.p, nc * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  API_END
}

int32_t mcba_reprojection_error(mcba_handle h, const double* x, double* err, uint8_t* valid) {
  API_BEGIN
  REQUIRE(h && x && err && valid, "null argument");
  g_fill_stream = h->stream;
  const Dims& d = h->d;
  const size_t nref = (size_t)d.C * d.F * d.B * d.P;
  h->out_big.alloc(nref, true);
  h->out_valid.alloc(nref, true);
  upload_x(h, x, h->x.p);
  eval_tables(h, h->x.p);
  h->ops->residual(d, h->t, h->stream, nullptr, nullptr, nullptr, h->out_big.p, h->out_valid.p);
  HIP_OK(hipMemcpyAsync(err, h->out_big.p, nref * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(valid, h->out_valid.p, nref, hipMemcpyDeviceToHost, h->stream));
  sync(h);
  API_END
}

int32_t mcba_project(mcba_handle h, const double* x, double* projected) {
  API_BEGIN
  REQUIRE(h && x && projected, "null argument");
  g_fill_stream = h->stream;
  const Dims& d = h->d;
  const size_t nref = (size_t)d.C * d.F * d.B * d.P;
  h->out_big.alloc(2 * nref, true);
  upload_x(h, x, h->x.p);
  eval_tables(h, h->x.p);
  h->ops->residual(d, h->t, h->stream, nullptr, nullptr, h->out_big.p, nullptr, nullptr);
  HIP_OK(hipMemcpyAsync(projected, h->out_big.p, 2 * nref * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  API_END
}

/* matrix.align_transforms_robust (transform/matrix.py:140-153) for a BATCH of problems on the device: the numeric core of
 * tables.estimate_transform (tables.py:153-176) and tables.relative_between_n (tables.py:334-345).  No handle: the
 * initialisation runs before a Calibration exists.  nA / nB = poses in A / B (ia, ib == nullptr: one per entry).
 * The stream and every buffer of a call are parked in the resource cache and taken back by the next call of the same
 * shape (an initialisation makes three calls; created and freed anew they cost 18 of its 32 ms at 16 x 1000 x 5). */
namespace {
void align_poses(int32_t n_problems, const int64_t* offsets, const double* A, int64_t nA, const int32_t* ia, const double* B,
                 int64_t nB, const int32_t* ib, const uint8_t* mask, double threshold, int32_t invert, double* out,
                 uint8_t* out_valid, uint8_t* inliers) {
  REQUIRE(n_problems >= 0 && offsets && A && B && out && out_valid, "bad argument");
  if (n_problems == 0) return;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    throw Error("no HIP device: the mcba back-end is GPU-only (there is no CPU fallback)");
  const int64_t total = offsets[n_problems];
  int64_t nmax = 1;
  for (int p = 0; p < n_problems; ++p) {
    REQUIRE(offsets[p + 1] >= offsets[p], "offsets must be non-decreasing");
    nmax = std::max<int64_t>(nmax, offsets[p + 1] - offsets[p]);
  }
  REQUIRE(offsets[0] == 0 && total >= 0, "offsets must start at 0");
  REQUIRE(nmax < (1ll << 24), "more than 2^24 pose pairs in one alignment problem");
  REQUIRE((ia == nullptr) == (ib == nullptr), "both index lists or none");
  if (ia != nullptr) {
    REQUIRE(nA > 0 && nB > 0 && nA < (1ll << 31) && nB < (1ll << 31), "bad pose table size");
    for (int64_t k = 0; k < total; ++k)
      REQUIRE(ia[k] >= 0 && ia[k] < nA && ib[k] >= 0 && ib[k] < nB, "pose index out of range");
  }
  const bool timing = getenv("MCBA_TIMING") != nullptr;
  const double t0 = now_seconds();
  hipStream_t st = resource_cache().take_stream();
  if (st == nullptr) HIP_OK(hipStreamCreate(&st));
  struct StreamGuard {     // the stream goes back to the resource cache (it is idle: every path below synchronises or throws)
    hipStream_t s;
    ~StreamGuard() {
      if (!resource_cache().park_stream(s)) (void)hipStreamDestroy(s);
    }
  } guard{st};
  g_fill_stream = st;
  {
  struct ParkReset { ~ParkReset() { g_park_on_release = false; } } park_reset;   // (declared first: runs after the buffers went)
  DevBuf<long long> d_off, d_prof;
  DevBuf<double> dA, dB, d_out, d_f64;
  DevBuf<uint8_t> d_mask, d_valid, d_inl;
  DevBuf<int> d_i32;
  DevBuf<int32_t> d_ia, d_ib;
  d_off.alloc((size_t)n_problems + 1, false);
  {
    static_assert(sizeof(long long) == sizeof(int64_t), "offsets are copied as they are");
    HIP_OK(hipMemcpyAsync(d_off.p, offsets, ((size_t)n_problems + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
  }
  const size_t tot = (size_t)std::max<int64_t>(total, 1);
  const size_t cntA = ia ? (size_t)nA : tot, cntB = ib ? (size_t)nB : tot;
  dA.alloc(16 * cntA, false);
  const bool same_table = ia != nullptr && A == B && nA == nB;
  if (!same_table) dB.alloc(16 * cntB, false);
  if (total > 0) {
    HIP_OK(hipMemcpyAsync(dA.p, A, 16 * (ia ? (size_t)nA : (size_t)total) * sizeof(double), hipMemcpyHostToDevice, st));
    if (!same_table) HIP_OK(hipMemcpyAsync(dB.p, B, 16 * (ib ? (size_t)nB : (size_t)total) * sizeof(double), hipMemcpyHostToDevice, st));
    if (ia != nullptr) {
      d_ia.alloc(tot, false);
      d_ib.alloc(tot, false);
      HIP_OK(hipMemcpyAsync(d_ia.p, ia, (size_t)total * sizeof(int32_t), hipMemcpyHostToDevice, st));
      HIP_OK(hipMemcpyAsync(d_ib.p, ib, (size_t)total * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
  }
  if (mask) {
    d_mask.alloc(tot, false);
    if (total > 0) HIP_OK(hipMemcpyAsync(d_mask.p, mask, (size_t)total, hipMemcpyHostToDevice, st));
  }
  d_out.alloc(16 * (size_t)n_problems, false);
  d_valid.alloc((size_t)n_problems, false);
  d_inl.alloc(tot, false);
  // per-problem scratch: 14 doubles + 7 ints per entry, sized for the largest problem
  const size_t per = (size_t)nmax, np_ = (size_t)n_problems;
  d_f64.alloc(np_ * per * 14, false);
  d_i32.alloc(np_ * per * 7, false);
  AlignScratch sc;
  sc.vec = d_f64.p;
  sc.cen = sc.vec + np_ * per * 6;
  sc.err = sc.cen + np_ * per * 6;
  sc.hgt = sc.err + np_ * per;
  sc.nd = sc.err;   // (nearest-neighbour distances of the clustering rounds: the error array is idle while a clustering runs)
  sc.size = d_i32.p;
  sc.chain = sc.size + np_ * per;
  sc.rep_a = sc.chain + np_ * per;
  sc.rep_b = sc.rep_a + np_ * per;
  sc.parent = sc.rep_b + np_ * per;
  sc.list = sc.parent + np_ * per;
  sc.live = sc.list + np_ * per;
  static const bool align_prof = dbg_switch("MCBA_ALIGN_PROF") != nullptr;
  sc.prof = nullptr;
  if (align_prof) {
    d_prof.alloc(np_ * 32);
    sc.prof = d_prof.p;
  }
  // dynamic LDS for the clustering state of problems with up to lds_cap selected entries (small batches of big problems get
  // the full 125 KB; batches of many small problems only what their largest problem needs, so that several fit a CU)
  // (a batch with a problem of more than ALIGN_LDS_CAP entries takes the full size: its error array -- 8 B per entry, up to
  //  18 150 entries -- is ranked from LDS too)
  const int lds_cap = (int)std::min<int64_t>(nmax, ALIGN_LDS_CAP);
  const size_t lds = align_lds_bytes(lds_cap);
  // (a per-DEVICE attribute: set on every call, it costs nothing next to the copies)
  HIP_OK(hipFuncSetAttribute((const void*)k_align_robust, hipFuncAttributeMaxDynamicSharedMemorySize, (int)align_lds_bytes(ALIGN_LDS_CAP)));
  const double t1 = now_seconds();
  static const bool no_staged = dbg_switch("MCBA_ALIGN_MONOLITHIC") != nullptr;
  // (the partial winners of the split scans take 24 B x ALIGN_SCAN_Z per entry and problem: batches beyond 4 M entries x problems
  //  -- 1.6 GB of them -- stay with the one-workgroup kernel)
  if (nmax > ALIGN_STAGED_MIN && n_problems <= 4096 && (size_t)n_problems * (size_t)nmax <= ((size_t)1 << 22) && !align_prof &&
      !no_staged) {
    // large problems: the rounds of the clustering as separate launches, the scans spread over the chip (k_align_stage_*)
    DevBuf<AlignStage> d_stage;
    DevBuf<int> d_done, d_pi;
    DevBuf<double> d_pf;
    d_stage.alloc(np_, true);
    d_done.alloc(2, true);
    d_pf.alloc(2 * np_ * ALIGN_SCAN_Z * per, false);     // partial winners of the split scans: key | d^2
    d_pi.alloc(2 * np_ * ALIGN_SCAN_Z * per + 2 * np_ * per, false);     // slot | size; + changed | work list
    unsigned long long* h_prog = (unsigned long long*)pinned_alloc(64);
    struct PinGuard { unsigned long long* p; ~PinGuard() { pinned_free(p, 64); } } pin_guard{h_prog};
    static std::atomic<unsigned long long> call_counter{0};
    AlignArgs a;
    a.off = d_off.p; a.A = dA.p; a.B = same_table ? dA.p : dB.p; a.ia = ia ? d_ia.p : nullptr; a.ib = ib ? d_ib.p : nullptr;
    a.mask = mask ? d_mask.p : nullptr; a.threshold = threshold; a.invert = (int)invert; a.scratch_stride = (long long)per;
    a.base = sc; a.out = d_out.p; a.out_valid = d_valid.p; a.inliers = d_inl.p; a.st = d_stage.p; a.done_count = d_done.p;
    a.host_progress = h_prog;
    a.pkey = d_pf.p; a.pd2 = d_pf.p + np_ * ALIGN_SCAN_Z * per; a.pidx = d_pi.p; a.pn = d_pi.p + np_ * ALIGN_SCAN_Z * per;
    a.changed = d_pi.p + 2 * np_ * ALIGN_SCAN_Z * per; a.work = a.changed + np_ * per;
    const int scan_y = (int)std::max<int64_t>(1, std::min<int64_t>(64, (nmax + 255) / 256));
    for (int pass = 0; pass < 2; ++pass) {
      const unsigned long long call = (++call_counter) & 0xffffull;
      *h_prog = 0ull;
      HIP_OK(hipMemsetAsync(d_done.p, 0, sizeof(int), st));
      hipLaunchKernelGGL(k_align_stage_pre, dim3(n_problems), dim3(ALIGN_THREADS), 0, st, a, pass);
      check_launch("k_align_stage_pre");
      // rounds: enqueued a few ahead of the progress word [call | problems whose clustering ended | round] that block 0 of every
      // merge kernel stores (kernels of finished problems return at once; at most nmax rounds can be needed)
      constexpr int LOOKAHEAD = 8;
      int enqueued = 0, seen_round = 0, spins = 0;
      double t_wait = 0.0;
      while (true) {
        const unsigned long long w = __atomic_load_n(h_prog, __ATOMIC_ACQUIRE);
        if ((w >> 48) == call) {
          const int rd = (int)(w & 0xffffffull);
          if (rd != seen_round) { seen_round = rd; t_wait = 0.0; spins = 0; }
          if ((int)((w >> 24) & 0xffffffull) >= n_problems) break;
        }
        if (enqueued - seen_round < LOOKAHEAD && enqueued <= nmax + 1) {
          hipLaunchKernelGGL(k_align_stage_scan, dim3(n_problems, scan_y, ALIGN_SCAN_Z), dim3(256), 0, st, a);
          ++enqueued;
          hipLaunchKernelGGL(k_align_stage_merge, dim3(n_problems), dim3(ALIGN_THREADS), 0, st, a, call, enqueued);
          if ((enqueued & 15) == 0) check_launch("k_align_stage rounds");
          t_wait = 0.0;
          spins = 0;
          continue;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if ((++spins & 1023) == 0) {
          const double now = now_seconds();
          if (t_wait == 0.0) t_wait = now;
          else if (now - t_wait > 0.05) {   // (a failed launch must not hang the caller)
            HIP_OK(hipStreamSynchronize(st));
            const unsigned long long w2 = __atomic_load_n(h_prog, __ATOMIC_ACQUIRE);
            REQUIRE((w2 >> 48) == call && ((int)(w2 & 0xffffffull) != seen_round || (int)((w2 >> 24) & 0xffffffull) >= n_problems),
                    "the staged alignment made no progress");
            t_wait = 0.0;
          }
        }
      }
      hipLaunchKernelGGL(k_align_stage_post, dim3(n_problems), dim3(ALIGN_THREADS), 0, st, a, pass);
      check_launch("k_align_stage_post");
    }
    HIP_OK(hipMemcpyAsync(out, d_out.p, 16 * (size_t)n_problems * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(out_valid, d_valid.p, (size_t)n_problems, hipMemcpyDeviceToHost, st));
    if (inliers && total > 0) HIP_OK(hipMemcpyAsync(inliers, d_inl.p, (size_t)total, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (timing) {
      std::vector<AlignStage> hs(np_);
      HIP_OK(hipMemcpy(hs.data(), d_stage.p, np_ * sizeof(AlignStage), hipMemcpyDeviceToHost));
      long long sl = 0, sw = 0, rd = 0;
      for (const AlignStage& q : hs) { sl += q.sum_live; sw += q.sum_work; rd = std::max<long long>(rd, q.rounds); }
      fprintf(stderr, "[align_poses] %d problems, %lld entries (staged): buffers + uploads %.2f ms, kernels + downloads %.2f ms; "
              "rounds of the last pass <= %lld, clusters alive / scanning summed over rounds %lld / %lld\n",
              n_problems, (long long)total, (t1 - t0) * 1e3, (now_seconds() - t1) * 1e3, rd, sl, sw);
    }
    g_park_on_release = true;
    return;
  }
  hipLaunchKernelGGL(k_align_robust, dim3(n_problems), dim3(ALIGN_THREADS), lds, st, d_off.p, dA.p,
                     (const double*)(same_table ? dA.p : dB.p), (const int32_t*)(ia ? d_ia.p : nullptr),
                     (const int32_t*)(ib ? d_ib.p : nullptr), (const uint8_t*)(mask ? d_mask.p : nullptr), threshold, (int)invert,
                     (long long)per, sc, d_out.p, d_valid.p, d_inl.p, lds_cap);
  check_launch("k_align_robust");
  HIP_OK(hipMemcpyAsync(out, d_out.p, 16 * (size_t)n_problems * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(out_valid, d_valid.p, (size_t)n_problems, hipMemcpyDeviceToHost, st));
  if (inliers && total > 0) HIP_OK(hipMemcpyAsync(inliers, d_inl.p, (size_t)total, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  if (timing)
    fprintf(stderr, "[align_poses] %d problems, %lld entries: buffers + uploads %.2f ms, kernel + downloads %.2f ms\n", n_problems,
            (long long)total, (t1 - t0) * 1e3, (now_seconds() - t1) * 1e3);
  if (align_prof) {   // phase cycles of the largest problem
    std::vector<long long> hp(np_ * 32);
    HIP_OK(hipMemcpy(hp.data(), d_prof.p, hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
    int big = 0;
    for (int p = 1; p < n_problems; ++p)
      if (offsets[p + 1] - offsets[p] > offsets[big + 1] - offsets[big]) big = p;
    const char* names[9] = {"compaction", "relative poses", "robust mean", "errors", "quantile + test", "  whitening", "  clustering",
                            "  cut+labels+mean", "  rounds (count)"};
    for (int pass = 0; pass < 2; ++pass)
      for (int k = 0; k < 9; ++k)
        fprintf(stderr, "[k_align_robust] problem %d (n = %lld) pass %d %-18s %10lld\n", big,
                (long long)(offsets[big + 1] - offsets[big]), pass, names[k], hp[(size_t)big * 32 + 16 * pass + k]);
  }
  g_park_on_release = true;   // regular end: the buffers of this scope are parked for the next call of the same shape
  }
}
}  // namespace

int32_t mcba_align_poses_robust(int32_t n_problems, const int64_t* offsets, const double* A, const double* B,
                                const uint8_t* mask, double threshold, int32_t invert, double* out, uint8_t* out_valid,
                                uint8_t* inliers) {
  API_BEGIN
  align_poses(n_problems, offsets, A, 0, nullptr, B, 0, nullptr, mask, threshold, invert, out, out_valid, inliers);
  API_END
}

int32_t mcba_align_poses_indexed(int32_t n_problems, const int64_t* offsets, const double* table_a, int64_t n_a,
                                 const int32_t* index_a, const double* table_b, int64_t n_b, const int32_t* index_b,
                                 const uint8_t* mask, double threshold, int32_t invert, double* out, uint8_t* out_valid,
                                 uint8_t* inliers) {
  API_BEGIN
  REQUIRE(index_a && index_b, "null index list");
  align_poses(n_problems, offsets, table_a, n_a, index_a, table_b, n_b, index_b, mask, threshold, invert, out, out_valid, inliers);
  API_END
}

int32_t mcba_project_model(mcba_handle h, const double* x, int32_t max_iterations, double* projected) {
  API_BEGIN
  REQUIRE(h && x && projected && max_iterations >= 0, "bad argument");
  g_fill_stream = h->stream;
  const Dims& d = h->d;
  const size_t nref = (size_t)d.C * d.F * d.B * d.P;
  h->out_big.alloc(2 * nref, true);
  upload_x(h, x, h->x.p);
  eval_tables(h, h->x.p);
  h->ops->project_model(d, h->t, h->stream, max_iterations, h->out_big.p);
  HIP_OK(hipMemcpyAsync(projected, h->out_big.p, 2 * nref * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  API_END
}

int32_t mcba_normal_equations(mcba_handle h, const double* x, const mcba_options* opt, double* cost, double* g,
                              double* diag) {
  API_BEGIN
  REQUIRE(h && x, "null argument");
  set_loss(h, opt);
  upload_x(h, x, h->x.p);
  launch_linearize(h, h->x.p);   // (k_tmat, the first kernel of the linearisation, prepares every table from x)
  launch_assemble(h);
  const Dims& d = h->d;
  if (g || diag) {   // frame-sharded: the caller of the HOST boundary gets complete vectors (the solver never needs them)
    gather_frame_entries(h, h->g());
    gather_frame_entries(h, h->diag());
  }
  // [g | diag | cost, count] comes down into pinned memory; only the two scalars when the vectors are not asked for
  const size_t first = (g || diag) ? 0 : 2 * (size_t)d.n, count = 2 * (size_t)d.n + 2 - first;
  HIP_OK(hipMemcpyAsync(h->h_gbuf + first, h->gbuf.p + first, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  if (g) to_caller(h, g, h->h_gbuf);
  if (diag) to_caller(h, diag, h->h_gbuf + d.n);
  if (cost) *cost = h->h_gbuf[2 * (size_t)d.n];
  API_END
}

// The same evaluation at the x that is already on the device (last mcba_normal_equations / mcba_solve), enqueued without
// any host transfer or synchronisation: this is how the trust-region driver itself evaluates (x, tables and the normal
// equations never leave HBM).  mcba_synchronize waits for the handle's stream.
int32_t mcba_normal_equations_device(mcba_handle h, const mcba_options* opt) {
  API_BEGIN
  REQUIRE(h, "null handle");
  set_loss(h, opt);
  // (replaying the four launches as a hipGraph was measured in round 3: 82.3 against 73.7 us at the north-star rig on
  //  ROCm 7.2 -- the stream-ordered launches already follow each other after ~1 us -- and is gone)
  launch_linearize(h, h->x.p);
  launch_assemble(h);
  API_END
}

int32_t mcba_synchronize(mcba_handle h) {
  API_BEGIN
  REQUIRE(h, "null handle");
  sync(h);
  API_END
}

int32_t mcba_dense_hessian(mcba_handle h, double* H) {
  API_BEGIN
  REQUIRE(h && H, "null argument");
  const Dims& d = h->d;
  const size_t nn = (size_t)d.n * d.n;
  REQUIRE(nn <= (1u << 26), "dense Hessian too large (debug API)");
  h->out_big.alloc(nn, false);
  hipLaunchKernelGGL(k_dense_hessian, dim3(std::min<size_t>(4096, (nn + 255) / 256)), dim3(256), 0, h->stream, d,
                     h->Hss.p, h->Hfs.p, h->Hff.p, h->out_big.p);
  if (h->ext2int.empty()) {
    HIP_OK(hipMemcpyAsync(H, h->out_big.p, nn * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    sync(h);
  } else {   // rows and columns of the caller's (ragged) parameter order
    std::vector<double> full(nn);
    HIP_OK(hipMemcpyAsync(full.data(), h->out_big.p, nn * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    sync(h);
    const int ne = h->n_ext;
    for (int i = 0; i < ne; ++i)
      for (int j = 0; j < ne; ++j) H[(size_t)i * ne + j] = full[(size_t)h->ext2int[i] * d.n + h->ext2int[j]];
  }
  API_END
}

// debug: the regularised Gauss-Newton direction in the scaled space for a given damping (parity tests of the Schur /
// Cholesky kernels).  Requires a preceding mcba_normal_equations at the same x.  gn_h and g_h are [n_params].
int32_t mcba_debug_gn_step(mcba_handle h, double reg, double* gn_h, double* g_h, double* scale_inv) {
  API_BEGIN
  REQUIRE(h && gn_h, "null argument");
  const Dims& d = h->d;
  hipLaunchKernelGGL(k_vec_scale, dim3(h->sl.nvb), dim3(256), 0, h->stream, d, h->x.p, h->g(), h->diag(), h->scale_inv.p,
                     h->dsc.p, h->gh.p, 1, h->scal.p + h->sl.vs, (const double*)nullptr, (double*)nullptr);
  launch_gn_solve(h, reg, h->shard_root);
  if (h->ext2int.empty()) {
    HIP_OK(hipMemcpyAsync(gn_h, h->gn.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (g_h) HIP_OK(hipMemcpyAsync(g_h, h->gh.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (scale_inv)
      HIP_OK(hipMemcpyAsync(scale_inv, h->scale_inv.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    sync(h);
  } else {
    h->xmap_tmp.resize(3 * (size_t)d.n);
    double* tmp = h->xmap_tmp.data();
    HIP_OK(hipMemcpyAsync(tmp, h->gn.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipMemcpyAsync(tmp + d.n, h->gh.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipMemcpyAsync(tmp + 2 * d.n, h->scale_inv.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    sync(h);
    to_caller(h, gn_h, tmp);
    if (g_h) to_caller(h, g_h, tmp + d.n);
    if (scale_inv) to_caller(h, scale_inv, tmp + 2 * d.n);
  }
  int info = 0;
  HIP_OK(hipMemcpy(&info, h->info.p, sizeof(int), hipMemcpyDeviceToHost));
  REQUIRE(info == 0, "Cholesky of the reduced system hit a non-positive pivot at column " + std::to_string(info));
  API_END
}

// debug: per-view cycle stamps of the k_linearize phases [views][8] (setup, rows, stage+mfma, epilogue, count, t0, t1)
int32_t mcba_debug_linearize_profile(mcba_handle h, const double* x, long long* out) {
  API_BEGIN
  REQUIRE(h && x && out, "null argument");
  const size_t n = ((size_t)h->d.views() + (h->d.views() + TMV - 1) / TMV) * 8;   // (+ one row per k_tmat workgroup)
  h->dbg.alloc(n, true);
  h->t.dbg = h->dbg.p;
  upload_x(h, x, h->x.p);
  launch_linearize(h, h->x.p);
  h->t.dbg = nullptr;
  HIP_OK(hipMemcpyAsync(out, h->dbg.p, n * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  API_END
}


// debug: solve (S + reg I) p = rhs with the device Cholesky (blocked != 0 forces the multi-workgroup path)
int32_t mcba_debug_chol(mcba_handle h, int32_t ns, const double* S, const double* rhs, double reg, int32_t blocked,
                        double* p_out) {
  API_BEGIN
  REQUIRE(h && S && rhs && p_out && ns > 0, "bad argument");
  DevBuf<double> buf, ps;
  buf.alloc((size_t)ns * ns + ns, false);
  ps.alloc((size_t)ns);
  HIP_OK(hipMemcpyAsync(buf.p, S, (size_t)ns * ns * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(buf.p + (size_t)ns * ns, rhs, (size_t)ns * sizeof(double), hipMemcpyHostToDevice, h->stream));
  REQUIRE(blocked == 0 || blocked == 1 || blocked == 4 || blocked == 6 || blocked == 7, "unknown Cholesky path");
  g_force_blocked_chol = blocked == 1;   // 0: automatic, 1: multi-workgroup kernels (k_cholb_*)
  g_force_panel2_chol = blocked == 6;    // 6: multi-launch panel kernels (k_cholp_*: the default for 160 < ns + 1 <= 1024)
  DevBuf<long long> stamps;
  if (blocked == 4) {                    // 4: k_chol_blk with phase stamps; p_out[0..7] receives the shader-clock totals
    REQUIRE(ns >= 8 && ns + 1 <= CHOL_BLK_MAX_N1, "profiling needs 8 <= ns < 160");
    stamps.alloc(8);
    g_chol_prof = stamps.p;
  }
  if (blocked == 7) {                    // 7: k_cholp_panel with phase stamps summed over the panels (p_out[0..5])
    REQUIRE(ns >= 8 && ns + 1 <= CHOLP_MAX_N1, "profiling needs 8 <= ns < 1024");
    stamps.alloc(8);
    g_chol_prof = stamps.p;
    g_force_panel2_chol = true;
  }
  try { launch_chol(h, ns, reg, buf.p, ps.p); }
  catch (...) { g_force_blocked_chol = g_force_panel2_chol = false; g_chol_prof = nullptr; throw; }
  g_force_blocked_chol = g_force_panel2_chol = false;
  g_chol_prof = nullptr;
  if (blocked == 4 || blocked == 7) {
    long long st[8];
    sync(h);
    HIP_OK(hipMemcpy(st, stamps.p, sizeof(st), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) p_out[i] = (double)st[i];
    return 0;
  }
  HIP_OK(hipMemcpyAsync(p_out, ps.p, (size_t)ns * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  int info = 0;
  HIP_OK(hipMemcpy(&info, h->info.p, sizeof(int), hipMemcpyDeviceToHost));
  REQUIRE(info == 0, "non-positive pivot at column " + std::to_string(info));
  API_END
}

// debug: one MFMA f64 16x16x4 with an asymmetric operand pair; out[16][16] = A^T B for V = [A | B] (4 x 32)
int32_t mcba_debug_mfma_probe(const double* V, double* out) {
  API_BEGIN
  DevBuf<double> dv, dout;
  dv.upload(std::vector<double>(V, V + 128));
  dout.alloc(256);
  hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, 0, dv.p, dout.p);
  HIP_OK(hipMemcpy(out, dout.p, 256 * sizeof(double), hipMemcpyDeviceToHost));
  API_END
}

// debug: workgroup dispatch rate (k_dispatch_probe): out[blocks][2] = 100 MHz wall-clock ticks at start / end of a workgroup
int32_t mcba_debug_dispatch_probe(int32_t blocks, int32_t threads, int32_t lds_bytes, int32_t spin, long long* out) {
  API_BEGIN
  REQUIRE(out && blocks > 0 && threads > 0 && threads <= 1024 && lds_bytes >= 0 && lds_bytes <= 64 * 1024, "bad argument");
  DevBuf<long long> buf;
  buf.alloc((size_t)blocks * 2);
  for (int rep = 0; rep < 3; ++rep)   // (the last launch is the one reported: code and buffers warm)
    hipLaunchKernelGGL(k_dispatch_probe, dim3(blocks), dim3(threads), (size_t)lds_bytes, 0, spin, buf.p);
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(out, buf.p, (size_t)blocks * 2 * sizeof(long long), hipMemcpyDeviceToHost));
  API_END
}

// debug: cycles ONE workgroup needs to read n doubles that workgroup `region` of the preceding kernel wrote (8 writer workgroups:
// region 0 shares the reader's XCD if the dispatcher starts every kernel on XCD 0); out[r] for r = 0 .. 7, then out[8] = the
// same region read a second time (warm in the reader's own caches)
int32_t mcba_debug_xcd_probe(int32_t n, long long* out) {
  API_BEGIN
  REQUIRE(out && n > 0 && n <= (1 << 22), "bad argument");
  DevBuf<double> buf, sink;
  DevBuf<long long> cyc;
  buf.alloc((size_t)8 * n, false);
  sink.alloc(512);
  cyc.alloc(16);
  for (int rep = 0; rep < 2; ++rep)
    for (int r = 0; r < 9; ++r) {
      if (r < 8) hipLaunchKernelGGL(k_xcd_write, dim3(8), dim3(256), 0, 0, buf.p, n);
      hipLaunchKernelGGL(k_xcd_read, dim3(1), dim3(512), 0, 0, (const double*)buf.p, n, r < 8 ? r : 7, sink.p, cyc.p + r);
    }
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(out, cyc.p, 9 * sizeof(long long), hipMemcpyDeviceToHost));
  API_END
}

// debug: FP64 VALU / FP64 MFMA pipe-sharing probe (k_pipe_probe); ms_out[3] = milliseconds of modes 0, 1, 2
int32_t mcba_debug_pipe_probe(int32_t iters, double* ms_out) {
  API_BEGIN
  REQUIRE(ms_out && iters > 0, "bad argument");
  DevBuf<double> sink;
  sink.alloc(8);
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  // (MCBA_PROBE_THREADS / MCBA_PROBE_BLOCKS: other occupancies, e.g. 256 threads = one wavefront per SIMD)
  const int thr = dbg_switch("MCBA_PROBE_THREADS") ? atoi(dbg_switch("MCBA_PROBE_THREADS")) : 512;
  const int blk = dbg_switch("MCBA_PROBE_BLOCKS") ? atoi(dbg_switch("MCBA_PROBE_BLOCKS")) : 256;
  REQUIRE(thr >= 64 && thr <= 512 && thr % 64 == 0 && blk > 0, "bad probe shape");
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k_pipe_probe, dim3(blk), dim3(thr), 0, 0, mode, 16, sink.p);   // warm-up
    HIP_OK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_pipe_probe, dim3(blk), dim3(thr), 0, 0, mode, iters, sink.p);
    HIP_OK(hipEventRecord(e1, 0));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    ms_out[mode] = ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  API_END
}

/* test hook (mcba_debug.h): [sum a b | sum a^2 | sum b^2] over n doubles by k_dot (one workgroup) and by k_dot3_part / k_dot3_fin (its 16
 * wavefronts on 16 CUs, the order of additions unchanged): out_single[3], out_wide[3] -- must agree bit for bit.  */
int32_t mcba_debug_dot3(const double* a, const double* b, int64_t n, double* out_single, double* out_wide) {
  API_BEGIN
  REQUIRE(a && b && n >= 0 && out_single && out_wide, "bad argument");
  DevBuf<double> da, db, part, out;
  da.alloc((size_t)std::max<int64_t>(n, 1)); db.alloc((size_t)std::max<int64_t>(n, 1)); part.alloc(3 * DOT_WAVES, true); out.alloc(6, true);
  if (n > 0) {
    HIP_OK(hipMemcpy(da.p, a, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db.p, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(k_dot, dim3(1), dim3(1024), 0, 0, (size_t)n, (const double*)da.p, (const double*)db.p, out.p, 1);
  hipLaunchKernelGGL(k_dot3_part, dim3(DOT_WAVES), dim3(64), 0, 0, (size_t)n, (const double*)da.p, (const double*)db.p, part.p);
  hipLaunchKernelGGL(k_dot3_fin, dim3(1), dim3(64), 0, 0, (const double*)part.p, out.p + 3);
  HIP_OK(hipDeviceSynchronize());
  double host[6];
  HIP_OK(hipMemcpy(host, out.p, sizeof(host), hipMemcpyDeviceToHost));
  for (int k = 0; k < 3; ++k) { out_single[k] = host[k]; out_wide[k] = host[3 + k]; }
  API_END
}

namespace {

struct LsmrOps {
  mcba_handle_s* h;
  int nblk;            // persistent single-wave workgroups of the two Jacobian products
  int part_stride;
  size_t m;            // residuals of THIS handle (its frame shard)
  bool trace = false;  // MCBA_SOLVE_TRACE (debug switch): per-solve lines on stderr
  size_t m_global = 0; // residuals of the whole problem (frame-sharded: summed over the ranks): scipy's maxiter = min(m, n)
  long long maxiter_override = 0;   // > 0: lsmr(..., maxiter = this) (test hook)
  bool sharded() const { return h->allreduce != nullptr; }
  double* bpart() const { return h->d.off_boards >= 0 ? h->ls_bpart.p : nullptr; }   // boards=True: jp^T u per observation
  LsmrGatherExtra extra() const { return LsmrGatherExtra{h->obs_index.p, h->board_off.p, bpart(), sharded() ? 1 : 0}; }
  int gather_grid() const {   // one wavefront per entry outside the per-frame pose block + one per frame (k_lsmr_gather)
    const int nfe = lsmr_gather_frame_entries(h->d);
    return h->d.n - nfe + (nfe > 0 ? h->d.Fl : 0);
  }
  // `count` doubles at `dev`: summed over the ranks of a frame-sharded problem (in place), then copied to the host
  void fetch_sum(double* dev, int count, double* out) {
    if (sharded()) call_allreduce(h, dev, (size_t)count, 0);
    HIP_OK(hipMemcpyAsync(h->h_scal, dev, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    sync(h);
    for (int k = 0; k < count; ++k) out[k] = h->h_scal[k];
  }
  double fold(double* part, int n) {   // sum of n partials (fixed order), one double back to the host
    hipLaunchKernelGGL(k_dot, dim3(1), dim3(1024), 0, h->stream, (size_t)n, (const double*)part, (const double*)nullptr, h->ls_out.p, 0);
    double v = 0.0;
    fetch_sum(h->ls_out.p, 1, &v);
    return v;
  }
  // u <- J_h v - alpha u (mode 0) | f (mode 1) | J_h v (mode 2); returns |u|^2 (over all ranks)
  double jv(int mode, const double* vin, double alpha, double* u) {
    h->ops->lsmr_jv(h->d, h->t, h->stream, h->view_first.p, mode, h->dsc.p, vin, alpha, u, h->ls_partial.p, nblk, nullptr);
    check_launch("k_lsmr_jv");
    return fold(h->ls_partial.p, nblk);
  }
  // the per-view partials of k_lsmr_jtu -> vout = D J^T u - beta vold, nrm[i] = (weighted) vout[i]^2.  Frame-sharded: the
  // shared entries are a sum over the ranks -- ONE all-reduce of ns doubles (k_lsmr_shard_pack / _finish).
  void gather(double beta, const double* vold, double* vout, const double* ls) {
    const Dims& d = h->d;
    hipLaunchKernelGGL(k_lsmr_gather, dim3(gather_grid()), dim3(64), 0, h->stream, d, (const double*)h->ls_part.p, part_stride,
                       (const double*)h->dsc.p, beta, vold, vout, h->ls_nrm.p, ls, extra());
    if (sharded()) {
      const int ns = std::max(d.ns, 1);
      if (h->ls_comm.n < (size_t)ns) h->ls_comm.alloc((size_t)ns, true);
      hipLaunchKernelGGL(k_lsmr_shard_pack, dim3((ns + 255) / 256), dim3(256), 0, h->stream, d, (const double*)vout, h->ls_comm.p);
      call_allreduce(h, h->ls_comm.p, (size_t)d.ns, 0);
      hipLaunchKernelGGL(k_lsmr_shard_finish, dim3((d.n + 255) / 256), dim3(256), 0, h->stream, d, (const double*)h->ls_comm.p,
                         (const double*)h->dsc.p, beta, vold, vout, h->ls_nrm.p, ls);
    }
    check_launch("k_lsmr_gather");
  }
  // vout <- D J^T (u inv_beta) - beta vold (u normalised in place); returns |vout|^2
  double jtu(double* u, double inv_beta, double beta, const double* vold, double* vout) {
    h->ops->lsmr_jtu(h->d, h->t, h->stream, h->view_first.p, inv_beta, u, h->ls_part.p, part_stride, bpart(), nblk, nullptr);
    gather(beta, vold, vout, nullptr);
    return fold(h->ls_nrm.p, h->d.n);
  }
  // one iteration of the device-resident solve (mcba_lsmr.h): six launches, every scalar read from the state `ls`
  // (frame-sharded: + three small all-reduces -- [|u|^2, |x|^2] (2 doubles), the shared sums of J^T u (ns), |v|^2 (1))
  void iteration(double* ls, double* u, double* v, double* vraw, unsigned long long call) {
    const Dims& d = h->d;
    const int nvb = (d.n + 255) / 256;
    h->ops->lsmr_jv(d, h->t, h->stream, h->view_first.p, 0, h->dsc.p, v, 0.0, u, h->ls_partial.p, nblk, ls);
    if (sharded()) {
      double* two = h->ls_out.p + 4;
      hipLaunchKernelGGL(k_lsmr_shard_fold_a, dim3(1), dim3(1024), 0, h->stream, d, (const double*)h->ls_partial.p, nblk,
                         (const double*)h->ls_nrm.p, two);
      call_allreduce(h, two, 2, 0);
      hipLaunchKernelGGL(k_lsmr_scal_a, dim3(1), dim3(1024), 0, h->stream, ls, (const double*)two, 1, (const double*)(two + 1), 1,
                         call, h->h_pub_seq + 1);
    } else {
      hipLaunchKernelGGL(k_lsmr_scal_a, dim3(1), dim3(1024), 0, h->stream, ls, (const double*)h->ls_partial.p, nblk,
                         (const double*)h->ls_nrm.p, d.n, call, h->h_pub_seq + 1);
    }
    h->ops->lsmr_jtu(d, h->t, h->stream, h->view_first.p, 0.0, u, h->ls_part.p, part_stride, bpart(), nblk, ls);
    gather(0.0, v, vraw, ls);
    if (sharded()) {
      double* one = h->ls_out.p + 6;
      hipLaunchKernelGGL(k_dot, dim3(1), dim3(1024), 0, h->stream, (size_t)d.n, (const double*)h->ls_nrm.p, (const double*)nullptr, one, 0);
      call_allreduce(h, one, 1, 0);
      hipLaunchKernelGGL(k_lsmr_scal_b, dim3(1), dim3(1024), 0, h->stream, ls, (const double*)one, 1);
    } else {
      hipLaunchKernelGGL(k_lsmr_scal_b, dim3(1), dim3(1024), 0, h->stream, ls, (const double*)h->ls_nrm.p, d.n);
    }
    hipLaunchKernelGGL(k_lsmr_update, dim3(nvb), dim3(256), 0, h->stream, d.n, 1.0, 0.0, 0.0, 0.0, vraw, h->ls_hbar.p, h->ls_x.p,
                       h->ls_h.p, h->ls_nrm.p, (const double*)ls);
  }
  // Round 5: the same iteration in THREE launches -- k_lsmr_fused (both products from one evaluation of the analytic rows),
  // k_lsmr_gather2 (+ beta, stopping tests), k_lsmr_update2 (+ alpha, rotations); state double-buffered A -> B -> A
  // (mcba_solver_kernels.h).  Frame-sharded: the same three collectives as above between them.
  void iteration_fused(double* lsA, double* lsB, double* u, double* v, double* vraw, unsigned long long call) {
    const Dims& d = h->d;
    const int nvb = (d.n + LSG_THREADS - 1) / LSG_THREADS;
    h->ops->lsmr_fused(d, h->t, h->stream, h->view_first.p, h->dsc.p, v, u, h->ls_partial.p, h->ls_part.p, part_stride, bpart(), nblk, lsA);
    const double* upart = h->ls_partial.p;
    const double* xpart = h->ls_xpart.p;
    int nu = nblk, nx = nvb;
    if (sharded()) {
      double* two = h->ls_out.p + 4;
      hipLaunchKernelGGL(k_lsmr_shard_fold_a2, dim3(1), dim3(LSG_THREADS), 0, h->stream, upart, nu, xpart, nx, two);
      call_allreduce(h, two, 2, 0);
      upart = two; nu = 1; xpart = two + 1; nx = 1;
    }
    hipLaunchKernelGGL(k_lsmr_gather2, dim3((gather_grid() + LSG_THREADS / 64 - 1) / (LSG_THREADS / 64)), dim3(LSG_THREADS), 0, h->stream, d,
                       (const double*)h->ls_part.p, part_stride, (const double*)h->dsc.p, (const double*)v, vraw, h->ls_nrm.p,
                       (const double*)lsA, lsB, upart, nu, xpart, nx, call, h->h_pub_seq + 1, extra());
    const double* vpart = h->ls_nrm.p;
    int nv = d.n;
    if (sharded()) {
      const int ns = std::max(d.ns, 1);
      if (h->ls_comm.n < (size_t)ns) h->ls_comm.alloc((size_t)ns, true);
      hipLaunchKernelGGL(k_lsmr_shard_pack, dim3((ns + 255) / 256), dim3(256), 0, h->stream, d, (const double*)vraw, h->ls_comm.p);
      call_allreduce(h, h->ls_comm.p, (size_t)d.ns, 0);
      hipLaunchKernelGGL(k_lsmr_shard_finish, dim3((d.n + 255) / 256), dim3(256), 0, h->stream, d, (const double*)h->ls_comm.p,
                         (const double*)h->dsc.p, 0.0, (const double*)v, vraw, h->ls_nrm.p, (const double*)lsB, 1);
      double* one = h->ls_out.p + 6;
      hipLaunchKernelGGL(k_dot, dim3(1), dim3(1024), 0, h->stream, (size_t)d.n, (const double*)h->ls_nrm.p, (const double*)nullptr, one, 0);
      call_allreduce(h, one, 1, 0);
      vpart = one; nv = 1;
    }
    hipLaunchKernelGGL(k_lsmr_update2, dim3(nvb), dim3(LSG_THREADS), 0, h->stream, d, (const double*)lsB, lsA, vpart, nv, vraw, h->ls_hbar.p,
                       h->ls_x.p, h->ls_h.p, h->ls_xpart.p);
  }
  // ... and in TWO launches: k_lsmr_fused2 carries the rotation + vector update of the previous step in its head, k_lsmr_gather3
  // folds its partials in a fifth wavefront beside the sums.  v is kept un-normalised (v = v_raw / alpha, alpha in the state).
  // State: k_lsmr_fused2 reads s0 (written by the gather / the initialisation) and writes s1; the gather reads s1 and writes s0.
  // k_lsmr_gather3: one workgroup per entry outside the frame block, one per four frames, + the publisher workgroup (no tasks)
  int gather3_grid() const {
    const int nfe = lsmr_gather_frame_entries(h->d);
    return (h->d.n - nfe) + ((nfe > 0 ? h->d.Fl : 0) + 3) / 4 + 1;
  }
  // per-view partials of the two-launch iteration in the TRANSPOSED layout (lsmr_part_index); views without inliers stay zero
  void ensure_part2(bool clear) {
    const size_t need = (size_t)std::max(h->d.views(), 1) * (size_t)(6 * h->d.NPB + h->d.KI);
    if (h->ls_part2.n < need) h->ls_part2.alloc(need, true);
    else if (clear) HIP_OK(hipMemsetAsync(h->ls_part2.p, 0, h->ls_part2.n * sizeof(double), h->stream));
  }
  void iteration_fused2(double* s0, double* s1, double* u, double* v, double* vraw, unsigned long long call, bool first_iteration) {
    const Dims& d = h->d;
    const double* vpart = h->ls_vpart.p;
    int nv = gather3_grid();
    if (sharded()) { vpart = h->ls_out.p + 6; nv = 1; }     // (|v_raw|^2 over all ranks, formed by k_lsmr_shard_finish2 of the previous iteration)
    // source of the observations (k_lsmr_fused2's MODE): the compacted tables, except with boards=True (masks form); lsmr_fused == 3:
    // the first iteration of a solve also stores the state of every observation, the others stream it back
    int mode = (d.off_boards < 0 && !h->lsmr_masks_form) ? 3 : 0;
    if (h->lsmr_fused == 3 && mode == 3) mode = first_iteration ? 4 : 2;
    if (mode == 4) {
      const size_t need = (((size_t)h->n_inliers + 63) / 64) * 64 * (size_t)lsmr_cache_components(d.motion, d.loss) + 64;
      if (h->ls_cache.n < need) h->ls_cache.alloc(need, false);
    }
    if (mode != 0) ensure_compact(h);
    h->ops->lsmr_fused2(d, h->t, h->stream, h->view_first.p, h->dsc.p, v, u, h->ls_partial.p, h->ls_xpart.p, h->ls_part2.p, part_stride,
                        bpart(), nblk, s0, s1, vpart, nv, h->ls_hbar.p, h->ls_x.p, h->ls_h.p, h->ls_cache.p, mode, compact_tables(h));
    // (the vector update is spread 64 entries per workgroup: only the first ceil(n / 64) workgroups hold a part of |x|^2)
    const int nu = nblk, nx = std::max(1, std::min(nblk, (d.n + 63) / 64));
    if (!sharded()) {
      hipLaunchKernelGGL(k_lsmr_gather3, dim3(gather3_grid()), dim3(LSG3_THREADS), 0, h->stream, d, (const double*)h->ls_part2.p, part_stride,
                         (const double*)h->dsc.p, (const double*)v, vraw, h->ls_nrm.p, h->ls_vpart.p, (const double*)s1, s0,
                         (const double*)h->ls_partial.p, nu, (const double*)h->ls_xpart.p, nx, call, h->h_pub_seq + 1, extra());
      return;
    }
    // Frame-sharded: ONE collective per iteration (k_lsmr_shard_pack2 / k_lsmr_shard_finish2, mcba_solver_kernels.h).  The gather only
    // forms raw sums (beta is not known before the message is back); message = [shared sums (ns) | |uhat|^2 | |x|^2 | a | b | c].
    LsmrGatherExtra ex = extra();
    ex.raw_shared = 2;
    hipLaunchKernelGGL(k_lsmr_gather3, dim3(gather3_grid()), dim3(LSG3_THREADS), 0, h->stream, d, (const double*)h->ls_part2.p, part_stride,
                       (const double*)h->dsc.p, (const double*)v, vraw, h->ls_nrm.p, h->ls_vpart.p, (const double*)s1, s0,
                       (const double*)h->ls_partial.p, nu, (const double*)h->ls_xpart.p, nx, call, h->h_pub_seq + 1, ex);
    const size_t msg = (size_t)d.ns + 5;
    if (h->ls_comm.n < msg) h->ls_comm.alloc(msg, true);
    hipLaunchKernelGGL(k_lsmr_shard_pack2, dim3(1), dim3(LSP_THREADS), 0, h->stream, d, (const double*)vraw, (const double*)v,
                       (const double*)h->dsc.p, (const double*)s1, (const double*)h->ls_partial.p, nu, (const double*)h->ls_xpart.p, nx,
                       h->ls_comm.p);
    call_allreduce(h, h->ls_comm.p, msg, 0);
    hipLaunchKernelGGL(k_lsmr_shard_finish2, dim3(1), dim3(LSP_THREADS), 0, h->stream, d, (const double*)h->ls_comm.p, (const double*)h->dsc.p,
                       (const double*)v, vraw, (const double*)s1, s0, h->ls_out.p + 6, call, h->h_pub_seq + 1);
  }
};

// every buffer the lsmr route needs on this handle (solve_lsmr, the debug and the timing entry points); idempotent
static LsmrOps lsmr_setup(mcba_handle_s* h) {
  const Dims& d = h->d;
  h->lsmr_fused = h->lsmr_fused_setting >= 0 ? h->lsmr_fused_setting : (d.motion == MOTION_ROLLING || d.off_boards >= 0 ? 2 : 3);
  ensure_view_first(h);
  const size_t m = 2 * (size_t)h->n_inliers;
  const int NL = 6 * d.NPB + d.KI;
  LsmrOps op{h, std::max(1, std::min(h->lsmr_grid, d.views())), (NL + 1) & ~1, m};
  for (DevBuf<double>* b : {&h->ls_u, &h->ls_ua, &h->ls_ub})
    if (b->n < std::max<size_t>(m, 2)) b->alloc(std::max<size_t>(m, 2), false);
  if (h->ls_part.n < (size_t)std::max(d.views(), 1) * op.part_stride) h->ls_part.alloc((size_t)std::max(d.views(), 1) * op.part_stride, true);
  else HIP_OK(hipMemsetAsync(h->ls_part.p, 0, h->ls_part.n * sizeof(double), h->stream));
  for (DevBuf<double>* b : {&h->ls_v, &h->ls_vraw, &h->ls_h, &h->ls_hbar, &h->ls_x, &h->ls_nrm})
    if (b->n < (size_t)d.n) b->alloc((size_t)d.n, true);
  if (h->ls_partial.n < (size_t)op.nblk) h->ls_partial.alloc((size_t)op.nblk, false);
  if (h->ls_out.n < (size_t)LS_NSLOTS + 8) h->ls_out.alloc((size_t)LS_NSLOTS + 8, false);
  if (h->ls_state.n < (size_t)2 * LS_NSLOTS) h->ls_state.alloc((size_t)2 * LS_NSLOTS, true);   // [A | B] (fused iterations)
  const size_t nvb = (size_t)(d.n + 255) / 256;
  if (h->ls_xpart.n < std::max(nvb, (size_t)op.nblk) + 1) h->ls_xpart.alloc(std::max(nvb, (size_t)op.nblk) + 1, true);
  if (h->ls_vpart.n < (size_t)op.gather3_grid() + 1) h->ls_vpart.alloc((size_t)op.gather3_grid() + 1, true);
  op.ensure_part2(true);
  if (d.off_boards >= 0) {   // boards=True: jp^T u per observation + the residual index of every slot (k_lsmr_gather)
    if (h->ls_bpart.n < std::max<size_t>(3 * (size_t)h->n_inliers, 3)) h->ls_bpart.alloc(std::max<size_t>(3 * (size_t)h->n_inliers, 3), false);
    ensure_obs_index(h);
  }
  return op;
}

// normr, normar, normA, condA, normx of the finished lsmr_solve from its state block (a copy + a wait: debug / trace only)
void lsmr_fetch_scalars(mcba_handle_s* h, mcba_handle_s::LsmrCall& c);

// the state block that holds the end of the last lsmr_solve (where the stopping tests of each iteration form write it)
const double* lsmr_final_state(const mcba_handle_s* h) {
  return h->lsmr_fused == 1 ? h->ls_state.p + LS_NSLOTS : h->ls_state.p;
}

void lsmr_fetch_scalars(mcba_handle_s* h, mcba_handle_s::LsmrCall& c) {
  double L[LS_NSLOTS];
  HIP_OK(hipMemcpyAsync(L, lsmr_final_state(h), sizeof(L), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  if (c.itn == 0) {   // (x = 0 returned from the prologue: the state was never initialised for this call)
    c.normr = c.normar = c.normA = c.condA = c.normx = std::numeric_limits<double>::quiet_NaN();
    return;
  }
  c.normr = L[LS_NORMR]; c.normar = L[LS_NORMAR]; c.normA = L[LS_NORMA]; c.condA = L[LS_CONDA]; c.normx = std::sqrt(L[LS_X2]);
}

// scipy.sparse.linalg.lsmr(J_h, f, damp, atol = btol = 1e-6, conlim = 1e8, maxiter = min(m, n)) -- the call of trf.py:481 --
// with the two products on the device and the scalar recurrences (lsmr.py:300-420, transcribed in order) on the host.
// The solution is left in h->ls_x; returns the number of iterations, *istop_out = scipy's stopping reason.
int lsmr_solve(LsmrOps& op, double damp, int* istop_out) {
  mcba_handle_s* h = op.h;
  const Dims& d = h->d;
  const int n = d.n;
  long long maxiter = std::min<long long>((long long)(op.m_global ? op.m_global : op.m), (long long)(h->ext2int.empty() ? n : h->n_ext));
  if (op.maxiter_override > 0) maxiter = op.maxiter_override;   // (mcba_debug_lsmr_solve: scipy's `maxiter` argument)
  const int nvb = (n + 255) / 256;
  double* u = h->ls_u.p;
  double* v = h->ls_v.p;
  double* vraw = h->ls_vraw.p;
  HIP_OK(hipMemsetAsync(h->ls_x.p, 0, (size_t)n * sizeof(double), h->stream));
  HIP_OK(hipMemsetAsync(h->ls_hbar.p, 0, (size_t)n * sizeof(double), h->stream));
  HIP_OK(hipMemsetAsync(v, 0, (size_t)n * sizeof(double), h->stream));
  const double normb = std::sqrt(op.jv(1, nullptr, 0.0, u));     // u = b = f
  double beta = normb, alpha = 0.0;
  if (beta > 0) {
    alpha = std::sqrt(op.jtu(u, 1.0 / beta, 0.0, v, vraw));       // v = A^T u (u normalised in place)
    std::swap(v, vraw);
  }
  if (alpha > 0) hipLaunchKernelGGL(k_scale_to, dim3(nvb), dim3(256), 0, h->stream, n, 1.0 / alpha, (const double*)v, v);
  HIP_OK(hipMemcpyAsync(h->ls_h.p, v, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (alpha * beta == 0 || normb == 0) {   // x = 0 is the exact solution (lsmr.py:292-296)
    *istop_out = 0;
    if (v != h->ls_v.p) std::swap(h->ls_v.p, h->ls_vraw.p);
    return 0;
  }
  // The iteration itself runs from a state block in HBM (mcba_lsmr.h): the host only ENQUEUES iterations, a bounded number
  // ahead of the progress word that k_lsmr_scal_a writes to pinned memory, until that word reports a stopping reason.  (The
  // first version fetched beta, alpha and |x| to the host in every iteration: three synchronisations of ~75 us.)  Kernels
  // enqueued behind the stop are empty launches; the call id in the word tells them from those of the next solve.
  if (h->ls_state.n < (size_t)2 * LS_NSLOTS) h->ls_state.alloc((size_t)2 * LS_NSLOTS, true);   // [A | B] (fused iteration)
  if (h->ls_xpart.n < (size_t)std::max(nvb, op.nblk) + 1) h->ls_xpart.alloc((size_t)std::max(nvb, op.nblk) + 1, true);
  if (h->ls_vpart.n < (size_t)op.gather3_grid() + 1) h->ls_vpart.alloc((size_t)op.gather3_grid() + 1, true);
  double* ls = h->ls_state.p;
  const unsigned long long call = (++h->ls_call) & 0xffffffull;
  hipLaunchKernelGGL(k_lsmr_init, dim3(1), dim3(64), 0, h->stream, ls, alpha, beta, damp, normb, (double)maxiter);
  check_launch("k_lsmr_init");
  constexpr long long LOOKAHEAD = 6;
  // Frame-sharded: every rank must enqueue the SAME number of iterations (each carries collectives).  The default (two-launch) form
  // enqueues in CHUNKS of LSMR_CHUNK iterations, two chunks ahead: chunk k + 1 goes out once the progress word of call k * CHUNK has
  // arrived without a stop (calls are numbered from 1; call j publishes the tests of step j - 1).  The word is monotone and the state
  // behind it is bit-identical on all ranks (computed from all-reduced sums by one workgroup in one order), so "was the solve stopped
  // when call k * CHUNK published?" has ONE answer whenever a rank happens to look: with the stop at step s every rank ends up having
  // enqueued (floor(s / CHUNK) + 2) * CHUNK calls (capped at maxiter + 1), and the collectives behind the stop stay matched -- their
  // kernels return on the flag.  The three- and six-launch forms (A/B runs) keep the lockstep of round 5 with three collectives each.
  constexpr long long LSMR_CHUNK = 8;
  const bool chunked = op.sharded() && h->lsmr_fused >= 2;
  const bool lockstep = op.sharded() && !chunked;
  long long enqueued = 0, done = 0;
  int istop = 0;
  double t_wait = 0.0;
  int spins = 0;
  unsigned long long seen = 0;
  while (true) {
    const unsigned long long w = __atomic_load_n(h->h_pub_seq + 1, __ATOMIC_ACQUIRE);
    if ((w >> 40) == call) {
      seen = w;
      const long long dn = (long long)(w & 0xffffffffull);
      if (dn != done) { done = dn; t_wait = 0.0; spins = 0; }
      istop = (int)((w >> 32) & 0xff);
      if (istop != 0 && !chunked) break;
    }
    if (chunked) {
      const bool have = (seen >> 40) == call;
      const long long allowed = lsmr_chunk_allowed(have, istop, done, LSMR_CHUNK, maxiter + 1);
      if (have && istop != 0 && enqueued >= allowed) break;
      if (enqueued < allowed) {
        op.iteration_fused2(ls, ls + LS_NSLOTS, u, v, vraw, call, enqueued == 0);
        std::swap(v, vraw);
        ++enqueued;
        if ((enqueued & 15) == 0) check_launch("lsmr iteration");
        t_wait = 0.0;
        spins = 0;
        continue;
      }
    }
    // Frame-sharded: every rank must enqueue the SAME number of iterations (each carries three collectives), so an iteration is
    // only enqueued once the word of the previous one (its k_lsmr_scal_a: completed = enqueued - 1, not stopped) has arrived --
    // the state is computed from all-reduced sums and is bit-identical on all ranks, hence so is the decision.
    const bool may_enqueue = lockstep ? (enqueued == 0 || ((seen >> 40) == call && done == enqueued - 1))
                                      : (enqueued - done < LOOKAHEAD);
    if (!chunked && may_enqueue && enqueued <= maxiter) {   // (iteration maxiter + 1 carries the tests of iteration maxiter)
      if (h->lsmr_fused >= 2) op.iteration_fused2(ls, ls + LS_NSLOTS, u, v, vraw, call, enqueued == 0);
      else if (h->lsmr_fused == 1) op.iteration_fused(ls, ls + LS_NSLOTS, u, v, vraw, call);
      else op.iteration(ls, u, v, vraw, call);
      std::swap(v, vraw);
      ++enqueued;
      if ((enqueued & 15) == 0) check_launch("lsmr iteration");
      t_wait = 0.0;
      spins = 0;
      continue;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if ((++spins & 1023) == 0) {
      const double now = now_seconds();
      if (t_wait == 0.0) t_wait = now;
      else if (now - t_wait > 0.05) {   // a failed launch must not hang the caller: everything enqueued has run after this
        HIP_OK(hipStreamSynchronize(h->stream));
        const unsigned long long w2 = __atomic_load_n(h->h_pub_seq + 1, __ATOMIC_ACQUIRE);
        REQUIRE((w2 >> 40) == call && w2 != seen, "the LSMR iteration on the device made no progress");
        t_wait = 0.0;
      }
    }
  }
  const int itn = (int)done;
  if (op.trace)
    fprintf(stderr, "[lsmr_solve rank %d/%d] call %llu: normb %.17g alpha %.17g beta %.17g damp %.17g maxiter %lld -> enqueued %lld done %lld istop %d\n",
            d.shard_rank, d.shard_world, call, normb, alpha, beta, damp, maxiter, enqueued, done, istop);
  if (v != h->ls_v.p) std::swap(h->ls_v.p, h->ls_vraw.p);   // (the handle's buffers keep their roles for the next call)
  *istop_out = istop;
  return itn;
}

}  // namespace

// g, diag, cost at dx (device) + every table the two products read (view chains, That); timing: events around k_linearize
static void lsmr_linearize(mcba_handle_s* h, const double* dx, bool timing = false) {
  const Dims& d = h->d;
  if (timing) HIP_OK(hipEventRecord(h->ev0, h->stream));
  launch_linearize(h, dx);
  if (timing) HIP_OK(hipEventRecord(h->ev1, h->stream));
  launch_assemble(h);
  const int nvw = d.views() * (d.motion == MOTION_ROLLING ? 2 : 1);
  if (nvw > 0) hipLaunchKernelGGL(k_views, dim3((nvw + 127) / 128), dim3(128), 0, h->stream, d, h->t);
  const int nb_views = std::max((d.views() + TMV - 1) / TMV, 1);
  if (tmat_local_poses(d) <= TM_LOCAL_POSES)
    hipLaunchKernelGGL(k_tmat<true>, dim3(nb_views), dim3(TM_THREADS), 0, h->stream, d, h->t, (double*)nullptr, 0, (double*)nullptr, 0,
                       (const double*)nullptr, nb_views);
  else
    hipLaunchKernelGGL(k_tmat<false>, dim3(nb_views), dim3(TM_THREADS), 0, h->stream, d, h->t, (double*)nullptr, 0, (double*)nullptr, 0,
                       (const double*)nullptr, nb_views);
  check_launch("linearisation");
}

/* scipy.optimize.least_squares(method='trf', tr_solver='lsmr', x_scale='jac') -- the reference's solver (calibration.py:209-210;
 * scipy picks 'lsmr' for a sparse Jacobian) -- with every product on the device: opt->tr_solver == MCBA_TR_LSMR.  The driver is
 * scipy's trf_no_bounds line by line (trf.py:401-560): Jacobian scaling, Cauchy regularisation, gn_h = lsmr(J_h, f, damp), the
 * 2-D subspace {g_h, gn_h} with B_S from the products J_h g_h and J_h gn_h, radius update, termination tests.  Unlike the exact
 * Schur / Cholesky steps of the default driver, the LSMR-truncated steps reproduce the reference's trajectory and END POINT.     */
static void solve_lsmr(mcba_handle h, double* x_inout, const mcba_options* opt, mcba_result* result) {
  const double t_start = now_seconds();
  const Dims& d = h->d;
  if (h->allreduce && d.shard_world <= 0)
    throw Error("frame-sharded handle without a rank: call mcba_set_shard_rank (or mcba_rccl_init)");
  const ScalLayout& sl = h->sl;
  const double ftol = opt->ftol, xtol = opt->xtol, gtol = opt->gtol;
  const int max_nfev = opt->max_nfev > 0 ? opt->max_nfev : d.n * 100;
  const double NaN = std::numeric_limits<double>::quiet_NaN();
  double* S = h->h_scal;
  // (persistent single-wave workgroups of the product kernels: h->lsmr_grid, default 2048; MCBA_LSMR_GRID: process-wide debug switch)
  if (const char* gsw = dbg_switch("MCBA_LSMR_GRID")) h->lsmr_grid = std::max(64, atoi(gsw));
  LsmrOps op = lsmr_setup(h);
  const size_t m = op.m;
  h->lsmr_trace.clear();

  // residuals of the WHOLE problem (scipy's maxiter = min(m, n) must be the same number on every rank of a frame-sharded solve,
  // whatever the rank's own shard holds -- an empty shard included): one 1-double all-reduce per solve
  size_t m_global = m;
  if (h->allreduce) {
    const double mine = (double)m;
    double all = 0.0;
    HIP_OK(hipMemcpyAsync(h->ls_out.p, &mine, sizeof(double), hipMemcpyHostToDevice, h->stream));
    op.fetch_sum(h->ls_out.p, 1, &all);
    m_global = (size_t)std::llround(all);
  }
  const bool trace = dbg_switch("MCBA_SOLVE_TRACE") != nullptr;
  op.trace = trace;
  upload_x(h, x_inout, h->x.p);
  float lin_ms_total = 0.f;
  bool lin_timed = false;
  auto linearize = [&](const double* dx) {   // g, diag, cost at dx + every table the two products read (view chains, That)
    const bool timing = !lin_timed;          // (only the first linearisation is timed: see mcba_solve)
    lin_timed = true;
    lsmr_linearize(h, dx, timing);
  };
  const int prep_blocks = (d.n_pose + d.C + d.B * d.P + 255) / 256;
  const int cost_grid = h->cost_blocks;
  linearize(h->x.p);
  int nfev = 1, njev = 1, iteration = 0, status = -100;
  bool first = true;
  double cost = 0, Delta = 0, step_norm = NaN, actual_reduction = NaN, g_norm = 0, initial_cost = 0;
  long long lsmr_iterations = 0;
  while (true) {
    hipLaunchKernelGGL(k_vec_scale, dim3(sl.nvb), dim3(256), 0, h->stream, d, h->x.p, h->g(), h->diag(), h->scale_inv.p, h->dsc.p,
                       h->gh.p, first ? 1 : 0, h->scal.p + sl.vs, h->costcount(), h->scal.p + TR_COST);
    if (h->allreduce) {   // the norms of the iterate, per rank ("gathered by summation", like message 2 of the exact solver)
      hipLaunchKernelGGL(k_shard_fold2, dim3(1), dim3(64), 0, h->stream, h->scal.p + sl.vs, sl.nvb, (const double*)nullptr, 0,
                         d.shard_rank, d.shard_world, h->scal.p + sl.shard2);
      call_allreduce(h, h->scal.p + sl.shard2, (size_t)4 * d.shard_world, 0);
      HIP_OK(hipMemcpyAsync(h->h_scal + sl.shard2, h->scal.p + sl.shard2, 4 * d.shard_world * sizeof(double),
                            hipMemcpyDeviceToHost, h->stream));
    }
    fetch_scalars(h, sl.q00p);
    if (lin_timed && lin_ms_total == 0.f) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) lin_ms_total = ms;
    }
    double mx = 0, gg = 0, xs = 0;
    const int nfold = h->allreduce ? d.shard_world : sl.nvb, at = h->allreduce ? sl.shard2 : sl.vs;
    for (int blk = 0; blk < nfold; ++blk) {
      mx = std::max(mx, S[at + 3 * blk]);
      gg += S[at + 3 * blk + 1];
      xs += S[at + 3 * blk + 2];
    }
    g_norm = mx;
    if (first) {
      cost = S[TR_COST];
      initial_cost = cost;
      if (!std::isfinite(cost)) throw Error("Residuals are not finite in the initial point.");   // scipy least_squares.py:844-845
      // (frame-sharded: the observation count of the linearisation is all-reduced with the cost -- the SAME maxiter on every rank,
      //  whatever its own shard holds; an empty shard must not stop enqueuing iterations and their collectives early)
      op.m_global = m_global;
      Delta = std::sqrt(xs);
      if (Delta == 0) Delta = 1.0;
      first = false;
    }
    if (g_norm < gtol) status = 1;
    if (h->log && opt->verbose >= 2) h->log(h->log_ctx, iteration, nfev, cost, actual_reduction, step_norm, g_norm);
    if (status != -100 || nfev >= max_nfev) break;

    // ---- trf.py:474-489 ------------------------------------------------------------------------------------------
    const double Q00 = op.jv(2, h->gh.p, 0.0, h->ls_ua.p);                       // |J_h g_h|^2 (build_quadratic_1d)
    const double reg_term = gg > 0 ? tr_reg_term(Q00, gg, Delta, 0.0) : 0.0;
    int istop = 0;
    const int itn = lsmr_solve(op, std::sqrt(reg_term), &istop);
    lsmr_iterations += itn;
    h->lsmr_trace.push_back({(double)iteration, std::sqrt(reg_term), Delta, (double)istop, (double)itn, NaN, NaN, NaN, NaN, NaN});
    if (h->lsmr_trace_scalars) lsmr_fetch_scalars(h, h->lsmr_trace.back());
    if (trace)
      fprintf(stderr, "[mcba_solve lsmr rank %d/%d] iteration %d: cost %.17g Q00 %.17g gg %.17g Delta %.17g reg_term %.17g m %zu -> lsmr itn %d istop %d\n",
              d.shard_rank, d.shard_world, iteration, cost, Q00, gg, Delta, reg_term, op.m_global, itn, istop);
    HIP_OK(hipMemcpyAsync(h->gn.p, h->ls_x.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    op.jv(2, h->gn.p, 0.0, h->ls_ub.p);
    // [Jg.Jgn | Jg.Jg | Jgn.Jgn] over the m rows and [g_h.gn | g_h.g_h | gn.gn] over the n entries (reordered below); frame-sharded:
    // per-rank partials (every parameter entry counted once: k_dot_weighted), summed by ONE all-reduce of 6 doubles
    if (h->dot_part.n < (size_t)3 * DOT_WAVES) h->dot_part.alloc((size_t)3 * DOT_WAVES, true);
    hipLaunchKernelGGL(k_dot3_part, dim3(DOT_WAVES), dim3(64), 0, h->stream, m, (const double*)h->ls_ua.p, (const double*)h->ls_ub.p, h->dot_part.p);
    hipLaunchKernelGGL(k_dot3_fin, dim3(1), dim3(64), 0, h->stream, (const double*)h->dot_part.p, h->ls_out.p);
    if (h->allreduce)
      hipLaunchKernelGGL(k_dot_weighted, dim3(1), dim3(1024), 0, h->stream, d, (const double*)h->gh.p, (const double*)h->gn.p,
                         h->ls_out.p + 3, 1);
    else
      hipLaunchKernelGGL(k_dot, dim3(1), dim3(1024), 0, h->stream, (size_t)d.n, (const double*)h->gh.p, (const double*)h->gn.p,
                         h->ls_out.p + 3, 1);
    double q6[6];
    op.fetch_sum(h->ls_out.p, 6, q6);
    const double* q3 = q6;
    const double* d3 = q6 + 3;
    S[TR_REG] = reg_term;
    S[TR_Q00] = q3[1];
    S[TR_D00] = d3[1]; S[TR_D01] = d3[0]; S[TR_D11] = d3[2];
    S[TR_GNORM] = g_norm; S[TR_GH2] = gg; S[TR_XS2] = xs;
    tr_subspace(S, true, q3[0], q3[2]);

    actual_reduction = -1;
    double cost_new = cost, ratio = 0;
    while (actual_reduction <= 0 && nfev < max_nfev) {
      tr_trial(S, Delta);
      hipLaunchKernelGGL(k_vec_step, dim3(sl.nvb + prep_blocks), dim3(256), 0, h->stream, d, h->t, h->x.p, h->dsc.p, h->gh.p, h->gn.p,
                         S[TR_ALPHA], S[TR_BETA], h->xnew.p, h->scal.p + sl.step, (double*)nullptr, h->scal.p + sl.dotp, 0, sl.nvb,
                         (double*)nullptr, (double*)nullptr);
      h->ops->cost(d, h->t, h->stream, h->scal.p + sl.costp, cost_grid);   // (an empty shard writes partial[0] = 0)
      if (h->allreduce) {   // [trial cost | step norms] of this rank, 4 doubles, summed over the ranks
        if (h->comm.n < 4) h->comm.alloc(4, false);
        hipLaunchKernelGGL(k_shard_trial_pack, dim3(1), dim3(64), 0, h->stream, h->scal.p + sl.costp, cost_grid, h->scal.p + sl.step,
                           sl.nvb, h->comm.p);
        call_allreduce(h, h->comm.p, 4, 0);
        hipLaunchKernelGGL(k_shard_trial_unpack, dim3(1), dim3(256), 0, h->stream, h->comm.p, h->scal.p + sl.costp,
                           h->scal.p + sl.step, sl.nvb);
      }
      const int cost_values = h->allreduce ? 1 : cost_grid;
      fetch_scalars(h, sl.costp + cost_values - sl.step, sl.step);
      double s3[3] = {0, 0, 0};
      for (int blk = 0; blk < sl.nvb; ++blk)
        for (int k = 0; k < 3; ++k) s3[k] += S[sl.step + 3 * blk + k];
      cost_new = host_sum(S + sl.costp, cost_values);
      const double predicted = S[TR_PRED];
      ++nfev;
      const double step_h_norm = std::sqrt(s3[0]);
      if (!std::isfinite(cost_new)) {
        Delta = 0.25 * step_h_norm;
        continue;
      }
      actual_reduction = cost - cost_new;
      double Delta_new = Delta;
      tr_update_radius(Delta_new, actual_reduction, predicted, step_h_norm, step_h_norm > 0.95 * Delta, ratio);
      step_norm = std::sqrt(s3[1]);
      status = tr_check_termination(actual_reduction, cost, step_norm, std::sqrt(s3[2]), ratio, ftol, xtol);
      if (status != -100) break;
      Delta = Delta_new;
    }
    if (actual_reduction > 0) {
      std::swap(h->x.p, h->xnew.p);
      cost = cost_new;
      linearize(h->x.p);
      ++njev;
    } else {
      step_norm = 0;
      actual_reduction = 0;
    }
    ++iteration;
  }
  if (status == -100) status = 0;
  if (trace)
    fprintf(stderr, "[mcba_solve lsmr rank %d/%d] %d trial steps, %lld LSMR iterations, status %d, %.3f ms\n", d.shard_rank, d.shard_world,
            nfev - 1, lsmr_iterations, status, (now_seconds() - t_start) * 1e3);
  gather_frame_entries(h, h->x.p);   // (frame-sharded: every rank returns the complete x; ONE n_motion message per solve)
  HIP_OK(hipMemcpyAsync(h->h_x, h->x.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  to_caller(h, x_inout, h->h_x);
  h->lsmr_iterations_last = lsmr_iterations;
  if (result) {
    result->cost = cost;
    result->initial_cost = initial_cost;
    result->optimality = g_norm;
    result->nfev = nfev;
    result->njev = njev;
    result->status = status;
    result->iterations = iteration;
    result->solve_seconds = now_seconds() - t_start;
    result->linearize_seconds = lin_ms_total * 1e-3;   // (the first linearisation, as mcba_solve reports it)
  }
}

/* test hook (mcba_debug.h): the two matrix-free products of the lsmr mode at x WITHOUT column scaling, through the very kernels
 * the solver iterates with (k_lsmr_jv / k_lsmr_jtu / k_lsmr_gather): jv_out[m] = J(x) v in the reference's residual order,
 * jtu_out[n] = J(x)^T u.  Either pair may be NULL.  Linear loss (J of `evaluate` itself, as mcba_jacobian returns it).      */
int32_t mcba_debug_lsmr_products(mcba_handle h, const double* x, const double* v, const double* u, double* jv_out, double* jtu_out) {
  API_BEGIN
  REQUIRE(h && x, "null argument");
  REQUIRE((v == nullptr) == (jv_out == nullptr) && (u == nullptr) == (jtu_out == nullptr), "v / jv_out and u / jtu_out come in pairs");
  g_fill_stream = h->stream;
  set_loss(h, nullptr);
  const Dims& d = h->d;
  LsmrOps op = lsmr_setup(h);
  const size_t m = op.m;
  upload_x(h, x, h->x.p);
  sync(h);   // (h_x is reused for v below)
  lsmr_linearize(h, h->x.p);
  std::vector<double> ones((size_t)d.n, 1.0);
  HIP_OK(hipMemcpyAsync(h->dsc.p, ones.data(), (size_t)d.n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  sync(h);
  if (v != nullptr) {
    upload_x(h, v, h->ls_v.p);
    op.jv(2, h->ls_v.p, 0.0, h->ls_u.p);
    HIP_OK(hipMemcpyAsync(jv_out, h->ls_u.p, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    sync(h);
  }
  if (u != nullptr) {
    HIP_OK(hipMemcpyAsync(h->ls_ua.p, u, m * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipMemsetAsync(h->ls_v.p, 0, (size_t)d.n * sizeof(double), h->stream));
    op.jtu(h->ls_ua.p, 1.0, 0.0, h->ls_v.p, h->ls_vraw.p);
    HIP_OK(hipMemcpyAsync(h->h_x, h->ls_vraw.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    sync(h);
    to_caller(h, jtu_out, h->h_x);
  }
  API_END
}

/* experiment / path-forcing switch `name` = `value` (see dbg_switch); value == NULL removes nothing -- set "" to override an
 * environment value of a variant build.  Process-wide; call before the first mcba_create.                                  */
int32_t mcba_debug_set_switch(const char* name, const char* value) {
  API_BEGIN
  REQUIRE(name && value && strncmp(name, "MCBA_", 5) == 0, "bad switch");
  std::lock_guard<std::mutex> lock(g_dbg_switch_mutex);
  auto& t = dbg_switch_table();
  if (t.find(name) == t.end()) t[name] = value;
  else REQUIRE(t[name] == value, "a switch can be set once per process (most are latched on first use)");
  API_END
}

/* -1 (default): automatic (3 for static / hand-eye rigs, 2 for rolling shutter and boards=True); 3: two launches, the per-observation
 * state of the linearisation cached by the first iteration of a solve and streamed back by the others; 2: the two-launch LSMR iteration
 * (k_lsmr_fused2 / k_lsmr_gather3); 1: three launches (k_lsmr_fused / k_lsmr_gather2 / k_lsmr_update2); 0: the six-launch form of
 * round 4 (A/B runs, test_lsmr_iteration_forms_agree) */
int32_t mcba_debug_set_lsmr_fused(mcba_handle h, int32_t on) {
  API_BEGIN
  REQUIRE(h, "null handle");
  REQUIRE(on >= -1 && on <= 3, "-1 = automatic, 0 = six launches, 1 = three, 2 = two, 3 = two with the per-observation state cached");
  h->lsmr_fused_setting = on;
  h->lsmr_fused = on >= 0 ? on : 2;
  API_END
}

/* test hook (mcba_debug.h): ONE call of the device's LSMR solve -- scipy.sparse.linalg.lsmr(J_h, f, damp, atol = btol = 1e-6), the
 * call of scipy/optimize/_lsq/trf.py:481 -- on the linearisation at x with scipy's Jacobian scaling of a FIRST iterate
 * (compute_jac_scale without a previous scale), through lsmr_solve itself (whichever iteration form the handle is set to).
 * scale_in (may be NULL) replaces that scaling (a later iterate: scipy keeps the running maximum of the column norms); maxiter > 0 =
 * scipy's `maxiter` argument (0: min(m, n)): the first few dozen Golub-Kahan steps can be compared with scipy's to rounding -- beyond
 * that the bidiagonalisation of these Jacobians loses orthogonality and any two roundings of it drift apart (profiles/r06_lsmr_sign.md).
 * gn_h_out[n] = the solution, scale_out[n] = d (J_h = J diag(d)), out[8] = scipy's return tuple {istop, itn, normr, normar, normA,
 * condA, normx} + normb.                                                                                                       */
int32_t mcba_debug_lsmr_solve(mcba_handle h, const double* x, const mcba_options* opt, double damp, const double* scale_in, int32_t maxiter,
                              double* gn_h_out, double* scale_out, double* out) {
  API_BEGIN
  REQUIRE(h && x && gn_h_out && out, "null argument");
  REQUIRE(h->allreduce == nullptr, "single handles only");
  REQUIRE(damp >= 0, "damp must not be negative");
  g_fill_stream = h->stream;
  set_loss(h, opt);
  const Dims& d = h->d;
  const ScalLayout& sl = h->sl;
  LsmrOps op = lsmr_setup(h);
  upload_x(h, x, h->x.p);
  sync(h);   // (h_x is reused below)
  lsmr_linearize(h, h->x.p);
  hipLaunchKernelGGL(k_vec_scale, dim3(sl.nvb), dim3(256), 0, h->stream, d, h->x.p, h->g(), h->diag(), h->scale_inv.p, h->dsc.p, h->gh.p, 1,
                     h->scal.p + sl.vs, h->costcount(), h->scal.p + TR_COST);
  if (scale_in != nullptr) {   // the scaling of a LATER iterate (scipy keeps the running maximum of the column norms: common.py:606-611)
    sync(h);
    upload_x(h, scale_in, h->dsc.p);
    sync(h);
  }
  int istop = 0;
  op.maxiter_override = maxiter;
  const int itn = lsmr_solve(op, damp, &istop);
  mcba_handle_s::LsmrCall c{0.0, damp, 0.0, (double)istop, (double)itn, 0, 0, 0, 0, 0};
  lsmr_fetch_scalars(h, c);
  double normb = 0.0;
  HIP_OK(hipMemcpyAsync(&normb, lsmr_final_state(h) + LS_NORMB, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(h->h_x, h->ls_x.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  to_caller(h, gn_h_out, h->h_x);
  if (scale_out != nullptr) {
    HIP_OK(hipMemcpyAsync(h->h_x, h->dsc.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    sync(h);
    to_caller(h, scale_out, h->h_x);
  }
  out[0] = c.istop; out[1] = c.itn; out[2] = c.normr; out[3] = c.normar; out[4] = c.normA; out[5] = c.condA; out[6] = c.normx;
  out[7] = itn > 0 ? normb : std::numeric_limits<double>::quiet_NaN();
  API_END
}

/* one row per LSMR call of the last mcba_solve with tr_solver = MCBA_TR_LSMR on this handle, 10 doubles each: {trust-region iteration,
 * damp, Delta, istop, itn, normr, normar, normA, condA, normx}; the last five are NaN unless mcba_debug_set_lsmr_trace(h, 1) asked
 * for them BEFORE the solve (one small copy + wait per call).  rows may be NULL (count only); at most cap rows are written.        */
int32_t mcba_debug_lsmr_trace(mcba_handle h, int32_t cap, double* rows, int32_t* n_rows) {
  API_BEGIN
  REQUIRE(h && n_rows, "null argument");
  *n_rows = (int32_t)h->lsmr_trace.size();
  if (rows != nullptr)
    for (int i = 0; i < std::min<int>(cap, *n_rows); ++i) {
      const auto& c = h->lsmr_trace[(size_t)i];
      const double r[10] = {c.tr_iteration, c.damp, c.Delta, c.istop, c.itn, c.normr, c.normar, c.normA, c.condA, c.normx};
      memcpy(rows + 10 * (size_t)i, r, sizeof(r));
    }
  API_END
}

int32_t mcba_debug_set_lsmr_trace(mcba_handle h, int32_t scalars) {
  API_BEGIN
  REQUIRE(h, "null handle");
  h->lsmr_trace_scalars = scalars != 0;
  API_END
}

/* 1: k_lsmr_fused2 reads masks / observations / board points from the frame-major tables on every rig (its form with boards=True and the
 * only one until round 6) instead of the compacted tables; A/B runs and test_lsmr_observation_sources_agree */
int32_t mcba_debug_set_lsmr_masks_form(mcba_handle h, int32_t on) {
  API_BEGIN
  REQUIRE(h, "null handle");
  h->lsmr_masks_form = on != 0;
  API_END
}

/* number of collective sizes mcba_allreduce_stats keeps (default 4096): the collective-sequence tests record whole lsmr solves */
int32_t mcba_debug_set_allreduce_trace(mcba_handle h, int32_t cap) {
  API_BEGIN
  REQUIRE(h && cap >= 0 && cap <= (1 << 22), "bad cap");
  h->ar_trace_cap = (size_t)cap;
  API_END
}

/* persistent single-wave workgroups of the LSMR product kernels on this handle (default 2048; >= 64): changes only the order in
 * which partial sums are folded -- the summation-order experiment of profiles/r06_lsmr_sign.* */
int32_t mcba_debug_set_lsmr_grid(mcba_handle h, int32_t grid) {
  API_BEGIN
  REQUIRE(h && grid >= 64 && grid <= 65536, "bad grid");
  h->lsmr_grid = grid;
  API_END
}

/* LSMR iterations of the last solve with tr_solver = MCBA_TR_LSMR on this handle */
int32_t mcba_debug_lsmr_info(mcba_handle h, int64_t* lsmr_iterations) {
  API_BEGIN
  REQUIRE(h && lsmr_iterations, "null argument");
  *lsmr_iterations = h->lsmr_iterations_last;
  API_END
}

int32_t mcba_solve(mcba_handle h, double* x_inout, const mcba_options* opt, mcba_result* result) {
  API_BEGIN
  REQUIRE(h && x_inout && opt, "null argument");
  if (opt->tr_solver == MCBA_TR_LSMR) {
    set_loss(h, opt);
    solve_lsmr(h, x_inout, opt, result);
    return 0;
  }
  REQUIRE(opt->tr_solver == MCBA_TR_EXACT, "unknown trust-region solver");
  const double t_start = now_seconds();
  set_loss(h, opt);
  const Dims& d = h->d;
  const ScalLayout& sl = h->sl;
  const double ftol = opt->ftol, xtol = opt->xtol, gtol = opt->gtol;
  const int max_nfev = opt->max_nfev > 0 ? opt->max_nfev : d.n * 100;
  const double NaN = std::numeric_limits<double>::quiet_NaN();
  const bool is_root = h->shard_root;
  // The scalar trust-region algebra runs in one-wave kernels between the vector kernels (k_tr_reg; the subspace step is the head of k_vec_step), so that a
  // whole iteration -- scaling, Cauchy curvature, damped Gauss-Newton solve, 2-D subspace step, trial cost -- is enqueued
  // at once and the host synchronises ONCE per iteration.  Frame-sharded handles insert their all-reduces into the same
  // chain (stream-ordered: native RCCL or the torch.distributed hook): H-dependent per-block partials are reduced element-
  // wise across ranks before the device (k_q00 partials) or the host (k_cost partials) folds them.  The host runs the
  // same algebra (mcba_trmath.h) only for the retries after a rejected step.
  double* S = h->h_scal;   // host copy of the scalar block scal[0 .. TR_NSLOTS)

  static const bool trace = dbg_switch("MCBA_SOLVE_TRACE") != nullptr;   // host wall clock of the driver's stages (stderr)
  auto mark = [&](const char* what) {
    if (trace) fprintf(stderr, "[mcba_solve] %8.3f ms  %s\n", (now_seconds() - t_start) * 1e3, what);
  };
  mark("enter");
  upload_x(h, x_inout, h->x.p);
  mark("x uploaded");
  float lin_ms_total = 0.f;
  // dx: linearise at that parameter vector (tables prepared by k_tmat itself); nullptr: the pose tables already hold the
  // point (the trial step's k_prep)
  // Only the FIRST linearisation of a solve is timed (result->linearize_seconds): an event record is a barrier packet in the
  // queue, and the two around every later linearisation cost 5 us between k_vec_step and k_linearize and 4.5 us between
  // k_linearize and k_assemble in every LM iteration (gaps in the rocprofv3 kernel trace of a long solve).
  bool lin_timed = false;
  auto timed_linearize = [&](const double* dx, unsigned long long publish_seq = 0, int cost_slot = 0, bool step_valid = false) {
    const bool timing = !lin_timed;
    lin_timed = true;
    if (timing) HIP_OK(hipEventRecord(h->ev0, h->stream));
    launch_linearize(h, dx);
    if (timing) HIP_OK(hipEventRecord(h->ev1, h->stream));
    launch_assemble(h, publish_seq, cost_slot, step_valid);
  };
  auto collect_lin_time = [&]() {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) lin_ms_total += ms;
  };
  // k_cost partials: a single GPU copies them to the host with the other scalars (no extra launch); a sharded handle folds
  // them on the device first, so that the trial cost crosses the ranks as ONE double
  const int cost_grid = h->cost_blocks;
  const int cost_fetch = h->allreduce ? 1 : cost_grid;
  // trial step for coefficients given by the host (retries) or computed on the device (tr_dev)
  // (tr_dev: k_vec_step first finishes the device-side algebra -- fold of the back-substitution's dots, 2-D subspace step;
  //  its trailing workgroups write the pose / camera / board-point tables of x_new for k_cost, which forms the view chains
  //  itself; k_tmat rebuilds the view table if the step is accepted)
  const int prep_blocks = (d.n_pose + d.C + d.B * d.P + 255) / 256;
  // (a frame-sharded handle folds the all-reduced per-RANK dots, one block per rank: sl.shard4)
  const int dot_blocks = h->allreduce ? d.shard_world : gn_dot_blocks(d);
  const int dot_slot = h->allreduce ? sl.shard4 : sl.dotp;
  if (h->allreduce && d.shard_world <= 0)
    throw Error("frame-sharded handle without a rank: call mcba_set_shard_rank (or mcba_rccl_init)");
  // with_cost = false (sharded handles, first trial of an iteration): no k_cost pass and no 1-double all-reduce -- the
  // speculative linearisation at x_new that follows delivers the cost of the very same point inside its own
  // [g | diag | cost] message (one dependent collective less per accepted iteration)
  auto enqueue_trial = [&](double alpha, double beta, double* tr_dev, bool with_cost = true, bool to_host = false) {
    hipLaunchKernelGGL(k_vec_step, dim3(sl.nvb + prep_blocks), dim3(256), 0, h->stream, d, h->t, h->x.p, h->dsc.p, h->gh.p,
                       h->gn.p, alpha, beta, h->xnew.p, h->scal.p + sl.step, tr_dev, h->scal.p + dot_slot, dot_blocks, sl.nvb,
                       to_host ? h->h_scal : nullptr, to_host ? h->h_scal + sl.step : nullptr);
    if (!with_cost) return;
    h->ops->cost(d, h->t, h->stream, h->scal.p + sl.costp, cost_grid);   // (an empty shard writes partial[0] = 0)
    if (h->allreduce) {   // retry: [trial cost | step norms] of this rank, 4 doubles
      hipLaunchKernelGGL(k_shard_trial_pack, dim3(1), dim3(64), 0, h->stream, h->scal.p + sl.costp, cost_grid, h->scal.p + sl.step,
                         sl.nvb, h->comm.p);
      call_allreduce(h, h->comm.p, 4, 0);
      hipLaunchKernelGGL(k_shard_trial_unpack, dim3(1), dim3(256), 0, h->stream, h->comm.p, h->scal.p + sl.costp,
                         h->scal.p + sl.step, sl.nvb);
    }
  };
  static const bool merge_off = dbg_switch("MCBA_NO_MERGED_TRIAL_COST") != nullptr && dbg_switch("MCBA_NO_MERGED_TRIAL_COST")[0] == '1';
  // (MCBA_FORCE_MERGED_TRIAL_COST=1: experiment -- a single GPU takes the trial cost from the speculative linearisation as well)
  static const bool merge_force = dbg_switch("MCBA_FORCE_MERGED_TRIAL_COST") != nullptr && dbg_switch("MCBA_FORCE_MERGED_TRIAL_COST")[0] == '1';
  const bool merged_trial_cost = (h->allreduce != nullptr || merge_force) && !merge_off;
  auto fold_trial = [&](double* step_h2, double* step2, double* x2, int n_cost) {   // after a fetch that covers [sl.step, ...)
    double s3[3] = {0, 0, 0};
    for (int blk = 0; blk < sl.nvb; ++blk)
      for (int k = 0; k < 3; ++k) s3[k] += S[sl.step + 3 * blk + k];
    *step_h2 = s3[0]; *step2 = s3[1]; *x2 = s3[2];
    return host_sum(S + sl.costp, n_cost);
  };
  const int trial_fetch_end = sl.costp + cost_fetch;

  timed_linearize(h->x.p);
  int nfev = 1, njev = 1, iteration = 0, status = -100;
  bool first = true, fresh_lin = true, lin_stale = false;
  // Speculative acceptance (single GPU, table-fed fused linearisation): NO trial-cost kernel on the path of an accepted step.
  // k_vec_step publishes its scalars to pinned host memory, the speculative linearisation at x_new delivers the cost of x_new
  // (k_shared_final writes it + a sequence number to the host), and the gradient scaling + Cauchy curvature of x_new are
  // enqueued behind it into the second set of scaling buffers BEFORE the host has decided: they run while the host folds,
  // updates the radius and enqueues the solve of the next iteration.  An accepted step swaps the buffer sets; a rejected one
  // leaves them behind and goes through the retry path (k_cost, stream-ordered fetch) as before.  Against the side-stream
  // form: k_linearize is not slowed by a concurrent k_cost (58 -> 49 us), no event packet between k_vec_step and
  // k_linearize (7 us), no copy.  MCBA_SPEC_ACCEPT=0 restores the side-stream form.
  // (blocks of the curvature sums: every k_schur_frame workgroup folds their partials -- MCBA_Q00_BLOCKS for experiments)
  static const int q00_blocks = dbg_switch("MCBA_Q00_BLOCKS") ? std::max(1, std::min(Q00_BLOCKS, atoi(dbg_switch("MCBA_Q00_BLOCKS")))) : Q00_BLOCKS;
  static const bool spec_accept_off = dbg_switch("MCBA_SPEC_ACCEPT") != nullptr && dbg_switch("MCBA_SPEC_ACCEPT")[0] == '0';
  // (k_fold_tr behind the speculative scaling: measured neutral against letting every k_schur_frame workgroup fold the
  //  72 + 512 partials itself -- 135.3 vs 136.1 us per trial step at the north-star rig, round 4 -- kept)
  bool scaled_ahead = false;     // the scaling / curvature of h->x are already in place (computed speculatively, swapped in)
  int trial_cost_values = cost_fetch;
  double cost = 0, Delta = 0, step_norm = NaN, actual_reduction = NaN, g_norm = 0, initial_cost = 0;

  while (true) {
    // a terminated / exhausted solve only needs the gradient norm of the final iterate (scipy reports it as optimality)
    const bool finishing = status != -100 || nfev >= max_nfev;
    if (lin_stale) {   // the last speculative linearisation was for a rejected point: rebuild g, H at x
      timed_linearize(h->x.p);
      lin_stale = false;
    }
    bool spec_lin = false, spec_scaled = false;
    // ---- enqueue: gradient scaling, Cauchy curvature (+ on a single GPU the whole step and its trial evaluation) ----
    // k_vec_scale also forwards {cost, count} of the linearisation into scal[TR_COST, TR_COUNT]
    static const bool split_q00 = dbg_switch("MCBA_SPLIT_Q00") != nullptr && dbg_switch("MCBA_SPLIT_Q00")[0] == '1';
    const bool scaled = scaled_ahead;   // (k_vec_scale's partials of this very point are in scal[sl.vs ..) already)
    scaled_ahead = false;
    if (scaled) {
    } else if (finishing || split_q00)
      hipLaunchKernelGGL(k_vec_scale, dim3(sl.nvb), dim3(256), 0, h->stream, d, h->x.p, h->g(), h->diag(), h->scale_inv.p,
                         h->dsc.p, h->gh.p, first ? 1 : 0, h->scal.p + sl.vs, h->costcount(), h->scal.p + TR_COST);
    bool have_trial = false;
    if (finishing) {
      if (h->allreduce) {   // the norms of the final iterate, per rank (message 2 without the curvature)
        hipLaunchKernelGGL(k_shard_fold2, dim3(1), dim3(64), 0, h->stream, h->scal.p + sl.vs, sl.nvb, (const double*)nullptr, 0,
                           d.shard_rank, d.shard_world, h->scal.p + sl.shard2);
        call_allreduce(h, h->scal.p + sl.shard2, (size_t)4 * d.shard_world, 0);
        HIP_OK(hipMemcpyAsync(h->h_scal + sl.shard2, h->scal.p + sl.shard2, 4 * d.shard_world * sizeof(double),
                              hipMemcpyDeviceToHost, h->stream));
      }
      fetch_scalars(h, sl.q00p);
    } else {
      if (scaled) {
      } else if (split_q00)
        hipLaunchKernelGGL(k_q00, dim3(q00_blocks), dim3(256), 0, h->stream, d, h->Hss.p, h->Hfs.p, h->Hff.p, h->dsc.p,
                           h->gh.p, h->scal.p + sl.q00p);
      else   // gradient scaling and Cauchy curvature in one launch (the curvature forms its scaled gradient on the fly)
        hipLaunchKernelGGL(k_vec_scale_q00, dim3(sl.nvb + q00_blocks), dim3(256), 0, h->stream, d, h->x.p, h->g(), h->diag(),
                           h->scale_inv.p, h->scale_inv.p, h->dsc.p, h->gh.p, first ? 1 : 0, h->scal.p + sl.vs, h->costcount(),
                           h->scal.p + TR_COST, sl.nvb, h->Hss.p, h->Hfs.p, h->Hff.p, h->scal.p + sl.q00p);
      if (h->allreduce && !scaled) {   // message 2: [|g|_inf, |g_h|^2, |x scale|^2, curvature] of every rank, 4 doubles each
        hipLaunchKernelGGL(k_shard_fold2, dim3(1), dim3(64), 0, h->stream, h->scal.p + sl.vs, sl.nvb, h->scal.p + sl.q00p,
                           q00_blocks, d.shard_rank, d.shard_world, h->scal.p + sl.shard2);
        call_allreduce(h, h->scal.p + sl.shard2, (size_t)4 * d.shard_world, 0);
      }          // (scaled: message 2 of this point went out speculatively, behind its linearisation)
      // (the fold of the k_vec_scale / k_q00 partials and the damping: head of the first kernel of the solve)
      // (scaled ahead: the partials were folded behind it by k_fold_tr -- [mx | gg | xs] is a k_vec_scale block of its own, q one value)
      const TrRegPartials trp = scaled && !h->allreduce ? TrRegPartials{h->scal.p + sl.fold, 1, h->scal.p + sl.fold + 3, 1, 0, Delta}
                              : h->allreduce ? TrRegPartials{h->scal.p + sl.shard2, d.shard_world, h->scal.p + sl.shard2 + 3 * d.shard_world,
                                                             d.shard_world, first ? 1 : 0, Delta}
                                       : TrRegPartials{h->scal.p + sl.vs, sl.nvb, h->scal.p + sl.q00p, q00_blocks, first ? 1 : 0, Delta};
      launch_gn_solve(h, 0.0, is_root, h->scal.p + sl.dotp, h->scal.p, &trp);
      // Single GPU, table-fed fused linearisation: the trial cost and the speculative linearisation both only READ the tables
      // that the tail of k_vec_step wrote, so k_cost + the scalar copy go to a side stream and run BESIDE k_linearize (13 us
      // of every iteration's critical path at the north-star rig); the host still decides on the trial cost alone.
      const bool spec_tables_ready = !linearize_table_form() && d.off_boards < 0 && h->use_mfma;
      static const bool side_off = dbg_switch("MCBA_NO_SIDE_COST") != nullptr && dbg_switch("MCBA_NO_SIDE_COST")[0] == '1';
      const bool side_cost = !h->allreduce && spec_tables_ready && !side_off && !merged_trial_cost;
      static const bool publish = !(dbg_switch("MCBA_NO_PUBLISH") != nullptr && dbg_switch("MCBA_NO_PUBLISH")[0] == '1');
      const bool spec_accept = side_cost && !spec_accept_off;
      if (spec_accept) {   // (see the declaration of scaled_ahead)
        enqueue_trial(0.0, 0.0, h->scal.p, false, true);
        ++h->pub_seq;
        timed_linearize(nullptr, h->pub_seq, sl.costp);
        hipLaunchKernelGGL(k_vec_scale_q00, dim3(sl.nvb + q00_blocks), dim3(256), 0, h->stream, d, h->xnew.p, h->g(), h->diag(),
                           h->scale_inv.p, h->scale_inv2.p, h->dsc2.p, h->gh2.p, 0, h->scal.p + sl.vs, h->costcount(),
                           h->scal.p + TR_COST, sl.nvb, h->Hss.p, h->Hfs.p, h->Hff.p, h->scal.p + sl.q00p);
        hipLaunchKernelGGL(k_fold_tr, dim3(1), dim3(64), 0, h->stream, h->scal.p + sl.vs, sl.nvb, h->scal.p + sl.q00p, q00_blocks,
                           h->scal.p + sl.fold);
        spec_lin = true;
        spec_scaled = true;
        trial_cost_values = 1;
        mark("iteration enqueued");
        publish_scalars_end(h, h->stream);
        mark("trial cost fetched");
        have_trial = true;
      } else {
      enqueue_trial(0.0, 0.0, h->scal.p, !merged_trial_cost && !side_cost);
      if (side_cost) {
        HIP_OK(hipEventRecord(h->ev_side, h->stream));
        HIP_OK(hipStreamWaitEvent(h->stream2, h->ev_side, 0));
        h->ops->cost(d, h->t, h->stream2, h->scal.p + sl.costp, cost_grid);
        if (publish) {
          publish_scalars_begin(h, trial_fetch_end, h->stream2);
        } else {
          HIP_OK(hipMemcpyAsync(h->h_scal, h->scal.p, trial_fetch_end * sizeof(double), hipMemcpyDeviceToHost, h->stream2));
          HIP_OK(hipEventRecord(h->ev_fetch, h->stream2));
        }
      } else if (!merged_trial_cost) {
        fetch_scalars_begin(h, trial_fetch_end);
      }
      // Speculation: most trial steps are accepted, so the linearisation at x_new is enqueued right behind the copy and
      // runs while the host looks at the trial cost and prepares the next iteration.  A rejected step leaves the
      // records / H / g of x_new behind (lin_stale): they are not needed by the retries with a smaller radius, and
      // are rebuilt before anything reads them again.
      // (fused form: straight from x_new; table form: k_tmat re-derives its entries; table-fed fused form: the tail of
      //  k_vec_step has just written the pose / camera tables of x_new -- no table kernel at all)
      timed_linearize(spec_tables_ready ? nullptr : h->xnew.p, 0, 0, h->allreduce != nullptr && merged_trial_cost);
      spec_lin = true;
      if (merged_trial_cost) {      // the cost of x_new arrived with the linearisation's all-reduced [g | diag | cost]
        HIP_OK(hipMemcpyAsync(h->scal.p + sl.costp, h->costcount(), sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        fetch_scalars_begin(h, trial_fetch_end);
      }
      if (merged_trial_cost) trial_cost_values = 1;
      if (h->allreduce && merged_trial_cost && !spec_accept_off) {
        // Frame-sharded speculation (round 4): the scaling and the curvature of x_new and their message -- [|g|_inf, |g_h|^2,
        // |x scale|^2, curvature] per rank -- go out BEHIND the copy the host decides on, into the second buffer set: every rank
        // takes the same decision from the same all-reduced cost, so the sequence of collectives stays the same on all of them;
        // an accepted step finds message 2 of its point done (the host used to wait for message 1 before it enqueued it), a
        // rejected one has wasted 4 W doubles.
        hipLaunchKernelGGL(k_vec_scale_q00, dim3(sl.nvb + q00_blocks), dim3(256), 0, h->stream, d, h->xnew.p, h->g(), h->diag(),
                           h->scale_inv.p, h->scale_inv2.p, h->dsc2.p, h->gh2.p, 0, h->scal.p + sl.vs, h->costcount(),
                           h->scal.p + TR_COST, sl.nvb, h->Hss.p, h->Hfs.p, h->Hff.p, h->scal.p + sl.q00p);
        hipLaunchKernelGGL(k_shard_fold2, dim3(1), dim3(64), 0, h->stream, h->scal.p + sl.vs, sl.nvb, h->scal.p + sl.q00p,
                           q00_blocks, d.shard_rank, d.shard_world, h->scal.p + sl.shard2);
        call_allreduce(h, h->scal.p + sl.shard2, (size_t)4 * d.shard_world, 0);
        spec_scaled = true;
      }
      mark("iteration enqueued");
      if (side_cost && publish) publish_scalars_end(h, h->stream2);
      else fetch_scalars_end(h);
      mark("trial cost fetched");
      have_trial = true;
      }
    }
    if (fresh_lin) { collect_lin_time(); fresh_lin = false; }
    if (!have_trial) {   // fold the k_vec_scale partials on the host (frame-sharded: the per-rank values, gathered by summation)
      double mx = 0, gg = 0, xs = 0;
      const int nfold = h->allreduce ? d.shard_world : sl.nvb, at = h->allreduce ? sl.shard2 : sl.vs;
      for (int blk = 0; blk < nfold; ++blk) {
        mx = std::max(mx, S[at + 3 * blk]);
        gg += S[at + 3 * blk + 1];
        xs += S[at + 3 * blk + 2];
      }
      S[TR_GNORM] = mx; S[TR_GH2] = gg; S[TR_XS2] = xs;
    }
    g_norm = S[TR_GNORM];
    if (first) {
      cost = S[TR_COST];
      initial_cost = cost;
      if (!std::isfinite(cost))
        throw Error("Residuals are not finite in the initial point.");   // scipy least_squares.py:844-845
      if (have_trial) {
        Delta = S[TR_DELTA];                       // trf.py:428-430, evaluated by k_tr_reg
      } else {
        Delta = std::sqrt(S[TR_XS2]);
        if (Delta == 0) Delta = 1.0;
      }
      first = false;
    }
    if (g_norm < gtol) status = 1;
    if (h->log && opt->verbose >= 2) h->log(h->log_ctx, iteration, nfev, cost, actual_reduction, step_norm, g_norm);
    if (status != -100 || nfev >= max_nfev) break;   // (a step the device has already evaluated is simply dropped)

    // ---- regularised Gauss-Newton step + 2-D subspace ------------------------------------------------------------
    const int32_t chol_info = (int32_t)S[TR_INFO];
    if (chol_info != 0)
      throw Error("reduced normal equations are not positive definite (pivot " + std::to_string(chol_info) +
                  "); non-finite Jacobian?");

    actual_reduction = -1;
    double cost_new = cost, ratio = 0;
    bool spec_valid = spec_lin;   // true while the trial under evaluation is the one the speculation was made for
    while (actual_reduction <= 0 && nfev < max_nfev) {
      if (!have_trial) {
        spec_valid = false;
        tr_trial(S, Delta);
        enqueue_trial(S[TR_ALPHA], S[TR_BETA], nullptr);
        fetch_scalars(h, trial_fetch_end - sl.step, sl.step);
      }
      have_trial = false;
      double step_h2, step2, x2;
      cost_new = fold_trial(&step_h2, &step2, &x2, trial_cost_values);
      trial_cost_values = cost_fetch;   // (retries evaluate the cost with k_cost)
      const double predicted = S[TR_PRED];
      ++nfev;
      const double step_h_norm = std::sqrt(step_h2);
      if (!std::isfinite(cost_new)) {
        Delta = 0.25 * step_h_norm;
        continue;
      }
      actual_reduction = cost - cost_new;
      double Delta_new = Delta;
      tr_update_radius(Delta_new, actual_reduction, predicted, step_h_norm, step_h_norm > 0.95 * Delta, ratio);
      step_norm = std::sqrt(step2);
      const double x_norm = std::sqrt(x2);
      status = tr_check_termination(actual_reduction, cost, step_norm, x_norm, ratio, ftol, xtol);
      if (status != -100) break;
      Delta = Delta_new;
    }

    if (actual_reduction > 0) {
      std::swap(h->x.p, h->xnew.p);
      cost = cost_new;
      if (spec_valid && spec_scaled) {   // the scaling / curvature computed ahead belong to the accepted point
        std::swap(h->scale_inv.p, h->scale_inv2.p);
        std::swap(h->dsc.p, h->dsc2.p);
        std::swap(h->gh.p, h->gh2.p);
        scaled_ahead = true;
      }
      if (!spec_valid) timed_linearize(h->x.p);   // (x now points at the accepted x_new)
      fresh_lin = true;
      ++njev;
    } else {
      step_norm = 0;
      actual_reduction = 0;
      if (spec_lin) lin_stale = true;
    }
    if (spec_lin && !spec_valid && actual_reduction > 0) lin_stale = false;   // re-linearised at the accepted retry
    ++iteration;
  }
  if (status == -100) status = 0;

  mark("loop left");
  // (through the pinned staging buffer, like every other small result.  Known one-time cost of the HIP runtime: ONE of the
  //  first device-to-host copies of this size in a process spends ~8 ms inside hipMemcpyAsync on the host -- seen in the
  //  first or second solve of a process at cfg3 (n = 6140), not at cfg2, pinned or pageable destination alike;
  //  MCBA_SOLVE_TRACE=1 prints the driver's stage times)
  gather_frame_entries(h, h->x.p);   // (frame-sharded: every rank returns the complete x; ONE n_motion message per solve)
  HIP_OK(hipMemcpyAsync(h->h_x, h->x.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  to_caller(h, x_inout, h->h_x);
  if (result) {
    result->cost = cost;
    result->initial_cost = initial_cost;
    result->optimality = g_norm;
    result->nfev = nfev;
    result->njev = njev;
    result->status = status;
    result->iterations = iteration;
    mark("done");
    result->solve_seconds = now_seconds() - t_start;
    result->linearize_seconds = lin_ms_total * 1e-3;
  }
  API_END
}


/* errors of Calibration.reprojection_error (inliers_only = 0) / reprojection_inliers (1) reduced ON THE DEVICE:
 * n = number of masked points, sum_sq = sum of squared errors, values[i] = exact order statistic of rank ranks[i]
 * (0-based, ascending) found by radix select -- the inputs numpy.quantile needs (calibration.py:37-40,304-310).       */
namespace {
// inlier_sums (may be null; single handles with n_ranks > 0 only): {sum of squares, count} of the INLIERS as well, from the same
// error pass and behind the same synchronisation (the report of the outlier loop asks for both: mcba_adjust_outliers)
void error_stats_impl(mcba_handle h, const double* x, int32_t inliers_only, int32_t n_ranks, const int64_t* ranks,
                      double* values, int64_t* n_out, double* sum_sq, double* inlier_sums) {
  REQUIRE(h && x && n_out && sum_sq, "null argument");
  REQUIRE(n_ranks == 0 || (ranks && values), "null argument");
  g_fill_stream = h->stream;
  const Dims& d = h->d;
  compute_errors(h, x);
  const uint8_t* m2 = inliers_only ? h->inlier.p : nullptr;
  const int grid = sel_grid(d);
  if (h->costpart.n < (size_t)2 * grid) h->costpart.alloc((size_t)2 * std::max(grid, COST_BLOCKS_MAX));
  hipLaunchKernelGGL(k_err_sums, dim3(grid), dim3(256), 0, h->stream, h->err_fm.p, h->evalid.p, m2, d.slots(), h->costpart.p);
  hipLaunchKernelGGL(k_sum2, dim3(1), dim3(256), 0, h->stream, h->costpart.p, grid, h->scal.p);
  call_allreduce(h, h->scal.p, 2, 0);
  int64_t n;
  const bool count_known = !h->allreduce;   // single handle: the counts are host-side facts (mcba_error_count)
  REQUIRE(inlier_sums == nullptr || (count_known && n_ranks > 0), "internal: inlier sums ride with a selection on a single handle");
  if (count_known && n_ranks > 0) {
    // one synchronisation for the whole call: the sums come down behind the selection passes
    n = inliers_only ? h->n_inliers : h->n_evalid;
    if (inlier_sums != nullptr) {
      hipLaunchKernelGGL(k_err_sums, dim3(grid), dim3(256), 0, h->stream, h->err_fm.p, h->evalid.p, (const uint8_t*)h->inlier.p,
                         d.slots(), h->costpart.p);
      hipLaunchKernelGGL(k_sum2, dim3(1), dim3(256), 0, h->stream, h->costpart.p, grid, h->scal.p + 2);
    }
    HIP_OK(hipMemcpyAsync(h->h_scal, h->scal.p, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  } else {
    fetch_scalars(h, 2);
    n = (int64_t)h->h_scal[1];
  }
  // consecutive ranks (floor / ceil of a virtual index) share one selection; all selections of a batch share the six
  // passes over the errors
  std::vector<long long> sel;          // distinct selections
  std::vector<int> which(n_ranks, -1), use_next(n_ranks, 0);
  for (int i = 0; i < n_ranks; ++i) {
    REQUIRE(n > 0 && ranks[i] >= 0 && ranks[i] < n, "order-statistic rank out of range");
    int found = -1;
    for (size_t k = 0; k < sel.size(); ++k) {
      if (sel[k] == ranks[i]) { found = (int)k; use_next[i] = 0; break; }
      if (sel[k] + 1 == ranks[i]) { found = (int)k; use_next[i] = 1; break; }
    }
    if (found < 0) { sel.push_back(ranks[i]); found = (int)sel.size() - 1; }
    which[i] = found;
  }
  std::vector<double> vk(sel.size()), vk1(sel.size());
  for (size_t k0 = 0; k0 < sel.size(); k0 += SEL_MAX) {
    const int nsel = (int)std::min<size_t>(SEL_MAX, sel.size() - k0);
    select_ranks_multi(h, m2, nsel, sel.data() + k0, vk.data() + k0, vk1.data() + k0);
  }
  for (int i = 0; i < n_ranks; ++i) values[i] = use_next[i] ? vk1[which[i]] : vk[which[i]];
  *sum_sq = h->h_scal[0];                     // (select_ranks_multi synchronised the stream)
  REQUIRE(!count_known || n_ranks == 0 || (int64_t)h->h_scal[1] == n, "inlier count out of sync with the device table");
  *n_out = n;
  if (inlier_sums != nullptr) {
    inlier_sums[0] = h->h_scal[2];
    inlier_sums[1] = h->h_scal[3];
  }
}
}  // namespace

int32_t mcba_error_stats(mcba_handle h, const double* x, int32_t inliers_only, int32_t n_ranks, const int64_t* ranks,
                         double* values, int64_t* n_out, double* sum_sq) {
  API_BEGIN
  error_stats_impl(h, x, inliers_only, n_ranks, ranks, values, n_out, sum_sq, nullptr);
  API_END
}

namespace {
// numpy.quantile(errors, q) (default method 'linear') from exact order statistics: virtual index (n - 1) q, its floor / ceil
// ranks by radix select on the device, numpy's _lerp on the host (numpy/lib/_function_base_impl.py)
struct ErrorStats { int64_t n = 0; double sum_sq = 0.0; std::vector<double> quantiles; };
int32_t error_stats_with_quantiles(mcba_handle h, const double* x, int inliers_only, const std::vector<double>& q, ErrorStats& out,
                                   ErrorStats* inl_out = nullptr) {
  out.quantiles.assign(q.size(), 0.0);
  if (inl_out != nullptr) inl_out->n = -1;      // (-1: not delivered -- the caller asks again)
  int64_t n = 0;
  if (int32_t rc = mcba_error_count(h, inliers_only, &n)) return rc;
  double ssq = 0.0;
  if (n < 0 || q.empty()) {
    if (int32_t rc = mcba_error_stats(h, x, inliers_only, 0, nullptr, nullptr, &n, &ssq)) return rc;
  }
  out.n = n;
  out.sum_sq = ssq;
  if (n == 0 || q.empty()) return 0;
  std::vector<int64_t> ranks(2 * q.size());
  std::vector<double> gamma(q.size()), vals(2 * q.size());
  for (size_t i = 0; i < q.size(); ++i) {
    const double virt = (double)(n - 1) * q[i], fl = std::floor(virt);
    const int64_t lo = std::min<int64_t>(std::max<int64_t>((int64_t)fl, 0), n - 1);
    ranks[2 * i] = lo;
    ranks[2 * i + 1] = std::min<int64_t>((int64_t)fl + 1, n - 1);
    gamma[i] = virt - fl;
  }
  double isums[2] = {0.0, 0.0};
  const bool with_inl = inl_out != nullptr && !h->allreduce;
  try {
    error_stats_impl(h, x, inliers_only, (int32_t)ranks.size(), ranks.data(), vals.data(), &n, &ssq, with_inl ? isums : nullptr);
  } catch (const std::exception& e) {
    g_error = e.what();
    return 1;
  }
  if (with_inl) {
    inl_out->sum_sq = isums[0];
    inl_out->n = (int64_t)isums[1];
  }
  out.sum_sq = ssq;
  for (size_t i = 0; i < q.size(); ++i) {
    const double a = vals[2 * i], b = vals[2 * i + 1], diff = b - a;
    out.quantiles[i] = gamma[i] >= 0.5 ? b - diff * (1.0 - gamma[i]) : a + diff * gamma[i];
  }
  return 0;
}
}  // namespace

/* Calibration.adjust_outliers (calibration.py:254-268) as ONE call: num_adjustments rounds of {report, optional f_scale from
 * a quantile of the errors, reject_outliers(quantile x factor), bundle_adjust}, then the final report -- the loop
 * Workspace.calibrate drives (workspace.py:238-244), without leaving the library between its steps: no Calibration objects,
 * no re-lowering and no rotation-vector <-> matrix round trips between the rounds (x continues with the solver's raw
 * rotation vectors; the reference canonicalises them in with_param_vec, which changes no projection).
 * rounds[i] (i <= num_adjustments) receives the report in front of round i (i = num_adjustments: the final report) and, for
 * i < num_adjustments, the threshold / f_scale / solve result of that round.  outlier_factor < 0: no rejection;
 * scale_factor < 0: f_scale = opt->f_scale.  inliers_out (may be NULL): the final inlier mask in [C,F,B,P] order.       */
int32_t mcba_adjust_outliers(mcba_handle h, double* x_inout, const mcba_options* opt, int32_t num_adjustments,
                             double outlier_quantile, double outlier_factor, double scale_quantile, double scale_factor,
                             mcba_round_report* rounds, uint8_t* inliers_out) {
  API_BEGIN
  REQUIRE(h && x_inout && opt && rounds && num_adjustments >= 0, "bad argument");
  const std::vector<double> five = {0.0, 0.25, 0.5, 0.75, 1.0};
  auto report = [&](mcba_round_report& r) -> int32_t {
    ErrorStats all, inl;
    if (int32_t rc = error_stats_with_quantiles(h, x_inout, 0, five, all, &inl)) return rc;    // (one pass, one synchronisation)
    if (inl.n < 0)
      if (int32_t rc = error_stats_with_quantiles(h, x_inout, 1, {}, inl)) return rc;
    r.n_all = all.n; r.n_inliers = inl.n;
    r.rms_all = all.n > 0 ? std::sqrt(all.sum_sq / (double)all.n) : 0.0;
    r.rms_inliers = inl.n > 0 ? std::sqrt(inl.sum_sq / (double)inl.n) : 0.0;
    for (int k = 0; k < 5; ++k) r.quantiles[k] = all.n > 0 ? all.quantiles[k] : 0.0;
    return 0;
  };
  auto quantile_of_all = [&](double q, const mcba_round_report& r, double* out) -> int32_t {
    for (int k = 0; k < 5; ++k)
      if (five[k] == q) { *out = r.quantiles[k]; return 0; }     // (the report has just selected it)
    ErrorStats st;
    if (int32_t rc = error_stats_with_quantiles(h, x_inout, 0, {q}, st)) return rc;
    *out = st.n > 0 ? st.quantiles[0] : 0.0;
    return 0;
  };
  const bool timing = getenv("MCBA_TIMING") != nullptr;
  double tt = now_seconds();
  auto lap = [&](const char* what, int i) {
    if (!timing) return;
    const double now = now_seconds();
    fprintf(stderr, "[mcba_adjust_outliers] round %d %-8s %.3f ms\n", i, what, (now - tt) * 1e3);
    tt = now;
  };
  for (int i = 0; i < num_adjustments; ++i) {
    mcba_round_report& r = rounds[i];
    memset(&r, 0, sizeof(r));
    if (int32_t rc = report(r)) return rc;
    lap("report", i);
    mcba_options o = *opt;
    r.f_scale = opt->f_scale;
    if (scale_factor >= 0.0) {
      double qv = 0.0;
      if (int32_t rc = quantile_of_all(scale_quantile, r, &qv)) return rc;
      r.f_scale = qv * scale_factor != 0.0 ? qv * scale_factor : 1.0;    // `... or 1.0` (calibration.py:259)
      o.f_scale = r.f_scale;
    }
    r.threshold = -1.0;
    if (outlier_factor >= 0.0) {
      double qv = 0.0;
      if (int32_t rc = quantile_of_all(outlier_quantile, r, &qv)) return rc;
      r.threshold = qv * outlier_factor;
      if (int32_t rc = mcba_reject_outliers(h, x_inout, r.threshold, &r.n_kept, &r.n_valid)) return rc;
    }
    lap("reject", i);
    if (int32_t rc = mcba_solve(h, x_inout, &o, &r.solve)) return rc;
    lap("solve", i);
  }
  memset(&rounds[num_adjustments], 0, sizeof(mcba_round_report));
  if (int32_t rc = report(rounds[num_adjustments])) return rc;
  lap("report", num_adjustments);
  if (inliers_out) {
    if (int32_t rc = mcba_get_inliers(h, inliers_out)) return rc;
  }
  API_END
}

/* number of points behind mcba_error_stats WITHOUT touching the device: lets the caller compute numpy's quantile ranks
 * first and make a single mcba_error_stats call.  -1 for a frame-sharded handle (the global count needs a reduction).  */
int32_t mcba_error_count(mcba_handle h, int32_t inliers_only, int64_t* n) {
  API_BEGIN
  REQUIRE(h && n, "null argument");
  *n = h->allreduce ? -1 : (inliers_only ? h->n_inliers : h->n_evalid);
  API_END
}

/* Calibration.reject_outliers on the device (calibration.py:240-252): inliers = (err < threshold) & valid, evaluated
 * at x; the handle's inlier table, per-view counts (and, lazily, the residual ordering) are replaced.               */
int32_t mcba_reject_outliers(mcba_handle h, const double* x, double threshold, int64_t* n_inliers, int64_t* n_valid) {
  API_BEGIN
  REQUIRE(h && x, "null argument");
  g_fill_stream = h->stream;
  const Dims& d = h->d;
  compute_errors(h, x);
  if (d.views() > 0)
    hipLaunchKernelGGL(k_reject, dim3(d.views()), dim3(64), 0, h->stream, d, h->err_fm.p, h->evalid.p, threshold, h->inlier.p,
                       h->view_count.p);
  refresh_active_views(h);
  const int grid = sel_grid(d);
  if (h->costpart.n < (size_t)2 * grid) h->costpart.alloc((size_t)2 * std::max(grid, COST_BLOCKS_MAX));
  hipLaunchKernelGGL(k_err_sums, dim3(grid), dim3(256), 0, h->stream, h->err_fm.p, h->evalid.p, h->inlier.p, d.slots(),
                     h->costpart.p);
  hipLaunchKernelGGL(k_sum2, dim3(1), dim3(256), 0, h->stream, h->costpart.p, grid, h->scal.p);
  hipLaunchKernelGGL(k_err_sums, dim3(grid), dim3(256), 0, h->stream, h->err_fm.p, h->evalid.p, (const uint8_t*)nullptr,
                     d.slots(), h->costpart.p);
  hipLaunchKernelGGL(k_sum2, dim3(1), dim3(256), 0, h->stream, h->costpart.p, grid, h->scal.p + 2);
  fetch_scalars(h, 4);
  h->n_inliers = (int64_t)h->h_scal[1];      // this shard's inliers
  h->obs_index_dirty = true; h->compact_dirty = true;
  h->view_first_dirty = true;
  h->out_r.alloc((size_t)std::max<int64_t>(2 * h->n_inliers, 1), false);
  double tot[2] = {h->h_scal[1], h->h_scal[3]};
  if (h->allreduce) {
    HIP_OK(hipMemcpyAsync(h->scal.p, tot, sizeof(tot), hipMemcpyHostToDevice, h->stream));
    call_allreduce(h, h->scal.p, 2, 0);
    fetch_scalars(h, 2);
    tot[0] = h->h_scal[0];
    tot[1] = h->h_scal[1];
  }
  if (n_inliers) *n_inliers = (int64_t)tot[0];
  if (n_valid) *n_valid = (int64_t)tot[1];
  API_END
}

/* current inlier table in the reference's [C,F,B,P] order (frames of other shards are zero)                          */
int32_t mcba_get_inliers(mcba_handle h, uint8_t* mask) {
  API_BEGIN
  REQUIRE(h && mask, "null argument");
  g_fill_stream = h->stream;
  const Dims& d = h->d;
  const size_t nref = (size_t)d.C * d.F * d.B * d.P;
  h->out_valid.alloc(nref, true);
  hipLaunchKernelGGL(k_inliers_to_ref, dim3(std::max(1, std::min(4096, (d.slots() + 255) / 256))), dim3(256), 0, h->stream, d,
                     h->inlier.p, h->out_valid.p);
  HIP_OK(hipMemcpyAsync(mask, h->out_valid.p, nref, hipMemcpyDeviceToHost, h->stream));
  sync(h);
  API_END
}

int32_t mcba_time_linearize(mcba_handle h, const double* x, const mcba_options* opt, int32_t repeats, double* avg_ms) {
  API_BEGIN
  REQUIRE(h && x && avg_ms && repeats > 0, "bad argument");
  set_loss(h, opt);
  upload_x(h, x, h->x.p);
  launch_linearize(h, h->x.p);   // warm-up (the table form: k_tmat prepares every table the dominant kernel reads)
  sync(h);
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < repeats; ++i) {   // the dominant kernel alone, as rocprofv3 reports it (table form: k_tmat ran above)
    const Dims& d = h->d;
    if (h->use_mfma && d.off_boards < 0 && !linearize_table_form())
      launch_fused_linearize_kernel(h);
    else
      h->ops->linearize(d, h->t, h->stream, h->rec.p, h->tri.p, h->use_mfma, h->lin_grid, nullptr, nullptr, 0, nullptr, 0, nullptr);
  }
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  sync(h);
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *avg_ms = ms / repeats;
  API_END
}

/* test hook (mcba_debug.h): one Golub-Kahan step through the kernels the default solver ITERATES with -- k_lsmr_fused2 (uhat = J v with
 * alpha = 0, and the per-view partials of J^T uhat from the same pass) and k_lsmr_gather3 (v_raw = J^T uhat / beta - beta v, beta = |uhat|)
 * -- at x, unscaled columns, linear loss: jv_out[m] = J(x) v, jtjv_out[n] = J(x)^T J(x) v (recovered as (v_raw + beta v) beta).  */
int32_t mcba_debug_lsmr_fused_products(mcba_handle h, const double* x, const double* v, double* jv_out, double* jtjv_out) {
  API_BEGIN
  REQUIRE(h && x && v && jv_out && jtjv_out, "null argument");
  REQUIRE(h->allreduce == nullptr, "single handles only");
  g_fill_stream = h->stream;
  set_loss(h, nullptr);
  const Dims& d = h->d;
  LsmrOps op = lsmr_setup(h);
  const size_t m = op.m;
  upload_x(h, x, h->x.p);
  sync(h);
  lsmr_linearize(h, h->x.p);
  std::vector<double> ones((size_t)d.n, 1.0);
  HIP_OK(hipMemcpyAsync(h->dsc.p, ones.data(), (size_t)d.n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  sync(h);
  upload_x(h, v, h->ls_v.p);                                   // (internal layout; padded coefficients zero)
  HIP_OK(hipMemsetAsync(h->ls_u.p, 0, std::max<size_t>(m, 2) * sizeof(double), h->stream));
  double* s0 = h->ls_state.p;
  double* s1 = s0 + LS_NSLOTS;
  op.ensure_part2(true);
  hipLaunchKernelGGL(k_lsmr_init, dim3(1), dim3(64), 0, h->stream, s0, /*alpha*/ 0.0, /*beta*/ 1.0, 0.0, 1.0, 1e9);
  int cached = (d.off_boards < 0 && !h->lsmr_masks_form) ? 3 : 0;   // (k_lsmr_fused2's MODE: compact tables unless boards=True)
  if (cached != 0) ensure_compact(h);
  if (h->lsmr_fused == 3 && cached == 3) {          // the cached form: one evaluating pass fills the cache, the pass under test streams it back
    const size_t need = (((size_t)h->n_inliers + 63) / 64) * 64 * (size_t)lsmr_cache_components(d.motion, d.loss) + 64;
    if (h->ls_cache.n < need) h->ls_cache.alloc(need, false);
    h->ops->lsmr_fused2(d, h->t, h->stream, h->view_first.p, h->dsc.p, h->ls_v.p, h->ls_u.p, h->ls_partial.p, h->ls_xpart.p, h->ls_part2.p,
                        op.part_stride, op.bpart(), op.nblk, s0, s1, h->ls_vpart.p, op.gather3_grid(), h->ls_hbar.p, h->ls_x.p, h->ls_h.p, h->ls_cache.p, 4,
                        compact_tables(h));
    cached = 2;
  }
  h->ops->lsmr_fused2(d, h->t, h->stream, h->view_first.p, h->dsc.p, h->ls_v.p, h->ls_u.p, h->ls_partial.p, h->ls_xpart.p, h->ls_part2.p,
                      op.part_stride, op.bpart(), op.nblk, s0, s1, h->ls_vpart.p, op.gather3_grid(), h->ls_hbar.p, h->ls_x.p, h->ls_h.p, h->ls_cache.p, cached, compact_tables(h));
  hipLaunchKernelGGL(k_lsmr_gather3, dim3(op.gather3_grid()), dim3(LSG3_THREADS), 0, h->stream, d, (const double*)h->ls_part2.p, op.part_stride,
                     (const double*)h->dsc.p, (const double*)h->ls_v.p, h->ls_vraw.p, h->ls_nrm.p, h->ls_vpart.p, (const double*)s1, h->ls_out.p,
                     (const double*)h->ls_partial.p, op.nblk, (const double*)h->ls_xpart.p, std::max(1, std::min(op.nblk, (d.n + 63) / 64)),
                     0ull, h->h_pub_seq + 1, op.extra());
  check_launch("k_lsmr_fused2 / k_lsmr_gather3");
  HIP_OK(hipMemcpyAsync(jv_out, h->ls_u.p, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  std::vector<double> vraw((size_t)d.n), vint((size_t)d.n);
  HIP_OK(hipMemcpyAsync(vraw.data(), h->ls_vraw.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipMemcpyAsync(vint.data(), h->ls_v.p, (size_t)d.n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  double beta = 0.0;   // the gather's own beta = |uhat| (published with its state): the recovery below is then exact in the zero columns
  HIP_OK(hipMemcpyAsync(&beta, h->ls_out.p + LS_BETA, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  sync(h);
  h->h_pub_seq[1] = 0;
  for (int i = 0; i < d.n; ++i) {
#pragma clang fp contract(off)      // (beta v must be rounded as the kernel rounded it before it is added back)
    const double bv = beta * vint[i];
    h->h_x[i] = (vraw[i] + bv) * beta;
  }
  to_caller(h, jtjv_out, h->h_x);
  API_END
}

/* average launch duration (HIP events on the handle's stream) of the two kernels of an LSMR iteration of the default solver at x:
 * ms[0] = k_lsmr_fused2 (both Jacobian products of a Golub-Kahan step), ms[1] = k_lsmr_gather3 -- the live numbers behind
 * bench.py's `parity_route.roofline`.  Single (unsharded) handles; linear loss.                                                  */
int32_t mcba_time_lsmr_iteration(mcba_handle h, const double* x, int32_t repeats, double* ms /*[2]*/) {
  API_BEGIN
  REQUIRE(h && x && ms && repeats > 0, "bad argument");
  REQUIRE(h->allreduce == nullptr, "single handles only");
  g_fill_stream = h->stream;
  set_loss(h, nullptr);
  const Dims& d = h->d;
  LsmrOps op = lsmr_setup(h);
  upload_x(h, x, h->x.p);
  sync(h);
  lsmr_linearize(h, h->x.p);
  std::vector<double> ones((size_t)d.n, 1.0);
  HIP_OK(hipMemcpyAsync(h->dsc.p, ones.data(), (size_t)d.n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->ls_v.p, ones.data(), (size_t)d.n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  double* s0 = h->ls_state.p;
  double* s1 = s0 + LS_NSLOTS;
  op.ensure_part2(true);
  hipLaunchKernelGGL(k_lsmr_init, dim3(1), dim3(64), 0, h->stream, s0, 1.0, 1.0, 0.0, 1.0, 1e9);   // (no pending rotation: the product alone)
  int cached = (d.off_boards < 0 && !h->lsmr_masks_form) ? 3 : 0;   // (k_lsmr_fused2's MODE: compact tables unless boards=True)
  if (cached != 0) ensure_compact(h);
  if (h->lsmr_fused == 3 && cached == 3) {          // the cached form: one evaluating pass fills the cache, the pass under test streams it back
    const size_t need = (((size_t)h->n_inliers + 63) / 64) * 64 * (size_t)lsmr_cache_components(d.motion, d.loss) + 64;
    if (h->ls_cache.n < need) h->ls_cache.alloc(need, false);
    h->ops->lsmr_fused2(d, h->t, h->stream, h->view_first.p, h->dsc.p, h->ls_v.p, h->ls_u.p, h->ls_partial.p, h->ls_xpart.p, h->ls_part2.p,
                        op.part_stride, op.bpart(), op.nblk, s0, s1, h->ls_vpart.p, op.gather3_grid(), h->ls_hbar.p, h->ls_x.p, h->ls_h.p, h->ls_cache.p, 4,
                        compact_tables(h));
    cached = 2;
  }
  sync(h);
  auto product = [&]() {
    h->ops->lsmr_fused2(d, h->t, h->stream, h->view_first.p, h->dsc.p, h->ls_v.p, h->ls_u.p, h->ls_partial.p, h->ls_xpart.p, h->ls_part2.p,
                        op.part_stride, op.bpart(), op.nblk, s0, s1, h->ls_vpart.p, op.gather3_grid(), h->ls_hbar.p, h->ls_x.p, h->ls_h.p, h->ls_cache.p, cached, compact_tables(h));
  };
  auto gather = [&]() {   // (writes its state to the spare half of s1's buffer is not possible: a scratch copy keeps s0 untouched)
    hipLaunchKernelGGL(k_lsmr_gather3, dim3(op.gather3_grid()), dim3(LSG3_THREADS), 0, h->stream, d, (const double*)h->ls_part2.p, op.part_stride,
                       (const double*)h->dsc.p, (const double*)h->ls_v.p, h->ls_vraw.p, h->ls_nrm.p, h->ls_vpart.p, (const double*)s1, h->ls_out.p,
                       (const double*)h->ls_partial.p, op.nblk, (const double*)h->ls_xpart.p, std::max(1, std::min(op.nblk, (d.n + 63) / 64)),
                       0ull, h->h_pub_seq + 1, op.extra());
  };
  product();
  gather();
  sync(h);
  float t = 0.f;
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < repeats; ++i) product();
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  sync(h);
  HIP_OK(hipEventElapsedTime(&t, h->ev0, h->ev1));
  ms[0] = t / repeats;
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < repeats; ++i) gather();
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  sync(h);
  HIP_OK(hipEventElapsedTime(&t, h->ev0, h->ev1));
  ms[1] = t / repeats;
#if defined(MCBA_EXP_F2_PROF)   // profiling build: one more product with per-workgroup stamps, summary on stderr
  {
    const size_t nw = 8 * (size_t)op.nblk;
    if (h->ls_cache.n < nw) h->ls_cache.alloc(nw, true);
    product();
    sync(h);
    std::vector<long long> pf(nw);
    HIP_OK(hipMemcpy(pf.data(), h->ls_cache.p, nw * sizeof(long long), hipMemcpyDeviceToHost));
    long long tmin = pf[0], tmax = pf[1];
    for (int i = 0; i < op.nblk; ++i) { tmin = std::min(tmin, pf[8 * i]); tmax = std::max(tmax, pf[8 * i + 1]); }
    std::vector<double> dur, st, ch, ep, endt;
    double views = 0, chunks = 0;
    for (int i = 0; i < op.nblk; ++i) {
      dur.push_back((double)(pf[8 * i + 1] - pf[8 * i])); st.push_back((double)pf[8 * i + 2]); ch.push_back((double)pf[8 * i + 3]);
      ep.push_back((double)pf[8 * i + 4]); endt.push_back((double)(pf[8 * i + 1] - tmin)); views += pf[8 * i + 5]; chunks += pf[8 * i + 6];
    }
    auto stat = [](std::vector<double> v) { std::sort(v.begin(), v.end()); double s = 0; for (double x : v) s += x;
                                            return std::array<double, 4>{s / v.size(), v[v.size() / 2], v[v.size() * 9 / 10], v.back()}; };
    auto pr = [&](const char* name, const std::vector<double>& v) { auto q = stat(v); fprintf(stderr, "  %-28s mean %9.0f  median %9.0f  p90 %9.0f  max %9.0f\n", name, q[0], q[1], q[2], q[3]); };
    fprintf(stderr, "[f2 profile] %d workgroups, %.0f views, %.0f chunks; kernel span %lld clocks (clock64 units)\n", op.nblk, views, chunks, tmax - tmin);
    pr("workgroup duration", dur); pr("workgroup end (from first start)", endt); pr("  staging + That v per wg", st); pr("  chunk loop per wg", ch); pr("  reduce + That^T + stores per wg", ep);
    std::vector<double> startt; for (int i = 0; i < op.nblk; ++i) startt.push_back((double)(pf[8 * i] - tmin));
    pr("workgroup start (from first)", startt);
    // per-SIMD load: group by HW_ID (cu / simd bits) -- sum of durations per (xcc-less) hardware id
    std::map<unsigned, double> simd;
    for (int i = 0; i < op.nblk; ++i) simd[(unsigned)pf[8 * i + 7] & 0xfffffff0u] += dur[i];
    std::vector<double> sv; for (auto& kv : simd) sv.push_back(kv.second);
    fprintf(stderr, "  distinct hardware slots (HW_ID without the wave bits): %zu\n", sv.size());
    pr("sum of wg durations per slot", sv);
  }
#endif
  h->h_pub_seq[1] = 0;   // (the timed gathers published progress words of call 0: forget them)
  API_END
}

int32_t mcba_time_residuals(mcba_handle h, const double* x, int32_t repeats, double* avg_ms) {
  API_BEGIN
  REQUIRE(h && x && avg_ms && repeats > 0, "bad argument");
  ensure_view_first(h);
  upload_x(h, x, h->x.p);
  eval_tables(h, h->x.p);
  h->ops->residual(h->d, h->t, h->stream, h->view_first.p, h->out_r.p, nullptr, nullptr, nullptr);
  sync(h);
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < repeats; ++i)
    h->ops->residual(h->d, h->t, h->stream, h->view_first.p, h->out_r.p, nullptr, nullptr, nullptr);
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  sync(h);
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *avg_ms = ms / repeats;
  API_END
}

}  // extern "C"
