#!/usr/bin/env python
"""bench.py -- the hot path's headline metric on MI355X (contract: see the task statement / DESIGN.md section 6).

  python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU, either started by the caller through
                                                          torch.distributed.run or, when WORLD_SIZE is not set, by
                                                          bench.py itself -- the same launcher on 127.0.0.1)

metric  : residual+Jacobian evaluations per second (BASELINE.json `metric`, first component); LM iterations/s and
          the final reprojection RMS of a full solve are reported alongside in the same JSON line.
step    : ONE fused residual+Jacobian evaluation reduced to the normal equations at one x -- table preparation,
          k_linearize (MFMA), assembly into H_ss / H_fs / H_ff / g, and (N > 1) the all-reduce of [g | diag | cost]:
          what one scipy `jac` call (1 + 34 finite-difference `evaluate` calls) plus J^T J / J^T f costs the reference.
          x, the tables and the results are resident in HBM (the solver's own evaluations never leave the device); the
          same evaluation through the host boundary (x up, cost down, one sync per call) is reported as `host_boundary`.
workload: BASELINE.json configs[2], the rig the north star quotes the metric on: 8 cameras x 500 frames x 2 boards
          (charuco_16x22 + aprilgrid_9x9), rolling-shutter motion model, intrinsics + extrinsics optimised, synthetic
          data (multical_amd.synthetic, seed 3).  Inputs are resident in HBM before the timed region.
scaling : `value` is evaluations/s of ONE FIXED rig -- BASELINE.json's north star: "on a synthetic 8-cam x 500-frame x 2-board
          rig reported at 1/2/4/8 GPUs".  For N > 1 that rig is frame-sharded over the N ranks (500 / N frames each)
          and every evaluation ends with the all-reduce of the shared [g | diag | cost]: STRONG scaling (`"scaling": "strong"`).
          At 62 frames per GPU k_linearize is latency-bound and ~23 us of fixed-cost kernels + the collective do not shrink, so
          the curve is Amdahl-limited (README).  The same line carries `weak_scaling` for N > 1: every rank owns a full
          500-frame shard of one 8 x (500 N) x 2 rig, shard evaluations/s of the whole job.
          --config cfg4 measures BASELINE configs[3] (16 cameras x 1000 frames x 5 cube faces, "frame-sharded across 8 x MI355X")
          with the same protocol.
parity  : `parity_route` = the solver that REPRODUCES THE REFERENCE'S END POINT (solver = "lsmr", the product default): whole-solve
          seconds, nfev / status, LM iterations/s, final RMS and its distance from the END POINT OF THE UNMODIFIED REFERENCE on this
          very rig (tests/golden/cfg3_endpoint.npz: Calibration.bundle_adjust() run to completion on one host core,
          oracle/make_endpoint.py), beside the reference's measured wall time.  `lm_iters_per_s` / `lm_long_solve` belong to the
          exact-step solver (solver = "native"), which ends at the converged optimum instead.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 matrix (= vector) peak: 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz (AMD spec;
                           # the guide's table stops at bf16/fp8 -- v_mfma_f64_16x16x4 measured at 64 cycles agrees)
RIGS = {   # BASELINE.json configs[2] (the north-star rig) and configs[3]
  "cfg3": dict(frames=500, label="BASELINE configs[2]: 8 cameras x 500 frames x 2 boards (charuco_16x22 + aprilgrid_9x9), "
                                 "rolling-shutter motion, intrinsics+extrinsics"),
  "cfg4": dict(frames=1000, label="BASELINE configs[3]: 16 cameras x 1000 frames x 5 boards (cube_10x10 faces), static frames, "
                                  "intrinsics+extrinsics"),
}
FRAMES_PER_SHARD = 500     # (cpu_baseline: the north-star rig)


def reference_endpoint(config):
  """END POINT of the unmodified reference on this rig at its stated size (committed fixture; oracle/make_endpoint.py)."""
  path = os.path.join(ROOT, "tests", "golden", f"{config}_endpoint.npz")
  if not os.path.exists(path):
    return None
  g = np.load(path, allow_pickle=False)
  out = dict(rms_px=float(g["ba_rms"]), cost=float(g["ba_cost"]), nfev=int(g["ba_nfev"]), njev=int(g["ba_njev"]),
             status=int(g["ba_status"]), seconds=float(g["ba_seconds"]), evaluate_calls=int(g["ba_evaluate_calls"]),
             peak_rss_gb=float(g["ba_peak_rss_gb"]), host=json.loads(str(g["host"])),
             source=f"tests/golden/{config}_endpoint.npz: Calibration.bundle_adjust() of the unmodified reference, one core of the build "
                    "container (8 vCPU Xeon 2.1 GHz), measured -- not extrapolated")
  if "ba_pert_rms" in g:
    out["spread_px"] = float(np.abs(g["ba_pert_rms"] - g["ba_rms"]).max())
    out["spread_runs"] = int(g["ba_pert_rms"].size)
  if "ba_tight_rms" in g:     # converged optimum of the reference's own residual function (oracle/make_endpoint.py tight)
    out["tight_rms_px"] = float(g["ba_tight_rms"])
    out["tight_cost"] = float(g["ba_tight_cost"])
  xp_path = os.path.join(ROOT, "tests", "golden", "exact_products.json")
  if os.path.exists(xp_path):   # scipy's own algorithm on the reference's residual function with 80-bit products (oracle/make_exact_products.py)
    xp = json.load(open(xp_path)).get(config)
    if xp and "longdouble_mean_rms" in xp:
      out["exact_product_rms_px"] = float(xp["longdouble_mean_rms"])
      out["exact_product_runs"] = len([r for r in xp["runs"] if r["arithmetic"] == "longdouble"])
  return out


def cpu_baseline(n_frames_sample=25, full_jacobian=False):
  """Reference CPU path (oracle = numpy/scipy port of the reference, bit-identical to it in-container) timed on THIS host.

  One residual+Jacobian evaluation of the reference = evaluate(x0) + scipy's grouped 2-point finite differences with the reference's
  sparsity: G + 1 `evaluate` calls for G column groups.  `value`: the port's `evaluate` timed at the FULL 500-frame rig (median of 5;
  nothing extrapolated in the number of frames) times the G + 1 = 40 calls the unmodified reference itself made per Jacobian on this rig
  (tests/golden/cfg3_endpoint.npz: 201 evaluate calls for nfev 5 / njev 5).  --cpu-baseline-full additionally runs ONE complete
  finite-difference Jacobian at full size (sparsity build ~75 s + G evaluations + the sparse assembly: several minutes).  The 25-frame
  sample of earlier rounds stays for the LSMR / trust-region figures, which need a Jacobian in memory."""
  from multical_amd import synthetic
  from oracle import restate
  from scipy.optimize._numdiff import approx_derivative, group_columns
  from scipy.sparse import csr_matrix
  # ---- full rig: evaluate(), measured
  rig_full = synthetic.make_rig("cfg3")
  oc_full = restate.from_rig(rig_full)
  xf = oc_full.param_vec
  oc_full.evaluate(xf)
  tev = []
  for _ in range(5):
    t0 = time.perf_counter()
    f_full = oc_full.evaluate(xf)
    tev.append(time.perf_counter() - t0)
  t_eval_full = sorted(tev)[2]
  ref = reference_endpoint("cfg3")
  calls_per_jacobian = 40
  if ref is not None and ref["njev"] > 0:   # (evaluate calls = nfev trial evaluations + 1 + njev * G  ->  G + 1 per residual+Jacobian evaluation)
    calls_per_jacobian = (ref["evaluate_calls"] - ref["nfev"] - 1) // ref["njev"] + 1
  full = dict(evaluate_seconds=t_eval_full, evaluate_seconds_all=tev, residuals=int(f_full.size), evaluate_calls_per_evaluation=calls_per_jacobian)
  if full_jacobian:
    t0 = time.perf_counter()
    S = csr_matrix(oc_full.sparsity_matrix)
    groups = group_columns(S)
    t_sp = time.perf_counter() - t0
    t0 = time.perf_counter()
    f0 = oc_full.evaluate(xf)
    Jf = approx_derivative(oc_full.evaluate, xf, method='2-point', f0=f0, sparsity=(S, groups))
    gf = Jf.T @ f0
    full.update(jacobian_seconds=time.perf_counter() - t0, sparsity_seconds=t_sp, groups=int(groups.max()) + 1,
                evals_per_s_measured=1.0 / (time.perf_counter() - t0))
    del Jf, S
  del oc_full, rig_full
  # ---- 25-frame sample: Jacobian, LSMR iteration, trust-region iterations
  rig = synthetic.make_rig("cfg3", frames=n_frames_sample)
  oc = restate.from_rig(rig)
  x0 = oc.param_vec
  t0 = time.perf_counter()
  S = csr_matrix(oc.sparsity_matrix)
  groups = group_columns(S)
  t_sparsity = time.perf_counter() - t0
  t0 = time.perf_counter()
  f0 = oc.evaluate(x0)
  J = approx_derivative(oc.evaluate, x0, method='2-point', f0=f0, sparsity=(S, groups))
  g = J.T @ f0                                    # the reduction the fused GPU pass also delivers
  dt = time.perf_counter() - t0
  evals_per_s_sample = 1.0 / dt
  # one LSMR iteration of the reference's linear solve = one J v and one J^T u (scipy lsmr: 2 sparse mat-vecs + O(m + n)
  # vector work); a TRF iteration runs up to min(m, n) of them (74 % of the reference's wall time)
  v = np.ones(x0.size)
  t1 = time.perf_counter()
  for _ in range(20):
    u = J @ v
    w = J.T @ u
  t_lsmr_iter = (time.perf_counter() - t1) / 20
  t1 = time.perf_counter()
  ba = oc.bundle_adjust(max_iterations=3, return_result=True)[1]
  t_ba = time.perf_counter() - t1
  lm_iters_per_s_sample = max(ba.nfev - 1, 1) / t_ba
  scale = n_frames_sample / FRAMES_PER_SHARD         # (only the LM figure below is still scaled from the sample)
  value = 1.0 / (calls_per_jacobian * t_eval_full)
  return dict(value=value, unit="evals/s", cores=1, kind="port",
              sample=(f"FULL rig ({FRAMES_PER_SHARD} frames, m={full['residuals']} residuals): evaluate() of the oracle port measured at full size, "
                      f"median of 5 = {t_eval_full:.2f} s, x {calls_per_jacobian} evaluate() calls per residual+Jacobian evaluation "
                      f"(the reference's own count on this rig: G = {calls_per_jacobian - 1} finite-difference column groups + f0); the sparsity "
                      f"build ({t_sparsity:.1f} s on the {n_frames_sample}-frame sample, ~75 s at full size) is not counted; host has "
                      f"{os.cpu_count()} logical cores, the numpy/scipy path is single-threaded"),
              full_rig=full,
              evals_per_s_on_sample=evals_per_s_sample, sample_frames=n_frames_sample,
              sample_extrapolated_evals_per_s=evals_per_s_sample * scale,
              lm_iters_per_s=lm_iters_per_s_sample * scale, lm_iters_per_s_on_sample=lm_iters_per_s_sample,
              lsmr_iteration_ms_on_sample=t_lsmr_iter * 1e3,
              lm_note=(f"reference TRF on the first {n_frames_sample} frames: {ba.nfev - 1} trial steps (max_nfev = 3) in {t_ba:.1f} s incl. "
                       f"finite-difference Jacobians and LSMR, scaled linearly to {FRAMES_PER_SHARD} frames; one LSMR iteration "
                       f"(J v + J^T u) = {t_lsmr_iter * 1e3:.1f} ms on the sample; the measured full-size figure is `reference_measured`"))


def self_launch(args):
  """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU) through
  torch.distributed.run on the loopback address, exactly as the driver's multi-GPU command line does, and pass rank 0's
  JSON line through.  Under torchrun (WORLD_SIZE set) this is never reached."""
  import socket
  import subprocess
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes needs dmabuf IPC on this driver
  return subprocess.call(cmd, env=env)


def dry_run(args):
  """`--dry-run`: the N-rank PLUMBING of this file without a GPU -- rank / world from the launcher's environment, gloo rendezvous on the
  loopback address, the frame-shard plan (with one EMPTY shard for N >= 3), the collective of an evaluation step with its real message
  layout [g_shared | diag_shared | cost, count | step norms] (2 n_s + 6 doubles) and real numbers (the rank's shard evaluated by the
  product's device functions compiled for the host: tests/hostmath, test infrastructure), the barrier + MAX-over-ranks timing of the
  contract, the watchdog and the one JSON line of rank 0 (flagged "dry_run": true; its `value` times the CPU stand-in and measures
  nothing).  So that the first multi-GPU run the driver manages is not also the first time this code path executes.
  Rank 0 checks the reduced [g | diag | cost] against the unsharded evaluation."""
  import torch
  import torch.distributed as dist
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  from hostmath_lib import HostMath
  from multical_amd import synthetic, calibration
  from multical_amd import distributed as mdist
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
  say = lambda msg: sys.stderr.write(f"[bench rank {rank}/{world}] {msg}\n")
  if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
  F = args.dry_frames
  rig = synthetic.make_rig(args.config, frames=F)
  calib = calibration.from_rig(rig)
  weights = calib.inliers.sum(axis=(0, 2, 3)).astype(np.float64)
  shards = mdist.frame_shards(F, world - 1, weights) + [(F, F)] if world >= 3 else mdist.frame_shards(F, world, weights)
  shard = shards[rank]
  hm = HostMath(calib, frame_range=shard if world > 1 else None)
  x0 = calib.param_vec
  n = x0.size
  C_, B_ = rig.valid.shape[0], rig.valid.shape[2]
  per = {"static": 6, "rolling": 12}.get(rig.cfg["motion"], 0)
  off = 6 * C_ + 6 * B_
  shared = np.ones(n, dtype=bool)
  shared[off:off + per * F] = False
  ns = int(shared.sum())
  say(f"dry run: frames [{shard[0]}, {shard[1]}) of {F}, {hm.m // 2} observations, n = {n}, n_s = {ns}, backend gloo")

  def barrier():
    if world > 1:
      dist.barrier()

  sizes = []

  def step():
    H, g, cost = hm.normal_equations(x0)
    msg = np.concatenate([g[shared], np.diag(H)[shared], [cost, hm.m // 2], np.zeros(4)])
    if world > 1:
      t = torch.from_numpy(msg)
      dist.all_reduce(t)
      sizes.append(int(t.numel()))
    return g, np.diag(H).copy(), msg

  watchdog = None
  if world > 1:
    import threading
    deadline_s = float(os.environ.get("MCBA_BENCH_DEADLINE_S", "420"))

    def expire():
      say(f"WATCHDOG: the dry run did not finish within {deadline_s:.0f} s")
      os._exit(7)
    watchdog = threading.Timer(deadline_s, expire)
    watchdog.daemon = True
    watchdog.start()
  for _ in range(args.warmup):
    step()
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    g_loc, diag_loc, msg = step()
  barrier()
  dt = time.perf_counter() - t0
  if world > 1:
    tmax = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
  counts = [None] * world
  if world > 1:
    dist.all_gather_object(counts, hm.m // 2)
  else:
    counts = [hm.m // 2]
  if watchdog is not None:
    watchdog.cancel()
  if rank == 0:
    whole = HostMath(calib)
    H, g, cost = whole.normal_equations(x0)
    ref = np.concatenate([g[shared], np.diag(H)[shared], [cost, whole.m // 2]])
    err = float(np.abs(msg[:2 * ns + 2] - ref).max() / np.abs(ref).max())
    print(json.dumps(dict(metric="residual+Jacobian evals/sec", value=args.steps / dt, unit="evals/s", n_gpus=world, steps=args.steps,
                          warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="strong", vs_baseline=None,
                          dtype="f64", data="synthetic", dry_run=True,
                          config=dict(workload=f"DRY RUN (no GPU): {RIGS[args.config]['label']} cut to {F} frames; the CPU stand-in of the step is "
                                               "test infrastructure and its timing means nothing",
                                      parallelism=f"frame-sharded x{world} ({[b - a for a, b in shards]} frames per rank), gloo all-reduce",
                                      n_params=int(n), n_shared=ns, observations_per_rank=counts),
                          step_collectives=dict(allreduce_calls_per_step=1 if world > 1 else 0, message_doubles=sorted(set(sizes))),
                          reduced_message_rel_error=err)), flush=True)
  if world > 1:
    dist.destroy_process_group()
  return 0


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
  ap.add_argument("--config", choices=sorted(RIGS), default="cfg3", help="cfg3 = BASELINE configs[2] (north star), cfg4 = configs[3]")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--cpu-baseline-full", action="store_true",
                  help="cpu_baseline: also run ONE complete finite-difference Jacobian of the oracle port at the full rig (several minutes)")
  ap.add_argument("--no-solve", action="store_true")
  ap.add_argument("--solve-repeats", type=int, default=5, help="default-tolerance solves; the median is reported")
  ap.add_argument("--no-scipy-mode", action="store_true")
  ap.add_argument("--no-lsmr-mode", action="store_true", help="skip `parity_route` (the solver = \"lsmr\" solve of the rig)")
  ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the weak-scaling measurement")
  ap.add_argument("--require-native-rccl", action="store_true",
                  help="N > 1: exit non-zero when the library's own RCCL communicator cannot be used (default: fall back to the "
                       "torch.distributed hook, loudly, and say so in the JSON line)")
  ap.add_argument("--scipy-frames", type=int, default=20, help="frames of the sample the scipy-driven product mode is timed on")
  ap.add_argument("--dry-run", action="store_true",
                  help="no GPU: run the N-rank plumbing (launcher, gloo rendezvous, shard plan incl. an empty shard, the step's collective with "
                       "its real message, barrier + MAX timing, watchdog, JSON line) with the shard evaluated by tests/hostmath on the CPU")
  ap.add_argument("--dry-frames", type=int, default=14, help="--dry-run: frames of the (cut-down) rig")
  args = ap.parse_args()
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(self_launch(args))
  if args.dry_run:
    sys.exit(dry_run(args))

  import torch
  import torch.distributed as dist
  from multical_amd import synthetic, calibration
  from multical_amd.backend import Handle, lower
  from multical_amd import distributed as mdist

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
  say = lambda msg: sys.stderr.write(f"[bench rank {rank}/{world}] {msg}\n")
  if not torch.cuda.is_available():
    # the product is HIP-only: there is no CPU path to fall back to, and the bench must not pretend otherwise
    say("no HIP device visible: the mcba back-end is GPU-only")
    sys.exit(3)
  # MCBA_BENCH_BACKEND=gloo is a test hook: it lets the N > 1 code path run with several ranks on ONE GPU (all-reduces
  # staged through the host); the measured configuration is always nccl (= RCCL over xGMI), one GPU per rank
  backend = os.environ.get("MCBA_BENCH_BACKEND", "nccl")
  n_dev = torch.cuda.device_count()
  if world > 1 and backend == "nccl" and n_dev < world:
    say(f"{world} ranks but only {n_dev} visible GPU(s): ranks would SHARE a device -- refusing to measure that")
    sys.exit(4)
  device_index = local_rank % max(n_dev, 1) if backend != "nccl" else local_rank
  torch.cuda.set_device(device_index)
  if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
      dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
    else:
      dist.init_process_group(backend)
  if world > 1 and backend == "nccl":     # one GPU per rank, checked: every rank reports the PCI bus id of its device
    ids = [None] * world
    dist.all_gather_object(ids, torch.cuda.get_device_properties(device_index).pci_bus_id
                           if hasattr(torch.cuda.get_device_properties(device_index), "pci_bus_id") else f"index{device_index}")
    if len(set(ids)) != world:
      say(f"ranks share a device: {ids}")
      sys.exit(4)

  rigdef = RIGS[args.config]
  F_rig = rigdef["frames"]
  tstream = torch.cuda.Stream()    # kernels and RCCL collectives share this (non-default) stream
  rccl_version = Handle.rccl_version()
  native_flags = []

  def make_handle(frames_total, shard):
    """this rank's frame shard [shard) of one rig of `frames_total` frames (a single rank owns the whole rig)"""
    rig_ = synthetic.make_rig(args.config, frames=frames_total, obs_frames=shard if world > 1 else None)
    calib_ = calibration.from_rig(rig_)
    h_ = Handle(lower(calib_), frame_range=shard if world > 1 else None, stream=tstream.cuda_stream)
    native_ = False
    if world > 1:
      # reductions: the library's own RCCL communicator (in-place ncclAllReduce on the handle's stream); the
      # torch.distributed hook is the fallback (gloo test hook, or if the native initialisation fails on any rank)
      if backend == "nccl" and os.environ.get("MCBA_NO_NATIVE_RCCL", "0") != "1":
        native_ = mdist.init_native_allreduce(h_, rank, world)
        if not native_:
          say("NATIVE RCCL INITIALISATION FAILED -- falling back to the torch.distributed all-reduce hook "
              "(~50 us of host time per reduction); the JSON line says so")
          if args.require_native_rccl:
            sys.exit(5)
      if not native_:
        h_.set_allreduce(mdist.make_allreduce_hook(stream=tstream))
      h_.set_shard_rank(rank, world)
    native_flags.append(native_)
    return rig_, calib_, h_, native_

  # ---- the workload: the FIXED rig of the north star; N > 1: this rank's frames of it ---------------------------
  shards = mdist.frame_shards(F_rig, world)
  shard = shards[rank]
  rig, calib, h, native = make_handle(F_rig, shard)
  x0 = calib.param_vec
  say(f"device {device_index} ({h.device_info().split(':')[0]}), frames [{shard[0]}, {shard[1]}) of {F_rig}, "
      f"native_rccl={'true' if native else 'false'}, rccl={rccl_version}, backend={backend if world > 1 else 'none'}")
  n_slots_local = int(np.prod(rig.valid.shape[0:1] + (shard[1] - shard[0],) + rig.valid.shape[2:]))
  n_obs_local = h.n_residuals // 2
  if world > 1:
    tot = torch.tensor([n_slots_local, n_obs_local], dtype=torch.float64, device="cuda")
    dist.all_reduce(tot)
    n_slots, n_obs = int(tot[0].item()), int(tot[1].item())
  else:
    n_slots, n_obs = n_slots_local, n_obs_local

  def barrier():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  # ---- timed region: K fused residual+Jacobian evaluations ------------------------------------------------------
  import ctypes as C
  from multical_amd.backend import make_options, _ptr, check
  xbuf = np.ascontiguousarray(x0)
  opt = make_options()
  cost = C.c_double()

  # x goes up once (and the host-boundary entry point is exercised once); inside the timed region everything is resident
  # in HBM -- x, tables, records, normal equations -- exactly as in the solver's own evaluations
  check(h.lib.mcba_normal_equations(h.h, _ptr(xbuf, C.c_double), C.byref(opt), C.byref(cost), None, None))

  def step():
    # tables + k_linearize + assembly (+ all-reduce), enqueued on the handle's stream; no host transfer
    check(h.lib.mcba_normal_equations_device(h.h, C.byref(opt)))

  def step_host():
    # the same evaluation through the host boundary: x upload (49 KB), cost download, one synchronisation per call
    check(h.lib.mcba_normal_equations(h.h, _ptr(xbuf, C.c_double), C.byref(opt), C.byref(cost), None, None))

  def timed(fn, all_times=None):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks.
    The timed region is repeated --repeats times (K steps each, nothing between them but the barrier) and the MEDIAN
    region is reported: at ~0.07 ms per step a single 20-step window is 1.5 ms of wall clock, too short for one sample."""
    for _ in range(args.warmup):
      fn()
    times = []
    for _ in range(max(1, args.repeats)):
      barrier()
      t0 = time.perf_counter()
      for _ in range(args.steps):
        fn()
      barrier()
      dt = time.perf_counter() - t0
      if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
      times.append(dt)
    if all_times is not None:
      all_times.extend(times)
    return sorted(times)[len(times) // 2]

  region_times = []
  dt = timed(step, region_times)
  ms_per_step = dt / args.steps * 1e3
  value = args.steps / dt                               # evaluations per second of the ONE fixed rig, whole job

  # ---- everything below adds detail to the line; `value` above is the contract's number -------------------------------------
  # N > 1 runs code that has never executed on more than one GPU (native RCCL with several ranks, the frame-sharded solvers over
  # xGMI): a WATCHDOG makes sure a hang there cannot cost the measurement -- past the deadline rank 0 prints the line with the
  # sections that did finish (marked "incomplete") and every rank leaves.
  weak = parity_route = step_comm = scipy_mode = roofline = host_ms_per_step = None
  extra = {}
  n_flops_obs = [None]

  def assemble(incomplete=None):
    C_, B_ = rig.valid.shape[0], rig.valid.shape[2]
    par = "single GPU"
    if world > 1:
      par = (f"ONE {C_} x {F_rig} x {B_} rig frame-sharded x{world} ({[b - a for a, b in shards]} frames per GPU), " +
             ("native RCCL all-reduce (librccl %d)" % rccl_version if native else f"torch.distributed ({backend}) all-reduce hook"))
    o = dict(metric="residual+Jacobian evals/sec", value=value, unit="evals/s", n_gpus=world, steps=args.steps,
             warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling="strong",
             vs_baseline=None, dtype="f64", data="synthetic",
             config=dict(workload=f"{rigdef['label']}; the whole rig on {world} GPU(s)",
                         n_params=int(h.n_params), n_slots=n_slots, n_observations=int(n_obs),
                         observation_fill=float(n_obs) / n_slots,   # evals/s is linear in observations, not in slots
                         parallelism=par, native_rccl=bool(native) if world > 1 else None, rccl_version=rccl_version,
                         device=h.device_info()),
             obs_per_s=value * float(n_obs),            # observations linearised per second, whole job
             timed_regions_ms=[t * 1e3 for t in region_times], repeats=len(region_times))
    if n_flops_obs[0] is not None:   # the STEP (all kernels of an evaluation + launch gaps + collective) against the FP64 peak of N GPUs
      o["step_roofline_frac"] = n_obs * n_flops_obs[0] / (ms_per_step * 1e-3) / 1e12 / (FP64_PEAK_TFLOPS * world)
    if host_ms_per_step is not None:
      o["host_boundary"] = dict(ms_per_step=host_ms_per_step, evals_per_s=1e3 / host_ms_per_step,
                                note="same evaluation with x uploaded and the cost downloaded on every call")
    if roofline is not None:
      o["roofline"] = roofline
    o.update(extra)
    for key, val in (("parity_route", parity_route), ("weak_scaling", weak), ("step_collectives", step_comm), ("scipy_mode", scipy_mode)):
      if val is not None:
        o[key] = val
    if incomplete:
      o["incomplete"] = incomplete
    return o

  watchdog = None
  if world > 1:
    import threading
    deadline_s = float(os.environ.get("MCBA_BENCH_DEADLINE_S", "420"))

    def expire():
      say(f"WATCHDOG: sections after the timed region did not finish within {deadline_s:.0f} s -- printing what exists and leaving")
      if rank == 0:
        print(json.dumps(assemble(f"watchdog after {deadline_s:.0f} s: a multi-GPU section hung; `value` and the listed sections are complete")),
              flush=True)
      os._exit(0 if rank == 0 else 7)
    watchdog = threading.Timer(deadline_s, expire)
    watchdog.daemon = True
    watchdog.start()

  # ---- weak scaling (N > 1): every rank owns a full F_rig-frame shard of one N-times-longer rig --------------------
  if world > 1 and not args.no_weak:
    wshard = (rank * F_rig, (rank + 1) * F_rig)
    wrig, wcalib, hw, wnative = make_handle(F_rig * world, wshard)
    wx = np.ascontiguousarray(wcalib.param_vec)
    check(hw.lib.mcba_normal_equations(hw.h, _ptr(wx, C.c_double), C.byref(opt), C.byref(cost), None, None))
    wdt = timed(lambda: check(hw.lib.mcba_normal_equations_device(hw.h, C.byref(opt))))
    weak = dict(value=world * args.steps / wdt, unit=f"shard evaluations/s (one {F_rig}-frame shard per GPU)", ms_per_step=wdt / args.steps * 1e3,
                frames_per_gpu=F_rig, scaling="weak", native_rccl=bool(wnative),
                note=f"{world} frame shards of one {rig.valid.shape[0]} x {F_rig * world} x {rig.valid.shape[2]} rig; every evaluation ends "
                     "with the all-reduce of the shared [g | diag | cost]")
    hw.close()
  # PCIe-inclusive variant (never `value`): every evaluation enters and leaves through the host boundary
  host_ms_per_step = timed(step_host) / args.steps * 1e3

  # ---- dominant kernel: k_linearize, HIP events on the handle's stream (this rank's shard) --------------------------
  lin_ms = h.time_linearize(x0, repeats=50)
  res_ms = h.time_residuals(x0, repeats=50)
  alg_bytes = 17 * n_slots_local + 16 * n_obs_local      # observed xy + mask per slot, residual per obs
  achieved = alg_bytes / (lin_ms * 1e-3) / 1e9
  d = h.problem
  NV = (12 if d.motion == 1 else 6) + (4 + d.n_dist if d.optimize & 8 else 0) + 1
  flops_per_obs = 2 * NV * (NV + 1) + 420 + (260 if d.motion == 1 else 0)
  n_flops_obs[0] = flops_per_obs
  alg_flops = n_obs_local * flops_per_obs
  # The fused pass is FP64-bound (~23 flop per algorithmic byte against a ridge of ~10 flop/B), so the
  # roofline that bounds k_linearize is the FP64 matrix/vector peak; the HBM view of the same launch is kept alongside.
  tflops = alg_flops / (lin_ms * 1e-3) / 1e12
  roofline = dict(bound="mfma", kernel="k_linearize", achieved=tflops, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                  frac=tflops / FP64_PEAK_TFLOPS, traffic=None, launch_ms=lin_ms, algorithmic_flops=alg_flops,
                  algorithmic_bytes=alg_bytes,
                  hbm=dict(achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS),
                  residual_kernel=dict(bound="hbm", launch_ms=res_ms, counter_bytes=None, achieved=None, frac=None,
                                       note="k_residual gathers the observations of the inlier slots only (mask bytes compacted "
                                            "per view), so its traffic is taken from the PMC counters of the committed profile "
                                            "(profiles/hbm_traffic.json: FETCH_SIZE x2 + WRITE_SIZE), not from the algorithmic "
                                            "17 B / slot + 16 B / observation, which it does not move"))
  # HBM bytes per launch from the PMC counters: they cannot be read inside an un-profiled run, so the figure of the
  # committed rocprofv3 passes of the same command is quoted and labelled as such (profiles/hbm_traffic.json: separate
  # --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction)
  traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
  if os.path.exists(traffic_file) and world == 1 and args.config == "cfg3":
    try:
      tj = json.load(open(traffic_file))
      roofline["traffic"] = tj.get("k_linearize_bytes_per_launch")
      rb = tj.get("k_residual_bytes_per_launch")
      if rb:
        rk = roofline["residual_kernel"]
        rk["counter_bytes"] = rb
        rk["achieved"] = rb / (res_ms * 1e-3) / 1e9
        rk["frac"] = rk["achieved"] / HBM_PEAK_GBS
      roofline["traffic_source"] = "profiles/hbm_traffic.json (" + str(tj.get("source", "rocprofv3 --pmc passes")) + "), not measured in this run"
    except Exception:
      pass

  def global_rms(hh, x):
    e, v = hh.reprojection_error(x)
    sq = torch.tensor([float((e[v] ** 2).sum()), float(v.sum())], dtype=torch.float64, device="cuda")
    if world > 1:
      dist.all_reduce(sq)
    return float(torch.sqrt(sq[0] / sq[1]).item())

  # ---- LM iterations/s and final RMS: one full bundle adjustment of the same problem (not part of `value`) --------
  ref_end = reference_endpoint(args.config)
  if not args.no_solve:
    # the default-tolerance solve of the rig with the EXACT-step solver (scipy's defaults: ftol 1e-4), repeated: median of the
    # repeats; and a LONG solve (tight tolerances, fixed number of trial steps from a perturbed start) whose per-trial-step time is
    # the steady-state figure
    solves = []
    for _ in range(max(1, args.solve_repeats)):
      barrier()
      t0 = time.perf_counter()
      res = h.solve(x0)
      barrier()
      solves.append((time.perf_counter() - t0, res.nfev))
    t_solve, nfev_med = sorted(solves)[len(solves) // 2]
    rng = np.random.default_rng(1)
    x1 = x0 + 1e-3 * rng.normal(size=x0.size)
    longs = []
    for _ in range(3):           # (median of three: one 0.25 ms hiccup in a 3 ms solve moved a single sample from 134 to 145 us)
      if world > 1:
        h.allreduce_stats(reset=True)
      barrier()
      t0 = time.perf_counter()
      lres = h.solve(x1, tolerance=1e-15, xtol=1e-15, gtol=1e-15, max_iterations=41)
      barrier()
      longs.append(time.perf_counter() - t0)
    t_long = sorted(longs)[1]
    native_rms = global_rms(h, res.x)
    extra = dict(lm_solver="native (exact Schur / Cholesky steps: the converged optimum, NOT the reference's end point -- see parity_route)",
                 lm_iters_per_s=(nfev_med - 1) / t_solve, lm_trial_steps=res.nfev - 1, lm_linearizations=res.njev,
                 solve_seconds=t_solve, solve_seconds_all=[t for t, _ in solves], solve_status=res.status,
                 lm_long_solve=dict(trial_steps=lres.nfev - 1, seconds=t_long, us_per_trial_step=t_long / max(lres.nfev - 1, 1) * 1e6,
                                    iters_per_s=max(lres.nfev - 1, 1) / t_long),
                 final_rms_px=native_rms, final_cost=res.cost, initial_cost=res.initial_cost)
    if world > 1:
      calls, doubles, sizes = h.allreduce_stats(reset=True)
      extra["lm_long_solve"]["allreduce"] = dict(calls=calls, bytes=8 * doubles, calls_per_trial_step=calls / max(lres.nfev - 1, 1),
                                                 bytes_per_trial_step=8 * doubles / max(lres.nfev - 1, 1),
                                                 message_doubles=sorted(set(abs(s) for s in sizes)))
  # collectives of ONE evaluation step (N > 1)
  if world > 1:
    h.allreduce_stats(reset=True)
    step()
    barrier()
    calls, doubles, sizes = h.allreduce_stats(reset=True)
    step_comm = dict(allreduce_calls=calls, allreduce_bytes=8 * doubles, message_doubles=[abs(s) for s in sizes])

  # ---- PARITY ROUTE: the solver that reproduces the reference's END POINT, on the whole rig (single GPU or frame-sharded) -----
  if not args.no_lsmr_mode and not args.no_solve:
    h.solve(x0, tr_solver="lsmr")                                   # warm-up (buffers, first-use allocations)
    runs = []
    for _ in range(3):
      barrier()
      t0 = time.perf_counter()
      pres = h.solve(x0, tr_solver="lsmr")
      barrier()
      runs.append(time.perf_counter() - t0)
    t_par = sorted(runs)[1]
    itn = h.lsmr_iterations()
    prms = global_rms(h, pres.x)
    parity_route = dict(solver="lsmr", seconds=t_par, seconds_all=runs, nfev=pres.nfev, njev=pres.njev, status=pres.status,
                        lm_iters_per_s=(pres.nfev - 1) / t_par, lsmr_iterations=itn, us_per_lsmr_iteration=t_par / max(itn, 1) * 1e6,
                        final_cost=pres.cost, final_rms_px=prms,
                        note="solver='lsmr' (the product default) on the whole rig: scipy's TRF driver + lsmr(J_h, f, damp) restated on the "
                             "device; us_per_lsmr_iteration = whole solve / LSMR iterations (linearisations and trial evaluations included)")
    if ref_end is not None:
      parity_route.update(reference_rms_px=ref_end["rms_px"], abs_delta_px=abs(prms - ref_end["rms_px"]),
                          reference_nfev=ref_end["nfev"], reference_status=ref_end["status"], reference_cost=ref_end["cost"],
                          reference_spread_px=ref_end.get("spread_px"), reference_spread_runs=ref_end.get("spread_runs"),
                          reference_seconds=ref_end["seconds"], speedup_vs_reference=ref_end["seconds"] / t_par,
                          within_tolerance=bool(abs(prms - ref_end["rms_px"]) <= max(1e-6, ref_end.get("spread_px") or 0.0)
                                                and pres.nfev == ref_end["nfev"] and pres.status == ref_end["status"]),
                          reference_source=ref_end["source"])
      if "final_rms_px" in extra:
        extra["native_abs_delta_px"] = abs(extra["final_rms_px"] - ref_end["rms_px"])
      if "exact_product_rms_px" in ref_end:
        # where scipy's OWN algorithm ends on the reference's residual function when its two sparse products are accumulated in 80-bit
        # precision: the device's products (tree reductions) are accurate to a few ulp and land there; the reference's single run sits
        # 1e-7 ... 2.5e-6 px above it (the footprint of sequential double accumulation in scipy.sparse: profiles/r06_lsmr_sign.md)
        parity_route.update(exact_product_rms_px=ref_end["exact_product_rms_px"],
                            exact_product_abs_delta_px=abs(prms - ref_end["exact_product_rms_px"]),
                            exact_product_source="tests/golden/exact_products.json (oracle/make_exact_products.py)")
      if "tight_rms_px" in ref_end:
        # SURVEY 7 protocol C at the stated size: the converged optimum of the reference's residual function, reached by the exact-step
        # solver at tight tolerance -- the sense in which the end point is defined beyond the reference's run-to-run spread
        cres = h.solve(x0, tolerance=1e-14, xtol=1e-14, gtol=1e-14, max_iterations=400)
        crms = global_rms(h, cres.x)
        parity_route.update(converged_rms_px=crms, converged_reference_rms_px=ref_end["tight_rms_px"],
                            converged_abs_delta_px=abs(crms - ref_end["tight_rms_px"]), converged_nfev=cres.nfev,
                            converged_cost_rel=cres.cost / ref_end["tight_cost"] - 1.0,
                            converged_source="tests/golden/%s_endpoint.npz: ba_tight_* (oracle/make_endpoint.py tight)" % args.config)
    # the product kernel of the route: J v and J^T u of one LSMR iteration
    parity_route["lsmr_iteration"] = dict(flops_per_observation=4 * (2 * (NV - 1)) + flops_per_obs - 2 * NV * (NV + 1),
                                          note="per observation and iteration: the analytic row pair (forward model + derivatives, "
                                               "as in k_linearize without the V^T V accumulation) + 2 x 2 x (NV - 1) multiply-adds for "
                                               "J v and J^T u")
    if world == 1:
      # the dominant kernel of THIS route, timed live with HIP events on the handle's stream like `roofline` above: the product kernel
      # of an LSMR iteration (both Jacobian products from one evaluation of the analytic rows) + the gather behind it
      f_ms, g_ms = h.time_lsmr_iteration(x0, repeats=50)
      fl_it = parity_route["lsmr_iteration"]["flops_per_observation"] * n_obs_local
      by_it = 48 * n_obs_local + n_slots_local        # observation in, uhat in + out (16 B each) per observation, mask byte per slot
      parity_route["roofline"] = dict(bound="fp64-valu", kernel="k_lsmr_fused2", achieved=fl_it / (f_ms * 1e-3) / 1e12, peak=FP64_PEAK_TFLOPS,
                                      unit="TFLOP/s", frac=fl_it / (f_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, launch_ms=f_ms,
                                      gather_launch_ms=g_ms, algorithmic_flops=fl_it, algorithmic_bytes=by_it,
                                      hbm=dict(achieved=by_it / (f_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                               frac=by_it / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
                                      traffic=None)
      pmc_file = os.path.join(ROOT, "profiles", "r06_lsmr_pmc.json")     # (PMC counters cannot be read inside an un-profiled run)
      if not os.path.exists(pmc_file):
        pmc_file = os.path.join(ROOT, "profiles", "r05_lsmr_pmc.json")
      if os.path.exists(pmc_file) and args.config == "cfg3":
        try:
          pj = json.load(open(pmc_file))
          k2 = sorted([k for k in pj if "k_lsmr_fused2" in k], key=lambda k: not k.rstrip().endswith(", 3>"))   # (MODE 3: the default form)
          if k2 and "FETCH_SIZE" in pj[k2[0]] and "WRITE_SIZE" in pj[k2[0]]:
            parity_route["roofline"]["traffic"] = 2 * pj[k2[0]]["FETCH_SIZE"] * 1024 + pj[k2[0]]["WRITE_SIZE"] * 1024
            parity_route["roofline"]["traffic_source"] = (os.path.relpath(pmc_file, ROOT) + ": separate --pmc FETCH_SIZE / WRITE_SIZE passes "
                                                          "(profiles/scripts/collect_r0*.sh), gfx950 x2 read correction; not measured in this run")
            parity_route["roofline"]["traffic_note"] = ("the compacted tables STREAM the board point (24 B) and the scan time (8 B) of every observation "
                                                        "instead of gathering board points through L2 behind mask bytes and a compaction: ~64 B per observation "
                                                        "and iteration + 2.3 KB of That per view, by design above the 48 B + 1 B / slot of `algorithmic_bytes` "
                                                        "-- coalesced bytes bought fewer dependent round trips per view (26.9 -> 24.3 us); at ~2.9 TB/s the kernel "
                                                        "is not HBM-bound")
        except Exception:
          pass
    fl = parity_route["lsmr_iteration"]["flops_per_observation"] * n_obs_local
    parity_route["lsmr_iteration"].update(us=parity_route["us_per_lsmr_iteration"],
                                          fp64_frac=fl / (parity_route["us_per_lsmr_iteration"] * 1e-6) / 1e12 / FP64_PEAK_TFLOPS)

  # ---- the reference's own solver on the device functions (product mode solver="scipy"), beside the native solve -----
  if world == 1 and not args.no_scipy_mode and args.config == "cfg3":
    srig = synthetic.make_rig("cfg3", frames=args.scipy_frames)
    sc = calibration.from_rig(srig)
    with Handle(lower(sc)) as hsm:
      sx0 = sc.param_vec
      hsm.solve(sx0)
      t0 = time.perf_counter(); nres = hsm.solve(sx0); t_nat = time.perf_counter() - t0
      t0 = time.perf_counter(); sres = hsm.solve_scipy(sx0, verbose=0); t_sci = time.perf_counter() - t0
      hsm.solve(sx0, tr_solver="lsmr")
      t0 = time.perf_counter(); mres = hsm.solve(sx0, tr_solver="lsmr"); t_lsm = time.perf_counter() - t0

      def rms_at(x):
        e_, v_ = hsm.reprojection_error(x)
        return float(np.sqrt(np.mean(e_[v_] ** 2)))
      scipy_mode = dict(sample=f"first {args.scipy_frames} of {F_rig} frames of the same rig ({hsm.n_residuals} residuals, "
                               f"{hsm.n_params} parameters)",
                        scipy_mode_seconds=t_sci, scipy_mode_nfev=sres.nfev, scipy_mode_rms_px=rms_at(sres.x),
                        lsmr_mode_seconds=t_lsm, lsmr_mode_nfev=mres.nfev, lsmr_mode_rms_px=rms_at(mres.x),
                        native_seconds=t_nat, native_nfev=nres.nfev, native_rms_px=rms_at(nres.x),
                        note="scipy mode = scipy.optimize.least_squares(method='trf', x_scale='jac') exactly as "
                             "optimization/calibration.py:209-210 on mcba_residuals + mcba_jacobian: the reference's end point "
                             "(profiles/parity_table.md), scipy's LSMR on the host")

  if watchdog is not None:
    watchdog.cancel()
  out = None
  if rank == 0:
    out = assemble()
    if world == 1 and not args.no_cpu_baseline and args.config == "cfg3":
      out["cpu_baseline"] = cpu_baseline(full_jacobian=args.cpu_baseline_full)
      if ref_end is not None:
        # the REAL reference, run to completion on this rig (build container): njev Jacobians (= residual+Jacobian evaluations by
        # SURVEY 8(d)'s counting convention) and nfev - 1 trial steps in `seconds` of one core
        out["cpu_baseline"]["reference_measured"] = dict(
          seconds=ref_end["seconds"], nfev=ref_end["nfev"], njev=ref_end["njev"], evaluate_calls=ref_end["evaluate_calls"],
          evals_per_s=ref_end["njev"] / ref_end["seconds"], lm_iters_per_s=(ref_end["nfev"] - 1) / ref_end["seconds"],
          peak_rss_gb=ref_end["peak_rss_gb"], final_rms_px=ref_end["rms_px"], cores=1, kind="reference", host=ref_end["host"],
          source=ref_end["source"])
    print(json.dumps(out), flush=True)
  h.close()
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
