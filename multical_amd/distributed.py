"""Frame sharding across GPUs: one process per GPU, torch.distributed for the cross-rank sums.

Every observation belongs to exactly one frame, the per-frame pose blocks are private to the shard that owns the
frame, and everything else is a sum -- so a rank keeps only its contiguous frame range of the observation table on
its GPU and the solver exchanges just the reduced quantities per iteration:

    shared entries of [g | diag] + cost + step norms    2 n_s + 6 doubles   after every linearisation
    norms + Cauchy curvature, per rank                  4 W doubles         once per trial point (speculatively, right behind
                                                                            the linearisation's message: a rejected step wastes it)
    Schur complement + right-hand side                  n_s^2 + n_s         once per iteration
    dots of the 2-D subspace, per rank (+ pivot flag)   3 W + 1 doubles     once per iteration
    trial cost + step norms                             4 doubles           per RETRY after a rejected step only
    eliminated-frame part of x                          n_motion            once per SOLVE (the complete x every rank returns)

(W ranks, n_s shared parameters.)  No message grows with the number of frames: the frame entries of every n-vector live on
the rank that owns the frame, sums over all parameters are all-reduced per-rank partial sums ("gather by summation": rank r
fills block r of a zeroed buffer).

The C library calls back into `allreduce_hook` with DEVICE pointers (include/mcba.h: mcba_allreduce_fn); with the
"nccl" backend (RCCL over xGMI on ROCm) the reduction runs in place on the handle's stream -- which is torch's
current stream, so no host synchronisation is involved -- and with "gloo" (CPU-side tests) the buffer is staged
through host memory.
"""
import os

import numpy as np


def frame_shards(n_frames, world_size, weights=None):
  """Contiguous frame ranges [(f0, f1)] * world_size, balanced by `weights` (e.g. inlier count per frame)."""
  if weights is None:
    weights = np.ones(n_frames)
  w = np.asarray(weights, dtype=np.float64)
  assert w.shape == (n_frames,)
  total = w.sum()
  if total <= 0:
    w = np.ones(n_frames)
    total = float(n_frames)
  cum = np.concatenate([[0.0], np.cumsum(w)])
  bounds = [0]
  for r in range(1, world_size):
    target = total * r / world_size
    f = int(np.searchsorted(cum, target, side='left'))
    f = min(max(f, bounds[-1]), n_frames)
    bounds.append(f)
  bounds.append(n_frames)
  return [(bounds[i], bounds[i + 1]) for i in range(world_size)]


class _DevArray(object):
  """Expose a raw device pointer of `count` doubles through __cuda_array_interface__ (zero-copy torch view)."""

  def __init__(self, ptr, count):
    self.__cuda_array_interface__ = dict(shape=(int(count),), typestr='<f8', data=(int(ptr), False), version=2,
                                         strides=None)


def make_allreduce_hook(group=None, device=None, stream=None):
  """Returns fn(ptr, count, op, stream) for Handle.set_allreduce, backed by torch.distributed.

  `stream` is the torch.cuda.Stream the handle was created on: the collective is enqueued inside
  `torch.cuda.stream(stream)`, i.e. ordered after the kernels that produced the buffer and before the ones that
  consume it, without a host synchronisation (nccl) -- no reliance on legacy default-stream semantics."""
  import contextlib
  import torch
  import torch.distributed as dist
  backend = dist.get_backend(group)
  dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())

  def hook(ptr, count, op, raw_stream):
    ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
    with ctx:
      t = torch.as_tensor(_DevArray(ptr, count), device=dev)
      rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX
      if backend == "nccl":
        dist.all_reduce(t, op=rop, group=group)        # RCCL over xGMI, in place
      else:
        host = t.cpu()                                 # gloo: stage through host memory
        dist.all_reduce(host, op=rop, group=group)
        t.copy_(host)
    return 0

  return hook


def init_native_allreduce(h, rank, world_size, group=None):
  """Switches handle `h` to the library's own RCCL communicator (collective over the group).

  torch.distributed only carries the 128-byte unique id and the success flags; the reductions themselves are
  ncclAllReduce calls issued by the C library on the handle's stream (no Python callback, no torch dispatch in the
  inner loop -- the callback path costs ~50 us of host time per reduction).  Returns True when EVERY rank succeeded;
  otherwise all ranks leave the native path again and the caller falls back to the torch.distributed hook."""
  import torch
  import torch.distributed as dist
  from .backend import Handle
  ok = 1
  payload = [None]
  if rank == 0:
    try:
      payload[0] = Handle.rccl_unique_id()
    except Exception:
      payload[0] = None
  dist.broadcast_object_list(payload, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
  if payload[0] is None:
    ok = 0
  else:
    try:
      h.rccl_init(payload[0], rank, world_size)
    except Exception:
      ok = 0
  flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
  dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
  if int(flag.item()) != 1:
    h.rccl_shutdown()
    return False
  return True


def sharded_handle(calib, rank=None, world_size=None, group=None, balance=True, native=None, shards=None):
  """Handle owning this rank's frame shard.  Reductions: the library's own RCCL communicator (`native`, default when the
  group's backend is nccl, i.e. one GPU per rank) or a torch.distributed hook (gloo tests, fallback).
  `shards`: explicit [(f0, f1)] * world_size instead of the balanced plan (empty ranges are legal)."""
  import torch
  import torch.distributed as dist
  from .backend import Handle, lower
  rank = dist.get_rank(group) if rank is None else rank
  world_size = dist.get_world_size(group) if world_size is None else world_size
  prob = lower(calib)
  weights = None
  if balance:
    inl = calib.inliers
    weights = inl.sum(axis=(0, 2, 3)).astype(np.float64)
  if shards is None:
    shards = frame_shards(prob.shape[1], world_size, weights)
  assert len(shards) == world_size and shards[0][0] == 0 and shards[-1][1] == prob.shape[1]
  tstream = torch.cuda.Stream()                       # dedicated (non-default) stream shared by kernels and collectives
  h = Handle(prob, frame_range=shards[rank], stream=tstream.cuda_stream)
  h.torch_stream = tstream                            # keep it alive as long as the handle
  if world_size > 1:
    if native is None:
      native = dist.get_backend(group) == "nccl" and os.environ.get("MCBA_NO_NATIVE_RCCL", "0") != "1"
    h.native_allreduce = bool(native) and init_native_allreduce(h, rank, world_size, group)
    if not h.native_allreduce:
      h.set_allreduce(make_allreduce_hook(group, stream=tstream))
    h.set_shard_rank(rank, world_size)
  h.frame_range = shards[rank]
  return h
