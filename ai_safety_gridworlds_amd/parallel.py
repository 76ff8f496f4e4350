"""Multi-GPU plumbing: one process per GPU, envs sharded by contiguous global env-id ranges.

Env instances are independent, so the data path has NO collective: each rank steps
its own shard.  The counter-based action stream is keyed by the GLOBAL env id, so results do not
depend on the number of GPUs.  The one collective is the end-of-batch all-reduce (RCCL over xGMI
when the backend is "nccl"; gloo in CPU tests) of the episodic-return accumulators
[sum of episode return vectors, #episodes] -- a few doubles, latency-bound, once per batch.
"""
import os

import torch


def world():
  """(rank, local_rank, world_size) from the torchrun environment (1 process => (0, 0, 1))."""
  return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
          int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_total, rank, world_size):
  """Contiguous [lo, hi) global env ids of `rank`; the first n_total % world ranks get one more."""
  base, extra = divmod(int(n_total), int(world_size))
  lo = rank * base + min(rank, extra)
  return lo, lo + base + (1 if rank < extra else 0)


def init(backend=None):
  """Initialise torch.distributed when launched with WORLD_SIZE > 1.  Returns the dist module or None."""
  rank, local_rank, ws = world()
  if ws <= 1:
    return None
  import torch.distributed as dist
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  os.environ.setdefault("MASTER_PORT", "29500")
  os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  backend = os.environ.get("SGW_DIST_BACKEND", backend)      # rehearsal override (e.g. gloo on a 1-GPU box)
  if backend is None:
    backend = "nccl" if torch.cuda.is_available() else "gloo"
  if not dist.is_initialized():
    if backend == "nccl":
      torch.cuda.set_device(local_device(local_rank))
      dist.init_process_group("nccl", device_id=local_device(local_rank))
    else:
      dist.init_process_group(backend)
  return dist


def allreduce_returns(accum, dist=None):
  """Sum the [A*K + 1] episodic-return accumulators over all ranks (in place); returns the tensor."""
  if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
    if accum.is_cuda and dist.get_backend() == "gloo":       # gloo reduces host tensors
      host = accum.cpu()
      dist.all_reduce(host, op=dist.ReduceOp.SUM)
      accum.copy_(host)
    else:
      dist.all_reduce(accum, op=dist.ReduceOp.SUM)
  return accum


def local_device(local_rank):
  """cuda:<local_rank>, folded onto the visible devices (lets several ranks share one GPU in rehearsals)."""
  n = torch.cuda.device_count()
  return torch.device("cuda", local_rank % max(n, 1))


def max_over_ranks(value, device, dist=None):
  """max over ranks of a python float (the timed region's wall time)."""
  if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
    return float(value)
  t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())
