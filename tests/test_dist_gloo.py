"""CPU tier: the N>1 path with world_size 2 over gloo.  Each rank owns a contiguous env-id shard; the
only collective is the end-of-batch all-reduce of the episodic-return accumulators.  (No GPU here:
the per-shard compute is done by the oracle, which is what the HIP path is parity-tested against.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TOTAL, T, SEED = 192, 60, 0x5AFE


def _returns(lo, hi):
  from ai_safety_gridworlds_amd import philox
  from oracle import oracle as O
  acts = philox.actions(SEED, np.arange(lo, hi), np.arange(T), 0, 5).T.copy()
  out = O.run_streams(O.make_config("island_navigation_ex", level=9), acts, fields=["step_type", "cumulative"])
  last = out["step_type"] == 2
  acc = np.zeros(out["K"] + 1)
  acc[:-1] = (out["cumulative"] * last[..., None]).sum(axis=(0, 1))
  acc[-1] = last.sum()
  return acc


def _worker(rank, world_size, port, q):
  sys.path.insert(0, REPO)
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size),
                    MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  from ai_safety_gridworlds_amd import parallel
  dist = parallel.init("gloo")
  lo, hi = parallel.shard_range(N_TOTAL, rank, world_size)
  acc = torch.from_numpy(_returns(lo, hi))
  parallel.allreduce_returns(acc, dist)
  tmax = parallel.max_over_ranks(float(rank + 1), torch.device("cpu"), dist)
  dist.barrier()
  q.put((rank, acc.numpy().tolist(), tmax, (lo, hi)))
  dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_process():
  s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  results = [q.get(timeout=120) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  want = _returns(0, N_TOTAL)
  spans = sorted(r[3] for r in results)
  assert spans == [(0, 96), (96, 192)]
  for rank, acc, tmax, _ in results:
    assert np.array_equal(np.array(acc), want), rank      # integer-valued sums: exact, order-free
    assert tmax == 2.0
  assert want[-1] > 0


def test_single_process_helpers_are_noops():
  from ai_safety_gridworlds_amd import parallel
  t = torch.ones(3, dtype=torch.float64)
  assert parallel.allreduce_returns(t, None) is t
  assert parallel.max_over_ranks(1.5, torch.device("cpu"), None) == 1.5


def _world8_worker(rank, world, port, out):
  """One rank of the 8-way rehearsal: bench.mixed_parts for this rank, a stand-in accumulator per family whose value is the
  sum of the global env ids the rank owns (so the all-reduced total identifies any hole or overlap), parallel.allreduce_returns."""
  import os
  import torch
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  import bench as B
  from ai_safety_gridworlds_amd import parallel
  dist = parallel.init("gloo")
  per_gpu = 32768
  parts = B.mixed_parts(rank, world, per_gpu)
  res = {}
  for fam in B.MIXED:
    acc = torch.zeros(3, dtype=torch.float64)
    for f, cnt, base in parts:
      if f == fam:
        ids = torch.arange(base, base + cnt, dtype=torch.float64)
        acc += torch.stack([ids.sum(), (ids * ids).sum(), torch.tensor(float(cnt), dtype=torch.float64)])
    parallel.allreduce_returns(acc, dist)
    res[fam] = acc.tolist()
  t = parallel.max_over_ranks(float(rank), torch.device("cpu"), dist)
  if rank == 0:
    out.put((res, t, [(f, c, b) for f, c, b in parts]))
  dist.destroy_process_group()


def test_eight_ranks_cover_the_mixed_suite_once_and_allreduce():
  """CPU tier, world size 8 (gloo): the 262 144-env mixed suite of BASELINE config 5 is cut into per-rank parts by
  bench.mixed_parts; the all-reduced per-family (sum of ids, sum of squares, count) equals what the three contiguous global id
  ranges give -- every env id owned exactly once -- and max_over_ranks sees the last rank."""
  import multiprocessing as mp
  import socket
  ctx = mp.get_context("spawn")
  with socket.socket() as s_:
    s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
  out = ctx.Queue()
  procs = [ctx.Process(target=_world8_worker, args=(r, 8, port, out)) for r in range(8)]
  for p in procs: p.start()
  res, t, parts0 = out.get(timeout=240)
  for p in procs:
    p.join(timeout=120)
    assert p.exitcode == 0
  import bench as B
  from ai_safety_gridworlds_amd import parallel
  total = 8 * 32768
  assert t == 7.0
  for i, fam in enumerate(B.MIXED):
    lo, hi = parallel.shard_range(total, i, 3)
    ids = np.arange(lo, hi, dtype=np.float64)
    assert res[fam] == [ids.sum(), (ids * ids).sum(), float(hi - lo)], fam
  assert parts0[0][0] == "island_navigation_ex" and parts0[0][2] == 0 and sum(c for _, c, _ in parts0) == 32768
