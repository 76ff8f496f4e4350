"""Host logic of bench.py (no GPU): the timed region must issue EXACTLY count x K steps per engine, batch j from action batch
j modulo the number resident, whatever grouping run_batches uses to hand contiguous batches to sgw_step_n in one call; and the
mixed suite's parts must cover the global env-id ranges exactly once over the ranks."""
import contextlib

import numpy as np
import pytest
import torch

import bench as B


class FakeActs(object):
  """Stands for the resident [steps, N] action tensor: slicing records (start, stop)."""
  def __init__(self, steps):
    self.shape = (steps, 8)
  def __getitem__(self, sl):
    return (sl.start, sl.stop)


class FakeEngine(object):
  def __init__(self):
    self.calls = []
  def step_n(self, span, accumulate=False):
    self.calls.append(span)


@pytest.mark.parametrize("K,nd,first,count", [(20, 204, 0, 3361), (2000, 2, 0, 41), (7, 5, 3, 23), (1, 1, 0, 10), (64, 64, 60, 200)])
def test_run_batches_issues_exactly_the_asked_steps(K, nd, first, count, monkeypatch):
  monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
  engines = [dict(eng=FakeEngine(), acts=FakeActs(K * nd), stream=None), dict(eng=FakeEngine(), acts=FakeActs(K * nd), stream=None)]
  B.run_batches(engines, K, first, count, True)
  for e in engines:
    steps, j = 0, first
    for (lo, hi) in e["eng"].calls:
      assert 0 <= lo < hi <= K * nd and lo % K == 0 and (hi - lo) % K == 0
      assert lo == (j % nd) * K                                  # batch j comes from action batch j modulo nd, in order
      g = (hi - lo) // K
      assert g * K <= max(K, 2000)                               # groups stay near 2 000 steps
      j += g; steps += hi - lo
    assert steps == count * K and j == first + count
  assert [c for c in engines[0]["eng"].calls] == [c for c in engines[1]["eng"].calls]      # batch-major: the same spans per family


@pytest.mark.parametrize("world", [1, 2, 8])
def test_mixed_parts_cover_the_global_ranges_once(world):
  per_gpu = 32768
  seen = {}
  for rank in range(world):
    for fam, cnt, base in B.mixed_parts(rank, world, per_gpu):
      assert cnt > 0
      seen.setdefault(fam, []).append((base, base + cnt))
  total = 0
  for fam, spans in seen.items():
    spans.sort()
    assert spans[0][0] >= 0
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
      assert a1 <= b0                                            # no env id belongs to two ranks
    total += sum(hi - lo for lo, hi in spans)
  assert total == world * per_gpu and set(seen) == set(B.MIXED)
