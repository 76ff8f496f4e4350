"""sgw_step_n replays its T launches as a captured hipGraph from the second call with the same device buffers on
(csrc/sgw_api.hip sgw_step_n).  The replayed launches must be the same launches: outputs, state and episodic returns of an
engine stepped by repeated step_n calls on ONE actions buffer refilled in place == an engine stepped one sgw_step at a time,
on the default stream and on a side stream, with write_every outputs, and after a setter invalidated the captures."""
import numpy as np
import pytest
import torch

from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OUTS = ("board", "reward", "step_type", "term_reason", "cumulative")


def make(name, n, **kw):
  spec = make_spec(name, **kw)
  eng = BatchedEngine(spec, n, device=DEV, outputs=OUTS)
  if name == "safe_interruptibility":
    eng.set_episode_bits(None, seed=5)
  eng.reset()
  return eng


@pytest.mark.parametrize("name,kw,side", [("island_navigation_ex", {}, False), ("island_navigation_ex", {}, True),
                                           ("safe_interruptibility", dict(level=1), True), ("boat_race_ex", dict(level=3), False)])
def test_step_n_graph_replay_equals_single_steps(name, kw, side):
  n, T, calls = 3000, 24, 5
  a, b = make(name, n, **kw), make(name, n, **kw)
  all_acts = a.fill_actions(T * calls, 11).clone()
  buf = torch.empty_like(all_acts[:T])
  stream = torch.cuda.Stream(DEV) if side else torch.cuda.current_stream(DEV)
  torch.cuda.synchronize()
  with torch.cuda.stream(stream):
    for c in range(calls):                      # call 0: direct launches, call 1: capture + replay, calls 2..: replay
      buf.copy_(all_acts[c * T:(c + 1) * T])
      got = {k: v.clone() for k, v in a.step_n(buf, accumulate=True).items()}
      if c == 3 and name == "safe_interruptibility":
        a.set_episode_bits(None, seed=5)        # same values: drops the captures, the next call launches directly again
    ra = a.read_returns().clone()
  for t in range(T * calls):
    want = b.step_n(all_acts[t:t + 1], accumulate=True)
  rb = b.read_returns()
  torch.cuda.synchronize()
  for k in OUTS:
    assert torch.equal(got[k], want[k]), k
  assert torch.equal(a.get_state(), b.get_state())
  assert torch.equal(ra, rb) and float(rb[-1]) > 0
  a.close(); b.close()


def test_step_n_graph_write_every():
  n, T = 1000, 16
  a, b = make("island_navigation_ex", n), make("island_navigation_ex", n)
  acts = a.fill_actions(T * 3, 3).clone()
  buf = torch.empty_like(acts[:T])
  for c in range(3):
    buf.copy_(acts[c * T:(c + 1) * T])
    got = {k: v.clone() for k, v in a.step_n(buf, write_every=True).items()}
  for t in range(2 * T):
    b.step(acts[t])
  for t in range(T):
    want = b.step(acts[2 * T + t])
    for k in OUTS:
      assert torch.equal(got[k][t], want[k]), (k, t)
  a.close(); b.close()


@pytest.mark.parametrize("name,kw", [("island_navigation_ex", {}), ("boat_race_ex", dict(level=3)), ("safe_interruptibility", dict(level=1)),
                                     ("island_navigation_ex_ma", {}), ("firemaker_ex_ma", dict(amount_agents=3))])
def test_replay_of_an_action_tape_equals_step_n(name, kw):
  """sgw_replay (one fused launch over the caller's [T, N, A] actions) == sgw_step_n over the same buffer: every output of every
  step, the final state and the episodic returns."""
  n, T = 2000, 40
  spec = make_spec(name, **kw)
  outs = OUTS if spec.A == 1 else ("board", "reward", "step_type", "term_reason", "cumulative")
  def mk():
    e = BatchedEngine(spec, n, device=DEV, outputs=outs)
    if name == "safe_interruptibility":
      e.set_episode_bits(None, seed=5)
    if getattr(spec, "needs_rng", False) or name == "firemaker_ex_ma":
      e.set_rng_seeds(np.arange(n) + 9)
    e.reset()
    return e
  a, b = mk(), mk()
  acts = a.fill_actions(T, 21)
  got = {k: v.clone() for k, v in a.replay(acts, write_every=True, accumulate=True).items()}
  want = {k: v.clone() for k, v in b.step_n(acts, write_every=True, accumulate=True).items()}
  torch.cuda.synchronize()
  for k in outs:
    assert torch.equal(got[k], want[k]), k
  assert torch.equal(a.get_state(), b.get_state())
  assert torch.equal(a.read_returns(), b.read_returns())
  a.close(); b.close()
