"""sgw_out.done / obs_dir / act_dir (ABI 8): the wrappers' per-step decodes of step_type and agent_flags, written by the step launch
itself -- through every launch kind (sgw_step, sgw_step_n with write_every, the fused rollout incl. the pipelined two-wave form,
masked reset) they must equal what the wrappers computed from the primary outputs (gridworld_gym_env.py:563-578,
gridworld_zoo_parallel_env.py:325-335, 589-600)."""
import numpy as np
import pytest
import torch

from ai_safety_gridworlds_amd import _native as N
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.helpers.gridworld_gym_env import GridworldVectorEnv
from ai_safety_gridworlds_amd.specs import make_spec

pytestmark = pytest.mark.gpu

CASES = [
    ("island_navigation_ex", dict(max_iterations=25), False),
    ("boat_race_ex", dict(max_iterations=30), False),
    ("safe_interruptibility_ex", dict(max_iterations=20), False),
    ("island_navigation_ex_ma", dict(max_iterations=40), True),
    ("island_navigation_ex_ma", dict(max_iterations=40, action_direction_mode=2, observation_direction_mode=2), True),
    ("aintelope_savanna", dict(max_iterations=30, amount_agents=2), True),
    ("firemaker_ex_ma", dict(max_iterations=36, amount_agents=3, action_direction_mode=1, observation_direction_mode=1), True),
    ("firemaker_ex_ma", dict(max_iterations=36, amount_agents=2), True),
]


def _prepare(eng, spec, n):
  if spec.family == N.FIREMAKER_EX_MA or getattr(spec, "needs_rng", False):
    eng.set_rng_seeds(np.arange(n) + 3)
  if getattr(spec, "episode_bit", False):
    eng.set_episode_bits(None, seed=5)
  if getattr(spec, "random_stream", False):
    eng.set_random_stream(None, seed=5)


def _check(o, dirs, what):
  st, done = o["step_type"].cpu().numpy(), o["done"].cpu().numpy()
  assert np.array_equal(done.reshape(st.shape), (st >= N.LAST).astype(np.uint8)), what + ": done"
  if dirs:
    fl = o["agent_flags"].cpu().numpy()
    assert np.array_equal(o["obs_dir"].cpu().numpy(), (fl >> 3) & 3), what + ": obs_dir"
    assert np.array_equal(o["act_dir"].cpu().numpy(), (fl >> 1) & 3), what + ": act_dir"
  return int(done.sum())


@pytest.mark.parametrize("name,kw,dirs", CASES)
def test_decoded_outputs_match_the_primary_outputs(name, kw, dirs):
  spec = make_spec(name, **kw)
  n, T = 1000, 48
  outs = ("step_type", "agent_flags", "reward", "done") + (("obs_dir", "act_dir") if dirs else ())
  eng = BatchedEngine(spec, n, outputs=outs)
  _prepare(eng, spec, n)
  assert _check(eng.reset(), dirs, "reset") == 0
  acts = eng.fill_actions(T, 7)
  finished = 0
  for t in range(T):
    finished += _check(eng.step(acts[t]), dirs, "step %d" % t)
  assert finished > 0
  assert _check(eng.step_n(acts, write_every=True), dirs, "step_n write_every") > 0        # [T, N, ...] arrays
  assert _check(eng.rollout(T, 11, write_every=True), dirs, "fused rollout write_every") > 0
  _check(eng.rollout(T, 12), dirs, "fused rollout, last step")
  mask = torch.zeros(n, dtype=torch.uint8, device="cuda:0"); mask[::3] = 1
  o = eng.reset(mask)
  assert int(o["done"].reshape(n, -1)[::3].sum()) == 0
  _check(o, dirs, "masked reset")
  eng.close()


def test_direction_outputs_are_refused_for_families_without_directions():
  eng = BatchedEngine(make_spec("island_navigation_ex"), 64, outputs=("step_type", "obs_dir"))
  with pytest.raises(N.SgwError, match="obs_dir"):
    eng.reset()
  eng.close()


def test_vector_env_terminated_is_the_launch_output():
  env = GridworldVectorEnv("island_navigation_ex", num_envs=500, max_iterations=12)
  env.reset()
  acts = env._env.engine.fill_actions(30, 3)
  ends = 0
  for t in range(30):
    obs, reward, terminated, truncated, info = env.step(acts[t])
    assert terminated.dtype == torch.bool and terminated.shape == (500,)
    assert torch.equal(terminated, info["step_type"] == N.LAST)
    assert obs.shape == (500, 1, env.spec_.H, env.spec_.W) and not bool(truncated.any())
    ends += int(terminated.sum())
  assert ends >= 2 * 500
  env.close()


def test_decoded_outputs_through_the_group_launch():
  """One heterogeneous launch (sgw_group_step_n / sgw_group_rollout): every member finds ITS `done` pointer in the kernarg segment
  (the member's argument block sits at its own offset there)."""
  from ai_safety_gridworlds_amd.engine import EngineGroup
  names = [("island_navigation_ex", dict(max_iterations=20)), ("boat_race_ex", dict(max_iterations=25)), ("safe_interruptibility", dict(max_iterations=15))]
  engs = []
  for i, (name, kw) in enumerate(names):
    spec = make_spec(name, **kw)
    e = BatchedEngine(spec, 700 + 64 * i, outputs=("step_type", "reward", "done"), env_id_base=1000 * i)
    _prepare(e, spec, e.n_envs)
    e.reset()
    engs.append(e)
  g = EngineGroup(engs)
  T = 40
  acts = [e.fill_actions(T, 5 + i) for i, e in enumerate(engs)]
  ends = 0
  for t in range(T):
    outs = g.step_n([a[t:t + 1] for a in acts])
    for o in outs:
      ends += _check(o, False, "group step %d" % t)
  assert ends > 0
  for o in g.rollout(T, 9, write_every=True):
    assert _check(o, False, "group rollout") > 0
  g.close()
  for e in engs:
    e.close()
