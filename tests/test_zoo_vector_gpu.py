"""GPU parity of the batched PettingZoo-parallel surface (GridworldZooVectorEnv) at BASELINE.json configs[3] size:
firemaker_ex_ma, 16 384 envs x 3 agents (the reference's maximum agent set), 64 rounds, every per-agent array of every round
against the multi-agent CPU oracle (OpenMP), and the reference's DEFAULT agent set (1 worker + supervisor) on a ragged batch.
Reference surface: helpers/gridworld_zoo_parallel_env.py:429-615; semantics firemaker_ex_ma.py:429-709, pycolab_interface_ma.py:173-246."""
import numpy as np
import pytest
import torch

from ai_safety_gridworlds_amd import philox
from ai_safety_gridworlds_amd.helpers.gridworld_zoo_vector_env import GridworldZooVectorEnv

pytestmark = pytest.mark.gpu
SLOT = {"agent_1": 0, "agent_2": 1, "agent_S": 2}


def _run(amount, E, T, seed, **kw):
  from oracle import oracle_ma as OM
  n_act = 9 if kw.get("action_direction_mode", 0) == 2 else 5           # mode 2: the turning actions 5-8
  actions = np.stack([philox.actions(seed, np.arange(E), np.arange(T), 0, n_act, agent=a) for a in range(3)], axis=-1)   # [T, E, 3]
  rng = np.stack([OM.rng_state_words(seed + e) for e in range(E)])
  want = OM.run_streams(OM.make_config(amount_agents=amount, **kw), np.transpose(actions, (1, 0, 2)).copy(), rng, nthreads=16)
  env = GridworldZooVectorEnv("firemaker_ex_ma", num_envs=E, amount_agents=amount, seed=seed, **kw)
  assert env.possible_agents == {1: ["agent_1"], 2: ["agent_1", "agent_S"], 3: ["agent_1", "agent_2", "agent_S"]}[amount]
  dev_actions = torch.from_numpy(actions).to(env.device)
  names = env.possible_agents
  K = {"agent_1": 3 if amount == 1 else 2, "agent_2": 2, "agent_S": 3}

  def check(t, obs, rewards, terms, infos):
    for a in names:
      q = SLOT[a]
      view = want["view_worker"][:, t, q] if q < 2 else want["view_supervisor"][:, t]
      assert obs[a].dtype == torch.uint8 and obs[a].is_cuda
      assert np.array_equal(obs[a].cpu().numpy(), view), (t, a, "observation window")
      assert np.array_equal(infos[a]["cumulative_reward"].cpu().numpy(), want["cumulative"][:, t, q, :K[a]]), (t, a)
      assert np.array_equal(infos[a]["agent_position"].cpu().numpy(), want["pos"][:, t, q]), (t, a)
      assert np.array_equal(infos[a]["step_type"].cpu().numpy(), want["step_type"][:, t, q]), (t, a)
      assert np.array_equal(infos[a]["observation_direction"].cpu().numpy(), want["observation_direction"][:, t, q]), (t, a, "observation direction")
      assert np.array_equal(infos[a]["action_direction"].cpu().numpy(), want["action_direction"][:, t, q]), (t, a, "action direction")
      if rewards is not None:
        assert rewards[a].shape == (E, K[a]) and rewards[a].dtype == torch.float64
        assert np.array_equal(rewards[a].cpu().numpy(), want["reward"][:, t, q, :K[a]]), (t, a, "reward")
        assert np.array_equal(terms[a].cpu().numpy(), want["step_type"][:, t, q] >= 2), (t, a, "terminated")
    assert np.array_equal(infos[names[0]]["board"].cpu().numpy().reshape(E, 17, 17), want["board"][:, t]), t

  obs, infos = env.reset()
  check(0, obs, None, None, infos)
  finished = 0
  for t in range(T):
    obs, rewards, terms, truncs, infos = env.step({a: dev_actions[t, :, SLOT[a]] for a in names})
    check(t + 1, obs, rewards, terms, infos)
    assert not any(bool(v.any()) for v in truncs.values())
    finished += int(terms[names[0]].sum())
  env.close()
  return finished


def test_zoo_vector_firemaker_at_baseline_size_matches_oracle():
  finished = _run(3, 16384, 64, seed=4242, max_iterations=90)      # 3 plays per round: episodes end (and auto-reset) after 30 rounds
  assert finished >= 2 * 16384


def test_zoo_vector_firemaker_default_agent_set_ragged_batch():
  assert _run(2, 1000, 80, seed=77, max_iterations=50, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.05) > 0
  assert _run(1, 333, 60, seed=5, max_iterations=40) > 0


def test_zoo_vector_firemaker_direction_modes():
  """Relative moves with windows rot90-ed by the observation direction, and the turning actions (firemaker_ex_ma.py:224-226, 472)."""
  assert _run(3, 2000, 70, seed=91, max_iterations=60, action_direction_mode=1, observation_direction_mode=1) > 0
  assert _run(3, 1500, 70, seed=92, max_iterations=60, action_direction_mode=2, observation_direction_mode=2,
              FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.04) > 0
  env = GridworldZooVectorEnv("firemaker_ex_ma", num_envs=4, amount_agents=3, seed=1, action_direction_mode=2, observation_direction_mode=2)
  assert env.action_range("agent_1") == (0, 8)                   # NOOP, four moves, four turns (firemaker_ex_ma.py:808-811)
  env.close()


def test_zoo_vector_value_mapped_observations_and_python_int_actions():
  env = GridworldZooVectorEnv("firemaker_ex_ma", num_envs=130, seed=1, ascii_observation_format=False)    # the reference's default: 2 agents
  obs, _ = env.reset()
  assert obs["agent_1"].dtype == torch.float32 and obs["agent_1"].shape == (130, 5, 5) and obs["agent_S"].shape == (130, 33, 33)
  o2, r, term, trunc, info = env.step({"agent_1": 2, "agent_S": torch.zeros(130, dtype=torch.int64, device=env.device)})
  assert r["agent_1"].shape == (130, 2) and r["agent_S"].shape == (130, 3) and float(r["agent_1"][:, 0].max()) == -1.0
  env.close()
