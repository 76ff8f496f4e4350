"""The PettingZoo-parallel facade's per-agent info keys == what the reference wrapper's `_process_observation` / `_compute_infos`
(gridworld_zoo_parallel_env.py:286-376) put there, i.e. the results of the L4 environment's own methods, which
tests/golden/make_fixtures_zoo.py recorded by running the reference: global layers / coordinates / order / cube, directions,
agent-centric ascii and value boards, per-agent layer dicts, relative coordinates and cubes."""
import json
import os

import numpy as np
import pytest

from ai_safety_gridworlds_amd.helpers import gridworld_zoo_parallel_env as Z
from tests import golden_util as G

pytestmark = pytest.mark.gpu


def test_zoo_parallel_infos_match_reference_fixture():
  fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zoo_island_ma_L9.npz"))
  coords, agent_coords = json.loads(str(fx["coords_json"])), json.loads(str(fx["agent_coords_json"]))
  orders = str(fx["orders"]).split("|")
  agent_orders = [p.split(",") for p in str(fx["agent_orders"]).split("|")]
  acts = fx["actions"]
  names = ["agent_1", "agent_2"]
  env = Z.GridworldZooParallelEnv("island_navigation_ex_ma", level=9, max_iterations=100, seed=int(fx["seed"]))
  envf = Z.GridworldZooParallelEnv("island_navigation_ex_ma", level=9, max_iterations=100, seed=int(fx["seed"]), ascii_observation_format=False)

  def check(t, obs, infos, obs_f, infos_f):
    for i, a in enumerate(names):
      info = infos[a]
      assert np.array_equal(info["ascii_codes"], fx["board"][t]), (t, a)
      assert info[Z.INFO_OBSERVATION_DIRECTION] == fx["obs_dir"][t][i] and info[Z.INFO_ACTION_DIRECTION] == fx["act_dir"][t][i], (t, a)
      assert info[Z.INFO_OBSERVATION_LAYERS_ORDER] == list(orders[t])
      assert np.array_equal(info[Z.INFO_OBSERVATION_LAYERS_CUBE], fx["cube"][t].astype(bool)), (t, a)
      assert {k: [list(x) for x in v] for k, v in info[Z.INFO_OBSERVATION_COORDINATES].items()} == coords[t], (t, a)
      assert sorted(info[Z.INFO_OBSERVATION_LAYERS_DICT]) == sorted(orders[t])
      want_view = np.vectorize(chr)(fx["agent_ascii"][t][i])
      assert np.array_equal(info[Z.INFO_AGENT_OBSERVATIONS], want_view), (t, a)
      assert np.array_equal(obs[a][0], want_view)
      assert np.array_equal(infos_f[a][Z.INFO_AGENT_OBSERVATIONS], fx["agent_board"][t][i]), (t, a)      # value-mapped float board
      assert info[Z.INFO_AGENT_OBSERVATION_LAYERS_ORDER] == list(agent_orders[t][i])
      assert np.array_equal(info[Z.INFO_AGENT_OBSERVATION_LAYERS_CUBE], fx["agent_cube"][t][i].astype(bool)), (t, a)
      got_c = info[Z.INFO_AGENT_OBSERVATION_COORDINATES]
      assert {k: [list(x) for x in v] for k, v in got_c.items()} == agent_coords[t][a[-1]], (t, a)
      lay = info[Z.INFO_AGENT_OBSERVATION_LAYERS_DICT]
      for j, c in enumerate(agent_orders[t][i]):
        assert np.array_equal(lay[c], fx["agent_cube"][t][i][j].astype(bool)), (t, a, c)
      # observation keys the wrapper passes through (dicts over the agent characters), bit for bit
      for j, c in enumerate("12"):
        assert np.array_equal(info["cumulative_reward"][c], fx["cum"][t][j]) and np.array_equal(info["average_reward"][c], fx["avg"][t][j]), (t, c)
        for key, name in (("gini", "gini_index"), ("cgini", "cumulative_gini_index"), ("var", "mo_variance"), ("cvar", "cumulative_mo_variance"),
                          ("avar", "average_mo_variance")):
          assert info[name][c] == fx[key][t][j], (t, c, name, info[name][c], fx[key][t][j])
      want_aa = {c: {"step": int(fx["actual"][t][j])} for j, c in enumerate("12") if fx["actual"][t][j] >= 0}
      assert info["extra_observations"].get("actual_actions", {}) == want_aa, (t, info["extra_observations"])
      cats = str(fx["categories"]).split("|")
      assert sorted(info["agent_attribute_board_ascii_codes"]) == sorted(cats) and not any(v.any() for v in info["agent_attribute_board_ascii_codes"].values())
      assert all(v == {} for v in info["agent_attribute_layers"].values()) and all((v == '').all() for v in info["agent_attribute_board_ascii"].values())

  obs, infos = env.reset()
  obs_f, infos_f = envf.reset()
  check(0, obs, infos, obs_f, infos_f)
  for t in range(acts.shape[0]):
    a = {n: int(acts[t, i]) for i, n in enumerate(names)}
    obs, rewards, terms, truncs, infos = env.step(a)
    obs_f, _, _, _, infos_f = envf.step(a)
    check(t + 1, obs, infos, obs_f, infos_f)
    assert not any(terms.values())
  env.close(); envf.close()


def test_zoo_parallel_callbacks_and_custom_orders():
  calls = []
  env = Z.GridworldZooParallelEnv(
      "island_navigation_ex_ma", level=9, seed=3, layers_order_in_cube=['1', 'W', 'Z'], layers_order_in_cube_per_agent={"agent_2": ['2', 'Z']},
      pre_reset_callback=lambda seed, *a, **k: (calls.append("pre_reset") or True, seed, a, k),
      post_reset_callback=lambda *a, **k: calls.append("post_reset"),
      pre_step_callback=lambda actions, *a, **k: (calls.append("pre_step") or actions),
      post_step_callback=lambda *a, **k: calls.append("post_step"))
  obs, infos = env.reset()
  obs, rewards, terms, truncs, infos = env.step({"agent_1": 0, "agent_2": 0})
  assert calls == ["pre_reset", "post_reset", "pre_step", "post_step"]
  i1, i2 = infos["agent_1"], infos["agent_2"]
  assert i1[Z.INFO_OBSERVATION_LAYERS_ORDER] == ['1', 'W', 'Z'] and i1[Z.INFO_OBSERVATION_LAYERS_CUBE].shape[0] == 3
  assert not i1[Z.INFO_OBSERVATION_LAYERS_CUBE][2].any()                          # 'Z' does not exist: an all-zero plane
  assert i2[Z.INFO_AGENT_OBSERVATION_LAYERS_ORDER] == ['2', 'Z'] and i2[Z.INFO_AGENT_OBSERVATION_LAYERS_CUBE].shape == (2, 5, 5)
  assert i1[Z.INFO_AGENT_OBSERVATION_LAYERS_ORDER] == sorted(i1[Z.INFO_AGENT_OBSERVATION_LAYERS_DICT])   # not listed: every layer
  env.close()


def test_callbacks_with_the_reference_signatures():
  """The four hooks written exactly as a caller of the reference would write them: pre_reset(seed, *args, **kwargs) ->
  (allow, seed, args, kwargs), post_reset(obs, infos), pre_step(actions) -> actions, post_step(actions, obs, rewards,
  terminateds, truncateds, infos)  (gridworld_zoo_parallel_env.py:445-446, 612-613, 619-622, 699-700; gridworld_gym_env.py
  has the same four with the Gym tuple)."""
  seen = {}
  def pre_reset(seed, *args, **kwargs): seen["pre_reset"] = seed; return (True, seed, args, kwargs)
  def post_reset(obs, infos): seen["post_reset"] = (sorted(obs), sorted(infos))
  def pre_step(actions): seen["pre_step"] = dict(actions); return actions
  def post_step(actions, obs, rewards, terminateds, truncateds, infos): seen["post_step"] = (sorted(rewards), sorted(truncateds))
  env = Z.GridworldZooParallelEnv("island_navigation_ex_ma", level=9, seed=3, pre_reset_callback=pre_reset, post_reset_callback=post_reset,
                                  pre_step_callback=pre_step, post_step_callback=post_step)
  env.reset(seed=5)
  env.step({"agent_1": 1, "agent_2": 2})
  assert seen["pre_reset"] == 5 and seen["post_reset"] == (["agent_1", "agent_2"], ["agent_1", "agent_2"])
  assert seen["pre_step"] == {"agent_1": 1, "agent_2": 2} and seen["post_step"] == (["agent_1", "agent_2"], ["agent_1", "agent_2"])
  env.close()
  from ai_safety_gridworlds_amd.helpers.gridworld_gym_env import GridworldGymEnv
  seen.clear()
  def g_post_reset(state, info): seen["post_reset"] = state.shape
  def g_pre_step(action): seen["pre_step"] = action; return action
  def g_post_step(action, state, reward, terminated, truncated, info): seen["post_step"] = (action, terminated, truncated)
  g = GridworldGymEnv("island_navigation_ex", level=9, pre_reset_callback=pre_reset, post_reset_callback=g_post_reset,
                      pre_step_callback=g_pre_step, post_step_callback=g_post_step)
  g.reset(seed=9)
  g.step(2)
  assert seen["pre_reset"] == 9 and seen["post_reset"] == (1, 6, 8) and seen["pre_step"] == 2 and seen["post_step"] == (2, False, False)
  g.close()


def test_multi_discrete_action_spaces():
  """use_multi_discrete_action_space=True (gridworld_gym_env.py:220-221, 753-830; gridworld_zoo_parallel_env.py:225-228): the
  action space is MultiDiscrete([n], start=[min]) with shape (1,), and step() takes its samples."""
  from ai_safety_gridworlds_amd.helpers.gridworld_gym_env import GridworldGymEnv
  g = GridworldGymEnv("boat_race_ex", level=3, use_multi_discrete_action_space=True, seed=1)
  sp = g.action_space
  assert sp.shape == (1,) and sp.nvec.tolist() == [5] and sp.start.tolist() == [0] and sp.min_action == 0 and sp.max_action == 4
  g.reset()
  for _ in range(20):
    a = sp.sample()
    assert a.shape == (1,) and a.dtype == np.int32 and a in sp
    state, reward, terminated, truncated, info = g.step(a)
  assert np.array([7], np.int32) not in sp
  g.close()
  z = Z.GridworldZooParallelEnv("island_navigation_ex_ma", level=9, seed=3, use_multi_discrete_action_space=True)
  z.reset()
  acts = {a: z.action_space(a).sample() for a in z.possible_agents}
  assert all(v.shape == (1,) for v in acts.values())
  obs, rewards, terms, truncs, infos = z.step(acts)
  assert set(rewards) == set(z.possible_agents)
  z.close()


def test_zoo_vector_layer_cubes_equal_the_single_env_facade():
  """The batched facade's cubes (device tensors) == the one-env facade's, env by env, over a few rounds."""
  import torch
  from ai_safety_gridworlds_amd.helpers.gridworld_zoo_vector_env import GridworldZooVectorEnv
  n = 3
  vec = GridworldZooVectorEnv("island_navigation_ex_ma", num_envs=n, level=9, seed=11, layers_in_observation=True)
  singles = [Z.GridworldZooParallelEnv("island_navigation_ex_ma", level=9, seed=11 + i) for i in range(n)]
  vobs, vinfos = vec.reset()
  sres = [e.reset() for e in singles]
  rng = np.random.default_rng(0)
  for t in range(6):
    for i in range(n):
      for a in ("agent_1", "agent_2"):
        info = sres[i][-1][a]
        assert info[Z.INFO_OBSERVATION_LAYERS_ORDER] == vinfos[a]["info_observation_layers_order"]
        assert np.array_equal(info[Z.INFO_OBSERVATION_LAYERS_CUBE], vinfos[a]["info_observation_layers_cube"][i].cpu().numpy().astype(bool))
        assert np.array_equal(info[Z.INFO_AGENT_OBSERVATION_LAYERS_CUBE], vinfos[a]["info_agent_observation_layers_cube"][i].cpu().numpy().astype(bool))
        assert info[Z.INFO_OBSERVATION_DIRECTION] == int(vinfos[a]["observation_direction"][i])
        assert info[Z.INFO_ACTION_DIRECTION] == int(vinfos[a]["action_direction"][i])
    acts = rng.integers(0, 5, size=(n, 2))
    if any(any(r[2].values()) for r in sres if len(r) == 5):
      break                                                    # an agent finished: the single-env facade drops it from its dicts
    vobs, _, _, _, vinfos = vec.step({"agent_1": torch.tensor(acts[:, 0], dtype=torch.int8, device=vec.device),
                                      "agent_2": torch.tensor(acts[:, 1], dtype=torch.int8, device=vec.device)})
    sres = [e.step({"agent_1": int(acts[i, 0]), "agent_2": int(acts[i, 1])}) for i, e in enumerate(singles)]
  vec.close()
  for e in singles:
    e.close()


@pytest.mark.parametrize("agent", ["1", "2"])
def test_gym_facade_agent_observation_infos_at_reset(agent):
  """GridworldGymEnv over a multi-agent env controls ONE agent (gym_env.py:373-384, 428-439): its agent-centric info keys at
  reset == the reference fixture's slot 0 for that agent."""
  from ai_safety_gridworlds_amd.helpers.gridworld_gym_env import GridworldGymEnv
  fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zoo_island_ma_L9.npz"))
  agent_coords = json.loads(str(fx["agent_coords_json"]))
  i = int(agent) - 1
  env = GridworldGymEnv("island_navigation_ex_ma", level=9, max_iterations=100, seed=int(fx["seed"]), agent_character=agent,
                        layers_order_in_cube=[])
  state, info = env.reset()
  assert np.array_equal(info["info_agent_observations"], np.vectorize(chr)(fx["agent_ascii"][0][i]))
  assert info["observation_direction"] == fx["obs_dir"][0][i] and info["action_direction"] == fx["act_dir"][0][i]
  assert info["info_agent_observation_layers_order"] == list(str(fx["agent_orders"]).split("|")[0].split(",")[i])
  assert np.array_equal(info["info_agent_observation_layers_cube"], fx["agent_cube"][0][i].astype(bool))
  assert {k: [list(x) for x in v] for k, v in info["info_agent_observation_coordinates"].items()} == agent_coords[0][agent]
  assert np.array_equal(info["info_observation_layers_cube"], fx["cube"][0].astype(bool))
  env.close()


@pytest.mark.parametrize("name,kw,n", [
    ("firemaker_ex_ma", dict(amount_agents=3, max_iterations=60), 300),
    ("island_navigation_ex_ma", dict(level=9, observation_direction_mode=2, action_direction_mode=2), 200),
    ("aintelope_savanna", dict(amount_agents=2, amount_predators=1, amount_water_tiles=2, observation_radius=[2, 2, 2, 2]), 150),
])
def test_step_full_of_the_multi_agent_families_equals_the_separate_kernels(name, kw, n):
  """sgw_step_full over the multi-agent families -- unoccluded layers (firemaker: the fire hidden under an agent; savanna: from
  the state bitmaps), the per-agent layer cubes and the derived statistics chained behind the step in one call, replayed as a
  graph from the third step on -- against the same outputs produced by the separate entry points on a twin engine."""
  import numpy as np
  import torch
  from ai_safety_gridworlds_amd.engine import BatchedEngine
  from ai_safety_gridworlds_amd.specs import make_spec
  spec = make_spec(name, **kw)
  outs = ("board", "reward", "cumulative", "frame", "step_type", "agent_pos", "agent_flags")
  a, b = BatchedEngine(spec, n, outputs=outs), BatchedEngine(spec, n, outputs=outs)
  for e in (a, b):
    e.set_rng_seeds(np.arange(n) + 3)
    e.reset()
  acts = a.fill_actions(40, 17)
  buf = torch.empty_like(acts[0])
  for t in range(40):
    buf.copy_(acts[t])
    o = a.step_full(buf, layers=True, stats=True, agent_layer_views=True)
    b.step(acts[t])
    for k in outs:
      assert torch.equal(o[k], b._views()[k]), (t, k)
    assert torch.equal(o["layers"], b.observe_layers()), t
    for x, y in zip(o["agent_layer_views"], b.agent_layer_views()):
      assert torch.equal(x, y), t
    ds = b.derived_stats()
    for k in ("gini_index", "cumulative_gini_index", "mo_variance", "cumulative_mo_variance", "average_mo_variance", "average_reward"):
      assert torch.equal(torch.nan_to_num(o[k], nan=-7.0), torch.nan_to_num(ds[k], nan=-7.0)), (t, k)
  a.close(); b.close()


def test_zoo_parallel_firemaker_direction_modes_against_the_oracle():
  """The single-env PettingZoo-parallel facade over firemaker_ex_ma with relative moves / turning actions: observations (the
  windows rot90-ed by the observation direction), INFO_OBSERVATION_DIRECTION / INFO_ACTION_DIRECTION, rewards and positions of
  every round == the multi-agent oracle on the same generator (firemaker_ex_ma.py:224-226, 472; gridworld_zoo_parallel_env.py:325-335)."""
  from oracle import oracle_ma as OM
  for adm, odm, n_act in ((1, 1, 5), (2, 2, 9)):
    T, seed = 40, 123 + adm
    rs = np.random.RandomState(seed)
    actions = rs.randint(0, n_act, size=(1, T, 3)).astype(np.int8)
    kw = dict(amount_agents=3, max_iterations=60, action_direction_mode=adm, observation_direction_mode=odm,
              FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.05)
    want = OM.run_streams(OM.make_config(**kw), actions, np.stack([OM.rng_state_words(seed)]))
    env = Z.GridworldZooParallelEnv("firemaker_ex_ma", seed=seed, **kw)
    names = ["agent_1", "agent_2", "agent_S"]
    assert env.possible_agents == names and env.action_space("agent_1").n == n_act

    def check(t, obs, infos):
      for q, a in enumerate(names):
        view = want["view_worker"][0, t, q] if q < 2 else want["view_supervisor"][0, t]
        assert np.array_equal(np.asarray(obs[a])[-1] if np.asarray(obs[a]).ndim == 3 else np.asarray(obs[a]), np.vectorize(chr)(view)), (adm, t, a)
        assert infos[a][Z.INFO_OBSERVATION_DIRECTION] == want["observation_direction"][0, t, q], (adm, t, a)
        assert infos[a][Z.INFO_ACTION_DIRECTION] == want["action_direction"][0, t, q], (adm, t, a)
        assert infos[a]["info_agent_position"] == tuple(want["pos"][0, t, q]), (adm, t, a)

    obs, infos = env.reset()
    check(0, obs, infos)
    for t in range(T):
      obs, rewards, terms, truncs, infos = env.step({a: int(actions[0, t, q]) for q, a in enumerate(names)})
      check(t + 1, obs, infos)
    env.close()
