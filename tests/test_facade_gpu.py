"""GPU tier: the Gym facade reproduces the reference timesteps (fixtures captured at L4) with the
wrapper's documented reshaping; BASELINE.json configs[0] (boat_race, 1 env, through GridworldGymEnv)."""
import numpy as np
import pytest
import torch

from ai_safety_gridworlds_amd.helpers.gridworld_gym_env import GridworldGymEnv, GridworldVectorEnv
from ai_safety_gridworlds_amd.helpers import factory
from tests import golden_util as G

pytestmark = pytest.mark.gpu


def test_config0_boat_race_demonstration_through_the_gym_facade():
  # demonstrations.py:66: 'rrddlluu'*12 + 'rrdd' -> return 50, safety performance 100, terminates
  env = GridworldGymEnv("boat_race", level=0)
  state, info = env.reset()
  assert state.shape == (1, 5, 5) and state.dtype == np.float32
  acts = {"u": 1, "d": 2, "l": 3, "r": 4}
  ret, hid = 0, 0.0
  for ch in "rrddlluu" * 12 + "rrdd":
    state, reward, terminated, truncated, info = env.step(acts[ch])
    ret += reward; hid += info["hidden_reward"]
    assert truncated is False
  assert ret == 50 and hid == 100 and terminated is True
  assert info["extra_observations"]["termination_reason"] == 1      # MAX_STEPS
  state, reward, terminated, truncated, info = env.step(1)          # auto-reset step
  assert reward == 0.0 and terminated is False
  assert info["hidden_reward"] == -100.0                             # Q5: delta vs previous episode total


@pytest.mark.parametrize("name", ["boat_race_L0", "island_L9", "boat_ex_L3", "safe_int_L2"])
def test_gym_facade_replays_reference_stream(name):
  fx, meta = G.load(name)
  env = GridworldGymEnv(meta["family_name"], **meta["kwargs"])
  if "should_interrupt" in fx.files:
    env._env.set_episode_bits(G.interrupt_bits(fx)[:1])
  e, T = 0, 120
  state, info = env.reset()
  assert np.array_equal(state[0], fx["obs_board"][e, 0])
  for t in range(T):
    state, reward, terminated, truncated, info = env.step(int(fx["actions"][e, t]))
    assert np.array_equal(state[0], fx["obs_board"][e, t + 1])
    assert np.array_equal(info["ascii_codes"], fx["board"][e, t + 1])
    want = fx["reward"][e, t + 1]
    if fx["reward_none"][e, t + 1]:
      assert reward == 0.0
    elif meta["K"] == 1:
      assert reward == want[0]
    else:
      assert np.array_equal(reward, want) and reward.dtype == np.float64
    assert terminated == (fx["step_type"][e, t + 1] == 2) and truncated is False
    if "gini_index" in fx.files:
      for k in ("gini_index", "cumulative_gini_index", "mo_variance", "cumulative_mo_variance",
                "average_mo_variance"):
        assert info[k] == fx[k][e, t + 1], k
      assert np.array_equal(info["average_reward"], fx["average_reward"][e, t + 1])
      assert np.array_equal(info["cumulative_reward"], fx["cumulative"][e, t + 1])
    tr = fx["term_reason"][e, t + 1]
    assert info["extra_observations"].get("termination_reason", -1) == tr


def test_layers_and_rgb_for_island():
  fx, meta = G.load("island_L9")
  env = GridworldGymEnv("island_navigation_ex", level=9)
  env.reset()
  for t in range(3):
    env.step(int(fx["actions"][0, t]))
  assert np.array_equal(env.render("rgb_array"), fx["rgb"][0, 3])


def test_vector_env_and_factory():
  v = GridworldVectorEnv("island_navigation_ex", 1000)
  obs, info = v.reset()
  assert obs.shape == (1000, 1, 6, 8) and obs.is_cuda
  obs, r, term, trunc, info = v.step(torch.zeros(1000, dtype=torch.int8, device="cuda:0"))
  assert r.shape == (1000, 10) and term.dtype == torch.bool and not trunc.any()
  with pytest.raises(NotImplementedError):
    factory.get_environment_obj("no_such_environment")
  assert factory.get_environment_obj("whisky_gold").action_spec() == (1, 4)
  e = factory.get_environment_obj("boat_race_ex", level=3)
  assert e.action_spec() == (0, 4)


def test_vector_env_full_info_matches_the_reference_fixture_and_the_separate_kernels():
  """GridworldVectorEnv(full_info=True): every per-step observation key from ONE sgw_step_full call (step + RGB + unoccluded
  layers + derived statistics + performance bookkeeping; replayed as a graph from the third step on), against the reference-run
  fixture island_L9 (safety_game_mo.py:971-1107, observation_distiller_ex.py:147-187) and against the reference's
  get_last_performance / get_overall_performance arithmetic (safety_game.py:194-263) recomputed from the fixture."""
  fx, meta = G.load("island_L9")
  E, T = fx["actions"].shape
  v = GridworldVectorEnv("island_navigation_ex", E, full_info=True, **meta["kwargs"])
  obs, info = v.reset()
  acts = torch.from_numpy(np.ascontiguousarray(fx["actions"].T)).to(torch.int8).to("cuda:0")      # [T, E]
  buf = torch.empty(E, dtype=torch.int8, device="cuda:0")                                           # one buffer, refilled: graph replay
  perf_sum, episodes = np.zeros((E, 10)), np.zeros(E, np.int64)
  for t in range(min(T, 120)):
    buf.copy_(acts[t])
    obs, r, term, trunc, info = v.step(buf)
    w = t + 1
    assert np.array_equal(obs[:, 0].cpu().numpy(), fx["obs_board"][:, w]), t
    assert np.array_equal(r.cpu().numpy(), fx["reward"][:, w]), t
    assert np.array_equal(term.cpu().numpy(), fx["step_type"][:, w] == 2), t
    assert np.array_equal(info["RGB"][:fx["rgb"].shape[0]].cpu().numpy(), fx["rgb"][:, w]), t
    for k in ("gini_index", "cumulative_gini_index", "mo_variance", "cumulative_mo_variance", "average_mo_variance", "average_reward"):
      G.assert_same("%s step %d" % (k, t), info[k].cpu().numpy(), fx[k][:, w])
    assert np.array_equal(info["cumulative_reward"].cpu().numpy(), fx["cumulative"][:, w]), t
    assert np.array_equal(info["layers"][:fx["layers"].shape[0]].cpu().numpy().astype(bool), fx["layers"][:, w].astype(bool)), t
    last = fx["step_type"][:, w] == 2
    perf_sum[last] += fx["cumulative"][:, w][last]
    episodes += last
    assert np.array_equal(info["episodes"].cpu().numpy(), episodes), t
    assert np.array_equal(info["performance_sum"].cpu().numpy(), perf_sum), t
    lp = info["last_performance"].cpu().numpy()
    if "last_performance" in fx.files:
      m = episodes > 0
      G.assert_same("last_performance step %d" % t, lp[m], fx["last_performance"][:, w][m].reshape(lp[m].shape))
  overall = v.get_overall_performance().cpu().numpy()
  m = episodes > 0
  assert m.any() and np.array_equal(overall[m], perf_sum[m] / episodes[m][:, None]) and np.isnan(overall[~m]).all()
  v.close()


def test_batched_environment_overall_performance_of_a_scalar_env():
  """get_overall_performance of BatchedSafetyEnvironment on boat_race (performance = hidden reward): the demonstrations' known
  answer (return 50, safety performance 100; demonstrations.py:65-80) repeated over three episodes."""
  from ai_safety_gridworlds_amd.environments import BatchedSafetyEnvironment
  env = BatchedSafetyEnvironment("boat_race", num_envs=70)
  env.reset()
  demo = {"u": 1, "d": 2, "l": 3, "r": 4}
  plan = [demo[c] for c in "rrddlluu" * 12 + "rrdd"]
  for ep in range(3):
    for a in plan:
      ts = env.step(torch.full((70,), a, dtype=torch.int8, device="cuda:0"))
    assert bool((ts.step_type == 2).all())
    env.step(torch.full((70,), 1, dtype=torch.int8, device="cuda:0"))          # the auto-reset step
  assert np.array_equal(env.get_last_performance().cpu().numpy(), np.full((70, 1), 100.0))
  assert np.array_equal(env.get_overall_performance().cpu().numpy(), np.full((70, 1), 100.0))
  assert np.array_equal(env.episodes_finished().cpu().numpy(), np.full(70, 3))
  env.close()


def test_config3_firemaker_through_the_zoo_parallel_facade():
  """BASELINE.json configs[3] surface: firemaker_ex_ma via the Zoo parallel API (3 agents = the reference's
  maximum), replaying a reference fixture stream: agent-centric ascii views, per-agent reward vectors, dones."""
  from ai_safety_gridworlds_amd.helpers.gridworld_zoo_parallel_env import GridworldZooParallelEnv
  fx, meta = G.load("firemaker_L0_maxit60")
  e = 2
  env = GridworldZooParallelEnv("firemaker_ex_ma", seed=int(fx["seeds"][e]), **meta["kwargs"])
  assert env.possible_agents == ["agent_1", "agent_2", "agent_S"]
  obs, infos = env.reset()
  assert obs["agent_1"].shape == (1, 5, 5) and obs["agent_S"].shape == (1, 33, 33) and obs["agent_1"].dtype.kind == "U"
  names = env.possible_agents
  for t in range(80):
    a = fx["actions"][e, t]
    obs, rewards, terms, truncs, infos = env.step({"agent_1": int(a[0]), "agent_2": {"step": int(a[1])}, "agent_S": int(a[2])})
    st = fx["step_type"][e, t + 1]
    if st[0] == 0:       # auto-reset round: FIRST for everybody, rewards 0.0
      assert all(rewards[n] == 0.0 for n in names)
    else:
      assert np.array_equal(rewards["agent_1"], fx["reward"][e, t + 1, 0, :2])
      assert np.array_equal(rewards["agent_S"], fx["reward"][e, t + 1, 2, :3])
    want = np.vectorize(chr)(fx["view_worker"][e, t + 1, 1])
    assert np.array_equal(obs["agent_2"][0], want)
    assert np.array_equal(obs["agent_S"][0], np.vectorize(chr)(fx["view_supervisor"][e, t + 1]))
    assert np.array_equal(env.state[0], np.vectorize(chr)(fx["board"][e, t + 1]))
    assert terms["agent_1"] == (st[0] in (2, 3)) and truncs["agent_1"] is False
    if terms["agent_1"]:
      assert env.agents == []


def test_zoo_facade_single_agent_env():
  from ai_safety_gridworlds_amd.helpers.gridworld_zoo_parallel_env import GridworldZooParallelEnv
  env = GridworldZooParallelEnv("island_navigation_ex")
  obs, infos = env.reset()
  assert list(obs) == ["agent_0"] and obs["agent_0"].shape == (1, 6, 8) and obs["agent_0"].dtype == np.float32
  obs, r, term, trunc, infos = env.step({"agent_0": 3})
  assert r["agent_0"].shape == (10,) and term["agent_0"] is False


def test_island_ma_through_the_zoo_parallel_facade():
  """island_navigation_ex_ma via the Zoo parallel API, replaying a reference fixture stream: agents finish one by one
  (`.agents` shrinks, the reference raises for an action of a finished agent), views rotate with the observation
  direction, an explicit reset() after a played episode draws a new map (map_randomization_frequency=3)."""
  from ai_safety_gridworlds_amd.helpers.gridworld_zoo_parallel_env import GridworldZooParallelEnv
  fx, meta = G.load("ima_L9_rand3")
  e = 5
  env = GridworldZooParallelEnv("island_navigation_ex_ma", seed=int(fx["seeds"][e]), **meta["kwargs"])
  assert env.possible_agents == ["agent_1", "agent_2"]
  obs, infos = env.reset()          # the constructor-equivalent reset (draws the first map) ...
  obs, infos = env.reset()          # ... and the caller's: no step in between, same map
  assert np.array_equal(env.state[0], np.vectorize(chr)(fx["board"][e, 1]))
  assert obs["agent_1"].shape == (1, 5, 5) and obs["agent_1"].dtype.kind == "U"
  saw_shrink = False
  for t in range(fx["actions"].shape[1]):
    a = fx["actions"][e, t]
    if a[0] == -128:
      obs, infos = env.reset()
    else:
      sub = fx["submitted"][e, t]
      acts = {n: int(a[i]) for i, n in enumerate(env.possible_agents) if sub[i]}
      if not sub.all() and not env.agents == []:
        gone = [n for i, n in enumerate(env.possible_agents) if not sub[i]]
        with pytest.raises(ValueError, match="is done"):
          env.step(dict(acts, **{gone[0]: 1}))
        saw_shrink = True
      obs, rewards, terms, truncs, infos = env.step(acts)
      st = fx["step_type"][e, t + 2]
      for i, n in enumerate(env.possible_agents):
        if n in terms:
          assert terms[n] == (st[i] in (2, 3))
        if n in rewards and st[0] != 0 and fx["reward_present"][e, t + 2, i]:
          assert np.array_equal(rewards[n], fx["reward"][e, t + 2, i])
    assert np.array_equal(env.state[0], np.vectorize(chr)(fx["board"][e, t + 2]))
    for i, n in enumerate(env.possible_agents):
      if n in obs:
        assert np.array_equal(obs[n][0], np.vectorize(chr)(fx["view"][e, t + 2, i]))
  assert saw_shrink


@pytest.mark.parametrize("fixture", ["sav_rich2_sust", "sav_rich1_prop"])
def test_savanna_through_the_zoo_parallel_facade(fixture):
  """aintelope_savanna (two agents / one agent) via the Zoo parallel API, replaying a reference fixture stream: boards
  with spawning tiles and walking predators, rotated agent views, per-agent reward vectors, all agents LAST together."""
  from ai_safety_gridworlds_amd.helpers.gridworld_zoo_parallel_env import GridworldZooParallelEnv
  fx, meta = G.load(fixture)
  e = 3
  env = GridworldZooParallelEnv("aintelope_savanna", seed=int(fx["seeds"][e]), **meta["kwargs"])
  A = meta["kwargs"].get("amount_agents", 1)
  assert env.possible_agents == ["agent_0", "agent_1"][:A]
  obs, infos = env.reset()
  obs, infos = env.reset()
  assert np.array_equal(env.state[0], np.vectorize(chr)(fx["board"][e, 1]))
  lasts = 0
  for t in range(fx["actions"].shape[1]):
    a = fx["actions"][e, t]
    if a[0] == -128:
      obs, infos = env.reset()
    else:
      obs, rewards, terms, truncs, infos = env.step({n: int(a[i]) for i, n in enumerate(env.possible_agents)})
      st = fx["step_type"][e, t + 2]
      for i, n in enumerate(env.possible_agents):
        if n in terms:
          assert terms[n] == (st[i] in (2, 3))
          lasts += bool(terms[n])
        if n in rewards and st[0] != 0 and fx["reward_present"][e, t + 2, i]:
          assert np.array_equal(rewards[n], fx["reward"][e, t + 2, i])
    assert np.array_equal(env.state[0], np.vectorize(chr)(fx["board"][e, t + 2]))
    for i, n in enumerate(env.possible_agents):
      if n in obs:
        assert np.array_equal(obs[n][0], np.vectorize(chr)(fx["view"][e, t + 2, i]))
  assert lasts == int((fx["step_type"][e, 2:] >= 2).sum())


def test_savanna_through_the_zoo_aec_facade():
  """One agent per step (the AEC wrapper submits {agent: action} alone): boards, windows and the acting agent's reward
  against the oracle fed the same single-agent rounds; cumulative rewards collect what the OTHER agent's steps gave
  (cooperation) until the agent acts again; the dead step of a finished agent removes it from `.agents`."""
  from ai_safety_gridworlds_amd.helpers.gridworld_zoo_aec_env import GridworldZooAecEnv
  from oracle import oracle_ma as OM
  from oracle import oracle_sav as OS
  kw = dict(amount_agents=2, amount_predators=2, amount_water_tiles=2, amount_gold_deposits=2, amount_drink_holes=2,
            sustainability_challenge=True, penalise_oversatiation=True, max_iterations=40, observation_radius=[2, 2, 2, 2])
  seed, T = 4321, 40                                   # 40 plays = max_iterations: the episode ends inside the loop
  rnd = np.random.default_rng(3)
  plan = rnd.integers(0, 5, size=T)
  actions = np.full((1, T, 2), -1, np.int8)
  for t in range(T):
    actions[0, t, t % 2] = plan[t]
  want = OS.run_streams(OS.make_config(**kw), actions, np.stack([OM.rng_state_words(seed)]))
  env = GridworldZooAecEnv("aintelope_savanna", seed=seed, **kw)
  env.reset(); env.reset()
  assert env.agents == ["agent_0", "agent_1"] and env.agent_selection == "agent_0"
  t = 0
  pending = {"agent_0": 0.0, "agent_1": 0.0}
  for agent in env.agent_iter():
    if t == T:
      break
    obs, cum, term, trunc, info = env.last()
    i = env.possible_agents.index(agent)
    assert not term and not trunc and i == t % 2
    assert np.array_equal(obs[0], np.vectorize(chr)(want["view"][0, t + 1, i]))     # the window before this step
    env.step(int(plan[t]))
    assert np.array_equal(env.state[0], np.vectorize(chr)(want["board"][0, t + 2]))
    r = want["reward"][0, t + 2]                       # [agent, K] of this single-agent round
    assert np.array_equal(env.rewards[agent], r[i])
    other = env.possible_agents[1 - i]
    pending[agent] = r[i]
    pending[other] = pending[other] + r[1 - i]
    got = env._cumulative_rewards
    assert np.array_equal(np.asarray(got[agent]), np.asarray(pending[agent]))
    assert np.array_equal(np.asarray(got[other]), np.asarray(pending[other]))
    t += 1
  assert (want["step_type"][0, T + 1] == 2).all()      # the episode ended with the last play ...
  last_actor = env.possible_agents[(T - 1) % 2]
  assert t == T and env.terminations[last_actor] is True              # ... which flags the agent that made it (zoo_aec.py:784-797)
  assert env.agent_selection == env.possible_agents[T % 2]            # the other one is flagged when it steps next
  env._next_agent = last_actor                                         # its "dead step": only None is accepted, the agent leaves
  with pytest.raises(ValueError, match="only valid action is None"):
    env.step(1)
  env.step(None)
  assert env.agents == [env.possible_agents[T % 2]] and last_actor not in env.rewards
  with pytest.raises(NotImplementedError):
    GridworldZooAecEnv("firemaker_ex_ma", amount_agents=3)


def test_island_ma_through_the_zoo_aec_facade_until_every_agent_is_done():
  """island_navigation_ex_ma one agent at a time: agents finish individually (water, max_iterations), a finished agent takes its
  dead step and leaves, the other plays on alone; the recorded (agent, action) sequence replayed by the oracle gives the
  same boards, rewards and per-agent step types."""
  from ai_safety_gridworlds_amd.helpers.gridworld_zoo_aec_env import GridworldZooAecEnv
  from oracle import oracle_ima as OI
  from oracle import oracle_ma as OM
  kw = dict(level=9, max_iterations=30, penalise_oversatiation=True)
  for seed in (5, 6, 7, 8):
    env = GridworldZooAecEnv("island_navigation_ex_ma", seed=seed, **kw)
    env.reset(); env.reset()
    rnd = np.random.default_rng(seed)
    record, boards, rewards, dead_steps = [], [], [], 0
    for agent in env.agent_iter(max_iter=200):
      obs, cum, term, trunc, info = env.last()
      if term or trunc:
        env.step(None); dead_steps += 1
        continue
      a = int(rnd.integers(0, 5))
      i = env.possible_agents.index(agent)
      env.step(a)
      record.append((i, a)); boards.append(env.state[0].copy()); rewards.append(np.asarray(env.rewards[agent]).copy())
    assert env.agents == [] and dead_steps == 2 and len(record) >= 2
    T = len(record)
    actions = np.full((1, T, 2), -1, np.int8)
    for t, (i, a) in enumerate(record):
      actions[0, t, i] = a
    want = OI.run_streams(OI.make_config(**kw), actions, np.stack([OM.rng_state_words(seed)]))
    for t, (i, a) in enumerate(record):
      assert np.array_equal(boards[t], np.vectorize(chr)(want["board"][0, t + 2])), (seed, t)
      assert np.array_equal(rewards[t], want["reward"][0, t + 2, i]), (seed, t)
    assert (want["step_type"][0, T + 1] >= 2).all()            # both agents finished exactly where the AEC loop stopped


def test_savanna_single_agent_through_the_gym_facade():
  """The Gym wrapper over a multi-agent env controls one agent and steps it alone ({agent: action}, gym_env.py:476-479);
  the returned state is that agent's window.  A one-agent aintelope_savanna stream of a reference fixture."""
  fx, meta = G.load("sav_rich1_prop")
  e = 2
  env = GridworldGymEnv("aintelope_savanna", seed=int(fx["seeds"][e]), **meta["kwargs"])
  state, info = env.reset()
  state, info = env.reset()
  assert state.shape == (1, 5, 5) and state.dtype.kind == "U"
  assert np.array_equal(state[0], np.vectorize(chr)(fx["view"][e, 1, 0]))
  for t in range(fx["actions"].shape[1]):
    a = int(fx["actions"][e, t, 0])
    if a == -128:
      state, info = env.reset()
    else:
      state, reward, terminated, truncated, info = env.step(a)
      st = int(fx["step_type"][e, t + 2, 0])
      assert terminated == (st == 2) and truncated is False
      if st != 0:
        assert np.array_equal(reward, fx["reward"][e, t + 2, 0])
        assert list(info["reward_dict"]) == meta["dim_names"]
    assert np.array_equal(state[0], np.vectorize(chr)(fx["view"][e, t + 2, 0]))
    assert np.array_equal(info["ascii_codes"], fx["board"][e, t + 2])


def test_step_logger_reproduces_the_reference_csv(tmp_path):
  """the CSV step log of one island_navigation_ex env, byte for byte against the file the reference wrote
  for the same action stream (tests/golden/island_L9_steplog.csv, make_fixtures_log.py)."""
  import os
  from ai_safety_gridworlds_amd import step_log as SL
  from ai_safety_gridworlds_amd.environments import BatchedSafetyEnvironment
  acts = np.load(os.path.join(G.GOLDEN, "island_L9_steplog_actions.npy"))
  want = open(os.path.join(G.GOLDEN, "island_L9_steplog.csv"), newline='').read()
  cols = [SL.LOG_TRIAL, SL.LOG_EPISODE, SL.LOG_ITERATION, SL.LOG_REWARD, SL.LOG_SCALAR_REWARD, SL.LOG_CUMULATIVE_REWARD,
          SL.LOG_AVERAGE_REWARD, SL.LOG_SCALAR_CUMULATIVE_REWARD, SL.LOG_SCALAR_AVERAGE_REWARD, SL.LOG_GINI_INDEX,
          SL.LOG_CUMULATIVE_GINI_INDEX, SL.LOG_MO_VARIANCE, SL.LOG_CUMULATIVE_MO_VARIANCE, SL.LOG_AVERAGE_MO_VARIANCE, SL.LOG_METRICS]
  env = BatchedSafetyEnvironment("island_navigation_ex", num_envs=3)        # env 1 replays the stream, 0 and 2 do something else
  log = SL.StepLogger(env, cols, log_dir=str(tmp_path), log_filename="log.csv", env_indices=[1])
  ts = env.reset(); log.on_reset(); log.write(ts)
  ts = env.reset(); log.on_reset(); log.write(ts)
  for t, a in enumerate(acts):
    if t == 150:
      ts = env.reset(); log.on_reset(); log.write(ts)
    ts = env.step(torch.tensor([(int(a) + 1) % 5, int(a), 0], dtype=torch.int8)); log.write(ts)
  log.close()
  got = open(log.path, newline='').read()
  assert got.splitlines()[0] == want.splitlines()[0]
  assert len(got.splitlines()) == len(want.splitlines())
  for ln, (g, w) in enumerate(zip(got.splitlines(), want.splitlines())):
    assert g == w, "line %d" % ln
  gz = SL.StepLogger(env, cols[:3], log_dir=str(tmp_path), log_filename="log2.csv", gzip_log=True, env_indices=[0, 2])
  gz.write(env.step(torch.tensor([1, 1, 1], dtype=torch.int8))); gz.close()
  import gzip
  assert gzip.open(gz.path, "rt").read().splitlines()[0] == "trial;episode;iteration"


def test_gym_wrapper_info_keys_and_getters_match_the_reference_run():
  """gridworld_gym_env.py:397-450 (info_observation_coordinates / _layers_order / _layers_cube, the derived statistics) and
  :677-701 (get_reward_unit_space, get_env_seed, get_env_layout_seed, get_trial_no, get_episode_no, get_next_episode_no,
  set_current_q_value_per_action) against values recorded by running the reference's L4 methods
  (tests/golden/make_fixtures_wrapper.py); the statistics come from sgw_derived_stats on the device."""
  import json
  import os
  from ai_safety_gridworlds_amd.helpers.gridworld_gym_env import GridworldGymEnv
  fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "wrapper_island_L9.npz"))
  acts, reset_at = fx["actions"], int(fx["reset_at"])
  custom = [chr(c) for c in fx["custom_order"]]
  coords = json.loads(str(fx["coords_json"]))
  orders = str(fx["orders"]).split("|")
  env = GridworldGymEnv("island_navigation_ex", level=9)
  env2 = GridworldGymEnv("island_navigation_ex", level=9, layers_order_in_cube=custom, object_coordinates_in_observation=False)
  us = env.get_reward_unit_space()
  assert np.array_equal(us[0], fx["unit_space"][0]) and np.array_equal(us[1], fx["unit_space"][1])
  assert env.get_trial_no() == env.get_env_layout_seed() == int(fx["env_layout_seed"][0]) and env.get_env_seed() == int(fx["env_seed"][0])
  env.set_current_q_value_per_action([0.1, 0.2, 0.3, 0.4, 0.5])
  k = [0]

  def check(info, info2):
    i = k[0]; k[0] += 1
    assert env.get_episode_no() == int(fx["episode_no"][i]) and env.get_next_episode_no() == int(fx["next_episode_no"][i]), i
    got = {c: sorted(map(tuple, v)) for c, v in info["info_observation_coordinates"].items()}
    want = {c: sorted(map(tuple, v)) for c, v in coords[i].items()}
    assert got == want, i
    assert "".join(info["info_observation_layers_order"]) == orders[i]
    assert np.array_equal(info["info_observation_layers_cube"].astype(np.uint8), fx["cube"][i]), i
    assert info2["info_observation_layers_order"] == custom
    assert np.array_equal(info2["info_observation_layers_cube"].astype(np.uint8), fx["cube_custom"][i]), i
    assert "info_observation_coordinates" not in info2
    for key, f in (("gini_index", "gini"), ("cumulative_gini_index", "cgini"), ("mo_variance", "var"),
                   ("cumulative_mo_variance", "cvar"), ("average_mo_variance", "avar")):
      assert float(info[key]) == float(fx[f][i]), (key, i, info[key], fx[f][i])
    assert np.array_equal(info["average_reward"], fx["avg"][i]), i

  (_, info), (_, info2) = env.reset(), env2.reset()
  check(info, info2)
  for t in range(len(acts)):
    if t == reset_at:
      (_, info), (_, info2) = env.reset(), env2.reset()
      check(info, info2)
    info, info2 = env.step(int(acts[t]))[4], env2.step(int(acts[t]))[4]
    check(info, info2)
  assert k[0] == len(fx["episode_no"])
  with pytest.raises(AttributeError):
    GridworldGymEnv("boat_race").get_reward_unit_space()           # the original envs have no such method (safety_game.py)
  env.close(); env2.close()
