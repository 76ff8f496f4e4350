"""GridworldZooParallelEnv: the reference PettingZoo-parallel wrapper's surface
(helpers/gridworld_zoo_parallel_env.py:100-702) over the batched HIP engine.

    env = GridworldZooParallelEnv("firemaker_ex_ma", amount_agents=3, seed=7)
    obs, infos = env.reset()
    obs, rewards, terminateds, truncateds, infos = env.step({"agent_1": 2, "agent_2": {"step": 0}, "agent_S": 4})

Kept from the reference: agent names `agent_<char>` (zoo.py:188-191; `agent_0` for single-agent envs), per-agent
observation = the agent-centric window of the board (`get_agent_perspective`, safety_game_moma.py:1996-2101)
as ascii characters by default for multi-agent envs (zoo.py:541-554) or the value-mapped float board,
`[np.newaxis]`-stacked; rewards = `np.ndarray float64[K_agent]`; terminated = LAST or DEAD, truncated = False;
`.agents` shrinks as agents finish, `.state` is the global board; `test_death` fault injection (zoo.py:577-586).
One round = ONE kernel launch; the shuffled per-agent sequential plays (the collision resolver) run on device.
pettingzoo is not required (absent in this image); if importable the class derives from ParallelEnv.
"""
import numpy as np
import torch

from .. import _native as N
from ..environments import BatchedSafetyEnvironment

try:
  from pettingzoo import ParallelEnv as _Base
except Exception:                       # pragma: no cover - pettingzoo absent in this image
  _Base = object

from .gridworld_gym_env import DiscreteActionSpace, BoxObservationSpace

OUTS = ("board", "obs_board", "reward", "cumulative", "step_type", "term_reason", "discount", "metrics", "frame",
        "agent_pos", "agent_flags", "hidden", "actual_action")


class GridworldZooParallelEnv(_Base):
  metadata = {"render.modes": ["human", "ansi", "rgb_array"], "name": "ai_safety_gridworlds_amd"}

  def __init__(self, env_name, use_transitions=False, flatten_observations=False, ascii_observation_format=True,
               test_death=False, test_death_probability=0.33, np_random=None, seed=None, device="cuda:0", **kwargs):
    self._env_name = env_name
    self._env = BatchedSafetyEnvironment(env_name, num_envs=1, device=device, outputs=OUTS, **kwargs)
    sp = self.spec_ = self._env.spec
    self._ma = sp.A > 1
    self._ascii_observation_format = ascii_observation_format if self._ma else False   # zoo.py:193
    self._use_transitions, self._flatten_observations = use_transitions, flatten_observations
    self._test_death, self._test_death_probability = test_death, test_death_probability
    chars = sp.agent_chars if self._ma else ['0']
    # column of each agent in the library's per-agent arrays (firemaker_ex_ma keeps the ('1','2','S') layout whatever amount_agents is)
    self._slots = list(getattr(sp, "agent_slots", range(len(chars))))
    self.possible_agents = ["agent_%s" % c for c in chars]
    self.agent_name_mapping = dict(zip(self.possible_agents, chars))
    self._np_random = np_random if np_random is not None else np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    self._seed_env(seed)
    self._dones = {a: False for a in self.possible_agents}
    self._test_deads = {a: False for a in self.possible_agents}
    self._last_agent_boards = {a: None for a in self.possible_agents}
    self._last_hidden_reward = {a: 0.0 for a in self.possible_agents}
    self._state = None
    self._vm = np.array([sp.native.value_map[i] for i in range(128)], np.float32)
    self._action_spaces = {a: DiscreteActionSpace(sp.action_lo, sp.n_actions, self._np_random) for a in self.possible_agents}
    vals = list(sp.value_mapping.values())
    self._observation_spaces = {}
    for i, a in enumerate(self.possible_agents):
      h, w = sp.view_shapes[self._slots[i]] if self._ma else (sp.H, sp.W)
      self._observation_spaces[a] = BoxObservationSpace((2 if use_transitions else 1, h, w), min(vals), max(vals))

  def _seed_env(self, seed):
    if self.spec_.family == N.FIREMAKER_EX_MA or getattr(self.spec_, "needs_rng", False):   # environment_data[NP_RANDOM] = seeding.np_random(seed)[0]
      st = np.random.PCG64(np.random.SeedSequence(seed)).state["state"]
      m = (1 << 64) - 1
      self._env.engine.set_rng_state(np.array([[st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m]],
                                              dtype=np.uint64))

  # ---- PettingZoo surface ---------------------------------------------------------------------
  @property
  def agents(self):
    return [a for a in self.possible_agents if not self._dones[a]]

  @property
  def num_agents(self):
    return len(self.agents)

  @property
  def max_num_agents(self):
    return len(self.possible_agents)

  @property
  def state(self):
    return None if self._state is None else self._state.copy()

  def observation_space(self, agent):
    return self._observation_spaces[agent]

  def action_space(self, agent):
    return self._action_spaces[agent]

  def seed(self, seed=None):
    self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    self._seed_env(seed)

  def close(self):
    self._env.close()

  # ---- helpers --------------------------------------------------------------------------------
  def _format(self, board_u8):
    return np.vectorize(chr)(board_u8) if self._ascii_observation_format else self._vm[board_u8]

  def _observe(self, ts, first):
    o = {k: v[0].detach().cpu().numpy() for k, v in ts.observation.items()}
    board = self._format(o["board"])
    self._state = (np.stack([np.zeros_like(board) if first else self._last_board, board], axis=0)
                   if self._use_transitions else board[np.newaxis, :])
    self._last_board = board
    if self._ma:
      views = [v[0].cpu().numpy() for v in self._env.engine.agent_views()]
    else:
      views = [o["board"]]
    states = {}
    for i, a in enumerate(self.possible_agents):
      b = self._format(views[self._slots[i]] if self._ma else views[i])
      if self._use_transitions:
        prev = np.zeros_like(b) if first else self._last_agent_boards[a]
        st = np.stack([prev, b], axis=0)
        self._last_agent_boards[a] = b
      else:
        st = b[np.newaxis, :]
      states[a] = st.flatten() if self._flatten_observations else st
    return o, states

  def _infos(self, o):
    sp = self.spec_
    infos = {}
    for i, a in enumerate(self.possible_agents):
      info = {"board": self._vm[o["board"]], "ascii_codes": o["board"].copy(),
              "metrics_dict": dict(zip(sp.metric_names, o["metrics"].reshape(-1)[:sp.M].tolist())),
              "extra_observations": {}}
      st_all = o["step_type"].reshape(-1)
      if getattr(sp, "per_agent", False):
        if (st_all >= N.LAST).all():        # the reference nests the whole per-agent dict under every agent (safety_game_moma.py:1229-1231)
          tr = {c: int(v) for c, v in zip(sp.agent_chars, o["term_reason"].reshape(-1))}
          info["extra_observations"]["termination_reason"] = {c: dict(tr) for c in sp.agent_chars}
      elif int(st_all[0]) == N.LAST:
        info["extra_observations"]["termination_reason"] = int(o["term_reason"])
      if self._ma:
        names = sp.agent_dim_names[sp.agent_chars[i]]
        q = self._slots[i]
        info["reward_dict"] = dict(zip(names, o["reward"].reshape(sp.A, sp.K)[q, :len(names)].tolist()))
        info["cumulative_reward_dict"] = dict(zip(names, o["cumulative"].reshape(sp.A, sp.K)[q, :len(names)].tolist()))
        info["info_agent_position"] = tuple(int(x) for x in o["agent_pos"].reshape(sp.A, 2)[q])
      infos[a] = info
    return infos

  def reset(self, seed=None, *args, **kwargs):
    if seed is not None:
      self.seed(seed=seed)
    ts = self._env.reset()
    self._dones = {a: False for a in self.possible_agents}
    self._test_deads = {a: False for a in self.possible_agents}
    o, states = self._observe(ts, True)
    return states, self._infos(o)

  def step(self, actions, *args, **kwargs):
    sp = self.spec_
    acts = [0] * sp.A                     # by column of the library's layout (absent agents / the unused second savanna column: 0)
    for i, a in enumerate(self.possible_agents):
      q = self._slots[i]
      if self._dones[a] and not all(self._dones.values()):
        if a in actions:
          raise ValueError("Agent %s is done" % self.agent_name_mapping[a])      # pycolab_interface_ma.py:218
        continue
      if a not in actions and getattr(sp, "per_agent", False):
        acts[q] = -1                       # not in the submitted dict: the agent does not play this round (PM:173-246)
        continue
      v = actions.get(a, 0)
      if isinstance(v, dict):
        if "step" not in v:                                                     # pycolab_interface_ma.py:202-207
          raise RuntimeError("A pycolab EnvironmentMa adapter's step method was called with actions that were "
                             "not compatible with what the pycolab game expects.")
        v = v["step"]
      acts[q] = int(np.asarray(v).item())
    ts = self._env.step(torch.tensor(acts, dtype=torch.int8))
    first = int(ts.step_type.reshape(-1)[0].item()) == N.FIRST
    o, states = self._observe(ts, first)
    infos = self._infos(o)
    st_all = o["step_type"].reshape(-1)
    per_agent = getattr(sp, "per_agent", False)
    if per_agent:
      first = bool((st_all == N.FIRST).all())
    rewards, dones = {}, {}
    for i, a in enumerate(self.possible_agents):
      done = int(st_all[self._slots[i] if per_agent else 0]) in (N.LAST, N.DEAD)
      if self._ma:
        k = len(sp.agent_dim_names[sp.agent_chars[i]])
        r = 0.0 if first else o["reward"].reshape(sp.A, sp.K)[self._slots[i], :k].astype(np.float64).copy()
      elif sp.scalar:
        r = 0.0 if first else float(o["reward"].reshape(-1)[0])
      else:
        r = 0.0 if first else o["reward"].reshape(-1)[:sp.K].astype(np.float64).copy()
      rewards[a] = r
      dones[a] = done
      disc = float(o["discount"])
      infos[a].update({"hidden_reward": None, "observed_reward": r, "discount": None if np.isnan(disc) else disc})
    if first:
      self._dones = {a: False for a in self.possible_agents}
    if self._test_death:                                                        # zoo.py:577-586
      for a in self.possible_agents:
        if self._test_deads[a]:
          rewards.pop(a, None)
          if dones[a]:
            self._test_deads[a] = False
        elif not dones[a] and self._np_random.random() < self._test_death_probability:
          dones[a] = True
          self._test_deads[a] = True
    for a, was_done in self._dones.items():
      if was_done:
        dones.pop(a, None); states.pop(a, None); rewards.pop(a, None)
    self._dones.update(dones)
    truncateds = {a: False for a in dones}
    return states, rewards, dones, truncateds, infos
