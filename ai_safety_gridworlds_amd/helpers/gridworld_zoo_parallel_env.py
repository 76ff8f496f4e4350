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

from .gridworld_gym_env import DiscreteActionSpace, MultiDiscreteActionSpace, BoxObservationSpace

OUTS = ("board", "obs_board", "reward", "cumulative", "step_type", "term_reason", "discount", "metrics", "frame",
        "agent_pos", "agent_flags", "hidden", "actual_action")

# info keys of the reference wrapper (gridworld_zoo_parallel_env.py:53-64; pycolab_interface_ma.py:37-39)
INFO_OBSERVED_REWARD = "observed_reward"
INFO_DISCOUNT = "discount"
INFO_OBSERVATION_DIRECTION = "observation_direction"
INFO_ACTION_DIRECTION = "action_direction"
INFO_OBSERVATION_COORDINATES = "info_observation_coordinates"
INFO_OBSERVATION_LAYERS_DICT = "info_observation_layers_dict"
INFO_OBSERVATION_LAYERS_ORDER = "info_observation_layers_order"
INFO_OBSERVATION_LAYERS_CUBE = "info_observation_layers_cube"
INFO_AGENT_OBSERVATIONS = "info_agent_observations"
INFO_AGENT_OBSERVATION_COORDINATES = "info_agent_observation_coordinates"
INFO_AGENT_OBSERVATION_LAYERS_DICT = "info_agent_observation_layers_dict"
INFO_AGENT_OBSERVATION_LAYERS_ORDER = "info_agent_observation_layers_order"
INFO_AGENT_OBSERVATION_LAYERS_CUBE = "info_agent_observation_layers_cube"


class GridworldZooParallelEnv(_Base):
  metadata = {"render.modes": ["human", "ansi", "rgb_array"], "name": "ai_safety_gridworlds_amd"}

  def __init__(self, env_name, use_transitions=False, render_animation_delay=0.1, flatten_observations=False,
               ascii_observation_format=True, object_coordinates_in_observation=True, layers_in_observation=True,
               occlusion_in_layers=False, layers_order_in_cube=[], layers_order_in_cube_per_agent={},
               ascii_attributes_format=False, attribute_coordinates_in_observation=True, layers_in_attribute_observation=False,
               occlusion_in_atribute_layers=False, observable_attribute_categories=None, observable_attribute_value_mapping=None,
               use_multi_discrete_action_space=False, np_random=None, seed=None, test_death=False, test_death_probability=0.33,
               pre_reset_callback=None, post_reset_callback=None, pre_step_callback=None, post_step_callback=None,
               render_mode=None, device="cuda:0", **kwargs):
    """Same parameters as the reference wrapper (gridworld_zoo_parallel_env.py:100-135).  The observable-attribute parameters are
    accepted and unused (no environment of the reference defines observable attributes)."""
    if occlusion_in_layers:
      raise NotImplementedError("occlusion_in_layers=True: the reference's branch (safety_game_moma.py:607-618) raises NameError "
                                "at this snapshot")
    self.render_mode = render_mode
    self._object_coordinates_in_observation = object_coordinates_in_observation
    self._layers_in_observation = layers_in_observation
    self._occlusion_in_layers = occlusion_in_layers
    self._layers_order_in_cube = layers_order_in_cube
    self._layers_order_in_cube_per_agent = layers_order_in_cube_per_agent
    self._observable_attribute_categories = (["expression", "action_direction", "observation_direction", "numeric_message", "public_metrics"]
                                             if observable_attribute_categories is None else list(observable_attribute_categories))
    self._pre_reset_callback, self._post_reset_callback = pre_reset_callback, post_reset_callback
    self._pre_step_callback, self._post_step_callback = pre_step_callback, post_step_callback
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
    space = MultiDiscreteActionSpace if use_multi_discrete_action_space else DiscreteActionSpace       # zoo.py:225-228
    self._action_spaces = {a: space(sp.action_lo, sp.n_actions, self._np_random) for a in self.possible_agents}
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

  def _layer_infos(self, o):
    """The observation-derived entries of `_process_observation` (zoo.py:286-316): computed once per step, shared by the agents."""
    sp = self.spec_
    out = {}
    need_layers = (self._layers_in_observation or self._object_coordinates_in_observation or self._layers_order_in_cube is not None
                   or (self._ma and self._layers_order_in_cube_per_agent is not None))
    if sp.scalar or not need_layers:
      return out
    lay = self._env.engine.observe_layers()
    glob = lay[0].cpu().numpy().astype(bool)
    layers = {c: glob[i] for i, c in enumerate(sp.layer_chars)}                 # observation['layers'] (unoccluded)
    out["layers"] = layers
    if self._object_coordinates_in_observation:                                  # safety_game_moma.py:583-603
      out["coords"] = {c: [tuple(x) for x in np.argwhere(layers[c]).tolist()] for c in layers}
    if self._layers_order_in_cube is not None:                                   # :621-672
      order = list(self._layers_order_in_cube) or sorted(layers.keys())
      zero = np.zeros_like(glob[0])
      out["order"], out["cube"] = order, np.stack([layers.get(c, zero) for c in order], axis=0)
    if self._ma:                                                                 # agent_perspectives_with_layers (:430-525)
      cubes = [v[0].cpu().numpy().astype(bool) for v in self._env.engine.agent_layer_views(layers=lay)]
      out["agent_layers"], out["agent_coords"], out["agent_order"], out["agent_cube"] = {}, {}, {}, {}
      for i, a in enumerate(self.possible_agents):
        ch = self.agent_name_mapping[a]
        al = {c: cubes[self._slots[i]][j] for j, c in enumerate(sp.layer_chars)}
        out["agent_layers"][a] = al
        if self._object_coordinates_in_observation:                              # calculate_agents_observation_coordinates (:528-580)
          me = np.argwhere(al[ch]) if ch in al else []
          if len(me) > 0:
            ay, ax = int(me[0][0]), int(me[0][1])
            out["agent_coords"][a] = {c: [(int(x) - ax, int(y) - ay) for y, x in np.argwhere(al[c]).tolist()] for c in al}
          else:
            out["agent_coords"][a] = []                                          # the agent is not in its own layers
        if self._layers_order_in_cube_per_agent is not None:
          order = list(self._layers_order_in_cube_per_agent.get(a, [])) or sorted(al.keys())
          zero = np.zeros_like(cubes[self._slots[i]][0])
          out["agent_order"][a], out["agent_cube"][a] = order, np.stack([al.get(c, zero) for c in order], axis=0)
    return out

  def _shared_infos(self, o):
    """Observation keys the wrapper passes through to every agent's info unchanged (zoo.py:378-381): for a multi-agent env they
    are dicts over the agent characters (safety_game_moma.py:1255-1379)."""
    sp = self.spec_
    out = {}
    if not self._ma:
      return out
    ds = {k: v[0].cpu().numpy() for k, v in self._env.engine.derived_stats().items()}    # computed on the device, numpy order
    cum = o["cumulative"].reshape(sp.A, sp.K)
    per = lambda f: {c: f(self._slots[i], len(sp.agent_dim_names[c])) for i, c in enumerate(sp.agent_chars)}
    out["cumulative_reward"] = per(lambda q, k: cum[q, :k].astype(np.float64).copy())
    out["average_reward"] = per(lambda q, k: ds["average_reward"][q, :k].astype(np.float64))
    for key in ("gini_index", "cumulative_gini_index", "mo_variance", "cumulative_mo_variance", "average_mo_variance"):
      out[key] = per(lambda q, k, key=key: np.float64(ds[key][q]))
    metrics = o["metrics"].reshape(-1)[:sp.M]
    mm = np.empty([sp.M, 2], object)
    for i, name in enumerate(sp.metric_names):
      mm[i, 0], mm[i, 1] = name, (None if np.isnan(metrics[i]) else metrics[i])
    out["metrics_matrix"] = mm
    # ACTUAL_ACTIONS: what each agent that played this round did (safety_game_ma.py:784-788; no policy wrapper changes actions
    # in the multi-agent envs): the submitted actions of the agents that were alive, nothing on a reset / auto-reset round
    acted = getattr(self, "_played", None)
    if acted:
      out["actual_actions"] = {c: {"step": int(v)} for c, v in acted.items()}
    # observable attributes (observation_distiller_ex.py:104-141): no environment of the reference sets any, so every category
    # is an all-zero board, an empty layer dict and a board of empty strings
    cats = self._observable_attribute_categories
    zero = np.zeros((sp.H, sp.W), np.uint8)
    out["agent_attribute_board_ascii_codes"] = {c: zero.copy() for c in cats}
    out["agent_attribute_layers"] = {c: {} for c in cats}
    out["agent_attribute_board_ascii"] = {c: np.full((sp.H, sp.W), '', dtype='<U1') for c in cats}
    return out

  def _infos(self, o, states=None):
    sp = self.spec_
    infos = {}
    li = self._layer_infos(o)
    shared = self._shared_infos(o)
    flags = o["agent_flags"].reshape(-1)
    for i, a in enumerate(self.possible_agents):
      info = {"board": self._vm[o["board"]], "ascii_codes": o["board"].copy(), "ascii": np.vectorize(chr)(o["board"]),
              "metrics_dict": dict(zip(sp.metric_names, o["metrics"].reshape(-1)[:sp.M].tolist())),
              "extra_observations": {}}
      st_all = o["step_type"].reshape(-1)
      if getattr(sp, "per_agent", False):
        if (st_all >= N.LAST).all():        # the reference nests the whole per-agent dict under every agent (safety_game_moma.py:1229-1231)
          tr = {c: int(v) for c, v in zip(sp.agent_chars, o["term_reason"].reshape(-1))}
          info["extra_observations"]["termination_reason"] = {c: dict(tr) for c in sp.agent_chars}
      elif int(st_all[0]) == N.LAST:
        info["extra_observations"]["termination_reason"] = int(o["term_reason"])
      for k, v in shared.items():
        if k == "actual_actions":
          info["extra_observations"][k] = v
        else:
          info[k] = v
      if self._ma:
        names = sp.agent_dim_names[sp.agent_chars[i]]
        q = self._slots[i]
        # zoo.py:325-335: directions as the env's observation carries them (Directions LEFT=0 RIGHT=1 UP=2 DOWN=3)
        info[INFO_OBSERVATION_DIRECTION] = int((flags[q] >> 3) & 3)
        info[INFO_ACTION_DIRECTION] = int((flags[q] >> 1) & 3)
        info["reward_dict"] = dict(zip(names, o["reward"].reshape(sp.A, sp.K)[q, :len(names)].tolist()))
        info["cumulative_reward_dict"] = dict(zip(names, o["cumulative"].reshape(sp.A, sp.K)[q, :len(names)].tolist()))
        info["info_agent_position"] = tuple(int(x) for x in o["agent_pos"].reshape(sp.A, 2)[q])
      if "coords" in li:
        info[INFO_OBSERVATION_COORDINATES] = li["coords"]                        # shared global observation, returned via every agent
      if self._layers_in_observation and "layers" in li:
        info[INFO_OBSERVATION_LAYERS_DICT] = li["layers"]
      if "cube" in li:
        info[INFO_OBSERVATION_LAYERS_ORDER], info[INFO_OBSERVATION_LAYERS_CUBE] = li["order"], li["cube"]
      if self._ma and states is not None:                                         # zoo.py:348-359
        view = states[a]
        info[INFO_AGENT_OBSERVATIONS] = view[-1] if not self._flatten_observations else view
        if "agent_layers" in li:
          if self._layers_in_observation:
            info[INFO_AGENT_OBSERVATION_LAYERS_DICT] = li["agent_layers"][a]
          if self._object_coordinates_in_observation:
            info[INFO_AGENT_OBSERVATION_COORDINATES] = li["agent_coords"][a]
          if self._layers_order_in_cube_per_agent is not None:
            info[INFO_AGENT_OBSERVATION_LAYERS_ORDER] = li["agent_order"][a]
            info[INFO_AGENT_OBSERVATION_LAYERS_CUBE] = li["agent_cube"][a]
      infos[a] = info
    return infos

  def observe_infos_from_location(self, agents_coordinates, agents_observation_directions=None):
    """zoo.py:395-412.  The reference cannot execute this at this snapshot: get_agent_perspective builds the overridden position
    with `Sprite.Position`, and safety_game_moma.py never imports `Sprite` (NameError at safety_game_moma.py:2000, checked by
    running it).  Kept with the reference's behaviour so that callers see the same failure."""
    raise NameError("name 'Sprite' is not defined")

  def reset(self, seed=None, *args, **kwargs):
    if self._pre_reset_callback is not None:                    # zoo.py:619-622
      (allow_reset, seed, args, kwargs) = self._pre_reset_callback(seed, *args, **kwargs)
      if not allow_reset:
        return
    if seed is not None:
      self.seed(seed=seed)
    ts = self._env.reset()
    self._played = None
    self._dones = {a: False for a in self.possible_agents}
    self._test_deads = {a: False for a in self.possible_agents}
    o, states = self._observe(ts, True)
    result = (states, self._infos(o, states))
    if self._post_reset_callback is not None:                   # zoo.py:699-700: exactly (obs, infos)
      self._post_reset_callback(*result)
    return result

  def step(self, actions, *args, **kwargs):
    if self._pre_step_callback is not None:                     # zoo.py:445-446
      actions = self._pre_step_callback(actions, *args, **kwargs)
    sp = self.spec_
    acts = [0] * sp.A                     # by column of the library's layout (absent agents / the unused second savanna column: 0)
    for i, a in enumerate(self.possible_agents):
      q = self._slots[i]
      if self._dones[a] and not all(self._dones.values()):
        if a in actions:
          raise ValueError("Agent %s is done" % self.agent_name_mapping[a])      # pycolab_interface_ma.py:218
        continue
      if a not in actions and (getattr(sp, "per_agent", False) or sp.family == N.FIREMAKER_EX_MA):
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
    st_host = o["step_type"].reshape(-1)
    all_first = bool((st_host[[self._slots[i] for i in range(len(self.possible_agents))]] == N.FIRST).all()) \
        if getattr(sp, "per_agent", False) else first
    self._played = None if all_first else {self.agent_name_mapping[a]: acts[self._slots[i]] for i, a in enumerate(self.possible_agents)
                                           if not self._dones[a] and acts[self._slots[i]] >= 0}
    infos = self._infos(o, states)
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
    result = (states, rewards, dones, truncateds, infos)
    if self._post_step_callback is not None:
      self._post_step_callback(actions, *result, *args, **kwargs)
    return result
