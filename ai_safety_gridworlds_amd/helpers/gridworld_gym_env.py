"""GridworldGymEnv: the reference Gym wrapper's surface (helpers/gridworld_gym_env.py:99-750) over the
batched HIP engine.

    env = GridworldGymEnv("island_navigation_ex", level=9)          # one logical env (batch of 1)
    state, info = env.reset()
    state, reward, terminated, truncated, info = env.step(action)

One logical env is a view over an engine batch of 1; every step is still a HIP kernel launch through
the C ABI (there is no CPU path).  For throughput use `GridworldVectorEnv(env_name, num_envs=N)`,
which returns device tensors for all envs at once.

Kept from the reference: `state = board[np.newaxis]` float32 (gym_env.py:525-536), `[2,H,W]` with
use_transitions, flatten_observations, reward type (ndarray float64[K] for MO envs, python number for the
original envs, 0.0 at reset), `terminated = step_type.last()` and `truncated = False` always
(gym_env.py:563-578), the info keys of gym_env.py:397-450 + hidden_reward / observed_reward / discount
(498-522) including the hidden-reward delta that is never reset between episodes (Q5), fresh copies
instead of the renderer's aliased buffer (Q16).  gymnasium/gym are not required (they are not installed
in this image); if gymnasium is importable the class derives from gymnasium.Env.
"""
import numpy as np
import torch

from .. import _native as N
from ..environments import BatchedSafetyEnvironment, StepType, TerminationReason

try:                                    # optional: only used as a base class / for spaces
  import gymnasium as _gym
  _Base = _gym.Env
except Exception:                       # pragma: no cover - gymnasium absent in this image
  _gym = None
  _Base = object

INFO_HIDDEN_REWARD = "hidden_reward"
INFO_OBSERVED_REWARD = "observed_reward"
INFO_DISCOUNT = "discount"
DIRECTION_UP = 2                        # safety_game_mo_base.py:62-73 Directions.UP


class DiscreteActionSpace(object):
  """Minimal Discrete(n, start) stand-in (gym_env.py:861-896) when gymnasium is absent."""

  def __init__(self, lo, n, np_random):
    self.start, self.n, self._rng = int(lo), int(n), np_random
    self.shape, self.dtype = (), np.int64

  def sample(self):
    return int(self.start + self._rng.integers(self.n))

  def contains(self, x):
    return self.start <= int(x) < self.start + self.n

  __contains__ = contains


class MultiDiscreteActionSpace(object):
  """MultiDiscrete(nvec=[n], start=[lo]) stand-in (use_multi_discrete_action_space=True: gym_env.py:220-221, 753-830): one
  component, so an action is an int32 array of shape (1,); sample() draws as gymnasium's MultiDiscrete does,
  (random(nvec.shape) * nvec).astype(dtype) + start."""

  def __init__(self, lo, n, np_random):
    self.start, self.nvec, self._rng = np.array([int(lo)], np.int32), np.array([int(n)], np.int32), np_random
    self.n, self.min_action, self.max_action = int(n), int(lo), int(lo) + int(n) - 1
    self.shape, self.dtype = (1,), np.int32

  def sample(self, mask=None):
    return (self._rng.random(self.nvec.shape) * self.nvec).astype(self.dtype) + self.start

  def contains(self, x):
    x = np.asarray(x)
    return x.shape == self.shape and bool(((x >= self.start) & (x < self.start + self.nvec)).all())

  __contains__ = contains


class BoxObservationSpace(object):
  def __init__(self, shape, low, high, dtype=np.float32):
    self.shape, self.low, self.high, self.dtype = tuple(shape), low, high, dtype

  def contains(self, x):
    x = np.asarray(x)
    return x.shape == self.shape and (x >= self.low).all() and (x <= self.high).all()

  __contains__ = contains


class GridworldGymEnv(_Base):
  metadata = {"render.modes": ["human", "ansi", "rgb_array"]}
  reward_range = (-float("inf"), float("inf"))

  def __init__(self, env_name, use_transitions=False, render_animation_delay=0.1, flatten_observations=False,
               ascii_observation_format=True, object_coordinates_in_observation=True, layers_in_observation=True,
               occlusion_in_layers=False, layers_order_in_cube=[], ascii_attributes_format=False,
               attribute_coordinates_in_observation=True, layers_in_attribute_observation=False,
               occlusion_in_atribute_layers=False, observable_attribute_categories=None,
               observable_attribute_value_mapping=None, use_multi_discrete_action_space=False, agent_character=None,
               np_random=None, seed=None, pre_reset_callback=None, post_reset_callback=None, pre_step_callback=None,
               post_step_callback=None, render_mode=None, device="cuda:0", **kwargs):
    self.render_mode = render_mode
    self._env_name = env_name
    # experiment bookkeeping of SafetyEnvironmentMo.__init__ (safety_game_mo.py:181-186, 338-385): per wrapper instance here
    # (the reference keeps them in CLASS attributes, i.e. per process)
    self._env_layout_seed = int(kwargs.pop("env_layout_seed", kwargs.pop("trial_no", None) or 1) or 1)
    ep = kwargs.pop("episode_no", None)
    self._episode_no = 1 if ep is None else int(ep)
    self._env_seed = self._derive_env_seed(seed, self._env_layout_seed)
    self._q_value_per_action = None
    self._env = BatchedSafetyEnvironment(env_name, num_envs=1, device=device, **kwargs)
    self.spec_ = self._env.spec
    # multi-agent env behind the single-agent wrapper (gym_env.py:182-189, 476-479): ONE agent is controlled -- the given
    # `agent_character` or the first player -- and stepped alone ({agent: action}); the state is that agent's window
    self._ma = bool(getattr(self.spec_, "per_agent", False)) or self.spec_.family == N.FIREMAKER_EX_MA
    cfg = getattr(self.spec_, "config", {}) or {}
    self._fixed_directions = (self.spec_.family == N.FIREMAKER_EX_MA and not cfg.get("action_direction_mode", 0)
                              and not cfg.get("observation_direction_mode", 0))      # firemaker's default: direction mode 0
    if self.spec_.A > 1 and not self._ma:
      raise NotImplementedError("%s: the batched engine plays whole rounds of this env (use GridworldZooParallelEnv)" % env_name)
    if self._ma:
      chars = list(self.spec_.agent_chars)
      slots = list(getattr(self.spec_, "agent_slots", range(len(chars))))     # the library's column of each agent character
      self._agent_chr = agent_character if agent_character is not None else chars[0]
      self._agent_index = slots[chars.index(self._agent_chr)]
      self._ma_ascii = bool(ascii_observation_format)
      self._vm = np.array([self.spec_.native.value_map[i] for i in range(128)], np.float32)
      self._seed_env(seed)
    self._use_transitions = use_transitions
    self._flatten_observations = flatten_observations
    self._layers_in_observation = layers_in_observation
    self._object_coordinates_in_observation = object_coordinates_in_observation
    self._occlusion_in_layers = occlusion_in_layers
    self._layers_order_in_cube = layers_order_in_cube          # None: no cube; []: every layer, sorted (safety_game_mo.py:460-485)
    self._ascii_observation_format = False      # non-MoMa envs force the float board (gym_env.py:191)
    self._pre_reset_callback, self._post_reset_callback = pre_reset_callback, post_reset_callback
    self._pre_step_callback, self._post_step_callback = pre_step_callback, post_step_callback
    self._last_board = None
    self._state = None
    self._rgb = None
    self._last_hidden_reward = 0.0              # never reset between episodes (Q5, gym_env.py:194)
    self._cumulative_reward = 0.0
    self._internal_np_random = np_random if np_random is not None else np.random.Generator(
        np.random.PCG64(np.random.SeedSequence(seed)))
    sp = self.spec_
    self._action_space = (MultiDiscreteActionSpace if use_multi_discrete_action_space else DiscreteActionSpace)(
        sp.action_lo, sp.n_actions, self._internal_np_random)
    vals = list(sp.value_mapping.values())
    shape = (2 if use_transitions else 1, sp.H, sp.W)
    if flatten_observations:
      shape = (int(np.prod(shape)),)
    self._observation_space = BoxObservationSpace(shape, min(vals), max(vals))

  # ---- gym surface ----------------------------------------------------------------------------
  @property
  def action_space(self):
    return self._action_space

  @property
  def observation_space(self):
    return self._observation_space

  def seed(self, seed=None):                     # gym_env.py:706-712
    self._internal_np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    self._action_space._rng = self._internal_np_random
    if self._ma:
      self._seed_env(seed)

  def _seed_env(self, seed):                     # environment_data[NP_RANDOM] = seeding.np_random(seed)[0]
    if getattr(self.spec_, "needs_rng", False) or self.spec_.family == N.FIREMAKER_EX_MA:
      st = np.random.PCG64(np.random.SeedSequence(seed)).state["state"]
      m = (1 << 64) - 1
      self._env.engine.set_rng_state(np.array([[st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m]],
                                              dtype=np.uint64))

  def close(self):
    self._env.close()

  def get_step_no(self):
    return int(self._env._last["frame"][0].item())

  # ---- experiment bookkeeping (gym_env.py:677-716 -> safety_game_mo.py:1230-1258) -----------------
  @staticmethod
  def _derive_env_seed(seed, env_layout_seed):
    """env_seed as SafetyEnvironmentMo.__init__ leaves it when a new layout starts (safety_game_mo.py:338, 358-385): the
    given seed (32 bits), else crc32(original seed, layout seed, 17122023) -- with no original seed: the layout seed."""
    if seed is None:
      return int(env_layout_seed)
    return int(seed) & 0xFFFFFFFF

  def get_reward_unit_space(self):
    """[min unit reward per enabled dimension, max ...] over the enabled reward flags (mo_reward.py:150-181); the original
    scalar envs have no such method in the reference either."""
    sp = self.spec_
    if sp.scalar:
      raise AttributeError("'%s' environment has no attribute 'get_reward_unit_space'" % sp.name)
    space = getattr(sp, "reward_unit_space", None)
    return None if space is None else [np.array(space[0], dtype=np.float64), np.array(space[1], dtype=np.float64)]

  def get_env_seed(self):
    return self._env_seed

  def get_env_layout_seed(self):
    return self._env_layout_seed

  def get_trial_no(self):                        # obsolete alias (gym_env.py:693-694)
    return self.get_env_layout_seed()

  def get_episode_no(self):
    return self._episode_no

  def get_next_episode_no(self):                 # safety_game_mo.py:1246-1253: + 1 once the running episode has a step
    played = self._env._last is not None and int(self._env._last["step_type"].reshape(-1)[0].item()) != N.FIRST
    return self._episode_no + (1 if played else 0)

  def set_current_q_value_per_action(self, q_value_per_action):   # kept for the step logger (safety_game_mo.py:1257-1258)
    self._q_value_per_action = q_value_per_action

  # ---- helpers --------------------------------------------------------------------------------
  def _host(self, ts):
    o = {k: v[0].detach().cpu().numpy() for k, v in ts.observation.items()}
    return o

  def _dim_names(self):
    """The controlled agent's reward dimensions (firemaker's workers and supervisor have different ones)."""
    sp = self.spec_
    if self._ma and getattr(sp, "agent_dim_names", None):
      return list(sp.agent_dim_names[self._agent_chr])
    return list(sp.dim_names)

  def _compute_info(self, o, first):
    sp = self.spec_
    K = sp.K
    ai = self._agent_index if self._ma else 0
    names = self._dim_names()
    reward = o["reward"].reshape(-1)[ai * K:ai * K + len(names)]
    cum = o["cumulative"].reshape(-1)[ai * K:ai * K + len(names)]
    frame = int(o["frame"])
    tri = ai if o["term_reason"].size > 1 else 0                # per agent only where agents finish one by one
    extra = {}
    if int(o["actual_action"].reshape(-1)[0]) >= 0:
      extra["actual_actions"] = int(o["actual_action"].reshape(-1)[0])
    if int(o["step_type"].reshape(-1)[ai]) == N.LAST:
      extra["termination_reason"] = TerminationReason(int(o["term_reason"].reshape(-1)[tri]))
    info = {
        "observation_direction": None, "action_direction": DIRECTION_UP,
        "board": o["obs_board"].copy(), "ascii_codes": o["board"].copy(),
        "ascii": np.vectorize(chr)(o["board"]), "extra_observations": extra,
    }
    if not sp.scalar:                           # safety_game_mo.py:1012-1084
      info["reward_dict"] = dict(zip(names, reward.tolist()))
      info["cumulative_reward_dict"] = dict(zip(names, cum.tolist()))
      metrics = o["metrics"].reshape(-1)[:sp.M]
      info["metrics_dict"] = dict(zip(sp.metric_names, metrics.tolist()))
      mm = np.empty([sp.M, 2], object)
      for i, name in enumerate(sp.metric_names):
        mm[i, 0], mm[i, 1] = name, metrics[i]
      info["metrics_matrix"] = mm
      # derived statistics of _process_timestep (safety_game_mo.py:1027-1084): computed ON THE DEVICE in numpy's summation
      # order (sgw_derived_stats); only the values come to the host
      ds = self._env.engine.derived_stats()
      pick = (lambda t: t[0, ai] if self._ma else t[0])
      info["cumulative_reward"] = cum.copy()
      info["average_reward"] = pick(ds["average_reward"]).cpu().numpy()[:len(names)].astype(np.float64)
      for key in ("gini_index", "cumulative_gini_index", "mo_variance", "cumulative_mo_variance", "average_mo_variance"):
        info[key] = np.float64(pick(ds[key]).item())
      lay = None
      if self._layers_in_observation or self._object_coordinates_in_observation or self._layers_order_in_cube is not None:
        lay = self._env.engine.observe_layers()[0].cpu().numpy().astype(bool)
        layers = {c: lay[i] for i, c in enumerate(sp.layer_chars)}
      if self._layers_in_observation:
        info["info_observation_layers_dict"] = layers
      if self._object_coordinates_in_observation:                 # calculate_observation_coordinates (safety_game_mo.py:422-457)
        if self._occlusion_in_layers:
          raise NotImplementedError("info_observation_coordinates with occlusion_in_layers=True: the reference's branch "
                                    "(safety_game_mo.py:443-457) raises NameError at this snapshot")
        info["info_observation_coordinates"] = {c: [tuple(x) for x in np.argwhere(layers[c]).tolist()] for c in layers}
      if self._layers_order_in_cube is not None:                  # get_layers_order / calculate_observation_layers_cube (:460-520)
        order = list(self._layers_order_in_cube)
        ascii_board = np.vectorize(chr)(o["board"])
        if order == []:
          order = sorted(layers.keys()) if not self._occlusion_in_layers else sorted(np.unique(ascii_board).tolist())
        info["info_observation_layers_order"] = order
        if not self._occlusion_in_layers:                           # absent layers read as zeros (cross-environment cubes)
          zero = np.zeros_like(next(iter(layers.values())))
          info["info_observation_layers_cube"] = np.stack([layers.get(c, zero) for c in order], axis=0)
        else:
          info["info_observation_layers_cube"] = np.stack([ascii_board == c for c in order], axis=0)
      if self._ma:                                                # gym_env.py:373-384, 428-439: the controlled agent's own window
        ch = self._agent_chr
        flags = int(o["agent_flags"].reshape(-1)[ai])
        if not self._fixed_directions:
          info["observation_direction"], info["action_direction"] = (flags >> 3) & 3, (flags >> 1) & 3
        view = self._env.engine.agent_views()[ai][0].cpu().numpy()
        info["info_agent_observations"] = np.vectorize(chr)(view) if self._ma_ascii else self._vm[view]
        if lay is not None:
          cube = self._env.engine.agent_layer_views()[ai][0].cpu().numpy().astype(bool)
          al = {c: cube[j] for j, c in enumerate(sp.layer_chars)}
          if self._layers_in_observation:
            info["info_agent_observation_layers_dict"] = al
          if self._object_coordinates_in_observation:             # calculate_agents_observation_coordinates (safety_game_moma.py:528-580)
            me = np.argwhere(al[ch]) if ch in al else []
            if len(me) > 0:
              ay, ax = int(me[0][0]), int(me[0][1])
              info["info_agent_observation_coordinates"] = {c: [(int(x) - ax, int(y) - ay) for y, x in np.argwhere(al[c]).tolist()] for c in al}
            else:
              info["info_agent_observation_coordinates"] = []
          if self._layers_order_in_cube is not None:              # the SAME order parameter as the global cube (gym_env.py:382-384)
            order = list(self._layers_order_in_cube) or sorted(al.keys())
            zero = np.zeros_like(cube[0])
            info["info_agent_observation_layers_order"] = order
            info["info_agent_observation_layers_cube"] = np.stack([al.get(c, zero) for c in order], axis=0)
    if sp.name == "island_navigation_ex":
      info["safety"] = int(o["safety"])
    return info

  def _state_from(self, o, first):
    board = o["obs_board"].copy()                # fresh copy (gym_env.py:525, Q16)
    if self._ma:                                 # the controlled agent's window, ascii by default (gym_env.py:537-553)
      view = self._env.engine.agent_views()[self._agent_index][0].cpu().numpy()
      board = np.vectorize(chr)(view) if self._ma_ascii else self._vm[view]
    if self._use_transitions:
      prev = np.zeros_like(board) if first else self._last_board
      state = np.stack([prev, board], axis=0)
      self._last_board = board
    else:
      state = board[np.newaxis, :]
    if self._flatten_observations:
      state = state.flatten()
    self._state = state
    return state

  def reset(self, seed=None, return_info=False, *args, **kwargs):
    if self._pre_reset_callback is not None:
      (allow_reset, seed, args, kwargs) = self._pre_reset_callback(seed, *args, **kwargs)
      if not allow_reset:
        return None
    if seed is not None:
      self.seed(seed=seed)
    # safety_game_mo.py:656-704: a reset after a played episode advances the episode counter; a new env_layout_seed restarts it
    layout = kwargs.pop("env_layout_seed", kwargs.pop("trial_no", None))
    if isinstance(kwargs.get("options"), dict):
      layout = kwargs["options"].get("env_layout_seed", kwargs["options"].get("trial_no", layout))
    if layout is not None and int(layout) != self._env_layout_seed:
      self._env_layout_seed, self._episode_no = int(layout), 1
      self._env_seed = self._derive_env_seed(seed, self._env_layout_seed)
    elif self._env._last is not None and int(self._env._last["step_type"].reshape(-1)[0].item()) != N.FIRST:
      self._episode_no += 1
    ts = self._env.reset()
    o = self._host(ts)
    info = self._compute_info(o, True)
    state = self._state_from(o, True)
    self._cumulative_reward = 0.0
    result = (state, info)
    if self._post_reset_callback is not None:
      self._post_reset_callback(*result)
    return result

  def step(self, action, *args, **kwargs):
    if self._pre_step_callback is not None:
      action = self._pre_step_callback(action, *args, **kwargs)
    a = np.asarray(action)
    if a.size != 1:                              # pycolab_interface_mo.py:168-171
      raise RuntimeError("A pycolab Environment adapter's step method was called with actions that were "
                         "not compatible with what the pycolab game expects.")
    sp = self.spec_
    acts = [int(a.reshape(-1)[0])]
    if self._ma:                                 # {agent: action}: the other agents are not in the dict (gym_env.py:476-479)
      acts = [-1] * sp.A
      acts[self._agent_index] = int(a.reshape(-1)[0])
    ts = self._env.step(torch.tensor(acts, dtype=torch.int8))
    o = self._host(ts)
    ai = self._agent_index if self._ma else 0
    first = int(o["step_type"].reshape(-1)[ai]) == N.FIRST         # auto-reset step: reward None -> 0.0
    info = self._compute_info(o, first)
    if first:
      reward = 0.0
    elif sp.scalar:
      r = float(o["reward"].reshape(-1)[0])
      reward = int(r) if r == int(r) else r
    else:
      reward = o["reward"].reshape(-1)[ai * sp.K:ai * sp.K + len(self._dim_names())].astype(np.float64).copy()
    if sp.scalar:                                # gym_env.py:498-505
      cumulative_hidden = float(o["hidden"])
      hidden_reward = cumulative_hidden - self._last_hidden_reward
      self._last_hidden_reward = cumulative_hidden
    else:
      hidden_reward = None
    disc = float(o["discount"])
    info.update({INFO_HIDDEN_REWARD: hidden_reward, INFO_OBSERVED_REWARD: reward,
                 INFO_DISCOUNT: None if np.isnan(disc) else disc})
    state = self._state_from(o, first)
    done = int(o["step_type"].reshape(-1)[ai]) == N.LAST
    self._cumulative_reward = self._cumulative_reward + reward
    result = (state, reward, done, False, info)
    if self._post_step_callback is not None:
      self._post_step_callback(action, *result, *args, **kwargs)
    return result

  def render(self, mode="rgb_array"):
    """rgb_array: uint8 [3, H, W] of the current board (gym_env.py:718-750; human/ansi UIs out of scope)."""
    rgb = self._env.engine.observe(rgb=True)["RGB"][0].cpu().numpy()
    if mode == "rgb_array":
      return rgb
    if mode == "ansi":
      b = self._env._last["board"][0].cpu().numpy()
      return "\n".join("".join(chr(c) for c in row) for row in b)
    raise NotImplementedError("render mode %r (curses/pyplot viewers are out of scope)" % mode)


class GridworldVectorEnv(object):
  """N envs at once; observations/rewards/dones stay on the device as torch tensors.

  reset() -> (obs [N,1,H,W] float32, info);  step(actions int8/int64 [N]) ->
  (obs, reward [N,K] float64 (or [N] for scalar envs), terminated bool [N], truncated bool [N], info).

  full_info=True: info carries everything the reference's env.step() computes per step (gridworld_gym_env.py:455-585,
  safety_game_mo.py:971-1107, observation_distiller_ex.py:147-187) as device tensors -- "RGB" uint8 [N, 3, H, W], "layers"
  uint8 [N, L, H, W] (unoccluded, `layers_order`), "cumulative_reward", "average_reward", "gini_index",
  "cumulative_gini_index", "mo_variance", "cumulative_mo_variance", "average_mo_variance", "metrics", "safety",
  "last_performance" / "performance_sum" / "episodes" -- produced by ONE library call per step (sgw_step_full: step kernel +
  RGB + layers + derived statistics + performance bookkeeping chained in C, replayed as one hipGraph)."""

  FULL_OUTPUTS = ("board", "obs_board", "reward", "cumulative", "step_type", "term_reason", "hidden", "frame", "metrics", "safety",
                  "discount", "actual_action")

  def __init__(self, env_name, num_envs, device="cuda:0", env_id_base=0,
               outputs=("board", "obs_board", "reward", "cumulative", "step_type", "term_reason", "hidden"), full_info=False,
               replay_graph=False, **kwargs):
    self._full = bool(full_info)
    self._replay = bool(replay_graph)       # full_info: one hipGraph launch per step (less host time, more GPU time: sgw_extras.replay)
    if self._full:
      outputs = tuple(dict.fromkeys(tuple(outputs) + self.FULL_OUTPUTS))
    else:
      outputs = tuple(dict.fromkeys(tuple(outputs) + ("done",)))       # `terminated` comes out of the step launch itself
    self._env = BatchedSafetyEnvironment(env_name, num_envs=num_envs, device=device, env_id_base=env_id_base,
                                         outputs=outputs, track_performance=False, **kwargs)
    self.spec_ = self._env.spec
    self.num_envs = num_envs
    self.layers_order = list(self.spec_.layer_chars)
    self._never = torch.zeros(num_envs, dtype=torch.bool, device=self._env.device)      # truncated: always False (gym_env.py:563-578)
    self._done = torch.zeros(num_envs, dtype=torch.bool, device=self._env.device)
    self._info = None
    self._lean = None

  def _pack(self, ts):
    o = ts.observation
    obs = o["obs_board"].unsqueeze(1)
    info = {"step_type": ts.step_type, "term_reason": o.get("term_reason"), "board": o.get("board"),
            "cumulative": o.get("cumulative"), "hidden": o.get("hidden")}
    return obs, info

  def reset(self, mask=None):
    return self._pack(self._env.reset(mask))

  def step(self, actions):
    if actions.dtype != torch.int8:
      actions = actions.to(torch.int8)
    if self._full:
      return self._step_full(actions)
    # a step from Python is host-bound: the outputs live in persistent buffers (and `terminated` is the launch's `done` output),
    # so the returned views are built once and a step is the library call alone (12.3 -> 8.9 us per step at 65 536 envs)
    o = self._env.engine.step(actions)
    self._env._last = o
    c = self._lean
    if c is None or c["_for"] is not o["step_type"]:
      st = o["step_type"].reshape(self.num_envs, -1)[:, 0]
      c = self._lean = {"_for": o["step_type"], "obs": o["obs_board"].unsqueeze(1),
                        "reward": o["reward"] if not self.spec_.scalar else o["reward"].reshape(-1),
                        "done": o["done"].reshape(self.num_envs, -1)[:, 0].view(torch.bool),
                        "info": {"step_type": st, "term_reason": o.get("term_reason"), "board": o.get("board"),
                                 "cumulative": o.get("cumulative"), "hidden": o.get("hidden")}}
    return c["obs"], c["reward"], c["done"], self._never, dict(c["info"])

  def _step_full(self, actions):
    """A step from Python is host-bound: the outputs live in persistent buffers, so the info dict of views is built once
    and a step is one library call plus one comparison."""
    o = self._env.engine.step_full(actions, rgb=True, layers=hasattr(self.spec_, "drape_chars"), stats=not self.spec_.scalar, performance=True,
                                   replay=self._replay)
    self._env._last = o
    if self._info is None or self._info["_for"] is not o["step_type"]:
      st = o["step_type"].reshape(self.num_envs, -1)[:, 0]
      info = {"step_type": st, "term_reason": o["term_reason"], "board": o["board"], "cumulative": o["cumulative"], "hidden": o["hidden"],
              "cumulative_reward": o["cumulative"], "RGB": o["RGB"], "layers": o.get("layers"), "layers_order": self.layers_order,
              "metrics": o["metrics"][:, :self.spec_.M], "safety": o["safety"], "frame": o["frame"], "discount": o["discount"],
              "actual_action": o["actual_action"], "last_performance": o["last_performance"], "performance_sum": o["performance_sum"],
              "episodes": o["episodes"]}
      for k in ("gini_index", "cumulative_gini_index", "mo_variance", "cumulative_mo_variance", "average_mo_variance", "average_reward"):
        if k in o:
          info[k] = o[k]
      self._info = {"_for": o["step_type"], "info": info, "done": o["done"], "obs": o["obs_board"].unsqueeze(1),
                    "reward": o["reward"] if not self.spec_.scalar else o["reward"].reshape(-1)}
    c = self._info
    return c["obs"], c["reward"], c["done"], self._never, dict(c["info"])

  def get_overall_performance(self):
    """[N, C] mean performance over each env's finished episodes (NaN = none yet); needs full_info=True (safety_game.py:194-208)."""
    o = self._env._last
    if o is None or "performance_sum" not in o:
      return None
    return o["performance_sum"] / o["episodes"].to(torch.float64)[:, None]

  def close(self):
    self._env.close()
