"""GridworldZooVectorEnv: N lockstep instances of a multi-agent safety gridworld behind the PettingZoo-parallel calling
convention, every array a DEVICE tensor and no host synchronisation on the step path.

    env = GridworldZooVectorEnv("firemaker_ex_ma", num_envs=16384, amount_agents=3, seed=0)
    obs, infos = env.reset()                                   # obs[agent]: uint8 [N, h_a, w_a] ascii codes (cuda)
    obs, rewards, terminateds, truncateds, infos = env.step({agent: int8 [N] cuda tensor, ...})

What one env of the batch returns is what `GridworldZooParallelEnv` (the reference's wrapper surface,
helpers/gridworld_zoo_parallel_env.py:429-615) returns for it:
  obs[agent]         the agent-centric window of the rendered board (get_agent_perspective, safety_game_moma.py:1996-2101):
                     ascii codes uint8 [N, h, w], or the value-mapped float32 board with ascii_observation_format=False
  rewards[agent]     float64 [N, K_agent] in the agent's sorted reward-dimension order (0 at an auto-reset round)
  terminateds[agent] bool [N]: the agent's StepType is LAST or DEAD;  truncateds[agent]: all False
  infos[agent]       device tensors: "step_type" [N], "cumulative_reward" [N, K_agent], "agent_position" [N, 2],
                     "discount" [N] (NaN = None), "metrics" [N, M], "board" uint8 [N, H, W] (the global board, shared),
                     "observation_direction" / "action_direction" uint8 [N] (Directions LEFT=0 RIGHT=1 UP=2 DOWN=3);
                     with layers_in_observation=True also the wrapper's layer cubes as tensors (zoo.py:296-316, 337-359):
                     "info_observation_layers_cube" uint8 [N, L, H, W] (shared) and "info_agent_observation_layers_cube"
                     uint8 [N, L, h_a, w_a], both in `layers_order` = the sorted layer characters
A finished env auto-resets at its next round exactly like the reference adapter (the round's actions are discarded); agents of
the per-agent families that are already LAST/DEAD can be given -1 ("not in the dict").  One round = ONE kernel launch: the
step (shuffled sequential plays, fire spread, rewards, auto-reset) writes the agent windows too (sgw_out.views / obs_views;
families without the in-kernel windows take a second launch, sgw_agent_views).
"""
import numpy as np
import torch

from .. import _native as N
from ..engine import fused_views
from ..environments import BatchedSafetyEnvironment
from ..specs import make_spec

OUTS = ("board", "reward", "cumulative", "step_type", "term_reason", "discount", "metrics", "agent_pos", "agent_flags", "done")


class GridworldZooVectorEnv(object):
  metadata = {"name": "ai_safety_gridworlds_amd_vector"}

  def __init__(self, env_name, num_envs, ascii_observation_format=True, layers_in_observation=False, seed=None, device="cuda:0",
               env_id_base=0, **kwargs):
    self._fused = fused_views(make_spec(env_name, **kwargs))
    self._ascii = bool(ascii_observation_format)
    cfg0 = getattr(make_spec(env_name, **kwargs), "config", None) or {}
    self._turning = bool(cfg0.get("action_direction_mode", 0) or cfg0.get("observation_direction_mode", 0))
    # `terminated` and the decoded directions leave with the step launch (sgw_out.done / obs_dir / act_dir): no torch launch per step
    outs = OUTS + (("obs_dir", "act_dir") if self._turning else ()) + ((("views",) if self._ascii else ("obs_views",)) if self._fused else ())
    self._env = BatchedSafetyEnvironment(env_name, num_envs=num_envs, device=device, env_id_base=env_id_base, outputs=outs,
                                         track_performance=False, **kwargs)
    sp = self.spec_ = self._env.spec
    if sp.A < 2 and not getattr(sp, "per_agent", False):
      raise NotImplementedError("%s is a single-agent env: use GridworldVectorEnv" % env_name)
    self.num_envs = int(num_envs)
    self.device = self._env.device
    self._per_agent = bool(getattr(sp, "per_agent", False))
    self._slots = list(getattr(sp, "agent_slots", range(len(sp.agent_chars))))
    self.possible_agents = ["agent_%s" % c for c in sp.agent_chars]
    self.agent_name_mapping = dict(zip(self.possible_agents, sp.agent_chars))
    self._k = {a: len(sp.agent_dim_names[c]) for a, c in self.agent_name_mapping.items()}
    self._layers = bool(layers_in_observation)
    self.layers_order = list(sp.layer_chars)                   # get_layers_order(...) of the reference: sorted layer keys
    self._vm = torch.tensor([sp.native.value_map[i] for i in range(128)], dtype=torch.float32, device=self.device)
    self._acts = torch.zeros((self.num_envs, sp.A), dtype=torch.int8, device=self.device)
    self._never = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)            # truncated: always False
    self._up = torch.full((self.num_envs,), 2, dtype=torch.uint8, device=self.device)         # Directions.UP: the fixed-direction envs
    if sp.family == N.FIREMAKER_EX_MA or getattr(sp, "needs_rng", False):
      # env i draws from Generator(PCG64(SeedSequence(seed + global id))): what seeding.np_random gives N separately seeded envs
      base = 0 if seed is None else int(seed)
      self._env.engine.set_rng_seeds(base + env_id_base + np.arange(self.num_envs))

  @property
  def agents(self):
    return list(self.possible_agents)

  def observation_shape(self, agent):
    return tuple(self.spec_.view_shapes[self._slots[self.possible_agents.index(agent)]])

  def action_range(self, agent):
    return (self.spec_.action_lo, self.spec_.action_lo + self.spec_.n_actions - 1)

  def close(self):
    self._env.close()

  def _pack(self, o):
    """A step from Python is host-bound (every torch op is a launch, every view a microsecond).  The step's outputs live in
    persistent buffers -- the `terminated` flags and the decoded directions among them (sgw_out.done / obs_dir / act_dir) -- so the
    per-agent dicts of views are built ONCE; a step then only refreshes the agent windows where the step launch does not write them
    (one launch into a reused buffer).  The returned tensors are valid until the next step / reset, like the engine's outputs."""
    sp = self.spec_
    n = self.num_envs
    eng = self._env.engine
    if getattr(self, "_packed_for", None) is not o.get("reward"):       # first call, or the engine reallocated its outputs
      self._packed_for = o.get("reward")
      if self._fused:                                              # the step launch wrote the windows (ascii or value-mapped)
        views = eng.split_views(o["views" if self._ascii else "obs_views"])
      else:
        vb = int(eng._lib.sgw_view_bytes(eng._h))
        self._view_buf = torch.empty((n, vb), dtype=torch.uint8, device=self.device)
        views = eng.agent_views(out=self._view_buf)
      st = o["step_type"].reshape(n, -1)
      self._done = o["done"].reshape(n, -1).view(torch.bool)
      rew = o["reward"].reshape(n, sp.A, sp.K)
      cum = o["cumulative"].reshape(n, sp.A, sp.K)
      pos = o["agent_pos"].reshape(n, sp.A, 2)
      if self._turning:
        self._odir, self._adir = o["obs_dir"].reshape(n, sp.A), o["act_dir"].reshape(n, sp.A)
      metrics = o["metrics"][:, :sp.M]
      obs, rewards, terms, truncs, infos = {}, {}, {}, {}, {}
      for i, a in enumerate(self.possible_agents):
        q, k = self._slots[i], self._k[a]
        c = q if self._per_agent else 0
        obs[a] = views[q]
        rewards[a] = rew[:, q, :k]
        terms[a] = self._done[:, c]
        truncs[a] = self._never
        infos[a] = {"step_type": st[:, c], "cumulative_reward": cum[:, q, :k], "agent_position": pos[:, q], "discount": o["discount"],
                    "metrics": metrics, "board": o["board"],
                    "observation_direction": self._odir[:, q] if self._turning else self._up,
                    "action_direction": self._adir[:, q] if self._turning else self._up}
      self._cached = (obs, rewards, terms, truncs, infos)
    elif not self._fused:
      eng.agent_views(out=self._view_buf)
    obs, rewards, terms, truncs, infos = self._cached
    obs, infos = dict(obs), {a: dict(d) for a, d in infos.items()}
    if not self._ascii and not self._fused:
      obs = {a: self._vm[v.long()] for a, v in obs.items()}
    if self._layers:                                            # two more launches, tensors stay on the device
      cube = eng.observe_layers()
      agent_cubes = eng.agent_layer_views(layers=cube)
      for i, a in enumerate(self.possible_agents):
        infos[a]["info_observation_layers_order"] = self.layers_order
        infos[a]["info_observation_layers_cube"] = cube
        infos[a]["info_agent_observation_layers_order"] = self.layers_order
        infos[a]["info_agent_observation_layers_cube"] = agent_cubes[self._slots[i]]
    return obs, dict(rewards), dict(terms), dict(truncs), infos

  def reset(self, mask=None):
    o = self._env.engine.reset(mask)             # straight to the engine: the L4 TimeStep bookkeeping is a dozen torch launches
    self._env._last = o
    obs, _, _, _, infos = self._pack(o)
    return obs, infos

  def step(self, actions):
    """actions: {agent: int8 / int64 tensor [N] on the device (or a python int for every env)}; a missing agent does not play
    this round (EnvironmentMa.step plays exactly the agents in the dict).  Or ONE int8 tensor [N, A] already in the library's column order
    (`agent_slots`): no per-agent copy."""
    sp = self.spec_
    if torch.is_tensor(actions):                                 # already the library's layout: int8 [N, A] by column (no copy)
      acts = actions
    else:
      for i, a in enumerate(self.possible_agents):
        col = self._acts[:, self._slots[i]]
        v = actions.get(a)
        if v is None:                                            # not in the dict: the agent does not play this round (PM:173-246)
          col.fill_(-1)
        elif torch.is_tensor(v):
          col.copy_(v.reshape(-1))
        else:
          col.fill_(int(v))
      acts = self._acts
    o = self._env.engine.step(acts.reshape(-1) if sp.A == 1 else acts)
    self._env._last = o
    return self._pack(o)
