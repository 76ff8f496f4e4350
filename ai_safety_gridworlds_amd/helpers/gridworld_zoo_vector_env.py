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
the per-agent families that are already LAST/DEAD can be given -1 ("not in the dict").  One round = ONE kernel launch for the
step (shuffled sequential plays, fire spread, rewards, auto-reset) + one for the agent windows.
"""
import numpy as np
import torch

from .. import _native as N
from ..environments import BatchedSafetyEnvironment

OUTS = ("board", "reward", "cumulative", "step_type", "term_reason", "discount", "metrics", "agent_pos", "agent_flags")


class GridworldZooVectorEnv(object):
  metadata = {"name": "ai_safety_gridworlds_amd_vector"}

  def __init__(self, env_name, num_envs, ascii_observation_format=True, layers_in_observation=False, seed=None, device="cuda:0",
               env_id_base=0, **kwargs):
    self._env = BatchedSafetyEnvironment(env_name, num_envs=num_envs, device=device, env_id_base=env_id_base, outputs=OUTS, **kwargs)
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
    self._ascii = bool(ascii_observation_format)
    self._layers = bool(layers_in_observation)
    self.layers_order = list(sp.layer_chars)                   # get_layers_order(...) of the reference: sorted layer keys
    self._vm = torch.tensor([sp.native.value_map[i] for i in range(128)], dtype=torch.float32, device=self.device)
    self._acts = torch.zeros((self.num_envs, sp.A), dtype=torch.int8, device=self.device)
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
    sp = self.spec_
    views = self._env.engine.agent_views()                      # per column: uint8 [N, h, w]
    st = o["step_type"].reshape(self.num_envs, -1)
    rew = o["reward"].reshape(self.num_envs, sp.A, sp.K)
    cum = o["cumulative"].reshape(self.num_envs, sp.A, sp.K)
    pos = o["agent_pos"].reshape(self.num_envs, sp.A, 2)
    flags = o["agent_flags"].reshape(self.num_envs, sp.A)
    cube = agent_cubes = None
    if self._layers:                                            # two more launches, tensors stay on the device
      cube = self._env.engine.observe_layers()
      agent_cubes = self._env.engine.agent_layer_views(layers=cube)
    obs, rewards, terms, truncs, infos = {}, {}, {}, {}, {}
    for i, a in enumerate(self.possible_agents):
      q, k = self._slots[i], self._k[a]
      v = views[q]
      obs[a] = v if self._ascii else self._vm[v.long()]
      s_a = st[:, q] if self._per_agent else st[:, 0]
      rewards[a] = rew[:, q, :k]
      terms[a] = s_a >= N.LAST
      truncs[a] = torch.zeros_like(terms[a])
      infos[a] = {"step_type": s_a, "cumulative_reward": cum[:, q, :k], "agent_position": pos[:, q], "discount": o["discount"],
                  "metrics": o["metrics"][:, :sp.M], "board": o["board"],
                  "observation_direction": (flags[:, q] >> 3) & 3, "action_direction": (flags[:, q] >> 1) & 3}
      if self._layers:
        infos[a]["info_observation_layers_order"] = self.layers_order
        infos[a]["info_observation_layers_cube"] = cube
        infos[a]["info_agent_observation_layers_order"] = self.layers_order
        infos[a]["info_agent_observation_layers_cube"] = agent_cubes[q]
    return obs, rewards, terms, truncs, infos

  def reset(self, mask=None):
    ts = self._env.reset(mask)
    obs, _, _, _, infos = self._pack(ts.observation)
    return obs, infos

  def step(self, actions):
    """actions: {agent: int8 / int64 tensor [N] on the device (or a python int for every env)}; a missing agent plays NOOP
    (per-agent families: does not play this round)."""
    sp = self.spec_
    for i, a in enumerate(self.possible_agents):
      col = self._acts[:, self._slots[i]]
      v = actions.get(a)
      if v is None:
        col.fill_(-1 if self._per_agent else 0)
      elif torch.is_tensor(v):
        col.copy_(v.reshape(-1))
      else:
        col.fill_(int(v))
    ts = self._env.step(self._acts.reshape(-1) if sp.A == 1 else self._acts)
    return self._pack(ts.observation)
