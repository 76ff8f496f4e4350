"""GridworldZooAecEnv: the reference PettingZoo-AEC wrapper's surface (helpers/gridworld_zoo_aec_env.py:67-985) over the
batched HIP engine.

    env = GridworldZooAecEnv("aintelope_savanna", amount_agents=2, seed=7)
    env.reset()
    for agent in env.agent_iter():
      obs, reward, terminated, truncated, info = env.last()
      env.step(None if terminated or truncated else policy(obs))

Kept from the reference: ONE agent acts per `step` -- the wrapper submits `{agent: action}` alone to the multi-agent
environment (zoo_aec.py:651-652), i.e. one Engine.play, no order shuffle; on the device that is a round whose other
agents carry the "not submitted" action -1.  `agent_selection` cycles over the agents that are not done
(zoo_aec.py:336-360); a done agent takes one "dead step" with action None, which removes it from `.agents` and
clears the others' step rewards (zoo_aec.py:627-648); `last()` returns the cumulative reward since the agent's own
previous step, reset to zero when it acts (zoo_aec.py:772-774); `rewards` / `terminations` / `truncations` / `infos`
are per-agent dicts.  Observations are the agent-centric ascii windows of the parallel wrapper.
Only the per-agent multi-agent families accept a subset of the agents (island_navigation_ex_ma, aintelope_savanna);
single-agent envs run with their one agent.  pettingzoo is not required (absent in this image).
"""
import numpy as np

from .gridworld_zoo_parallel_env import GridworldZooParallelEnv

try:
  from pettingzoo import AECEnv as _Base
except Exception:                       # pragma: no cover - pettingzoo absent in this image
  _Base = object


class GridworldZooAecEnv(_Base):
  metadata = {"render.modes": ["human", "ansi", "rgb_array"], "name": "ai_safety_gridworlds_amd", "is_parallelizable": True}

  def __init__(self, env_name, *args, **kwargs):
    self._par = GridworldZooParallelEnv(env_name, *args, **kwargs)
    sp = self._par.spec_
    if sp.A > 1 and not getattr(sp, "per_agent", False):
      raise NotImplementedError("%s: the batched engine plays whole rounds of this env; one-agent-at-a-time stepping "
                                "needs a family with per-agent rounds (island_navigation_ex_ma, aintelope_savanna)" % env_name)
    self.possible_agents = list(self._par.possible_agents)
    self.agent_name_mapping = dict(self._par.agent_name_mapping)
    self._agents = []
    self._next_agent = None
    self._next_agent_index = -1
    self._all_agents_done = True

  # ---- PettingZoo AEC surface ---------------------------------------------------------------------------------------
  @property
  def agents(self):
    return list(self._agents)

  @property
  def num_agents(self):
    return len(self._agents)

  @property
  def max_num_agents(self):
    return len(self.possible_agents)

  @property
  def agent_selection(self):
    return self._next_agent

  @property
  def rewards(self):
    return self._rewards

  @property
  def infos(self):
    return self._infos

  @property
  def state(self):
    return self._par.state

  def observation_space(self, agent):
    return self._par.observation_space(agent)

  def action_space(self, agent):
    return self._par.action_space(agent)

  def seed(self, seed=None):
    self._par.seed(seed)

  def close(self):
    self._par.close()

  def agent_iter(self, max_iter=2 ** 63):
    n = 0
    while n < max_iter and not self._all_agents_done:      # yields the current agent; `step` moves on (zoo_aec.py:327-378)
      n += 1
      yield self._next_agent

  def _move_to_next_agent(self):                           # zoo_aec.py:336-360
    for _ in range(len(self.possible_agents)):
      self._next_agent_index = (self._next_agent_index + 1) % len(self.possible_agents)
      agent = self.possible_agents[self._next_agent_index]
      if agent in self._agents:
        self._next_agent = agent
        return
    self._next_agent_index, self._next_agent, self._all_agents_done = -1, None, True

  def reset(self, seed=None, *args, **kwargs):
    states, infos = self._par.reset(seed=seed)
    self._agents = list(self.possible_agents)
    self._all_agents_done = False
    self._states = dict(states)
    self._infos = dict(infos)
    self._rewards = {a: 0.0 for a in self._agents}
    self._cumulative_rewards = {a: 0.0 for a in self._agents}
    self.terminations = {a: False for a in self._agents}
    self.truncations = {a: False for a in self._agents}
    self._next_agent_index = 0
    self._next_agent = self.possible_agents[0]

  def observe(self, agent):
    """The agent's window onto the CURRENT board (every step refreshes every agent's window)."""
    return self._states.get(agent)

  def last_for_agent(self, agent=None, observe=True):
    agent = self._next_agent if agent is None else agent
    state = self.observe(agent) if observe else None
    return state, self._cumulative_rewards[agent], self.terminations[agent], self.truncations[agent], self._infos[agent]

  def last(self, observe=True):
    return self.last_for_agent(self._next_agent, observe)

  def step(self, action, *args, **kwargs):
    agent = self._next_agent
    if agent is None:
      raise RuntimeError("every agent is done: call reset()")
    if self.terminations[agent] or self.truncations[agent]:               # the "dead step" (zoo_aec.py:627-648)
      act = action["step"] if isinstance(action, dict) else action
      if act is not None:
        raise ValueError("When an agent is dead, the only valid action is None")
      for d in (self.terminations, self.truncations, self._cumulative_rewards, self._infos, self._states):
        d.pop(agent, None)
      self._agents.remove(agent)
      self._rewards = {a: 0.0 for a in self._agents}
      self._move_to_next_agent()
      return
    states, rewards, terms, truncs, infos = self._par.step({agent: action})
    self._states.update(states)
    self._infos[agent] = infos[agent]
    self._cumulative_rewards[agent] = 0.0                                 # zoo_aec.py:772: the acting agent starts afresh
    for a, r in rewards.items():
      if a in self._cumulative_rewards:
        self._cumulative_rewards[a] = self._cumulative_rewards[a] + r
    self._rewards.update({a: r for a, r in rewards.items() if a in self._agents})
    for a in self._agents:
      if self.terminations[a] or self.truncations[a]:
        self._rewards[a] = 0.0
    self.terminations[agent] = bool(terms.get(agent, False))
    self.truncations[agent] = False
    self._move_to_next_agent()
