"""BatchedSafetyEnvironment: the L4 `SafetyEnvironment*` surface (reset / step / get_last_performance /
environment_data-like accessors) for N lockstep env instances on one MI355X.

Mirrors shared/safety_game.py:82-316 and shared/safety_game_mo.py:148-1107 for what is on the step
path; logging, Q-value dumps and the class-level episode/trial counters are host-side observability
and out of scope.
"""
import collections
import enum

import numpy as np
import torch

from . import _native as N
from .engine import BatchedEngine, ALL_OUTPUTS
from .specs import make_spec


class StepType(enum.IntEnum):      # shared/rl/environment.py:62-80
  FIRST = 0
  MID = 1
  LAST = 2

  def first(self): return self is StepType.FIRST
  def mid(self): return self is StepType.MID
  def last(self): return self is StepType.LAST


class TerminationReason(enum.IntEnum):   # shared/termination_reason_enum.py:25-39
  TERMINATED = 0
  MAX_STEPS = 1
  INTERRUPTED = 2
  QUIT = 3


class TimeStep(collections.namedtuple("TimeStep", ["step_type", "reward", "discount", "observation"])):
  """Batched analogue of shared/rl/environment.py:29-60: every field is an array over envs."""
  __slots__ = ()


class BatchedSafetyEnvironment(object):

  def __init__(self, env_name, num_envs=1, device="cuda:0", env_id_base=0, outputs=None, track_performance=True, **kwargs):
    self.env_name = env_name
    self._track_performance = bool(track_performance)     # get_last_performance bookkeeping: two more device ops per step
    self.spec = make_spec(env_name, **kwargs)
    self.num_envs = int(num_envs)
    if outputs is None:        # everything the family produces ('safety2_<agent>' exists in aintelope_savanna only)
      outputs = ALL_OUTPUTS + (("safety2",) if self.spec.name == "aintelope_savanna" else ())
    self.engine = BatchedEngine(self.spec, self.num_envs, device=device, env_id_base=env_id_base,
                                outputs=outputs)
    self.device = self.engine.device
    self._last = None
    self._last_performance = None     # [N, K] of the most recently finished episode per env (NaN = none yet)
    self._performance_sum = None      # [N, K] running sum over the env's finished episodes; _episodes int64 [N] their number
    self._episodes = None

  # ---- specs ----------------------------------------------------------------------------------
  def action_spec(self):
    """(minimum, maximum) inclusive, like BoundedArraySpec (pycolab_interface_mo.py:235-262)."""
    return (self.spec.action_lo, self.spec.action_lo + self.spec.n_actions - 1)

  @property
  def enabled_reward_dimension_keys(self):
    return list(self.spec.dim_names)

  # ---- stepping -------------------------------------------------------------------------------
  def _timestep(self, o):
    self._last = o
    st = o["step_type"].reshape(self.num_envs, -1)
    if getattr(self.spec, "per_agent", False):                # agents finish individually: the env is LAST once all are LAST/DEAD
      done_all, first_all = (st >= N.LAST).all(dim=1), (st == N.FIRST).all(dim=1)
      st = torch.where(done_all, torch.full_like(st[:, 0], N.LAST), torch.where(first_all, torch.full_like(st[:, 0], N.FIRST),
                                                                               torch.full_like(st[:, 0], N.MID)))
    else:
      st = st[:, 0]                                           # all agents of an env share the step type
    if self._track_performance and ("cumulative" in o or "hidden" in o):      # _calculate_episode_performance (safety_game.py:253-263)
      self._track(o)
    return TimeStep(step_type=st, reward=o.get("reward"), discount=o.get("discount"), observation=o)

  def _perf_source(self, o):
    use_hidden = self.spec.scalar and getattr(self.spec, "performance", "hidden") == "hidden"   # distributional_shift keeps the default: episode return
    return ("hidden", 1) if use_hidden and "hidden" in o else ("cumulative", self.spec.A * self.spec.K)

  def _track(self, o):
    """Device-side _episodic_performances bookkeeping (sgw_track_performance): last performance, running sum and episode count
    per env, updated where the step that just ran was LAST."""
    import ctypes as C
    name, cols = self._perf_source(o)
    if o.get(name) is None or "step_type" not in o:
      return
    if self._last_performance is None:
      self._last_performance = torch.full((self.num_envs, cols), float("nan"), dtype=torch.float64, device=self.device)
      self._performance_sum = torch.zeros((self.num_envs, cols), dtype=torch.float64, device=self.device)
      self._episodes = torch.zeros(self.num_envs, dtype=torch.int64, device=self.device)
    eng = self.engine
    N.check(eng._lib.sgw_track_performance(eng._h, eng._bufs[name].data_ptr(), cols, eng._bufs["step_type"].data_ptr(),
                                           self._last_performance.data_ptr(), self._performance_sum.data_ptr(),
                                           self._episodes.data_ptr(), None, eng._stream()), "sgw_track_performance")

  def reset(self, mask=None):
    return self._timestep(self.engine.reset(mask))

  def step(self, actions):
    if not torch.is_tensor(actions):
      actions = torch.as_tensor(np.asarray(actions).reshape(-1), dtype=torch.int8)
    return self._timestep(self.engine.step(actions))

  def get_last_performance(self, default=None):
    """Per env: performance of the last finished episode (safety_game.py:229-251); NaN rows = none yet."""
    return default if self._last_performance is None else self._last_performance

  def get_overall_performance(self, default=None):
    """Per env: the mean of the performances of its finished episodes, sum(_episodic_performances) / len(...) in the reference's
    left-to-right order (safety_game.py:194-208, 234-244; safety_game_mo.py:917-938); NaN rows = no episode finished yet.
    `episodes_finished()` gives the counts."""
    if self._last_performance is None:
      return default
    return self._performance_sum / self._episodes.to(torch.float64)[:, None]      # 0 / 0 = NaN where none finished

  def episodes_finished(self):
    return None if self._last_performance is None else self._episodes

  def set_episode_bits(self, bits, seed=0):
    self.engine.set_episode_bits(bits, seed)

  def close(self):
    self.engine.close()
