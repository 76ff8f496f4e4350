"""ctypes front-end of the CPU oracle for island_navigation_ex_ma.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import oracle as _o

A, NU, MAXCELLS, MAXM, VIEW = 2, 12, 128, 16, 25

_INT_FIELDS = ("level", "max_iterations", "randomize_agent_actions_order", "sustainability_challenge",
               "thirst_hunger_death", "penalise_oversatiation", "use_satiation_proportional_reward",
               "map_randomization_frequency", "action_direction_mode", "observation_direction_mode", "map_width", "map_height",
               "remove_unused_tile_types_from_layers")
_F64_FIELDS = (
    "movement_reward", "final_reward", "drink_deficiency_reward", "food_deficiency_reward", "drink_reward", "food_reward",
    "non_drink_reward", "non_food_reward", "gap_reward_food", "gap_reward_drink", "gap_reward_gold", "gap_reward_silver",
    "gold_reward", "silver_reward", "danger_tile_reward", "thirst_hunger_death_reward",
    "drink_oversatiation_reward", "food_oversatiation_reward",
    "drink_deficiency_initial", "drink_extraction_rate", "drink_deficiency_rate", "drink_deficiency_limit",
    "drink_oversatiation_limit", "drink_oversatiation_threshold", "drink_deficiency_threshold",
    "food_deficiency_initial", "food_extraction_rate", "food_deficiency_rate", "food_deficiency_limit",
    "food_oversatiation_limit", "food_oversatiation_threshold", "food_deficiency_threshold",
    "drink_regrowth_exponent", "drink_growth_limit", "drink_availability_initial",
    "food_regrowth_exponent", "food_growth_limit", "food_availability_initial")


class Config(C.Structure):
  _fields_ = [(n, C.c_int32) for n in _INT_FIELDS] + [(n, C.c_double) for n in _F64_FIELDS]


class TimeStep(C.Structure):
  _fields_ = [
      ("step_type", C.c_int32 * A), ("reward_none", C.c_int32), ("K", C.c_int32),
      ("reward", (C.c_double * NU) * A), ("cumulative", (C.c_double * NU) * A), ("discount", C.c_double),
      ("term_reason", C.c_int32 * A), ("frame", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
      ("board", C.c_uint8 * MAXCELLS), ("M", C.c_int32), ("metrics", C.c_double * MAXM), ("pos", (C.c_int32 * 2) * A),
      ("action_direction", C.c_int32 * A), ("observation_direction", C.c_int32 * A), ("safety", C.c_int32 * A),
      ("rng", C.c_uint64 * 4), ("rng_has_uint32", C.c_int32), ("rng_uinteger", C.c_uint32),
      ("view", (C.c_uint8 * VIEW) * A)]


TS_DTYPE = np.dtype(TimeStep)
RESET = -128            # actions[..., 0] == RESET: explicit reset() at that tick


def _lib():
  L = _o.lib()
  if not getattr(L, "_ima_ready", False):
    L.or_ima_default_config.argtypes = [C.POINTER(Config)]
    L.or_ima_run_streams.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.or_ima_last_error.restype = C.c_char_p
    L._ima_ready = True
  return L


def make_config(**kw):
  """kwargs use the reference's constructor / flag names (island_navigation_ex_ma.py:276-420); mo_reward-valued flags
  are given as the value of their own dimension, GAP_REWARD as a dict."""
  cfg = Config()
  _lib().or_ima_default_config(C.byref(cfg))
  names = {f[0] for f in Config._fields_}
  for k, v in kw.items():
    k = k.lower()
    if k == "gap_reward":
      for dim, val in v.items():
        setattr(cfg, "gap_reward_" + dim.lower().replace("_reward", ""), float(val))
      continue
    if k == "amount_agents":
      if int(v) != 2:
        raise ValueError("the oracle covers amount_agents == 2")
      continue
    if k in ("map_width", "map_height") and v is None:
      v = 0
    if k not in names:
      raise KeyError("island_navigation_ex_ma oracle config has no field %r" % k)
    setattr(cfg, k, v)
  return cfg


def run_streams(cfg, actions, rng_states, nthreads=1):
  """actions int8 [E, T, 2] (RESET in agent 0's slot = explicit reset), rng_states uint64 [E, 4] (generator right after
  seeding) -> dict of arrays [E, T+2, ...]: slot 0 constructor reset, slot 1 first reset(), then one per tick."""
  actions = np.ascontiguousarray(actions, dtype=np.int8)
  rng_states = np.ascontiguousarray(rng_states, dtype=np.uint64)
  E, T, _ = actions.shape
  outs = np.zeros((E, T + 2), dtype=TS_DTYPE)
  rc = _lib().or_ima_run_streams(C.byref(cfg), E, T, actions.ctypes.data, rng_states.ctypes.data, outs.ctypes.data, int(nthreads))
  if rc:
    raise ValueError("island_navigation_ex_ma oracle failed: %s" % _lib().or_ima_last_error().decode())
  d = {n: outs[n] for n in TS_DTYPE.names}
  H, W, K, M = int(outs["H"][0, 0]), int(outs["W"][0, 0]), int(outs["K"][0, 0]), int(outs["M"][0, 0])
  d["board"] = d["board"][..., :H * W].reshape(E, T + 2, H, W)
  d["reward"] = d["reward"][..., :K]; d["cumulative"] = d["cumulative"][..., :K]
  d["metrics"] = d["metrics"][..., :M]
  d["view"] = d["view"].reshape(E, T + 2, A, 5, 5)
  return d
