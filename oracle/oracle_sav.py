"""ctypes front-end of the CPU oracle for aintelope_savanna.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import oracle as _o

A, NU, MAXCELLS, MAXM, MAXVIEW, NLAYER = 2, 13, 192, 32, 441, 9
LAYER_CHRS = "WPDFdfGS1"

_INT_FIELDS = ("level", "max_iterations", "amount_agents", "randomize_agent_actions_order", "sustainability_challenge",
               "thirst_hunger_death", "penalise_oversatiation", "use_satiation_proportional_reward",
               "map_randomization_frequency", "action_direction_mode", "observation_direction_mode", "observation_radius",
               "use_food_availability_metric_instead_of_spawning_tiles", "use_drink_availability_metric_instead_of_spawning_tiles",
               "amount_food_patches", "amount_drink_holes", "amount_small_food_patches", "amount_small_drink_holes",
               "amount_gold_deposits", "amount_silver_deposits", "amount_water_tiles", "amount_predators", "map_width", "map_height",
               "remove_unused_tile_types_from_layers")
_F64_FIELDS = (
    "movement_score", "final_score", "drink_deficiency_score", "food_deficiency_score", "drink_score", "food_score",
    "small_drink_score", "small_food_score", "non_drink_score", "non_food_score",
    "gap_score_food", "gap_score_drink", "gap_score_gold", "gap_score_silver",
    "gold_visits_log_base", "gold_score", "silver_visits_log_base", "silver_score",
    "danger_tile_score", "predator_npc_score", "predator_movement_probability",
    "cooperation_score", "small_cooperation_score", "drink_oversatiation_score", "food_oversatiation_score",
    "drink_deficiency_initial", "drink_extraction_rate", "small_drink_extraction_rate", "drink_deficiency_rate",
    "drink_oversatiation_limit", "drink_oversatiation_threshold", "drink_deficiency_threshold",
    "food_deficiency_initial", "food_extraction_rate", "small_food_extraction_rate", "food_deficiency_rate",
    "food_oversatiation_limit", "food_oversatiation_threshold", "food_deficiency_threshold",
    "drink_regrowth_exponent", "drink_growth_limit", "food_regrowth_exponent", "food_growth_limit")


class Config(C.Structure):
  _fields_ = [(n, C.c_int32) for n in _INT_FIELDS] + [(n, C.c_double) for n in _F64_FIELDS]


class TimeStep(C.Structure):
  _fields_ = [
      ("step_type", C.c_int32 * A), ("reward_none", C.c_int32), ("K", C.c_int32),
      ("reward", (C.c_double * NU) * A), ("cumulative", (C.c_double * NU) * A), ("discount", C.c_double),
      ("term_reason", C.c_int32 * A), ("frame", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
      ("board", C.c_uint8 * MAXCELLS), ("M", C.c_int32), ("view_side", C.c_int32), ("metrics", C.c_double * MAXM),
      ("pos", (C.c_int32 * 2) * A), ("action_direction", C.c_int32 * A), ("observation_direction", C.c_int32 * A),
      ("safety", C.c_int32 * A), ("safety2", C.c_int32 * A),
      ("rng", C.c_uint64 * 4), ("rng_has_uint32", C.c_int32), ("rng_uinteger", C.c_uint32),
      ("view", (C.c_uint8 * MAXVIEW) * A), ("layers", (C.c_uint8 * MAXCELLS) * NLAYER)]


TS_DTYPE = np.dtype(TimeStep)
RESET = -128            # actions[..., 0] == RESET: explicit reset() at that tick


def _lib():
  L = _o.lib()
  if not getattr(L, "_sav_ready", False):
    L.or_sav_default_config.argtypes = [C.POINTER(Config)]
    L.or_sav_run_streams.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.or_sav_last_error.restype = C.c_char_p
    L._sav_ready = True
  return L


def make_config(**kw):
  """kwargs use the reference's constructor / flag names (aintelope_savanna.py:417-590); mo_reward-valued flags are
  given as the value of their own dimension, GAP_SCORE as a dict, observation_radius as one int for all four sides."""
  cfg = Config()
  _lib().or_sav_default_config(C.byref(cfg))
  names = {f[0] for f in Config._fields_}
  for k, v in kw.items():
    k = k.lower()
    if k == "gap_score":
      for dim, val in v.items():
        setattr(cfg, "gap_score_" + dim.lower(), float(val))
      continue
    if isinstance(v, dict):                       # a mo_reward flag given as {dimension: value}: one dimension per flag
      if len(v) != 1:
        raise ValueError("flag %r: the oracle keeps each event on its own dimension" % k)
      (v,) = v.values()
    if k == "observation_radius" and not isinstance(v, int):
      if len(set(v)) != 1:
        raise ValueError("the oracle covers square views")
      v = int(v[0])
    if k in ("map_width", "map_height") and v is None:
      v = 0
    if k not in names:
      raise KeyError("aintelope_savanna oracle config has no field %r" % k)
    setattr(cfg, k, v)
  return cfg


def run_streams(cfg, actions, rng_states, nthreads=1):
  """actions int8 [E, T, 2] (RESET in agent 0's slot = explicit reset), rng_states uint64 [E, 4] -> dict of arrays
  [E, T+2, ...]: slot 0 constructor reset, slot 1 first reset(), then one per tick."""
  actions = np.ascontiguousarray(actions, dtype=np.int8)
  rng_states = np.ascontiguousarray(rng_states, dtype=np.uint64)
  E, T, _ = actions.shape
  outs = np.zeros((E, T + 2), dtype=TS_DTYPE)
  rc = _lib().or_sav_run_streams(C.byref(cfg), E, T, actions.ctypes.data, rng_states.ctypes.data, outs.ctypes.data, int(nthreads))
  if rc:
    raise ValueError("aintelope_savanna oracle failed: %s" % _lib().or_sav_last_error().decode())
  d = {n: outs[n] for n in TS_DTYPE.names}
  H, W, K, M = int(outs["H"][0, 0]), int(outs["W"][0, 0]), int(outs["K"][0, 0]), int(outs["M"][0, 0])
  S = int(outs["view_side"][0, 0]); n = cfg.amount_agents
  d["board"] = d["board"][..., :H * W].reshape(E, T + 2, H, W)
  d["layers"] = d["layers"][..., :H * W].reshape(E, T + 2, NLAYER, H, W)
  d["reward"] = d["reward"][:, :, :n, :K]; d["cumulative"] = d["cumulative"][:, :, :n, :K]
  d["metrics"] = d["metrics"][..., :M]
  d["view"] = d["view"][:, :, :n, :S * S].reshape(E, T + 2, n, S, S)
  for f in ("step_type", "term_reason", "pos", "action_direction", "observation_direction", "safety", "safety2"):
    d[f] = d[f][:, :, :n]
  return d
