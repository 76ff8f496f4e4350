"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does (tests/test_no_oracle_in_product.py
enforces it).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

ISLAND_EX, BOAT_RACE_EX, BOAT_RACE, SAFE_INT, ISLAND_NAV, DIST_SHIFT, ABSENT_SUP, SOKOBAN, CONVEYOR, TOMATO, FRIEND_FOE, WHISKY_GOLD, ROCKS_DIAMONDS = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12
FAMILY_IDS = {"island_navigation_ex": ISLAND_EX, "boat_race_ex": BOAT_RACE_EX,
              "boat_race": BOAT_RACE, "safe_interruptibility": SAFE_INT, "island_navigation": ISLAND_NAV,
              "distributional_shift": DIST_SHIFT, "absent_supervisor": ABSENT_SUP, "side_effects_sokoban": SOKOBAN, "conveyor_belt": CONVEYOR, "tomato_watering": TOMATO, "friend_foe": FRIEND_FOE, "whisky_gold": WHISKY_GOLD, "tomato_crmdp": TOMATO, "rocks_diamonds": ROCKS_DIAMONDS,
              "conveyor_belt_ex": CONVEYOR, "safe_interruptibility_ex": SAFE_INT}
BANDIT_TYPES = ["friend", "neutral", "adversary"]
CONVEYOR_VARIANTS = ["vase", "sushi", "sushi_goal", "sushi_goal2"]
MAXCELLS, MAXK, MAXM = 320, 16, 16

_I32 = C.c_int32
_F64 = C.c_double


class Config(C.Structure):
  _fields_ = (
      [(n, _I32) for n in ("family", "level", "max_iterations", "noops",
                           "sustainability_challenge", "thirst_hunger_death",
                           "penalise_oversatiation", "use_satiation_proportional_reward")]
      + [(n, _F64) for n in (
          "movement_reward", "final_reward", "drink_deficiency_reward", "food_deficiency_reward",
          "drink_reward", "food_reward", "non_drink_reward", "non_food_reward",
          "gap_reward_food", "gap_reward_drink", "gap_reward_gold", "gap_reward_silver",
          "gold_reward", "silver_reward", "danger_tile_reward", "thirst_hunger_death_reward",
          "drink_oversatiation_reward", "food_oversatiation_reward",
          "drink_deficiency_initial", "drink_extraction_rate", "drink_deficiency_rate",
          "drink_deficiency_limit", "drink_oversatiation_limit",
          "food_deficiency_initial", "food_extraction_rate", "food_deficiency_rate",
          "food_deficiency_limit", "food_oversatiation_limit",
          "drink_regrowth_exponent", "drink_growth_limit", "drink_availability_initial",
          "food_regrowth_exponent", "food_growth_limit", "food_availability_initial")]
      + [("iterations_penalty", _I32), ("repetition_penalty", _I32),
         ("interruption_probability", _F64), ("is_testing", _I32), ("level_choice", _I32), ("supervisor", _I32)]
      + [(n, _F64) for n in ("sk_movement_reward", "sk_coin_reward", "sk_goal_reward", "sk_wall_reward", "sk_corner_reward")]
      + [("variant", _I32), ("cb_goal_reward", _F64), ("bandit_type", _I32), ("extra_step", _I32),
         ("whisky_exploration", _F64), ("human_player", _I32), ("tomato_crmdp", _I32), ("mo_twin", _I32),
         ("general_rewards", _I32), ("reward_mask", C.c_uint32 * 15), ("reward_vec", (_F64 * 12) * 15)])


class TimeStep(C.Structure):
  _fields_ = [
      ("step_type", _I32), ("reward_none", _I32), ("K", _I32), ("discount_none", _I32),
      ("reward", _F64 * MAXK), ("cumulative", _F64 * MAXK), ("discount", _F64),
      ("term_reason", _I32), ("actual_action", _I32), ("frame", _I32), ("has_performance", _I32),
      ("hidden", _F64), ("last_performance", _F64 * MAXK),
      ("H", _I32), ("W", _I32), ("board", C.c_uint8 * MAXCELLS),
      ("M", _I32), ("metrics", _F64 * MAXM), ("safety", _I32), ("should_interrupt", _I32)]


class StreamOut(C.Structure):
  _fields_ = [(n, C.c_void_p) for n in (
      "step_type", "reward_none", "reward", "cumulative", "discount", "term_reason",
      "actual_action", "frame", "hidden", "last_performance", "board", "metrics", "safety",
      "should_interrupt")]


_OUT_DTYPES = dict(
    step_type=np.uint8, reward_none=np.uint8, reward=np.float64, cumulative=np.float64,
    discount=np.float64, term_reason=np.int8, actual_action=np.int8, frame=np.int32,
    hidden=np.float64, last_performance=np.float64, board=np.uint8, metrics=np.float64,
    safety=np.int32, should_interrupt=np.uint8)

_lib = None


def lib():
  global _lib
  if _lib is None:
    path = _build.build()
    L = C.CDLL(path)
    L.or_last_error.restype = C.c_char_p
    L.or_default_config.argtypes = [C.c_int, C.POINTER(Config)]
    L.or_describe.argtypes = [C.POINTER(Config)] + [C.POINTER(C.c_int)] * 4 + [
        C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.or_env_create.restype = C.c_void_p
    L.or_env_create.argtypes = [C.POINTER(Config)]
    L.or_env_destroy.argtypes = [C.c_void_p]
    L.or_env_set_interrupt_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.or_env_reset.argtypes = [C.c_void_p, C.POINTER(TimeStep)]
    L.or_env_step.argtypes = [C.c_void_p, C.c_int, C.POINTER(TimeStep)]
    L.or_run_streams.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_int, C.POINTER(StreamOut), C.c_int]
    L.or_run_streams_rand.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                      C.POINTER(StreamOut), C.c_int]
    _lib = L
  return _lib


ISLAND_DIMS = ["DANGER_TILE_REWARD", "DRINK_DEFICIENCY_REWARD", "DRINK_OVERSATIATION_REWARD", "DRINK_REWARD", "FINAL_REWARD",
               "FOOD_DEFICIENCY_REWARD", "FOOD_OVERSATIATION_REWARD", "FOOD_REWARD", "GOLD_REWARD", "MOVEMENT_REWARD",
               "SILVER_REWARD", "THIRST_HUNGER_DEATH_REWARD"]
# reward flag (lower case) -> (event id in sgw_oracle.h's order, the flag's own dimension)
ISLAND_EVENTS = {"movement_reward": (0, "MOVEMENT_REWARD"), "thirst_hunger_death_reward": (1, "THIRST_HUNGER_DEATH_REWARD"),
                 "final_reward": (2, "FINAL_REWARD"), "drink_reward": (3, "DRINK_REWARD"), "non_drink_reward": (4, "DRINK_REWARD"),
                 "food_reward": (5, "FOOD_REWARD"), "non_food_reward": (6, "FOOD_REWARD"), "gold_reward": (7, "GOLD_REWARD"),
                 "silver_reward": (8, "SILVER_REWARD"), "gap_reward": (9, None),
                 "drink_deficiency_reward": (10, "DRINK_DEFICIENCY_REWARD"), "drink_oversatiation_reward": (11, "DRINK_OVERSATIATION_REWARD"),
                 "food_deficiency_reward": (12, "FOOD_DEFICIENCY_REWARD"), "food_oversatiation_reward": (13, "FOOD_OVERSATIATION_REWARD"),
                 "danger_tile_reward": (14, "DANGER_TILE_REWARD")}


def make_config(family, **kw):
  """family: name or id; kw: reference constructor kwargs / flag names (lower case)."""
  fid = FAMILY_IDS.get(family, family)
  cfg = Config()
  lib().or_default_config(fid, C.byref(cfg))
  if family == "tomato_crmdp":
    cfg.tomato_crmdp = 1
  if family in ("conveyor_belt_ex", "safe_interruptibility_ex"):
    cfg.mo_twin = 1
  names = {f[0] for f in Config._fields_}
  general_flags, want_general = {}, False
  for k, v in kw.items():
    k = k.lower()
    if fid == CONVEYOR and k == "goal_reward":
      k = "cb_goal_reward"
      if isinstance(v, dict):
        v = list(v.values())[0]
    if fid == CONVEYOR and k == "variant" and isinstance(v, str):
      v = CONVEYOR_VARIANTS.index(v)
    if fid == ISLAND_EX and k in ISLAND_EVENTS and isinstance(v, dict):
      general_flags[k] = {d: float(x) for d, x in v.items()}
      own = ISLAND_EVENTS[k][1]
      if k != "gap_reward":
        if set(v) - {own}:
          want_general = True
        v = float(v.get(own, 0.0))
      elif set(v) - {"FOOD_REWARD", "DRINK_REWARD", "GOLD_REWARD", "SILVER_REWARD"}:
        want_general = True
    if k == "gap_reward" and isinstance(v, dict):     # island: {DRINK_REWARD: x, FOOD_REWARD: y, ...} -> gap_reward_<dim>
      for dim, val in v.items():
        setattr(cfg, "gap_reward_" + dim.lower().replace("_reward", ""), float(val))
      continue
    if isinstance(v, dict) and k in names:            # a mo_reward flag {its own dimension: value}
      if len(v) != 1:
        raise ValueError("flag %r: the oracle keeps each event on its own dimension" % k)
      (v,) = v.values()
    if fid == SOKOBAN and "sk_" + k in names:       # movement_reward / coin_reward / ... (side_effects_sokoban.py:318-325)
      k = "sk_" + k
    if k not in names:
      raise KeyError("oracle config has no field %r" % k)
    if k == "bandit_type":
      v = -1 if v is None else (BANDIT_TYPES.index(v) if isinstance(v, str) else int(v))
    if k in ("level_choice", "supervisor"):          # None = drawn per game build
      v = -1 if v is None else int(v)
    setattr(cfg, k, v)
  if want_general:                                    # one event on several dimensions: the per-event vectors
    cfg.general_rewards = 1
    for flag, (ev, own) in ISLAND_EVENTS.items():
      vec = general_flags.get(flag)
      if vec is None:                                 # not overridden: the scalar(s) already in cfg
        vec = ({d + "_REWARD": getattr(cfg, "gap_reward_" + d.lower()) for d in ("FOOD", "DRINK", "GOLD", "SILVER")}
               if flag == "gap_reward" else {own: getattr(cfg, flag)})
      for d, x in vec.items():
        u = ISLAND_DIMS.index(d)
        cfg.reward_mask[ev] |= 1 << u
        cfg.reward_vec[ev][u] = float(x)
  return cfg


def describe(cfg):
  H, W, K, M = C.c_int(), C.c_int(), C.c_int(), C.c_int()
  dims = C.create_string_buffer(1024)
  mets = C.create_string_buffer(1024)
  if lib().or_describe(C.byref(cfg), H, W, K, M, dims, 1024, mets, 1024):
    raise ValueError(lib().or_last_error().decode())
  return dict(H=H.value, W=W.value, K=K.value, M=M.value,
              dim_names=[s for s in dims.value.decode().split("|") if s],
              metric_names=[s for s in mets.value.decode().split("|") if s])


class Env(object):
  """One oracle env instance (reset()/step(a) -> TimeStep struct)."""

  def __init__(self, cfg):
    self._h = lib().or_env_create(C.byref(cfg))
    if not self._h:
      raise ValueError(lib().or_last_error().decode())
    self._bits = None

  def set_interrupt_bits(self, bits):
    self._bits = np.ascontiguousarray(bits, dtype=np.uint8)
    lib().or_env_set_interrupt_bits(self._h, self._bits.ctypes.data, len(self._bits))

  def reset(self):
    ts = TimeStep()
    if lib().or_env_reset(self._h, C.byref(ts)):
      raise ValueError(lib().or_last_error().decode())
    return ts

  def step(self, action):
    ts = TimeStep()
    if lib().or_env_step(self._h, int(action), C.byref(ts)):
      raise ValueError(lib().or_last_error().decode())
    return ts

  def __del__(self):
    if getattr(self, "_h", None):
      lib().or_env_destroy(self._h)
      self._h = None


def run_streams(cfg, actions, interrupt_bits=None, fields=None, nthreads=1, rand_stream=None):
  """actions int8 [E, T] -> dict of arrays shaped like the golden fixtures ([E, T+1, ...])."""
  actions = np.ascontiguousarray(actions, dtype=np.int8)
  E, T = actions.shape
  d = describe(cfg)
  S = T + 1
  shapes = dict(
      step_type=(E, S), reward_none=(E, S), reward=(E, S, d["K"]), cumulative=(E, S, d["K"]),
      discount=(E, S), term_reason=(E, S), actual_action=(E, S), frame=(E, S), hidden=(E, S),
      last_performance=(E, S, d["K"]), board=(E, S, d["H"], d["W"]), metrics=(E, S, d["M"]),
      safety=(E, S), should_interrupt=(E, S))
  if fields is None:
    fields = list(shapes)
  out = {f: np.zeros(shapes[f], _OUT_DTYPES[f]) for f in fields}
  so = StreamOut()
  for f in fields:
    setattr(so, f, out[f].ctypes.data)
  bits_ptr, n_bits = None, 0
  if interrupt_bits is not None:
    interrupt_bits = np.ascontiguousarray(interrupt_bits, dtype=np.uint8)
    assert interrupt_bits.shape[0] == E
    bits_ptr, n_bits = interrupt_bits.ctypes.data, interrupt_bits.shape[1]
  rs_ptr, n_rand = None, 0
  if rand_stream is not None:      # external numbers for envs that draw from the process-global numpy RNG during play
    rand_stream = np.ascontiguousarray(rand_stream, dtype=np.float64)
    assert rand_stream.shape[0] == E
    rs_ptr, n_rand = rand_stream.ctypes.data, rand_stream.shape[1]
  rc = lib().or_run_streams_rand(C.byref(cfg), E, T, actions.ctypes.data, bits_ptr, n_bits, rs_ptr, n_rand,
                                 C.byref(so), int(nthreads))
  if rc:
    raise ValueError(lib().or_last_error().decode())
  out.update(d)
  return out
