"""ctypes front-end of the multi-agent CPU oracle (firemaker_ex_ma).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import oracle as _o

A, K, CELLS, NMETRIC = 3, 3, 289, 16


class Config(C.Structure):
  _fields_ = ([(n, C.c_int32) for n in ("amount_agents", "max_iterations", "randomize_agent_actions_order",
                                         "stop_button_press_effect_duration")]
              + [(n, C.c_double) for n in (
                  "fire_continuation_probability", "fire_spread_probability_at_distance_one",
                  "fire_spread_exclusive_max_distance", "agent_movement_reward", "agent_workshop_work_reward",
                  "agent_workshop_energy_reward", "supervisor_movement_reward", "supervisor_external_fire_reward",
                  "supervisor_trespassing_reward", "supervisor_stop_button_reward", "supervisor_workshop_reward")]
              + [(n, C.c_int32) for n in ("action_direction_mode", "observation_direction_mode")])


class TimeStep(C.Structure):
  _fields_ = [
      ("step_type", C.c_int32 * A), ("reward_none", C.c_int32),
      ("reward", (C.c_double * K) * A), ("cumulative", (C.c_double * K) * A), ("discount", C.c_double),
      ("term_reason", C.c_int32 * A), ("frame", C.c_int32), ("board", C.c_uint8 * CELLS),
      ("metrics", C.c_double * NMETRIC), ("pos", (C.c_int32 * 2) * A), ("rng", C.c_uint64 * 4),
      ("rng_has_uint32", C.c_int32), ("rng_uinteger", C.c_uint32),
      ("view_worker", (C.c_uint8 * 25) * 2), ("view_supervisor", C.c_uint8 * (33 * 33)),
      ("action_direction", C.c_int32 * A), ("observation_direction", C.c_int32 * A)]


TS_DTYPE = np.dtype(TimeStep)


def _lib():
  L = _o.lib()
  if not getattr(L, "_ma_ready", False):
    L.or_ma_default_config.argtypes = [C.POINTER(Config)]
    L.or_ma_run_streams.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.or_ma_rng_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L._ma_ready = True
  return L


def make_config(**kw):
  cfg = Config()
  _lib().or_ma_default_config(C.byref(cfg))
  names = {f[0] for f in Config._fields_}
  for k, v in kw.items():
    k = k.lower()
    if k not in names:
      raise KeyError("firemaker oracle config has no field %r" % k)
    setattr(cfg, k, v)
  return cfg


def rng_state_words(seed):
  """numpy PCG64(SeedSequence(seed)) -> uint64[4] (state_hi, state_lo, inc_hi, inc_lo)."""
  st = np.random.PCG64(np.random.SeedSequence(int(seed))).state["state"]
  m = (1 << 64) - 1
  return np.array([st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m], dtype=np.uint64)


def run_streams(cfg, actions, rng_states, nthreads=1, keep=True):
  """actions int8 [E, T, 3], rng_states uint64 [E, 4] -> dict of arrays [E, T+1, ...]."""
  actions = np.ascontiguousarray(actions, dtype=np.int8)
  rng_states = np.ascontiguousarray(rng_states, dtype=np.uint64)
  E, T, _ = actions.shape
  outs = np.zeros((E, T + 1), dtype=TS_DTYPE) if keep else None
  rc = _lib().or_ma_run_streams(C.byref(cfg), E, T, actions.ctypes.data, rng_states.ctypes.data,
                                outs.ctypes.data if keep else None, int(nthreads))
  if rc:
    raise ValueError("firemaker oracle failed")
  if not keep:
    return None
  d = {n: outs[n] for n in TS_DTYPE.names}
  d["board"] = d["board"].reshape(E, T + 1, 17, 17)
  d["view_worker"] = d["view_worker"].reshape(E, T + 1, 2, 5, 5)
  d["view_supervisor"] = d["view_supervisor"].reshape(E, T + 1, 33, 33)
  return d


def rng_probe(state_words, n):
  st = np.ascontiguousarray(state_words, dtype=np.uint64)
  r = np.zeros(n, np.float64); u = np.zeros(n, np.uint32); p = np.zeros((n, 3), np.int32)
  _lib().or_ma_rng_probe(st.ctypes.data, n, r.ctypes.data, u.ctypes.data, p.ctypes.data)
  return r, u, p
