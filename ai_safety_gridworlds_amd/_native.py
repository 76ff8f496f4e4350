"""ctypes binding of libsgw.so (include/sgw.h).  No CPU fallback: if the library is missing or a
HIP call fails, the error is raised -- the product path never silently degrades."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsgw.so")

MAX_CELLS, MAX_K, MAX_M, MAX_AGENTS, N_PARAMS, ENV_ALIGN = 320, 16, 32, 4, 72, 64
ABI_VERSION = 8

ISLAND_NAVIGATION_EX, BOAT_RACE_EX, BOAT_RACE, SAFE_INTERRUPTIBILITY, FIREMAKER_EX_MA, ISLAND_NAVIGATION_EX_MA, TILE_EVENTS, SIDE_EFFECTS_SOKOBAN, CONVEYOR_BELT, TOMATO_WATERING, FRIEND_FOE, WHISKY_GOLD, ROCKS_DIAMONDS, AINTELOPE_SAVANNA = range(14)
FIRST, MID, LAST, DEAD = 0, 1, 2, 3
TERM_NONE = 255


class SgwError(RuntimeError):
  pass


class Spec(C.Structure):
  """Mirror of `struct sgw_spec` (include/sgw.h)."""
  _fields_ = [
      ("family", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("K", C.c_int32),
      ("M", C.c_int32), ("A", C.c_int32), ("max_iterations", C.c_int32),
      ("start_cell", C.c_int32 * MAX_AGENTS),
      ("action_lo", C.c_int32), ("n_actions", C.c_int32), ("flags", C.c_int32),
      ("view_radius", (C.c_int32 * 4) * MAX_AGENTS),
      ("view_outside", C.c_int32),
      ("reserved", C.c_int32 * 2),
      ("dim_slot", (C.c_int8 * MAX_K) * MAX_AGENTS),
      ("metric_slot", C.c_int8 * MAX_M),
      ("params", C.c_double * N_PARAMS),
      ("value_map", C.c_float * 128),
      ("static_board", C.c_uint8 * MAX_CELLS),
      ("art", C.c_uint8 * MAX_CELLS),
      ("aux", C.c_uint8 * MAX_CELLS),
  ]


OUT_FIELDS = ("board", "obs_board", "reward", "cumulative", "step_type", "term_reason",
              "actual_action", "discount", "hidden", "safety", "metrics", "frame", "agent_pos", "agent_flags", "safety2",
              "views", "obs_views", "done", "obs_dir", "act_dir")


class Out(C.Structure):
  """Mirror of `struct sgw_out`: device pointers (0 = skip that output)."""
  _fields_ = [(n, C.c_void_p) for n in OUT_FIELDS]


class Extras(C.Structure):
  """Mirror of `struct sgw_extras` (sgw_step_full)."""
  _fields_ = [("rgb", C.c_void_p), ("rgb_lut_dev", C.c_void_p), ("layers", C.c_void_p), ("layer_chars_dev", C.c_void_p),
              ("layer_static_dev", C.c_void_p), ("n_layers", C.c_int32), ("gap_index", C.c_int32), ("hidden_layer", C.c_int32),
              ("perf_from_hidden", C.c_int32), ("stats", C.c_void_p), ("k_agent", C.c_int32 * MAX_AGENTS),
              ("agent_layer_views", C.c_void_p), ("perf_last", C.c_void_p), ("perf_sum", C.c_void_p), ("perf_count", C.c_void_p),
              ("done", C.c_void_p), ("replay", C.c_int32), ("reserved_", C.c_int32)]


_lib = None


def lib():
  """Load libsgw.so (building it with hipcc if the sources are newer / it is absent)."""
  global _lib
  if _lib is not None:
    return _lib
  # One HIP runtime per process: torch ships its own libamdhip64; load torch first so libsgw.so binds
  # to the runtime torch's tensors/streams live in (two runtimes => "no ROCm-capable device").
  import torch  # noqa: F401
  from . import build as _build
  override = os.environ.get("SGW_LIBRARY")   # diagnostics: load this build of the library instead (tools/diag/)
  if override:
    if not os.path.exists(override):
      raise SgwError("SGW_LIBRARY=%s does not exist" % override)
    path = override
  else:
    try:
      path = _build.build()
    except Exception as ex:   # no hipcc: use a prebuilt .so if one travelled with the tree
      if not os.path.exists(LIB_PATH):
        raise SgwError("libsgw.so is missing and cannot be built (%s); the engine has no CPU fallback" % ex)
      path = LIB_PATH
  L = C.CDLL(path)
  L.sgw_last_error.restype = C.c_char_p
  L.sgw_build_info.restype = C.c_char_p
  L.sgw_create.argtypes = [C.POINTER(Spec), C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
  L.sgw_destroy.argtypes = [C.c_void_p]
  for f in ("sgw_n_envs", "sgw_n_pad", "sgw_state_bytes"):
    getattr(L, f).restype = C.c_int64
    getattr(L, f).argtypes = [C.c_void_p]
  L.sgw_state_words.argtypes = [C.c_void_p]
  L.sgw_set_episode_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64]
  L.sgw_set_rng_state.argtypes = [C.c_void_p, C.c_void_p]
  L.sgw_set_random_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64]
  L.sgw_set_family_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
  L.sgw_pow_selfcheck.restype = C.c_int64
  L.sgw_pow_selfcheck.argtypes = []
  L.sgw_pow_f64.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
  L.sgw_reset.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Out), C.c_void_p]
  L.sgw_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Out), C.c_void_p]
  L.sgw_step_n.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Out), C.c_int, C.c_void_p]
  L.sgw_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Out), C.c_int, C.c_void_p]
  L.sgw_rollout.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int64, C.c_int, C.POINTER(Out),
                            C.c_int, C.c_void_p]
  L.sgw_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
  L.sgw_group_destroy.argtypes = [C.c_void_p]
  L.sgw_group_step_n.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(Out), C.c_int, C.c_void_p]
  L.sgw_group_rollout.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int64, C.c_int, C.POINTER(Out), C.c_int, C.c_void_p]
  L.sgw_read_returns.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
  L.sgw_fill_actions.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]
  L.sgw_accumulate_returns.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
  L.sgw_observe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                            C.c_void_p, C.c_void_p]
  L.sgw_derived_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
  L.sgw_observe_layers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_void_p]
  L.sgw_state_layers.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
  L.sgw_view_bytes.argtypes = [C.c_void_p]
  L.sgw_agent_views.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint8, C.c_void_p, C.c_void_p]
  L.sgw_agent_layer_views.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint8, C.c_void_p, C.c_void_p]
  L.sgw_track_performance.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
  L.sgw_step_full.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Out), C.POINTER(Extras), C.c_void_p]
  L.sgw_sizeof_extras.restype = C.c_int
  L.sgw_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
  L.sgw_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
  if L.sgw_abi_version() != ABI_VERSION:
    raise SgwError("libsgw.so ABI %d != binding ABI %d" % (L.sgw_abi_version(), ABI_VERSION))
  if L.sgw_sizeof_spec() != C.sizeof(Spec) or L.sgw_sizeof_out() != C.sizeof(Out):
    raise SgwError("struct layout mismatch between libsgw.so and the ctypes mirror "
                   "(spec %d vs %d, out %d vs %d)" % (L.sgw_sizeof_spec(), C.sizeof(Spec),
                                                      L.sgw_sizeof_out(), C.sizeof(Out)))
  if L.sgw_sizeof_extras() != C.sizeof(Extras):
    raise SgwError("struct layout mismatch: sgw_extras %d vs %d" % (L.sgw_sizeof_extras(), C.sizeof(Extras)))
  _lib = L
  return L


EXPORTS = [
    "sgw_abi_version", "sgw_last_error", "sgw_build_info", "sgw_sizeof_spec", "sgw_sizeof_out", "sgw_sizeof_extras", "sgw_create", "sgw_track_performance", "sgw_step_full",
    "sgw_destroy", "sgw_n_envs", "sgw_n_pad", "sgw_state_bytes", "sgw_set_episode_bits",
    "sgw_set_rng_state", "sgw_set_random_stream", "sgw_set_family_table", "sgw_pow_f64", "sgw_pow_selfcheck", "sgw_reset", "sgw_step", "sgw_step_n", "sgw_rollout", "sgw_replay", "sgw_group_create", "sgw_group_destroy", "sgw_group_step_n", "sgw_group_rollout", "sgw_read_returns", "sgw_fill_actions",
    "sgw_accumulate_returns", "sgw_observe", "sgw_derived_stats", "sgw_observe_layers", "sgw_state_layers", "sgw_view_bytes", "sgw_agent_views", "sgw_agent_layer_views", "sgw_state_words", "sgw_get_state", "sgw_set_state"]


def check(rc, what=""):
  if rc != 0:
    raise SgwError("%s failed (%d): %s" % (what or "libsgw call", rc, lib().sgw_last_error().decode()))
