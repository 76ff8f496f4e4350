"""CPU tier: host logic (spec tables, Philox, sharding) against the reference-run fixtures."""
import numpy as np
import pytest

from ai_safety_gridworlds_amd import philox, parallel
from ai_safety_gridworlds_amd.specs import make_spec, environment_names
from tests import golden_util as G


def test_philox_known_answers():
  # Random123 kat_vectors, philox4x32 10 rounds
  got = philox.philox4x32_10(0, 0, 0, 0, 0, 0)
  assert [int(x) for x in got] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
  f = 0xffffffff
  got = philox.philox4x32_10(f, f, f, f, f, f)
  assert [int(x) for x in got] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
  got = philox.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
  assert [int(x) for x in got] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_action_stream_is_keyed_by_global_env_id():
  full = philox.actions(7, np.arange(100), np.arange(9), 0, 5)
  part = philox.actions(7, np.arange(40, 70), np.arange(9), 0, 5)
  assert np.array_equal(full[:, 40:70], part)
  assert full.min() >= 0 and full.max() <= 4 and len(np.unique(full)) == 5


@pytest.mark.parametrize("name", G.fixture_names(G.SCALAR_PREFIXES))
def test_spec_tables_match_reference_fixture(name):
  fx, meta = G.load(name)
  spec = make_spec(meta["family_name"], **meta["kwargs"])
  assert (spec.H, spec.W, spec.K) == (meta["H"], meta["W"], meta["K"])
  if meta["K"] > 1:
    assert spec.dim_names == meta["dim_names"]
  assert spec.metric_names == meta["metric_labels"]
  if not name.endswith("_oob"):        # the _oob fixtures drive the env with action VALUES beyond its action_spec on purpose
    assert (spec.action_lo, spec.n_actions) == (meta["action_lo"], meta["n_actions"])
  board = fx["board"]
  # value_mapping LUT (observation['board']) and RGB LUT reproduce the reference's distiller output
  vm = np.array([spec.native.value_map[i] for i in range(128)], np.float32)
  assert np.array_equal(vm[board], fx["obs_board"])
  lut = spec.rgb_lut()
  n = fx["rgb"].shape[0]
  assert np.array_equal(np.moveaxis(lut[board[:n]], -1, 2), fx["rgb"])
  # reset board: static board + agent at its start cell
  table = spec.native.static_board
  if hasattr(spec, "art_variants") and "should_interrupt" in fx.files and fx["should_interrupt"][0, 0]:
    table = spec.native.aux                      # tile-event envs: the board variant of the first game build
  sb = np.array(list(table[:spec.H * spec.W]), np.uint8).reshape(spec.H, spec.W).copy()
  r, c = divmod(spec.native.start_cell[0], spec.W)
  sb[r, c] = ord('A')
  if spec.name in ("side_effects_sokoban", "whisky_gold", "rocks_diamonds"):   # boxes / coins / the whisky drape are dynamic entities: the reset board is the level art itself
    sb = np.array([[ord(ch) for ch in row] for row in spec.art], np.uint8)
  if spec.name == "friend_foe":                   # the floor drape depends on the bandit drawn for the episode
    sb = board[0, 0].copy()
    assert set(np.unique(sb)) <= set(map(ord, "#*AFNB"))
  if spec.name in ("tomato_watering", "tomato_crmdp"):              # tomatoes are dynamic; some of the initially watered ones dried during its_showtime
    sb = board[0, 0].copy()
    assert set(np.unique(sb[(sb != np.array([[ord(ch) for ch in row] for row in spec.art], np.uint8))])) <= {ord('t')}
  if spec.name in ("conveyor_belt", "conveyor_belt_ex"):   # the belt drape is stretched over its row at construction; the object rides on it
    orow, ocol = divmod("".join(spec.art).index('O'), spec.W)
    sb[orow, ocol] = ord('O')
  assert np.array_equal(sb, board[0, 0])
  if "layers" in fx.files:    # occluded layers == (board == char) except the unoccluded extras
    assert sorted(meta["layer_chars"]) == sorted(spec.layer_chars)


def test_island_safety_table_matches_fixture():
  fx, meta = G.load("island_L9")
  spec = make_spec("island_navigation_ex", level=9)
  board, safety, st = fx["board"], fx["safety"], fx["step_type"]
  aux = np.array(list(spec.native.aux[:48]), np.int32)
  pos = (board.reshape(board.shape[0], board.shape[1], -1) == ord('A')).argmax(-1)
  mid = st != 0
  assert np.array_equal(aux[pos][mid], safety[mid])
  assert (safety[~mid] == 3).all()


def test_spec_errors_mirror_the_reference():
  with pytest.raises(NotImplementedError):           # factory.py:201-202
    make_spec("no_such_environment")
  with pytest.raises(ValueError, match="DRINK_DEFICIENCY_REWARD is not enabled"):   # mo_reward.py:196-198
    make_spec("island_navigation_ex", level=0)
  with pytest.raises(TypeError):
    make_spec("island_navigation_ex", nonsense=1)
  with pytest.raises(ValueError, match="unknown reward dimensions"):
    make_spec("island_navigation_ex", MOVEMENT_REWARD={"SOME_OTHER_DIM": -1})
  g = make_spec("island_navigation_ex", DRINK_REWARD={"DRINK_REWARD": 2.0, "FOOD_REWARD": -1.0})   # one event, two dimensions
  with pytest.raises(NotImplementedError, match="device pow"):     # csrc/sgw_pow.hpp: glibc's main path only
    make_spec("island_navigation_ex", DRINK_REGROWTH_EXPONENT=0.0)
  assert g.native.flags & 16 and g.family_table.shape == (195,) and g.family_table[3 * 12 + 7] == -1.0
  s = make_spec("island_navigation_ex", movement_reward="{'MOVEMENT_REWARD': -2.5}", GOLD_REWARD={"GOLD_REWARD": 0})
  assert "GOLD_REWARD" not in s.dim_names and s.K == 9          # zero units drop out (mo_reward.py:131-135)
  assert s.native.params[0] == -2.5
  assert set(environment_names()) >= {"island_navigation_ex", "boat_race_ex", "boat_race", "safe_interruptibility"}


def test_savanna_spec_mirrors_the_reference_constructor():
  from tests import golden_util as G
  for name in G.fixture_names(["sav_"]):
    fx, meta = G.load(name)
    sp = make_spec("aintelope_savanna", **meta["kwargs"])
    assert sp.dim_names == meta["dim_names"] and sp.metric_names == meta["metric_labels"], name
    assert (sp.H, sp.W) == fx["board"].shape[2:]
  with pytest.raises(NotImplementedError, match="NameError"):       # safety_game_moma.py:1636
    make_spec("aintelope_savanna", thirst_hunger_death=True)
  with pytest.raises(RuntimeError, match="ObservationToArray"):     # fixed level-0 map holds '1' but one agent has no value for it
    make_spec("aintelope_savanna", map_randomization_frequency=0)
  with pytest.raises(ValueError, match="is not enabled"):           # drink / gold / ... tiles stay on a fixed map while their amount is 0
    make_spec("aintelope_savanna", amount_agents=2, map_randomization_frequency=0)
  t = make_spec("aintelope_savanna", amount_gold_deposits=1, max_iterations=10).family_table
  import math
  assert len(t) == 24 and t[0] == 40 * (math.log(2, 1.5) - math.log(1, 1.5)) and t[12 + 3] == 30 * (math.log(5, 1.5) - math.log(4, 1.5))


def test_shard_ranges_partition_the_env_ids():
  for n, w in [(65536, 8), (262144, 8), (10, 3), (7, 8)]:
    spans = [parallel.shard_range(n, r, w) for r in range(w)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_experiment_presets_and_factory_aliases():
  """helpers/factory.py:148-170 name forms; experiments/** presets = base env + recorded flag overrides
  (tests/golden/make_experiment_presets.py).  Four island presets put one event's reward on several dimensions (the
  island kernel's per-event reward vectors)."""
  from ai_safety_gridworlds_amd.specs import EXPERIMENT_PRESETS
  assert len(EXPERIMENT_PRESETS) == 24
  for name, (base, package, flags) in EXPERIMENT_PRESETS.items():
    sp = make_spec(name)
    assert sp.name == base
    assert make_spec(package + "." + name).native.params[:] == sp.native.params[:]
    assert make_spec("ai_safety_gridworlds." + package + "." + name, max_iterations=7).max_iterations == 7   # kwargs override the preset
  assert make_spec("food_drink_rolf").native.flags & 16          # per-event reward vectors (one event on several dimensions)
  assert make_spec("environments.boat_race_ex").name == "boat_race_ex"
  assert make_spec("aintelope.aintelope_savanna").name == "aintelope_savanna"
  assert make_spec("ai_safety_gridworlds.environments.aintelope.aintelope_savanna").name == "aintelope_savanna"
  sd = make_spec("savanna_demo")
  assert sd.n_agents == 2 and sd.view_shapes[0] == (9, 9) and sd.max_iterations == 100



def test_firemaker_flag_surface():
  """Direction modes as island_navigation_ex_ma's (firemaker_ex_ma.py:224-226, 808-811); map_width / map_height behave as in the
  reference, whose builder refuses to resize a map that is never randomised (firemaker_ex_ma.py:373, safety_game_mo_base.py:984-991:
  run against the reference, 19 x 19 and 15 x 17 raise AssertionError, 17 x 17 constructs)."""
  import pytest
  assert make_spec("firemaker_ex_ma", map_width=17, map_height=17).H == 17
  for kw in (dict(map_width=19, map_height=19), dict(map_width=15)):
    with pytest.raises(AssertionError):
      make_spec("firemaker_ex_ma", **kw)
  sp = make_spec("firemaker_ex_ma", action_direction_mode=2, observation_direction_mode=2)
  assert sp.n_actions == 9 and sp.rotating_views and sp.native.flags & 32 and sp.native.flags & 64
  sp = make_spec("firemaker_ex_ma", action_direction_mode=1, observation_direction_mode=1, noops=False)
  assert (sp.action_lo, sp.n_actions) == (1, 4) and sp.native.flags & 8 and sp.native.flags & 16
  assert not make_spec("firemaker_ex_ma").rotating_views
  wide = make_spec("firemaker_ex_ma", FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE=4.5)              # radius 4: the WIDE kernels (flag bit 7)
  assert wide.native.flags & 128 and wide.native.params[30] == 4.0 and wide.native.params[32] == 0.0 and wide.native.params[33] > 0
  assert not make_spec("firemaker_ex_ma", FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE=2.5).native.flags & 128
  with pytest.raises(NotImplementedError):
    make_spec("firemaker_ex_ma", FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE=5.5)
  with pytest.raises(NotImplementedError):
    make_spec("firemaker_ex_ma", action_direction_mode=0, observation_direction_mode=2)       # raises in the reference too (MA:670)
