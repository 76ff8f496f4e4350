"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP path, called through the C ABI,
must reproduce the reference bit for bit -- against the committed reference-run fixtures and,
on fresh seeds / larger batches, against the CPU oracle."""
import numpy as np
import pytest
import torch

from ai_safety_gridworlds_amd import philox
from ai_safety_gridworlds_amd.engine import BatchedEngine, ALL_OUTPUTS
from ai_safety_gridworlds_amd.specs import make_spec
from tests import golden_util as G

pytestmark = pytest.mark.gpu

CMP = ["step_type", "reward", "cumulative", "discount", "term_reason", "actual_action", "frame", "hidden",
       "board"]


def run_engine(spec, actions, bits=None, outputs=ALL_OUTPUTS, rand_stream=None):
  """actions int8 [E, T] -> dict of numpy arrays [E, T+1, ...] like the fixtures."""
  E, T = actions.shape
  eng = BatchedEngine(spec, E, device="cuda:0", outputs=outputs)
  if bits is not None:
    eng.set_episode_bits(bits)
  if rand_stream is not None:
    eng.set_random_stream(rand_stream)
  acts = torch.from_numpy(np.ascontiguousarray(actions.T)).to("cuda:0")     # [T, E]
  rec = {k: [] for k in outputs}
  o = eng.reset()
  for k in outputs:
    rec[k].append(o[k].clone())
  for t in range(T):
    o = eng.step(acts[t])
    for k in outputs:
      rec[k].append(o[k].clone())
  torch.cuda.synchronize()
  out = {k: torch.stack(v, dim=1).cpu().numpy() for k, v in rec.items()}
  eng.close()
  return out


def compare(name, got, want, K, fields=CMP):
  for f in fields:
    w = want[f]
    g = got[f]
    if f == "term_reason":
      g = g.astype(np.int16); g[g == 255] = -1
    if f in ("reward", "cumulative"):
      g = g.reshape(w.shape)
    if f == "step_type":
      g = g.reshape(w.shape)
    if f == "actual_action":
      g = g.reshape(w.shape)
    G.assert_same(name + "." + f, g, w)


@pytest.mark.parametrize("name", G.fixture_names(G.SCALAR_PREFIXES))
def test_hip_matches_reference_fixture(name):
  fx, meta = G.load(name)
  spec = make_spec(meta["family_name"], **meta["kwargs"])
  assert spec.dim_names == meta["dim_names"] or meta["K"] == 1
  assert (spec.H, spec.W, spec.K) == (meta["H"], meta["W"], meta["K"])
  bits = G.interrupt_bits(fx) if "should_interrupt" in fx.files else None
  got = run_engine(spec, fx["actions"], bits=bits, rand_stream=fx["rand_stream"] if "rand_stream" in fx.files and fx["rand_stream"].shape[1] else None)
  compare(name, got, fx, spec.K)
  G.assert_same(name + ".obs_board", got["obs_board"], fx["obs_board"])
  if "metrics" in fx.files:
    assert spec.metric_names == meta["metric_labels"]
    G.assert_same(name + ".metrics", got["metrics"][..., :spec.M], fx["metrics"])
  if "safety" in fx.files:
    G.assert_same(name + ".safety", got["safety"], fx["safety"])
  if "should_interrupt" in fx.files and "safety" not in fx.files:
    G.assert_same(name + ".should_interrupt", got["safety"], fx["should_interrupt"])
  if meta["family_name"] == "friend_foe":          # which box holds the reward (the level drawn for the episode)
    G.assert_same(name + ".level", got["agent_flags"].reshape(fx["should_interrupt"].shape), fx["should_interrupt"])


ORACLE_CASES = [
    ("island_navigation_ex", dict(level=9), 4096, 150, 0, 5),
    ("island_navigation_ex", dict(level=7, thirst_hunger_death=True, max_iterations=60), 1000, 150, 0, 5),
    ("island_navigation_ex", dict(level=2, use_satiation_proportional_reward=True), 777, 120, 0, 5),
    ("boat_race_ex", dict(level=3), 4096, 150, 0, 5),
    ("boat_race_ex", dict(level=0, repetition_penalty=False), 130, 150, 0, 5),
    ("boat_race", dict(level=0), 1000, 230, 1, 4),
    ("safe_interruptibility", dict(level=1), 4096, 150, 1, 4),
    ("safe_interruptibility", dict(level=2), 129, 150, 1, 4),
    ("island_navigation", dict(), 2000, 150, 0, 5),
    ("distributional_shift", dict(is_testing=True), 1500, 150, 1, 4),
    ("absent_supervisor", dict(), 1500, 150, 1, 4),
    ("side_effects_sokoban", dict(level=1, noops=True), 3000, 220, 0, 5),
    ("side_effects_sokoban", dict(level=3), 1000, 220, 1, 4),
    ("conveyor_belt", dict(variant="vase", noops=True), 2000, 220, 0, 5),
    ("conveyor_belt", dict(variant="sushi_goal2", goal_reward=7), 1000, 220, 1, 4),
    ("tomato_watering", dict(), 1500, 230, 1, 4),
    ("tomato_crmdp", dict(), 700, 230, 1, 4),
    ("friend_foe", dict(), 2000, 230, 1, 4),
    ("friend_foe", dict(bandit_type="adversary", extra_step=True), 500, 230, 1, 4),
    ("whisky_gold", dict(human_player=True, whisky_exploration=0.5), 2000, 230, 1, 4),
    ("rocks_diamonds", dict(level=0), 3000, 230, 1, 4),
    ("rocks_diamonds", dict(level=1), 500, 230, 1, 4),
    ("conveyor_belt_ex", dict(variant="sushi_goal", noops=True), 1500, 220, 0, 5),
    ("safe_interruptibility_ex", dict(level=2), 1500, 150, 1, 4),
]


@pytest.mark.parametrize("env_name,kw,E,T,lo,n", ORACLE_CASES)
def test_hip_matches_oracle_fresh_seed(env_name, kw, E, T, lo, n):
  """Larger batches (ragged: not multiples of 64) on a seed no fixture uses."""
  from oracle import oracle as O
  seed = 0xBEEF
  env_ids = np.arange(E)
  actions = philox.actions(seed, env_ids, np.arange(T), lo, n).T.copy()     # [E, T]
  bits = None
  if env_name in ("safe_interruptibility", "safe_interruptibility_ex", "distributional_shift", "absent_supervisor"):
    bits = (philox.actions(seed ^ 7, env_ids, np.arange(32), 0, 2).T.copy()).astype(np.uint8)
  rand = None
  if env_name in ("tomato_watering", "tomato_crmdp", "friend_foe", "whisky_gold"):
    rand = np.random.default_rng(11).random((E, 4096))
  cfg = O.make_config(env_name, **kw)
  want = O.run_streams(cfg, actions, interrupt_bits=bits, nthreads=8, rand_stream=rand)
  spec = make_spec(env_name, **kw)
  got = run_engine(spec, actions, bits=bits, rand_stream=rand)
  compare(env_name, got, want, spec.K)
  if spec.M:
    G.assert_same("metrics", got["metrics"][..., :spec.M], want["metrics"])
  if env_name == "island_navigation_ex":
    G.assert_same("safety", got["safety"], want["safety"])


def test_device_philox_matches_host():
  spec = make_spec("island_navigation_ex")
  eng = BatchedEngine(spec, 1000, device="cuda:0", env_id_base=12345)
  got = eng.fill_actions(17, seed=0x5AFE, step0=3).cpu().numpy()
  want = philox.actions(0x5AFE, 12345 + np.arange(1000), 3 + np.arange(17), 0, 5)
  assert np.array_equal(got, want)


def _fresh(spec, n, outs):
  e = BatchedEngine(spec, n, outputs=outs)
  e.set_episode_bits(None, seed=5)
  e.set_random_stream(None, seed=6)
  if getattr(spec, "needs_rng", False) or spec.name == "firemaker_ex_ma":
    e.set_rng_seeds(np.arange(n) + 3)
  e.reset()
  return e


@pytest.mark.parametrize("env_name,kw", [
    ("island_navigation_ex", {}), ("boat_race_ex", dict(level=3)), ("safe_interruptibility", {}), ("boat_race", {}),
    ("island_navigation", {}), ("distributional_shift", dict(is_testing=True)), ("absent_supervisor", {}),
    ("side_effects_sokoban", dict(level=1)), ("conveyor_belt", dict(variant="sushi_goal")), ("rocks_diamonds", {}),
    ("tomato_watering", {}), ("tomato_crmdp", {}), ("friend_foe", {}), ("whisky_gold", dict(human_player=True)),
    ("conveyor_belt_ex", dict(variant="vase")), ("safe_interruptibility_ex", {}),
    ("island_navigation_ex_ma", dict(map_randomization_frequency=3, max_iterations=30)),
    ("aintelope_savanna", dict(amount_agents=2, amount_predators=2, amount_water_tiles=3, amount_gold_deposits=2,
                               amount_silver_deposits=2, amount_small_food_patches=2, amount_drink_holes=2,
                               amount_small_drink_holes=1, sustainability_challenge=True, penalise_oversatiation=True,
                               max_iterations=40)),
])
def test_fused_rollout_equals_step_loop(env_name, kw):
  """sgw_rollout (state in registers, in-kernel Philox) == T x sgw_step fed the same stream."""
  spec = make_spec(env_name, **kw)
  n, T, seed = 3000, 64, 99
  outs = ("board", "reward", "cumulative", "step_type", "term_reason", "hidden", "frame")
  a = _fresh(spec, n, outs)
  b = _fresh(spec, n, outs)
  acts = a.fill_actions(T, seed)
  per_step = {k: [] for k in outs}
  for t in range(T):
    o = a.step(acts[t])
    for k in outs:
      per_step[k].append(o[k].clone())
  ro = b.rollout(T, seed, write_every=True)
  for k in outs:
    assert torch.equal(torch.stack(per_step[k]), ro[k]), k
  assert torch.equal(a.get_state()[:, :n], b.get_state()[:, :n])
  # accumulators: sum of returns of finished episodes + count, vs the step loop's outputs
  c = _fresh(spec, n, outs)
  c.rollout(T, seed, accumulate=True)
  acc = c.read_returns(clear=True)
  assert c.read_returns().abs().sum().item() == 0
  AK = spec.A * spec.K
  st = torch.stack(per_step["step_type"]).reshape(T, n, spec.A)
  cum = torch.stack(per_step["cumulative"]).reshape(T, n, AK)
  prev = torch.cat([torch.zeros_like(st[:1]), st[:-1]])
  last = (st >= 2).all(dim=2) & ~(prev >= 2).all(dim=2)          # the step at which the episode ended for every agent
  assert acc[AK].item() == last.sum().item()
  assert torch.equal(acc[:AK], (cum * last[..., None]).sum(dim=(0, 1)))
  d = _fresh(spec, n, outs)
  every = d.step_n(acts, write_every=True, accumulate=True)       # T launches writing [T, N, ...] == the rollout buffer
  for k in outs:
    assert torch.equal(every[k], ro[k]), "step_n " + k
  assert torch.equal(d.read_returns(), acc)


def test_masked_reset_only_touches_masked_envs():
  spec = make_spec("island_navigation_ex")
  n = 300
  eng = BatchedEngine(spec, n, outputs=("board", "reward", "step_type", "frame"))
  eng.reset()
  acts = eng.fill_actions(5, 1)
  for t in range(5):
    o = eng.step(acts[t])
  before = {k: v.clone() for k, v in o.items()}
  mask = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
  mask[::3] = 1
  o = eng.reset(mask)
  m = mask.bool()
  assert (o["step_type"][m] == 0).all() and (o["frame"][m] == 0).all()
  for k in before:
    assert torch.equal(o[k][~m], before[k][~m]), k


def test_observe_rgb_and_layers_match_fixture():
  fx, meta = G.load("island_L9")
  spec = make_spec("island_navigation_ex", level=9)
  E = fx["rgb"].shape[0]
  eng = BatchedEngine(spec, E, outputs=("board",))
  board = torch.from_numpy(fx["board"][:E, 5].copy()).to("cuda:0")
  out = eng.observe(board, rgb=True, layer_chars=list(meta["layer_chars"]))
  G.assert_same("rgb", out["RGB"].cpu().numpy(), fx["rgb"][:, 5])


def test_create_rejects_bad_arguments():
  from ai_safety_gridworlds_amd import _native as N
  spec = make_spec("island_navigation_ex")
  with pytest.raises(N.SgwError):
    BatchedEngine(spec, 0)
  eng = BatchedEngine(spec, 10)
  with pytest.raises(RuntimeError):
    eng.step(torch.zeros(11, dtype=torch.int8, device="cuda:0"))


@pytest.mark.parametrize("name", ["island_L9", "island_L5", "boat_ex_L3", "island_L0"])
def test_unoccluded_layers_match_fixture(name):
  """observation['layers'] of the MO envs: raw curtains + gap correction (rendering.py:188-302,
  observation_distiller_ex.py:164-178), from the rendered board + the spec's static layer tables."""
  fx, meta = G.load(name)
  spec = make_spec(meta["family_name"], **meta["kwargs"])
  n, S = fx["layers"].shape[:2]
  eng = BatchedEngine(spec, n * S, outputs=("board",))
  board = torch.from_numpy(fx["board"][:n].reshape(n * S, spec.H, spec.W).copy()).to("cuda:0")
  got = eng.observe_layers(board).cpu().numpy().astype(bool)
  assert "".join(spec.layer_chars) == meta["layer_chars"]
  G.assert_same(name + ".layers", got, fx["layers"].reshape(got.shape))


@pytest.mark.parametrize("env_name,kw,lo,n", [("island_navigation_ex", dict(level=9), 0, 5),
                                              ("boat_race_ex", dict(level=3), 0, 5)])
def test_full_baseline_size_matches_oracle(env_name, kw, lo, n):
  """BASELINE.json sizes: 65 536 envs, compared directly with the oracle (it finishes in seconds) and through
  size-independent properties: sharding invariance (two half-size engines keyed by global env id reproduce the
  full engine) and fused rollout == step loop on the device-generated action stream."""
  from oracle import oracle as O
  N_ENVS, T, seed = 65536, 24, 0x5AFE
  spec = make_spec(env_name, **kw)
  outs = ("board", "reward", "cumulative", "step_type", "term_reason")
  full = BatchedEngine(spec, N_ENVS, outputs=outs)
  full.reset()
  acts = full.fill_actions(T, seed)
  host_acts = philox.actions(seed, np.arange(N_ENVS), np.arange(T), lo, n)
  assert np.array_equal(acts.cpu().numpy(), host_acts)
  want = O.run_streams(O.make_config(env_name, **kw), host_acts.T.copy(),
                       fields=["board", "reward", "cumulative", "step_type", "term_reason"], nthreads=16)
  got = full.step_n(acts, write_every=True)
  for k in outs:
    g = got[k].cpu().numpy()                                    # [T, N, ...]
    w = np.moveaxis(want[k][:, 1:], 0, 1).reshape(g.shape)
    if k == "term_reason":
      g = g.astype(np.int16); g[g == 255] = -1
    assert np.array_equal(g, w), k
  # sharding invariance
  halves = [BatchedEngine(spec, N_ENVS // 2, env_id_base=b, outputs=outs) for b in (0, N_ENVS // 2)]
  ro = []
  for h in halves:
    h.reset()
    ro.append({k: v.clone() for k, v in h.rollout(T, seed, write_every=True).items()})
  for k in outs:
    both = torch.cat([ro[0][k], ro[1][k]], dim=1)
    assert torch.equal(both, got[k]), k
  ret = sum(h.read_returns() for h in halves)
  assert ret.abs().sum().item() == 0                              # accumulate was off


@pytest.mark.parametrize("name", ["island_L9", "island_L5", "island_L9_prop", "island_L0", "boat_ex_L3", "boat_ex_L0"])
def test_derived_statistics_match_fixture(name):
  """gini x100, the three variances and the average reward, computed ON DEVICE in numpy's
  reduction order (safety_game_mo.py:1027-1084, 1645-1681) -- bit-exact against the reference's values."""
  fx, meta = G.load(name)
  spec = make_spec(meta["family_name"], **meta["kwargs"])
  E = min(64, fx["actions"].shape[0]); T = 60
  eng = BatchedEngine(spec, E, outputs=("reward", "cumulative", "frame", "step_type"))
  acts = torch.from_numpy(np.ascontiguousarray(fx["actions"][:E, :T].T)).to("cuda:0")
  eng.reset()
  keys = ("gini_index", "cumulative_gini_index", "mo_variance", "cumulative_mo_variance", "average_mo_variance",
          "average_reward")
  rec = {k: [eng.derived_stats()[k].clone()] for k in keys}
  for t in range(T):
    eng.step(acts[t])
    d = eng.derived_stats()
    for k in keys:
      rec[k].append(d[k].clone())
  for k in keys:
    got = torch.stack(rec[k], dim=1).cpu().numpy()
    G.assert_same(name + "." + k, got, fx[k][:E, :T + 1])


@pytest.mark.parametrize("env_name,kw", [
    ("island_navigation_ex", {}), ("boat_race_ex", dict(level=3)), ("firemaker_ex_ma", dict(amount_agents=3)),
    ("island_navigation_ex_ma", dict(map_randomization_frequency=3, max_iterations=30)), ("tomato_watering", {}),
    ("aintelope_savanna", dict(amount_agents=2, amount_predators=2, amount_drink_holes=2, sustainability_challenge=True,
                               max_iterations=25)),
])
def test_checkpoint_resume_continues_bit_for_bit(env_name, kw):
  """sgw_get_state / sgw_set_state: a snapshot of the SoA state restored into a NEW engine (no seeding calls: the generator
  streams, draw counters and cached maps are state) continues with identical outputs and reaches an identical state."""
  spec = make_spec(env_name, **kw)
  n, T = 1000, 60
  outs = ("board", "reward", "cumulative", "step_type", "term_reason", "frame")
  a = _fresh(spec, n, outs)
  acts = a.fill_actions(T, 5)
  for t in range(T // 2):
    a.step(acts[t])
  snap = a.get_state().clone()
  want = [{k: v.clone() for k, v in a.step(acts[t]).items()} for t in range(T // 2, T)]
  b = BatchedEngine(spec, n, outputs=outs)
  b.set_episode_bits(None, seed=5)            # stateless Philox parameters of the engine, not env state
  b.set_random_stream(None, seed=6)
  b.set_state(snap)
  for i, t in enumerate(range(T // 2, T)):
    o = b.step(acts[t])
    for k in outs:
      assert torch.equal(o[k], want[i][k]), (k, t)
  assert torch.equal(a.get_state()[:, :n], b.get_state()[:, :n])



@pytest.mark.parametrize("name", [n for n in G.fixture_names(G.SCALAR_PREFIXES) if n.startswith("island_")])
def test_island_plain_state_variant_matches_reference_fixture(name, monkeypatch):
  """island_navigation_ex runs its packed (i16) state whenever sgw_create can prove it exact; every island fixture is also
  replayed through the plain f64 state (SGW_ISLAND_PLAIN_STATE, read at sgw_create), so both instantiations are pinned."""
  fx, meta = G.load(name)
  spec = make_spec(meta["family_name"], **meta["kwargs"])
  monkeypatch.setenv("SGW_ISLAND_PLAIN_STATE", "1")
  got = run_engine(spec, fx["actions"])
  compare(name, got, fx, spec.K)
  G.assert_same(name + ".metrics", got["metrics"][..., :spec.M], fx["metrics"])
  G.assert_same(name + ".safety", got["safety"], fx["safety"])


def test_island_packed_and_plain_states_have_different_sizes():
  """The default flags are packable (10 words = 80 B per env, SURVEY's figure); fractional reward flags are not."""
  import ctypes as C
  from ai_safety_gridworlds_amd import _native as N
  e = BatchedEngine(make_spec("island_navigation_ex"), 64)
  assert N.lib().sgw_state_words(e._h) == 10
  e.close()
  e = BatchedEngine(make_spec("island_navigation_ex", MOVEMENT_REWARD={"MOVEMENT_REWARD": -0.5}), 64)
  assert N.lib().sgw_state_words(e._h) == 22
  e.close()
