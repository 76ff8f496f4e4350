"""GPU parity for aintelope_savanna: fixtures from the reference run and fresh seeds vs the oracle.  Map generation
(tile-count removal + interior shuffle), resource tiles spawning / vanishing through Generator.choice, predators,
cooperation and visit-count rewards, overlapping layers on the rendered board, the None (NaN) rows of the metrics matrix
and the numpy PCG64 stream position must all match bit for bit."""
import numpy as np
import pytest
import torch

from ai_safety_gridworlds_amd import philox
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec
from tests import golden_util as G

pytestmark = pytest.mark.gpu

OUTS = ("board", "reward", "cumulative", "step_type", "term_reason", "discount", "metrics", "frame", "agent_pos",
        "safety", "safety2", "obs_board", "agent_flags")
RESET = -128


def run(spec, actions, rng_states):
  E, T, _ = actions.shape
  eng = BatchedEngine(spec, E, outputs=OUTS)
  eng.set_rng_state(rng_states)
  acts = torch.from_numpy(np.ascontiguousarray(np.transpose(actions, (1, 0, 2)))).to("cuda:0")   # [T, E, 2]
  rec = {k: [] for k in OUTS}
  views = [[], []]
  rng = []
  lay = []
  def grab(o):
    for k in OUTS:
      rec[k].append(o[k].clone())
    lay.append(eng.observe_layers().clone())
    for i, v in enumerate(eng.agent_views()):
      views[i].append(v.clone())
    st = eng.get_state()[:, :E].cpu().numpy().view(np.uint64)
    rng.append(np.concatenate([np.stack([st[3], st[4], st[5], st[6]], axis=1),
                               ((st[0] >> np.uint64(27)) & np.uint64(1))[:, None], (st[2] & np.uint64(0xffffffff))[:, None]], axis=1))
  grab(eng.reset())
  grab(eng.reset())
  for t in range(T):
    if actions[0, t, 0] == RESET:
      assert (actions[:, t, 0] == RESET).all()
      grab(eng.reset())
    else:
      grab(eng.step(acts[t]))
  torch.cuda.synchronize()
  out = {k: torch.stack(v, dim=1).cpu().numpy() for k, v in rec.items()}
  out["view"] = np.stack([torch.stack(v, dim=1).cpu().numpy() for v in views], axis=2)    # [E, S, 2, VS, VS]
  out["rng_all"] = np.stack(rng, axis=1)
  out["layers"] = torch.stack(lay, dim=1).cpu().numpy()                                   # [E, S, L, H, W], L = spec.layer_chars
  eng.close()
  return out


def check(name, got, want, K, A):
  sl = slice(1, None)
  E, S = want["step_type"].shape[:2]
  G.assert_same(name + ".step_type", got["step_type"][:, sl, :A], want["step_type"][:, sl])
  G.assert_same(name + ".frame", got["frame"][:, sl], want["frame"][:, sl])
  H, W = want["board"].shape[2:]
  G.assert_same(name + ".board", got["board"][:, sl].reshape(E, S - 1, H, W), want["board"][:, sl])
  G.assert_same(name + ".pos", got["agent_pos"][:, sl].reshape(E, S - 1, 2, 2)[:, :, :A], want["pos"][:, sl])
  G.assert_same(name + ".reward", got["reward"][:, sl].reshape(E, S - 1, 2, K)[:, :, :A], want["reward"][:, sl])
  G.assert_same(name + ".cumulative", got["cumulative"][:, sl].reshape(E, S - 1, 2, K)[:, :, :A], want["cumulative"][:, sl])
  G.assert_same(name + ".discount", got["discount"][:, sl], want["discount"][:, sl])
  G.assert_same(name + ".metrics", got["metrics"][:, sl], want["metrics"][:, sl])
  tr = got["term_reason"].astype(np.int16); tr[tr == 255] = -1
  G.assert_same(name + ".term_reason", tr[:, sl, :A], want["term_reason"][:, sl])
  G.assert_same(name + ".safety", got["safety"][:, sl, :A], want["safety"][:, sl])
  G.assert_same(name + ".safety2", got["safety2"][:, sl, :A], want["safety2"][:, sl])       # SV:837-844
  G.assert_same(name + ".action_direction", ((got["agent_flags"][:, sl] >> 1) & 3)[:, :, :A], want["action_direction"][:, sl])
  G.assert_same(name + ".observation_direction", ((got["agent_flags"][:, sl] >> 3) & 3)[:, :, :A], want["observation_direction"][:, sl])
  G.assert_same(name + ".view", got["view"][:, sl, :A], want["view"][:, sl])
  G.assert_same(name + ".rng", got["rng_all"][:, :, :4], want["rng"])                       # incl. slot 0: the constructor's draws
  G.assert_same(name + ".rng_has32", got["rng_all"][:, sl, 4], want["rng_has_uint32"][:, sl])
  G.assert_same(name + ".rng_u32", got["rng_all"][:, sl, 5] * got["rng_all"][:, sl, 4],
                want["rng_uinteger"][:, sl].astype(np.uint64) * want["rng_has_uint32"][:, sl].astype(np.uint64))


@pytest.mark.parametrize("name", G.fixture_names(["sav_"]))
def test_savanna_hip_matches_reference_fixture(name):
  fx, meta = G.load(name)
  spec = make_spec("aintelope_savanna", **meta["kwargs"])
  assert spec.dim_names == meta["dim_names"] and spec.metric_names == meta["metric_labels"]
  got = run(spec, fx["actions"], fx["rng_seeded"])
  check(name, got, fx, spec.K, spec.n_agents)
  G.assert_same(name + ".obs_board", got["obs_board"][:, 1:].reshape(fx["obs_board"][:, 1:].shape), fx["obs_board"][:, 1:])
  # unoccluded layers from the state: the reference's drape curtains (W P D F d f G S), the agents, walls, and the gap layer
  # "only where every other layer is blank"
  chars = spec.layer_chars
  if "layer_keys" in meta:                  # the keys of the reference's observation['layers'] (remove_unused_tile_types_from_layers)
    assert "".join(chars) == meta["layer_keys"], (chars, meta["layer_keys"])
  for li, ch in enumerate("WPDFdfGS"):
    if ch not in chars:                     # a drape the game was built without: the reference has no such layer (and no tile)
      assert not fx["layers"][:, 1:, li].any()
      continue
    G.assert_same(name + ".layer " + ch, got["layers"][:, 1:, chars.index(ch)], fx["layers"][:, 1:, li])
  occupied = fx["layers"][:, 1:, :8].any(axis=2) | (fx["board"][:, 1:] == ord('#'))
  for i in range(spec.n_agents):
    agent = np.zeros_like(occupied)
    E, S1 = agent.shape[:2]
    ee, ss = np.meshgrid(np.arange(E), np.arange(S1), indexing="ij")
    agent[ee, ss, fx["pos"][:, 1:, i, 0], fx["pos"][:, 1:, i, 1]] = True
    G.assert_same(name + ".layer agent", got["layers"][:, 1:, chars.index("01"[i])], agent.astype(np.uint8))
    occupied |= agent
  G.assert_same(name + ".layer gap", got["layers"][:, 1:, chars.index(' ')], (~occupied).astype(np.uint8))


RICH = dict(amount_predators=3, amount_water_tiles=3, amount_gold_deposits=3, amount_silver_deposits=2,
            amount_small_food_patches=2, amount_drink_holes=2, amount_small_drink_holes=2, observation_radius=[3, 3, 3, 3])


@pytest.mark.parametrize("kw,E,T,resets", [
    (dict(amount_agents=2, sustainability_challenge=True, penalise_oversatiation=True, max_iterations=90, **RICH), 700, 130, (40, 41, 99)),
    (dict(amount_agents=1, max_iterations=50, map_randomization_frequency=1, **RICH), 500, 120, (30, 77)),
    (dict(level=13, amount_agents=2, map_randomization_frequency=3, amount_food_patches=1, amount_drink_holes=1,
          amount_small_food_patches=1, amount_small_drink_holes=1, sustainability_challenge=True,
          use_satiation_proportional_reward=True, penalise_oversatiation=True, randomize_agent_actions_order=False,
          max_iterations=64, observation_radius=[1, 1, 1, 1]), 400, 100, (33,)),
])
def test_savanna_hip_matches_oracle_fresh_seed(kw, E, T, resets):
  from oracle import oracle_ma as OM
  from oracle import oracle_sav as OS
  seed = 0x5A7
  actions = np.stack([philox.actions(seed, np.arange(E), np.arange(T), 0, 5, agent=a) for a in range(2)], axis=-1)
  actions = np.transpose(actions, (1, 0, 2)).astype(np.int8).copy()                        # [E, T, 2]
  for t in resets:
    actions[:, t, :] = RESET
  rng = np.stack([OM.rng_state_words(9000 + e) for e in range(E)])
  want = dict(OS.run_streams(OS.make_config(**kw), actions, rng, nthreads=8))
  spec = make_spec("aintelope_savanna", **kw)
  got = run(spec, actions, rng)
  want["step_type"] = want["step_type"].astype(np.uint8)
  want["term_reason"] = want["term_reason"].astype(np.int16)
  check("fresh", got, want, spec.K, spec.n_agents)


def test_savanna_long_streams_match_oracle():
  """~10^6 rounds: rare paths (tile floods, predators on agents, long regrowth chains, many map generations) against the
  oracle; boards, rewards, metrics and the generator position of every tick."""
  from oracle import oracle_ma as OM
  from oracle import oracle_sav as OS
  kw = dict(amount_agents=2, sustainability_challenge=True, penalise_oversatiation=True, use_satiation_proportional_reward=True,
            max_iterations=150, amount_predators=4, amount_water_tiles=4, amount_gold_deposits=3, amount_silver_deposits=3,
            amount_small_food_patches=2, amount_drink_holes=3, amount_small_drink_holes=2, observation_radius=[1, 1, 1, 1])
  E, T = 3000, 330
  actions = np.stack([philox.actions(0xBEEF, np.arange(E), np.arange(T), 0, 5, agent=a) for a in range(2)], axis=-1)
  actions = np.transpose(actions, (1, 0, 2)).astype(np.int8).copy()
  actions[:, 100, :] = RESET; actions[:, 101, :] = RESET; actions[:, 250, :] = RESET
  rng = np.stack([OM.rng_state_words(50000 + e) for e in range(E)])
  want = dict(OS.run_streams(OS.make_config(**kw), actions, rng, nthreads=16))
  spec = make_spec("aintelope_savanna", **kw)
  got = run(spec, actions, rng)
  want["step_type"] = want["step_type"].astype(np.uint8)
  want["term_reason"] = want["term_reason"].astype(np.int16)
  check("long", got, want, spec.K, spec.n_agents)

