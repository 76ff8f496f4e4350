"""GPU parity for island_navigation_ex_ma: fixtures from the reference run and fresh seeds vs the oracle.  Per-agent
step types (LAST/DEAD), relative directions, rotated agent views, map randomisation draws and the numpy PCG64 stream
position must all match bit for bit."""
import numpy as np
import pytest
import torch

from ai_safety_gridworlds_amd import philox
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec
from tests import golden_util as G

pytestmark = pytest.mark.gpu

OUTS = ("board", "reward", "cumulative", "step_type", "term_reason", "discount", "metrics", "frame", "agent_pos",
        "safety", "obs_board", "agent_flags")
RESET = -128


def run(spec, actions, rng_states):
  """Replays the fixture protocol: constructor reset (slot 0), the caller's reset (slot 1), then per tick either an
  explicit reset (actions[:, t, 0] == RESET for every env) or one round."""
  E, T, A = actions.shape
  eng = BatchedEngine(spec, E, outputs=OUTS)
  eng.set_rng_state(rng_states)
  acts = torch.from_numpy(np.ascontiguousarray(np.transpose(actions, (1, 0, 2)))).to("cuda:0")   # [T, E, A]
  rec = {k: [] for k in OUTS}
  views = [[], []]
  rng = []
  def grab(o):
    for k in OUTS:
      rec[k].append(o[k].clone())
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
  out["view"] = np.stack([torch.stack(v, dim=1).cpu().numpy() for v in views], axis=2)    # [E, S, A, 5, 5]
  out["rng_all"] = np.stack(rng, axis=1)                                                  # [E, S, 6]
  eng.close()
  return out


def check(name, got, want, K):
  sl = slice(1, None)
  E, S = want["step_type"].shape[:2]
  G.assert_same(name + ".step_type", got["step_type"][:, sl], want["step_type"][:, sl])
  G.assert_same(name + ".reward", got["reward"][:, sl].reshape(E, S - 1, 2, K), want["reward"][:, sl])
  G.assert_same(name + ".cumulative", got["cumulative"][:, sl].reshape(E, S - 1, 2, K), want["cumulative"][:, sl])
  G.assert_same(name + ".discount", got["discount"][:, sl], want["discount"][:, sl])
  G.assert_same(name + ".frame", got["frame"][:, sl], want["frame"][:, sl])
  H, W = want["board"].shape[2:]
  G.assert_same(name + ".board", got["board"][:, sl].reshape(E, S - 1, H, W), want["board"][:, sl])
  G.assert_same(name + ".metrics", got["metrics"][:, sl], want["metrics"][:, sl])
  G.assert_same(name + ".pos", got["agent_pos"][:, sl].reshape(E, S - 1, 2, 2), want["pos"][:, sl])
  tr = got["term_reason"].astype(np.int16); tr[tr == 255] = -1
  G.assert_same(name + ".term_reason", tr[:, sl], want["term_reason"][:, sl])
  G.assert_same(name + ".safety", got["safety"][:, sl], want["safety"][:, sl])
  G.assert_same(name + ".action_direction", (got["agent_flags"][:, sl] >> 1) & 3, want["action_direction"][:, sl])
  G.assert_same(name + ".observation_direction", (got["agent_flags"][:, sl] >> 3) & 3, want["observation_direction"][:, sl])
  G.assert_same(name + ".view", got["view"][:, sl], want["view"][:, sl])
  G.assert_same(name + ".rng", got["rng_all"][:, :, :4], want["rng"])                       # incl. slot 0: the constructor's draws
  G.assert_same(name + ".rng_has32", got["rng_all"][:, sl, 4], want["rng_has_uint32"][:, sl])
  G.assert_same(name + ".rng_u32", got["rng_all"][:, sl, 5] * got["rng_all"][:, sl, 4],
                want["rng_uinteger"][:, sl].astype(np.uint64) * want["rng_has_uint32"][:, sl].astype(np.uint64))


@pytest.mark.parametrize("name", G.fixture_names(["ima_"]))
def test_island_ma_hip_matches_reference_fixture(name):
  fx, meta = G.load(name)
  spec = make_spec("island_navigation_ex_ma", **meta["kwargs"])
  if "layer_keys" in meta:                  # the keys of the reference's observation['layers'] (remove_unused_tile_types_from_layers)
    assert "".join(spec.layer_chars) == meta["layer_keys"], (spec.layer_chars, meta["layer_keys"])
  assert spec.dim_names == meta["dim_names"] and spec.metric_names == meta["metric_labels"]
  got = run(spec, fx["actions"], fx["rng_seeded"])
  check(name, got, fx, spec.K)
  G.assert_same(name + ".obs_board", got["obs_board"][:, 1:].reshape(fx["obs_board"][:, 1:].shape), fx["obs_board"][:, 1:])
  G.assert_same(name + ".art0", got["board"][:, 0].reshape(fx["art0"].shape), fx["art0"])


@pytest.mark.parametrize("kw,E,T,resets", [
    (dict(level=9, max_iterations=60), 1000, 150, (40, 41, 99)),
    (dict(level=10, map_randomization_frequency=3, penalise_oversatiation=True, sustainability_challenge=True,
          max_iterations=24), 600, 120, (10, 30, 31, 77)),
    (dict(level=4, map_randomization_frequency=2, randomize_agent_actions_order=False, max_iterations=30), 300, 80, (33,)),
    # resized maps of more than 64 cells: the 8-word map instantiation (IslandMaT<8>), 126 cells and the 128-cell ceiling
    (dict(level=9, map_width=14, map_height=9, map_randomization_frequency=3, action_direction_mode=2, observation_direction_mode=2,
          max_iterations=40), 500, 110, (20, 21, 80)),
    (dict(level=9, map_width=8, map_height=16, map_randomization_frequency=1, sustainability_challenge=True, max_iterations=36), 400, 90, (45,)),
])
def test_island_ma_hip_matches_oracle_fresh_seed(kw, E, T, resets):
  from oracle import oracle_ima as OI
  from oracle import oracle_ma as OM
  seed = 0x1A3
  actions = np.stack([philox.actions(seed, np.arange(E), np.arange(T), 0, 5, agent=a) for a in range(2)], axis=-1)
  actions = np.transpose(actions, (1, 0, 2)).astype(np.int8).copy()                        # [E, T, 2]
  for t in resets:
    actions[:, t, :] = RESET
  rng = np.stack([OM.rng_state_words(7000 + e) for e in range(E)])
  want = OI.run_streams(OI.make_config(**kw), actions, rng, nthreads=8)
  spec = make_spec("island_navigation_ex_ma", **kw)
  got = run(spec, actions, rng)
  want = dict(want)
  want["rng_has_uint32"] = want["rng_has_uint32"]; want["step_type"] = want["step_type"].astype(np.uint8)
  want["term_reason"] = want["term_reason"].astype(np.int16)
  check("fresh", got, want, spec.K)
