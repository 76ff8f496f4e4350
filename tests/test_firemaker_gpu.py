"""GPU parity for the multi-agent path (firemaker_ex_ma): fixtures from the patched reference run and
fresh seeds vs the multi-agent oracle.  Everything incl. the numpy PCG64 stream position must match."""
import numpy as np
import pytest
import torch

from ai_safety_gridworlds_amd import philox
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec
from tests import golden_util as G

pytestmark = pytest.mark.gpu

OUTS = ("board", "reward", "cumulative", "step_type", "term_reason", "discount", "metrics", "frame", "agent_pos",
        "safety", "obs_board", "agent_flags", "views", "obs_views")


def run(spec, actions, rng_states):
  E, T, A = actions.shape
  eng = BatchedEngine(spec, E, outputs=OUTS)
  eng.set_rng_state(rng_states)
  acts = torch.from_numpy(np.ascontiguousarray(np.transpose(actions, (1, 0, 2)))).to("cuda:0")   # [T, E, A]
  rec = {k: [] for k in OUTS}
  views = [[], [], []]
  def grab(o):
    for k in OUTS:
      rec[k].append(o[k].clone())
    for i, v in enumerate(eng.agent_views()):
      views[i].append(v.clone())
  grab(eng.reset())
  for t in range(T):
    grab(eng.step(acts[t]))
  torch.cuda.synchronize()
  out = {k: torch.stack(v, dim=1).cpu().numpy() for k, v in rec.items()}
  # the windows written by the step launch itself (sgw_out.views / obs_views) == the separate window kernel's, and the float
  # windows are the value-mapped ascii ones
  fused = [w.cpu().numpy() for w in eng.split_views(torch.stack(rec["views"], dim=1))]
  fused_f = [w.cpu().numpy() for w in eng.split_views(torch.stack(rec["obs_views"], dim=1))]
  vm = np.array([spec.native.value_map[i] for i in range(128)], np.float32)
  out["views"] = [torch.stack(v, dim=1).cpu().numpy() for v in views]
  for i in range(3):
    assert np.array_equal(fused[i], out["views"][i]), "in-kernel window of agent %d differs from sgw_agent_views" % i
    assert np.array_equal(fused_f[i], vm[fused[i]]), "float window of agent %d" % i
  st = eng.get_state()[:, :E].cpu().numpy().view(np.uint64)
  out["rng_final"] = np.stack([st[3], st[4], st[5], st[6]], axis=1)
  out["rng_has32_final"] = (st[0] >> np.uint64(27)) & np.uint64(1)
  out["rng_u32_final"] = st[2] & np.uint64(0xffffffff)
  eng.close()
  return out


TEMPLATE = ["ExternalVisits_1", "ExternalVisits_2", "ExternalVisits_S", "InternalVisits_1", "InternalVisits_2", "InternalVisits_S",
            "WorkshopVisits_1", "WorkshopVisits_2", "WorkshopVisits_S", "FireVisits_1", "FireVisits_2", "FireVisits_S",
            "StopButtonVisits_1", "StopButtonVisits_2", "StopButtonVisits_S", "StopButtonPressCountdown"]   # firemaker_ex_ma.py:123-140


def check(name, got, want, spec=None):
  """Columns follow the fixed ('1','2','S') layout; with amount_agents < 3 only the present agents' columns are compared
  (the reference's dicts have no key for the others) and the engine's metrics (the present rows, in order) are placed at
  their METRICS_LABELS_TEMPLATE rows."""
  slots = list(getattr(spec, "agent_slots", [0, 1, 2])) if spec is not None else [0, 1, 2]
  for f in ("step_type", "reward", "cumulative"):
    G.assert_same(name + "." + f, got[f][:, :, slots], want[f][:, :, slots])
  G.assert_same(name + ".discount", got["discount"], want["discount"])
  G.assert_same(name + ".frame", got["frame"], want["frame"])
  G.assert_same(name + ".board", got["board"], want["board"])
  metrics = np.zeros(got["metrics"].shape[:2] + (16,))
  names = spec.metric_names if spec is not None else TEMPLATE
  for j, lab in enumerate(names):
    metrics[..., TEMPLATE.index(lab)] = got["metrics"][..., j]
  G.assert_same(name + ".metrics", metrics, want["metrics"])
  G.assert_same(name + ".pos", got["agent_pos"][:, :, slots], want["pos"][:, :, slots])
  tr = got["term_reason"].astype(np.int16); tr[tr == 255] = -1
  G.assert_same(name + ".term_reason", tr, want["term_reason"][..., 0])
  if "layers" in getattr(want, "files", []):      # fire hidden under an agent sprite, from the engine state
    nl = want["layers"].shape[0]
    Fi = sorted(" #-12BFSW").index('F')
    for a in slots:
      p = want["pos"][:nl, :, a]
      hid = np.take_along_axis(want["layers"][:, :, Fi].reshape(nl, -1, 289), (p[..., 0] * 17 + p[..., 1])[..., None], axis=2)[..., 0]
      G.assert_same(name + ".agent_flags%d" % a, got["agent_flags"][:nl, :, a] & 1, hid.astype(np.uint8))
  if "action_direction" in (getattr(want, "files", None) or want):        # Directions per agent (bits 1-2 / 3-4 of agent_flags)
    G.assert_same(name + ".action_direction", ((got["agent_flags"] >> 1) & 3)[:, :, slots], want["action_direction"][:, :, slots])
    G.assert_same(name + ".observation_direction", ((got["agent_flags"] >> 3) & 3)[:, :, slots], want["observation_direction"][:, :, slots])
  for a in slots:
    if a < 2: G.assert_same(name + ".view_worker%d" % a, got["views"][a], want["view_worker"][:, :, a])
    else: G.assert_same(name + ".view_supervisor", got["views"][2], want["view_supervisor"])
  G.assert_same(name + ".rng", got["rng_final"], want["rng"][:, -1])
  G.assert_same(name + ".rng_has32", got["rng_has32_final"], want["rng_has_uint32"][:, -1])
  G.assert_same(name + ".rng_u32", got["rng_u32_final"] * got["rng_has32_final"],
                want["rng_uinteger"][:, -1].astype(np.uint64) * want["rng_has_uint32"][:, -1].astype(np.uint64))


@pytest.mark.parametrize("name", G.fixture_names(["firemaker_"]))
def test_firemaker_hip_matches_reference_fixture(name):
  fx, meta = G.load(name)
  spec = make_spec("firemaker_ex_ma", **meta["kwargs"])
  got = run(spec, fx["actions"], fx["rng_init"])
  check(name, got, fx, spec)
  G.assert_same(name + ".obs_board", got["obs_board"], fx["obs_board"])


@pytest.mark.parametrize("kw,E,T", [
    (dict(amount_agents=3, max_iterations=150), 700, 120),
    (dict(amount_agents=2, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.05, max_iterations=110), 500, 140),   # the reference's default agent set
    (dict(amount_agents=1, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.08, FIRE_CONTINUATION_PROBABILITY=0.97, max_iterations=200), 400, 220),
    (dict(amount_agents=3, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.04, FIRE_CONTINUATION_PROBABILITY=0.9,
          max_iterations=300), 300, 150),
    # direction modes: relative moves + rotated windows; the turning actions (action range 0..8); turning with fixed windows
    (dict(amount_agents=3, action_direction_mode=1, observation_direction_mode=1, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.04, max_iterations=130), 400, 150),
    (dict(amount_agents=3, action_direction_mode=2, observation_direction_mode=2, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.04, max_iterations=130), 400, 150),
    (dict(amount_agents=2, action_direction_mode=2, observation_direction_mode=0, max_iterations=80), 300, 100),
    (dict(amount_agents=1, action_direction_mode=1, observation_direction_mode=0, max_iterations=80), 200, 100),
    # FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE > 3: the WIDE kernels (sources up to 3 / 4 cells away; firemaker_ex_ma.py:255, 566-606)
    (dict(amount_agents=3, FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE=4.0, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.03, max_iterations=150), 300, 120),
    (dict(amount_agents=3, FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE=5.0, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.02, max_iterations=120), 300, 120),
    (dict(amount_agents=2, FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE=3.5, FIRE_CONTINUATION_PROBABILITY=0.9, max_iterations=100,
          action_direction_mode=1, observation_direction_mode=1), 200, 120),
])
def test_firemaker_hip_matches_oracle_fresh_seed(kw, E, T):
  from oracle import oracle_ma as OM
  seed = 0xF1E
  n_act = 9 if kw.get("action_direction_mode", 0) == 2 else 5
  actions = np.stack([philox.actions(seed, np.arange(E), np.arange(T), 0, n_act, agent=a) for a in range(3)], axis=-1)
  actions = np.transpose(actions, (1, 0, 2)).copy()                      # [E, T, 3]
  rng = np.stack([OM.rng_state_words(5000 + e) for e in range(E)])
  want = OM.run_streams(OM.make_config(**kw), actions, rng, nthreads=8)
  spec = make_spec("firemaker_ex_ma", **kw)
  got = run(spec, actions, rng)
  check("fresh", got, want, spec)


def test_firemaker_needs_rng_and_rollout_matches_steps():
  from ai_safety_gridworlds_amd import _native as N
  spec = make_spec("firemaker_ex_ma", amount_agents=3, max_iterations=90)
  n, T, seed = 640, 48, 3
  a = BatchedEngine(spec, n, outputs=("board", "reward", "step_type", "views"))
  with pytest.raises(N.SgwError, match="sgw_set_rng_state"):
    a.reset()
  b = BatchedEngine(spec, n, outputs=("board", "reward", "step_type", "views"))
  a.set_rng_seeds(np.arange(n) + 11); b.set_rng_seeds(np.arange(n) + 11)
  a.reset(); b.reset()
  acts = a.fill_actions(T, seed)
  assert acts.shape == (T, n, 3)
  ref = philox.actions(seed, np.arange(n), np.arange(T), 0, 5, agent=2)
  assert np.array_equal(acts[..., 2].cpu().numpy(), ref)
  per = {k: [] for k in ("board", "reward", "step_type", "views")}
  for t in range(T):
    o = a.step(acts[t])
    for k in per:
      per[k].append(o[k].clone())
  ro = b.rollout(T, seed, write_every=True)
  for k in per:
    assert torch.equal(torch.stack(per[k]), ro[k]), k
  assert torch.equal(a.get_state()[:, :n], b.get_state()[:, :n])


def test_firemaker_layers_and_agent_layer_cubes_match_fixture():
  """observation['layers'] and the per-agent crops of every layer (agent_perspectives_with_layers,
  safety_game_moma.py:430-525) from the rendered board + agent positions."""
  fx, meta = G.load("firemaker_L0_maxit60")
  spec = make_spec("firemaker_ex_ma", **meta["kwargs"])
  assert "".join(spec.layer_chars) == meta["layer_chars"]
  nl, S = fx["layers"].shape[:2]
  eng = BatchedEngine(spec, nl * S, outputs=("board", "agent_pos"))
  board = torch.from_numpy(fx["board"][:nl].reshape(nl * S, 17, 17).copy()).to("cuda:0")
  pos = torch.from_numpy(fx["pos"][:nl].reshape(nl * S, 3, 2).astype(np.uint8)).to("cuda:0")
  # fire hidden under an agent: in the fixture it is exactly (layer F at the agent's cell)
  fl = np.zeros((nl * S, 3), np.uint8)
  Fi = spec.layer_chars.index('F')
  fxl = fx["layers"].reshape(nl * S, 9, 17, 17); fxp = fx["pos"][:nl].reshape(nl * S, 3, 2)
  for i in range(nl * S):
    for a in range(3):
      fl[i, a] = fxl[i, Fi, fxp[i, a, 0], fxp[i, a, 1]]
  layers = eng.observe_layers(board, pos, torch.from_numpy(fl).to("cuda:0"))
  G.assert_same("layers", layers.cpu().numpy().astype(bool), fx["layers"].reshape(nl * S, 9, 17, 17))
  cubes = eng.agent_layer_views(layers, pos)
  G.assert_same("worker1", cubes[0].cpu().numpy().astype(bool), fx["agent_layers_worker"][:, :, 0].reshape(nl * S, 9, 5, 5))
  G.assert_same("worker2", cubes[1].cpu().numpy().astype(bool), fx["agent_layers_worker"][:, :, 1].reshape(nl * S, 9, 5, 5))
  G.assert_same("supervisor", cubes[2].cpu().numpy().astype(bool), fx["agent_layers_supervisor"].reshape(nl * S, 9, 33, 33))


def test_firemaker_views_of_a_masked_reset_and_of_step_n():
  """sgw_out.views under a masked reset (only the reset envs' rows change) and through sgw_step_n with write_every (the
  [T, N_pad, view_bytes] time slices), against the separate window kernel on the same boards."""
  spec = make_spec("firemaker_ex_ma", amount_agents=3, max_iterations=40)
  n, T = 200, 24                                                    # ragged: the last env-wave is partly padding
  eng = BatchedEngine(spec, n, outputs=("board", "agent_pos", "step_type", "views"))
  eng.set_rng_seeds(np.arange(n) + 5)
  eng.reset()
  acts = eng.fill_actions(T, 9)
  o = eng.step_n(acts, write_every=True)
  boards, pos, views = o["board"].clone(), o["agent_pos"].clone(), o["views"].clone()
  for t in range(T):
    want = eng.agent_views(board=boards[t].contiguous(), agent_pos=pos[t].contiguous())
    got = eng.split_views(views[t])
    for i in range(3):
      assert torch.equal(got[i], want[i]), (t, i)
  before = eng.step(acts[0])["views"].clone()
  mask = (torch.arange(n, device="cuda:0") % 3 == 0).to(torch.uint8)
  o = eng.reset(mask)
  want = eng.agent_views()
  got = eng.split_views(o["views"])
  m = mask.bool()
  for i in range(3):
    assert torch.equal(got[i][m], want[i][m]), i
  assert torch.equal(o["views"][~m], before[~m])                    # the other envs' rows were left alone
  eng.close()


@pytest.mark.parametrize("agent", ["1", "S"])
def test_firemaker_through_the_gym_facade_with_agent_character(agent):
  """GridworldGymEnv("firemaker_ex_ma", agent_character=...) steps ONE agent per step ({agent: action}: gridworld_gym_env.py:
  182-189, 476-479; EnvironmentMa.step plays exactly the submitted agents, pycolab_interface_ma.py:173-246) -- against the
  multi-agent oracle fed the same single-agent rounds (pinned on the reference run firemaker_L0_subset)."""
  from oracle import oracle_ma as OM
  from ai_safety_gridworlds_amd.helpers.gridworld_gym_env import GridworldGymEnv
  kw = dict(amount_agents=3, max_iterations=50, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.05)
  T, seed, q = 120, 31, {"1": 0, "2": 1, "S": 2}[agent]
  own = philox.actions(seed, np.arange(1), np.arange(T), 0, 5, agent=q)[:, 0]
  actions = np.full((1, T, 3), -1, np.int8); actions[0, :, q] = own
  want = OM.run_streams(OM.make_config(**kw), actions, OM.rng_state_words(seed)[None])
  env = GridworldGymEnv("firemaker_ex_ma", agent_character=agent, seed=seed, **kw)
  state, info = env.reset()
  view = want["view_worker"][0, :, q] if q < 2 else want["view_supervisor"][0]
  assert np.array_equal(state[0], np.vectorize(chr)(view[0]))
  k = 2 if q < 2 else 3
  for t in range(T):
    state, reward, terminated, truncated, info = env.step(int(own[t]))
    st = int(want["step_type"][0, t + 1, q])
    assert np.array_equal(state[-1] if state.ndim == 3 else state, np.vectorize(chr)(view[t + 1])), t
    assert terminated == (st == 2) and truncated is False, t
    if st == 0:
      assert np.all(np.asarray(reward) == 0.0), t
    else:
      assert np.array_equal(reward, want["reward"][0, t + 1, q, :k]), t
      assert list(info["reward_dict"]) == (["ENERGY", "WORKSHOP"] if q < 2 else ["ENERGY", "EXTERNAL_FIRE", "TRESPASSING"])
      assert np.array_equal(info["cumulative_reward"], want["cumulative"][0, t + 1, q, :k]), t
  env.close()
