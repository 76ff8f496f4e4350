"""Randomised configuration cases, HIP engine vs the C oracle, for the flag-rich families (island_navigation_ex,
island_navigation_ex_ma, aintelope_savanna, firemaker_ex_ma): random levels / flags / tile amounts / map sizes / direction modes /
single-agent rounds / explicit resets; every output every step, metrics and the numpy generator position.  Used by
tests/test_fuzz_gpu.py (a bounded fixed-seed slice, in the GPU tier) and by tools/diag/fuzz_parity.py (open-ended sweeps).
A case returns (family, kwargs, True | False | reason) or None when the drawn configuration is one the reference refuses."""
import numpy as np
import torch

from ai_safety_gridworlds_amd import philox
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec
from oracle import oracle as O, oracle_ima as OI, oracle_ma as OM, oracle_sav as OS


def island_case(rnd, E=1500, T=160, nthreads=16):
  level = int(rnd.choice([2, 3, 4, 5, 6, 7, 8, 9]))
  kw = dict(level=level, sustainability_challenge=bool(rnd.integers(2)), thirst_hunger_death=bool(rnd.integers(2)),
            penalise_oversatiation=bool(rnd.integers(2)), use_satiation_proportional_reward=bool(rnd.integers(2)),
            max_iterations=int(rnd.integers(15, 120)))
  acts = philox.actions(int(rnd.integers(1 << 30)), np.arange(E), np.arange(T), 0, 5).T.copy()
  try:
    spec = make_spec("island_navigation_ex", **kw)
  except ValueError:
    return None
  want = O.run_streams(O.make_config("island_navigation_ex", **kw), acts, nthreads=nthreads)
  eng = BatchedEngine(spec, E, outputs=("board", "reward", "cumulative", "step_type", "metrics", "frame"))
  a = torch.from_numpy(np.ascontiguousarray(acts.T)).to("cuda:0")
  rec = {k: [eng.reset()[k].clone()] for k in ("board", "reward", "cumulative", "step_type", "metrics", "frame")}
  for t in range(T):
    o = eng.step(a[t])
    for k in rec: rec[k].append(o[k].clone())
  got = {k: torch.stack(v, 1).cpu().numpy() for k, v in rec.items()}
  eng.close()
  ok = True
  for k in ("reward", "cumulative", "step_type", "frame", "metrics"):
    g, w = got[k].reshape(want[k].shape) if k != "metrics" else got[k][..., :spec.M], want[k]
    ok &= bool(((g == w) | ((g != g) & (w != w))).all())
  ok &= bool((got["board"].reshape(want["board"].shape) == want["board"]).all())
  return ("island", kw, ok)

def ma_case(rnd, which, E=800, T=150, nthreads=16, strict=False):
  seed = int(rnd.integers(1 << 30))
  modes = [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 2)][int(rnd.integers(6))]     # (action, observation) direction modes
  n_act = 9 if modes[0] == 2 else 5                          # mode 2: the turning actions 5-8 join the action set
  actions = np.stack([philox.actions(seed, np.arange(E), np.arange(T), 0, n_act, agent=a) for a in range(2)], axis=-1)
  actions = np.transpose(actions, (1, 0, 2)).astype(np.int8).copy()
  for t in rnd.choice(T, 3, replace=False): actions[:, int(t), :] = -128
  if rnd.integers(2):                                      # some single-agent rounds
    for t in range(0, T, 4):
      if actions[0, t, 0] != -128: actions[:, t, int(rnd.integers(2))] = -1
  rng = np.stack([OM.rng_state_words(int(seed % 100000) + e) for e in range(E)])
  if which == "ima":
    kw = dict(level=int(rnd.choice([2, 3, 4, 5, 6, 7, 8, 9, 10])), sustainability_challenge=bool(rnd.integers(2)),
              penalise_oversatiation=bool(rnd.integers(2)), use_satiation_proportional_reward=bool(rnd.integers(2)),
              map_randomization_frequency=int(rnd.integers(4)), action_direction_mode=modes[0],
              observation_direction_mode=modes[1], max_iterations=int(rnd.integers(12, 80)))
    if kw["map_randomization_frequency"] >= 1 and rnd.integers(3) == 0:      # resized island: up to 128 cells (> 64: the 8-word map kernels)
      w = int(rnd.integers(4, 17)); kw.update(map_width=w, map_height=int(rnd.integers(4, min(13, 128 // w + 1))))
    name, Or = "island_navigation_ex_ma", OI
  else:
    two = bool(rnd.integers(2))
    kw = dict(amount_agents=2 if two else 1, sustainability_challenge=bool(rnd.integers(2)), penalise_oversatiation=bool(rnd.integers(2)),
              use_satiation_proportional_reward=bool(rnd.integers(2)), map_randomization_frequency=int(rnd.integers(1, 4)),
              amount_food_patches=int(rnd.integers(1, 4)), amount_drink_holes=int(rnd.integers(0, 3)), amount_small_food_patches=int(rnd.integers(0, 3)),
              amount_small_drink_holes=int(rnd.integers(0, 3)), amount_gold_deposits=int(rnd.integers(0, 4)), amount_silver_deposits=int(rnd.integers(0, 4)),
              amount_water_tiles=int(rnd.integers(0, 5)), amount_predators=int(rnd.integers(0, 5)), max_iterations=int(rnd.integers(20, 120)),
              observation_radius=[2, 2, 2, 2], action_direction_mode=modes[0], observation_direction_mode=modes[1])
    if rnd.integers(3) == 0: kw.update(map_width=int(rnd.integers(7, 14)), map_height=int(rnd.integers(7, 13)))
    if not two: actions[:, :, 1] = np.where(actions[:, :, 0:1].repeat(1, 2)[:, :, 0] == -128, actions[:, :, 1], 0)
    name, Or = "aintelope_savanna", OS
  try:
    spec = make_spec(name, **kw)
  except (ValueError, NotImplementedError, AssertionError, RuntimeError):
    return None
  if name == "aintelope_savanna" and kw["amount_agents"] == 1:
    actions[:, :, 0] = np.where(actions[:, :, 0] == -1, 0, actions[:, :, 0])
  try:
    want = Or.run_streams(Or.make_config(**kw), actions, rng, nthreads=nthreads)
  except ValueError as ex:
    return (which, kw, "oracle refused: %s" % ex)
  from ai_safety_gridworlds_amd.engine import fused_views
  fused = fused_views(spec)
  outs = ("board", "reward", "cumulative", "step_type", "frame", "metrics", "agent_pos", "agent_flags") + (("views",) if fused else ())
  eng = BatchedEngine(spec, E, outputs=outs); eng.set_rng_state(rng)
  a = torch.from_numpy(np.ascontiguousarray(np.transpose(actions, (1, 0, 2)))).to("cuda:0")
  rec = {k: [] for k in outs}
  views_ok = [True]
  def grab(o):
    for k in outs: rec[k].append(o[k].clone())
    # the windows written by the round's launch == the separate window kernel on the same board (rot90-ed by the observation direction)
    if fused:
      for f, w in zip(eng.split_views(o["views"]), eng.agent_views()):
        views_ok[0] &= bool(torch.equal(f, w))
  grab(eng.reset()); grab(eng.reset())
  for t in range(T):
    grab(eng.reset() if actions[0, t, 0] == -128 else eng.step(a[t]))
  st = eng.get_state()[:, :E].cpu().numpy().view(np.uint64)
  got = {k: torch.stack(v, 1).cpu().numpy() for k, v in rec.items()}
  eng.close()
  A = want["step_type"].shape[2]
  S = T + 2
  # Rounds on a FINISHED episode: the agents that count are the DEAD ones when there is one (the reference raises for a LAST
  # agent next to a DEAD one), everybody otherwise; with none of them submitted nothing resets and LAST turns into DEAD.
  # Engine and oracle follow the same rule; `strict` as 4th argument leaves such streams out of the comparison
  done_before = (want["step_type"][:, 1:S - 1] >= 2).all(axis=2)                      # [E, T]: status when tick t is submitted
  partial = (actions[:, :, :A] == -1).any(axis=2)
  valid = ~(done_before & partial).any(axis=1) if strict else np.ones(E, bool)
  v = valid
  ok = views_ok[0] and bool((got["board"][v, 1:].reshape(want["board"][v, 1:].shape) == want["board"][v, 1:]).all())
  ok &= bool((got["step_type"][v, 1:, :A] == want["step_type"][v, 1:]).all())
  g = got["reward"][v, 1:].reshape(int(v.sum()), S - 1, 2, spec.K)[:, :, :A]; ok &= bool((g == want["reward"][v, 1:]).all())
  g = got["cumulative"][v, 1:].reshape(int(v.sum()), S - 1, 2, spec.K)[:, :, :A]; ok &= bool((g == want["cumulative"][v, 1:]).all())
  g, w = got["metrics"][v, 1:], want["metrics"][v, 1:]; ok &= bool(((g == w) | ((g != g) & (w != w))).all())
  rngw = np.stack([st[3], st[4], st[5], st[6]], axis=1); ok &= bool((rngw[v] == want["rng"][v, -1]).all())
  return (which, dict(kw, _valid_streams=int(v.sum())), ok)


def firemaker_case(rnd, E=400, T=120, nthreads=16):
  """firemaker_ex_ma: agent set, fire probabilities, episode length, action-order shuffle; engine vs the multi-agent oracle incl.
  the windows written by the step launch and the generator position."""
  seed = int(rnd.integers(1 << 30))
  kw = dict(amount_agents=int(rnd.integers(1, 4)), max_iterations=int(rnd.integers(20, 200)),
            FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=float(rnd.choice([0.01, 0.02, 0.05, 0.08])),
            FIRE_CONTINUATION_PROBABILITY=float(rnd.choice([0.9, 0.95, 0.97])))
  adm, odm = [(0, 0), (0, 0), (1, 1), (2, 2), (2, 0), (1, 0), (0, 1)][int(rnd.integers(7))]      # the combinations the reference survives
  kw.update(action_direction_mode=adm, observation_direction_mode=odm)
  kw["FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE"] = float(rnd.choice([3.0, 3.0, 3.0, 2.5, 3.5, 4.0, 4.5, 5.0]))      # > 3: the WIDE kernels
  actions = np.stack([philox.actions(seed, np.arange(E), np.arange(T), 0, 9 if adm == 2 else 5, agent=a) for a in range(3)], axis=-1)
  actions = np.transpose(actions, (1, 0, 2)).copy()
  rng = np.stack([OM.rng_state_words(int(seed % 100000) + e) for e in range(E)])
  spec = make_spec("firemaker_ex_ma", **kw)
  want = OM.run_streams(OM.make_config(**kw), actions, rng, nthreads=nthreads)
  outs = ("board", "reward", "cumulative", "step_type", "frame", "metrics", "agent_pos", "agent_flags", "views")
  eng = BatchedEngine(spec, E, outputs=outs); eng.set_rng_state(rng)
  a = torch.from_numpy(np.ascontiguousarray(np.transpose(actions, (1, 0, 2)))).to("cuda:0")
  rec = {k: [eng.reset()[k].clone()] for k in outs}
  for t in range(T):
    o = eng.step(a[t])
    for k in outs: rec[k].append(o[k].clone())
  st = eng.get_state()[:, :E].cpu().numpy().view(np.uint64)
  got = {k: torch.stack(v, 1).cpu() for k, v in rec.items()}
  views = [w.numpy() for w in eng.split_views(got["views"])]
  got = {k: v.numpy() for k, v in got.items()}
  eng.close()
  slots = list(getattr(spec, "agent_slots", [0, 1, 2]))
  ok = bool((got["board"] == want["board"]).all()) and bool((got["frame"] == want["frame"]).all())
  for f in ("step_type", "reward", "cumulative"):
    ok &= bool((got[f][:, :, slots] == want[f][:, :, slots]).all())
  ok &= bool((got["agent_pos"][:, :, slots] == want["pos"][:, :, slots]).all())
  ok &= bool((((got["agent_flags"] >> 1) & 3)[:, :, slots] == want["action_direction"][:, :, slots]).all())
  ok &= bool((((got["agent_flags"] >> 3) & 3)[:, :, slots] == want["observation_direction"][:, :, slots]).all())
  for q in slots:
    ok &= bool((views[q] == (want["view_worker"][:, :, q] if q < 2 else want["view_supervisor"])).all())
  ok &= bool((np.stack([st[3], st[4], st[5], st[6]], axis=1) == want["rng"][:, -1]).all())
  return ("firemaker", kw, ok)
