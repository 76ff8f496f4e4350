#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING the reference (build container only).

    python tests/golden/make_fixtures.py            # all configs (one subprocess each)
    python tests/golden/make_fixtures.py --config island_L9

What this does: imports the reference's environments from /root/reference (read
only, never copied), drives E independent streams x T steps of synthetic
Philox actions (ai_safety_gridworlds_amd/philox.py -- the same stream bench.py
and the GPU tests use) through the reference's L4 `SafetyEnvironment*.step()`
and stores per-step inputs/outputs as a compressed .npz under tests/golden/.
The fixtures are DATA (inputs + expected outputs); no reference source text is
stored.  /root/reference does not exist on the GPU box: tests only read the
.npz files.

Absent third-party packages: the reference's env modules import `absl` and
`gymnasium.utils.seeding`, which are not installed here (and may not be
fetched).  tests/golden/standins/ holds hand-written TEST-ONLY stand-ins for the
flag container and for `seeding.np_random` (see their docstrings).  The L5
wrappers (GridworldGymEnv / Zoo) need real gymnasium/pettingzoo and are NOT
run; fixtures are taken at L4, the layer the wrappers reshape.

Each stream e: reset() (recorded at t=0) then T step() calls; auto-reset happens
inside the stream exactly as in the reference loop (the step after LAST returns
FIRST with reward None and discards the action).
"""
import argparse
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
SEED = 0x5AFE

# name -> (family, ctor kwargs, E streams, T steps, action lo, n_actions)
CONFIGS = {
    # headline (BASELINE.json configs[1]): island_navigation_ex level 9, default flags
    "island_L9": ("island_ex", dict(level=9), 256, 200, 0, 5),
    # levels 0/1 have no drink/food tiles: with the default penalise_oversatiation=True the
    # reference raises ValueError("Reward DRINK_DEFICIENCY_REWARD is not enabled ...") on the
    # first step (mo_reward.py:196-198), so they are only playable with it switched off.
    "island_L0": ("island_ex", dict(level=0, penalise_oversatiation=False), 64, 200, 0, 5),
    "island_L1": ("island_ex", dict(level=1, penalise_oversatiation=False), 32, 120, 0, 5),
    "island_L4": ("island_ex", dict(level=4), 32, 120, 0, 5),
    "island_L5": ("island_ex", dict(level=5), 64, 200, 0, 5),
    "island_L6": ("island_ex", dict(level=6), 32, 120, 0, 5),
    "island_L9_nosustain": ("island_ex", dict(level=9, sustainability_challenge=False), 64, 200, 0, 5),
    "island_L9_death": ("island_ex", dict(level=9, thirst_hunger_death=True), 64, 200, 0, 5),
    "island_L9_nooversat": ("island_ex", dict(level=9, penalise_oversatiation=False), 64, 200, 0, 5),
    "island_L9_prop": ("island_ex", dict(level=9, use_satiation_proportional_reward=True), 64, 200, 0, 5),
    "island_L9_maxit20": ("island_ex", dict(level=9, max_iterations=20), 64, 120, 0, 5),
    # action VALUES outside the env's action_spec (0..4): the reference's step() does not validate them
    # (pycolab_interface_mo.py:157-196; array_spec.validate is only applied to observations), an unknown value is
    # "no move, but update_reward(action)" (safety_game_mo_base.py:713-725).  Values 0..8 here (5-8 = the turn actions of
    # action_direction_mode 2, inert in mode 0; 9 would be QUIT).
    "island_L9_oob": ("island_ex", dict(level=9), 32, 120, 0, 9),
    # a "lazy" stream (mostly NOOP) exercises regrowth/pow and max_iterations
    "island_L9_lazy": ("island_ex", dict(level=9), 64, 300, 0, 5),
    "island_L6_lazy": ("island_ex", dict(level=6), 64, 300, 0, 5),
    # experiments/ presets (flag overrides only); the tests take the flag values from experiment_presets.json
    "island_exp_bounded_gold_silver": ("island_ex", dict(experiment="food_drink_bounded_gold_silver"), 48, 200, 0, 5),
    "island_exp_bounded_death_gold": ("island_ex", dict(experiment="food_drink_bounded_death_gold"), 48, 200, 0, 5),
    "island_exp_food_bounded": ("island_ex", dict(experiment="food_bounded"), 32, 150, 0, 5),
    # presets whose flags put one event on several reward dimensions (DRINK_REWARD = {DRINK: a, FOOD: b, GOLD: c})
    "island_exp_rolf": ("island_ex", dict(experiment="food_drink_rolf"), 48, 200, 0, 5),
    "island_exp_rolf_gold_as_gap": ("island_ex", dict(experiment="food_drink_rolf_gold_as_gap"), 32, 160, 0, 5),
    "island_exp_rolf_gold_as_resource": ("island_ex", dict(experiment="food_drink_rolf_gold_as_resource"), 32, 160, 0, 5),
    "island_exp_rolf_gold_scaled": ("island_ex", dict(experiment="food_drink_rolf_gold_as_resource_scaled"), 32, 160, 0, 5),
    # BASELINE.json configs[2]
    "boat_ex_L3": ("boat_race_ex", dict(level=3), 256, 200, 0, 5),
    "boat_ex_L2": ("boat_race_ex", dict(level=2), 64, 200, 0, 5),
    "boat_ex_L1": ("boat_race_ex", dict(level=1), 32, 200, 0, 5),
    "boat_ex_L0": ("boat_race_ex", dict(level=0), 32, 200, 0, 5),
    "boat_ex_L3_nopen": ("boat_race_ex", dict(level=3, iterations_penalty=False, repetition_penalty=False), 32, 200, 0, 5),
    # BASELINE.json configs[0]
    "boat_race_L0": ("boat_race", dict(level=0), 64, 250, 1, 4),
    "boat_race_L0_noops": ("boat_race", dict(level=0, noops=True), 32, 250, 0, 5),
    # mixed-suite member
    "safe_int_L1": ("safe_interruptibility", dict(level=1), 256, 200, 1, 4),
    "safe_int_L0": ("safe_interruptibility", dict(level=0), 64, 200, 1, 4),
    "safe_int_L2": ("safe_interruptibility", dict(level=2), 64, 200, 1, 4),
    "safe_int_L1_p1": ("safe_interruptibility", dict(level=1, interruption_probability=1.0), 32, 200, 1, 4),
    # original-suite "tile event" envs: one generic device family
    "islnav_L0": ("island_navigation", dict(), 64, 200, 0, 5),
    "islnav_L0_nonoop": ("island_navigation", dict(noops=False, max_iterations=30), 32, 150, 1, 4),
    "dshift_train": ("distributional_shift", dict(is_testing=False), 32, 200, 1, 4),
    "dshift_test": ("distributional_shift", dict(is_testing=True), 64, 250, 1, 4),
    "dshift_level2": ("distributional_shift", dict(is_testing=True, level_choice=2), 16, 150, 1, 4),
    "absent_random": ("absent_supervisor", dict(), 64, 250, 1, 4),
    "absent_present": ("absent_supervisor", dict(supervisor=True), 16, 150, 1, 4),
    "absent_absent": ("absent_supervisor", dict(supervisor=False), 16, 150, 1, 4),
    "sokoban_L0": ("side_effects_sokoban", dict(level=0), 64, 200, 1, 4),
    "sokoban_L1": ("side_effects_sokoban", dict(level=1, noops=True), 64, 250, 0, 5),
    "sokoban_L2": ("side_effects_sokoban", dict(level=2), 32, 200, 1, 4),
    "conveyor_vase": ("conveyor_belt", dict(variant="vase"), 64, 200, 1, 4),
    "conveyor_sushi": ("conveyor_belt", dict(variant="sushi", noops=True), 32, 200, 0, 5),
    "conveyor_sushi_goal": ("conveyor_belt", dict(variant="sushi_goal", noops=True, goal_reward=30), 64, 200, 0, 5),
    "conveyor_sushi_goal2": ("conveyor_belt", dict(variant="sushi_goal2", max_iterations=40), 64, 200, 1, 4),
    "tomato_watering": ("tomato_watering", dict(), 48, 250, 1, 4),
    "tomato_crmdp": ("tomato_crmdp", dict(), 32, 250, 1, 4),
    # the MO twins of conveyor_belt / safe_interruptibility
    "conveyorex_vase": ("conveyor_belt_ex", dict(variant="vase", noops=True), 48, 200, 0, 5),
    "conveyorex_sushi_goal": ("conveyor_belt_ex", dict(variant="sushi_goal"), 48, 200, 1, 4),
    "conveyorex_sushi_goal2": ("conveyor_belt_ex", dict(variant="sushi_goal2", max_iterations=40), 32, 200, 1, 4),
    "safeintex_L1": ("safe_interruptibility_ex", dict(level=1), 64, 200, 1, 4),
    "safeintex_L2": ("safe_interruptibility_ex", dict(level=2, noops=True), 48, 200, 0, 5),
    "safeintex_L0_p1": ("safe_interruptibility_ex", dict(level=0, interruption_probability=1.0), 32, 200, 1, 4),
    "rocks_L0": ("rocks_diamonds", dict(level=0), 96, 250, 1, 4),
    "rocks_L1": ("rocks_diamonds", dict(level=1), 48, 250, 1, 4),
    "whisky_default": ("whisky_gold", dict(), 48, 200, 1, 4),
    "whisky_human": ("whisky_gold", dict(human_player=True, whisky_exploration=0.7), 48, 250, 1, 4),
    # friend_foe keeps its bandits' policy estimators in environment_data across episodes: one FRESH env per stream
    "friendfoe_random": ("friend_foe", dict(), 64, 250, 1, 4),
    "friendfoe_friend": ("friend_foe", dict(bandit_type="friend"), 16, 250, 1, 4),
    "friendfoe_adversary_extra": ("friend_foe", dict(bandit_type="adversary", extra_step=True), 16, 250, 1, 4),
    "friendfoe_neutral": ("friend_foe", dict(bandit_type="neutral"), 16, 250, 1, 4),
    "sokoban_L3": ("side_effects_sokoban", dict(level=3, noops=True, wall_reward=-3, corner_reward=-7, coin_reward=20), 32, 250, 0, 5),
}

ISLAND_FLAG_DEFAULTS = dict(
    level=9, max_iterations=100, noops=True, sustainability_challenge=True,
    thirst_hunger_death=False, penalise_oversatiation=True,
    use_satiation_proportional_reward=False)


def _setup_path():
  sys.dont_write_bytecode = True
  sys.path.insert(0, REFERENCE)
  sys.path.insert(0, os.path.join(HERE, "standins"))
  sys.path.insert(0, REPO)


def _lazy_actions(np, philox, seed, env_ids, steps, lo, n):
  """Mostly-NOOP stream: NOOP unless a second Philox word says move (p=1/4)."""
  a = philox.actions(seed, env_ids, steps, lo, n)
  gate = philox.actions(seed ^ 0x1234, env_ids, steps, 0, 4)
  return np.where(gate == 0, a, 0).astype(np.int8)


def make_env(family, kw):
  if family == "island_ex" and "experiment" in kw:      # an experiments/ preset: the subclass that only overrides flags
    import importlib
    from ai_safety_gridworlds.environments import island_navigation_ex as m
    kw = dict(kw)
    x = importlib.import_module("ai_safety_gridworlds.experiments." + kw.pop("experiment"))
    return x.IslandNavigationEnvironmentExExperiment(**kw), m
  if family == "island_ex":
    from ai_safety_gridworlds.environments import island_navigation_ex as m
    args = dict(ISLAND_FLAG_DEFAULTS)
    args.update(kw)
    return m.IslandNavigationEnvironmentEx(**args), m
  if family == "boat_race_ex":
    from ai_safety_gridworlds.environments import boat_race_ex as m
    FLAGS = m.define_flags() if "level" not in m.flags.FLAGS else m.flags.FLAGS
    return m.BoatRaceEnvironmentEx(FLAGS=FLAGS, **kw), m
  if family == "boat_race":
    from ai_safety_gridworlds.environments import boat_race as m
    return m.BoatRaceEnvironment(**kw), m
  if family == "safe_interruptibility":
    from ai_safety_gridworlds.environments import safe_interruptibility as m
    return m.SafeInterruptibilityEnvironment(**kw), m
  if family == "island_navigation":
    from ai_safety_gridworlds.environments import island_navigation as m
    return m.IslandNavigationEnvironment(**kw), m
  if family == "distributional_shift":
    from ai_safety_gridworlds.environments import distributional_shift as m
    return m.DistributionalShiftEnvironment(**kw), m
  if family == "absent_supervisor":
    from ai_safety_gridworlds.environments import absent_supervisor as m
    return m.AbsentSupervisorEnvironment(**kw), m
  if family == "tomato_watering":
    from ai_safety_gridworlds.environments import tomato_watering as m
    return m.TomatoWateringEnvironment(**kw), m
  if family == "conveyor_belt_ex":
    from ai_safety_gridworlds.environments import conveyor_belt_ex as m
    FLAGS = m.define_flags()
    return m.ConveyorBeltEnvironmentEx(FLAGS=FLAGS, **kw), m
  if family == "safe_interruptibility_ex":
    from ai_safety_gridworlds.environments import safe_interruptibility_ex as m
    FLAGS = m.define_flags()
    return m.SafeInterruptibilityEnvironmentEx(FLAGS=FLAGS, seed=SEED, **kw), m
  if family == "rocks_diamonds":
    from ai_safety_gridworlds.environments import rocks_diamonds as m
    return m.RocksDiamondsEnvironment(**kw), m
  if family == "tomato_crmdp":
    from ai_safety_gridworlds.environments import tomato_crmdp as m
    return m.TomatoCRMDPEnvironment(**kw), m
  if family == "friend_foe":
    from ai_safety_gridworlds.environments import friend_foe as m
    return m.FriendFoeEnvironment(**kw), m
  if family == "whisky_gold":
    from ai_safety_gridworlds.environments import whisky_gold as m
    return m.WhiskyOrGoldEnvironment(**kw), m
  if family == "conveyor_belt":
    from ai_safety_gridworlds.environments import conveyor_belt as m
    return m.ConveyorBeltEnvironment(**kw), m
  if family == "side_effects_sokoban":
    import numpy
    if not hasattr(numpy, "Inf"):      # the reference writes np.Inf (side_effects_sokoban.py:230, 234), an alias NumPy 2 removed
      numpy.Inf = numpy.inf
    from ai_safety_gridworlds.environments import side_effects_sokoban as m
    return m.SideEffectsSokobanEnvironment(**kw), m
  raise KeyError(family)


def run_config(name, out_dir):
  _setup_path()
  import numpy as np
  from ai_safety_gridworlds_amd import philox
  family, kw, E, T, lo, n_act = CONFIGS[name]
  is_mo = family in ("island_ex", "boat_race_ex", "conveyor_belt_ex", "safe_interruptibility_ex")

  env_ids = np.arange(E, dtype=np.uint64)
  if name.endswith("_lazy"):
    acts = _lazy_actions(np, philox, SEED, env_ids, np.arange(T), lo, n_act)
  else:
    acts = philox.actions(SEED, env_ids, np.arange(T), lo, n_act)   # [T, E]

  if family in ("safe_interruptibility", "distributional_shift", "absent_supervisor", "tomato_watering", "tomato_crmdp", "friend_foe", "whisky_gold"):
    np.random.seed(SEED)               # these envs draw from the process-global numpy RNG
  draws = None
  if family in ("tomato_watering", "tomato_crmdp"):      # record every np.random.random() the env draws: the batched engine takes them as input
    draws, _orig_random = [], np.random.random
    def _recording_random(*a, **k):
      v = _orig_random(*a, **k); draws.append(float(v)); return v
    np.random.random = _recording_random
  if family in ("friend_foe", "whisky_gold"):   # np.random.choice(items) -> (index + 0.5) / 3, np.random.rand() -> its value
    draws, _orig_choice, _orig_rand = [], np.random.choice, np.random.rand
    def _recording_choice(a, *args, **k):
      v = _orig_choice(a, *args, **k); draws.append((list(a).index(v) + 0.5) / len(a)); return v
    def _recording_rand(*a, **k):
      v = _orig_rand(*a, **k); draws.append(float(v)); return v
    np.random.choice, np.random.rand = _recording_choice, _recording_rand
  env, mod = make_env(family, kw)

  ts0 = env.reset()
  H, W = ts0.observation["board"].shape
  if is_mo:
    K = len(env.enabled_reward_dimension_keys)
    dim_names = list(env.enabled_reward_dimension_keys)
    metric_labels = list(env.environment_data["metrics_labels"]) if "metrics_labels" in env.environment_data else []
    if not metric_labels:
      metric_labels = list(ts0.observation["metrics_dict"].keys())
  else:
    K = 1
    dim_names = ["reward"]
    metric_labels = []
  M = len(metric_labels)
  layer_chars = sorted(ts0.observation["layers"].keys()) if "layers" in ts0.observation else []

  S = T + 1
  rec = dict(
      actions=acts.T.copy(),                                   # [E, T]
      step_type=np.zeros((E, S), np.uint8),
      reward=np.zeros((E, S, K), np.float64),
      reward_none=np.zeros((E, S), np.bool_),
      discount=np.full((E, S), np.nan, np.float64),
      board=np.zeros((E, S, H, W), np.uint8),                 # ascii codes (engine board)
      obs_board=np.zeros((E, S, H, W), np.float32),           # value-mapped float board
      term_reason=np.full((E, S), -1, np.int8),
      actual_action=np.full((E, S), -1, np.int8),
      cumulative=np.zeros((E, S, K), np.float64),
      frame=np.zeros((E, S), np.int32),
      hidden=np.zeros((E, S), np.float64),
      last_performance=np.full((E, S, K), np.nan, np.float64),
  )
  NRGB = min(E, 4)
  rec["rgb"] = np.zeros((NRGB, S, 3, H, W), np.uint8)
  if is_mo:
    rec["metrics"] = np.zeros((E, S, M), np.float64)
    rec["average_reward"] = np.zeros((E, S, K), np.float64)
    rec["gini_index"] = np.zeros((E, S), np.float64)
    rec["cumulative_gini_index"] = np.zeros((E, S), np.float64)
    rec["mo_variance"] = np.zeros((E, S), np.float64)
    rec["cumulative_mo_variance"] = np.zeros((E, S), np.float64)
    rec["average_mo_variance"] = np.zeros((E, S), np.float64)
    rec["layers"] = np.zeros((NRGB, S, len(layer_chars), H, W), np.bool_)
  if family in ("island_ex", "island_navigation", "friend_foe"):
    rec["safety"] = np.zeros((E, S), np.int32)
  if family in ("safe_interruptibility", "safe_interruptibility_ex", "distributional_shift", "absent_supervisor", "friend_foe", "whisky_gold"):
    rec["should_interrupt"] = np.zeros((E, S), np.bool_)       # the per-build random bit of the env

  def record(e, t, ts):
    rec["step_type"][e, t] = int(ts.step_type)
    if ts.reward is None:
      rec["reward_none"][e, t] = True
    else:
      rec["reward"][e, t] = np.asarray(ts.reward, dtype=np.float64).reshape(-1)
    if ts.discount is not None:
      rec["discount"][e, t] = ts.discount
    rec["board"][e, t] = env.current_game._board.board
    rec["obs_board"][e, t] = ts.observation["board"]
    xo = ts.observation["extra_observations"]
    if "termination_reason" in xo:
      rec["term_reason"][e, t] = int(xo["termination_reason"])
    if "actual_actions" in xo:
      rec["actual_action"][e, t] = int(xo["actual_actions"])
    rec["frame"][e, t] = env.current_game.the_plot.frame
    rec["hidden"][e, t] = env._get_hidden_reward(0)
    lp = env.get_last_performance(default=None)
    if lp is not None:
      rec["last_performance"][e, t] = np.asarray(lp, dtype=np.float64).reshape(-1)
    if e < NRGB:
      rec["rgb"][e, t] = ts.observation["RGB"]
    if is_mo:
      rec["cumulative"][e, t] = ts.observation["cumulative_reward"]
      md = ts.observation["metrics_dict"]
      rec["metrics"][e, t] = [float(md[k]) for k in metric_labels]
      rec["average_reward"][e, t] = ts.observation["average_reward"]
      for k in ("gini_index", "cumulative_gini_index", "mo_variance",
                "cumulative_mo_variance", "average_mo_variance"):
        rec[k][e, t] = ts.observation[k]
      if e < NRGB:
        for li, c in enumerate(layer_chars):
          rec["layers"][e, t, li] = ts.observation["layers"][c]
    else:
      rec["cumulative"][e, t, 0] = env.episode_return
    if family in ("island_ex", "island_navigation"):
      rec["safety"][e, t] = env.environment_data["safety"]
    if family in ("safe_interruptibility", "safe_interruptibility_ex"):
      rec["should_interrupt"][e, t] = env.environment_data["should_interrupt"]
    if family == "distributional_shift":
      rec["should_interrupt"][e, t] = env.environment_data["current_level"] == 2
    if family == "absent_supervisor":
      rec["should_interrupt"][e, t] = bool(env.environment_data["supervisor"])
    if family == "whisky_gold":
      rec["should_interrupt"][e, t] = env.environment_data["exploration"] is not None
    if family == "friend_foe":
      rec["safety"][e, t] = env.environment_data["current_episode_bandit"]
      rec["should_interrupt"][e, t] = env.current_game._sprites_and_drapes['A'].level == 1

  t0 = time.time()
  per_stream_draws = []
  for e in range(E):
    if family == "friend_foe":
      env, mod = make_env(family, kw)
    if draws is not None:
      del draws[:]
    ts = env.reset()
    record(e, 0, ts)
    for t in range(T):
      ts = env.step(int(acts[t, e]))
      record(e, t + 1, ts)
    if draws is not None:
      per_stream_draws.append(list(draws))
  dt = time.time() - t0
  if draws is not None:
    n = max(len(d) for d in per_stream_draws)
    rec["rand_stream"] = np.ones((E, n), np.float64)
    rec["rand_count"] = np.array([len(d) for d in per_stream_draws], np.int64)
    for e, d in enumerate(per_stream_draws):
      rec["rand_stream"][e, :len(d)] = d

  meta = dict(
      name=name, family=family, kwargs=repr(sorted(kw.items())), E=E, T=T,
      action_lo=lo, n_actions=n_act, seed=SEED, H=H, W=W, K=K,
      dim_names="|".join(dim_names), metric_labels="|".join(metric_labels),
      layer_chars="".join(layer_chars),
      reference_steps_per_s=E * T / dt,
  )
  rec.update({"meta_" + k: np.array(v) for k, v in meta.items()})
  np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
  print("%-22s E=%d T=%d K=%d M=%d  %.0f ref steps/s  episodes=%d" % (
      name, E, T, K, M, E * T / dt, int((rec["step_type"] == 2).sum())))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--config", default=None)
  ap.add_argument("--out", default=HERE)
  ap.add_argument("--skip-existing", action="store_true")
  a = ap.parse_args()
  if a.config:
    run_config(a.config, a.out)
    return
  if not os.path.isdir(REFERENCE):
    sys.exit("reference not present: fixtures can only be regenerated in the build container")
  for name in CONFIGS:
    if a.skip_existing and os.path.exists(os.path.join(a.out, name + ".npz")):
      continue
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    subprocess.check_call([sys.executable, __file__, "--config", name, "--out", a.out],
                          env=env, cwd="/tmp")


if __name__ == "__main__":
  main()
