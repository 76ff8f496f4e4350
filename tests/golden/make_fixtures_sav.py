#!/usr/bin/env python3
"""Golden fixtures for aintelope_savanna, produced by RUNNING the reference (build container only).

    python tests/golden/make_fixtures_sav.py [config ...]

Same rules and stream protocol as make_fixtures_ima.py: data only; test-only stand-ins for absl and gymnasium seeding;
the one documented patch (`_last_reward = _default_reward` when still None, pycolab_interface_ma.py:415-417).
Every agent is submitted every tick (no agent of this env can finish on its own: thirst_hunger_death and the 'U' goal
raise NameError in the reference, safety_game_moma.py:1636); the tick after all agents are LAST auto-resets.
`metrics` is column 1 of the reference's metrics matrix with None recorded as NaN; `layers` are the drape curtains
(W P D F d f G S and the dummy '1' drape of an absent second agent), which overlap freely in this env.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SEED = 0x5AFE

RICH = dict(amount_predators=2, amount_water_tiles=3, amount_gold_deposits=2, amount_silver_deposits=2,
            amount_small_food_patches=2, amount_drink_holes=2, amount_small_drink_holes=1)
R2 = [2, 2, 2, 2]

CONFIGS = {
    # name: (kwargs, E, T, reset_ticks)
    "sav_default": (dict(max_iterations=40), 10, 100, (30, 31, 75)),
    "sav_rich2_sust": (dict(amount_agents=2, sustainability_challenge=True, penalise_oversatiation=True,
                            max_iterations=120, observation_radius=R2, **RICH), 16, 160, (100, 101, 140)),
    "sav_rich2": (dict(amount_agents=2, max_iterations=80, map_randomization_frequency=1, observation_radius=R2, **RICH),
                  12, 100, (50,)),
    "sav_rich1_prop": (dict(amount_agents=1, sustainability_challenge=True, penalise_oversatiation=True,
                            use_satiation_proportional_reward=True, max_iterations=70, observation_radius=R2, **RICH),
                       12, 120, (40, 90)),
    "sav_L13_fixed": (dict(level=13, amount_agents=2, map_randomization_frequency=0, amount_food_patches=1,
                           amount_drink_holes=1, amount_small_food_patches=1, amount_small_drink_holes=1,
                           sustainability_challenge=True, penalise_oversatiation=True, max_iterations=90,
                           observation_radius=R2), 12, 120, (60,)),
    "sav_L0_fixed2": (dict(level=0, amount_agents=2, map_randomization_frequency=0, amount_food_patches=2,
                           amount_drink_holes=5, amount_small_food_patches=2, amount_small_drink_holes=2,
                           amount_gold_deposits=5, amount_silver_deposits=5, amount_water_tiles=5, amount_predators=5,
                           max_iterations=100, observation_radius=R2), 12, 130, (55,)),
    "sav_L16_fixeddir": (dict(level=16, amount_agents=2, amount_food_patches=1, sustainability_challenge=True,
                              penalise_oversatiation=True, use_satiation_proportional_reward=True,
                              action_direction_mode=0, observation_direction_mode=0, max_iterations=60,
                              observation_radius=R2), 8, 80, ()),
    # experiments/aintelope presets (flag overrides only); the tests take the flag values from experiment_presets.json
    "sav_exp_predators_gold_silver": (dict(experiment="food_drink_homeostasis_predators_gold_silver", max_iterations=60), 12, 120, (45,)),
    "sav_exp_demo": (dict(experiment="savanna_demo", max_iterations=80), 12, 110, (50, 51)),
    "sav_exp_sharing": (dict(experiment="food_sharing", max_iterations=60), 8, 90, ()),
    # EnvironmentMa.step with ONE agent per call on two ticks out of three (the AEC wrapper's way); actions == -1 = not submitted
    "sav_rich2_sust_aec": (dict(amount_agents=2, sustainability_challenge=True, penalise_oversatiation=True,
                                max_iterations=90, observation_radius=R2, _aec=True, **RICH), 12, 150, (80, 81)),
    # map_width / map_height (MA:1113-1170): a frame of walls around an interior filled with the tile counts, then shuffled
    "sav_resize_9x11": (dict(map_width=11, map_height=9, amount_agents=2, amount_predators=1, amount_water_tiles=2,
                             amount_gold_deposits=1, amount_drink_holes=1, amount_small_food_patches=1,
                             sustainability_challenge=True, penalise_oversatiation=True, max_iterations=50,
                             observation_radius=R2), 12, 110, (30, 31, 80)),
    "sav_resize_13x14": (dict(map_width=14, amount_agents=1, amount_predators=3, amount_silver_deposits=2, max_iterations=40,
                              map_randomization_frequency=2, observation_radius=R2), 8, 90, (45,)),
    # direction mode 2: separate turning actions (Actions 5-8), action set 0..8 (aintelope_savanna.py:1646-1647)
    "sav_rich2_turn": (dict(amount_agents=2, action_direction_mode=2, observation_direction_mode=2, sustainability_challenge=True,
                            penalise_oversatiation=True, max_iterations=60, observation_radius=R2, _n_actions=9, **RICH), 10, 100, (70,)),
    "sav_turn_fixedobs": (dict(amount_agents=1, action_direction_mode=2, observation_direction_mode=0, max_iterations=40,
                               observation_radius=R2, _n_actions=9, **RICH), 8, 70, ()),
    # remove_unused_tile_types_from_layers (safety_game_ma.py:1256-1262): drapes of tile types that are not on the map are dropped
    # -- their layers disappear, `things.get(...)` finds nothing (safety_ / safety2_ stay at their initial 3) and their update()
    # (availability metrics) never runs
    "sav_unused_removed": (dict(amount_agents=2, remove_unused_tile_types_from_layers=True, amount_water_tiles=0, amount_predators=0,
                                amount_gold_deposits=0, amount_silver_deposits=2, amount_food_patches=2, amount_drink_holes=0,
                                amount_small_food_patches=0, amount_small_drink_holes=1, sustainability_challenge=True,
                                penalise_oversatiation=True, max_iterations=50, observation_radius=R2), 10, 90, (40,)),
    "sav_L3_tiny": (dict(level=3, amount_food_patches=1, sustainability_challenge=True, penalise_oversatiation=True,
                         max_iterations=30, observation_radius=R2), 8, 70, (20,)),
    "sav_L14_metric_only": (dict(level=14, amount_agents=2, amount_food_patches=1, amount_drink_holes=1,
                                 amount_small_food_patches=1, amount_small_drink_holes=1, sustainability_challenge=True,
                                 use_food_availability_metric_instead_of_spawning_tiles=True,
                                 use_drink_availability_metric_instead_of_spawning_tiles=True, max_iterations=50,
                                 map_randomization_frequency=2, observation_radius=R2), 8, 80, (30,)),
}

LAYER_CHRS = "WPDFdfGS1"


def main():
  # scratch cwd: the reference's step logger writes ./logs/*.csv into the current directory
  import tempfile
  os.chdir(tempfile.mkdtemp(prefix="sgw_fixtures_"))
  sys.dont_write_bytecode = True
  sys.path.insert(0, "/root/reference")
  sys.path.insert(0, os.path.join(HERE, "standins"))
  sys.path.insert(0, REPO)
  import numpy as np
  from ai_safety_gridworlds_amd import philox
  from ai_safety_gridworlds.environments.shared.rl import pycolab_interface_ma
  from gymnasium.utils import seeding

  _orig = pycolab_interface_ma.EnvironmentMa._update_for_game_step
  def _patched(self, observations, reward, discount):      # the documented patch
    if self._last_reward is None:
      self._last_reward = self._default_reward
    return _orig(self, observations, reward, discount)
  pycolab_interface_ma.EnvironmentMa._update_for_game_step = _patched
  from ai_safety_gridworlds.environments.aintelope import aintelope_savanna as m

  def words(st):
    mask = (1 << 64) - 1
    return [st['state']['state'] >> 64, st['state']['state'] & mask, st['state']['inc'] >> 64, st['state']['inc'] & mask]

  only = sys.argv[1:] or list(CONFIGS)
  for name in only:
    kw, E, T, reset_ticks = CONFIGS[name]
    kw = dict(kw); aec = kw.pop('_aec', False); n_act = kw.pop('_n_actions', 5)
    S = T + 2
    ctor = m.AIntelopeSavannaEnvironmentMa
    ctor_kw = dict(kw)
    eff = dict(kw)
    if "experiment" in kw:
      import importlib
      x = importlib.import_module("ai_safety_gridworlds.experiments.aintelope." + ctor_kw.pop("experiment"))
      ctor = x.AIntelopeSavannaEnvironmentMaExperiment
      f = x.init_experiment_flags()
      eff = dict(amount_agents=f.amount_agents, observation_radius=f.observation_radius); eff.update(kw)
    A = eff.get('amount_agents', 1)
    AGENTS = ['0', '1'][:A]
    VS = 2 * eff.get('observation_radius', [10])[0] + 1
    acts = np.stack([philox.actions(SEED, np.arange(E), np.arange(T), 0, n_act, agent=a) for a in range(2)], axis=-1)  # [T,E,2]
    acts = np.transpose(acts, (1, 0, 2)).astype(np.int8).copy()     # [E, T, A]
    for t in reset_ticks:
      acts[:, t, :] = -128
    rec = None
    t0 = time.time()
    labels = dims = None
    n_steps = 0
    # The very first construction of the class in a process re-seeds environment_data[NP_RANDOM] AFTER the constructor's
    # reset drew the map (class attribute env_layout_seed still -1, safety_game_moma.py:353-390); every later one does
    # not.  A batch has no "first": the streams are recorded as later constructions.
    ctor(seed=1, **ctor_kw)
    for e in range(E):
      seed = 2000 + e
      seeded = seeding.np_random(seed)[0].bit_generator.state
      env = ctor(seed=seed, **ctor_kw)
      art0 = env.environment_data['ascii_art']
      H, W = len(art0), len(art0[0])
      if rec is None:
        dims = list(env.enabled_agents_reward_dimensions['0']) if hasattr(env, "enabled_agents_reward_dimensions") else None
        labels = list(env.environment_data["metrics_labels"])
        K = len(dims); M = len(labels)
        rec = dict(
            actions=acts, submitted=np.zeros((E, T, 2), bool), seeds=np.zeros(E, np.int64), rng_seeded=np.zeros((E, 4), np.uint64),
            step_type=np.zeros((E, S, A), np.uint8), reward=np.zeros((E, S, A, K)), reward_present=np.zeros((E, S, A), bool),
            reward_none=np.zeros((E, S), bool), cumulative=np.zeros((E, S, A, K)), discount=np.full((E, S), np.nan),
            term_reason=np.full((E, S, A), -1, np.int8), frame=np.zeros((E, S), np.int32), board=np.zeros((E, S, H, W), np.uint8),
            metrics=np.zeros((E, S, M)), pos=np.zeros((E, S, A, 2), np.int32), action_direction=np.zeros((E, S, A), np.int8),
            observation_direction=np.zeros((E, S, A), np.int8), safety=np.zeros((E, S, A), np.int32),
            safety2=np.zeros((E, S, A), np.int32), layers=np.zeros((E, S, len(LAYER_CHRS), H, W), np.uint8),
            rng=np.zeros((E, S, 4), np.uint64), rng_has_uint32=np.zeros((E, S), np.uint8), rng_uinteger=np.zeros((E, S), np.uint32),
            view=np.zeros((E, S, A, VS, VS), np.uint8), obs_board=np.zeros((E, S, H, W), np.float32),
            art0=np.zeros((E, H, W), np.uint8))
      rec["seeds"][e] = seed
      rec["rng_seeded"][e] = words(seeded)
      assert seeded['has_uint32'] == 0

      def record_state(t):
        st = env.environment_data['np_random'].bit_generator.state
        rec["rng"][e, t] = words(st)
        rec["rng_has_uint32"][e, t] = st['has_uint32']; rec["rng_uinteger"][e, t] = st['uinteger']
        rec["frame"][e, t] = env.current_game.the_plot.frame
        rec["board"][e, t] = env.current_game._board.board
        for ai, ch in enumerate(AGENTS):
          sp = env.environment_data['agent_sprite'][ch]
          rec["pos"][e, t, ai] = [sp.position.row, sp.position.col]
          rec["action_direction"][e, t, ai] = int(sp.action_direction)
          rec["observation_direction"][e, t, ai] = int(sp.observation_direction)
          rec["safety"][e, t, ai] = int(env.environment_data['safety_' + ch])
          rec["safety2"][e, t, ai] = int(env.environment_data['safety2_' + ch])
        things = env.current_game._sprites_and_drapes
        for li, lc in enumerate(LAYER_CHRS):
          if lc in things and hasattr(things[lc], 'curtain'):
            rec["layers"][e, t, li] = things[lc].curtain

      def record(t, ts):
        record_state(t)
        for ai, ch in enumerate(AGENTS):
          rec["step_type"][e, t, ai] = int(ts.step_type[ch])
          full = ts.observation["reward_dict"][ch]
          rec["reward"][e, t, ai] = [float(full[d]) for d in dims]
          if ts.reward is not None and ts.reward.get(ch) is not None:
            rec["reward_present"][e, t, ai] = True
            assert np.array_equal(np.asarray(ts.reward[ch], dtype=np.float64), rec["reward"][e, t, ai])
          c = np.asarray(ts.observation["cumulative_reward"][ch], dtype=np.float64)
          rec["cumulative"][e, t, ai] = c
          tr = ts.observation["extra_observations"].get("termination_reason")
          if tr is not None:
            v = tr[ch]
            v = v[ch] if isinstance(v, dict) else v        # the reference nests the whole dict per agent
            rec["term_reason"][e, t, ai] = int(v)
        rec["reward_none"][e, t] = ts.reward is None
        if ts.discount is not None:
          rec["discount"][e, t] = ts.discount
        rec["obs_board"][e, t] = ts.observation["board"]
        mm = env.environment_data['metrics_matrix']
        rec["metrics"][e, t] = [np.nan if v is None else float(v) for v in mm[:, 1]]
        per = env.agent_perspectives(env.current_game._board.board)
        for ai, ch in enumerate(AGENTS):
          rec["view"][e, t, ai] = per[ch]

      # slot 0: the constructor dropped its game after computing the observation spec; what remains is the generator
      # state and the map it drew (environment_data['ascii_art'])
      st0 = env.environment_data['np_random'].bit_generator.state
      rec["rng"][e, 0] = words(st0); rec["rng_has_uint32"][e, 0] = st0['has_uint32']; rec["rng_uinteger"][e, 0] = st0['uinteger']
      rec["art0"][e] = np.array([[ord(c) for c in row] for row in art0], np.uint8)
      ts = env.reset()
      record(1, ts)
      for t in range(T):
        if acts[e, t, 0] == -128:
          ts = env.reset()
        else:
          stp = [int(ts.step_type[ch]) for ch in AGENTS]
          assert 3 not in stp and len(set(stp)) == 1, stp
          sub = [True] * A
          if aec and t % 3 != 2 and stp[0] != 2:        # one agent per env.step, in turn (a finished episode resets with everybody)
            sub = [i == t % A for i in range(A)]
            for i in range(A):
              if not sub[i]: acts[e, t, i] = -1
          rec["submitted"][e, t, :A] = sub
          ts = env.step({ch: {'step': int(acts[e, t, ai])} for ai, ch in enumerate(AGENTS) if sub[ai]})
          n_steps += 1
        record(t + 2, ts)
    dt = time.time() - t0
    meta = dict(name=name, family="aintelope_savanna", kwargs=repr(sorted(kw.items())), E=E, T=T, seed=SEED,
                layer_keys="".join(sorted(ts.observation["layers"].keys())),
                metric_labels="|".join(labels), dim_names="|".join(dims), reference_rounds_per_s=n_steps / dt,
                reset_ticks=np.array(reset_ticks, np.int32))
    rec.update({"meta_" + k: np.array(v) for k, v in meta.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    st = rec["step_type"]
    print("%-24s E=%d T=%d  %.0f ref rounds/s  K=%d M=%d  LAST=%d DEAD=%d  distinct maps=%d" % (
        name, E, T, n_steps / dt, len(dims), len(labels), int((st == 2).sum()), int((st == 3).sum()),
        len({rec["board"][e, t].tobytes() for e in range(E) for t in range(S) if rec["frame"][e, t] == 0})))


if __name__ == "__main__":
  main()
