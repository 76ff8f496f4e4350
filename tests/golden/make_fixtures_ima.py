#!/usr/bin/env python3
"""Golden fixtures for island_navigation_ex_ma, produced by RUNNING the reference (build container only).

    python tests/golden/make_fixtures_ima.py [config ...]

Same rules as make_fixtures.py / make_fixtures_ma.py: data only; test-only stand-ins for absl and gymnasium seeding;
the one documented patch (`_last_reward = _default_reward` when still None, pycolab_interface_ma.py:415-417).

Stream protocol (one env per seed, every env follows the same tick schedule so a lockstep batch can replay it):
  slot 0  = state after the CONSTRUCTOR (which resets once internally; with map randomisation this is where the
            first map is drawn).  `rng_seeded` is the generator right after seeding.np_random(seed), i.e. before it.
  slot 1  = the caller's first env.reset() (no step was taken: the episode counter and the cached map stay)
  slot 2+ = one per tick: an explicit env.reset() at the ticks listed in `reset_ticks` (actions[..., 0] == -128),
            otherwise env.step(actions of the agents that are not LAST/DEAD; once every agent is done: the DEAD
            agents only, or all agents when all are LAST -- that round auto-resets and discards the actions).
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SEED = 0x5AFE

CONFIGS = {
    # name: (kwargs, E, T, reset_ticks)
    "ima_L9": (dict(level=9), 24, 160, (60, 61, 130)),
    # thirst_hunger_death=True and the 'U' goal of level 0 cannot be exercised: AgentSafetySpriteMo.terminate_episode
    # refers to `safety_game_ma`, which safety_game_moma.py never imports (NameError at safety_game_moma.py:1636)
    "ima_L9_homeo": (dict(level=9, sustainability_challenge=True, penalise_oversatiation=True,
                                max_iterations=120), 16, 140, (90,)),
    "ima_L9_prop": (dict(level=9, penalise_oversatiation=True, use_satiation_proportional_reward=True,
                               max_iterations=60), 12, 100, ()),
    "ima_L6": (dict(level=6, max_iterations=40), 12, 80, (33,)),
    "ima_L5": (dict(level=5, max_iterations=50), 12, 80, ()),
    "ima_L10_rand3": (dict(level=10, map_randomization_frequency=3, max_iterations=30), 16, 120, (20, 21, 50, 80, 110)),
    "ima_L9_rand3": (dict(level=9, map_randomization_frequency=3, max_iterations=40), 16, 120, (25, 70, 71)),
    "ima_L8_rand1": (dict(level=8, map_randomization_frequency=1, max_iterations=30), 8, 80, (25, 50)),
    "ima_L9_fixeddir": (dict(level=9, action_direction_mode=0, observation_direction_mode=0, max_iterations=50), 8, 80, ()),
    # EnvironmentMa.step with a SUBSET of the agents, the way the AEC wrapper steps (gridworld_zoo_aec_env.py:651-652): two
    # ticks with one agent each (alive agents in turn), then a tick with every alive agent; actions[...] == -1 = not submitted
    "ima_L9_aec": (dict(level=9, max_iterations=50, _aec=True), 16, 150, (70,)),
    # direction mode 2: separate turning actions (Actions 5-8, safety_game_ma.py:608-634, 674-697, 733-758); the action set grows
    # to 0..8 (island_navigation_ex_ma.py:944-945).  The reference only survives turning actions with action mode 2 and
    # observation mode 0 or 2 (mode 1 of either asserts on them)
    "ima_L9_turn": (dict(level=9, action_direction_mode=2, observation_direction_mode=2, max_iterations=60, _n_actions=9), 12, 100, (70,)),
    "ima_L9_turn_fixedobs": (dict(level=9, action_direction_mode=2, observation_direction_mode=0, max_iterations=40, _n_actions=9), 8, 70, ()),
    # amount_agents=1 cannot be constructed in the reference: without map randomisation the '2' of the art stays on the board and the
    # observation distiller has no value for it (RuntimeError, rendering.py:529); with it make_safety_game asserts that a tile type
    # with count 0 has a sprite or drape (safety_game_ma.py:1183).  Nothing to pin.
    # map_width / map_height: make_game's tile_type_counts only lists the agent characters (IM:484-492), so a resized island is a
    # water frame around two agents and gaps
    "ima_resize_7x9": (dict(level=9, map_width=9, map_height=7, map_randomization_frequency=3, max_iterations=30), 10, 90, (20, 21, 60)),
    # more than 64 cells (the 8-word map of the wide kernel instantiation): 120 cells, and the 128-cell ceiling
    "ima_resize_10x12": (dict(level=9, map_width=12, map_height=10, map_randomization_frequency=3, max_iterations=40), 10, 100, (25, 26, 70)),
    "ima_resize_8x16": (dict(level=9, map_width=16, map_height=8, map_randomization_frequency=2, max_iterations=36), 8, 90, (30,)),
    # remove_unused_tile_types_from_layers (safety_game_mo_base.py:1113-1120): the game is built without the drapes of tile types that
    # are not on its map: no such layer, things.get() finds nothing (safety_ stays 3), the drape's update() never runs (metrics None)
    "ima_L10_unused_removed": (dict(level=10, remove_unused_tile_types_from_layers=True, max_iterations=40), 10, 80, (30,)),
    "ima_L1_unused_removed": (dict(level=1, penalise_oversatiation=False, remove_unused_tile_types_from_layers=True, max_iterations=30), 8, 70, ()),
    "ima_L10_rand3_aec": (dict(level=10, map_randomization_frequency=3, penalise_oversatiation=True, max_iterations=36, _aec=True), 12, 140, (50, 51, 100)),
}

A = 2
AGENTS = ['1', '2']


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
  from ai_safety_gridworlds.environments import island_navigation_ex_ma as m

  def words(st):
    mask = (1 << 64) - 1
    return [st['state']['state'] >> 64, st['state']['state'] & mask, st['state']['inc'] >> 64, st['state']['inc'] & mask]

  # every recorded stream is a LATER construction of the class: the very first construction in a process re-seeds its generator
  # after the constructor drew the map (safety_game_moma.py:353-390), which no batch of envs can reproduce
  m.IslandNavigationEnvironmentExMa(seed=1, level=9)
  only = sys.argv[1:] or list(CONFIGS)
  for name in only:
    kw, E, T, reset_ticks = CONFIGS[name]
    kw = dict(kw); aec = kw.pop('_aec', False); n_act = kw.pop('_n_actions', 5)
    AGENTS = ['1', '2'][:kw.get('amount_agents', 2)]      # absent agents keep their (zero / -1) columns in the [.., 2] arrays
    S = T + 2
    acts = np.stack([philox.actions(SEED, np.arange(E), np.arange(T), 0, n_act, agent=a) for a in range(A)], axis=-1)  # [T,E,A]
    acts = np.transpose(acts, (1, 0, 2)).astype(np.int8).copy()     # [E, T, A]
    for t in reset_ticks:
      acts[:, t, :] = -128
    rec = None
    t0 = time.time()
    labels = dims = None
    n_steps = 0
    for e in range(E):
      seed = 2000 + e
      seeded = seeding.np_random(seed)[0].bit_generator.state
      env = m.IslandNavigationEnvironmentExMa(seed=seed, **kw)
      art0 = env.environment_data['ascii_art']
      H, W = len(art0), len(art0[0])
      if rec is None:
        dims = list(env.enabled_agents_reward_dimensions['1']) if hasattr(env, "enabled_agents_reward_dimensions") else None
        labels = list(env.environment_data["metrics_labels"])
        K = len(dims); M = len(labels)
        rec = dict(
            actions=acts, submitted=np.zeros((E, T, A), bool), seeds=np.zeros(E, np.int64), rng_seeded=np.zeros((E, 4), np.uint64),
            step_type=np.zeros((E, S, A), np.uint8), reward=np.zeros((E, S, A, K)), reward_present=np.zeros((E, S, A), bool),
            reward_none=np.zeros((E, S), bool), cumulative=np.zeros((E, S, A, K)), discount=np.full((E, S), np.nan),
            term_reason=np.full((E, S, A), -1, np.int8), frame=np.zeros((E, S), np.int32), board=np.zeros((E, S, H, W), np.uint8),
            metrics=np.zeros((E, S, M)), pos=np.zeros((E, S, A, 2), np.int32), action_direction=np.zeros((E, S, A), np.int8),
            observation_direction=np.zeros((E, S, A), np.int8), safety=np.zeros((E, S, A), np.int32),
            rng=np.zeros((E, S, 4), np.uint64), rng_has_uint32=np.zeros((E, S), np.uint8), rng_uinteger=np.zeros((E, S), np.uint32),
            view=np.zeros((E, S, A, 5, 5), np.uint8), obs_board=np.zeros((E, S, H, W), np.float32),
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
        md = ts.observation["metrics_dict"]
        rec["metrics"][e, t] = [float('nan') if md.get(k) is None else float(md[k]) for k in labels]      # a removed drape never saves its metric
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
          done = [v in (2, 3) for v in stp]
          if all(done):     # the reference raises for a LAST agent submitted next to a DEAD one (PM:213-216)
            sub = [v == 3 for v in stp] if 3 in stp else [True] * len(AGENTS)
          else:
            sub = [not d for d in done]
            if aec and t % 3 != 2:                     # one agent per env.step: the alive agents in turn
              alive = [i for i in range(len(AGENTS)) if not done[i]]
              pick = alive[t % len(alive)]
              sub = [i == pick for i in range(len(AGENTS))]
          if aec:
            for i in range(len(AGENTS)):
              if not sub[i]: acts[e, t, i] = -1
          rec["submitted"][e, t, :len(AGENTS)] = sub
          ts = env.step({ch: {'step': int(acts[e, t, ai])} for ai, ch in enumerate(AGENTS) if sub[ai]})
          n_steps += 1
        record(t + 2, ts)
    dt = time.time() - t0
    meta = dict(name=name, family="island_navigation_ex_ma", aec=int(aec), kwargs=repr(sorted(kw.items())), E=E, T=T, seed=SEED,
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
