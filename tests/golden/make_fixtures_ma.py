#!/usr/bin/env python3
"""Golden fixtures for firemaker_ex_ma, produced by RUNNING the reference (build container only).

    python tests/golden/make_fixtures_ma.py

Same rules as make_fixtures.py (data only, test-only stand-ins for absl / gymnasium seeding).
Additionally this env needs the two patches SURVEY.md §8c documents, because at this snapshot the
reference cannot construct it:
  (1) SafetyEnvironmentMoMa.reset leaves `_last_reward` None and FireDrape adds a (zero) reward during
      its_showtime -> `None + ma_reward` raises.  Patch: treat None as the default reward in
      EnvironmentMa._update_for_game_step (what EnvironmentMa.reset itself does, pycolab_interface_ma.py:164).
  (2) FireDrape calls NP_RANDOM.rand() (legacy-gym RandomNumberGenerator); the stand-in Generator
      subclass provides rand == random.
amount_agents=3 (workers '1','2' + supervisor 'S') is the reference's maximum (firemaker_ex_ma.py:113-118).
Each stream: a fresh env seeded `seed`, reset, T rounds of {agent: {"step": a}} actions (Philox, per agent).
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SEED = 0x5AFE

CONFIGS = {
    # name: (kwargs, E, T)
    "firemaker_L0": (dict(amount_agents=3), 24, 160),
    "firemaker_L0_maxit60": (dict(amount_agents=3, max_iterations=60), 16, 120),
    "firemaker_L0_hot": (dict(amount_agents=3, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.05, max_iterations=240), 12, 200),
    # the reference's DEFAULT (firemaker_ex_ma.py:160): one worker + the supervisor; and the lone worker.  Arrays keep the
    # three-column ('1','2','S') layout, absent agents' columns stay zero; metrics sit at their METRICS_LABELS_TEMPLATE rows.
    "firemaker_L0_a2": (dict(amount_agents=2, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.03, max_iterations=120), 16, 200),
    "firemaker_L0_a1": (dict(amount_agents=1, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.03, max_iterations=120), 16, 200),
    # rounds with a SUBSET of the agents in the submitted dict (EnvironmentMa.step plays exactly those, pycolab_interface_ma.py:
    # 173-246; what the Gym wrapper with agent_character does every step): actions < 0 in the recorded array = not submitted
    "firemaker_L0_subset": (dict(amount_agents=3, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.04, max_iterations=90), 16, 200),
    # direction modes (firemaker_ex_ma.py:224-226, 331-336, 472; safety_game_ma.py:515-787): relative moves / windows rot90-ed by the
    # observation direction (mode 1), the turning actions 5-8 (mode 2; action range 0..8), and the mixed combinations
    "firemaker_L0_reldir": (dict(amount_agents=3, observation_direction_mode=1, action_direction_mode=1,
                                 FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.03, max_iterations=120), 16, 160),
    "firemaker_L0_turn": (dict(amount_agents=3, observation_direction_mode=2, action_direction_mode=2,
                               FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.03, max_iterations=120), 16, 160),
    "firemaker_L0_turn_fixedobs": (dict(amount_agents=2, observation_direction_mode=0, action_direction_mode=2,
                                        FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.03, max_iterations=90), 8, 120),
    "firemaker_L0_relact_fixedobs": (dict(amount_agents=3, observation_direction_mode=0, action_direction_mode=1, max_iterations=90), 8, 120),
    "firemaker_L0_fixedact_relobs": (dict(amount_agents=3, observation_direction_mode=1, action_direction_mode=0, max_iterations=90), 8, 120),
    # FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE > 3 (firemaker_ex_ma.py:255, 566-606): sources up to 3 / 4 cells away in each direction
    "firemaker_L0_dist4": (dict(amount_agents=3, FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE=4.0, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.02,
                                max_iterations=120), 12, 160),
    "firemaker_L0_dist4p5": (dict(amount_agents=2, FIRE_SPREAD_EXCLUSIVE_MAX_DISTANCE=4.5, FIRE_SPREAD_PROBABILITY_AT_DISTANCE_ONE=0.015,
                                  max_iterations=90), 8, 140),
    # randomize_agent_actions_order=False cannot be configured through the reference constructor: it passes the
    # flag explicitly AND leaves it in **kwargs (firemaker_ex_ma.py:816-847) -> TypeError "multiple values".
}


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
  from ai_safety_gridworlds.environments.shared import safety_game_moma

  _orig = pycolab_interface_ma.EnvironmentMa._update_for_game_step
  def _patched(self, observations, reward, discount):      # documented patch (1)
    if self._last_reward is None:
      self._last_reward = self._default_reward
    return _orig(self, observations, reward, discount)
  pycolab_interface_ma.EnvironmentMa._update_for_game_step = _patched
  from ai_safety_gridworlds.environments import firemaker_ex_ma as m

  only = sys.argv[1:] or list(CONFIGS)
  for name in only:
    kw, E, T = CONFIGS[name]
    A, K, S = 3, 3, T + 1
    NL, LAYER_CHARS = 2, sorted(" #-12BFSW")
    agents = {1: ['1'], 2: ['1', 'S'], 3: ['1', '2', 'S']}[kw["amount_agents"]]
    SLOT = {'1': 0, '2': 1, 'S': 2}
    TEMPLATE = list(m.METRICS_LABELS_TEMPLATE)
    n_act = 9 if kw.get("action_direction_mode", 0) == 2 else 5          # mode 2 adds the turning actions 5-8 (firemaker_ex_ma.py:808-811)
    acts = np.stack([philox.actions(SEED, np.arange(E), np.arange(T), 0, n_act, agent=a) for a in range(A)], axis=-1)  # [T,E,A]
    if name.endswith("_subset"):        # two ticks out of three: one or two agents only (which ones varies with the tick and the stream)
      acts = acts.astype(np.int8)
      for t in range(T):
        for e in range(E):
          if t % 3 == 1:
            keep = (t // 3 + e) % 3
            for q in range(3):
              if q != keep: acts[t, e, q] = -1
          elif t % 3 == 2:
            acts[t, e, (t // 3 + 2 * e) % 3] = -1
    rec = dict(
        actions=np.transpose(acts, (1, 0, 2)).copy(),          # [E, T, A]
        seeds=np.zeros(E, np.int64), rng_init=np.zeros((E, 4), np.uint64),
        step_type=np.zeros((E, S, A), np.uint8), reward=np.zeros((E, S, A, K)), reward_none=np.zeros((E, S), bool),
        cumulative=np.zeros((E, S, A, K)), discount=np.full((E, S), np.nan), term_reason=np.full((E, S, A), -1, np.int8),
        frame=np.zeros((E, S), np.int32), board=np.zeros((E, S, 17, 17), np.uint8), metrics=np.zeros((E, S, 16)),
        pos=np.zeros((E, S, A, 2), np.int32), action_direction=np.zeros((E, S, A), np.int32),
        observation_direction=np.zeros((E, S, A), np.int32), rng=np.zeros((E, S, 4), np.uint64), rng_has_uint32=np.zeros((E, S), np.uint8),
        rng_uinteger=np.zeros((E, S), np.uint32), view_worker=np.zeros((E, S, 2, 5, 5), np.uint8),
        view_supervisor=np.zeros((E, S, 33, 33), np.uint8), obs_board=np.zeros((E, S, 17, 17), np.float32),
        # observation['layers'] (unoccluded + gap correction) and the per-agent crops of every layer, first NL streams
        layers=np.zeros((NL, S, 9, 17, 17), bool), agent_layers_worker=np.zeros((NL, S, 2, 9, 5, 5), bool),
        agent_layers_supervisor=np.zeros((NL, S, 9, 33, 33), bool))
    t0 = time.time()
    labels = None
    for e in range(E):
      seed = 1000 + e
      env = m.FiremakerExMa(seed=seed, **kw)
      rng = env.environment_data['np_random']
      st = rng.bit_generator.state
      assert st['has_uint32'] == 0
      rec["seeds"][e] = seed
      rec["rng_init"][e] = [st['state']['state'] >> 64, st['state']['state'] & (2**64 - 1),
                            st['state']['inc'] >> 64, st['state']['inc'] & (2**64 - 1)]

      def record(t, ts):
        rng = env.environment_data['np_random']
        st = rng.bit_generator.state
        rec["rng"][e, t] = [st['state']['state'] >> 64, st['state']['state'] & (2**64 - 1),
                            st['state']['inc'] >> 64, st['state']['inc'] & (2**64 - 1)]
        rec["rng_has_uint32"][e, t] = st['has_uint32']; rec["rng_uinteger"][e, t] = st['uinteger']
        for ch in agents:
          ai = SLOT[ch]
          rec["step_type"][e, t, ai] = int(ts.step_type[ch])
          if ts.reward is not None and ts.reward.get(ch) is not None:
            r = np.asarray(ts.reward[ch], dtype=np.float64)
            rec["reward"][e, t, ai, :len(r)] = r
          cr = ts.observation["cumulative_reward"]
          if ch in cr:                # (an agent outside the submitted dict may have no observation entry)
            c = np.asarray(cr[ch], dtype=np.float64)
            rec["cumulative"][e, t, ai, :len(c)] = c
            rec.setdefault("cumulative_present", np.zeros((E, S, A), bool))[e, t, ai] = True
          sp = env.environment_data['agent_sprite'][ch]
          rec["pos"][e, t, ai] = [sp.position.row, sp.position.col]
          rec["action_direction"][e, t, ai] = int(sp.action_direction); rec["observation_direction"][e, t, ai] = int(sp.observation_direction)
          tr = ts.observation["extra_observations"].get("termination_reason")
          if tr is not None:
            v = tr[ch]
            v = v[ch] if isinstance(v, dict) else v        # the reference nests the whole dict per agent
            rec["term_reason"][e, t, ai] = int(v)
        rec["reward_none"][e, t] = ts.reward is None
        if ts.discount is not None:
          rec["discount"][e, t] = ts.discount
        rec["frame"][e, t] = env.current_game.the_plot.frame
        rec["board"][e, t] = env.current_game._board.board
        rec["obs_board"][e, t] = ts.observation["board"]
        md = ts.observation["metrics_dict"]
        for k in labels:              # the template's 16 labels; the dict holds only the present agents' rows
          if k in md:
            rec["metrics"][e, t, TEMPLATE.index(k)] = float(md[k])
        board = env.current_game._board.board
        for ch in agents:
          ai = SLOT[ch]
          sp = env.environment_data['agent_sprite'][ch]
          view = safety_game_moma.get_agent_perspective(sp, board, ord('#'))
          if ai < 2:
            rec["view_worker"][e, t, ai] = view
          else:
            rec["view_supervisor"][e, t] = view
        if e < NL:
          for li, c in enumerate(LAYER_CHARS):
            rec["layers"][e, t, li] = ts.observation["layers"][c]
          per = env.agent_perspectives_with_layers(ts.observation, include_layers=True, board=False, ascii=True)
          for ch in agents:
            ai = SLOT[ch]
            for li, c in enumerate(LAYER_CHARS):
              lay = per[ch]["layers"][c]
              if ai < 2:
                rec["agent_layers_worker"][e, t, ai, li] = lay
              else:
                rec["agent_layers_supervisor"][e, t, li] = lay

      ts = env.reset()
      if labels is None:
        labels = list(env.environment_data["metrics_labels"])
      record(0, ts)
      for t in range(T):
        a = acts[t, e]
        ts = env.step({ch: {'step': int(a[SLOT[ch]])} for ch in agents if a[SLOT[ch]] >= 0})
        record(t + 1, ts)
    dt = time.time() - t0
    meta = dict(name=name, family="firemaker_ex_ma", kwargs=repr(sorted(kw.items())), E=E, T=T, seed=SEED,
                metric_labels="|".join(labels), reference_rounds_per_s=E * T / dt, layer_chars="".join(LAYER_CHARS))
    rec.update({"meta_" + k: np.array(v) for k, v in meta.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("%-24s E=%d T=%d  %.0f ref rounds/s  fires(max cells)=%d  episodes=%d" % (
        name, E, T, E * T / dt, int((rec["board"] == ord('F')).sum(axis=(2, 3)).max()),
        int((rec["step_type"][:, :, 0] == 2).sum())))


if __name__ == "__main__":
  main()
