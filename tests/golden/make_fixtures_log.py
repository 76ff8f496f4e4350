#!/usr/bin/env python3
"""Golden CSV step log, produced by RUNNING the reference with logging switched on (build container only).

    python tests/golden/make_fixtures_log.py

One island_navigation_ex level-9 env (default flags), `log_columns` as in the reference's own main()
(island_navigation_ex.py:815-834), reset + T Philox actions with auto-resets and one explicit reset; the file the
reference wrote is stored as data (tests/golden/island_L9_steplog.csv) next to the action stream (…_actions.npy).
Same stand-ins as make_fixtures.py.
"""
import glob
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SEED, T, RESET_AT = 0x5AFE, 260, 150


def main():
  # scratch cwd: the reference's step logger writes ./logs/*.csv into the current directory
  import tempfile
  os.chdir(tempfile.mkdtemp(prefix="sgw_fixtures_"))
  sys.dont_write_bytecode = True
  sys.path.insert(0, "/root/reference"); sys.path.insert(0, os.path.join(HERE, "standins")); sys.path.insert(0, REPO)
  import numpy as np
  from ai_safety_gridworlds_amd import philox
  from ai_safety_gridworlds.environments import island_navigation_ex as m
  from ai_safety_gridworlds.environments.shared import safety_game_mo as mo
  cols = [mo.LOG_TRIAL, mo.LOG_EPISODE, mo.LOG_ITERATION, mo.LOG_REWARD, mo.LOG_SCALAR_REWARD, mo.LOG_CUMULATIVE_REWARD,
          mo.LOG_AVERAGE_REWARD, mo.LOG_SCALAR_CUMULATIVE_REWARD, mo.LOG_SCALAR_AVERAGE_REWARD, mo.LOG_GINI_INDEX,
          mo.LOG_CUMULATIVE_GINI_INDEX, mo.LOG_MO_VARIANCE, mo.LOG_CUMULATIVE_MO_VARIANCE, mo.LOG_AVERAGE_MO_VARIANCE, mo.LOG_METRICS]
  acts = philox.actions(SEED, np.arange(1) + 77, np.arange(T), 0, 5)[:, 0]
  tmp = tempfile.mkdtemp()
  env = m.IslandNavigationEnvironmentEx(level=9, max_iterations=100, noops=True, sustainability_challenge=True,
                                        thirst_hunger_death=False, penalise_oversatiation=True,
                                        use_satiation_proportional_reward=False, log_columns=cols, log_dir=tmp,
                                        log_arguments_to_separate_file=False)
  env.reset()
  env.reset()      # the reference opens its log file in a reset() that finds the env in its FIRST state (safety_game_mo.py:576-650)
  for t in range(T):
    if t == RESET_AT:
      env.reset()
    env.step(int(acts[t]))
  f = getattr(env.__class__, "log_file_handle", None)
  if f:
    f.flush(); f.close()
  files = [p for p in glob.glob(os.path.join(tmp, "*.csv"))]
  assert len(files) == 1, files
  shutil.copy(files[0], os.path.join(HERE, "island_L9_steplog.csv"))
  np.save(os.path.join(HERE, "island_L9_steplog_actions.npy"), acts.astype(np.int8))
  n = sum(1 for _ in open(files[0]))
  print("wrote island_L9_steplog.csv: %d lines" % n)
  shutil.rmtree(tmp)


if __name__ == "__main__":
  main()
