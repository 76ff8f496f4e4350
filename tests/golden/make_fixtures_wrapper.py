#!/usr/bin/env python3
"""Golden values for the Gym wrapper's info keys and getters, produced by RUNNING the reference (build container only).

    python tests/golden/make_fixtures_wrapper.py

The L5 wrapper itself needs gymnasium (absent); what its info keys and getters RETURN comes from methods of the L4
environment, which run here: calculate_observation_coordinates / get_layers_order / calculate_observation_layers_cube
(gridworld_gym_env.py:357-370, 397-450 -> safety_game_mo.py:422-520), get_reward_unit_space / get_env_seed /
get_env_layout_seed / get_episode_no / get_next_episode_no (gridworld_gym_env.py:677-701 -> safety_game_mo.py:1230-1253)
and the derived statistics of the info dict.  One island_navigation_ex level-9 env (default flags), T steps with an explicit
reset in the middle; written as data to tests/golden/wrapper_island_L9.npz (+ a JSON string for the coordinate dicts).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
  import tempfile
  os.chdir(tempfile.mkdtemp(prefix="sgw_fixtures_"))
  import make_fixtures as MF
  MF._setup_path()
  import numpy as np
  from ai_safety_gridworlds_amd import philox
  T, RESET_AT = 60, 23
  acts = philox.actions(MF.SEED, np.arange(1), np.arange(T), 0, 5)[:, 0]
  env, _ = MF.make_env("island_ex", dict(level=9))
  custom_order = ['A', 'W', 'Z', 'D']                         # 'Z' does not exist: an all-zero plane (cross-environment cubes)
  rec = dict(actions=acts.astype(np.int8), reset_at=np.array(RESET_AT), custom_order=np.array([ord(c) for c in custom_order], np.uint8))
  us = env.get_reward_unit_space()
  rec["unit_space"] = np.stack([np.asarray(us[0], np.float64), np.asarray(us[1], np.float64)])
  rec["dim_names"] = np.array("|".join(env.enabled_reward_dimension_keys))
  rows = dict(episode_no=[], next_episode_no=[], env_seed=[], env_layout_seed=[], cube=[], cube_custom=[], gini=[], cgini=[],
              var=[], cvar=[], avar=[], avg=[])
  coords, orders = [], []

  def record(ts):
    obs = ts.observation
    rows["episode_no"].append(env.get_episode_no()); rows["next_episode_no"].append(env.get_next_episode_no())
    rows["env_seed"].append(env.get_env_seed()); rows["env_layout_seed"].append(env.get_env_layout_seed())
    c = env.calculate_observation_coordinates(obs, occlusion_in_layers=False, ascii=True)
    coords.append({k: [list(map(int, x)) for x in v] for k, v in c.items()})
    order = env.get_layers_order(obs, occlusion_in_layers=False, layers_order=[])
    orders.append("".join(order))
    rows["cube"].append(env.calculate_observation_layers_cube(obs, occlusion_in_layers=False, layers_order=order).astype(np.uint8))
    rows["cube_custom"].append(env.calculate_observation_layers_cube(obs, occlusion_in_layers=False, layers_order=custom_order).astype(np.uint8))
    rows["gini"].append(float(obs["gini_index"])); rows["cgini"].append(float(obs["cumulative_gini_index"]))
    rows["var"].append(float(obs["mo_variance"])); rows["cvar"].append(float(obs["cumulative_mo_variance"]))
    rows["avar"].append(float(obs["average_mo_variance"])); rows["avg"].append(np.asarray(obs["average_reward"], np.float64))

  record(env.reset())
  for t in range(T):
    if t == RESET_AT:
      record(env.reset())
    record(env.step(int(acts[t])))
  for k, v in rows.items():
    rec[k] = np.asarray(v)
  rec["coords_json"] = np.array(json.dumps(coords))
  rec["orders"] = np.array("|".join(orders))
  np.savez_compressed(os.path.join(HERE, "wrapper_island_L9.npz"), **rec)
  print("episode_no", rows["episode_no"][:3], rows["episode_no"][-3:], "next", rows["next_episode_no"][:3], "seed", rows["env_seed"][0],
        rows["env_layout_seed"][0], "order", orders[0], "unit space", rec["unit_space"].tolist())


if __name__ == "__main__":
  main()
