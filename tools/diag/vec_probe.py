"""GridworldVectorEnv.step: host time per call vs end-to-end time per step, with the default outputs and with full_info=True
(every observation key of the reference's env.step(): RGB, layers, derived statistics, performance -- one sgw_step_full call),
and the step kernel alone with the same outputs."""
import sys, time, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ai_safety_gridworlds_amd.helpers.gridworld_gym_env import GridworldVectorEnv
n = 65536
for full in (False, True):
  env = GridworldVectorEnv("island_navigation_ex", num_envs=n, full_info=full)
  env.reset()
  acts = env._env.engine.fill_actions(1200, 1)
  buf = torch.empty_like(acts[0])
  for t in range(100): buf.copy_(acts[t]); env.step(buf)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for t in range(100, 1100): buf.copy_(acts[t]); env.step(buf)
  t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
  print("GridworldVectorEnv.step(full_info=%s): host %.2f us per call (incl. the action copy), end to end %.2f us per step"
        % (full, (t1 - t0) / 1000 * 1e6, (t2 - t0) / 1000 * 1e6))
  eng = env._env.engine
  for rep in range(3): eng.step_n(acts[:300])
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for rep in range(4): eng.step_n(acts[:300])
  torch.cuda.synchronize()
  print("  the step kernel alone, same outputs: %.2f us per launch" % ((time.perf_counter() - t0) / 1200 * 1e6))
  if full:
    pr = cProfile.Profile(); pr.enable()
    for t in range(100, 600): buf.copy_(acts[t]); env.step(buf)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(10)
  env.close()
