"""GridworldZooVectorEnv.step per round: host time per call, end-to-end time, and the step kernel alone with the same outputs
(sgw_step_n graph replay) -- how far the Python surface is from the kernel under it."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from ai_safety_gridworlds_amd.helpers.gridworld_zoo_vector_env import GridworldZooVectorEnv
for name, n, kw, layers in (("island_navigation_ex_ma", 65536, {}, False), ("island_navigation_ex_ma", 65536, {}, True),
                            ("aintelope_savanna", 65536, dict(amount_agents=2), False), ("firemaker_ex_ma", 16384, dict(amount_agents=3), False),
                            ("firemaker_ex_ma", 16384, dict(amount_agents=3), True)):
  z = GridworldZooVectorEnv(name, num_envs=n, seed=0, layers_in_observation=layers, **kw)
  z.reset()
  A = z.spec_.A
  rnd = torch.randint(0, 5, (320, n, A), dtype=torch.int8, device=z.device)
  for t in range(20): z.step(rnd[t])
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for t in range(20, 320): z.step(rnd[t])
  t1 = time.perf_counter(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
  eng = z._env.engine
  for rep in range(3): eng.step_n(rnd[:300])
  torch.cuda.synchronize(); k0 = time.perf_counter()
  for rep in range(3): eng.step_n(rnd[:300])
  torch.cuda.synchronize(); kdt = (time.perf_counter() - k0) / 900
  print("%-26s %6d envs layers=%-5s %.1f us per round (%.2e rounds/s), host %.1f us per call, step kernel alone %.1f us" % (
      name, n, layers, dt * 1e6, n / dt, (t1 - t0) / 300 * 1e6, kdt * 1e6), flush=True)
  z.close()
