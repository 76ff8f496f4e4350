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
  rnd = torch.randint(0, 5, (120, n, A), dtype=torch.int8, device=z.device)
  for t in range(20): z.step(rnd[t])
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for t in range(20, 120): z.step(rnd[t])
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
  print("%-26s %6d envs layers=%-5s %.1f us per round (%.2e rounds/s)" % (name, n, layers, dt * 1e6, n / dt), flush=True)
  z.close()
