import sys, time
sys.path.insert(0, "/root/repo")
import torch
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec
spec = make_spec("island_navigation_ex")
eng = BatchedEngine(spec, 65536, outputs=("board", "reward", "step_type", "term_reason"))
eng.reset()
acts = eng.fill_actions(2000, 1)
for t in range(200): eng.step(acts[t])
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(2000): eng.step(acts[t])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("BatchedEngine.step from Python: host %.2f us per call, end to end %.2f us per step" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
# GridworldVectorEnv.step has its own probe (tools/diag/vec_probe.py: host time per call, end to end, the kernel alone)
from ai_safety_gridworlds_amd.helpers.gridworld_zoo_vector_env import GridworldZooVectorEnv
for n in (64, 16384):
  z = GridworldZooVectorEnv("firemaker_ex_ma", num_envs=n, amount_agents=3, seed=0)
  z.reset()
  acts3 = {a: torch.zeros(n, dtype=torch.int8, device=z.device) for a in z.possible_agents}
  rnd = torch.randint(0, 5, (300, 3, n), dtype=torch.int8, device=z.device)
  for t in range(50): z.step({a: rnd[t, i] for i, a in enumerate(z.possible_agents)})
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for t in range(50, 300): z.step({a: rnd[t, i] for i, a in enumerate(z.possible_agents)})
  torch.cuda.synchronize(); t2 = time.perf_counter()
  print("GridworldZooVectorEnv(firemaker, %d envs).step: %.1f us per round (%.2e rounds/s)" % (n, (t2 - t0) / 250 * 1e6, n * 250 / (t2 - t0)))
  z.close()
