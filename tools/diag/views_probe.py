import sys, time
sys.path.insert(0, "/root/repo")
import torch
from ai_safety_gridworlds_amd.helpers.gridworld_zoo_vector_env import GridworldZooVectorEnv
n = 16384
z = GridworldZooVectorEnv("firemaker_ex_ma", num_envs=n, amount_agents=3, seed=0)
z.reset()
eng = z._env.engine
buf = z._view_buf
for i in range(20): eng.agent_views(out=buf)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(300): eng.agent_views(out=buf)
torch.cuda.synchronize(); print("agent_views: %.1f us per call (%d envs, %d view bytes per env)" % ((time.perf_counter() - t0) / 300 * 1e6, n, buf.shape[1]))
acts = torch.zeros((n, 3), dtype=torch.int8, device=z.device)
for i in range(20): z._env.engine.step(acts)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(300): z._env.engine.step(acts)
torch.cuda.synchronize(); print("engine.step (NOOP rounds): %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
