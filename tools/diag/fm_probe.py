"""Where a firemaker round goes: launch time with random actions vs all-NOOP rounds (no fire ever starts: the lane-per-env part,
the dilations and the barriers alone) and the share of envs that burn."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec

dev, n, K = "cuda:0", 16384, 400
spec = make_spec("firemaker_ex_ma", amount_agents=3)
for mode in ("noop", "random", "random-noboard"):
  eng = BatchedEngine(spec, n, device=dev, outputs=("board", "reward", "step_type", "term_reason", "agent_pos") if mode != "random-noboard"
                      else ("reward", "step_type", "term_reason", "agent_pos"))
  eng.set_rng_seeds(np.arange(n))
  eng.reset()
  acts = eng.fill_actions(K, 1)
  if mode == "noop":
    acts.zero_()
  torch.cuda.synchronize()
  for rep in range(3):
    eng.step_n(acts)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for rep in range(3):
    out = eng.step_n(acts)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / (3 * K)
  if mode == "random-noboard":
    print("%-7s %.2f us per round without the board output" % (mode, dt * 1e6), flush=True)
    eng.close()
    continue
  b = out["board"][:n]
  burning = (b == ord('F')).any(dim=1).float().mean().item() if True else 0
  fires = (b == ord('F')).sum(dim=1).float().mean().item()
  print("%-7s %.2f us per round; envs with a visible fire %.2f, fire cells per env %.1f" % (mode, dt * 1e6, burning, fires), flush=True)
  eng.close()
