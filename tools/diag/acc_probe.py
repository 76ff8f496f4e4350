"""Does the episodic-return staging (5.6 KB of LDS per env-wave: 2 instead of 3 workgroups per CU) cost anything at large N?
step_n with and without accumulate, launch-to-launch."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec
spec = make_spec("island_navigation_ex")
for n in (65536, 262144, 1048576):
  eng = BatchedEngine(spec, n, outputs=("board", "reward", "step_type", "term_reason", "safety", "frame"))
  eng.reset()
  K = 200
  acts = eng.fill_actions(K, 1)
  for acc in (True, False):
    for rep in range(3): eng.step_n(acts, accumulate=acc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(5): eng.step_n(acts, accumulate=acc)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (5 * K)
    print("n=%8d accumulate=%-5s %.2f us per launch  frac %.3f" % (n, acc, dt * 1e6, n * 299 / dt / 8e12), flush=True)
  eng.close()
