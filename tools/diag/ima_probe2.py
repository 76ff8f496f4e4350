import sys
import numpy as np, torch
sys.path.insert(0, ".")
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec
spec = make_spec("island_navigation_ex_ma", map_randomization_frequency=3, max_iterations=30)
n = 3000
def dec(s):
  w0, w1 = int(s[0]), int(s[1])
  return dict(map_cached=(w0 >> 45) & 1, episode_no=(w1 >> 32) & 0xffff, map_episode=(w1 >> 48) & 0xffff, st=(w0 >> 32) & 15)
for T in (1, 2, 3):
  for wev in (False, True):
    e = BatchedEngine(spec, n, outputs=("board", "reward", "step_type")); e.set_rng_seeds(np.arange(n) + 3); e.reset()
    before = dec(e.get_state()[:, 7].cpu().numpy().view(np.uint64))
    e.rollout(T, 99, write_every=wev)
    after = dec(e.get_state()[:, 7].cpu().numpy().view(np.uint64))
    allst = e.get_state()[:, :n].cpu().numpy().view(np.uint64)
    ep = (allst[1] >> np.uint64(32)) & np.uint64(0xffff)
    print("T", T, "write_every", wev, before, "->", after, "| envs with episode_no != 1:", int((ep != 1).sum()))
    e.close()
