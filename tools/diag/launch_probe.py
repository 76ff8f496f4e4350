"""Host-side cost of one sgw_step launch: enqueue time of sgw_step_n (host clock, before any sync) and the
launch-to-launch time on the device, for a tiny batch (the kernel itself is ~3 us) and for the bench sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec

dev = "cuda:0"
K = 2000
for name, n in (("island_navigation_ex", 64), ("island_navigation_ex", 65536), ("safe_interruptibility", 64),
                ("safe_interruptibility", 21845), ("island_navigation", 65536)):
  spec = make_spec(name)
  outs = ("board", "reward", "step_type", "term_reason")
  eng = BatchedEngine(spec, n, device=dev, outputs=outs)
  if getattr(spec, "episode_bit", False) or name == "safe_interruptibility":
    eng.set_episode_bits(None, seed=1)
  eng.reset()
  acts = eng.fill_actions(K, 1)
  torch.cuda.synchronize()
  for rep in range(3):
    eng.step_n(acts, accumulate=True)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for rep in range(5):
    eng.step_n(acts, accumulate=True)
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print("%-24s n=%6d  host enqueue %.2f us/launch   end-to-end %.2f us/launch" % (name, n, (t1 - t0) / (5 * K) * 1e6, (t2 - t0) / (5 * K) * 1e6), flush=True)
  eng.close()
