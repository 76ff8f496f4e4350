"""Diagnostic: sgw_agent_views alone (the board windows of every agent, one launch) at the BASELINE sizes of the two families whose
windows are larger than their boards."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec
from ai_safety_gridworlds_amd import _native as N
L_ = N.lib()
def timed(fn, reps=200, warm=20):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3
for name, n, kw in (("firemaker_ex_ma", 16384, dict(amount_agents=3)), ("aintelope_savanna", 65536, dict(amount_agents=2))):
  sp = make_spec(name, **kw)
  eng = BatchedEngine(sp, n, outputs=("board", "agent_pos", "agent_flags", "step_type"))
  eng.set_rng_seeds(np.arange(n)); eng.reset()
  acts = eng.fill_actions(20, 1)
  for t in range(20): eng.step(acts[t])
  B = eng._bufs
  vb = int(L_.sgw_view_bytes(eng._h))
  vbuf = torch.empty((n, vb), dtype=torch.uint8, device="cuda:0")
  us = timed(lambda: L_.sgw_agent_views(eng._h, B["board"].data_ptr(), B["agent_pos"].data_ptr(), None, ord('#'), vbuf.data_ptr(), eng._stream()))
  print("%s n=%d view_bytes=%d: %.2f us (%.2f TB/s of output)" % (name, n, vb, us, n * vb / us / 1e6), flush=True)
  eng.close()
