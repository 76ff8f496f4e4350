"""Diagnostic: what each requested output costs a family's step kernel (sgw_step_n graph replay, us per launch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec

name = sys.argv[1] if len(sys.argv) > 1 else "island_navigation_ex_ma"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
kw = eval(sys.argv[3]) if len(sys.argv) > 3 else {}
spec = make_spec(name, **kw)
BASE = ("board", "reward", "step_type", "term_reason")
ZOO = ("board", "reward", "cumulative", "step_type", "term_reason", "discount", "metrics", "agent_pos", "agent_flags", "done")
sets = [("bench outputs", BASE), ("+ views", BASE + ("views",)), ("+ cumulative", BASE + ("cumulative",)), ("+ metrics", BASE + ("metrics",)),
        ("+ discount, agent_pos, agent_flags, done", BASE + ("discount", "agent_pos", "agent_flags", "done")),
        ("zoo outputs", ZOO), ("zoo outputs + views", ZOO + ("views",)), ("zoo + views + dirs", ZOO + ("views", "obs_dir", "act_dir"))]
for label, outs in sets:
  try:
    eng = BatchedEngine(spec, n, outputs=outs)
    eng.set_rng_seeds(np.arange(n))
    eng.reset()
    acts = eng.fill_actions(300, 1)
    for r in range(3): eng.step_n(acts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in range(4): eng.step_n(acts)
    torch.cuda.synchronize()
    print("%-44s %.2f us" % (label, (time.perf_counter() - t0) / 1200 * 1e6), flush=True)
    eng.close()
  except Exception as ex:
    print("%-44s %s" % (label, ex))
