"""Diagnostic: where does a savanna round go?  Times step launches for output / config variants."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec

RICH = dict(amount_agents=2, amount_predators=2, amount_water_tiles=3, amount_gold_deposits=2, amount_silver_deposits=2,
            amount_small_food_patches=2, amount_drink_holes=2, amount_small_drink_holes=1, penalise_oversatiation=True)
def run(name, kw, outs, n=65536, T=200):
  spec = make_spec("aintelope_savanna", **kw)
  e = BatchedEngine(spec, n, outputs=outs)
  e.set_rng_seeds(np.arange(n))
  e.reset()
  acts = e.fill_actions(T + 20, 1)
  for t in range(20): e.step(acts[t])
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for t in range(20, 20 + T): e.step(acts[t])
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / T
  print("%-46s %7.1f us" % (name, dt * 1e6)); e.close()

full = ("board", "reward", "step_type", "term_reason", "safety")
run("rich sust, all outputs", dict(sustainability_challenge=True, **RICH), full)
run("rich sust, no board", dict(sustainability_challenge=True, **RICH), ("reward", "step_type"))
run("rich sust, step_type only", dict(sustainability_challenge=True, **RICH), ("step_type",))
run("rich no-sust, all outputs", dict(**RICH), full)
run("default (1 agent, 2 food), all outputs", dict(), full)
run("2 agents, food only, sust", dict(amount_agents=2, sustainability_challenge=True), full)
run("rich sust, no predators", dict(sustainability_challenge=True, **dict(RICH, amount_predators=0)), full)
