"""Diagnostic: fused rollout vs step loop on long streams for the register-heavy families (raw state + outputs)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec
CASES = [
  ("island_navigation_ex_ma", dict(map_randomization_frequency=3, max_iterations=30, penalise_oversatiation=True, sustainability_challenge=True)),
  ("island_navigation_ex_ma", dict(level=10, map_randomization_frequency=1, max_iterations=45)),
  ("aintelope_savanna", dict(amount_agents=2, amount_predators=3, amount_water_tiles=3, amount_gold_deposits=2, amount_silver_deposits=2,
                             amount_small_food_patches=2, amount_drink_holes=2, amount_small_drink_holes=1, sustainability_challenge=True,
                             penalise_oversatiation=True, max_iterations=50)),
  ("aintelope_savanna", dict(max_iterations=35)),
  ("island_navigation_ex", dict(level=9, max_iterations=40)),
  ("island_navigation_ex", dict(level=3, DRINK_REWARD={"DRINK_REWARD": 2.0, "FOOD_REWARD": -1.0})),
]
ok = True
for name, kw in CASES:
  spec = make_spec(name, **kw)
  n, T = 5000, 300
  outs = ("board", "reward", "cumulative", "step_type", "frame")
  def fresh():
    e = BatchedEngine(spec, n, outputs=outs)
    if getattr(spec, "needs_rng", False): e.set_rng_seeds(np.arange(n) + 11)
    e.reset(); return e
  a, b, c = fresh(), fresh(), fresh()
  acts = a.fill_actions(T, 77)
  per = {k: [] for k in outs}
  for t in range(T):
    o = a.step(acts[t])
    for k in outs: per[k].append(o[k].clone())
  ro = b.rollout(T, 77, write_every=True)
  same = all(torch.equal(torch.stack(per[k]), ro[k]) for k in outs) and torch.equal(a.get_state()[:, :n], b.get_state()[:, :n])
  c.rollout(T, 77)                       # no per-step outputs: a third instantiation path
  same2 = torch.equal(a.get_state()[:, :n], c.get_state()[:, :n])
  print(name, sorted(kw)[:2], "write_every:", same, "silent:", same2)
  ok &= same and same2
  for e in (a, b, c): e.close()
sys.exit(0 if ok else 1)
