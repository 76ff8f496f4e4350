"""What the outputs cost: a workload's step launches with all of bench.py's outputs, without the board, and with step_type only."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from ai_safety_gridworlds_amd.specs import make_spec

K = 300
for name in (sys.argv[1:] or ["tomato_watering", "side_effects_sokoban", "whisky_gold", "boat_race_ex", "island_navigation_ex_ma", "rocks_diamonds"]):
  wl = B.WORKLOADS[name]
  spec = make_spec(name, **wl["kwargs"])
  n = wl["envs"]
  row = []
  for outs in (wl["outputs"], tuple(o for o in wl["outputs"] if o != "board"), ("step_type",)):
    eng = B.prepare_engine(name, spec, n, 0, torch.device("cuda:0"), outs)
    acts = eng.fill_actions(K, 1)
    for rep in range(3): eng.step_n(acts, accumulate=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(4): eng.step_n(acts, accumulate=True)
    torch.cuda.synchronize()
    row.append((time.perf_counter() - t0) / (4 * K) * 1e6)
    eng.close()
  print("%-26s all outputs %6.2f us   without board %6.2f us   step_type only %6.2f us" % (name, *row), flush=True)
