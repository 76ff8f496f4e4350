"""Diagnostic build only: wave cycles per phase of ANY family's step kernel (k_engine's SGW_STAMP marks, -DSGW_PHASE_PROF).
    hipcc ... -DSGW_PHASE_PROF -o tools/diag/libsgw_phaseprof.so ; SGW_LIBRARY=tools/diag/libsgw_phaseprof.so python tools/diag/phase_prof.py [workload ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench as B
from ai_safety_gridworlds_amd import _native as N
from ai_safety_gridworlds_amd.specs import make_spec

NAMES = ["", "prologue: loads issued -> state + tables there, auto-reset", "play (rules)", "outputs staged and copied out",
         "finished-episode returns", "state stores issued"]
K = 200
lib = N.lib()
buf = (C.c_ulonglong * (4096 * 8))()
for name in (sys.argv[1:] or ["island_navigation_ex", "island_navigation_ex_ma", "boat_race_ex", "tomato_watering", "side_effects_sokoban"]):
  wl = B.WORKLOADS[name]
  spec = make_spec(name, **wl["kwargs"])
  n = wl["envs"]
  eng = B.prepare_engine(name, spec, n, 0, torch.device("cuda:0"), wl["outputs"])
  acts = eng.fill_actions(K, 1)
  eng.step_n(acts, accumulate=True); torch.cuda.synchronize()
  lib.sgw_debug_phase_prof(buf, 1)
  t0 = time.perf_counter()
  eng.step_n(acts, accumulate=True); torch.cuda.synchronize()
  us = (time.perf_counter() - t0) / K * 1e6
  assert lib.sgw_debug_phase_prof(buf, 1) == 0
  arr = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.float64)
  waves = int((arr.sum(1) > 0).sum())
  tot = arr.sum()
  print("%s: %.2f us per launch (profiled build), %d waves marked, %.0f cycles per wave and launch" % (name, us, waves, tot / max(waves, 1) / K))
  for k in range(1, 6):
    print("  %-62s %8.0f  %5.1f %%" % (NAMES[k], arr[:, k].sum() / max(waves, 1) / K, 100.0 * arr[:, k].sum() / tot))
  eng.close()
