"""Diagnostic build only: wave cycles per phase of an aintelope_savanna round (csrc/sgw_savanna.hpp SAV_T marks, -DSGW_SAV_PROF).
    hipcc ... -DSGW_SAV_PROF -o tools/diag/libsgw_savprof.so ; SGW_LIBRARY=tools/diag/libsgw_savprof.so python tools/diag/sav_prof.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench as B
from ai_safety_gridworlds_amd import _native as N
from ai_safety_gridworlds_amd.specs import make_spec

NAMES = ["between the plays (action-order shuffle)", "agent move + update_reward", "water + predators", "reward bookkeeping",
         "prologue: state load issued, tables staged, auto-reset of finished envs", "outputs staged and copied out", "state stores issued", "-",
         "drapes: availability (regrowth pow)", "drapes: sampling (tiles taken away / put)"]
n, K = 65536, 200
wl = B.WORKLOADS["aintelope_savanna"]
spec = make_spec("aintelope_savanna", **wl["kwargs"])
eng = B.prepare_engine("aintelope_savanna", spec, n, 0, torch.device("cuda:0"), wl["outputs"])
acts = eng.fill_actions(K, 1)
lib = N.lib()
buf = (C.c_ulonglong * (4096 * 16))()
eng.step_n(acts); torch.cuda.synchronize()
lib.sgw_debug_sav_prof(buf, 1)
t0 = time.perf_counter()
eng.step_n(acts); torch.cuda.synchronize()
print("%.2f us per round (profiled build)" % ((time.perf_counter() - t0) / K * 1e6))
assert lib.sgw_debug_sav_prof(buf, 1) == 0
waves = n // 64
arr = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16)[:waves].astype(np.float64)
tot = arr.sum()
print("cycles per wave per round, state load to state store: %.0f" % (tot / waves / K))
for k in (4, 0, 1, 2, 3, 8, 9, 5, 6):
  print("  %-60s %8.0f  %5.1f %%" % (NAMES[k], arr[:, k].sum() / waves / K, 100.0 * arr[:, k].sum() / tot))
