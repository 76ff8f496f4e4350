"""Diagnostic build only: wave cycles per phase of a firemaker round (csrc/sgw_firemaker.hpp FM_T marks, -DSGW_FM_PROF).
    hipcc ... -DSGW_FM_PROF -o tools/diag/libsgw_fmprof.so ; SGW_LIBRARY=tools/diag/libsgw_fmprof.so python tools/diag/fm_prof.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ai_safety_gridworlds_amd import _native as N
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec

NAMES = ["outside fire_update (3 plays, shuffle, outputs)", "dilation / candidates / work mask", "ticket", "broadcast + PCG jump-ahead + 128 draws",
         "scratch fill + candidate list", "neighbourhoods + chains", "draw ranks + ring + atomicOr", "new fire words back",
         "continue passes + final state", "publish", "barrier wait", "tail"]
n, K = 16384, 300
spec = make_spec("firemaker_ex_ma", amount_agents=3)
eng = BatchedEngine(spec, n, device="cuda:0", outputs=("board", "reward", "step_type", "term_reason", "agent_pos"))
eng.set_rng_seeds(np.arange(n)); eng.reset()
acts = eng.fill_actions(K, 1)
if len(sys.argv) > 1 and sys.argv[1] == "noop":
  acts.zero_()
lib = N.lib()
buf = (C.c_ulonglong * (4096 * 16))()
eng.step_n(acts); torch.cuda.synchronize()
lib.sgw_debug_fm_prof(buf, 1)
import time
t0 = time.perf_counter()
eng.step_n(acts); torch.cuda.synchronize()
print("%.2f us per round (profiled build)" % ((time.perf_counter() - t0) / K * 1e6))
assert lib.sgw_debug_fm_prof(buf, 1) == 0
waves = n // 64 * 8
arr = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16)[:waves].astype(np.float64)
served = (np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16)[:waves, 11] >> np.uint64(32)).astype(np.float64)
arr[:, 11] -= served * 4294967296.0
print("env-spreads served per wave per round: %.2f (per workgroup and play: %.1f of 64 envs)" % (served.sum() / waves / K, served.sum() / (waves / 8) / K / 3))
cnt = arr[:, 12:16].sum(0); arr = arr[:, :12]
print("env-spreads by size: <= 32 candidates and <= 32 fire cells %.1f %%, 33-64 candidates %.1f %%, > 64 %.1f %%; mean candidates %.1f" % (
    100 * cnt[0] / served.sum(), 100 * cnt[1] / served.sum(), 100 * cnt[2] / served.sum(), cnt[3] / served.sum()))
tot = arr.sum()
print("cycles per wave per round: %.0f (slowest wave %.0f, fastest %.0f)" % (tot / waves / K, arr.sum(1).max() / K, arr.sum(1).min() / K))
for k in range(12):
  print("  %-52s %8.0f  %5.1f %%" % (NAMES[k], arr[:, k].sum() / waves / K, 100.0 * arr[:, k].sum() / tot))
wg = arr.reshape(-1, 8, 12)
work = wg[:, :, 2:10].sum(2)            # spread work per wave
print("spread work per wave within a workgroup: mean %.0f, mean of max %.0f (per round)" % (work.mean() / K, work.max(1).mean() / K))
