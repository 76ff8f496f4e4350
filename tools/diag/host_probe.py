"""Diagnostic: where the host time of one Python step goes (island_navigation_ex, 65 536 envs).  Each line: host microseconds per
call (the loop's wall time before the final synchronize) and end-to-end microseconds per step.

    python tools/diag/host_probe.py [n_envs] [calls]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ai_safety_gridworlds_amd import _native as N
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec
from ai_safety_gridworlds_amd.helpers.gridworld_gym_env import GridworldVectorEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
OUTS = ("board", "obs_board", "reward", "cumulative", "step_type", "term_reason", "hidden")


def timed(label, fn, warm=200):
  for _ in range(warm): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(calls): fn()
  t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
  print("%-62s host %6.2f us   end to end %6.2f us" % (label, (t1 - t0) / calls * 1e6, (t2 - t0) / calls * 1e6), flush=True)


eng = BatchedEngine(make_spec("island_navigation_ex"), n, outputs=OUTS)
eng.reset()
acts = eng.fill_actions(8, 1)
a0 = acts[0].contiguous()
ptr, outp, h, lib = a0.data_ptr(), C.byref(eng._out), eng._h, eng._lib
stream = eng._stream()
timed("lib.sgw_step (bare ctypes call, fixed stream)", lambda: lib.sgw_step(h, ptr, outp, stream))
timed("lib.sgw_step + eng._stream()", lambda: lib.sgw_step(h, ptr, outp, eng._stream()))
timed("eng.step_ptr", lambda: eng.step_ptr(ptr))
timed("eng.step(tensor)", lambda: eng.step(a0))
timed("eng.step_full(tensor) no extras", lambda: eng.step_full(a0))
timed("eng.step_full(tensor, performance=True)", lambda: eng.step_full(a0, performance=True))
timed("eng.step(tensor) + (step_type == LAST)", lambda: eng.step(a0)["step_type"] == 2)
buf = torch.empty_like(a0)
timed("buf.copy_(actions) alone", lambda: buf.copy_(a0))
eng.close()
for full in (False, True):
  env = GridworldVectorEnv("island_navigation_ex", num_envs=n, full_info=full)
  env.reset()
  timed("GridworldVectorEnv.step(full_info=%s)" % full, lambda: env.step(a0))
  env.close()
