"""Diagnostic: open-ended randomised configuration sweep, HIP engine vs the C oracle (the cases live in tests/fuzz_cases.py;
tests/test_fuzz_gpu.py runs a bounded fixed-seed slice of them in the GPU tier).  Exit code 1 on a mismatch.

    python tools/diag/fuzz_parity.py [seed] [seconds] [ima|sav|fm|''] [strict]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import fuzz_cases as F

rnd = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
only = sys.argv[3] if len(sys.argv) > 3 else ""
strict = len(sys.argv) > 4 and sys.argv[4] == "strict"
cases = {"ima": [lambda: F.ma_case(rnd, "ima", strict=strict)], "sav": [lambda: F.ma_case(rnd, "sav", strict=strict)],
         "fm": [lambda: F.firemaker_case(rnd)]}.get(
    only, [lambda: F.island_case(rnd), lambda: F.ma_case(rnd, "ima", strict=strict), lambda: F.ma_case(rnd, "sav", strict=strict),
           lambda: F.firemaker_case(rnd)])
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < (float(sys.argv[2]) if len(sys.argv) > 2 else 60):
  res = cases[n % len(cases)]()
  n += 1
  if res is None: continue
  print(res[0], res[2], {k: v for k, v in res[1].items() if k not in ("observation_radius",)}, flush=True)
  if res[2] is not True: bad += 1
print("cases", n, "bad", bad)
sys.exit(1 if bad else 0)
