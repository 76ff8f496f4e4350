#!/usr/bin/env python3
"""Time the REFERENCE (CPython object graph, one env, one core) on the bench workloads, in the build container.

    python tools/time_reference.py [--seconds 10]

The reference cannot travel to the GPU box (only /root/repo ships), so its steps/s cannot be measured next to the GPU
numbers; bench.py quotes the figure this script writes to profiles/reference_cpython.json (labelled with where it was
measured) beside the C port that IS timed on the GPU box.  Envs are built exactly as the fixture generator builds them
(tests/golden/make_fixtures.py: the reference imported from /root/reference with the test-only stand-ins for absl /
gymnasium.utils.seeding), stepped at the SafetyEnvironment*.step() level with uniform Philox actions, auto-reset included.
"""
import argparse
import json
import os
import platform
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))

WORKLOADS = {
    "island_navigation_ex": ("island_ex", dict(level=9), 0, 5),
    "boat_race_ex": ("boat_race_ex", dict(level=3), 0, 5),
    "safe_interruptibility": ("safe_interruptibility", dict(level=1), 1, 4),
    "boat_race": ("boat_race", dict(level=0), 1, 4),
}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--seconds", type=float, default=10.0)
  a = ap.parse_args()
  import make_fixtures as MF
  if not os.path.isdir(MF.REFERENCE):
    sys.exit("reference not present: it can only be timed in the build container")
  os.chdir(tempfile.mkdtemp(prefix="sgw_time_ref_"))     # the reference's logger writes into the cwd
  MF._setup_path()
  import numpy as np
  from ai_safety_gridworlds_amd import philox
  out = {"where": "build container: %s, %d cpus visible, 1 core used, CPython %s, numpy %s"
                  % (platform.processor() or platform.machine(), os.cpu_count(), platform.python_version(), np.__version__),
         "how": "tools/time_reference.py: SafetyEnvironment*.step() loop, uniform Philox actions, auto-reset included"}
  for name, (family, kw, lo, n) in WORKLOADS.items():
    env, _ = MF.make_env(family, kw)
    env.reset()
    acts = philox.actions(MF.SEED, np.arange(1), np.arange(100000), lo, n)[:, 0]
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < a.seconds:
      for k in range(500):
        env.step(int(acts[(steps + k) % len(acts)]))
      steps += 500
    dt = time.perf_counter() - t0
    out[name] = {"steps_per_s": steps / dt, "steps": steps, "seconds": dt, "kwargs": kw}
    print(name, "%.0f steps/s" % (steps / dt), flush=True)
  path = os.path.join(REPO, "profiles", "reference_cpython.json")
  json.dump(out, open(path, "w"), indent=1, sort_keys=True)
  print("wrote", path)


if __name__ == "__main__":
  main()
