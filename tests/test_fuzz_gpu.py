"""GPU tier: a bounded, fixed-seed slice of the randomised configuration sweep (tests/fuzz_cases.py; the open-ended tool is
tools/diag/fuzz_parity.py): HIP engine vs the C oracle on random levels / flags / tile amounts / map sizes / direction modes /
single-agent rounds / explicit resets of island_navigation_ex, island_navigation_ex_ma, aintelope_savanna and firemaker_ex_ma --
every output of every step, the metrics and the numpy generator position.  Every kernel edit is validated by this in the GPU
tier, not only by a tool somebody has to remember to run."""
import numpy as np
import pytest

from tests import fuzz_cases as F

pytestmark = pytest.mark.gpu


def _run(case, n_cases, seed):
  rnd = np.random.default_rng(seed)
  ran, tries = 0, 0
  while ran < n_cases and tries < 4 * n_cases:
    tries += 1
    res = case(rnd)
    if res is None:                      # a configuration the reference (and make_spec) refuses
      continue
    assert res[2] is True, "mismatch: %s %r -> %r" % (res[0], res[1], res[2])
    ran += 1
  assert ran == n_cases


def test_fuzz_island_navigation_ex():
  _run(lambda r: F.island_case(r, E=512, T=120, nthreads=8), 150, seed=20261)


def test_fuzz_island_navigation_ex_ma():
  _run(lambda r: F.ma_case(r, "ima", E=320, T=100, nthreads=8), 150, seed=20262)


def test_fuzz_aintelope_savanna():
  _run(lambda r: F.ma_case(r, "sav", E=256, T=100, nthreads=8), 150, seed=20263)


def test_fuzz_firemaker_ex_ma():
  _run(lambda r: F.firemaker_case(r, E=192, T=110, nthreads=8), 60, seed=20264)
