"""Pin the multi-agent oracle (firemaker_ex_ma): numpy PCG64 conformance + bit-for-bit against fixtures
captured from the reference run with the two documented patches (tests/golden/make_fixtures_ma.py)."""
import ast

import numpy as np
import pytest

from oracle import oracle_ma as OM
from tests import golden_util as G

FIELDS = ["step_type", "reward", "cumulative", "discount", "term_reason", "frame", "board", "metrics", "pos",
          "rng", "rng_has_uint32", "rng_uinteger", "view_worker", "view_supervisor"]


def test_pcg64_random_uint32_and_list_shuffle_match_numpy():
  for seed in (0, 1, 1000, 2**31 + 7):
    st = OM.rng_state_words(seed)
    n = 200
    r, u, p = OM.rng_probe(st, n)
    g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    for i in range(n):
      assert g.random() == r[i]
      assert int(g.integers(0, 2**32, dtype=np.uint32)) == int(u[i])      # one buffered next_uint32
      items = [0, 1, 2]
      g.shuffle(items)                                                     # list path: Fisher-Yates + random_interval
      assert items == list(p[i])


def _cfg(meta):
  return OM.make_config(**dict(meta["kwargs"]))


@pytest.mark.parametrize("name", G.fixture_names(["firemaker_"]))
def test_ma_oracle_matches_reference_fixture(name):
  fx, meta = G.load(name)
  cfg = _cfg(meta)
  out = OM.run_streams(cfg, fx["actions"], fx["rng_init"])
  for f in FIELDS:
    G.assert_same(name + "." + f, out[f], fx[f])
  if "action_direction" in fx:           # the direction-mode fixtures (round 3): Directions per agent after every round
    for f in ("action_direction", "observation_direction"):
      G.assert_same(name + "." + f, out[f], fx[f])
  assert (out["reward_none"].astype(bool) == fx["reward_none"]).all()
  assert OM.rng_state_words(int(fx["seeds"][3]))[1] == fx["rng_init"][3][1]   # seeding is plain numpy
