"""Pin the aintelope_savanna oracle bit-for-bit against fixtures captured from the reference
(tests/golden/make_fixtures_sav.py): tile-count removal + interior shuffle of the map, resource tiles spawning and
vanishing through Generator.choice(replace=False), predators, cooperation and logarithmic gold / silver rewards,
overlapping drape layers, and the None rows of the metrics matrix."""
import numpy as np
import pytest

from oracle import oracle_sav as OS
from tests import golden_util as G

FIELDS = ["step_type", "reward", "cumulative", "discount", "term_reason", "frame", "board", "layers", "metrics", "pos",
          "action_direction", "observation_direction", "safety", "safety2", "rng", "rng_has_uint32", "rng_uinteger", "view"]


@pytest.mark.parametrize("name", G.fixture_names(["sav_"]))
def test_sav_oracle_matches_reference_fixture(name):
  fx, meta = G.load(name)
  cfg = OS.make_config(**meta["kwargs"])
  out = OS.run_streams(cfg, fx["actions"], fx["rng_seeded"])
  G.assert_same(name + ".rng[0]", out["rng"][:, 0], fx["rng"][:, 0])
  for f in FIELDS:
    G.assert_same(name + "." + f, out[f][:, 1:], fx[f][:, 1:])
  assert (out["reward_none"][:, 1:].astype(bool) == fx["reward_none"][:, 1:]).all()
