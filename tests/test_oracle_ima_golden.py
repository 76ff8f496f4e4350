"""Pin the island_navigation_ex_ma oracle bit-for-bit against fixtures captured from the reference
(tests/golden/make_fixtures_ima.py): per-agent termination (LAST/DEAD), relative action and observation
directions, rotated agent views, and map randomisation (Generator.shuffle of the interior, cached per episode)."""
import numpy as np
import pytest

from oracle import oracle_ima as OI
from tests import golden_util as G

FIELDS = ["step_type", "reward", "cumulative", "discount", "term_reason", "frame", "board", "metrics", "pos",
          "action_direction", "observation_direction", "safety", "rng", "rng_has_uint32", "rng_uinteger", "view"]


@pytest.mark.parametrize("name", G.fixture_names(["ima_"]))
def test_ima_oracle_matches_reference_fixture(name):
  fx, meta = G.load(name)
  cfg = OI.make_config(**meta["kwargs"])
  out = OI.run_streams(cfg, fx["actions"], fx["rng_seeded"])
  # slot 0 is the constructor's internal reset: the reference only leaves the generator state and the map behind
  G.assert_same(name + ".rng[0]", out["rng"][:, 0], fx["rng"][:, 0])
  G.assert_same(name + ".art0", out["board"][:, 0], fx["art0"])    # at frame 0 the board IS the (shuffled) ascii art
  for f in FIELDS:
    G.assert_same(name + "." + f, out[f][:, 1:], fx[f][:, 1:])
  assert (out["reward_none"][:, 1:].astype(bool) == fx["reward_none"][:, 1:]).all()
  if meta["kwargs"].get("map_randomization_frequency", 0):
    # the randomised map of slot 0 (agents at their shuffled start cells) is the board of slot 1 (cached, no new draw)
    G.assert_same(name + ".map0", out["board"][:, 0], out["board"][:, 1])
    firsts = fx["board"][:, 1:][fx["frame"][:, 1:] == 0]
    n_maps = len({b.tobytes() for b in firsts})
    if meta["kwargs"]["map_randomization_frequency"] == 3:
      assert n_maps > fx["board"].shape[0]                                      # explicit resets drew new maps
    else:
      assert n_maps == fx["board"].shape[0]                                     # once per experiment
