"""TEST-ONLY stand-in for gymnasium.utils.seeding.np_random.

Published gymnasium behaviour restated: a numpy Generator over PCG64 seeded from
SeedSequence(seed).  `.rand` is added (legacy-gym RandomNumberGenerator semantics,
= Generator.random) because firemaker_ex_ma.py:615,621 calls NP_RANDOM.rand().
"""
import numpy as np


class RandomNumberGenerator(np.random.Generator):

  def rand(self, *size):
    return self.random(size if size else None)


def np_random(seed=None):
  if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
    raise ValueError("Seed must be a non-negative integer or omitted, not %r" % (seed,))
  seed_seq = np.random.SeedSequence(seed)
  np_seed = seed_seq.entropy
  rng = RandomNumberGenerator(np.random.PCG64(seed_seq))
  return rng, np_seed
