"""Host (numpy) statement of the synthetic-action stream used by bench/tests.

Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3",
SC'11), counter-based so that the HIP kernels (csrc/sgw_philox.hpp), this numpy
version, the CPU oracle replay and the reference fixture generator all see the
*same* action for (seed, global env id, agent, step) regardless of GPU count or
launch shape (SURVEY.md §8d "Synthetic inputs").

  key     = (seed & 0xffffffff, env_id & 0xffffffff)
  counter = (step, stream_tag, agent, env_id >> 32)
  action  = lo + ((x0 * n_actions) >> 32)          # multiply-shift, x0 = word 0

stream_tag separates uses: 0 = actions, 1 = per-episode Bernoulli draws
(safe_interruptibility should_interrupt), 2 = reserved.
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)

TAG_ACTION = 0
TAG_EPISODE = 1


def philox4x32_10(c0, c1, c2, c3, k0, k1):
  """Vectorised Philox-4x32-10. All args broadcastable integer arrays (< 2**32).

  Returns four uint32 arrays.
  """
  c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(
      *[np.asarray(x, dtype=np.uint64) & _MASK for x in (c0, c1, c2, c3, k0, k1)])
  c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
  k0, k1 = k0.copy(), k1.copy()
  for _ in range(10):
    p0 = _M0 * c0            # < 2**64, exact in uint64
    p1 = _M1 * c2
    hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
    hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
    c0, c1, c2, c3 = (hi1 ^ c1 ^ k0), lo1, (hi0 ^ c3 ^ k1), lo0
    k0 = (k0 + np.uint64(_W0)) & _MASK
    k1 = (k1 + np.uint64(_W1)) & _MASK
  return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def actions(seed, env_ids, steps, lo, n_actions, agent=0):
  """int8 actions for every (step, env): shape [len(steps), len(env_ids)]."""
  env_ids = np.asarray(env_ids, dtype=np.uint64)
  steps = np.asarray(steps, dtype=np.uint64)
  x0, _, _, _ = philox4x32_10(
      steps[:, None], TAG_ACTION, agent, env_ids[None, :] >> np.uint64(32),
      seed, env_ids[None, :])
  a = (x0.astype(np.uint64) * np.uint64(n_actions)) >> np.uint64(32)
  return (a.astype(np.int64) + lo).astype(np.int8)


def episode_uniform(seed, env_ids, episode_idx):
  """Per-(env, episode) uniform double in [0,1): (x0<<21 | x1>>11) * 2**-53."""
  env_ids = np.asarray(env_ids, dtype=np.uint64)
  episode_idx = np.asarray(episode_idx, dtype=np.uint64)
  x0, x1, _, _ = philox4x32_10(
      episode_idx, TAG_EPISODE, 0, env_ids >> np.uint64(32), seed, env_ids)
  bits = (x0.astype(np.uint64) << np.uint64(21)) | (x1.astype(np.uint64) >> np.uint64(11))
  return bits.astype(np.float64) * (1.0 / 9007199254740992.0)
