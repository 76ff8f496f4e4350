"""GPU parity test of BASELINE.json configs[4]: the mixed suite (island_navigation_ex + boat_race_ex + safe_interruptibility on
three streams, env-id ranges sharded over ranks, summed episodic returns) exactly as bench.py builds and launches it
(`bench.mixed_parts`, `build_engines`, `run_batches`, per-family `read_returns`).

  (a) every family's outputs == the CPU oracle on the family's GLOBAL id range, every step;
  (b) a 2-shard run (both ranks' engines on this one GPU) concatenates to the 1-shard run bit for bit;
  (c) the summed `read_returns` are equal, and equal to the oracle's finished-episode sums;
  (d) at the BASELINE per-GPU size the multi-launch / multi-stream path (no host sync between launches) ends in the same
      outputs and returns for 1 and 2 shards (size-independent property; no oracle at that size).
Reference semantics: island_navigation_ex.py:449-704, boat_race_ex.py:201-257, safe_interruptibility.py:192-269."""
import numpy as np
import pytest
import torch

import bench as B
from ai_safety_gridworlds_amd import philox
from tests import golden_util as G

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ACTION_RANGE = {"island_navigation_ex": (0, 5), "boat_race_ex": (0, 5), "safe_interruptibility": (1, 4)}
ORACLE_NAME = {"hidden": "hidden", "safety": "safety", "frame": "frame", "board": "board", "reward": "reward",
               "step_type": "step_type", "term_reason": "term_reason"}


def oracle_part(fam, cnt, base, T):
  """The oracle on global env ids [base, base + cnt): the action / episode-bit streams the engine derives on the device."""
  from oracle import oracle as O
  lo, n = ACTION_RANGE[fam]
  ids = base + np.arange(cnt)
  acts = philox.actions(B.SEED, ids, np.arange(T), lo, n).T.copy()                       # [E, T]
  bits = None
  kw = B.WORKLOADS[fam]["kwargs"]
  if fam == "safe_interruptibility":                                                       # should_interrupt of the k-th game build
    u = philox.episode_uniform(B.SEED, ids[:, None], np.arange(T + 2)[None, :])
    bits = (u <= 0.5).astype(np.uint8)
  return O.run_streams(O.make_config(fam, **kw), acts, interrupt_bits=bits, nthreads=8)


def oracle_returns(want):
  last = want["step_type"][:, 1:] == 2
  acc = np.zeros(want["K"] + 1)
  acc[:-1] = (want["cumulative"][:, 1:] * last[..., None]).sum(axis=(0, 1))
  acc[-1] = last.sum()
  return acc


def snapshot(engines):
  """Current outputs of every engine -> {family: {field: [n, ...] numpy}} concatenated in global-id order."""
  torch.cuda.synchronize()
  out = {}
  for e in sorted(engines, key=lambda e: e["base"]):
    v = e["eng"]._views()
    d = out.setdefault(e["fam"], {})
    for k in e["wl"]["outputs"]:
      d.setdefault(k, []).append(v[k].cpu().numpy())
  return {fam: {k: np.concatenate(v, axis=0) for k, v in d.items()} for fam, d in out.items()}


def family_returns(engines):
  acc = {}
  for e in engines:
    r = e["eng"].read_returns().cpu().numpy()
    acc[e["fam"]] = acc.get(e["fam"], 0) + r
  return acc


def close(engines):
  for e in engines:
    e["eng"].close()


def test_mixed_suite_matches_oracle_and_shards_concatenate():
  per_gpu, T = 3000, 110                       # three ragged 1000-env ranges; world 2: 1500 per rank, cut inside boat_race_ex
  one = B.build_engines(B.mixed_parts(0, 1, per_gpu), DEV)
  two = B.build_engines(B.mixed_parts(0, 2, per_gpu // 2) + B.mixed_parts(1, 2, per_gpu // 2), DEV)
  assert [e["fam"] for e in one] == list(B.MIXED) and len(two) == 4
  assert sum(e["n"] for e in two) == per_gpu and sorted(e["base"] for e in two) == [0, 1000, 1500, 2000]
  B.fill_action_batches(one, 1, T)             # T distinct one-step batches: (global id, step) keyed Philox
  B.fill_action_batches(two, 1, T)
  want = {fam: oracle_part(fam, cnt, base, T) for fam, cnt, base in B.mixed_parts(0, 1, per_gpu)}
  for t in range(T):
    B.run_batches(one, 1, t, 1, True)          # one launch per family on its own stream, as the bench issues them
    B.run_batches(two, 1, t, 1, True)
    s1, s2 = snapshot(one), snapshot(two)
    for fam in B.MIXED:
      for k, g in s1[fam].items():
        w = want[fam][ORACLE_NAME[k]][:, t + 1]
        if k == "term_reason":
          g = g.astype(np.int16); g[g == 255] = -1
        G.assert_same("%s.%s step %d" % (fam, k, t), g.reshape(w.shape), w)                        # (a)
        assert np.array_equal(s1[fam][k], s2[fam][k]), "2 shards != 1 shard: %s.%s step %d" % (fam, k, t)   # (b)
  r1, r2 = family_returns(one), family_returns(two)
  for fam in B.MIXED:
    assert np.array_equal(r1[fam], r2[fam]), fam                                                    # (c)
    assert np.array_equal(r1[fam], oracle_returns(want[fam])), fam
    assert r1[fam][-1] > 0
  close(one); close(two)


def test_mixed_suite_multistream_launcher_at_bench_size():
  per_gpu, K, R = 32768, 25, 6                 # bench.py's mixed size per GPU; 6 back-to-back batches of 25 launches per stream
  one = B.build_engines(B.mixed_parts(0, 1, per_gpu), DEV)
  two = B.build_engines(B.mixed_parts(0, 2, per_gpu // 2) + B.mixed_parts(1, 2, per_gpu // 2), DEV)
  B.fill_action_batches(one, K, R)
  B.fill_action_batches(two, K, R)
  B.run_batches(one, K, 0, R, True)            # no host synchronisation between the launches of a stream or between streams
  B.run_batches(two, K, 0, R, True)
  s1, s2 = snapshot(one), snapshot(two)
  for fam in B.MIXED:
    for k in s1[fam]:
      assert np.array_equal(s1[fam][k], s2[fam][k]), (fam, k)
  r1, r2 = family_returns(one), family_returns(two)
  for fam in B.MIXED:
    assert np.array_equal(r1[fam], r2[fam]) and r1[fam][-1] > 0, fam
  # the same run again from scratch is bit-identical: nothing in the launch sequence races the reset / action kernels
  again = B.build_engines(B.mixed_parts(0, 1, per_gpu), DEV)
  B.fill_action_batches(again, K, R)
  B.run_batches(again, K, 0, R, True)
  s3, r3 = snapshot(again), family_returns(again)
  for fam in B.MIXED:
    for k in s1[fam]:
      assert np.array_equal(s1[fam][k], s3[fam][k]), (fam, k)
    assert np.array_equal(r1[fam], r3[fam])
  close(one); close(two); close(again)


def test_group_launch_equals_per_family_launches_and_the_oracle():
  """sgw_group_step_n / sgw_group_rollout (ONE heterogeneous launch per step over the three families) against the per-family
  launches and the oracle: every output every step, the returns, and the raw state; two shards in one group == one shard."""
  from ai_safety_gridworlds_amd.engine import EngineGroup
  per_gpu, T = 3000, 90
  parts = B.mixed_parts(0, 1, per_gpu)
  solo = B.build_engines(parts, DEV)
  grp = B.build_engines(parts, DEV, streams=False)
  two = B.build_engines(B.mixed_parts(0, 2, per_gpu // 2) + B.mixed_parts(1, 2, per_gpu // 2), DEV, streams=False)
  g1, g2 = EngineGroup([e["eng"] for e in grp]), EngineGroup([e["eng"] for e in two])
  for es in (solo, grp, two):
    B.fill_action_batches(es, 1, T)
  want = {fam: oracle_part(fam, cnt, base, T) for fam, cnt, base in parts}
  for t in range(T):
    B.run_batches(solo, 1, t, 1, True)
    B.run_batches(grp, 1, t, 1, True, g1)
    B.run_batches(two, 1, t, 1, True, g2)
    s0, s1, s2 = snapshot(solo), snapshot(grp), snapshot(two)
    for fam in B.MIXED:
      for k, g in s1[fam].items():
        w = want[fam][ORACLE_NAME[k]][:, t + 1]
        assert np.array_equal(s0[fam][k], g), "group != per-family launch: %s.%s step %d" % (fam, k, t)
        assert np.array_equal(s2[fam][k], g), "2 shards in a group != 1 shard: %s.%s step %d" % (fam, k, t)
        if k == "term_reason":
          g = g.astype(np.int16); g[g == 255] = -1
        G.assert_same("%s.%s step %d" % (fam, k, t), g.reshape(w.shape), w)
  r0, r1, r2 = family_returns(solo), family_returns(grp), family_returns(two)
  for fam in B.MIXED:
    assert np.array_equal(r0[fam], r1[fam]) and np.array_equal(r1[fam], r2[fam]) and r1[fam][-1] > 0, fam
    assert np.array_equal(r1[fam], oracle_returns(want[fam])), fam
  for a, b in zip(solo, grp):
    assert torch.equal(a["eng"].get_state()[:, :a["n"]], b["eng"].get_state()[:, :b["n"]]), a["fam"]
  # K >= 8: the second identical call is captured, the third replays the graph; the fused group rollout == the step loop
  K = 12
  for es in (solo, grp):
    B.fill_action_batches(es, K, 3, step0=T)
  B.run_batches(solo, K, 0, 3, True)
  for j in range(3):
    g1.step_n([e["acts"][j * K:(j + 1) * K] for e in grp], accumulate=True)
  s0, s1 = snapshot(solo), snapshot(grp)
  for fam in B.MIXED:
    for k in s0[fam]:
      assert np.array_equal(s0[fam][k], s1[fam][k]), (fam, k)
  same = [e["acts"][:K] for e in grp]
  before = [e["eng"].get_state().clone() for e in grp]
  outs = []
  for rep in range(3):                       # direct, capture, replay -- from the same start state
    for e, st in zip(grp, before):
      e["eng"].set_state(st.clone())
    g1.step_n(same)
    outs.append(snapshot(grp))
  for fam in B.MIXED:
    for k in outs[0][fam]:
      assert np.array_equal(outs[0][fam][k], outs[1][fam][k]) and np.array_equal(outs[0][fam][k], outs[2][fam][k]), (fam, k)
  # fused: one group rollout of T2 steps == T2 single group steps over the same Philox stream
  T2 = 40
  for e, st in zip(grp, before):
    e["eng"].set_state(st.clone())
  for e, st in zip(solo, before):
    e["eng"].set_state(st.clone())
  ro = g1.rollout(T2, B.SEED, step0=500, write_every=True)
  ro = [{k: v.clone() for k, v in o.items()} for o in ro]
  for e, o in zip(solo, ro):
    acts = e["eng"].fill_actions(T2, B.SEED, step0=500)
    per = e["eng"].step_n(acts, write_every=True)
    for k in e["wl"]["outputs"]:
      assert torch.equal(per[k], o[k]), (e["fam"], k)
  for a, b in zip(solo, grp):
    assert torch.equal(a["eng"].get_state()[:, :a["n"]], b["eng"].get_state()[:, :b["n"]]), a["fam"]
  g1.close(); g2.close()
  close(solo); close(grp); close(two)


def test_group_refuses_round_kernels_and_foreign_engines():
  from ai_safety_gridworlds_amd import _native as N
  from ai_safety_gridworlds_amd.engine import BatchedEngine, EngineGroup
  from ai_safety_gridworlds_amd.specs import make_spec
  a = BatchedEngine(make_spec("island_navigation_ex"), 128)
  b = BatchedEngine(make_spec("firemaker_ex_ma"), 64)
  with pytest.raises(N.SgwError, match="not a group member"):
    EngineGroup([a, b])
  with pytest.raises(N.SgwError, match="listed twice"):
    EngineGroup([a, a])
  a.close(); b.close()
