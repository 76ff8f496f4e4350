"""Helpers shared by the oracle and GPU parity tests: load fixtures, map them to configs."""
import ast
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FAMILY_NAMES = {"island_ex": "island_navigation_ex", "boat_race_ex": "boat_race_ex",
                "boat_race": "boat_race", "safe_interruptibility": "safe_interruptibility"}
SCALAR_PREFIXES = ["island_", "boat_", "safe_int_", "islnav_", "dshift_", "absent_", "sokoban_", "conveyor_", "tomato_", "friendfoe_", "whisky_", "rocks_", "conveyorex_", "safeintex_"]     # single-agent fixture families


def fixture_names(prefixes=None):
  names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
  if prefixes:
    names = [n for n in names if any(n.startswith(p) for p in prefixes)]
  return names


def load(name):
  fx = np.load(os.path.join(GOLDEN, name + ".npz"))
  meta = {k[5:]: (fx[k].item() if fx[k].ndim == 0 else fx[k]) for k in fx.files if k.startswith("meta_")}
  meta["family_name"] = FAMILY_NAMES.get(meta["family"], meta["family"])
  meta["kwargs"] = dict(ast.literal_eval(meta["kwargs"]))
  if "experiment" in meta["kwargs"]:      # an experiments/ preset: the flag values come from the package's preset table,
    from ai_safety_gridworlds_amd.specs import EXPERIMENT_PRESETS   # so the fixture (made by the reference's subclass) pins them
    kw = dict(meta["kwargs"])
    meta["experiment"] = kw.pop("experiment")
    merged = dict(EXPERIMENT_PRESETS[meta["experiment"]][2])
    merged.update(kw)
    meta["kwargs"] = merged
  meta["dim_names"] = [s for s in meta.get("dim_names", "").split("|") if s]
  meta["metric_labels"] = [s for s in meta.get("metric_labels", "").split("|") if s]
  return fx, meta


def interrupt_bits(fx, n_bits=None):
  """safe_interruptibility: per-stream should_interrupt bit of the k-th game build,
  recovered from the recorded per-step values (one build per FIRST step)."""
  st = fx["step_type"]
  si = fx["should_interrupt"]
  E = st.shape[0]
  per = [si[e][st[e] == 0].astype(np.uint8) for e in range(E)]
  n = max(len(p) for p in per) if n_bits is None else n_bits
  bits = np.zeros((E, n), np.uint8)
  for e, p in enumerate(per):
    bits[e, :len(p)] = p[:n]
  return bits


def performance_mask(fx):
  """last_performance is only comparable once an episode has ended INSIDE the stream (the
  generator reuses one env object across streams, so earlier values leak from stream e-1)."""
  ended = np.cumsum(fx["step_type"] == 2, axis=1) > 0
  return ended


def assert_same(name, got, want):
  got = np.asarray(got)
  want = np.asarray(want)
  assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
  if got.dtype.kind == "f" or want.dtype.kind == "f":
    same = (got == want) | (np.isnan(got) & np.isnan(want))
  else:
    same = got.astype(np.int64) == want.astype(np.int64)
  if not same.all():
    bad = np.argwhere(~same)
    raise AssertionError("%s: %d mismatches, first at %s: got %r want %r" % (
        name, len(bad), tuple(bad[0]), got[tuple(bad[0])], want[tuple(bad[0])]))
