"""The per-lane device code of the three big families (csrc/sgw_island.hpp, sgw_island_ma.hpp, sgw_savanna.hpp), compiled
for the HOST through a minimal HIP stand-in (tests/host_shim/) with clang's UndefinedBehaviorSanitizer + AddressSanitizer and
pattern-initialised locals, driven one env at a time through k_engine's sequence and compared with the REFERENCE fixtures.
No GPU needed: this is where undefined behaviour or an uninitialised read in the kernel source would show up (a sanitizer
report fails the run; a pattern-initialised local changes the outputs).  Test infrastructure only -- the product has no
CPU path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from ai_safety_gridworlds_amd import _native as N
from ai_safety_gridworlds_amd.specs import make_spec
from tests import golden_util as G

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "host_shim")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
  out = str(tmp_path_factory.mktemp("host") / "host_families")
  cc = CLANG if os.path.exists(CLANG) else "g++"
  flags = ["-std=c++17", "-O1", "-g", "-ffp-contract=off", "-mfma", "-fsanitize=undefined,address", "-fno-sanitize-recover=all"]
  if cc == CLANG:
    flags.append("-ftrivial-auto-var-init=pattern")
  subprocess.check_call([cc] + flags + ["-I" + SHIM, os.path.join(SHIM, "host_families.cpp"), "-o", out])
  return out


def run(exe, tmp_path, spec, actions, rng, proto, bits=None, stream=None):
  """actions int8 [E, T, A]; returns dict of arrays [E, S, ...] (S = T + 1 + proto)."""
  E, T, A = actions.shape
  inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
  table = getattr(spec, "family_table", None)
  with open(inp, "wb") as f:
    f.write(np.array([spec.family, E, T, proto], np.int32).tobytes())
    f.write(bytes(spec.native))
    f.write(np.array([0 if table is None else len(table)], np.int32).tobytes())
    if table is not None:
      f.write(np.ascontiguousarray(table, np.float64).tobytes())
    f.write(np.ascontiguousarray(rng, np.uint64).tobytes())
    f.write(np.ascontiguousarray(actions, np.int8).tobytes())
    for arr, dt in ((bits, np.uint8), (stream, np.float64)):
      if arr is None or arr.shape[1] == 0:
        f.write(np.array([0], np.int32).tobytes())
      else:
        f.write(np.array([arr.shape[1]], np.int32).tobytes())
        f.write(np.ascontiguousarray(arr, dt).tobytes())
  env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
  p = subprocess.run([exe, inp, outp], capture_output=True, text=True, env=env)
  assert p.returncode == 0, p.stderr[-3000:]
  S = T + 1 + proto
  LA = 2 if spec.family in (N.ISLAND_NAVIGATION_EX_MA, N.AINTELOPE_SAVANNA) else 1
  AK, HW = LA * spec.K, spec.H * spec.W
  dt = np.dtype([("step_type", np.int32, (LA,)), ("frame", np.int32), ("reward", np.float64, (AK,)),
                 ("cumulative", np.float64, (AK,)), ("board", np.uint8, (HW,))], align=False)
  rec = np.fromfile(outp, dtype=dt).reshape(E, S)
  return {k: rec[k] for k in dt.names}


MA_FIXTURES = [n for n in G.fixture_names(["ima_", "sav_"])]


@pytest.mark.parametrize("name", MA_FIXTURES)
def test_multi_agent_family_source_on_the_host_matches_reference(name, exe, tmp_path):
  fx, meta = G.load(name)
  env_name = "island_navigation_ex_ma" if name.startswith("ima_") else "aintelope_savanna"
  spec = make_spec(env_name, **meta["kwargs"])
  actions = fx["actions"]
  if actions.shape[2] == 1:
    actions = np.concatenate([actions, np.zeros_like(actions)], axis=2)
  got = run(exe, tmp_path, spec, actions, fx["rng_seeded"], 1)
  A = fx["step_type"].shape[2]
  E, S = fx["step_type"].shape[:2]
  sl = slice(1, None)
  G.assert_same(name + ".step_type", got["step_type"][:, sl, :A], fx["step_type"][:, sl])
  G.assert_same(name + ".frame", got["frame"][:, sl], fx["frame"][:, sl])
  G.assert_same(name + ".board", got["board"][:, sl].reshape(fx["board"][:, sl].shape), fx["board"][:, sl])
  G.assert_same(name + ".reward", got["reward"][:, sl].reshape(E, S - 1, 2, spec.K)[:, :, :A], fx["reward"][:, sl])
  G.assert_same(name + ".cumulative", got["cumulative"][:, sl].reshape(E, S - 1, 2, spec.K)[:, :, :A], fx["cumulative"][:, sl])


@pytest.mark.parametrize("name", [n for n in G.fixture_names(["island_"]) if not n.startswith("island_nav")])
def test_island_source_on_the_host_matches_reference(name, exe, tmp_path):
  fx, meta = G.load(name)
  spec = make_spec("island_navigation_ex", **meta["kwargs"])
  actions = fx["actions"][:, :, None]
  E = actions.shape[0]
  got = run(exe, tmp_path, spec, actions, np.zeros((E, 4), np.uint64), 0)
  G.assert_same(name + ".step_type", got["step_type"][:, :, 0], fx["step_type"])
  G.assert_same(name + ".frame", got["frame"], fx["frame"])
  G.assert_same(name + ".board", got["board"].reshape(fx["board"].shape), fx["board"])
  G.assert_same(name + ".reward", got["reward"], fx["reward"].reshape(got["reward"].shape))
  G.assert_same(name + ".cumulative", got["cumulative"], fx["cumulative"].reshape(got["cumulative"].shape))


@pytest.mark.parametrize("name", G.fixture_names(["boat_", "sokoban_", "conveyor_", "rocks_", "conveyorex_"]))
def test_deterministic_scalar_family_source_on_the_host_matches_reference(name, exe, tmp_path):
  fx, meta = G.load(name)
  spec = make_spec(meta["family_name"], **meta["kwargs"])
  actions = fx["actions"][:, :, None]
  got = run(exe, tmp_path, spec, actions, np.zeros((actions.shape[0], 4), np.uint64), 0)
  G.assert_same(name + ".step_type", got["step_type"][:, :, 0], fx["step_type"])
  G.assert_same(name + ".frame", got["frame"], fx["frame"])
  G.assert_same(name + ".board", got["board"].reshape(fx["board"].shape), fx["board"])
  G.assert_same(name + ".reward", got["reward"], fx["reward"].reshape(got["reward"].shape))


@pytest.mark.parametrize("name", G.fixture_names(["safe_int_", "islnav_", "dshift_", "absent_", "tomato_", "friendfoe_", "whisky_", "safeintex_"]))
def test_externally_randomised_scalar_family_source_on_the_host_matches_reference(name, exe, tmp_path):
  """Families that take the reference's process-global random numbers as inputs (episode bits / in-play stream)."""
  fx, meta = G.load(name)
  spec = make_spec(meta["family_name"], **meta["kwargs"])
  actions = fx["actions"][:, :, None]
  bits = G.interrupt_bits(fx) if "should_interrupt" in fx.files else None
  stream = fx["rand_stream"] if "rand_stream" in fx.files and fx["rand_stream"].shape[1] else None
  got = run(exe, tmp_path, spec, actions, np.zeros((actions.shape[0], 4), np.uint64), 0, bits=bits, stream=stream)
  G.assert_same(name + ".step_type", got["step_type"][:, :, 0], fx["step_type"])
  G.assert_same(name + ".frame", got["frame"], fx["frame"])
  G.assert_same(name + ".board", got["board"].reshape(fx["board"].shape), fx["board"])
  G.assert_same(name + ".reward", got["reward"], fx["reward"].reshape(got["reward"].shape))



def test_reciprocal_row():
  """csrc/sgw_savanna.hpp row_of(): cell / W as (cell * (65536 / W + 1)) >> 16 is exact for every board the family can hold."""
  for W in range(1, 192):
    inv = 65536 // W + 1
    assert all(((cell * inv) & 0xffffffff) >> 16 == cell // W for cell in range(192)), W
