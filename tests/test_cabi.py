"""CPU tier: libsgw.so loads and exports every symbol include/sgw.h declares; struct mirrors match;
the product fails loudly (no CPU fallback) when no HIP device exists; the product never touches oracle/."""
import ctypes as C
import os
import re

import pytest

from ai_safety_gridworlds_amd import _native as N

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
  text = open(os.path.join(REPO, "include", "sgw.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(sgw_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
  L = N.lib()
  names = declared_symbols()
  assert len(names) >= 20
  missing = [n for n in names if not hasattr(L, n)]
  assert not missing, missing
  assert sorted(N.EXPORTS) == names, "binding list and header disagree"


def test_struct_mirrors_match_the_compiled_layout():
  L = N.lib()
  assert L.sgw_abi_version() == N.ABI_VERSION
  assert L.sgw_sizeof_spec() == C.sizeof(N.Spec)
  assert L.sgw_sizeof_out() == C.sizeof(N.Out)


def test_product_does_not_reference_the_oracle():
  """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
  pkg = os.path.join(REPO, "ai_safety_gridworlds_amd")
  for root, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
        text = open(os.path.join(root, f)).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
        assert "sgw_oracle" not in text and "libsgw_oracle" not in text, f
  header = open(os.path.join(REPO, "include", "sgw.h")).read()
  assert "oracle" not in header.lower()


def test_engine_fails_loudly_without_a_gpu():
  import torch
  if torch.cuda.is_available():
    pytest.skip("a GPU is present")
  from ai_safety_gridworlds_amd.engine import BatchedEngine
  from ai_safety_gridworlds_amd.specs import make_spec
  with pytest.raises(N.SgwError):
    BatchedEngine(make_spec("island_navigation_ex"), 64, device="cuda:0")
  with pytest.raises(N.SgwError):
    BatchedEngine(make_spec("island_navigation_ex"), 64, device="cpu")


def test_host_libm_matches_the_pow_tables():
  """sgw_create refuses the regrowth families when the running libm's pow() (the reference's math.pow) is not the one the
  device tables were extracted from; on the build host the probe finds no difference."""
  assert N.lib().sgw_pow_selfcheck() == 0
