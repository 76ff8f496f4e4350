"""CPU tier: the gfx950 code of EVERY kernel in libsgw.so is checked for the compiler defect that corrupted
k_engine<IslandMa, K_ROLLOUT> in round 1 (VGPR saves / live-range copies placed AHEAD of the `s_or_b64 exec` that re-enables
the lanes at an `s_cbranch_execz` join; tools/isa_lint.py).  hipcc cross-compiles the assembly without a GPU."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import isa_lint  # noqa: E402

BAD = """
_ZN3sgw8k_engineINS_8IslandMaELi1EEEvNS_5KArgsE:
; %bb.0:
\tv_cmp_ne_u32_e64 s[82:83], 0, v113
\ts_and_saveexec_b64 s[86:87], s[82:83]
\ts_cbranch_execz .LBB28_70
; %bb.67:
\tv_add_f64 v[0:1], v[2:3], v[4:5]
.LBB28_70:                              ; %Flow1679
\tv_mov_b32_e32 v205, v193
\tv_readlane_b32 s38, v251, 45
\tv_accvgpr_write_b32 a2, v122
\ts_or_b64 exec, exec, s[86:87]
\tv_accvgpr_read_b32 v122, a2
\ts_endpgm
\t.section\t.rodata
"""
GOOD = BAD.replace("\tv_mov_b32_e32 v205, v193\n\tv_readlane_b32 s38, v251, 45\n\tv_accvgpr_write_b32 a2, v122\n\ts_or_b64 exec, exec, s[86:87]\n",
                   "\tv_readlane_b32 s38, v251, 45\n\ts_or_b64 exec, exec, s[86:87]\n\tv_mov_b32_e32 v205, v193\n\tv_accvgpr_write_b32 a2, v122\n")


def test_lint_recognises_the_round1_pattern():
  found = isa_lint.lint_text(BAD)
  assert [f[2] for f in found] == ["v_mov_b32_e32 v205, v193", "v_accvgpr_write_b32 a2, v122"]
  assert isa_lint.lint_text(GOOD) == []


@pytest.fixture(scope="module")
def asm():
  """The assembly of the INSTALLED build (build.build() keeps it as libsgw.s and refuses to install a flagged build)."""
  if not os.path.exists("/opt/rocm/bin/hipcc"):
    pytest.skip("hipcc not present")
  from ai_safety_gridworlds_amd import build as B
  B.build()
  if not os.path.exists(B.ASM) or os.path.getmtime(B.ASM) < max(os.path.getmtime(d) for d in B._deps()):
    B.build(force=True)
  return open(B.ASM).read()


def test_build_refuses_to_install_a_flagged_library(tmp_path, monkeypatch):
  """build.build() lints the assembly of the compile it is about to install: with a lint that reports a finding the library
  on disk is left as it was and the build raises (a rebuild on another compiler cannot ship the round-1 pattern silently)."""
  if not os.path.exists("/opt/rocm/bin/hipcc"):
    pytest.skip("hipcc not present")
  from ai_safety_gridworlds_amd import build as B, isa_lint as pkg_lint
  B.build()
  before = os.path.getmtime(B.LIB)
  small = tmp_path / "k.hip"
  small.write_text("#include <hip/hip_runtime.h>\n__global__ void k(int* p) { p[threadIdx.x] = 1; }\n")
  monkeypatch.setattr(B, "SOURCES", [str(small)])
  monkeypatch.setattr(pkg_lint, "lint_text", lambda text: [("k", 1, "v_mov_b32 v0, v1", "s_or_b64 exec, exec, s[0:1]")])
  with pytest.raises(RuntimeError, match="NOT installed"):
    B.build(force=True)
  assert os.path.getmtime(B.LIB) == before


def test_library_names_its_compiler():
  from ai_safety_gridworlds_amd import _native as N, build as B
  info = N.lib().sgw_build_info().decode()
  assert info.startswith("HIP ") and "clang" in info and info == B.compiler_version()


def test_no_kernel_saves_registers_ahead_of_an_exec_restore(asm):
  findings = isa_lint.lint_text(asm)
  assert not findings, "\n".join("%s: line %d `%s` before `%s`" % f for f in findings[:20])


def test_register_budget(asm):
  """The defect needs VGPR pressure (live-range splitting / AGPR saves at a join).  Budgets of this build
  (profiles/r02_kernel_registers.md): no step / reset kernel spills a VGPR to scratch and every fused-rollout kernel but the
  two register-bound families' stays below 64 SGPR spills (before the per-step kernarg re-read they sat at 200-630)."""
  stats = isa_lint.metadata_stats(asm)
  assert len(stats) >= 50
  assert sum(1 for k in stats if "IslandMaTILi" in k and "EEELi0E" in k) == 2      # (the budget below finds its kernels: 4- and 8-word maps)
  heavy = ("Savanna", "IslandTILb1", "IslandMa", "Firemaker")
  for k, d in stats.items():
    if "k_engine" not in k:
      continue
    if "k_engine_group" in k:                            # eleven family bodies in one kernel: the counts are sums over them
      assert d.get("vgpr_spill_count", 0) == 0 and d.get("private_segment_fixed_size", 0) == 0 and d.get("sgpr_spill_count", 0) < 11 * 64, (k, d)
      continue
    if "Li1E" not in k and "Savanna" not in k:           # step and reset kernels: nothing spilled to scratch
      assert d.get("vgpr_spill_count", 0) == 0, k
    if "IslandMaTILi4EEELi0E" in k or "IslandMaTILi8EEELi0E" in k:      # round 3: the cumulative vectors wait in LDS while the rules run --
      # the one-step round kernel of island_navigation_ex_ma fits the register file (no AGPR saves, nothing in scratch)
      assert d.get("vgpr_count", 999) <= 256 and d.get("agpr_count", 1) == 0 and d.get("private_segment_fixed_size", 1) == 0, (k, d)
    if "SavannaELi0E" in k:                              # (aintelope_savanna: 349 + 93 AGPR at the start of round 3)
      assert d.get("vgpr_count", 999) + d.get("agpr_count", 999) <= 352, (k, d)
    if "Li1E" in k and not any(h in k for h in heavy):
      assert d.get("sgpr_spill_count", 0) < 64 and d.get("private_segment_fixed_size", 0) == 0, (k, d)
