#!/usr/bin/env python3
"""ISA lint for libsgw's gfx950 code object: lanes must be re-enabled BEFORE anything is saved for them.

    python tools/isa_lint.py [--build] [file.s ...]          (exit code 1 when a kernel is flagged)

Background (DESIGN.md "The fused-kernel fault: cause"): at a control-flow join the backend re-enables the lanes that skipped
a divergent region with `s_or_b64 exec, exec, s[a:b]` (SI_END_CF), which must come first in the join block.  Under VGPR
pressure ROCm 7.2's register allocator splits live ranges / spills VGPRs to AGPRs exactly there, and
`SIInstrInfo::isBasicBlockPrologue` (the function that keeps such code behind the exec restore) gives up at the first COPY in
the block -- its source carries a FIXME saying so.  The result is

    .LBB28_70:                          ; join block of `while (pend) { pow }`
        v_mov_b32  v205, v193           ; live-range-split copy       } executed ONLY by the lanes that were inside the
        v_accvgpr_write_b32 a2, v122    ; VGPR -> AGPR spill          } region (no lane at all when the wave skipped it)
        s_or_b64   exec, exec, s[86:87] ; the other lanes come back here
        ...                             ; v122 reused as a temporary by every lane
        v_accvgpr_read_b32 v122, a2     ; every lane reloads: the lanes that skipped the region get a STALE a2

i.e. a loop-carried value (episode_no in k_engine<IslandMa, K_ROLLOUT>) silently turns to garbage, or -- when the value is
an address -- the wave faults.  The bug is in the compiler, it is deterministic per build and depends only on register
pressure, so the guard is this lint, run over every build (tests/test_isa_lint.py): within one basic block, no instruction
whose effect depends on EXEC (VALU other than v_readlane / v_writelane / v_readfirstlane, VMEM, LDS, scratch) may precede an
`s_or_b64 exec, exec, ...` / `s_mov_b64 exec, ...` that widens EXEC.
Also reports, per kernel, the register / spill / scratch figures from the code object's metadata.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "ai_safety_gridworlds_amd", "csrc")
sys.path.insert(0, REPO)
from ai_safety_gridworlds_amd.isa_lint import lint_text, kernel_stats, metadata_stats      # noqa: E402,F401  (the checks themselves live in the package: build.build() runs them on every build)


def demangle(names):
  try:
    p = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, p.stdout.splitlines()))
  except Exception:
    return {n: n for n in names}


def build_asm(extra_flags=()):
  """Compile csrc/sgw_api.hip for gfx950 to assembly with the library's own flags (device side only)."""
  sys.path.insert(0, REPO)
  from ai_safety_gridworlds_amd import build as B
  tmp = tempfile.mkdtemp(prefix="sgw_isa_")
  out = os.path.join(tmp, "sgw_api.s")
  flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
  cmd = [B.HIPCC] + flags + list(extra_flags) + ["--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, "sgw_api.hip")]
  subprocess.check_call(cmd)
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("files", nargs="*")
  ap.add_argument("--build", action="store_true", help="compile csrc/sgw_api.hip to assembly first and lint that")
  ap.add_argument("--table", default=None, help="write the per-kernel register / spill table (markdown) here")
  a = ap.parse_args()
  files = list(a.files)
  if a.build or not files:
    files.append(build_asm())
  rc = 0
  for f in files:
    text = open(f).read()
    findings = lint_text(text)
    stats = metadata_stats(text)
    names = demangle(sorted(stats))
    if a.table:
      with open(a.table, "w") as t:
        t.write("| kernel | VGPR | AGPR | SGPR | SGPR spills | VGPR spills | scratch B | exec-restore lint |\n|---|---|---|---|---|---|---|---|\n")
        bad = {k for k, _, _, _ in findings}
        for k in sorted(stats, key=lambda k: names[k]):
          d = stats[k]
          t.write("| `%s` | %d | %d | %d | %d | %d | %d | %s |\n" % (
              names[k].replace("sgw::", "").replace("(sgw::KArgs)", ""), d.get("vgpr_count", 0), d.get("agpr_count", 0), d.get("sgpr_count", 0),
              d.get("sgpr_spill_count", 0), d.get("vgpr_spill_count", 0), d.get("private_segment_fixed_size", 0),
              "**FLAGGED**" if k in bad else "clean"))
    print("%s: %d kernels, %d finding(s)" % (f, len(stats), len(findings)))
    dn = demangle(sorted({k for k, _, _, _ in findings}))
    for k, no, code, widen in findings:
      rc = 1
      print("  %s: line %d `%s` executes before `%s` in the same block" % (dn[k], no, code, widen))
  return rc


if __name__ == "__main__":
  sys.exit(main())
