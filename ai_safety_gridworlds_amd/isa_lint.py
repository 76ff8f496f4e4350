"""ISA lint for libsgw's gfx950 code object: lanes must be re-enabled BEFORE anything is saved for them.

Run by `build.build()` on the assembly of EVERY build (the library is not installed when a kernel is flagged) and by
tools/isa_lint.py / tests/test_isa_lint.py.  Background (DESIGN.md "The fused-kernel fault: cause"): at a control-flow join the
backend re-enables the lanes that skipped a divergent region with `s_or_b64 exec, exec, s[a:b]` (SI_END_CF), which must come
first in the join block.  Under VGPR pressure ROCm 7.2's register allocator splits live ranges / spills VGPRs to AGPRs exactly
there, and `SIInstrInfo::isBasicBlockPrologue` gives up at the first COPY in the block: a loop-carried value silently turns to
garbage for the lanes that skipped the region, or -- when it is an address -- the wave faults.  The defect is deterministic per
build and depends only on register pressure, so every build is checked: within one basic block that is the target of an
`s_cbranch_execz`, no instruction whose effect depends on EXEC (VALU other than v_readlane / v_writelane / v_readfirstlane,
VMEM, LDS, scratch) may precede the `s_or_b64 exec, exec, ...` / `s_mov_b64 exec, ...` that widens EXEC."""
import re

EXEC_FREE = ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32", "v_nop")
LANE_PREFIXES = ("v_", "global_", "flat_", "scratch_", "buffer_", "ds_", "image_")
WIDEN = re.compile(r"^\s*(s_or_b64\s+exec,\s*exec,|s_or_saveexec_b64\s|s_mov_b64\s+exec,)")
KERNEL_LABEL = re.compile(r"^(_Z\w+):")
BLOCK_START = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)")
TERMINATOR = re.compile(r"^\s*(s_cbranch_\w+|s_branch|s_endpgm|s_setpc_b64)\b")


EXEC_WRITE = re.compile(r"^\s*s_\w+\s+exec\b|^\s*s_\w*saveexec\w*\s")


def lint_text(text):
  """-> list of (kernel, line number, offending instruction, exec-widening instruction).

  A label that is the target of an `s_cbranch_execz` is a JOIN: the wave arrives there with EXEC == 0 when every lane skipped
  the region (and with the region's lanes only when it falls through from the region's body).  Whatever lane-dependent
  instruction sits between such a label and the block's first write to EXEC ran for the wrong lane set when that first
  write widens the mask (`s_or_b64 exec, exec, saved`)."""
  lines = text.splitlines()
  joins, kernel = set(), None
  for line in lines:
    m = KERNEL_LABEL.match(line)
    if m:
      kernel = m.group(1)
    m = re.match(r"^\s*s_cbranch_execz\s+(\.LBB\d+_\d+)", line)
    if m:
      joins.add((kernel, m.group(1)))
  findings = []
  kernel, pending, in_kernel, armed = None, [], False, False
  for no, line in enumerate(lines, 1):
    m = KERNEL_LABEL.match(line)
    if m:
      kernel, pending, in_kernel, armed = m.group(1), [], True, False
      continue
    if not in_kernel:
      continue
    if line.startswith("\t.section") or ".end_amdhsa_kernel" in line:
      in_kernel = False
      continue
    m = BLOCK_START.match(line)
    if m or TERMINATOR.match(line):
      pending = []
      armed = bool(m) and (kernel, m.group(1).rstrip(":")) in joins
      continue
    code = line.split(";")[0].strip()
    if not code or code.startswith("."):
      continue
    op = code.split()[0]
    if EXEC_WRITE.match(line):
      if WIDEN.match(line) and armed:
        for pno, pcode in pending:
          findings.append((kernel, pno, pcode, code))
      pending, armed = [], False
      continue
    if armed and op.startswith(LANE_PREFIXES) and op not in EXEC_FREE:
      pending.append((no, code))
  return findings


def kernel_stats(text):
  """Per-kernel metadata from the assembly's .amdhsa / remark comments."""
  out = {}
  for m in re.finditer(r"; Function info:.*?\n(.*?)(?=\n\t\.text|\n\t\.section|\Z)", text, flags=re.S):
    pass
  cur = None
  for line in text.splitlines():
    m = KERNEL_LABEL.match(line)
    if m:
      cur = m.group(1)
    m = re.match(r"; (NumSgprs|NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|SGPRSpill|VGPRSpill): (\d+)", line.strip())
    if m and cur:
      out.setdefault(cur, {})[m.group(1)] = int(m.group(2))
  # spill counts are in the metadata note
  for m in re.finditer(r"\.name:\s+(\S+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", text, flags=re.S):
    pass
  return out


def metadata_stats(text):
  """{kernel: {sgpr_spill_count, vgpr_spill_count, vgpr_count, agpr_count, sgpr_count, private_segment_fixed_size}} from the
  amdhsa.kernels YAML note at the end of the assembly."""
  out = {}
  note = text[text.rfind("amdhsa.kernels:"):] if "amdhsa.kernels:" in text else ""
  for block in note.split("  - .agpr_count:")[1:]:
    block = ".agpr_count:" + block
    d = {}
    for key in ("agpr_count", "vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size",
                "group_segment_fixed_size"):
      m = re.search(r"\.%s:\s+(\d+)" % key, block)
      if m:
        d[key] = int(m.group(1))
    m = re.search(r"\.name:\s+(\S+)", block)
    if m:
      out[m.group(1)] = d
  return out


