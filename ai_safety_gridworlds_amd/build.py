"""Build recipe for libsgw.so (the HIP engine behind include/sgw.h), gfx950 only.

    python -m ai_safety_gridworlds_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is built IN-TREE next to this file so that it
travels with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
-ffp-contract=off: the reference's float arithmetic is plain IEEE-754 double add/mul (CPython);
bit-exact parity forbids the compiler from fusing them into FMAs.

Every build is LINTED before it is installed: the compile keeps its gfx950 assembly (-save-temps), `isa_lint.lint_text`
looks for the register-allocator defect that corrupted a fused kernel in round 1 (VGPR saves placed ahead of the
`s_or_b64 exec` of a control-flow join), and a flagged build raises instead of replacing libsgw.so -- an
automatic rebuild on another box / compiler cannot ship the broken pattern unnoticed.  The compiler's version is compiled into
the library (`sgw_build_info()`); the assembly of the installed build stays beside it as libsgw.s (git-ignored) for the
register tables of tools/isa_lint.py.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsgw.so")
ASM = os.path.join(HERE, "libsgw.s")
SOURCES = ["sgw_api.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical",
         # k_engine's leading scalar arguments (state / tables / actions pointers, sizes) arrive preloaded in SGPRs at wave
         # launch, so the prologue's global loads do not wait for a scalar load of the kernarg segment first
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _deps():
  out = [os.path.join(HERE, "..", "include", "sgw.h")]
  for f in os.listdir(CSRC):
    if f.endswith((".hip", ".hpp", ".h", ".inc")):
      out.append(os.path.join(CSRC, f))
  return out


def compiler_version():
  """One line naming the compiler that builds the library (compiled in as SGW_BUILD_COMPILER)."""
  try:
    out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True, check=True).stdout.splitlines()
    hip = next((l.split(":", 1)[1].strip() for l in out if l.startswith("HIP version")), "?")
    clang = next((l.strip() for l in out if "clang version" in l), "?")
    return ("HIP %s; %s" % (hip, clang)).replace('"', "'")[:200]
  except Exception:
    return "unknown"


def _up_to_date():
  return os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in _deps())


def build(force=False, verbose=False, extra_flags=(), lint=True):
  if not force and _up_to_date():
    return LIB
  if not os.path.exists(HIPCC):
    raise RuntimeError("hipcc not found at %s: libsgw.so cannot be built (no CPU fallback exists)" % HIPCC)
  # several ranks of one node may get here at once (torchrun): one compiles, the others wait for it and reuse the
  # result; the library appears atomically
  import fcntl
  with open(LIB + ".lock", "w") as lock:
    fcntl.flock(lock, fcntl.LOCK_EX)
    if not force and _up_to_date():
      return LIB
    work = tempfile.mkdtemp(prefix="sgw_build_")
    try:
      tmp = os.path.join(work, "libsgw.so")
      cmd = ([HIPCC] + FLAGS + list(extra_flags) + ['-DSGW_BUILD_COMPILER="%s"' % compiler_version()] +
             (["-save-temps=obj"] if lint else []) + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES])
      if verbose:
        print(" ".join(cmd).replace(tmp, LIB))
      subprocess.check_call(cmd, cwd=work)
      if lint:
        from . import isa_lint
        asm = [os.path.join(work, f) for f in os.listdir(work) if f.endswith(".s") and "amdgcn" in f]
        if not asm:
          raise RuntimeError("build: the compile left no gfx950 assembly to lint (-save-temps=obj)")
        findings = []
        for f in asm:
          findings += isa_lint.lint_text(open(f).read())
        if findings:
          raise RuntimeError("build: libsgw.so NOT installed -- %d instruction(s) execute ahead of an exec-mask restore at a "
                             "control-flow join (the round-1 miscompile, tools/isa_lint.py), first: %s line %d `%s` before `%s`"
                             % ((len(findings),) + tuple(findings[0])))
        shutil.copyfile(asm[0], ASM + ".tmp")
        os.replace(ASM + ".tmp", ASM)
        if verbose:
          print("isa lint: clean (%d assembly file(s))" % len(asm))
      os.replace(tmp, LIB)
    finally:
      shutil.rmtree(work, ignore_errors=True)
  return LIB


if __name__ == "__main__":
  print(build(force="--force" in sys.argv, verbose=True))
