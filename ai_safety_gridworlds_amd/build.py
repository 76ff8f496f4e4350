"""Build recipe for libsgw.so (the HIP engine behind include/sgw.h), gfx950 only.

    python -m ai_safety_gridworlds_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is built IN-TREE next to this file so that it
travels with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
-ffp-contract=off: the reference's float arithmetic is plain IEEE-754 double add/mul (CPython);
bit-exact parity forbids the compiler from fusing them into FMAs.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsgw.so")
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


def build(force=False, verbose=False, extra_flags=()):
  if (not force and os.path.exists(LIB)
      and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in _deps())):
    return LIB
  if not os.path.exists(HIPCC):
    raise RuntimeError("hipcc not found at %s: libsgw.so cannot be built (no CPU fallback exists)" % HIPCC)
  # several ranks of one node may get here at once (torchrun): one compiles, the others wait for it and reuse the
  # result; the library appears atomically
  import fcntl
  with open(LIB + ".lock", "w") as lock:
    fcntl.flock(lock, fcntl.LOCK_EX)
    if (not force and os.path.exists(LIB)
        and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in _deps())):
      return LIB
    tmp = "%s.tmp.%d" % (LIB, os.getpid())
    cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
      print(" ".join(cmd).replace(tmp, LIB))
    try:
      subprocess.check_call(cmd)
      os.replace(tmp, LIB)
    finally:
      if os.path.exists(tmp):
        os.remove(tmp)
  return LIB


if __name__ == "__main__":
  print(build(force="--force" in sys.argv, verbose=True))
