"""Build recipe for the CPU oracle (TEST INFRASTRUCTURE ONLY; see sgw_oracle.h).

    python oracle/build.py        -> oracle/libsgw_oracle.so

-ffp-contract=off: the reference's float arithmetic is plain IEEE double ops
(CPython floats); no fused multiply-adds may be introduced.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsgw_oracle.so")
SOURCES = ["sgw_oracle.c", "sgw_oracle_ma.c", "sgw_oracle_ima.c", "sgw_oracle_sav.c"]


def build(force=False, verbose=False):
  srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
  deps = srcs + [os.path.join(HERE, "sgw_oracle.h"), os.path.join(HERE, "sgw_pcg.h")]
  if (not force and os.path.exists(LIB)
      and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in deps)):
    return LIB
  cmd = ["gcc", "-std=gnu11", "-O2", "-g0", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
         "-Wall", "-Wextra", "-Wno-unused-parameter", "-o", LIB] + srcs + ["-lm"]
  if verbose:
    print(" ".join(cmd))
  subprocess.check_call(cmd)
  return LIB


if __name__ == "__main__":
  print(build(force="--force" in sys.argv, verbose=True))
