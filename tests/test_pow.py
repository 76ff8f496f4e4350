"""The device's pow (csrc/sgw_pow.hpp: glibc's algorithm and tables) compiled for the host and compared with the C
library's pow() -- the arithmetic behind the reference's math.pow -- on 10^7 inputs of the regrowth domain.  Also checks
that the committed tables are the ones the running libm holds (tools/gen_pow_tables.py regenerates them)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def test_restated_pow_matches_libm_bit_for_bit(tmp_path):
  exe = str(tmp_path / "pow_check")
  subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(HERE, "pow_check.c"), "-lm"])
  out = subprocess.run([exe, "10000000"], capture_output=True, text=True, check=True)
  assert int(out.stdout.strip()) == 0, out.stderr


def test_committed_tables_match_the_running_libm(tmp_path):
  inc = os.path.join(REPO, "ai_safety_gridworlds_amd", "csrc", "sgw_pow_tables.inc")
  before = open(inc).read()
  fresh = str(tmp_path / "tables.inc")
  subprocess.check_call([sys.executable, os.path.join(REPO, "tools", "gen_pow_tables.py"), "--out=" + fresh], stdout=subprocess.DEVNULL)
  after = open(fresh).read()
  strip = lambda t: "\n".join(l for l in t.splitlines() if not l.startswith("//"))
  assert strip(before) == strip(after)


def test_python_math_pow_is_libm_pow():
  import ctypes, math, random
  libm = ctypes.CDLL("libm.so.6"); libm.pow.restype = ctypes.c_double; libm.pow.argtypes = [ctypes.c_double, ctypes.c_double]
  rnd = random.Random(5)
  for _ in range(20000):
    x = 1.0 + 60.0 * rnd.random()
    assert math.pow(x, 1.1) == libm.pow(x, 1.1)
