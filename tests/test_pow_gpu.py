"""Device pow (glibc's algorithm and tables, csrc/sgw_pow.hpp) against the host C library's pow() -- the arithmetic of the
reference's math.pow -- bit for bit, through the C ABI."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from ai_safety_gridworlds_amd import _native as N

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("y", [1.1, 0.7, 2.3])
def test_device_pow_matches_libm(y):
  rng = np.random.default_rng(11)
  x = np.concatenate([1.0 + 60.0 * rng.random(4_000_000), np.arange(2, 122) * 0.5])
  libm = C.CDLL("libm.so.6"); libm.pow.restype = C.c_double; libm.pow.argtypes = [C.c_double, C.c_double]
  xd = torch.from_numpy(x).to("cuda:0")
  out = torch.empty_like(xd)
  N.check(N.lib().sgw_pow_f64(xd.data_ptr(), float(y), out.data_ptr(), xd.numel(), 0, None), "sgw_pow_f64")
  torch.cuda.synchronize()
  got = out.cpu().numpy()
  # numpy's float64 power loop calls the same libm pow for scalars only; use libm directly on a sample and math.pow on all
  want = np.array([math.pow(v, y) for v in x[:300000]] + [math.pow(v, y) for v in x[-120:]])
  sel = np.concatenate([np.arange(300000), np.arange(len(x) - 120, len(x))])
  assert (got[sel].view(np.uint64) == want.view(np.uint64)).all()
  assert math.pow(3.3, y) == libm.pow(3.3, y)
