"""Side kernels of the step path timed on their own at the BASELINE sizes: device time per launch by
HIP events over back-to-back launches on the current stream, algorithmic bytes (inputs read + outputs written, each once) and
the fraction of the HBM roofline.  `python tools/diag/side_probe.py [out.json]` (run on the GPU box)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec

HBM = 8000.0


def timed(fn, reps=200, warm=20):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3          # us


def row(name, us, nbytes, rows):
  gbs = nbytes / us / 1e3
  rows.append({"kernel": name, "us_per_launch": round(us, 2), "algorithmic_bytes": int(nbytes), "gb_per_s": round(gbs, 1),
               "frac_of_hbm_peak": round(gbs / HBM, 3)})
  print("%-44s %8.2f us  %10.1f KB  %8.1f GB/s  %.3f of HBM" % (name, us, nbytes / 1e3, gbs, gbs / HBM))


def main():
  import ctypes as C
  from ai_safety_gridworlds_amd import _native as N
  rows = []
  L_ = N.lib()

  def stream(eng):
    return eng._stream()

  def u8(*shape, dev="cuda:0"):
    return torch.empty(shape, dtype=torch.uint8, device=dev)

  def probe_family(name, n, kw, steps, rng):
    sp = make_spec(name, **kw)
    outs = ("board", "reward", "cumulative", "frame", "step_type", "agent_pos", "agent_flags")
    eng = BatchedEngine(sp, n, outputs=outs)
    if rng:
      eng.set_rng_seeds(np.arange(n))
    eng.reset()
    acts = eng.fill_actions(steps, 1)
    for t in range(steps):
      eng.step(acts[t])
    return sp, eng

  # ---- island_navigation_ex L9, 65 536 envs
  n = 65536
  sp, eng = probe_family("island_navigation_ex", n, {}, 30, False)
  HW, K, L = sp.H * sp.W, sp.K, len(sp.layer_chars)
  B = eng._bufs
  stats = torch.empty((n, 1, 5 + K), dtype=torch.float64, device=eng.device)
  karr = (C.c_int32 * N.MAX_AGENTS)(K, 0, 0, 0)
  row("k_derived_stats (island_navigation_ex, 65 536)",
      timed(lambda: L_.sgw_derived_stats(eng._h, B["reward"].data_ptr(), B["cumulative"].data_ptr(), B["frame"].data_ptr(), karr,
                                         stats.data_ptr(), stream(eng))), n * (2 * K * 8 + 4 + (5 + K) * 8), rows)
  lut = torch.from_numpy(sp.rgb_lut().reshape(-1)).to(eng.device)
  rgb = u8(n, 3, HW)
  row("k_observe: RGB (island_navigation_ex, 65 536)",
      timed(lambda: L_.sgw_observe(eng._h, B["board"].data_ptr(), lut.data_ptr(), rgb.data_ptr(), None, 0, None, stream(eng))), n * (HW + 3 * HW), rows)
  chars = torch.tensor([ord(c) for c in sp.layer_chars], dtype=torch.uint8, device=eng.device)
  lay = u8(n, L, HW)
  row("k_observe: RGB + occluded layers (island_navigation_ex, 65 536)",
      timed(lambda: L_.sgw_observe(eng._h, B["board"].data_ptr(), lut.data_ptr(), rgb.data_ptr(), chars.data_ptr(), L, lay.data_ptr(), stream(eng))),
      n * (HW + 3 * HW + L * HW), rows)
  stat = torch.from_numpy(sp.layer_static()).to(eng.device)
  gap = sp.layer_chars.index(sp.what_lies_beneath)
  row("k_observe_layers (island_navigation_ex, 65 536)",
      timed(lambda: L_.sgw_observe_layers(eng._h, B["board"].data_ptr(), chars.data_ptr(), stat.data_ptr(), L, gap, None, None, -1,
                                          lay.data_ptr(), stream(eng))), n * (HW + L * HW), rows)
  eng.close()
  # ---- firemaker_ex_ma, 16 384 envs x 3 agents
  n = 16384
  sp, eng = probe_family("firemaker_ex_ma", n, dict(amount_agents=3), 60, True)
  HW, L = sp.H * sp.W, len(sp.layer_chars)
  B = eng._bufs
  vb = int(L_.sgw_view_bytes(eng._h))
  chars = torch.tensor([ord(c) for c in sp.layer_chars], dtype=torch.uint8, device=eng.device)
  stat = torch.from_numpy(sp.layer_static()).to(eng.device)
  gap = sp.layer_chars.index(sp.what_lies_beneath)
  hid = sp.layer_chars.index(sp.hidden_layer_char)
  lay = u8(n, L, HW)
  row("k_observe_layers (firemaker_ex_ma, 16 384)",
      timed(lambda: L_.sgw_observe_layers(eng._h, B["board"].data_ptr(), chars.data_ptr(), stat.data_ptr(), L, gap, B["agent_pos"].data_ptr(),
                                          B["agent_flags"].data_ptr(), hid, lay.data_ptr(), stream(eng))), n * (HW + 6 + L * HW), rows)
  cube = u8(n, vb * L)
  row("k_agent_layer_views (firemaker_ex_ma, 16 384)",
      timed(lambda: L_.sgw_agent_layer_views(eng._h, lay.data_ptr(), B["agent_pos"].data_ptr(), None, chars.data_ptr(), L, ord('#'),
                                             cube.data_ptr(), stream(eng)), reps=50, warm=5), n * (L * HW + 6 + L * vb), rows)
  vbuf = u8(n, vb)
  row("k_agent_views (firemaker_ex_ma, 16 384)",
      timed(lambda: L_.sgw_agent_views(eng._h, B["board"].data_ptr(), B["agent_pos"].data_ptr(), None, ord('#'), vbuf.data_ptr(), stream(eng))),
      n * (HW + 6 + vb), rows)
  stats = torch.empty((n, 3, 5 + 3), dtype=torch.float64, device=eng.device)
  karr = (C.c_int32 * N.MAX_AGENTS)(2, 2, 3, 0)
  row("k_derived_stats (firemaker_ex_ma, 16 384 x 3)",
      timed(lambda: L_.sgw_derived_stats(eng._h, B["reward"].data_ptr(), B["cumulative"].data_ptr(), B["frame"].data_ptr(), karr,
                                         stats.data_ptr(), stream(eng))), n * 3 * (2 * 3 * 8 + (5 + 3) * 8) + n * 4, rows)
  eng.close()
  # ---- aintelope_savanna: layers from the state bitmaps
  kw = dict(amount_agents=2, amount_predators=2, amount_water_tiles=3, amount_gold_deposits=2, amount_silver_deposits=2,
            amount_small_food_patches=2, amount_drink_holes=2, amount_small_drink_holes=1)
  n = 65536
  sp, eng = probe_family("aintelope_savanna", n, kw, 10, True)
  HW, L = sp.H * sp.W, len(sp.layer_chars)
  chars = torch.tensor([ord(c) for c in sp.layer_chars], dtype=torch.uint8, device=eng.device)
  lay = u8(n, L, HW)
  row("k_savanna_layers (aintelope_savanna, 65 536)",
      timed(lambda: L_.sgw_state_layers(eng._h, chars.data_ptr(), L, 1, lay.data_ptr(), stream(eng)), reps=50, warm=5), n * (9 * 24 + L * HW), rows)
  eng.close()
  if len(sys.argv) > 1:
    json.dump({"rows": rows, "hbm_peak_gbs": HBM}, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
  main()
