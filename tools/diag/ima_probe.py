import sys
import numpy as np, torch
sys.path.insert(0, ".")
from ai_safety_gridworlds_amd.engine import BatchedEngine
from ai_safety_gridworlds_amd.specs import make_spec
spec = make_spec("island_navigation_ex_ma", map_randomization_frequency=3, max_iterations=30)
n, T, seed = 3000, 4, 99
outs = ("board", "reward", "step_type", "frame", "agent_pos", "agent_flags")
def fresh():
  e = BatchedEngine(spec, n, outputs=outs); e.set_rng_seeds(np.arange(n) + 3); e.reset(); return e
a, b = fresh(), fresh()
acts = a.fill_actions(T, seed)
e0 = 7
print("actions env7", acts[:, e0].tolist())
o0 = a._bufs
per = {k: [] for k in outs}
for t in range(T):
  o = a.step(acts[t])
  for k in outs: per[k].append(o[k].clone())
ro = b.rollout(T, seed, write_every=True)
for t in range(T):
  for nm, src in (("step", {k: torch.stack(per[k]) for k in outs}), ("roll", ro)):
    bd = src["board"][t, e0].cpu().numpy().reshape(6, 8)
    print(t, nm, "pos", src["agent_pos"][t, e0].reshape(-1).tolist(), "flags", src["agent_flags"][t, e0].tolist(), "st", src["step_type"][t, e0].tolist(), "rew0", src["reward"][t, e0].reshape(2, -1)[0].tolist())
    print("   ", " | ".join("".join(chr(c) for c in r) for r in bd))
sa = a.get_state()[:, e0].cpu().numpy().view(np.uint64); sb = b.get_state()[:, e0].cpu().numpy().view(np.uint64)
print("state words differ at", [i for i in range(len(sa)) if sa[i] != sb[i]])
for nm, s in (("step", sa), ("roll", sb)):
  w0, w1 = int(s[0]), int(s[1])
  print(nm, "map_cached", (w0 >> 45) & 1, "episode_no", (w1 >> 32) & 0xffff, "map_episode", (w1 >> 48) & 0xffff, "step_type", (w0 >> 32) & 15, "ast", (w0 >> 16) & 7, (w0 >> 19) & 7)
