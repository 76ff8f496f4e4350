"""Diagnostic: the headline workload (island_navigation_ex L9, 65 536 envs) cut into S env-id shards, each an engine of its own
stepped on its own HIP stream (graph replay, sgw_step_n) -- does overlapping the shards' kernel boundaries beat one 1 024-wave
launch per step?

    python tools/diag/shard_streams_probe.py [total_envs] [steps] [shard counts ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B

total = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
shards = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for S in shards:
  per = total // S
  parts = [("island_navigation_ex", per, i * per) for i in range(S)]
  engines = B.build_engines(parts, dev, streams=True)
  B.fill_action_batches(engines, K, 1)
  B.run_batches(engines, K, 0, 3, True)
  torch.cuda.synchronize(dev)
  best = None
  for rep in range(3):
    t0 = time.perf_counter()
    B.run_batches(engines, K, 0, 4, True)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / (4 * K)
    best = dt if best is None else min(best, dt)
  print("shards %d x %d envs: %.3f us per step of all %d envs, %.3e env-steps/s" % (S, per, best * 1e6, S * per, S * per / best), flush=True)
  del engines
