#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched safety-gridworld step path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the config the metric is quoted on): 65 536 lockstep
island_navigation_ex envs (level 9, default flags) PER GPU, synthetic uniform actions from
Philox-4x32-10 keyed by the global env id.  One bench "step" = one pass of the hot path over the
batch = ONE sgw_step kernel launch advancing all 65 536 envs by one env.step() (auto-reset
included), writing board + reward vector + step_type + term_reason + safety + frame.  The K
action batches are resident in HBM before the timed region starts; the K launches are issued
back to back by sgw_step_n (host loop in C).  value = envs x K x N_gpus / max-over-ranks time.
Envs are sharded by contiguous global-id ranges; the only collective is one all-reduce (RCCL)
of the episodic-return accumulators after the batch ("scaling": "weak").

Also reported (same JSON line): the fused-rollout mode (K steps in ONE launch, state in
registers, outputs written every step), the roofline of the step kernel, and a CPU baseline =
the C oracle (oracle/, "port" of the reference semantics, bit-identical outputs) on a bounded
sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from ai_safety_gridworlds_amd import parallel                   # noqa: E402
from ai_safety_gridworlds_amd.engine import BatchedEngine      # noqa: E402
from ai_safety_gridworlds_amd.specs import make_spec            # noqa: E402

SEED = 0x5AFE
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# algorithmic bytes per env-step, island_navigation_ex L9 (step-per-launch mode):
# action 1 + state 80 read + 80 write + board 48 + reward 80 + done 1 + term 1 + safety/hidden 8
B_ALG_STEP = 299
B_ALG_FUSED = 139                # fused rollout: the two state terms drop out
OUTPUTS = ("board", "reward", "step_type", "term_reason", "safety", "frame")


def cpu_baseline(n_envs, seconds, threads):
  """Time the C oracle on a bounded sample of the same workload (rank 0, N=1 only)."""
  from oracle import oracle as O
  from ai_safety_gridworlds_amd import philox
  cfg = O.make_config("island_navigation_ex", level=9)
  E, T = min(n_envs, 8192), 500
  acts = philox.actions(SEED, np.arange(E), np.arange(T), 0, 5).T.copy()      # [E, T], reused by every repeat
  def run(repeats, nthreads):
    t0 = time.perf_counter()
    for _ in range(repeats):
      O.run_streams(cfg, acts, fields=["step_type"], nthreads=nthreads)
    return repeats * E * T / (time.perf_counter() - t0)
  rate1 = run(1, 1)                                        # single thread, ~2 s
  est = run(1, threads)                                    # calibrate the all-core rate
  repeats = int(max(1, min(200, est * seconds / (E * T))))
  rate = run(repeats, threads)
  return {"value": rate, "unit": "env-steps/s", "cores": threads, "kind": "port",
          "sample": "%d x (%d island_navigation_ex L9 envs x %d steps), same Philox action stream, C oracle "
                    "(oracle/sgw_oracle.c), OpenMP over envs on %d threads; 1 thread: %.0f env-steps/s"
                    % (repeats, E, T, threads, rate1),
          "single_thread_value": rate1}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=2000)
  ap.add_argument("--warmup", type=int, default=200)
  ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
  ap.add_argument("--cpu-seconds", type=float, default=12.0)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-fused", action="store_true")
  a = ap.parse_args()

  rank, local_rank, world = parallel.world()
  if world != a.gpus:
    if world == 1 and a.gpus > 1:
      sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % a.gpus)
    a.gpus = world
  dist = parallel.init("nccl") if world > 1 else None
  device = torch.device("cuda", local_rank)
  torch.cuda.set_device(device)

  K, W, n = a.steps, a.warmup, a.envs
  spec = make_spec("island_navigation_ex")            # level 9, default flags
  eng = BatchedEngine(spec, n, device=device, env_id_base=rank * n, outputs=OUTPUTS)
  eng.reset()
  acts = eng.fill_actions(W + K, SEED)                # [W+K, n] int8, resident in HBM
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

  def barrier():
    torch.cuda.synchronize(device)
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize(device)

  if W > 0:
    eng.step_n(acts[:W])
  barrier()
  t0 = time.perf_counter()
  ev0.record()
  eng.step_n(acts[W:], accumulate=True)               # K launches, one per step
  ev1.record()
  torch.cuda.synchronize(device)
  elapsed = time.perf_counter() - t0
  barrier()
  kernel_ms = ev0.elapsed_time(ev1) / K               # avg launch duration on the launch stream
  accum = eng.read_returns()                          # [K+1] per-GPU (sum of episode returns, #episodes)
  elapsed = parallel.max_over_ranks(elapsed, device, dist)
  parallel.allreduce_returns(accum, dist)             # the one collective: episodic returns, end of batch
  acc = accum.cpu().numpy()

  fused = None
  if not a.no_fused:
    eng2 = BatchedEngine(spec, n, device=device, env_id_base=rank * n, outputs=OUTPUTS)
    eng2.reset()
    Tf = min(K, 512)
    eng2.rollout(min(W, 64) or 1, SEED, step0=0, write_every=True)
    barrier()
    f0 = time.perf_counter()
    ev0.record()
    eng2.rollout(Tf, SEED, step0=W, write_every=True)
    ev1.record()
    torch.cuda.synchronize(device)
    felapsed = time.perf_counter() - f0
    barrier()
    fms = ev0.elapsed_time(ev1)
    felapsed = parallel.max_over_ranks(felapsed, device, dist)
    fused = {"value": world * n * Tf / felapsed, "unit": "env-steps/s", "steps_per_launch": Tf,
             "ms_per_step": fms / Tf, "bytes_per_env_step": B_ALG_FUSED,
             "hbm_gbs": n * B_ALG_FUSED / (fms / Tf * 1e-3) / 1e9,
             "note": "ONE launch advances every env Tf steps (state in registers, in-kernel Philox "
                     "actions), board/reward/step_type/term_reason/safety/frame written every step"}
    eng2.close()

  if rank == 0:
    achieved = n * B_ALG_STEP / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(REPO, "profiles", "r01_traffic.json")
    if os.path.exists(tpath):
      try:
        traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
      except Exception:
        traffic = None
    line = {
        "metric": "env-steps/sec (whole node), 65 536 batched envs per GPU",
        "value": world * n * K / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "island_navigation_ex level 9 default flags, %d envs/GPU, one sgw_step launch "
                               "per step, uniform Philox actions resident in HBM" % n,
                   "envs_per_gpu": n, "reward_dims": spec.K, "board": "%dx%d" % (spec.H, spec.W),
                   "outputs": list(OUTPUTS), "sharding": "env-id ranges, dp%d" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "sgw::k_engine<sgw::Island>", "avg_launch_us": kernel_ms * 1e3,
                     "algorithmic_bytes_per_env_step": B_ALG_STEP, "env_steps_per_launch": n},
        "episodes_finished": float(acc[spec.K]),
        "mean_episode_return": (acc[:spec.K] / max(acc[spec.K], 1.0)).tolist(),
        "reward_dim_names": spec.dim_names,
    }
    if fused is not None:
      line["fused_rollout"] = fused
    if world == 1 and not a.no_cpu_baseline:
      threads = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
      threads = max(1, min(threads, 64))
      line["cpu_baseline"] = cpu_baseline(n, a.cpu_seconds, threads)
    print(json.dumps(line))
  eng.close()
  if dist is not None:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
