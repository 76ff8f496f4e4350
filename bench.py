#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched safety-gridworld step path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the config the metric is quoted on): 65 536 lockstep
island_navigation_ex envs (level 9, default flags) PER GPU, synthetic uniform actions from
Philox-4x32-10 keyed by the global env id.  One bench "step" = one pass of the hot path over the
batch = ONE sgw_step kernel launch advancing all 65 536 envs by one env.step() (auto-reset
included), writing board + reward vector + step_type + term_reason + safety + frame.  The
action batches are resident in HBM before the timed region starts; the K launches of a batch are
issued back to back by sgw_step_n (host loop in C).  A K-step batch at this size lasts K x ~9 us, so
the timed region REPEATS the K-step batch R times (R chosen after warmup so that the region lasts
>= --min-seconds, default 2 s; "repeats" / "timed_steps" in the JSON line; --min-seconds 0 times
exactly K steps): with the driver's `--steps 20` alone the region would be 0.2 ms of event / sync
overhead and clock ramp.  value = envs x K x R x N_gpus / max-over-ranks time.
Envs are sharded by contiguous global-id ranges; the only collective is one all-reduce (RCCL)
of the episodic-return accumulators after the batch ("scaling": "weak").

Also reported (same JSON line): the fused-rollout mode (512 steps in ONE launch, state in
registers, outputs written every step), the roofline of the step kernel, and a CPU baseline =
the C oracle (oracle/, "port" of the reference semantics, bit-identical outputs) on a bounded
sample of the same workload on this box's host cores, with the reference's own CPython rate
(measured in the build container, profiles/reference_cpython.json) beside it.
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
from ai_safety_gridworlds_amd.engine import BatchedEngine, EngineGroup, fused_views      # noqa: E402
from ai_safety_gridworlds_amd.specs import make_spec            # noqa: E402

SEED = 0x5AFE
METRIC = json.load(open(os.path.join(REPO, "BASELINE.json")))["metric"]      # BASELINE.json's metric string, verbatim
TRAFFIC_FILES = ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json")  # newest first
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_MEASURED_COPY_GBS = 6290.0   # MI355X_MICROARCH.md: the guide's measured device-to-device copy rate (reported beside the spec peak)
# MI355X_MICROARCH.md "SIMD": a wave64 VALU instruction occupies its SIMD for 2 cycles (32 lanes/cycle); 256 CUs x 4 SIMDs at
# the 2.4 GHz maximum clock -> 1 228.8 G wave-instructions/s (f64 add/mul cost more than one slot: the real ceiling is lower)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0
# families whose round kernel is bound by instruction issue, not by HBM: SQ_INSTS_VALU per launch comes from this round's PMC pass
VALU_BOUND = {"firemaker_ex_ma": ("r03_pmc_firemaker_ex_ma.json", "r02_pmc_firemaker_ex_ma.json"),
              "aintelope_savanna": ("r03_pmc_aintelope_savanna.json", "r02_pmc_aintelope_savanna.json")}     # newest first
# algorithmic bytes per env-step, island_navigation_ex L9 (step-per-launch mode):
# action 1 + state 80 read + 80 write + board 48 + reward 80 + done 1 + term 1 + safety/hidden 8
# (per-workload figures live in WORKLOADS below; fused rollout: the two state terms drop out)


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


def reference_cpython():
  """The reference itself (CPython object graph, one env) cannot travel to the GPU box; it is timed in the build container by
  tools/time_reference.py, which writes profiles/reference_cpython.json.  Reported beside the C port, never measured here."""
  path = os.path.join(REPO, "profiles", "reference_cpython.json")
  try:
    d = json.load(open(path))
    return {"value": d["island_navigation_ex"]["steps_per_s"], "unit": "env-steps/s", "cores": 1,
            "where": d["where"], "source": "profiles/reference_cpython.json (tools/time_reference.py)"}
  except Exception:
    return None


# SURVEY.md §8(d) algorithmic bytes per env-step: (step-per-launch, fused rollout)
WORKLOADS = {
    "island_navigation_ex": dict(kwargs={}, envs=65536, b_step=299, b_fused=139,
                                 outputs=("board", "reward", "step_type", "term_reason", "safety", "frame")),
    "boat_race_ex": dict(kwargs=dict(level=3), envs=65536, b_step=318, b_fused=108,
                         outputs=("board", "reward", "step_type", "term_reason", "frame")),
    "safe_interruptibility": dict(kwargs=dict(level=1), envs=65536, b_step=107, b_fused=75,
                                  outputs=("board", "reward", "step_type", "term_reason", "hidden")),
    "boat_race": dict(kwargs=dict(level=0), envs=65536, b_step=76, b_fused=44,
                      outputs=("board", "reward", "step_type", "term_reason", "hidden")),
    # one env-step = one ROUND (3 agents act); agent-centric views are produced by a separate kernel and are
    # not part of this loop: 1815 B/round of SURVEY minus the 1139 view bytes
    "firemaker_ex_ma": dict(kwargs=dict(amount_agents=3), envs=16384, b_step=676, b_fused=354,
                            outputs=("board", "reward", "step_type", "term_reason", "agent_pos")),
    # one env-step = one ROUND (the agents that are still alive act once); 38-word state (per-env 4-bit map, PCG64,
    # 2 x K cumulative): 2 + 2*304 + 48 + 128 + 2 + 2 + 8
    "island_navigation_ex_ma": dict(kwargs={}, envs=65536, b_step=798, b_fused=190,
                                    outputs=("board", "reward", "step_type", "term_reason", "safety")),
    # original-suite families: 1 action + state read + state write + board + reward 8 + step_type 1 + term 1 + hidden 8
    "island_navigation": dict(kwargs={}, envs=65536, b_step=131, b_fused=67, outputs=("board", "reward", "step_type", "term_reason", "hidden")),
    "distributional_shift": dict(kwargs=dict(is_testing=True), envs=65536, b_step=146, b_fused=82,
                                 outputs=("board", "reward", "step_type", "term_reason", "hidden")),
    "absent_supervisor": dict(kwargs={}, envs=65536, b_step=131, b_fused=67, outputs=("board", "reward", "step_type", "term_reason", "hidden")),
    "side_effects_sokoban": dict(kwargs=dict(level=1), envs=65536, b_step=183, b_fused=119,
                                 outputs=("board", "reward", "step_type", "term_reason", "hidden")),
    "conveyor_belt": dict(kwargs=dict(variant="sushi_goal"), envs=65536, b_step=132, b_fused=68,
                          outputs=("board", "reward", "step_type", "term_reason", "hidden")),
    "rocks_diamonds": dict(kwargs={}, envs=65536, b_step=146, b_fused=82, outputs=("board", "reward", "step_type", "term_reason", "hidden")),
    "tomato_watering": dict(kwargs={}, envs=65536, b_step=146, b_fused=82, outputs=("board", "reward", "step_type", "term_reason", "hidden")),
    "friend_foe": dict(kwargs={}, envs=65536, b_step=193, b_fused=49, outputs=("board", "reward", "step_type", "term_reason", "hidden")),
    "whisky_gold": dict(kwargs=dict(human_player=True), envs=65536, b_step=131, b_fused=67,
                        outputs=("board", "reward", "step_type", "term_reason", "hidden")),
    # one env-step = one ROUND of two agents on the 13x13 savanna; 68 of the 84 state words move per step (the cached
    # initial layers only when an episode begins): 2 + 2*544 + 169 + 2*11*8 + 2 + 2 + 8
    "aintelope_savanna": dict(kwargs=dict(amount_agents=2, amount_predators=2, amount_water_tiles=3, amount_gold_deposits=2,
                                          amount_silver_deposits=2, amount_small_food_patches=2, amount_drink_holes=2,
                                          amount_small_drink_holes=1, sustainability_challenge=True, penalise_oversatiation=True),
                              envs=65536, b_step=1447, b_fused=357, outputs=("board", "reward", "step_type", "term_reason", "safety")),
}
MIXED = ("island_navigation_ex", "boat_race_ex", "safe_interruptibility")    # BASELINE.json configs[4]


def mixed_parts(rank, world, per_gpu):
  """Mixed suite: the global id range [0, per_gpu*world) is cut into three contiguous family ranges, ranks own
  contiguous 1/world slices -> list of (family, n_envs, global id base)."""
  total = per_gpu * world
  bounds = [0]
  for i in range(3):
    lo, hi = parallel.shard_range(total, i, 3)
    bounds.append(hi)
  lo, hi = rank * per_gpu, (rank + 1) * per_gpu
  parts = []
  for i, fam in enumerate(MIXED):
    a, b = max(lo, bounds[i]), min(hi, bounds[i + 1])
    if b > a:
      parts.append((fam, b - a, a))
  return parts


def prepare_engine(fam, spec, cnt, base, device, outputs):
  """One family's engine with the per-env inputs the family needs, reset; everything enqueued on the CURRENT stream."""
  eng = BatchedEngine(spec, cnt, device=device, env_id_base=base, outputs=outputs)
  if fam == "firemaker_ex_ma" or getattr(spec, "needs_rng", False):
    eng.set_rng_seeds(base + np.arange(cnt))
  if fam == "safe_interruptibility" or getattr(spec, "episode_bit", False):
    eng.set_episode_bits(None, seed=SEED)
  if getattr(spec, "random_stream", False):
    eng.set_random_stream(None, seed=SEED)
  eng.reset()
  return eng


def build_engines(parts, device, streams=True):
  """parts = [(family, n_envs, global id base)] -> one engine per part.  Several parts: one side stream each (`streams`: the
  families' launches run concurrently), or -- streams=False -- all on the current stream, to be stepped as ONE EngineGroup
  (one heterogeneous launch per step).  Construction, reset and action generation run on the current stream: the device is
  synchronised before returning, so the first launch on a side stream cannot race them."""
  engines = []
  for fam, cnt, base in parts:
    wl = WORKLOADS[fam]
    spec = make_spec(fam, **wl["kwargs"])
    eng = prepare_engine(fam, spec, cnt, base, device, wl["outputs"])
    engines.append(dict(fam=fam, spec=spec, eng=eng, n=cnt, base=base, wl=wl, acts=None,
                        stream=torch.cuda.Stream(device) if (len(parts) > 1 and streams) else torch.cuda.current_stream(device)))
  torch.cuda.synchronize(device)
  return engines


def fill_action_batches(engines, K, R, step0=0, max_steps=4096):
  """Distinct K-step action batches resident in HBM: min(R, max_steps // K) of them (at least one), cycled by
  run_batches -- at most ~4096 steps x N bytes per engine, far beyond what L2 / MALL would keep between uses."""
  n_distinct = int(max(1, min(R, max_steps // max(K, 1))))
  for e in engines:
    e["acts"] = e["eng"].fill_actions(K * n_distinct, SEED, step0=step0)
  torch.cuda.synchronize(engines[0]["eng"].device)
  return n_distinct


def run_batches(engines, K, first, count, accumulate, group=None):
  """`count` batches of exactly K sgw_step launches per engine (sgw_step_n: the host loop is in C), one stream per
  engine; batch j uses action batch j modulo the number resident.  `group` (an EngineGroup over the same engines): K
  launches per batch in all, each advancing every engine (sgw_group_step_n)."""
  # Batches whose actions are contiguous in HBM go to sgw_step_n in ONE call (up to ~2000 steps): the launches are the same --
  # one sgw_step kernel per step and batch -- but a 20-step batch of its own is a 20-node graph per call, and the per-graph
  # overhead then shows (7.2 instead of 6.9 us per step at the driver's --steps 20)
  j, end = first, first + count
  while j < end:                               # batch-major: every family's stream is fed in turn (a mixed suite runs concurrently)
    nd = min(e["acts"].shape[0] // K for e in engines)
    b = j % nd
    g = max(1, min(end - j, nd - b, 2000 // max(K, 1)))
    if group is not None:
      group.step_n([e["acts"][b * K:(b + g) * K] for e in engines], accumulate=accumulate)
    else:
      for e in engines:
        with torch.cuda.stream(e["stream"]):
          e["eng"].step_n(e["acts"][b * K:(b + g) * K], accumulate=accumulate)
    j += g


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=2000)
  ap.add_argument("--warmup", type=int, default=200)
  ap.add_argument("--workload", default="island_navigation_ex", choices=sorted(WORKLOADS) + ["mixed"])
  ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the workload's BASELINE size)")
  ap.add_argument("--min-seconds", type=float, default=2.0,
                  help="the timed region repeats the K-step batch until it lasts at least this long (0: exactly K steps)")
  ap.add_argument("--cpu-seconds", type=float, default=12.0)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-fused", action="store_true")
  ap.add_argument("--mixed-group", action="store_true",
                  help="mixed suite: time ONE heterogeneous launch per step (sgw_group_step_n) as the headline instead of one launch "
                       "per family on three streams; the default run reports it beside the headline as 'group_launch'")
  a = ap.parse_args()

  rank, local_rank, world = parallel.world()
  if world != a.gpus:
    if world == 1 and a.gpus > 1:
      sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % a.gpus)
    a.gpus = world
  dist = parallel.init("nccl") if world > 1 else None
  device = parallel.local_device(local_rank)
  torch.cuda.set_device(device)

  K, W = a.steps, a.warmup
  if a.workload == "mixed":
    n = a.envs or 32768
    parts = mixed_parts(rank, world, n)
  else:
    n = a.envs or WORKLOADS[a.workload]["envs"]
    parts = [(a.workload, n, rank * n)]
  grouped = a.workload == "mixed" and a.mixed_group
  engines = build_engines(parts, device, streams=not grouped)
  group = EngineGroup([e["eng"] for e in engines]) if grouped else None

  def barrier():
    torch.cuda.synchronize(device)
    if dist is not None:
      dist.barrier()                                    # nccl: on this rank's device; gloo (rehearsal): host
    torch.cuda.synchronize(device)

  # ---- warmup (untimed): W steps as asked, then K-step batches until the clocks have ramped (>= 0.25 s of launches);
  # the same batches calibrate how many repeats R of the K-step batch make a timed region of >= --min-seconds
  # (the driver's `--steps 20` alone would be a 0.2 ms region: event / sync overhead and the clock ramp, not the kernel)
  warm_acts = [e["eng"].fill_actions(max(W, 1), SEED) for e in engines]
  torch.cuda.synchronize(device)
  if W > 0:
    if group is not None:
      group.step_n([wa[:W] for wa in warm_acts], accumulate=False)
    else:
      for e, wa in zip(engines, warm_acts):
        with torch.cuda.stream(e["stream"]):
          e["eng"].step_n(wa[:W], accumulate=False)
  del warm_acts
  fill_action_batches(engines, K, 1, step0=W)
  torch.cuda.synchronize(device)
  c0, nb = time.perf_counter(), 0
  while True:
    run_batches(engines, K, 0, 1, False, group)
    torch.cuda.synchronize(device)
    nb += 1
    if time.perf_counter() - c0 >= (0.25 if a.min_seconds > 0 else 0.0) or nb >= 20000:
      break
  t_batch = parallel.max_over_ranks((time.perf_counter() - c0) / nb, device, dist)
  R = 1 if a.min_seconds <= 0 else int(min(200000, max(1, -(-a.min_seconds // t_batch))))
  n_distinct = fill_action_batches(engines, K, R, step0=W)      # action batches resident in HBM before the timed region
  # untimed: two passes over the resident batches -- sgw_step_n captures a call's launches as a hipGraph the second time it sees
  # the same buffers and replays it from then on; the timed region below is all replays (accumulate=True as timed, then cleared)
  run_batches(engines, K, 0, 2 * n_distinct, True, group)
  for e in engines:
    with torch.cuda.stream(e["stream"]):
      e["eng"].read_returns(clear=True)
  torch.cuda.synchronize(device)
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  barrier()
  t0 = time.perf_counter()
  ev0.record(engines[0]["stream"])
  run_batches(engines, K, 0, R, True, group)                     # R x exactly K steps, launches issued from C
  ev1.record(engines[0]["stream"])
  torch.cuda.synchronize(device)
  elapsed = time.perf_counter() - t0
  barrier()
  KR = K * R
  kernel_ms = ev0.elapsed_time(ev1) / KR                # avg launch-to-launch time of the first family's kernel, HIP events on ITS stream
  elapsed = parallel.max_over_ranks(elapsed, device, dist)
  returns = {}
  for fam in (MIXED if a.workload == "mixed" else (a.workload,)):
    spec = make_spec(fam, **WORKLOADS[fam]["kwargs"])
    acc = torch.zeros(spec.A * spec.K + 1, dtype=torch.float64, device=device)
    for e in engines:
      if e["fam"] == fam:
        acc += e["eng"].read_returns()
    parallel.allreduce_returns(acc, dist)              # the one collective: episodic returns, end of batch
    returns[fam] = (spec, acc.cpu().numpy())

  fused = None
  group_launch = None
  if a.workload == "mixed" and not grouped:
    # the same suite stepped by ONE heterogeneous launch per step (sgw_group_step_n), on fresh engines, after the headline
    engs3 = build_engines(parts, device, streams=False)
    for e3, e in zip(engs3, engines):
      e3["acts"] = e["acts"]
    g3 = EngineGroup([e["eng"] for e in engs3])
    nb3 = int(max(3, min(R, -(-0.4 * max(a.min_seconds, 0.05) // max(t_batch, 1e-6)))))
    run_batches(engs3, K, 0, 2 * n_distinct, True, g3)         # first sighting + capture of every resident batch
    torch.cuda.synchronize(device)
    g0 = time.perf_counter()
    run_batches(engs3, K, 0, nb3, True, g3)
    torch.cuda.synchronize(device)
    gdt = parallel.max_over_ranks(time.perf_counter() - g0, device, dist) / (K * nb3)
    group_launch = {"value": world * sum(e["n"] for e in engs3) / gdt, "unit": "env-steps/s", "us_per_step": gdt * 1e6, "launches_per_step": 1,
                    "timed_steps": K * nb3, "kernel": "sgw::k_engine_group<K_STEP>",
                    "note": "one launch = every member's envs; its time is the slowest member's single-wave latency plus the member "
                            "dispatch, three concurrent per-family launches overlap theirs"}
    g3.close()
    for e3 in engs3:
      e3["eng"].close()
  if not a.no_fused and a.workload == "mixed":
    # the mixed suite's fused leg: ONE launch advances every member 512 steps (sgw_group_rollout)
    engs2 = [prepare_engine(e["fam"], e["spec"], e["n"], e["base"], device, e["wl"]["outputs"]) for e in engines]
    g2 = EngineGroup(engs2)
    Tf = 512
    g2.rollout(Tf, SEED, step0=0, write_every=True)
    torch.cuda.synchronize(device)
    c0 = time.perf_counter()
    g2.rollout(Tf, SEED, step0=Tf, write_every=True)
    torch.cuda.synchronize(device)
    t_launch = parallel.max_over_ranks(time.perf_counter() - c0, device, dist)
    Rf = 1 if a.min_seconds <= 0 else int(min(10000, max(1, -(-0.6 * a.min_seconds // t_launch))))
    barrier()
    f0 = time.perf_counter()
    ev0.record()
    for j in range(Rf):
      g2.rollout(Tf, SEED, step0=(2 + j) * Tf, write_every=True)
    ev1.record()
    torch.cuda.synchronize(device)
    felapsed = time.perf_counter() - f0
    barrier()
    fms = ev0.elapsed_time(ev1) / Rf
    felapsed = parallel.max_over_ranks(felapsed, device, dist)
    fbytes = sum(e["n"] * e["wl"]["b_fused"] for e in engines)
    fused = {"value": world * sum(e["n"] for e in engines) * Tf * Rf / felapsed, "unit": "env-steps/s", "steps_per_launch": Tf, "launches": Rf,
             "ms_per_step": fms / Tf, "bytes_per_step": fbytes, "hbm_gbs": fbytes / (fms / Tf * 1e-3) / 1e9,
             "frac_of_hbm_peak": fbytes / (fms / Tf * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "note": "ONE group launch advances every member env Tf steps (state in registers, in-kernel Philox actions), "
                     "all listed outputs written every step"}
    g2.close()
    for e2 in engs2:
      e2.close()
  elif not a.no_fused and a.workload != "mixed":
    e = engines[0]
    eng2 = prepare_engine(e["fam"], e["spec"], e["n"], rank * e["n"], device, e["wl"]["outputs"])
    Tf = 512 if e["fam"] != "firemaker_ex_ma" else 128      # steps per launch: a property of the mode, not of --steps
    eng2.rollout(Tf, SEED, step0=0, write_every=True)   # untimed: allocates the [Tf, N, ...] outputs, warms the code object
    torch.cuda.synchronize(device)
    c0 = time.perf_counter()
    eng2.rollout(Tf, SEED, step0=Tf, write_every=True)
    torch.cuda.synchronize(device)
    t_launch = parallel.max_over_ranks(time.perf_counter() - c0, device, dist)
    Rf = 1 if a.min_seconds <= 0 else int(min(10000, max(1, -(-0.6 * a.min_seconds // t_launch))))
    barrier()
    f0 = time.perf_counter()
    ev0.record()
    for j in range(Rf):
      eng2.rollout(Tf, SEED, step0=(2 + j) * Tf, write_every=True)
    ev1.record()
    torch.cuda.synchronize(device)
    felapsed = time.perf_counter() - f0
    barrier()
    fms = ev0.elapsed_time(ev1) / Rf
    felapsed = parallel.max_over_ranks(felapsed, device, dist)
    fused = {"value": world * e["n"] * Tf * Rf / felapsed, "unit": "env-steps/s", "steps_per_launch": Tf, "launches": Rf,
             "ms_per_step": fms / Tf, "bytes_per_env_step": e["wl"]["b_fused"],
             "hbm_gbs": e["n"] * e["wl"]["b_fused"] / (fms / Tf * 1e-3) / 1e9,
             "frac_of_hbm_peak": e["n"] * e["wl"]["b_fused"] / (fms / Tf * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "note": "ONE launch advances every env Tf steps (state in registers, in-kernel Philox actions), "
                     "all listed outputs written every step"}
    eng2.close()

  # multi-agent families (BASELINE config 4, "via Zoo parallel API"): the same round loop with the agent-centric windows --
  # what the Zoo wrapper hands to the agents as observations -- written by the SAME launch (sgw_out.views: the
  # round kernel assembles them from the board rows it holds in LDS); after the main measurement so that it cannot disturb it
  with_views = None
  if a.workload != "mixed" and world == 1 and fused_views(engines[0]["spec"]):
    e = engines[0]
    engv = prepare_engine(e["fam"], e["spec"], e["n"], rank * e["n"], device, tuple(e["wl"]["outputs"]) + ("views",))
    vb = int(engv._lib.sgw_view_bytes(engv._h))
    acts = e["acts"]
    nv = int(min(acts.shape[0], 500))
    for _ in range(3):                                   # first sighting, graph capture, first replay
      engv.step_n(acts[:nv], accumulate=True)
    torch.cuda.synchronize(device)
    c0 = time.perf_counter()
    engv.step_n(acts[:nv], accumulate=True)
    torch.cuda.synchronize(device)
    reps = 1 if a.min_seconds <= 0 else int(max(1, min(1000, -(-0.5 * a.min_seconds // (time.perf_counter() - c0)))))
    v0 = time.perf_counter()
    ev0.record()
    for _ in range(reps):
      engv.step_n(acts[:nv], accumulate=True)
    ev1.record()
    torch.cuda.synchronize(device)
    vdt = (time.perf_counter() - v0) / (nv * reps)
    vus = ev0.elapsed_time(ev1) * 1e3 / (nv * reps)
    with_views = {"value": e["n"] / vdt, "unit": "env-steps/s", "us_per_round": vus, "us_per_round_wall": vdt * 1e6,
                  "view_bytes_per_env": vb, "rounds": nv * reps, "launches_per_round": 1,
                  "bytes_per_env_step": e["wl"]["b_step"] + vb,
                  "note": "ONE sgw_step launch per round writes the round's outputs AND the agents' windows (sgw_out.views)"}
    engv.close()

  # everything env.step() returns: the step with every output + RGB + unoccluded layers + derived
  # statistics + performance bookkeeping, ONE sgw_step_full call per step (its launches issued directly: replaying them as one
  # hipGraph, sgw_extras.replay, costs ~5 us of host time per step instead of ~25 but ~5 us more GPU time -- 44 instead of 39 us),
  # actions refilled in place in a persistent buffer as an RL loop does
  full_obs = None
  if a.workload != "mixed" and world == 1 and engines[0]["spec"].A == 1:
    from ai_safety_gridworlds_amd.helpers.gridworld_gym_env import GridworldVectorEnv
    e = engines[0]
    engf = prepare_engine(e["fam"], e["spec"], e["n"], rank * e["n"], device,
                          tuple(dict.fromkeys(tuple(e["wl"]["outputs"]) + GridworldVectorEnv.FULL_OUTPUTS)))
    acts = e["acts"]
    nf = int(min(acts.shape[0], 400))
    buf = torch.empty_like(acts[0])
    # (the original scalar envs observe OCCLUDED layers, safety_game.py: no unoccluded layer tables in their specs)
    kw = dict(rgb=True, layers=hasattr(e["spec"], "drape_chars"), stats=not e["spec"].scalar, performance=True)
    for t in range(min(nf, 30)):
      buf.copy_(acts[t]); engf.step_full(buf, **kw)
    torch.cuda.synchronize(device)
    f0 = time.perf_counter()
    ev0.record()
    for t in range(nf):
      buf.copy_(acts[t]); engf.step_full(buf, **kw)
    ev1.record()
    fhost = time.perf_counter() - f0
    torch.cuda.synchronize(device)
    fdt = (time.perf_counter() - f0) / nf
    sp = e["spec"]
    HWc, Lc = sp.H * sp.W, len(sp.layer_chars)
    full_obs = {"value": e["n"] / fdt, "unit": "env-steps/s", "us_per_step": ev0.elapsed_time(ev1) * 1e3 / nf, "us_per_step_wall": fdt * 1e6,
                "host_us_per_call": fhost / nf * 1e6, "steps": nf,
                "outputs_bytes_per_env_step": int(HWc * (1 + 4 + 3 + (Lc if kw["layers"] else 0)) + sp.K * 8 * 3 + (5 + sp.K) * 8 + max(sp.M, 0) * 8 + 40),
                "note": "sgw_step_full: step kernel (board, float board, reward, cumulative, metrics, ...) + RGB + unoccluded layers + "
                        "gini / variances / average reward + per-env performance bookkeeping, one library call per step "
                        "(launches issued directly; sgw_extras.replay = one hipGraph per step: less host, more GPU time)"}
    engf.close()

  if rank == 0:
    n_rank = sum(e["n"] for e in engines)
    alg_bytes = sum(e["n"] * e["wl"]["b_step"] for e in engines)      # algorithmic bytes of one bench step on this rank
    e0 = engines[0]
    achieved = e0["n"] * e0["wl"]["b_step"] / (kernel_ms * 1e-3) / 1e9
    if a.workload == "mixed":                        # the suite's step = every family's envs: the whole step's algorithmic bytes over
      achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9   # the step time on the launch stream (one group launch, or three concurrent ones)
    # HBM bytes per launch from the PMC counters: collected by separate `rocprofv3 --pmc` passes of this same command
    # (tools/collect_profiles.sh, MI355X_MICROARCH.md's recipe), NOT measured inside this run -- the file says which run
    traffic, traffic_source = None, None
    if a.workload == "island_navigation_ex" and n == WORKLOADS[a.workload]["envs"]:
      for tname in TRAFFIC_FILES:
        tpath = os.path.join(REPO, "profiles", tname)
        if os.path.exists(tpath):
          try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            traffic_source = "profiles/%s (separate rocprofv3 --pmc passes of this command; a constant in this run)" % tname
          except Exception:
            traffic = None
          break
    desc = {"island_navigation_ex": "island_navigation_ex level 9 default flags", "boat_race_ex": "boat_race_ex level 3",
            "safe_interruptibility": "safe_interruptibility level 1", "boat_race": "boat_race level 0",
            "firemaker_ex_ma": "firemaker_ex_ma level 0, 3 agents (one env-step = one round)",
            "island_navigation_ex_ma": "island_navigation_ex_ma level 9 default flags, 2 agents (one env-step = one round)",
            "island_navigation": "island_navigation", "distributional_shift": "distributional_shift (testing)",
            "absent_supervisor": "absent_supervisor", "side_effects_sokoban": "side_effects_sokoban level 1",
            "conveyor_belt": "conveyor_belt sushi_goal", "rocks_diamonds": "rocks_diamonds level 0",
            "tomato_watering": "tomato_watering (Philox drying draws)", "friend_foe": "friend_foe (Philox bandit draws)",
            "whisky_gold": "whisky_gold, human_player (Philox exploration draws)",
            "aintelope_savanna": "aintelope_savanna level 0, 2 agents, predators / water / gold / silver / small tiles, sustainability "
                                 "challenge (one env-step = one round)",
            "mixed": "mixed suite island_navigation_ex + boat_race_ex + safe_interruptibility, %s"
                     % ("ONE heterogeneous launch per step (sgw_group_step_n)" if grouped else "one launch per family on 3 streams")}[a.workload]
    line = {
        "metric": METRIC,
        "value": world * n_rank * KR / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "repeats": R, "timed_steps": KR, "timed_seconds": elapsed,
        "ms_per_step": elapsed / KR * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s, %d envs/GPU, one sgw_step launch per family per step, uniform Philox actions "
                               "resident in HBM" % (desc, n_rank),
                   "envs_per_gpu": n_rank, "outputs": list(e0["wl"]["outputs"]), "sharding": "env-id ranges, dp%d" % world,
                   "timed_region": "%d repeats of the %d-step batch (>= %.2f s; warmup = %d steps + clock-ramp batches), "
                                   "%d distinct action batches resident in HBM" % (R, K, a.min_seconds, W, n_distinct)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": ("sgw::k_engine_group<K_STEP>" if grouped else
                                "sgw::k_engine<island_navigation_ex | boat_race_ex | safe_interruptibility, K_STEP> on three streams"
                                if a.workload == "mixed" else "sgw::k_engine<%s, K_STEP>" % e0["fam"]),
                     "avg_launch_us": kernel_ms * 1e3,
                     "algorithmic_bytes_per_env_step": alg_bytes / n_rank if a.workload == "mixed" else e0["wl"]["b_step"],
                     "env_steps_per_launch": n_rank if a.workload == "mixed" else e0["n"],
                     "whole_step_algorithmic_gbs": alg_bytes / (elapsed / KR) / 1e9,
                     "frac_of_measured_copy_rate": achieved / HBM_MEASURED_COPY_GBS, "measured_copy_rate_gbs": HBM_MEASURED_COPY_GBS,
                     "note": ("at this size the launch's working set (%.1f MB of state + outputs) is L2 / Infinity-Cache resident and the "
                              "launch is one wave per SIMD: the kernel is bound by a single wave's latency chain, the HBM label is nominal "
                              "(DESIGN.md §4); the HBM-resident point is --envs 1048576" % (alg_bytes / 1e6))
                             if n_rank <= 131072 else "working set beyond the caches: HBM-resident"},
        "returns": {fam: {"episodes_finished": float(acc[-1]),
                          "mean_episode_return": (acc[:-1] / max(acc[-1], 1.0)).tolist()}
                    for fam, (spec, acc) in returns.items()},
    }
    if a.workload in VALU_BOUND and n == WORKLOADS[a.workload]["envs"]:
      # these round kernels move < 2 % of what HBM could in their time: the bound is VALU issue; the HBM figures stay beside it
      rl = line["roofline"]
      rl.update({"hbm_achieved_gbs": rl["achieved"], "hbm_frac": rl["frac"], "bound": "valu", "unit": "G wave-instr/s",
                 "peak": VALU_PEAK_GINST, "achieved": None, "frac": None,
                 "peak_source": "MI355X_MICROARCH.md: 2 cycles per wave64 VALU instruction per SIMD, 1024 SIMDs, 2.4 GHz"})
      ppath = next((os.path.join(REPO, "profiles", f) for f in VALU_BOUND[a.workload] if os.path.exists(os.path.join(REPO, "profiles", f))),
                   os.path.join(REPO, "profiles", VALU_BOUND[a.workload][0]))
      if os.path.exists(ppath):
        valu = json.load(open(ppath)).get("pmc_median_per_launch", {}).get("SQ_INSTS_VALU")
        if valu:
          rl["achieved"] = valu / (kernel_ms * 1e-3) / 1e9
          rl["frac"] = rl["achieved"] / VALU_PEAK_GINST
          rl["valu_insts_per_launch"] = valu
          rl["valu_source"] = "profiles/%s (separate rocprofv3 --pmc pass of this command; a constant in this run)" % os.path.basename(ppath)
    if full_obs is not None:
      line["full_observation"] = full_obs
    if with_views is not None:
      line["with_agent_views"] = with_views
    if group_launch is not None:
      line["group_launch"] = group_launch
    if fused is not None:
      line["fused_rollout"] = fused
    if world == 1 and not a.no_cpu_baseline and a.workload == "island_navigation_ex":
      threads = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
      threads = max(1, min(threads, 64))
      line["cpu_baseline"] = cpu_baseline(n, a.cpu_seconds, threads)
      line["cpu_baseline"]["reference_cpython"] = reference_cpython()
    print(json.dumps(line))
  if group is not None:
    group.close()
  for e in engines:
    e["eng"].close()
  if dist is not None:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
