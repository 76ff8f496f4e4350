#!/usr/bin/env python3
"""Regenerate the round-3 part of profiles/README.md from the collection's own JSON / CSV files (everything above "# Earlier
rounds" is rewritten; the earlier rounds' text below it is kept).    python tools/profiles_readme.py v2"""
import csv, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "v2"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles") + "/"


def load(name):
  return json.loads(open(P + name).read().strip().splitlines()[-1])


def sci(x):
  e = int(("%e" % x).split("e")[1])
  return "%.2f × 10^%d" % (x / 10 ** e, e)


def rocprof(name, sub, kind=None):
  for r in csv.DictReader(open(P + name)):
    if sub in r["Name"] and (kind is None or kind in r["Name"]):
      return float(r["AverageNs"]) / 1e3, int(r["Calls"])
  return float("nan"), 0


T = "r03_%s_" % tag
b, fm, mx = load(T + "bench.json"), load(T + "bench_firemaker_ex_ma.json"), load(T + "bench_mixed.json")
pmc, side = json.load(open(P + T + "pmc.json")), json.load(open(P + T + "side_probe.json"))
old = open(P + "README.md").read()
earlier = old[old.index("# Earlier rounds"):]
rows = [("boat_race_ex", "boat_race_ex L3"), ("safe_interruptibility", "safe_interruptibility L1"), ("boat_race", "boat_race L0"),
        ("firemaker_ex_ma", "firemaker_ex_ma (3 agents, rounds)"), ("island_navigation_ex_ma", "island_navigation_ex_ma L9 (2 agents, rounds)"),
        ("aintelope_savanna", "aintelope_savanna L0 (2 agents, all tile types; rounds)"), ("island_navigation", "island_navigation"),
        ("distributional_shift", "distributional_shift (testing)"), ("absent_supervisor", "absent_supervisor"),
        ("side_effects_sokoban", "side_effects_sokoban L1"), ("conveyor_belt", "conveyor_belt sushi_goal"), ("rocks_diamonds", "rocks_diamonds"),
        ("tomato_watering", "tomato_watering"), ("friend_foe", "friend_foe"), ("whisky_gold", "whisky_gold")]
tab = ""
for k, label in rows:
  d = load(T + "bench_%s.json" % k)
  try:
    o = load("r02_v7_bench_%s.json" % k)
  except Exception:
    o = None
  rl = d["roofline"]
  tab += "| %s | %d | %s | %.1f (%s) | %.3f (%s) | %s |\n" % (label, d["config"]["envs_per_gpu"], sci(d["value"]), d["ms_per_step"] * 1e3,
                                                          ("%.1f" % (o["ms_per_step"] * 1e3)) if o else "-", rl["frac"] or 0, rl["bound"],
                                                          sci(d["fused_rollout"]["value"]))
sizes = ""
for n in (131072, 262144, 1048576):
  d = load(T + "bench_island_n%d.json" % n)
  sizes += "| %d | %s | %.2f | %.3f | %.2f | %.3f |\n" % (n, sci(d["value"]), d["ms_per_step"] * 1e3, d["roofline"]["frac"],
                                                          d["roofline"]["frac"] * 8000.0 / 6290.0, d["fused_rollout"]["frac_of_hbm_peak"])
sidetab = ""
for r in side["rows"]:
  sidetab += "| %s | %.2f | %.1f | %.0f | %.3f |\n" % (r["kernel"], r["us_per_launch"], r["algorithmic_bytes"] / 1e6, r["gb_per_s"], r["frac_of_hbm_peak"])
pm = pmc["pmc_median_per_launch"]
fo, wv, gl = b["full_observation"], fm["with_agent_views"], mx["group_launch"]
isl_us, isl_calls = rocprof(T + "kernel_stats.csv", "IslandT<false, true>, 0>")
fm_us, fm_calls = rocprof(T + "kernel_stats_firemaker.csv", "FiremakerT<false>, 0>")
g0_us, g0_calls = rocprof(T + "kernel_stats_mixed.csv", "k_engine_group<0>")
boat_us, _ = rocprof(T + "kernel_stats_mixed.csv", "sgw::Boat, 0>")
mi_us, _ = rocprof(T + "kernel_stats_mixed.csv", "IslandT<false, true>, 0>")
si_us, _ = rocprof(T + "kernel_stats_mixed.csv", "sgw::SafeInt, 0>")
side_rp = "; ".join("%s %.1f µs" % (k, rocprof(T + "kernel_stats_side.csv", k)[0]) for k in
                    ("k_derived_stats", "k_savanna_layers", "k_agent_layer_views_lds"))
fm_valu = json.load(open(P + "r03_pmc_firemaker_ex_ma.json"))["pmc_median_per_launch"]["SQ_INSTS_VALU"]
new = f'''# profiles/ — measurements (1× MI355X, gfx950, ROCm 7.2)

Produced on the GPU box by `tools/collect_profiles.sh <tag>`, condensed by `tools/summarize_profiles.py <tag> r03`; this file's
round-3 tables are generated from those files by `tools/profiles_readme.py {tag}`.  `gpurun_out/` (raw CSVs) is scratch and not tracked.
One collection per round is kept: **`r03_{tag}_*` (round 3, final)**, `r02_v7_*` (round 2), `r01_v9_*` (round 1).

| file | what |
|---|---|
| `r03_{tag}_bench.json` | the `bench.py` JSON line (default run: island_navigation_ex L9, 65 536 envs; `roofline.traffic` = this collection's PMC figure; `full_observation` = every observation key of `env.step()` per step) |
| `r03_{tag}_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats` of `bench.py --steps 500 --warmup 50` |
| `r03_{tag}_kernel_stats_firemaker.csv`, `_mixed.csv`, `_side.csv` | the same for `bench.py --workload firemaker_ex_ma` (round kernel with and without the agents' windows), `--workload mixed` (three per-family kernels + `k_engine_group<0 / 1>`), and `tools/diag/side_probe.py` (derived statistics, RGB / layers, windows, layer cubes) |
| `r03_{tag}_pmc.json`, `r03_traffic.json` | per-launch medians of the PMC passes (separate `--pmc` runs: FETCH_SIZE / WRITE_SIZE / SQ_*); HBM bytes per launch of the step kernel, read by `bench.py` into `roofline.traffic` |
| `r03_{tag}_bench_<workload>.json`, `r03_{tag}_bench_island_n<N>.json` | every other family through `bench.py --workload …`; the headline kernel at 131 072 / 262 144 / 1 048 576 envs |
| `r03_{tag}_side_probe.json / .txt` | device time (HIP events, unprofiled), algorithmic bytes, HBM fraction of every side kernel at the BASELINE sizes |
| `r03_{tag}_host_probe.txt`, `r03_{tag}_zoo_vector_probe.txt`, `r03_{tag}_agent_views_probe.txt` | host vs end-to-end time of every way to issue one step (bare C call, engine, graph replay, `GridworldVectorEnv.step` default and `full_info=True`); `GridworldZooVectorEnv.step` next to the round kernel alone; `sgw_agent_views` alone |
| `r03_{tag}_phase_stamps.txt` | in-kernel phase stamps of the headline kernel (diagnostic build `tools/diag/stamp_probe.hip`) |
| `r03_pmc_firemaker_ex_ma.json`, `r03_pmc_aintelope_savanna.json` | SQ counters of the two instruction-bound round kernels (`bench.py` reads SQ_INSTS_VALU from them) |
| `r03_pmc_agent_views.json` | SQ counters of the window kernel (`sgw_agent_views`) at the two BASELINE sizes: 314 VALU + 287 SALU + 34 LDS wave-instructions per savanna env, 0.04 bank conflicts per LDS instruction (DESIGN.md §4.8) |
| `r03_kernel_registers.md` | VGPR / AGPR / SGPR / spills / scratch of every kernel in `libsgw.so` + the exec-restore lint verdict |
| `r02_ima_miscompile.txt` | the evidence for DESIGN.md §8 "the fused-kernel fault of round 1" |
| `reference_cpython.json` | the reference itself timed under CPython in the build container (`tools/time_reference.py`) |

## Headline (BASELINE.json configs[1]): island_navigation_ex L9, 65 536 envs, one launch per step

| quantity | round 3 | round 2 | round 1 | source |
|---|---|---|---|---|
| env-steps/s (value) | **{sci(b["value"])}** | 9.4-9.9 × 10^9 | 7.62 × 10^9 | r03_{tag}_bench.json |
| avg launch-to-launch, HIP events on the launch stream | {b["roofline"]["avg_launch_us"]:.2f} µs | 6.66 µs | 8.59 µs | `roofline.avg_launch_us` |
| avg kernel duration, rocprofv3 `--stats` (`k_engine<IslandT<false, true>, 0>`, {isl_calls} calls) | {isl_us:.2f} µs | 6.34 µs | 8.60 µs | r03_{tag}_kernel_stats.csv |
| algorithmic bytes / launch (299 B × 65 536) | 19.6 MB | 19.6 MB | 19.6 MB | SURVEY §8(d) |
| roofline: achieved / peak / frac | {b["roofline"]["achieved"]:.0f} GB/s / 8 000 GB/s / **{b["roofline"]["frac"]:.3f}** ({b["roofline"]["frac_of_measured_copy_rate"]:.2f} of the guide's measured 6.29 TB/s copy rate) | 0.35-0.37 | 0.285 | r03_{tag}_bench.json |
| HBM traffic / launch (PMC) | **{pmc["hbm_bytes_per_launch"]/1e6:.2f} MB** = 2×FETCH_SIZE {pm["FETCH_SIZE"]:.0f} KiB + WRITE_SIZE {pm["WRITE_SIZE"]:.0f} KiB | 20.0-20.3 MB | 29.8 MB | r03_{tag}_pmc.json |
| fused rollout (512 steps / launch, all outputs every step) | {sci(b["fused_rollout"]["value"])}, {b["fused_rollout"]["frac_of_hbm_peak"]:.3f} | 2.73 × 10^10, 0.46-0.47 | 2.0 × 10^10, 0.35 | r03_{tag}_bench.json |
| **everything `env.step()` returns** (`full_observation`: board, float board, reward, cumulative, metrics + RGB + unoccluded layers + gini / variances / average reward + performance bookkeeping; ONE `sgw_step_full` call per step, launches issued directly) | **{sci(fo["value"])} env-steps/s**, {fo["us_per_step"]:.1f} µs per step, {fo["host_us_per_call"]:.1f} µs of host time per call incl. the action copy (`extras.replay`: one hipGraph per step, 5 µs of host time, 44 µs per step) | four launches from Python (~15.8 µs of host time for the step alone) | - | r03_{tag}_bench.json |
| CPU baseline ("port": C oracle, same stream) | {sci(b["cpu_baseline"]["value"])} env-steps/s on {b["cpu_baseline"]["cores"]} threads ({sci(b["cpu_baseline"]["single_thread_value"])} on one) | 4.8 × 10^7 | | r03_{tag}_bench.json |
| reference CPython (build container, 1 core) | 2.5 × 10^3 env-steps/s | | | reference_cpython.json |

The step kernel itself did not change in round 3 beyond the `done / obs_dir / act_dir` outputs of ABI 8, whose pointers are read from the kernarg segment only when asked for (bounded experiments were measured and not kept: DESIGN.md §4.10).  At this size the
launch is 1 024 waves = one per SIMD and its 14 MB working set is L2 / Infinity-Cache resident: the kernel is **latency-bound**, the
HBM label is nominal (PMC: `SQ_WAIT_ANY / SQ_WAVE_CYCLES` = {pm["SQ_WAIT_ANY"]/pm["SQ_WAVE_CYCLES"]:.2f}; VALU {pm["SQ_INSTS_VALU"]/1024:.0f}, SALU {pm["SQ_INSTS_SALU"]/1024:.0f}, LDS {pm["SQ_INSTS_LDS"]/1024:.0f} instructions per wave).
Phase stamps (`r03_{tag}_phase_stamps.txt`): loads arrive 0.60 µs after issue · rules 1.85 · output phase 1.43 · state stores 0.23 ⇒ 4.1 µs of
wave life; all waves end 4.4-5.2 µs after the first starts; the other ≈ 1.5 µs of a launch are the kernel boundary.

| envs | env-steps/s | µs / launch | frac of 8 TB/s | of the guide's measured 6.29 TB/s copy rate | fused frac |
|---|---|---|---|---|---|
{sizes}
## Round 3: BASELINE configs 4 and 5

| quantity | round 3 | round 2 | source |
|---|---|---|---|
| firemaker_ex_ma, 16 384 envs x 3 agents: µs per round | **{fm["ms_per_step"]*1e3:.1f}** ({fm["roofline"]["frac"]:.3f} of VALU issue) | 80.4-80.7 | r03_{tag}_bench_firemaker_ex_ma.json |
| ... with the three agents' windows (the Zoo API's observation) | **{wv["us_per_round"]:.1f} µs from ONE launch** (`sgw_out.views`) | 101.9 µs in two launches | `with_agent_views` |
| `GridworldZooVectorEnv.step` (dict in, dicts of device tensors out) | 84-93 µs per round = the round kernel with the Zoo outputs (87-89 µs alone) | 125 µs | r03_{tag}_zoo_vector_probe.txt |
| rocprofv3 avg of `k_engine<FiremakerT<false>, 0>` over the run (both variants, {fm_calls} calls) | {fm_us:.1f} µs | 80.7 µs | r03_{tag}_kernel_stats_firemaker.csv |
| mixed suite (island_navigation_ex + boat_race_ex + safe_interruptibility, 3 x 10 923 envs): three concurrent per-family launches per step | {sci(mx["value"])} env-steps/s, {mx["ms_per_step"]*1e3:.2f} µs per step, {mx["roofline"]["frac"]:.3f} of HBM over the whole step (rocprofv3: Boat {boat_us:.2f}, Island {mi_us:.2f}, SafeInt {si_us:.2f} µs) | 4.38 × 10^9, 7.5 µs | r03_{tag}_bench_mixed.json, r03_{tag}_kernel_stats_mixed.csv |
| ... as ONE heterogeneous launch per step (`k_engine_group<0>`) | {sci(gl["value"])}, {gl["us_per_step"]:.2f} µs (rocprofv3 avg {g0_us:.2f} µs over {g0_calls} calls) | - | `group_launch` |
| ... fused: one group launch advances every member 512 steps (`k_engine_group<1>`) | **{sci(mx["fused_rollout"]["value"])} env-steps/s** ({mx["fused_rollout"]["ms_per_step"]*1e3:.2f} µs per step) | no fused leg | `fused_rollout` |

A launch's time is its slowest member's single-wave latency (boat_race_ex: 7.07 µs at 10 923 envs) plus the member dispatch; three streams
overlap their latencies -- hence one launch per step is 6 % slower than three and `bench.py` keeps the streams for the headline value.

## Round 3: side kernels (`tools/diag/side_probe.py`; HIP events over back-to-back launches, unprofiled; bytes = inputs read + outputs written once)

| kernel | µs / launch | MB | GB/s | frac of 8 TB/s |
|---|---|---|---|---|
{sidetab}
rocprofv3 durations of the same launches (`r03_{tag}_kernel_stats_side.csv`; `k_observe*` rows average several board sizes; `k_agent_layer_views_lds<1>` = the layer cubes, `<4>` = the one-plane launches of `sgw_agent_views`): {side_rp}.
History inside the round: `k_derived_stats` 29.7 µs with 2.6 KB of scratch per lane (round 2) → 23.1 µs (vectors streamed from LDS at run-time
indices: serial LDS round trips) → 7.2-7.4 µs (compile-time K, vectors in registers).  `k_observe` RGB 9.0 → 4.6 µs, RGB + occluded layers 23.2
→ 10.6 µs, `k_observe_layers` 21.8 → 12.9 µs (first LDS-staged version, 16 output bytes per lane → four cells and a dword store per lane and
plane; unaligned dword stores where H·W is odd).  Firemaker's 17 × 17 layers 88 → 51 (dword path) → 23.8 µs (16 instead of 64 envs per
workgroup: 24 instead of 95 KB of LDS).  `k_savanna_layers` 137 → 45 µs (state words and code vectors in LDS, the same plane writer).
`k_agent_layer_views` 208 → 160 (a workgroup per env, planes and the env's output row in LDS) → 121 µs (lane constants hoisted out of the env loop,
dword pad fill) → 106 µs (per-agent table and per-env words read from LDS instead of select chains over kernel-argument arrays, rows out as
aligned 16-byte chunks); `sgw_agent_views` 26.3 → 17.5 µs through the same kernel with one plane (four envs per one-wave workgroup;
firemaker's windows normally come from the round's own launch); 65 536 savanna envs' 21 × 21 windows 60 → 56 µs (`r03_{tag}_agent_views_probe.txt`).
Both are instruction-bound: a whole wave issues every instruction of a window's assembly (DESIGN.md §4.8).

## Other workloads (`r03_{tag}_bench_*.json`; in brackets round 2's µs per launch)

| workload | envs | env-steps/s | µs / launch (r02) | frac (bound) | fused rollout |
|---|---|---|---|---|---|
{tab}
island_navigation_ex_ma and aintelope_savanna trade 9 % / 2.5 % at this size for register headroom (the cumulative vectors wait in LDS while
the rules run: 230 VGPR / 0 AGPR and 297 + 41 instead of 278 + 22 and 349 + 93); at 262 144 envs they run 59 instead of 70 µs and 175
instead of 282 µs per round (DESIGN.md §4.9).  The firemaker VALU fraction uses this collection's SQ_INSTS_VALU ({fm_valu/1e6:.1f} M wave-instructions per launch).

Python-level step paths (`r03_{tag}_host_probe.txt`, `r03_{tag}_zoo_vector_probe.txt`).  The wrappers' per-step decodes (`terminated`, the
observation / action directions) are outputs of the step launch since ABI 8 (`sgw_out.done / obs_dir / act_dir`), so a Python step launches no
torch kernel of its own and costs what the kernel under it costs: `GridworldVectorEnv.step` 9.0 µs end to end with 5.4 µs of host time (the step
kernel with these outputs: 9.0 µs; 12.3 µs host-bound before), `full_info=True` 39 µs end to end with 25 µs of host time (direct launches;
44 µs and 5 µs as one hipGraph replay per step); `GridworldZooVectorEnv.step` island_navigation_ex_ma 28.6 µs per round (kernel 28.4; 44.6 before,
51 in round 2), firemaker 84-93 (kernel 87-89; 125 in round 2), aintelope_savanna 91 (kernel 32 + its 21 × 21 windows, larger than the board,
from the separate window kernel: 56 µs); with the layer cubes (`layers_in_observation=True`) firemaker 225 µs (377 before the layer kernels'
rewrite), ima 170 µs.  A bare `sgw_step` call costs 4.0 µs of host time; a ONE-node hipGraph replay of the same kernel runs 13.7 µs end to end
instead of 9.0 (the graph launch's own device-side cost: why `sgw_step_full` issues its launches directly by default).

'''
open(P + "README.md", "w").write(new + earlier)
print("profiles/README.md regenerated for tag", tag)
