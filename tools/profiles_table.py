#!/usr/bin/env python3
"""Markdown rows of profiles/README.md's "Other workloads" table from profiles/r02_<tag>_bench_*.json (and, in brackets, an older tag).
    python tools/profiles_table.py v6 v3"""
import json, sys

tag, old = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
ROWS = [("boat_race_ex", "boat_race_ex L3"), ("safe_interruptibility", "safe_interruptibility L1"), ("boat_race", "boat_race L0"),
        ("firemaker_ex_ma", "firemaker_ex_ma (3 agents, rounds)"), ("island_navigation_ex_ma", "island_navigation_ex_ma L9 (2 agents, rounds)"),
        ("aintelope_savanna", "aintelope_savanna L0 (2 agents, predators, all tile types, sustainability; rounds)"),
        ("island_navigation", "island_navigation"), ("distributional_shift", "distributional_shift (testing)"),
        ("absent_supervisor", "absent_supervisor"), ("side_effects_sokoban", "side_effects_sokoban L1"),
        ("conveyor_belt", "conveyor_belt sushi_goal"), ("rocks_diamonds", "rocks_diamonds"), ("tomato_watering", "tomato_watering"),
        ("friend_foe", "friend_foe"), ("whisky_gold", "whisky_gold"),
        ("mixed", "mixed suite (island_navigation_ex + boat_race_ex + safe_interruptibility, 3 streams, 10 923 envs each)")]


def load(t, name):
  try:
    return json.loads(open("profiles/r02_%s_bench_%s.json" % (t, name)).read().strip().splitlines()[-1])
  except Exception:
    return None


def sci(x):
  e = int(("%e" % x).split("e")[1])
  return "%.2f × 10%s" % (x / 10 ** e, str(e).translate(str.maketrans("0123456789", "⁰¹²³⁴⁵⁶⁷⁸⁹")))


for key, label in ROWS:
  d, o = load(tag, key), (load(old, key) if old else None)
  if d is None:
    continue
  fu, ofu = d.get("fused_rollout") or {}, (o or {}).get("fused_rollout") or {}
  envs = d["config"].get("envs_per_gpu", 65536)
  print("| %s | %s | %s | %.1f%s | %.3f (%s) | %s |" % (
      label, format(envs, ",").replace(",", " "), sci(d["value"]), d["ms_per_step"] * 1e3, (" (%.1f)" % (o["ms_per_step"] * 1e3)) if o else "",
      d["roofline"]["frac"], d["roofline"]["bound"],
      ("%s%s" % (sci(fu["value"]), (" (%s)" % sci(ofu["value"])) if ofu else "")) if fu else "—"))
