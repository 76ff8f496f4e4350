#!/usr/bin/env python3
"""Experiment presets (reference ai_safety_gridworlds/experiments/**): the flag values each preset's
init_experiment_flags() sets, recorded as DATA by running it in the build container and diffing against the base
environment's define_flags() defaults.

    python tests/golden/make_experiment_presets.py   ->  ai_safety_gridworlds_amd/experiment_presets.json

Same rules as the fixture generators: test-only stand-ins for absl / gymnasium seeding, nothing of the reference's
source is stored -- only {preset name: base environment, {flag: value}}.
"""
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(REPO, "ai_safety_gridworlds_amd", "experiment_presets.json")


def plain(v):
  if hasattr(v, "_reward_dimensions_dict"):
    return {"__mo_reward__": {k: float(x) for k, x in v._reward_dimensions_dict.items()}}
  if isinstance(v, (list, tuple)):
    return [plain(x) for x in v]
  if isinstance(v, (bool, int, float, str)) or v is None:
    return v
  raise TypeError("flag value of type %r" % type(v))


def flag_values(FLAGS):
  out = {}
  for name in FLAGS:
    out[name] = plain(FLAGS[name].value)
  return out


def main():
  # scratch cwd: the reference's step logger writes ./logs/*.csv into the current directory
  import tempfile
  os.chdir(tempfile.mkdtemp(prefix="sgw_fixtures_"))
  sys.dont_write_bytecode = True
  sys.path.insert(0, "/root/reference")
  sys.path.insert(0, os.path.join(HERE, "standins"))
  root = "/root/reference/ai_safety_gridworlds/experiments"
  presets = {}
  for sub, pkg in (("", "experiments"), ("aintelope", "experiments.aintelope")):
    d = os.path.join(root, sub)
    for fn in sorted(os.listdir(d)):
      if not fn.endswith(".py") or fn.startswith("__init__"):
        continue
      name = fn[:-3]
      mod = importlib.import_module("ai_safety_gridworlds." + pkg + "." + name)
      base_mod = "aintelope_savanna" if sub == "aintelope" else "island_navigation_ex"
      base = importlib.import_module("ai_safety_gridworlds.environments." + ("aintelope." if sub else "") + base_mod)
      defaults = flag_values(base.define_flags())
      values = flag_values(mod.init_experiment_flags())
      diff = {k: v for k, v in values.items() if defaults.get(k, "<absent>") != v and not k.endswith("_flags_defined") and k != "eval"}
      assert name not in presets, name
      presets[name] = {"base": base_mod, "package": pkg, "flags": diff}
      print("%-50s %-22s %2d flags" % (pkg + "." + name, base_mod, len(diff)))
  json.dump(presets, open(OUT, "w"), indent=1, sort_keys=True)
  print("wrote", OUT)


if __name__ == "__main__":
  main()
