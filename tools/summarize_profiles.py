#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/: kernel stats CSV copy, PMC table,
r<NN>_traffic.json (HBM bytes per launch of the step kernel, corrected as MI355X_MICROARCH.md §HBM prescribes)."""
import csv, glob, json, os, shutil, statistics, sys

tag, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r01")
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
KERNEL = "Island"

def counters(d):
  # gpurun_out/ is scratch that survives rounds: a tag reused from an earlier round still holds that round's CSVs -> newest file
  f = sorted(glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
  acc = {}
  if not f:
    return acc
  for r in csv.DictReader(open(f[-1])):
    if KERNEL in r["Kernel_Name"] and "Li0E" in r["Kernel_Name"] or (KERNEL in r["Kernel_Name"] and "K_STEP" in r["Kernel_Name"]) or (KERNEL in r["Kernel_Name"] and ", 0>" in r["Kernel_Name"]):
      acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
  return {k: statistics.median(v[len(v) // 5:]) for k, v in acc.items()}

pm = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
  pm.update(counters(d))
stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if stats:
  shutil.copy(stats[-1], os.path.join("profiles", "%s_%s_kernel_stats.csv" % (rnd, tag)))
bench_dst = os.path.join("profiles", "%s_%s_bench.json" % (rnd, tag))
shutil.copy(os.path.join(src, "bench.json"), bench_dst)
out = {"tag": tag, "kernel": "sgw::k_engine<sgw::Island, K_STEP> (65536 envs, 1024 waves)", "pmc_median_per_launch": pm}
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
  # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly half of a wide coalesced streaming read
  # (MI355X_MICROARCH.md §HBM) -> doubled.  Our state loads are 8 B/lane (uncalibrated width): both figures are kept.
  out["fetch_kib_raw"] = pm["FETCH_SIZE"]; out["write_kib"] = pm["WRITE_SIZE"]
  out["hbm_bytes_per_launch"] = (2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024
  out["hbm_bytes_per_launch_uncorrected"] = (pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024
  out["algorithmic_bytes_per_launch"] = 299 * 65536
  out["layout"] = "80-byte packed i16 state, pairs of words per lane (16-byte accesses)"
json.dump(out, open(os.path.join("profiles", "%s_%s_pmc.json" % (rnd, tag)), "w"), indent=1)
if "hbm_bytes_per_launch" in out:
  json.dump({"hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "source": "%s_%s_pmc.json" % (rnd, tag)},
            open(os.path.join("profiles", "%s_traffic.json" % rnd), "w"))
for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
  shutil.copy(f, os.path.join("profiles", "%s_%s_%s" % (rnd, tag, os.path.basename(f))))
if os.path.exists(os.path.join(src, "phase_stamps.txt")):
  shutil.copy(os.path.join(src, "phase_stamps.txt"), os.path.join("profiles", "%s_%s_phase_stamps.txt" % (rnd, tag)))
for wl, kname in (("firemaker_ex_ma", "Firemaker"), ("aintelope_savanna", "Savanna")):
  fm = {}
  for f in sorted(glob.glob(os.path.join(src, "pmc_" + wl, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
    acc = {}
    for r in csv.DictReader(open(f)):
      if kname in r["Kernel_Name"] and (", 0>(" in r["Kernel_Name"] or "Li0E" in r["Kernel_Name"]):
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    fm = {k: statistics.median(v[len(v) // 5:]) for k, v in acc.items()}
  if fm:
    body = {"kernel": "sgw::k_engine<sgw::%s, K_STEP> (bench.py --workload %s)" % (kname, wl), "tag": tag, "pmc_median_per_launch": fm}
    json.dump(body, open(os.path.join("profiles", "%s_%s_pmc_%s.json" % (rnd, tag, wl)), "w"), indent=1)
    json.dump(body, open(os.path.join("profiles", "%s_pmc_%s.json" % (rnd, wl)), "w"), indent=1)     # the copy bench.py reads
for extra in ("side", "firemaker", "mixed"):            # round 3: rocprofv3 kernel stats of the side kernels / firemaker with windows / the group launch
  st = sorted(glob.glob(os.path.join(src, "stats_" + extra, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
  if st:
    shutil.copy(st[-1], os.path.join("profiles", "%s_%s_kernel_stats_%s.csv" % (rnd, tag, extra)))
for f in ("side_probe.json", "side_probe.txt", "vec_probe.txt", "host_probe.txt", "agent_views_probe.txt", "zoo_vector_probe.txt"):
  if os.path.exists(os.path.join(src, f)):
    shutil.copy(os.path.join(src, f), os.path.join("profiles", "%s_%s_%s" % (rnd, tag, f)))
if "hbm_bytes_per_launch" in out:      # the bench ran before this tag's PMC passes were summarised: carry their traffic figure
  line = json.loads(open(bench_dst).read().strip().splitlines()[-1])
  line["roofline"]["traffic"] = out["hbm_bytes_per_launch"]
  open(bench_dst, "w").write(json.dumps(line) + "\n")
print(json.dumps(out, indent=1))
