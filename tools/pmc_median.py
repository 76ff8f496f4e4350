#!/usr/bin/env python3
"""Median per-launch PMC values of one kernel from a rocprofv3 --pmc output dir: tools/pmc_median.py <dir> <kernel substring> [<kind tag>]"""
import csv, glob, statistics, sys
d, sub = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "Li0E"
acc = {}
for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if sub in n and (tag in n or ", 0>" in n):
      acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
  print("%-24s median %14.1f  (n=%d)" % (k, statistics.median(v[len(v) // 5:]), len(v)))
