#!/usr/bin/env python3
"""Kernel durations and launch-to-launch intervals of the step kernel from a rocprofv3 --kernel-trace CSV."""
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_engine" in r["Kernel_Name"] and ", 0>" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
iv = [b - a for a, b in zip(st, st[1:])]
n = len(d)
def med(x): return statistics.median(x) if x else float("nan")
w = int(sys.argv[2]) if len(sys.argv) > 2 else n // 10
print("kernels", n, "warmup(no accumulate) median dur ns", med(d[2:w]), "timed median dur ns", med(d[w + 5:]),
      "timed median start-to-start ns", med(iv[w + 5:]), "min dur", min(d), "VGPR", rows[-1]["VGPR_Count"], "LDS", rows[-1]["LDS_Block_Size"])
