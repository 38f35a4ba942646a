#!/usr/bin/env python3
"""Lists the kernels of one step (between two Adam launches) of a rocprofv3 kernel trace with start, gap and duration."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_shard/s_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
seg = rows[idx[-3] + 1:idx[-2] + 1]
t0 = int(seg[0]["Start_Timestamp"]); prev = None; tot = 0.0
for r in seg:
  s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
  print("%8.1f us +%6.1f dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3 if prev else 0, (e - s) / 1e3, r["Kernel_Name"][:60]))
  prev = e; tot += (e - s) / 1e3
print("sum of durations %.1f us, span %.1f us, %d kernels" % (tot, (prev - t0) / 1e3, len(seg)))
