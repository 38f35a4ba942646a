#!/usr/bin/env python3
"""Per-kernel launch durations from a rocprofv3 --kernel-trace CSV, with the gated no-op launches of the range-guarded
kernels (second launch of a call, returns at once) kept apart from the real ones.

usage: trace_summarize.py OUT.json KERNEL_TRACE.csv
"""
import csv
import json
import re
import sys


def short(name):
  name = re.sub(r"^void ", "", name)
  return re.sub(r"\(.*$", "", name)


def main():
  out_path, trace = sys.argv[1], sys.argv[2]
  durs = {}
  with open(trace, newline="") as f:
    for row in csv.DictReader(f):
      k = short(row["Kernel_Name"])
      if not k.startswith("oivae::"):
        continue
      durs.setdefault(k, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
  res = {}
  for k, v in sorted(durs.items()):
    top = max(v)
    real = [x for x in v if x >= 0.1 * top]
    noop = [x for x in v if x < 0.1 * top]
    res[k] = {"launches": len(real), "avg_ms": sum(real) / len(real), "median_ms": sorted(real)[len(real) // 2],
              "min_ms": min(real), "max_ms": max(real), "launch_ms": [round(x, 4) for x in real],
              "gated_noop_launches": len(noop), "gated_noop_avg_ms": (sum(noop) / len(noop)) if noop else None}
  with open(out_path, "w") as f:
    json.dump(res, f, indent=1, sort_keys=True)
  for k, e in sorted(res.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"])[:8]:
    print("%-40s %3d x avg %8.4f median %8.4f ms  (+%d gated no-ops)" % (k[:40], e["launches"], e["avg_ms"], e["median_ms"], e["gated_noop_launches"]))


if __name__ == "__main__":
  main()
