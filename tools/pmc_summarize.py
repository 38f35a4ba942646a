#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes into one JSON of per-kernel averages.

usage: pmc_summarize.py OUT.json DIR [DIR ...]

Every DIR is the -d output directory of one `rocprofv3 --pmc ... --output-format csv`
pass (tools/pmc_collect.sh).  Counters are averaged over the launches of each kernel
of the `oivae::` namespace.  HBM bytes follow MI355X_MICROARCH.md section HBM:
FETCH_SIZE is in KiB and reports half of the bytes of wide coalesced reads on
gfx950 (so x1024 x2); WRITE_SIZE x1024 is exact for 16-byte stores and float atomics.
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
  name = re.sub(r"^void ", "", name)
  return re.sub(r"\(.*$", "", name)


def main():
  out_path, dirs = sys.argv[1], sys.argv[2:]
  acc = {}
  for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
      with open(path, newline="") as f:
        for row in csv.DictReader(f):
          k = short(row["Kernel_Name"])
          if not k.startswith("oivae::"):
            continue
          acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
          meta = acc[k].setdefault("_meta", {})
          meta["VGPR_Count"] = int(row["VGPR_Count"])
          meta["Accum_VGPR_Count"] = int(row["Accum_VGPR_Count"])
          meta["SGPR_Count"] = int(row["SGPR_Count"])
          meta["LDS_Block_Size"] = int(row["LDS_Block_Size"])
          meta["Scratch_Size"] = int(row["Scratch_Size"])
  res = {}
  for k, counters in sorted(acc.items()):
    # Kernels behind a range guard are launched twice per call; the second launch returns at once unless the guard
    # asked for a repeat (csrc/mfma_decoder.hip range_guard).  Such no-op launches (a counter below 10 % of the
    # kernel's largest launch) are left out of the averages.
    e = {}
    n_used = 0
    for c, v in counters.items():
      if c == "_meta":
        continue
      top = max(v)
      used = [x for x in v if x >= 0.1 * top]
      e[c] = sum(used) / len(used)
      n_used = max(n_used, len(used))
      e.setdefault("launches_skipped_as_gated", 0)
      e["launches_skipped_as_gated"] = max(e["launches_skipped_as_gated"], len(v) - len(used))
    e["launches_averaged"] = n_used
    e.update(counters.get("_meta", {}))
    if "FETCH_SIZE" in e:
      e["hbm_read_bytes_corrected"] = e["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in e:
      e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024.0
    res[k] = e
  # stamp: the decoder sources these counters were taken from (bench.py reports `traffic` only for a matching tree)
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import bench
  res["_decoder_source_digest"] = bench.decoder_source_digest()
  with open(out_path, "w") as f:
    json.dump(res, f, indent=1, sort_keys=True)
  print("wrote %s (%d kernels)" % (out_path, len(res)))


if __name__ == "__main__":
  main()
