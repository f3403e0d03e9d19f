#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs (one directory per pass) into per-kernel averages."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "ffsa::" not in k:
            continue
        name = k.split("ffsa::")[1].split("(")[0]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        acc[name]["_VGPR"].append(float(row.get("VGPR_Count", 0) or 0))
        acc[name]["_LDS"].append(float(row.get("LDS_Block_Size", 0) or 0))
for name, cs in acc.items():
    print("==", name)
    for c in sorted(cs):
        v = cs[c]
        print("   %-28s avg %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
