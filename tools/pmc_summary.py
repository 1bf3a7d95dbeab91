#!/usr/bin/env python3
"""Collapse the rocprofv3 output of tools/pmc_collect.sh into one JSON: per kernel the average duration
(kernel trace) and the per-dispatch average of every collected counter (summed over the counter's dimensions)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir = sys.argv[1]
res = {"kernels_ms": {}, "counters": {}}
for f in glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        res["kernels_ms"][row["Name"]] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6,
                                          "pct": float(row["Percentage"])}
for f in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(float))   # (kernel, counter) -> dispatch -> value
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"], row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (k, cn), d in acc.items():
        if not k.startswith("void qs::") and "qs::" not in k:
            continue
        res["counters"].setdefault(cn, {})[k.split("(")[0]] = {"dispatches": len(d), "avg_per_dispatch": sum(d.values()) / len(d)}
json.dump(res, sys.stdout, indent=1)
