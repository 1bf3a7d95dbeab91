#!/usr/bin/env python3
"""Collapse the rocprofv3 output of tools/pmc_collect.sh into one JSON.

kernels_ms : per kernel the call count and average duration (kernel trace of the stats pass)
counters   : per counter and kernel the per-dispatch average (summed over the counter's dimensions)
entries    : what bench.py attaches to its roofline object -- one entry for the count kernel of the profiled
             workload: the workload key and kernel variant bench.py printed in that run, the sha of the kernel sources
             the run was built from, SQ_INSTS_VALU per launch, and the HBM bytes per launch = 2 x FETCH_SIZE +
             WRITE_SIZE (KB -> bytes; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).
"""
import csv
import datetime
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

out_dir = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {"kernels_ms": {}, "counters": {}, "entries": []}
prev = os.path.join(out_dir, "summary.json")
if len(sys.argv) > 2 and sys.argv[2] == "--resummarise" and os.path.exists(prev):   # CSVs gone: start from the old summary
    old = json.load(open(prev))
    res["kernels_ms"], res["counters"] = old["kernels_ms"], old["counters"]
    old_entry = (old.get("entries") or [{}])[0]
else:
    old_entry = {}
for f in glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        res["kernels_ms"][row["Name"]] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6,
                                          "pct": float(row["Percentage"])}
for f in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(float))   # (kernel, counter) -> dispatch -> value
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"], row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (k, cn), d in acc.items():
        if "qs::" not in k:
            continue
        res["counters"].setdefault(cn, {})[k.split("(")[0]] = {"dispatches": len(d), "avg_per_dispatch": sum(d.values()) / len(d)}

# the bench line of the stats pass names the workload and the kernel variant
bench = None
try:
    with open(os.path.join(out_dir, "stats.log")) as f:
        for line in f:
            if line.startswith("{") and '"metric"' in line:
                bench = json.loads(line)
except (OSError, ValueError):
    pass
if bench:
    h = hashlib.sha256()
    for fn in ("qs_count.hip", "qs_bitslice3.hpp", "qs_count_fused.hip", "qs_common.hpp"):   # = bench.py kernel_source_sha
        with open(os.path.join(root, "quartetscores_amd", "csrc", fn), "rb") as f:
            h.update(f.read())
    kname = bench["roofline"]["kernel"]

    # every instance of the count kernel that works on the workload's table type (the depth classes of one batch run
    # different template instances; bench.py's lookup gate runs one launch on a scratch u16 table): per-launch average
    # over all of them, weighted by their dispatch counts
    cell = "unsigned int" if bench["dtype"] == "u32" else "unsigned short"

    def mine(name):
        return kname in name and (cell + ">") in name.replace(" >", ">")

    def avg(counter):
        tot = cnt = 0.0
        for k, v in res["counters"].get(counter, {}).items():
            if mine(k):
                tot += v["avg_per_dispatch"] * v["dispatches"]; cnt += v["dispatches"]
        return tot / cnt if cnt else None
    fetch, write, valu = avg("FETCH_SIZE"), avg("WRITE_SIZE"), avg("SQ_INSTS_VALU")
    hit, miss = avg("TCC_HIT_sum"), avg("TCC_MISS_sum")
    kms_all = [v for k, v in res["kernels_ms"].items() if mine(k)]
    kms = [{"avg_ms": sum(v["avg_ms"] * v["calls"] for v in kms_all) / sum(v["calls"] for v in kms_all),
            "calls": sum(v["calls"] for v in kms_all)}] if kms_all else []
    res["entries"].append({
        "workload_key": bench["config"]["workload_key"], "variant": bench["config"]["algo"],
        "kernel_source_sha": old_entry.get("kernel_source_sha") or h.hexdigest()[:16], "kernel": kname,
        "collected": old_entry.get("collected") or datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
        "rocprof_avg_launch_ms": kms[0]["avg_ms"] if kms else None, "rocprof_calls": kms[0]["calls"] if kms else None,
        "bench_avg_launch_ms": bench["roofline"]["avg_launch_ms"],
        "valu_insts_per_launch": valu, "fetch_size_kb": fetch, "write_size_kb": write,
        "hbm_bytes_per_launch": (2 * fetch + write) * 1024 if fetch is not None and write is not None else None,
        "l2_hit": hit / (hit + miss) if hit is not None and miss else None,
    })
    res["bench_line_of_stats_pass"] = bench
json.dump(res, sys.stdout, indent=1)
