#!/bin/bash
# L1 / L2 request counters of a stand-alone micro-benchmark: tools/pmc_bin.sh <tag> <binary> [args...]  (through gpurun from the repo root)
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
BIN=$ROOT/$1; shift
cd /tmp
for grp in "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    name=$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$OUT/pmc_$name" -o run -- "$BIN" "$@" > "$OUT/pmc_$name.log" 2>&1 || echo "pmc group failed: $grp" >> "$OUT/errors.txt"
done
cd "$ROOT"
python3 - "$OUT" <<'PY' > "$OUT/counters.txt"
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float); key_of = {}
    for r in csv.DictReader(open(f)):
        d = (r["Dispatch_Id"], r["Counter_Name"])
        per[d] += float(r["Counter_Value"])
        key_of[r["Dispatch_Id"]] = (r["Kernel_Name"][:70], r.get("Grid_Size"), r.get("Workgroup_Size"))
    for (d, c), v in per.items(): agg[key_of[d]][c].append(v)
for k in sorted(agg):
    c = {n: min(v) for n, v in agg[k].items()}     # (min over the repeats = the best run)
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    req, lat = c.get("TCP_TCC_READ_REQ_sum", 0), c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0)
    line = f"{k[0]} grid {k[1]} wg {k[2]}: cycles/XCD {g:.4g} req {req:.4g} lat/req {lat / max(req, 1):.0f} outstanding/TCP {lat / max(256 * g, 1):.1f}"
    line += f" pend_stall {c.get('TCP_PENDING_STALL_CYCLES_sum', 0) / max(c.get('TCP_GATE_EN1_sum', 1), 1):.2f} acc {c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0):.4g} tcc hit/miss {c.get('TCC_HIT_sum', 0):.4g}/{c.get('TCC_MISS_sum', 0):.4g} tcc_req {c.get('TCC_REQ_sum', 0):.4g} ea_rd {c.get('TCC_EA0_RDREQ_sum', 0):.4g}"
    print(line)
PY
find "$OUT" -type f ! -name "*.log" ! -name "*.txt" -delete
cat "$OUT/counters.txt"
