#!/bin/bash
# A count table that does not fit the GPU: 1200 taxa, u16 cells = 515 GB through one 288 GB MI355X in shards by largest
# taxon id. RANDOM evaluation trees (round 2 used copies of the reference tree: every score 1 -- that would pass with most
# counting bugs). Two checks:
#   1. library level: every shard is counted the way `QuartetScores --table-shards K` counts it (qs_create with
#      [d_lo, d_hi), all trees) and 10 000 random quartets whose largest id lies in the shard are looked up (qs_lookup) and
#      compared with the split-based brute force of tests/bruteforce.py on the same trees;
#   2. product level: the CLI with --table-shards 0 (as many shards as the free device memory asks for, finished shards
#      counted again for the second scoring pass) against the CLI with one shard more (other cut points): the annotated
#      trees must be identical.
# (The scores themselves are compared with the oracle at sizes the oracle can do: tests/test_cli.py.)
set -u
N=${1:-1200}; M=${2:-40}
D=$(mktemp -d)
python3 - "$N" "$M" "$D" <<'PY'
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import bruteforce
from quartetscores_amd import distributed, engine, flatten, native_ingest, ranks
n, m, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ref_nw = native_ingest.synth_trees(n, 1, 5000).decode().strip()
text = native_ingest.synth_trees(n, m, 5001)
open(d + "/ref.nwk", "w").write(ref_nw + "\n")
open(d + "/eval.nwk", "wb").write(text)
ref = flatten.flatten_reference(ref_nw)
batch, _ = native_ingest.ingest_text(ref_nw, text, 0, m, want_ranges=False)
trees = [ln.decode() for ln in text.split(b"\n") if ln.strip()]
import torch
free_b, _tot = torch.cuda.mem_get_info(0)
table_bytes = ranks.n_quartets(n) * 6
K = max(2, int(np.ceil(table_bytes / (0.70 * free_b))))
print(f"{n} taxa, {m} random trees: table {table_bytes / 1e9:.0f} GB, device free {free_b / 1e9:.0f} GB -> {K} shards")
rng = np.random.default_rng(7)
bad = total = 0
for k in range(K):
    d_lo, d_hi = distributed.shard_of_largest_id(n, K, k)
    t0 = time.perf_counter()
    ctx = engine.Context(n, 16, d_lo=d_lo, d_hi=d_hi)
    ctx.table_alloc()
    ctx.count_trees(batch, engine.QS_ALGO_GATHER)
    ctx.sync()
    tops = rng.integers(max(d_lo, 3), d_hi, size=10000)          # the largest id of every probe lies in this shard
    q = np.sort(np.stack([np.append(rng.choice(int(t), size=3, replace=False), t) for t in tops]), axis=1).astype(np.uint16)
    got = ctx.lookup(q)
    want = bruteforce.quartet_counts_for(trees, ref.names, q.astype(np.int64))
    nb = int((got != want).any(axis=1).sum())
    bad += nb; total += len(q)
    print(f"shard {k}: largest id in [{d_lo},{d_hi}), {ctx.table_bytes / 1e9:.0f} GB, {ctx.last_count_variant()}, {time.perf_counter() - t0:.1f} s; "
          f"{len(q)} lookups against the brute force: {nb} differ")
    ctx.close()
print(f"lookups checked: {total}, differing: {bad}")
PY
for extra in "--table-shards 0" "--table-shards 4"; do
  rm -f $D/out.nwk
  T0=$SECONDS
  quartetscores_amd/bin/QuartetScores -r $D/ref.nwk -e $D/eval.nwk -o $D/out.nwk $extra --spill recount 2>&1 | grep -vE "^Counting quartets|^Note:|^      "
  echo "CLI $extra: $((SECONDS - T0)) s wall"
  cp $D/out.nwk "$D/out_$(echo $extra | tr -d ' -').nwk"
done
python3 - "$D" <<'PY'
import re, sys
a = open(sys.argv[1] + "/out_tableshards0.nwk").read(); b = open(sys.argv[1] + "/out_tableshards4.nwk").read()
vals = re.findall(r"(qp-ic|lq-ic|eqp-ic):([-0-9.e+]+)", a)
print("annotated internodes:", len(vals) // 3, "| automatic shard count and 4 shards give identical files:", a == b,
      "| distinct lq-ic values:", len({v for k, v in vals if k == "lq-ic"}))
PY
rm -rf $D
