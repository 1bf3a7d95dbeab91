"""Tree-sharded against table-sharded multi-GPU counting, settled on ONE GPU.

    python tools/scaling_model.py [--configs 2,3] [--ns 2,4,8] [--steps 3] [--out profiles/r06_scaling_model.json]

With the gather kernel the work of a launch is (trees) x (table cells the context owns), so a rank of an N-GPU job can be cut
either way at the same count work:
  * tree-sharded: m/N trees into a private FULL table, then one collective on the table (reduce-scatter, distributed.py);
  * table-sharded: ALL trees into the ranks whose largest taxon id lies in this rank's [d_lo, d_hi) (contiguous because the
    rank's leading term is C(s3,4): /root/reference/src/quartet_lookup_table.hpp:161-165) -- no table collective at all, only
    the panel build is replicated.
Every rank's step of both modes is a single-GPU computation, so this tool runs each of them in turn on cuda:0: for N in --ns
and k = 0..N-1 shard k of N (all trees: panel build + count kernel + corrections, then the sharded scoring passes), and rank
0's and rank N-1's tree shares into the full table. What one GPU cannot measure is the collective of the tree-sharded mode; it is
MODELLED from the bytes a reduce-scatter moves per rank and the xGMI link rate of MI355X_MICROARCH.md (153 GB/s per link, 7
links), as two bounds: a ring (every byte crosses one link per step, N-1 steps) and the full mesh (all peers' links at once).
The JSON says which numbers are measured and which are the model. Timings: HIP events inside qs_count_batch (QS_COUNT_TIMED)
for the kernels, wall clock around synchronised steps for ms_per_step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {1: (128, 1000), 2: (512, 10000), 3: (256, 100000)}
XGMI_LINK_GBS = 153.0            # MI355X_MICROARCH.md: per link and direction; 7 links per GPU (one to every peer of an 8-GPU node)


def collective_model(table_tuples, total_trees, n, binary_full=True):
    """Bytes a reduce-scatter of the table moves out of one rank, and two time bounds for it (MODEL, not measured)."""
    if total_trees < 65536:
        wire, bpt = ("u16x2", 4) if binary_full else ("u16", 6)
    else:
        wire, bpt = ("u32x2", 8) if binary_full else ("u32", 12)
    size = table_tuples * bpt
    sent = size * (n - 1) / n                                  # per rank, reduce-scatter
    return {"wire": wire, "wire_bytes": size, "sent_bytes_per_rank": sent,
            "ring_ms": sent / (XGMI_LINK_GBS * 1e9) * 1e3,     # N-1 steps of size/N over ONE link each
            "mesh_ms": (size / n) / (XGMI_LINK_GBS * 1e9) * 1e3,   # every peer's chunk over its own link at once
            "all_reduce_factor": 2.0,
            "note": "MODEL: reduce-scatter bytes / 153 GB/s per xGMI link; ring = one link at a time, mesh = all peers' links"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3")
    ap.add_argument("--ns", default="2,4,8")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--trees", type=int, default=0, help="override the tree count (smoke runs)")
    ap.add_argument("--taxa", type=int, default=0)
    ap.add_argument("--balance", choices=["c4", "cost"], default="cost", help="how the shard bounds are chosen (distributed.shard_of_largest_id / qs_shard_bounds)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_scaling_model.json"))
    args = ap.parse_args()
    import torch
    from quartetscores_amd import distributed, engine, flatten, native_ingest, ranks

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream(dev)
    algo = engine.QS_ALGO_GATHER | engine.QS_COUNT_OVERWRITE
    doc = {"what": "per-rank steps of table-sharded and tree-sharded counting, each run alone on one MI355X",
           "device": torch.cuda.get_device_name(0), "steps_timed": args.steps, "balance": args.balance, "configs": {}}

    def time_steps(ctx, hb, steps):
        ctx.count_batch(hb, algo)                        # warm-up (code objects, panel allocation, tile order)
        ctx.sync()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(steps):
            ctx.count_batch(hb, algo | (engine.QS_COUNT_TIMED if k == steps - 1 else 0))
        ctx.sync()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 1e3 / steps
        panel, count, _tot = ctx.last_count_ms()
        return {"ms_per_step": round(ms, 3), "panel_ms": round(panel, 3), "count_ms": round(count, 3),
                "fix_ms": round(ctx.last_count_fix_ms(), 3), "launches": ctx.last_count_launches(), "variant": ctx.last_count_variant()}

    for cfg_no in [int(x) for x in args.configs.split(",")]:
        n, m = CONFIGS[cfg_no]
        n, m = args.taxa or n, args.trees or m
        nq = ranks.n_quartets(n)
        ref_nw = native_ingest.synth_trees(n, 1, 1000 * cfg_no).decode().strip()
        ref = flatten.flatten_reference(ref_nw)
        text = native_ingest.synth_trees(n, m, 1000 * cfg_no + 1)
        batch, _ = native_ingest.ingest_text(ref_nw, text, 0, m)
        entry = {"taxa": n, "trees": m, "quartets": nq, "units_per_step": m * nq, "table": {}, "tree": {}}
        print(f"configs[{cfg_no}]: {n} taxa x {m} trees", flush=True)

        def run_ctx(d_lo, d_hi, b, score, shard_index=None, shards=1):
            ctx = engine.Context(n, 32, device=0, stream=stream.cuda_stream, d_lo=d_lo, d_hi=d_hi)
            ctx.table_alloc()
            hb = ctx.batch_upload(b, with_nodes=False)
            r = time_steps(ctx, hb, args.steps)
            r.update({"d_lo": d_lo, "d_hi": d_hi, "tuples": ctx.table_tuples, "table_bytes": ctx.table_bytes, "trees": b.n_trees})
            if score:
                ts = []
                for _ in range(3):
                    torch.cuda.synchronize(dev)
                    s0 = time.perf_counter()
                    if shards > 1:
                        distributed.score_table_shards(lambda k: ctx, [shard_index], ref, device=dev)
                    else:
                        ctx.score(ref)
                    ts.append((time.perf_counter() - s0) * 1e3)
                r["score_ms"] = round(min(ts), 3)
            ctx.batch_free(hb)
            ctx.close()
            torch.cuda.synchronize(dev)
            return r

        one = run_ctx(0, n, batch, True)
        one["value"] = m * nq / (one["ms_per_step"] * 1e-3)
        entry["n1"] = one
        print("  N=1:", one["ms_per_step"], "ms, score", one.get("score_ms"), flush=True)
        for N in [int(x) for x in args.ns.split(",")]:
            # ---- table-sharded: every shard with ALL trees ----
            shards = []
            for k in range(N):
                d_lo, d_hi = distributed.shard_of_largest_id(n, N, k, by=args.balance)
                r = run_ctx(d_lo, d_hi, batch, True, shard_index=k, shards=N)
                r["shard"] = k
                shards.append(r)
                print(f"  table N={N} shard {k} d[{d_lo},{d_hi}): {r['ms_per_step']} ms (panel {r['panel_ms']}, count {r['count_ms']}, fix {r['fix_ms']}), score {r.get('score_ms')}", flush=True)
            mx = max(s["ms_per_step"] for s in shards)
            mean = sum(s["ms_per_step"] for s in shards) / N
            entry["table"][str(N)] = {
                "shards": shards, "max_ms": mx, "mean_ms": round(mean, 3), "imbalance": round(mx / mean - 1.0, 4),
                "panel_share_of_max": round(max(s["panel_ms"] for s in shards) / mx, 4),
                "value": m * nq / (mx * 1e-3), "speedup_vs_n1": round(one["ms_per_step"] / mx, 3),
                "efficiency": round(one["ms_per_step"] / mx / N, 4), "collective": None,
                "score_max_ms": max(s.get("score_ms", 0.0) for s in shards),
                "note": "measured: each shard alone on one GPU; an N-GPU job runs them side by side, no table collective"}
            # ---- tree-sharded: rank 0's and rank N-1's share into the full table ----
            shares = []
            for r_ in sorted({0, N - 1}):
                lo, hi = distributed.shard_range(m, N, r_)
                s = run_ctx(0, n, batch.slice(lo, hi), False)
                s["rank"] = r_
                shares.append(s)
                print(f"  tree  N={N} rank {r_} trees [{lo},{hi}): {s['ms_per_step']} ms", flush=True)
            cm = collective_model(nq, m, N)
            cnt = max(s["ms_per_step"] for s in shares)
            entry["tree"][str(N)] = {
                "shares": shares, "count_max_ms": cnt, "collective_model": cm,
                "step_ms_overlapped": {"ring": round(max(cnt, cm["ring_ms"]), 3), "mesh": round(max(cnt, cm["mesh_ms"]), 3)},
                "step_ms_serial": {"ring": round(cnt + cm["ring_ms"], 3), "mesh": round(cnt + cm["mesh_ms"], 3)},
                "value_overlapped": {"ring": m * nq / (max(cnt, cm["ring_ms"]) * 1e-3), "mesh": m * nq / (max(cnt, cm["mesh_ms"]) * 1e-3)},
                "value_serial": {"ring": m * nq / ((cnt + cm["ring_ms"]) * 1e-3), "mesh": m * nq / ((cnt + cm["mesh_ms"]) * 1e-3)},
                "note": "count measured (one rank's share alone on one GPU); collective = MODEL; bench.py overlaps the collective of step k with step k+1 (a one-shot CLI run cannot)"}
        doc["configs"][str(cfg_no)] = entry
        del batch, text
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
