#!/usr/bin/env python3
"""bench.py -- quartets counted per second on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C]

`--gpus N` with N > 1 (or `--via-launcher`) and no WORLD_SIZE in the environment: this process is only the PARENT. Before any
GPU call it starts one fresh child `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py <same arguments>`, relays rank 0's JSON line as its own last line and exits with the child's code (no
retry; `--dry-launch` prints the command). Started by torch.distributed.run itself (WORLD_SIZE set) it is a rank.

Workloads (BASELINE.json `configs`; seeded synthetic trees from csrc/host/synth.hpp):
    --config 1   configs[1]: 128 taxa x 1 000 trees, u32 table (128 MB)
    --config 2   configs[2]: 512 taxa x 10 000 trees, u32 table (34 GB)            <- default at EVERY N (one workload per driver curve)
    --config 3   configs[3]: 256 taxa x 100 000 trees, u32 table (2.1 GB)
    --config 4   configs[4]: 1024 taxa x 5 000 trees, u16 table sharded by the largest taxon id over max(N, 8) shards;
                 every rank counts all trees into its shard(s), no table collective
    --taxa/--trees/--count-bits override the sizes (the workload label then says "custom"); --dropout / --collapse / --mixed
    make binary trees with missing taxa / multifurcating trees / a third of each (numpy generator, small tree counts).
N > 1: --mode table = every rank counts ALL trees into its shard of the table (by largest taxon id, cost-balanced bounds:
qs_shard_bounds), no table collective; --mode tree = trees / N per rank into a full table + ONE RCCL collective on the table per step;
--mode auto (default) = by the model of the one-GPU measurements (distributed.auto_mode: table for configs[2], tree for configs[3]).
The line carries a few steps of the OTHER mode on the same trees as config.other_mode_leg.

A step = one pass of the hot path over the rank's batch of trees, which is already resident in HBM (the task's measurement
contract; the upload-inclusive step is reported as e2e.upload_in_step_ms, +1 %): build the pair-depth panel and run the count
kernel class by class and slice by slice (first slice stores, the others accumulate), then for N > 1 the collective on the
table. Exactly K steps are timed between barrier + torch.cuda.synchronize() on both sides; value = quartet-tree units counted
by all ranks / max-over-ranks time. N > 1: `collective` = {ranks, proof (all-reduce of ones), comm_init_ms,
collective_alone_ms}; the cpu_baseline leg runs at N = 1 only (task statement 4; `--cpu-baseline-at-n` forces it on rank 0 of an N > 1 run).

roofline: the dominant kernel is the count kernel. The gather formulation keeps every counter in a
register and writes each table cell once per panel slice, so it is bound by VALU issue, not by HBM:
    achieved = ALGORITHMIC vector lane-operations per launch (the minimal compare chain of the bit-sliced four-point
               test per class: (2(B+1)+2)/32 per tree x quartet for full binary trees with B depth bits, 2(B+1)+5 with
               missing taxa, 3(B+1)+3 / +7 for multifurcating trees: ops_of) / the kernel's average launch duration, measured live
               with HIP events on the launch stream around every launch of the LAST timed step (qs_last_count_ms);
    peak     = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz (MI355X_MICROARCH.md: one wave64 VALU instruction per 2 cycles
               per SIMD = the 157.3 TFLOP/s fp32 vector peak / 2 flops).
`issued` (instructions really issued, from rocprofv3 SQ_INSTS_VALU) and `traffic` (HBM bytes from FETCH_SIZE /
WRITE_SIZE) come from the PMC summary under profiles/ ONLY when that summary was collected for this workload, this
kernel variant and this kernel source (sha256 of qs_count.hip); otherwise they are null. `hbm_algorithmic_ratio` is
SURVEY.md 8(d)'s figure: the RMW bytes of the reference formulation (2 x sizeof(counter) per unit) / kernel time /
8 TB/s -- a speed-up over a perfect scatter implementation, not a fraction.
cpu_baseline: the oracle (CPU restatement of the reference, kind "port") timed in a child process on this host on a
bounded sample of the same trees at -t 1 and at the best thread count; reported, never the target.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BACKEND = os.environ.get("QS_BENCH_BACKEND") or os.environ.get("QS_DIST_BACKEND", "nccl")   # "gloo": several ranks share one GPU, collectives staged through the host (tests only)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TLOPS = 256 * 4 * 32 * 2.4e9 / 1e12   # 78.6 T lane-ops/s: one wave64 VALU op per 2 cycles per SIMD-32

CONFIGS = {
    1: dict(taxa=128, trees=1000, bits=32, shards=1, split=False),
    2: dict(taxa=512, trees=10000, bits=32, shards=1, split=False),
    3: dict(taxa=256, trees=100000, bits=32, shards=1, split=True),
    4: dict(taxa=1024, trees=5000, bits=16, shards=8, split=False),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = automatic: about 6 s of timed steps, between 5 and 500")
    ap.add_argument("--warmup", type=int, default=-1, help="-1 = automatic: 2 for steps >= 100 ms, else 20")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4], help="BASELINE.json configs[k]; 0 = 2 (the config the metric is quoted on) at every N")
    ap.add_argument("--prewarm-ms", type=float, default=150.0, help="untimed pre-conditioning before the warm-up steps (GPU clock ramp); 0 = off")
    ap.add_argument("--taxa", type=int, default=0)
    ap.add_argument("--trees", type=int, default=0, help="trees in total (split over the ranks when the config splits), else per rank")
    ap.add_argument("--split-trees", type=int, default=-1, help="1: --trees are split over the ranks (strong scaling); 0: every rank counts --trees of its own")
    ap.add_argument("--algo", choices=["gather", "scatter"], default="gather")
    ap.add_argument("--count-bits", type=int, default=0)
    ap.add_argument("--cpu-budget-s", type=float, default=30.0, help="seconds of CPU counting at -t 1 and at the best thread count of the cpu_baseline leg (the scan in between: a tenth each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-at-n", action="store_true", help="run the cpu_baseline leg on rank 0 of an N > 1 run too (default: N = 1 only)")
    ap.add_argument("--no-score", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the e2e legs (CLI child process; steps with the host-array upload inside)")
    ap.add_argument("--no-impl-check", action="store_true", help="skip the on-device comparison with the byte-SWAR implementation")
    ap.add_argument("--nni", action="store_true", help="evaluation trees = reference tree + Poisson(n/8) random NNIs (concentrated counts)")
    ap.add_argument("--shape", choices=["random", "ladder"], default="random",
                    help="ladder: the reference tree is a caterpillar and the evaluation trees are that ladder + Poisson(n/8) NNIs (deep trees: 9 depth bits at 512 taxa)")
    ap.add_argument("--collapse", type=float, default=0.0, help="collapse each internal edge with this probability (multifurcating trees; numpy generator, small sizes)")
    ap.add_argument("--dropout", type=float, default=0.0, help="drop each taxon from a tree with this probability (partial trees; numpy generator, small sizes)")
    ap.add_argument("--mixed", action="store_true",
                    help="a third of the trees binary and full, a third with --dropout (default 0.1), a third with --collapse (default 0.2), interleaved")
    ap.add_argument("--reduce", choices=["scatter", "all"], default="scatter",
                    help="N>1: reduce-scatter (each rank keeps and scores a shard of the reduced table) or all-reduce")
    ap.add_argument("--wire", choices=["auto", "u16x2", "u16", "u32x2", "u32"], default="auto",
                    help="N>1: format of the table on the wire (auto: fewer than 65536 trees in total: u16x2 for binary full trees, else u16; "
                         "from 65536 trees on: u32x2 = two u32 cells per tuple for binary full trees with reduce-scatter, else u32)")
    ap.add_argument("--mode", choices=["auto", "tree", "table"], default="auto",
                    help="N > 1: tree = every rank counts trees/N into a full table + ONE RCCL collective on the table; table = every rank counts ALL "
                         "trees into its shard of the table (by largest taxon id), no table collective; auto = by the model (auto_mode)")
    ap.add_argument("--balance", choices=["auto", "c4", "cost"], default="auto",
                    help="table mode: shard bounds balanced by the tuples held (c4) or by the count kernel's work (cost: qs_shard_bounds); auto = cost, c4 for configs[4]")
    ap.add_argument("--other-leg", type=int, default=-1, help="N > 1: also time a few steps of the OTHER mode (tree <-> table) in the same run; -1 = on for binary full workloads")
    ap.add_argument("--table-shards", type=int, default=0, help="table-sharded mode: split the table by the largest taxon id into this many shards")
    ap.add_argument("--shard-index", type=int, default=-1, help="table-sharded mode on fewer ranks than shards: which shard this rank owns (default: its rank)")
    ap.add_argument("--slice-bytes", type=int, default=0, help="qs_set_tuning(QS_TUNE_PANEL_SLICE_BYTES); 0 = automatic")
    ap.add_argument("--p2p-leg", type=int, default=-1, help="1: also time the C++ host's peer-access reduction (QuartetScores --gpus N --reduce p2p) in a child before rank 0 uses its GPU; 0: never; -1: at N > 1 on a split workload")
    ap.add_argument("--via-launcher", action="store_true",
                    help="go through the spawn path of --gpus N > 1 even at N = 1 (a fresh torch.distributed.run child; this process never touches the GPU)")
    ap.add_argument("--dry-launch", action="store_true", help="print the child command of the spawn path as one JSON line and exit")
    ap.add_argument("--secondary", type=int, default=-1,
                    help="after the timed region also measure 3 steps each of 512 taxa x 1500 trees with --collapse 0.2 / --dropout 0.1 / --mixed on the resident "
                         "table, and configs[1], one rank's share of configs[3], one shard of configs[4] (config.secondary); -1 = on for the default N = 1 line")
    ap.add_argument("--cpu-child", default="", help=argparse.SUPPRESS)
    ap.add_argument("--gen-child", default="", help=argparse.SUPPRESS)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process is only the PARENT. It never
# makes a HIP call and never imports torch (the GPUs are counted from the KFD topology in sysfs: visible_gpus); it
# starts ONE fresh child `python -m torch.distributed.run ... bench.py <same arguments>`, passes the child's stderr
# through, relays rank 0's JSON line as its own last line of stdout and exits with the child's code. No retry.
# ---------------------------------------------------------------------------------------------------------------
LAUNCH_ONLY_FLAGS = ("--via-launcher", "--dry-launch")


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_command(n_gpus, argv, port=None):
    """The command the driver itself uses for N > 1 (task contract), with this script's own arguments."""
    rest = [a for a in argv if a not in LAUNCH_ONLY_FLAGS]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__)] + rest


def visible_gpus():
    """GPUs this process could use, counted WITHOUT torch and without any HIP / HSA call: the KFD topology in sysfs (a node
    with simd_count > 0 whose render node /dev/dri/renderD<drm_render_minor> this process may open is a GPU), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (index lists;
    UUID entries count as one device each)."""
    import glob
    n = 0
    # (no KFD topology = no amdgpu driver = no GPU: the ROCm runtime enumerates devices from these very files)
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            # a container / cgroup may expose only some of the host's GPUs: the topology lists them all, the render nodes say
            # which ones this process can open (drm_render_minor of the node's properties)
            minor = int(props.get("drm_render_minor", "0"))
            if minor > 0 and os.path.isdir("/dev/dri") and not os.access(f"/dev/dri/renderD{minor}", os.R_OK | os.W_OK):
                continue
            n += 1
        except (OSError, ValueError):
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is not None:
            listed = [x for x in val.split(",") if x.strip() != ""]
            n = min(n, len(listed))
    return n


def launch(args, argv):
    cmd = child_command(args.gpus, argv)
    if args.dry_launch:
        print(json.dumps({"launch": cmd, "n_ranks": args.gpus, "visible_gpus": visible_gpus()}))
        return 0
    have = visible_gpus()
    if have < args.gpus and not (BACKEND == "gloo" and have >= 1):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs on this node, {have} visible "
                         "(no CPU fallback in quartetscores_amd); nothing was launched\n")
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this pool
    env["QS_BENCH_LAUNCHED_BY_PARENT"] = "1"
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, cwd=ROOT)
    line = None
    for ln in p.stdout:                                     # relay progress as it comes, keep the JSON line for the end
        if ln.startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if rc != 0 or line is None:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank child failed (rc {rc}, JSON line {'present' if line else 'absent'})\n")
        return rc or 1
    try:
        doc = json.loads(line)
        doc.setdefault("config", {})["launcher"] = {"spawned_by": "bench.py parent (subprocess, no GPU call in the parent)",
                                                     "child_wall_s": round(time.perf_counter() - t0, 2), "ranks": args.gpus}
        line = json.dumps(doc)
    except ValueError:
        pass
    sys.stdout.flush()
    print(line)
    sys.stdout.flush()
    return 0


# ---------------------------------------------------------------------------------------------------------------
# cpu_baseline leg: runs in a CHILD process (no GPU, only the oracle) so that a host-memory problem with the
# reference's n^4 table can never take the bench line down with it.
# ---------------------------------------------------------------------------------------------------------------
def host_info():
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    mem = 0
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    mem = int(line.split()[1]) * 1024
    except OSError:
        pass
    try:
        with open("/sys/fs/cgroup/memory.max") as f:
            v = f.read().strip()
            if v != "max":
                mem = min(mem, int(v)) if mem else int(v)
    except (OSError, ValueError):
        pass
    import glob
    nodes = len(glob.glob("/sys/devices/system/node/node[0-9]*")) or 1
    return {"host_cpus": os.cpu_count() or 1, "cpu_model": model, "mem_available_bytes": mem, "numa_nodes": nodes,
            "threads_note": "OpenMP threads are not pinned; beyond one NUMA node's cores the n^4 table's remote accesses cost more than the threads add"[:100]}


def numa_node0_cpus():
    """CPUs of NUMA node 0 that this process may use (sysfs cpulist), or None."""
    try:
        with open("/sys/devices/system/node/node0/cpulist") as f:
            txt = f.read().strip()
        cpus = set()
        for part in txt.split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= set(os.sched_getaffinity(0))
        return sorted(cpus) or None
    except (OSError, ValueError, AttributeError):
        return None


def cpu_child(spec_path):
    """Times the oracle on a bounded sample; prints one JSON object. The process is pinned to the CPUs of NUMA node 0 and the
    OpenMP threads are bound to them (OMP_PLACES / OMP_PROC_BIND) BEFORE the oracle's library (libgomp) is loaded: the table is
    first-touched by those threads, so table and threads share one node."""
    with open(spec_path) as f:
        spec = json.load(f)
    # placement "all" (second, short child): no affinity, threads spread over every CPU of the host -- on a multi-socket box, or
    # where spreading suits the n^4 table better, the one-node placement would understate the CPU
    spread = spec.get("placement") == "all"
    pinned = None if spread else numa_node0_cpus()
    if pinned:
        try:
            os.sched_setaffinity(0, pinned)
        except OSError:
            pinned = None
    os.environ.setdefault("OMP_PROC_BIND", "spread" if spread else "close")
    os.environ.setdefault("OMP_PLACES", "threads")
    from oracle_api import Oracle
    with open(spec["trees_path"]) as f:
        text = f.read()
    n, m, nq, budget = spec["n"], spec["m"], spec["nq"], spec["budget_s"]
    info = host_info()
    ncpu = len(pinned) if pinned else info["host_cpus"]
    info["pinned"] = (f"NUMA node 0: {len(pinned)} CPUs ({pinned[0]}..{pinned[-1]}), OMP_PROC_BIND={os.environ['OMP_PROC_BIND']} OMP_PLACES={os.environ['OMP_PLACES']}"
                      if pinned else f"not pinned, all {ncpu} CPUs, OMP_PROC_BIND={os.environ['OMP_PROC_BIND']}" if spread else "not pinned (no sysfs NUMA topology)")
    info["threads_note"] = ("threads spread over all CPUs of the host" if spread else
                            "threads and table on one NUMA node; thread counts beyond the node's CPUs are not run")
    cint_bits = 8 if m < 256 else 16 if m < 65536 else 32               # QuartetScores.cpp:115-147
    fast_bytes = n ** 4 * cint_bits // 8
    # QuartetScoreComputer.hpp:739: the n^4 table unless it exceeds 0.9 x RAM (here: half of what is available, so
    # that the leg stays a guest on the box)
    savemem = bool(info["mem_available_bytes"] and fast_bytes > 0.5 * info["mem_available_bytes"])
    o = Oracle(spec["ref"])
    out = dict(info)
    out.update({"kind": "port", "unit": "quartets/s", "table": "compact table" if savemem else "n^4 table",
                "cint_bits": cint_bits, "runs": []})

    def run(th, secs):
        o.set_budget(secs, True)
        t = o.count(text, savemem=savemem, cint_bits=cint_bits, nthreads=th)
        inc = o.increments_done()
        # the reference increments twice per displayed quartet (SURVEY.md 3.2 iii): units = increments / 2
        r = {"threads": th, "seconds": t, "units": inc / 2.0, "trees_equiv": inc / (2.0 * nq), "value": inc / 2.0 / t}
        out["runs"].append(r)
        return r

    if spread:
        # the short scan only; the full budget is spent here only when the scan beats the pinned placement's best (spec["beat"])
        best = None
        if not savemem:
            for th in sorted({min(t, ncpu) for t in (16, 64, ncpu)} - {1}):
                r = run(th, max(2.0, budget / 10))
                if best is None or r["value"] > best["value"]:
                    best = r
                elif r["value"] < 0.95 * best["value"]:
                    break                                               # past the best thread count: more threads only add remote accesses
            if best is not None and best["value"] > float(spec.get("beat") or 0.0):
                best = run(best["threads"], max(10.0, budget / 2))   # a longer run is the figure, not the 3 s of the scan (half the budget: the default line stays within minutes)
        if best is None:
            out.update({"value": None, "cores": 0, "sample": "not run (the compact table with threads is racy in the reference)"})
            o.close()
            print(json.dumps(out))
            return
        r1 = None
    else:
        r1 = run(1, budget)                                             # -t 1 for the full budget (>= 30 s by default)
        best = r1
        if not savemem:                                                 # savemem + threads is racy in the reference
            scan = None
            for th in sorted({min(t, ncpu) for t in (8, 16, 32, 64, ncpu)} - {1}):   # short scan for the best thread count ...
                r = run(th, max(2.0, budget / 10))
                if scan is None or r["value"] > scan["value"]:
                    scan = r
                elif r["value"] < 0.95 * scan["value"] and th >= 32:
                    break                                               # ... which stops at the first count from 32 threads on that loses 5 % (the default line stays within minutes)
            if scan is not None:
                r = run(scan["threads"], budget)                        # ... which then runs for the full budget
                if r["value"] > best["value"]:
                    best = r
    if r1 is not None:
        out["t1"] = {"value": r1["value"], "cores": 1}
    out["value"], out["cores"] = best["value"], best["threads"]
    # counting is linear in the number of trees: the rate measured on a prefix stands for the whole batch
    out["extrapolated_from_trees"] = round(best["trees_equiv"], 3)
    out["batch_trees"] = m
    out["sample"] = (f"{best['trees_equiv']:.2f} trees' worth of increments of the {m}-tree batch, {out['table']} u{cint_bits}, "
                     f"-t {best['threads']}, {best['seconds']:.1f} s")[:100]
    o.close()
    print(json.dumps(out))


def run_cpu_baseline(ref_nw, eval_text, n, m, nq, budget_s):
    import tempfile
    d = tempfile.mkdtemp(prefix="qsbench_")
    tp, sp = os.path.join(d, "trees.nwk"), os.path.join(d, "spec.json")
    # the child needs only as many trees as it can count in its budget: cap the sample text
    lines = eval_text.split(b"\n") if isinstance(eval_text, bytes) else eval_text.encode().split(b"\n")
    keep = lines[: max(8, min(len(lines), 4096))]
    with open(tp, "wb") as f:
        f.write(b"\n".join(keep))
    spec = {"trees_path": tp, "ref": ref_nw, "n": n, "m": m, "nq": nq, "budget_s": budget_s}
    env = dict(os.environ)
    env.pop("HIP_VISIBLE_DEVICES", None)

    def child(extra):
        with open(sp, "w") as f:
            json.dump(dict(spec, **extra), f)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", sp], capture_output=True, text=True,
                           timeout=120 + 4 * budget_s, env=env)
        last = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not last:
            return {"value": None, "unit": "quartets/s", "cores": 0, "kind": "port", "sample": f"failed (rc {p.returncode}): {p.stderr[-300:]}"}
        return json.loads(last[-1])
    try:
        res = child({})                      # threads and table on NUMA node 0
        if res.get("value") and res.get("numa_nodes", 1) >= 1 and os.environ.get("QS_BENCH_CPU_SPREAD", "1") != "0":
            # one unpinned placement beside it (all CPUs, OMP_PROC_BIND=spread): a short scan, the full budget only if it wins
            try:
                alt = child({"placement": "all", "beat": res["value"]})
            except Exception as e:
                alt = {"value": None, "sample": f"failed: {e}"}
            res["unpinned"] = {k_: alt.get(k_) for k_ in ("value", "cores", "sample", "pinned", "runs")}
            if alt.get("value") and alt["value"] > res["value"]:
                for k_ in ("value", "cores", "sample", "extrapolated_from_trees"):
                    res[k_] = alt.get(k_)
                res["placement_won"] = "unpinned, spread over all CPUs"
            else:
                res["placement_won"] = "pinned to NUMA node 0"
        return res
    except Exception as e:  # the baseline is reported, never required for the metric
        return {"value": None, "unit": "quartets/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    finally:
        for q in (tp, sp):
            try:
                os.remove(q)
            except OSError:
                pass
        try:
            os.rmdir(d)
        except OSError:
            pass


def cli_phases(stdout, stderr):
    """Phases of one QuartetScores run from what it prints: the two "It took: <us> microseconds." lines of the reference's stdout
    protocol (counting, scoring) and, with --trace on a multi-GPU run, when all GPUs had counted and how long the table
    reduction behind that took. None when the protocol lines are missing."""
    import re
    took = [int(x) for x in re.findall(r"It took: (\d+) microseconds", stdout)]
    if len(took) < 2:
        return None
    res = {"counting_phase_ms": took[0] / 1e3, "scoring_phase_ms": took[1] / 1e3}
    stamps = {what.strip(): float(ms) for ms, what in re.findall(r"\[trace\] \+\s*([0-9.]+) ms\s+(main: [^\n]*)", stderr)}
    if "main: all GPUs counted" in stamps and "main: tables reduced" in stamps:
        res["all_gpus_counted_at_ms"] = stamps["main: all GPUs counted"]
        res["table_reduction_ms"] = round(stamps["main: tables reduced"] - stamps["main: all GPUs counted"], 3)
    return res


def same_workload_scaling(value, world, units_per_rank_per_step, count_only_ms, ms_per_step):
    """What an N > 1 line says about scaling by itself: the ranks' own shares counted without the table collective in the same
    run (count_only_ms, max over ranks) against the whole-job value."""
    rate = units_per_rank_per_step / (count_only_ms * 1e-3)
    return {"count_only_ms_per_step": round(count_only_ms, 3), "units_per_rank_per_step": units_per_rank_per_step,
            "rate_quartets_per_s": rate, "scaling_efficiency": value / (world * rate),
            "collective_exposed_ms": round(ms_per_step - count_only_ms, 3),
            "note": "same ranks, shares, run and store path (wire words / pack pass); only the dist.* call is left out (max over ranks)"[:100]}


def run_cli_e2e(ref_nw, eval_text, threads=8, extra=()):
    """The product's own counting phase (QuartetScores CLI: Newick text on disk -> table in HBM) on the same trees, in a
    child process, BEFORE this process touches the GPU. Returns the phases the CLI prints (Appendix A protocol). `extra`:
    more CLI arguments (the peer-access leg: --gpus N --reduce p2p --trace; its trace stamps are parsed as well)."""
    import re
    import tempfile
    exe = os.path.join(ROOT, "quartetscores_amd", "bin", "QuartetScores")
    if not os.path.exists(exe):
        return {"error": "quartetscores_amd/bin/QuartetScores not built"}
    d = tempfile.mkdtemp(prefix="qsbench_cli_")
    rp, ep, op = (os.path.join(d, x) for x in ("ref.nwk", "eval.nwk", "out.nwk"))
    try:
        with open(rp, "w") as f:
            f.write(ref_nw + "\n")
        with open(ep, "wb") as f:
            f.write(eval_text if isinstance(eval_text, bytes) else eval_text.encode())
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-r", rp, "-e", ep, "-o", op, "-t", str(threads)] + list(extra), capture_output=True, text=True, timeout=600)
        wall = time.perf_counter() - t0
        res = cli_phases(p.stdout, p.stderr)
        if p.returncode != 0 or res is None:
            return {"error": f"rc {p.returncode}: {p.stderr[-200:]}"}
        res.update({"process_wall_ms": wall * 1e3, "host_threads": threads, "newick_bytes": os.path.getsize(ep)})
        return res
    except Exception as e:  # reported, never required for the metric
        return {"error": str(e)[:200]}
    finally:
        for q in (rp, ep, op):
            try:
                os.remove(q)
            except OSError:
                pass
        try:
            os.rmdir(d)
        except OSError:
            pass


# ---------------------------------------------------------------------------------------------------------------
# kernel variant -> classes and the minimal instruction count the roofline prices
#   "gather/<mode>/bitslice_b4x2:2358+bitslice_b5x2:7642/count_u32", or classes of several modes:
#   "gather/mixed/binary_full.bitslice_b4x2:120+binary_partial.bitslice_b5x2:80+.../count_u32"
MODES = ("binary_full", "binary_partial", "general_full", "partial")
POPS = {"binary_full": 2, "binary_partial": 3, "general_full": 3, "partial": 3}        # v_bcnt per (quartet, 32 trees)


def ops_of(bits_, mode_):
    """Minimal VALU wave-instructions per (quartet, 32 trees) of the bit-sliced four-point test, DESIGN.md 3.1:
      binary_full:    [L > R] and [L < R] over B+1 planes (2 v_bitop3 per plane) + 2 v_bcnt
      binary_partial: + 1 presence AND + 2 masks + 1 v_bcnt (the trees holding all four; the third topology is that minus the other two)
      general_full:   + [M[ad]-M[cd] > M[ab]-M[bc]] (B+1) + 1 v_bcnt;   partial: + 3 presence masks + the presence AND"""
    return {"binary_full": 2 * (bits_ + 1) + 2, "binary_partial": 2 * (bits_ + 1) + 5,
            "general_full": 3 * (bits_ + 1) + 3, "partial": 3 * (bits_ + 1) + 7}[mode_]


def parse_variant(variant, m):
    """(mode | "mixed" | None, [(depth bits, trees, mode)], mean minimal instructions per (quartet, 32 trees) or None)."""
    import re
    head = variant.split("/")[1] if variant.count("/") >= 2 else ""
    mode = head if head in MODES else ("mixed" if head == "mixed" else None)
    classes = []
    for mo_, b_, cnt_ in re.findall(r"(?:(binary_full|binary_partial|general_full|partial)\.)?bitslice_b(\d+)(?:x2)?(?::(\d+))?", variant):
        classes.append((int(b_), int(cnt_) if cnt_ else m, mo_ or mode))
    ops32 = None
    if classes and "depth_u" not in variant and mode and all(mo_ in MODES for _, _, mo_ in classes):
        ops32 = sum(ops_of(b_, mo_) * cnt_ for b_, cnt_, mo_ in classes) / float(sum(cnt_ for _, cnt_, _ in classes))
    return mode, classes, ops32


# ---------------------------------------------------------------------------------------------------------------
def kernel_source_sha():
    h = hashlib.sha256()
    for fn in ("qs_count.hip", "qs_bitslice3.hpp", "qs_count_fused.hip", "qs_common.hpp"):
        with open(os.path.join(ROOT, "quartetscores_amd", "csrc", fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_for(workload_key, variant):
    """PMC figures of the count kernel for exactly this workload / variant / kernel source, or None."""
    import glob
    sha = kernel_source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
        except (OSError, ValueError):
            continue
        for ent in doc.get("entries", []):
            if ent.get("workload_key") == workload_key and ent.get("variant") == variant and ent.get("kernel_source_sha") == sha:
                ent = dict(ent)
                ent["file"] = os.path.relpath(path, ROOT)
                return ent
    return None


# ---------------------------------------------------------------------------------------------------------------
# secondary workloads of the default line (config.secondary): multifurcating / incomplete / mixed trees come from the numpy
# generator (11 s per 1500 trees at 512 taxa), so they are generated and flattened in CHILD processes started before anything
# else -- beside the cpu_baseline leg -- and only loaded when the timed region is over.
SECONDARY = [
    {"name": "collapse0.2", "label": "512 taxa x 1500 trees, 20 % of the inner edges collapsed", "kw": {"collapse": 0.2}},
    {"name": "dropout0.1", "label": "512 taxa x 1500 trees, 10 % of the taxa dropped per tree", "kw": {"dropout": 0.1}},
    {"name": "mixed", "label": "512 taxa x 1500 trees, a third each full / dropout 0.1 / collapse 0.2", "kw": {"mixed": True}},
]


def gen_child(spec_path):
    """Generates one secondary workload exactly as `bench.py --taxa n --trees m --collapse/--dropout/--mixed` does (same generator,
    same seeds) and writes its flattened arrays to spec["out"] (npz)."""
    import numpy as np
    from quartetscores_amd import flatten, synth
    with open(spec_path) as f:
        spec = json.load(f)
    n, m, seed, kw = spec["n"], spec["m"], spec["seed"], spec["kw"]
    ref = flatten.flatten_reference(spec["ref"])
    if kw.get("mixed"):
        k3 = m // 3
        sets = [synth.tree_set(n, m - 2 * k3, seed), synth.tree_set(n, k3, seed + 1, dropout=0.1), synth.tree_set(n, k3, seed + 2, collapse=0.2)]
        trees = [sets[i % 3][i // 3] if i // 3 < len(sets[i % 3]) else None for i in range(3 * len(sets[0]))]
        trees = [t_ for t_ in trees if t_ is not None]
    else:
        trees = synth.tree_set(n, m, seed, collapse=kw.get("collapse", 0.0), dropout=kw.get("dropout", 0.0))
    b = flatten.flatten_eval_trees(trees, ref.name_to_id)
    np.savez(spec["out"], leaf_off=b.leaf_off, leaf_ids=b.leaf_ids, adj_depth=b.adj_depth, n_trees=np.array([b.n_trees]))


def start_secondary_generators(ref_nw, n, m, seed):
    import tempfile
    d = tempfile.mkdtemp(prefix="qsbench_sec_")
    jobs = []
    for w in SECONDARY:
        sp, out = os.path.join(d, w["name"] + ".json"), os.path.join(d, w["name"] + ".npz")
        with open(sp, "w") as f:
            json.dump({"ref": ref_nw, "n": n, "m": m, "seed": seed, "kw": w["kw"], "out": out}, f)
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--gen-child", sp], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        jobs.append((w, p, sp, out))
    return d, jobs


def valu_frac(variant, m, units, count_ms):
    """(roofline.frac, minimal instructions per (quartet, 32 trees)) of a count step: algorithmic lane-ops of the classes that ran /
    the step's count-kernel time / the VALU peak; (None, None) for variants the issue model does not price (byte-SWAR, scatter)."""
    _mode, _classes, ops32 = parse_variant(variant, m)
    if not ops32 or not count_ms:
        return None, None
    return units * ops32 / 32.0 / (count_ms * 1e-3) / 1e12 / VALU_PEAK_TLOPS, ops32




def auto_mode(n, m_total, world, binary_full=True):
    """tree- or table-sharded for N > 1: quartetscores_amd.distributed.auto_mode (the Python multi-GPU driver and the C++ CLI hold the same
    arithmetic: table_shards.hpp prefer_table_shards)."""
    from quartetscores_amd import distributed
    return distributed.auto_mode(n, m_total, world, binary_full)


def run_mode_leg(mode, world, rank, local_rank, n, count_bits, batch_all, m_total, steps, use_dist, balance):
    """A few steps of one multi-GPU mode, self-contained (own context, table, buffers; everything is released before it returns): the
    N > 1 line carries the OTHER mode as a leg, so that one N-GPU lease measures tree- and table-sharded counting of the same trees.
    Binary trees holding all taxa only. Every rank calls it; the result is the same on every rank (max over ranks)."""
    import torch
    import torch.distributed as dist
    from quartetscores_amd import collectives as coll
    from quartetscores_amd import distributed, engine, ranks
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream(dev)
    nq_all = ranks.n_quartets(n)
    algo = engine.QS_ALGO_GATHER | engine.QS_COUNT_OVERWRITE
    pending = [None, None]
    info = {"mode": mode}
    if mode == "table":
        d_lo, d_hi = distributed.shard_of_largest_id(n, world, rank, by=balance)
        ctx = engine.Context(n, count_bits, device=local_rank, stream=stream.cuda_stream, d_lo=d_lo, d_hi=d_hi)
        table = torch.zeros(max((ctx.table_bytes + 3) // 4, 1), dtype=torch.int32, device=dev)
        ctx.table_attach(table)
        hb = ctx.batch_upload(batch_all, with_nodes=False)
        info.update({"table_shard_rank0": None, "balance": balance, "collective": None,
                     "step": "panel build (all trees) + count kernel on the rank's table shard; no table collective"})
        if rank == 0:
            info["table_shard_rank0"] = [d_lo, d_hi]

        def step(i):
            ctx.count_batch(hb, algo)
    else:
        lo, hi = distributed.shard_range(m_total, world, rank)
        ctx = engine.Context(n, count_bits, device=local_rank, stream=stream.cuda_stream)
        wire_fmt = (("u16x2" if m_total < 65536 else "u32x2") if count_bits == 32 else 16)
        _, chunk_words = distributed.scatter_layout(ctx.table_tuples, world, wire_fmt)
        send_words = world * chunk_words
        recv = [torch.zeros(chunk_words, dtype=torch.int32, device=dev) for _ in range(2)]
        wire = [torch.zeros(send_words, dtype=torch.int32, device=dev) for _ in range(2)]
        table = None
        if wire_fmt == "u32x2":
            table = torch.zeros((ctx.table_bytes + 3) // 4, dtype=torch.int32, device=dev)
            ctx.table_attach(table)
        hb = ctx.batch_upload(batch_all.slice(lo, hi), with_nodes=False)
        info.update({"wire": str(wire_fmt), "collective": "scatter", "collective_input_bytes_per_rank": send_words * 4,
                     "step": "panel build (trees/N) + count kernel on the full table + RCCL reduce-scatter, async, 2 buffers"})

        def step(i):
            k = i & 1
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
            if wire_fmt == "u16x2":
                ctx.wire_attach(wire[k])
                ctx.count_batch(hb, algo | engine.QS_COUNT_WIRE16X2)
            elif wire_fmt == "u32x2":
                ctx.count_batch(hb, algo)
                ctx.table_pack32x2(wire[k])
            else:                                   # u16 table: counted in place into the (padded) send buffer
                ctx.table_attach(wire[k])
                ctx.count_batch(hb, algo)
            if use_dist and world > 1:
                pending[k] = coll.reduce_scatter_tensor(recv[k], wire[k][:send_words], op=dist.ReduceOp.SUM, async_op=True)
            else:
                recv[k].copy_(wire[k][:chunk_words])

    def fence():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        ctx.sync()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    step(0)
    step(1)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        coll.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # gate: every tuple this rank holds after the step sums to the number of trees (binary trees, all taxa)
    if mode == "table":
        k_ = min(ctx.table_tuples, 1 << 24)
        cells = table[: 3 * k_] if count_bits == 32 else (table.view(torch.int16)[: 3 * k_].to(torch.int32) & 0xFFFF)
        ok_local = bool((cells.view(k_, 3).sum(dim=1) == m_total).all().item()) if k_ else True
    else:
        _own_lo, own_n = distributed.scatter_owned(ctx.table_tuples, world, rank, wire_fmt)
        k_ = min(own_n, 1 << 24)
        red = recv[(steps - 1) & 1]
        if wire_fmt == "u16x2":
            w_ = red[:k_]
            ok_local = bool((((w_ & 0xFFFF) + ((w_ >> 16) & 0xFFFF)) <= m_total).all().item())
        elif wire_fmt == "u32x2":
            ok_local = bool((red[: 2 * k_].view(-1, 2).to(torch.int64).sum(dim=1) <= m_total).all().item())
        else:
            cells = red.view(torch.int16)[: 3 * k_].to(torch.int32) & 0xFFFF
            ok_local = bool((cells.view(k_, 3).sum(dim=1) == m_total).all().item()) if k_ else True
    if use_dist:
        ok = torch.tensor([int(ok_local)], device=dev)
        coll.all_reduce(ok, op=dist.ReduceOp.MIN)
        ok_local = bool(ok.item())
    info.update({"steps": steps, "ms_per_step": elapsed / steps * 1e3, "value": m_total * nq_all * steps / elapsed,
                 "parity_tuple_sums_ok": ok_local, "algo": ctx.last_count_variant()[:100]})
    ctx.batch_free(hb)
    ctx.close()                      # (fence() above has drained the stream and the collectives)
    del table
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    return info


def main():
    args = parse_args()
    if args.cpu_child:
        return cpu_child(args.cpu_child)
    if args.gen_child:
        return gen_child(args.gen_child)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.via_launcher or args.dry_launch):
        sys.exit(launch(args, sys.argv[1:]))     # parent only: BEFORE torch is imported, never after a GPU call
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if BACKEND == "gloo":                # rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks: the ranks share the devices
        local_rank %= max(1, torch.cuda.device_count())
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if torch.cuda.device_count() <= local_rank:   # (counting devices does not initialise the GPU)
        raise SystemExit(f"bench.py rank {rank} needs GPU {local_rank}: {torch.cuda.device_count()} visible (no CPU fallback in quartetscores_amd)")
    # a rank started by torch.distributed.run (WORLD_SIZE set) always takes the RCCL path, also with one rank
    use_dist = "WORLD_SIZE" in os.environ or os.environ.get("QS_BENCH_FORCE_DIST") == "1"

    from quartetscores_amd import _lib, distributed, engine, flatten, native_ingest, ranks, synth
    from quartetscores_amd import collectives as coll

    # ---- workload -----------------------------------------------------------------------------------------
    cfg_no = args.config or 2        # the config the metric is quoted on, at every N (a driver curve is ONE workload)
    cfg = dict(CONFIGS[cfg_no])
    custom = bool(args.taxa or args.trees or args.count_bits or args.table_shards or args.split_trees >= 0 or args.shape != "random")
    n = args.taxa or cfg["taxa"]
    m_total = args.trees or cfg["trees"]
    count_bits = args.count_bits or cfg["bits"]
    shards = args.table_shards or cfg["shards"]
    binary_full_trees = not (args.collapse or args.dropout or args.mixed)
    # ---- how N > 1 ranks share the work ----
    #   table: rank r counts ALL trees into shard r of the table (largest taxon id in [d_lo, d_hi)): no table collective;
    #   tree:  rank r counts trees/N into a full table, one RCCL collective on the table per step;
    #   tree-weak (--split-trees 0): every rank counts --trees of its own + the collective (weak scaling; the pre-round-6 default of configs 1/2)
    # explicit shards (configs[4], --table-shards K): the table-sharded path with K >= N shards, as before.
    par_mode, mode_why, mode_est = "single", "one rank", None
    multi = world > 1 or os.environ.get("QS_BENCH_FORCE_DIST") == "1"
    if shards > 1:
        shards = max(shards, world)
        split = False
        par_mode, mode_why = "table", ("--table-shards" if args.table_shards else "configs[4]: the table is sharded by definition")
    elif args.split_trees == 0 and multi:
        split = False
        par_mode, mode_why = "tree-weak", "--split-trees 0"
    elif multi:
        if args.mode == "auto" and args.split_trees < 0 and world > 1:
            par_mode, mode_est = auto_mode(n, m_total, world, binary_full_trees)
            mode_why = "auto (model)"
        else:
            par_mode = "tree" if (args.mode == "auto" or args.split_trees == 1) else args.mode
            mode_why = "--mode " + args.mode if args.mode != "auto" else "one rank under the launcher: the tree-sharded path with nothing to exchange"
        split = par_mode == "tree"
        if par_mode == "table":
            shards = world               # (one shard per rank; world == 1: the whole table)
    else:
        split = False
    balance = args.balance if args.balance != "auto" else ("c4" if cfg_no == 4 and not args.table_shards else "cost")
    t_lo, t_hi = distributed.shard_range(m_total, world, rank) if split else (0, m_total)
    m = t_hi - t_lo                                                      # trees THIS rank counts per step
    # seeded inputs: seed = 1000 * config + tree-set id. Split configs: ONE set of m_total trees, rank
    # r takes trees [t_lo, t_hi); otherwise rank r counts its own set r (tree t of a set depends only on (seed, t)).
    seed_ref, seed_set = 1000 * cfg_no, 1000 * cfg_no + 1 + (0 if (split or par_mode == "table") else rank)
    want_other_leg = multi and par_mode in ("tree", "table") and not (args.table_shards or cfg["shards"] > 1) and binary_full_trees and args.algo == "gather" \
        and (args.other_leg == 1 or args.other_leg < 0)
    t_gen = time.perf_counter()
    batch_all = None
    if args.shape == "ladder":
        lad = "(t0,t1)"
        for i in range(2, n - 2):
            lad = "(" + lad + f",t{i})"
        ref_nw = f"({lad},t{n - 2},t{n - 1});"
        args.nni = True
    else:
        ref_nw = native_ingest.synth_trees(n, 1, seed_ref).decode().strip()
    ref = flatten.flatten_reference(ref_nw)
    if binary_full_trees:
        all_text = native_ingest.synth_trees(n, m_total if split else m, seed_set, kind="nni" if args.nni else "random",
                                             ref_text=ref_nw if args.nni else None)
        if want_other_leg and split:         # the other-mode leg (table) counts ALL trees on every rank
            batch_all, _ = native_ingest.ingest_text(ref_nw, all_text, 0, m_total, want_ranges=False)
            batch = batch_all.slice(t_lo, t_hi)
        else:
            batch, _ = native_ingest.ingest_text(ref_nw, all_text, t_lo if split else 0, t_hi if split else m, want_ranges=(args.algo == "scatter"))
            if want_other_leg:
                batch_all = batch
        sample_text = all_text
    else:                                   # multifurcating / partial trees: the numpy generator (small sizes only)
        if args.mixed:
            k3 = m // 3
            sets = [synth.tree_set(n, m - 2 * k3, seed_set), synth.tree_set(n, k3, seed_set + 1, dropout=args.dropout or 0.1),
                    synth.tree_set(n, k3, seed_set + 2, collapse=args.collapse or 0.2)]
            trees = [sets[i % 3][i // 3] if i // 3 < len(sets[i % 3]) else None for i in range(3 * len(sets[0]))]
            trees = [t_ for t_ in trees if t_ is not None]
        else:
            trees = synth.tree_set(n, m, seed_set, collapse=args.collapse, dropout=args.dropout)
        batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
        sample_text = "\n".join(trees).encode()
    assert batch.n_trees == m
    gen_s = time.perf_counter() - t_gen

    # the secondary workloads' trees: generated beside the cpu_baseline leg, loaded after the timed region
    want_secondary = args.secondary == 1 or (args.secondary < 0 and world == 1 and not custom and cfg_no == 2 and binary_full_trees
                                             and args.algo == "gather" and shards == 1 and not args.nni and not multi)
    sec_dir, sec_jobs = (None, [])
    if want_secondary and rank == 0:
        sec_dir, sec_jobs = start_secondary_generators(ref_nw, n, 1500, seed_set)

    # cpu_baseline leg first, in a child process, BEFORE this process touches the GPU (N = 1 only)
    cpu_baseline = None
    # (N = 1 only, as the measurement contract says: at N > 1 the other ranks would sit at the rendezvous for the two minutes it takes)
    if not args.no_cpu_baseline and rank == 0 and (world == 1 or args.cpu_baseline_at_n):
        cpu_baseline = run_cpu_baseline(ref_nw, sample_text, n, m, ranks.n_quartets(n), args.cpu_budget_s)

    cli_e2e = None
    if not args.no_e2e and world == 1 and binary_full_trees and shards == 1 and args.algo == "gather":
        cli_e2e = run_cli_e2e(ref_nw, sample_text)
        if "error" not in cli_e2e:      # the first process on a fresh box pays the driver's cold start (HIP runtime up after ~250 ms instead of ~80)
            time.sleep(3.0)             # (a process that starts right after another one released a 17-34 GB table waits in its own hipMalloc
                                        #  while the driver reclaims that memory: 750 ms instead of 540-570 without the pause)
            second = run_cli_e2e(ref_nw, sample_text)
            if "error" not in second:
                cli_e2e["second_run"] = {k_: second[k_] for k_ in ("counting_phase_ms", "scoring_phase_ms", "process_wall_ms")}
    # The peer-access leg (`--p2p-leg`, default at N > 1 on a split workload): the C++ host's communicator-free reduction
    # (QuartetScores --gpus N --reduce p2p: one process, hipDeviceEnablePeerAccess, qs_sum_words over xGMI) on the SAME trees, as
    # a child process of rank 0 BEFORE rank 0 touches its GPU (the other ranks wait at the rendezvous with idle devices), so that
    # one N-GPU run measures the RCCL path (`value`) and the p2p path (`config.p2p_leg`) side by side.
    p2p_leg = None
    want_p2p = args.p2p_leg == 1 or (args.p2p_leg < 0 and world > 1 and split)
    if want_p2p and rank == 0 and binary_full_trees and shards == 1 and args.algo == "gather":
        p2p_leg = run_cli_e2e(ref_nw, sample_text, threads=0, extra=["--gpus", str(world), "--reduce", "p2p", "--trace"])
        if p2p_leg.get("counting_phase_ms"):
            p2p_leg["counting_quartets_per_s"] = m_total * ranks.n_quartets(n) / (p2p_leg["counting_phase_ms"] * 1e-3)
        p2p_leg["what"] = f"QuartetScores -t 0 --gpus {world} --reduce p2p: Newick on disk -> reduced table, all {m_total} trees"[:100]

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # QS_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, barrier, collective) even with one rank
    comm = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        c0_ = time.perf_counter()
        if BACKEND == "gloo":          # rehearsal: the ranks share one GPU (collectives.py)
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
        # proof that RCCL's communicator spans the ranks the line claims: an all-reduce of ones must give the world size
        # on every rank (the first collective also completes the lazy parts of the communicator set-up)
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        coll.all_reduce(ones, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize(dev)
        comm = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "proof": int(ones.item()),
                "proof_ok": int(ones.item()) == world == dist.get_world_size(),
                "comm_init_ms": round((time.perf_counter() - c0_) * 1e3, 1),
                "comm_init_note": "rank 0: init_process_group + first all-reduce; outside the timed region"}
        if not comm["proof_ok"]:
            raise SystemExit(f"bench.py rank {rank}: all-reduce of ones gave {comm['proof']}, expected {world}")

    stream = torch.cuda.current_stream(dev)
    d_lo, d_hi = 0, n
    shard_index = None
    if shards > 1:
        shard_index = args.shard_index if args.shard_index >= 0 else rank
        d_lo, d_hi = distributed.shard_of_largest_id(n, shards, shard_index, by=balance)
    nq_all = ranks.n_quartets(n)
    nq = ranks.n_quartets(d_hi) - ranks.n_quartets(d_lo)                 # quartets this rank's table holds
    ctx = engine.Context(n, count_bits, device=local_rank, stream=stream.cuda_stream, d_lo=d_lo, d_hi=d_hi)
    if args.slice_bytes:
        ctx.set_tuning(_lib.QS_TUNE_PANEL_SLICE_BYTES, args.slice_bytes)
    n_words = (ctx.table_bytes + 3) // 4
    table = torch.zeros(n_words, dtype=torch.int32, device=dev)          # u16 tables reduce as packed words
    ctx.table_attach(table)

    # N > 1, tree-sharded: the per-rank tables are combined with ONE collective per step, asynchronous on RCCL's
    # stream, so that it overlaps the counting of the next step (two buffers in flight).
    #   --reduce scatter (default): rank r ends with tuples [r*T, (r+1)*T) of the reduced table and scores that shard
    #       (distributed.score_sharded): half the bytes per link of an all-reduce.
    #   --reduce all: every rank ends with the full table (the wording of BASELINE.json north_star).
    # Wire format: while the summed counts stay below 2^16 (the reference's own CINT rule, QuartetScores.cpp:115-147)
    # the u32 table travels as u16 cells, binary full batches as ONE word per tuple; else the table's own cells.
    # (one rank under torch.distributed.run -- `--via-launcher` at N = 1 -- initialises RCCL and proves its communicator, but has
    # no peer to combine a table with: the step is the N = 1 step)
    collective = (world > 1 or os.environ.get("QS_BENCH_FORCE_DIST") == "1") and shards == 1 and par_mode != "table"
    total_trees_reduced = m_total if split else m * world
    tables = [table]
    wire_fmt = None
    if collective and count_bits == 32 and args.algo == "gather":
        small = total_trees_reduced < 65536
        if args.wire == "u16x2" or (args.wire == "auto" and small and binary_full_trees):
            wire_fmt = "u16x2"
        elif args.wire == "u16" or (args.wire == "auto" and small):
            wire_fmt = "u16"
        elif args.wire == "u32x2" or (args.wire == "auto" and binary_full_trees and args.reduce == "scatter"):
            wire_fmt = "u32x2"      # configs[3]: 100 000 trees need u32 cells; (n0, n1) travel, n2 = total - n0 - n1 is restored: 8 B / tuple
    wire32x2 = wire_fmt == "u32x2"
    wire16 = wire_fmt is not None and not wire32x2
    if wire16:
        assert total_trees_reduced < 65536, "--wire u16 / u16x2 need fewer than 65536 trees in total"
    assert wire_fmt not in ("u16x2", "u32x2") or binary_full_trees, "--wire u16x2 / u32x2 need binary trees that hold all taxa"
    assert not wire32x2 or args.reduce == "scatter", "--wire u32x2 is a reduce-scatter format"
    reduce_mode = args.reduce if collective else None
    bits_wire = 16 if wire16 else count_bits
    layout_wire = wire_fmt or count_bits
    send_words = 0
    if reduce_mode == "scatter":
        _, chunk_words = distributed.scatter_layout(ctx.table_tuples, world, layout_wire)
        send_words = world * chunk_words
        recv = [torch.zeros(chunk_words, dtype=torch.int32, device=dev) for _ in range(2)]
    elif reduce_mode == "all":
        send_words = ctx.table_tuples if wire_fmt == "u16x2" else distributed.table_words(ctx.table_tuples, bits_wire)
    if wire16 or wire32x2:
        wire = [torch.zeros(send_words, dtype=torch.int32, device=dev) for _ in range(2)]
    elif collective:
        table = torch.zeros(max(n_words, send_words), dtype=torch.int32, device=dev)  # padded to world chunks
        ctx.table_attach(table)
        tables = [table]
        if 2 * table.numel() * 4 < 64 * (1 << 30):
            tables.append(torch.zeros_like(table))
    pending = [None] * 2
    step_no = [0]
    last_buf = [0]
    coll_ms = []
    hb = ctx.batch_upload(batch, with_nodes=(args.algo == "scatter"))    # inputs resident in HBM before the timed region
    algo = engine.QS_ALGO_GATHER if args.algo == "gather" else engine.QS_ALGO_SCATTER
    # gather: QS_COUNT_OVERWRITE = "clear + count" in one pass (the first slice stores instead of accumulating)
    step_algo = algo | engine.QS_COUNT_OVERWRITE if args.algo == "gather" else algo

    def step(timed=False, exchange=True):
        """One step. exchange=False (the count-only leg of an N > 1 line) keeps the counting path IDENTICAL -- wire words, pack
        pass, buffers -- and leaves out nothing but the dist.* call."""
        i = step_no[0] % (2 if (wire16 or wire32x2) else len(tables))
        step_no[0] += 1
        if pending[i] is not None:       # the collective that last used this buffer must be done
            pending[i].wait()
            pending[i] = None
        if len(tables) > 1:
            ctx.table_attach(tables[i])
        if args.algo != "gather":
            ctx.table_clear()
        t_flag = engine.QS_COUNT_TIMED if timed else 0
        if collective and wire_fmt == "u16x2":
            ctx.wire_attach(wire[i])     # counted straight into the wire words: no table write, no pack pass
            ctx.count_batch(hb, step_algo | engine.QS_COUNT_WIRE16X2 | t_flag)
        else:
            ctx.count_batch(hb, step_algo | t_flag)
        if collective:
            if wire_fmt == "u16x2":
                src = wire[i]
            elif wire16:
                (ctx.table_pack16x2 if wire_fmt == "u16x2" else ctx.table_pack16)(wire[i])
                src = wire[i]
            elif wire32x2:
                ctx.table_pack32x2(wire[i])      # (n0, n1) of every tuple: one pass over the table, 8 instead of 12 bytes per tuple on xGMI
                src = wire[i]
            else:
                src = tables[i]
            if not exchange:
                pass
            elif reduce_mode == "scatter":
                pending[i] = coll.reduce_scatter_tensor(recv[i], src[:send_words], op=dist.ReduceOp.SUM, async_op=True)
            else:
                pending[i] = coll.all_reduce(src, op=dist.ReduceOp.SUM, async_op=True)
        last_buf[0] = i

    def drain():
        for i, w in enumerate(pending):
            if w is not None:
                w.wait()
                pending[i] = None

    def fence():
        drain()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # first step: lazy initialisation (code objects, RCCL communicator); second step: its duration sizes the run
    step()
    drain()
    torch.cuda.synchronize(dev)
    t_pre = time.perf_counter()
    step()
    drain()
    torch.cuda.synchronize(dev)
    one_ms = max((time.perf_counter() - t_pre) * 1e3, 1e-3)
    if use_dist:
        tt = torch.tensor([one_ms], dtype=torch.float64, device=dev)
        coll.all_reduce(tt, op=dist.ReduceOp.MAX)
        one_ms = float(tt.item())
    steps = args.steps or int(min(500, max(5, 6000.0 / one_ms)))
    warmup = args.warmup if args.warmup >= 0 else (2 if one_ms >= 100 else 20)
    # untimed pre-conditioning (a sub-millisecond step is far shorter than the GPU's clock ramp: 20 steps of configs[1]
    # from idle measure 0.233 ms per step where 1000 steps measure 0.192 ms)
    if args.prewarm_ms > 0:
        for _ in range(int(min(1024, max(0, args.prewarm_ms / one_ms - 1)))):
            step()
        drain()
        torch.cuda.synchronize(dev)
    for _ in range(warmup):
        step()
    ctx.sync()
    fence()
    # HIP events on the launch stream bracket the whole timed region (GPU time per step); the kernels of the LAST
    # timed step are bracketed individually inside qs_count_batch (QS_COUNT_TIMED: an event after every launch)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for k_ in range(steps):
        step(timed=(k_ == steps - 1))
    ev1.record(stream)
    fence()
    t1 = time.perf_counter()
    ctx.sync()
    elapsed = t1 - t0
    region_gpu_ms = ev0.elapsed_time(ev1) / max(steps, 1)
    last_step_ms = ctx.last_count_ms() if steps > 0 else None
    last_fix_ms = ctx.last_count_fix_ms() if steps > 0 else None     # depth-clamp corrections inside count_kernels_ms (QS_TUNE_DEPTH_CLAMP)
    clamp_info = ctx.batch_clamp_info(hb)
    last_events = ctx.last_count_events() if steps > 0 else []
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        coll.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    drain()
    # collective alone (N > 1): the same buffers, nothing else on the GPU, a few repetitions
    coll_alone_ms = None
    if collective and steps > 0:
        src = (wire if (wire16 or wire32x2) else tables)[last_buf[0]]
        fence()
        c0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            if reduce_mode == "scatter":
                coll.reduce_scatter_tensor(recv[last_buf[0]], src[:send_words], op=dist.ReduceOp.SUM)
            else:
                coll.all_reduce(src, op=dist.ReduceOp.SUM)
        fence()
        coll_alone_ms = (time.perf_counter() - c0) * 1e3 / reps
        step()                              # restore a freshly counted + reduced buffer for the gates below
        drain()
        torch.cuda.synchronize(dev)
    # the SAME ranks count the SAME shares without the table collective, inside this run: what the step costs when nothing is
    # exchanged (max over ranks). scaling_efficiency = value / (N x one rank's count-only rate) then reads from this line alone.
    count_only_ms = None
    if collective and steps > 0:
        fence()
        step(exchange=False)
        ctx.sync()
        fence()
        k_co = max(3, min(steps, 10))
        c0 = time.perf_counter()
        for _ in range(k_co):
            step(exchange=False)            # same store path as the timed step (wire words / pack pass), no dist.* call
        ctx.sync()
        torch.cuda.synchronize(dev)
        count_only_ms = (time.perf_counter() - c0) * 1e3 / k_co
        if use_dist:
            tt = torch.tensor([count_only_ms], dtype=torch.float64, device=dev)
            coll.all_reduce(tt, op=dist.ReduceOp.MAX)
            count_only_ms = float(tt.item())
        step()                              # a freshly counted + reduced buffer for the gates below
        drain()
        torch.cuda.synchronize(dev)
    # gate on the REDUCED table of the last step: every tuple sums to the total number of trees (binary, full trees)
    reduced_ok = None
    shard16 = None
    if collective and steps > 0:
        if reduce_mode == "scatter":
            own_lo, own_n = distributed.scatter_owned(ctx.table_tuples, world, rank, layout_wire)
            red, n_red = recv[last_buf[0]], own_n
        else:
            red, n_red = (wire if (wire16 or wire32x2) else tables)[last_buf[0]], nq
        if wire_fmt == "u16x2":            # restore the third cell of the reduced tuples (a u16 table again)
            shard16 = torch.zeros(distributed.table_words(max(n_red, 1), 16), dtype=torch.int32, device=dev)
            ctx.unpack16x2(red, n_red, total_trees_reduced, shard16)
            ctx.sync()                     # raises if a reduced tuple exceeds the total
            w_ = red[:n_red]
            ok_local = bool((((w_ & 0xFFFF) + ((w_ >> 16) & 0xFFFF)) <= total_trees_reduced).all().item())
        elif wire32x2:                     # restore the third cell of the reduced tuples (u32 cells again); raises if a tuple exceeds the total
            shard16 = torch.zeros(distributed.table_words(max(n_red, 1), 32), dtype=torch.int32, device=dev)
            ctx.unpack32x2(red, n_red, total_trees_reduced, shard16)
            ctx.sync()
            pr = red[: 2 * n_red].view(-1, 2).to(torch.int64)
            ok_local = bool((pr.sum(dim=1) <= total_trees_reduced).all().item())
            del pr
        elif binary_full_trees:
            ok_local = True
            for c0_ in range(0, n_red, 1 << 26):   # chunked: the shard can be GBs
                c1_ = min(n_red, c0_ + (1 << 26))
                cells = red[c0_ * 3: c1_ * 3] if bits_wire == 32 else (red.view(torch.int16)[c0_ * 3: c1_ * 3].to(torch.int32) & 0xFFFF)
                ok_local = ok_local and bool((cells.view(c1_ - c0_, 3).sum(dim=1) == total_trees_reduced).all().item())
                del cells
        else:
            ok_local = None
        if ok_local is not None:
            ok = torch.tensor([int(ok_local)], device=dev)
            coll.all_reduce(ok, op=dist.ReduceOp.MIN)      # every rank's shard must pass
            reduced_ok = bool(ok.item())
    if len(tables) > 1:                  # measurements below run on one table without collectives
        ctx.table_attach(table)
        tables[:] = [table]
        pending[:] = [None]
    collective_saved, collective = collective, False
    # per-kernel durations = the events qs_count_batch recorded around every launch of the LAST step of the timed region
    # (the roofline's `avg_launch_ms`). A sub-millisecond step is too short for one sample: there, and only there, the
    # average over a few extra event-bracketed steps is used and labelled `kernel_ms_source`.
    last_launches = ctx.last_count_launches() if steps > 0 else 0
    kernel_ms_source = "last step of the timed region"
    if last_step_ms and last_launches and one_ms >= 5:
        panel_ms, count_ms, launches = float(last_step_ms[0]), float(last_step_ms[1]), int(last_launches)
    else:
        kern = []
        for _ in range(max(3, min(steps, 10))):
            step(timed=True)
            kern.append(ctx.last_count_ms() + (ctx.last_count_launches(),))
        panel_ms = float(np.mean([k[0] for k in kern]))
        count_ms = float(np.mean([k[1] for k in kern]))                  # all count-kernel launches of one step
        launches = int(kern[-1][3]) or 1
        kernel_ms_source = f"mean of {len(kern)} extra event-bracketed steps after the timed region"
    variant = ctx.last_count_variant()

    # ---- parity gates run with every measurement -------------------------------------------------------------
    step()
    ctx.sync()
    # ---- scoring (secondary metric of SURVEY 8(d)): ONE cold call (first use: reference tree, LCA matrix, plans,
    # accumulator allocation) and then warm calls; phases from qs_last_score_ms. It runs BEFORE the gates below: they
    # allocate and free tens of GB of scratch (a 17 GB table, a 34 GB clone), and the first hipMalloc after such a free
    # blocks for ~0.5 s in the runtime -- that, not qs_score, was the 492 ms of round 2's driver line
    score_cold_ms = score_ms = None
    score_phases_cold = score_phases = None
    score_mode = None
    if not args.no_score:
        def score_once():
            torch.cuda.synchronize(dev)
            s0 = time.perf_counter()
            if shards > 1:
                # table-sharded mode (configs[4]): this rank's shard stays resident; pass 1 -> SUM / MIN over the ranks ->
                # pass 2 -> gather of the candidates -> host finish (distributed.score_table_shards). With fewer ranks than
                # shards (N = 1: one shard of 8) the other shards are simply absent: the TIME is that of one rank's share,
                # the scores are those of the quartets this shard owns.
                distributed.score_table_shards(lambda k: ctx, [shard_index], ref, device=dev)
                ph = None
            elif reduce_mode == "scatter" and steps > 0:
                # every rank scores the shard it received (view), accumulators combined with small collectives
                own_lo, own_n = distributed.scatter_owned(ctx.table_tuples, world, rank, layout_wire)
                ctx.score_set_view(shard16 if wire_fmt in ("u16x2", "u32x2") else recv[last_buf[0]], bits_wire, own_lo, own_n)
                distributed.score_sharded(ctx, ref)
                ctx.score_set_view(None, 0, 0, 0)
                ph = None
            else:
                ctx.score(ref)
                ph = {k_: round(v_, 3) for k_, v_ in ctx.last_score_ms().items()}
                ph["log_records"] = ctx.last_score_log()          # > 0: the table was read ONCE (pass2 = filter over the log)
                ph["log_predicted"] = ctx.last_score_estimate()
            return (time.perf_counter() - s0) * 1e3, ph
        score_cold_ms, score_phases_cold = score_once()
        warm = [score_once() for _ in range(3)]
        score_ms, score_phases = min(warm, key=lambda x: x[0])
        score_mode = ("table shards: pass 1, SUM/MIN, pass 2, gather, finish" + ("" if world >= shards else f" ({world} of {shards} shards present)")) if shards > 1 \
            else "reduce-scattered shard per rank" if (reduce_mode == "scatter" and steps > 0) \
            else ("qs_score: one read (sampled bounds, candidate log)" if (score_phases or {}).get("log_records") else "qs_score: two passes")

    # the score kernels are HBM-bound: the warm call's table pass(es) against the 8 TB/s peak (6-7 TB/s is what a plain streaming
    # read reaches on this chip: tools/read_bw.hip); `reads_of_table` = 1 when pass 2 was a filter over pass 1's candidate log
    score_roofline = None
    if score_phases and score_phases.get("pass1"):
        one_read = bool(score_phases.get("log_records"))
        t_ms = score_phases["pass1"] + (0.0 if one_read else score_phases.get("pass2", 0.0))
        moved = ctx.table_bytes * (1 if one_read else 2)
        score_roofline = {"bound": "hbm", "achieved": round(moved / t_ms * 1e-6, 1), "peak": 8000.0, "unit": "GB/s",
                          "frac": round(moved / t_ms * 1e-6 / 8000.0, 4), "reads_of_table": 1 if one_read else 2,
                          "table_pass_ms": round(t_ms, 3), "note": "pass 1 incl. the two samples" if one_read else "pass 1 + pass 2"}

    parity = None
    if binary_full_trees:              # tuples sum to m only when every tree resolves every quartet
        parity = True
        t16 = table.view(torch.int16)
        for c0_ in range(0, nq, 1 << 26):
            c1_ = min(nq, c0_ + (1 << 26))
            cells = table[c0_ * 3: c1_ * 3] if count_bits == 32 else (t16[c0_ * 3: c1_ * 3].to(torch.int32) & 0xFFFF)
            parity = parity and bool((cells.view(c1_ - c0_, 3).sum(dim=1) == m).all().item())
            del cells
    # stronger gate (the tuple sums are trivially m in the binary_full variant): the table must equal the one the
    # independent byte-SWAR implementation of the same count produces, bit for bit, on the device
    impl_match = None
    free_b, _tot = torch.cuda.mem_get_info(dev)
    if args.algo == "gather" and "bitslice" in variant and not args.no_impl_check and free_b > table.numel() * 4 + (4 << 30):
        mine = table.clone()
        ctx.set_tuning(_lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_SWAR)
        step()
        ctx.sync()
        swar_variant = ctx.last_count_variant()
        impl_match = bool(torch.equal(mine, table))
        ctx.set_tuning(_lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_AUTO)
        step()                          # restore the default implementation's table (and variant string)
        ctx.sync()
        assert "bitslice" in ctx.last_count_variant() and "depth_u" in swar_variant
        del mine
    # third gate, independent of every kernel: random quartets of this rank's table against the split-based brute
    # force (tests/bruteforce.py) on a short prefix of this rank's trees, counted into a scratch context
    lookup_ok = None
    if rank == 0 and binary_full_trees:
        import bruteforce
        k_trees = min(m, 48)
        rng = np.random.default_rng(12345)
        qs_ = np.sort(np.stack([rng.choice(d_hi, size=4, replace=False) for _ in range(20000)]), axis=1)
        qs_ = qs_[qs_[:, 3] >= d_lo].astype(np.uint16)
        prefix = sample_text.split(b"\n")[(t_lo if split else 0):][:k_trees]
        ctx_s = engine.Context(n, 16, device=local_rank, stream=stream.cuda_stream, d_lo=d_lo, d_hi=d_hi)   # scratch u16 table
        if torch.cuda.mem_get_info(dev)[0] > ctx_s.table_bytes + (4 << 30):
            ctx_s.table_alloc()
            ctx_s.count_trees(batch.slice(0, k_trees), algo)
            got = ctx_s.lookup(qs_)
            # lookup ids = the reference tree's leaf order (ref.names): the brute force works on the same ids
            want = bruteforce.quartet_counts_for([ln.decode() for ln in prefix], ref.names, qs_.astype(np.int64))
            lookup_ok = bool((got == want).all())
        ctx_s.close()

    # ---- e2e leg 2: the same step with the host-array upload (qs_batch_upload: validation, pinned staging, H2D copy)
    # inside the timed region; the boundary hands over host arrays, so this is the PCIe-inclusive rate
    upload_step_ms = None
    if not args.no_e2e and args.algo == "gather":
        def up_step():
            h2 = ctx.batch_upload(batch, with_nodes=False)
            ctx.count_batch(h2, step_algo)
            ctx.batch_free(h2)
        up_step()
        ctx.sync()
        k_up = max(3, min(steps, 20)) if one_ms >= 100 else max(3, min(steps, 50))   # (a timed region of its own, not a 3-step aside)
        torch.cuda.synchronize(dev)
        u0 = time.perf_counter()
        for _ in range(k_up):
            up_step()
        ctx.sync()
        torch.cuda.synchronize(dev)
        upload_step_ms = (time.perf_counter() - u0) * 1e3 / k_up
        step()                           # the gates / scoring below read the table of a plain step
        ctx.sync()

    # ---- secondary workloads on the resident table (config.secondary): multifurcating / incomplete / mixed gene-tree batches at the
    # taxon count of the default line, and configs[1]; 1 warm + 3 timed steps each, kernels bracketed by the library's own events ----
    secondary = None
    if want_secondary and rank == 0:
        secondary = []

        def measure(ctx_, hb_, m_, label, k_steps=3, nq_=None):
            ctx_.count_batch(hb_, step_algo)
            ctx_.sync()
            torch.cuda.synchronize(dev)
            c0_ = time.perf_counter()
            for i_ in range(k_steps):
                ctx_.count_batch(hb_, step_algo | (engine.QS_COUNT_TIMED if i_ == k_steps - 1 else 0))
            ctx_.sync()
            torch.cuda.synchronize(dev)
            ms_ = (time.perf_counter() - c0_) * 1e3 / k_steps
            _p, cnt_ms, _t = ctx_.last_count_ms()
            v_ = ctx_.last_count_variant()
            nq_ = nq_ or ranks.n_quartets(ctx_.n)     # (a table shard: the quartets it owns)
            # frac: the same definition as roofline.frac (count kernels incl. corrections of the last step, event-timed); for steps below
            # a millisecond the wall-clock mean over the steps stands in for one event sample
            fr, ops_ = valu_frac(v_, m_, m_ * nq_, cnt_ms if ms_ >= 5 else ms_)
            return {"workload": label[:100], "algo": v_[:100], "value": m_ * nq_ / (ms_ * 1e-3), "ms_per_step": round(ms_, 4), "steps": k_steps,
                    "count_kernels_ms_last_step": round(cnt_ms, 4), "launches": ctx_.last_count_launches(),
                    "frac": round(fr, 4) if fr else None, "ops_per_unit32": ops_}
        for w, p_, sp_, out_ in sec_jobs:
            try:
                _o, err_ = p_.communicate(timeout=600)
                if p_.returncode != 0:
                    raise RuntimeError(err_[-200:])
                z = np.load(out_)
                empty = np.zeros(0, dtype=np.uint32)
                mt = int(z["n_trees"][0])
                b_ = flatten.TreeBatch(mt, z["leaf_off"], z["leaf_ids"], z["adj_depth"], np.zeros(mt + 1, dtype=np.uint32), np.zeros(1, dtype=np.uint32), empty.astype(np.uint16))
                hb_ = ctx.batch_upload(b_, with_nodes=False)
                secondary.append(measure(ctx, hb_, mt, w["label"]))
                ctx.batch_free(hb_)
            except Exception as e:           # reported, never required for the metric
                secondary.append({"workload": w["label"][:100], "error": str(e)[:100]})
            finally:
                for q in (sp_, out_):
                    try:
                        os.remove(q)
                    except OSError:
                        pass
        try:
            os.rmdir(sec_dir)
        except OSError:
            pass
        try:                                 # configs[1]: its own 128 MB table
            c1 = CONFIGS[1]
            ref1 = native_ingest.synth_trees(c1["taxa"], 1, 1000).decode().strip()
            text1 = native_ingest.synth_trees(c1["taxa"], c1["trees"], 1001)
            b1, _ = native_ingest.ingest_text(ref1, text1, 0, c1["trees"], want_ranges=False)
            ctx1 = engine.Context(c1["taxa"], c1["bits"], device=local_rank, stream=stream.cuda_stream)
            ctx1.table_alloc()
            hb1 = ctx1.batch_upload(b1, with_nodes=False)
            for _ in range(300):             # clock ramp: a 0.17 ms step from idle measures the ramp, not the kernel
                ctx1.count_batch(hb1, step_algo)
            secondary.append(measure(ctx1, hb1, c1["trees"], "configs[1]: 128 taxa x 1000 trees, u32 table, seeds 1000/1001", k_steps=500))
            ctx1.batch_free(hb1)
            ctx1.close()
        except Exception as e:
            secondary.append({"workload": "configs[1]", "error": str(e)[:100]})
        try:                                 # configs[3]: one GPU's share of 8 (12 500 of the 100 000 trees into the full 2.1 GB table)
            c3 = CONFIGS[3]
            ref3 = native_ingest.synth_trees(c3["taxa"], 1, 3000).decode().strip()
            text3 = native_ingest.synth_trees(c3["taxa"], c3["trees"] // 8, 3001)
            b3, _ = native_ingest.ingest_text(ref3, text3, 0, c3["trees"] // 8, want_ranges=False)
            ctx3 = engine.Context(c3["taxa"], c3["bits"], device=local_rank, stream=stream.cuda_stream)
            ctx3.table_alloc()
            hb3 = ctx3.batch_upload(b3, with_nodes=False)
            secondary.append(measure(ctx3, hb3, c3["trees"] // 8, "configs[3]: one rank's share of 8: 256 taxa x 12500 of the 100000 trees, u32 table, seeds 3000/3001", k_steps=10))
            ctx3.batch_free(hb3)
            ctx3.close()
            del b3, text3
        except Exception as e:
            secondary.append({"workload": "configs[3] share", "error": str(e)[:100]})
        try:                                 # configs[4]: one of its 8 table shards (34 GB of u16 cells), all 5000 trees
            c4 = CONFIGS[4]
            lo4, hi4 = distributed.shard_of_largest_id(c4["taxa"], c4["shards"], 0, by="c4")
            if torch.cuda.mem_get_info(dev)[0] > (ranks.n_quartets(hi4) - ranks.n_quartets(lo4)) * 6 + (8 << 30):
                ref4 = native_ingest.synth_trees(c4["taxa"], 1, 4000).decode().strip()
                text4 = native_ingest.synth_trees(c4["taxa"], c4["trees"], 4001)
                b4, _ = native_ingest.ingest_text(ref4, text4, 0, c4["trees"], want_ranges=False)
                ctx4 = engine.Context(c4["taxa"], c4["bits"], device=local_rank, stream=stream.cuda_stream, d_lo=lo4, d_hi=hi4)
                ctx4.table_alloc()
                hb4 = ctx4.batch_upload(b4, with_nodes=False)
                secondary.append(measure(ctx4, hb4, c4["trees"], f"configs[4]: shard d[{lo4},{hi4}) of 8: 1024 taxa x 5000 trees, u16 table, seeds 4000/4001", k_steps=3,
                                         nq_=ranks.n_quartets(hi4) - ranks.n_quartets(lo4)))
                ctx4.batch_free(hb4)
                ctx4.close()
                del b4, text4
        except Exception as e:
            secondary.append({"workload": "configs[4] shard", "error": str(e)[:100]})
        step()                               # the resident table holds the default workload's counts again
        ctx.sync()

    # how fast THIS device runs the count kernel's bare instruction slot (qs_issue_probe: 24 v_bitop3 + 4 v_bcnt in registers, 4
    # waves per SIMD): boxes of one pool differ by several per cent; the figure makes lines from different boxes comparable
    box_probe_ns = None
    if rank == 0:
        try:
            box_probe_ns = ctx.issue_probe(60000)
        except Exception:        # diagnostic only
            box_probe_ns = None

    # ---- the OTHER multi-GPU mode on the same trees, a few steps, every rank (the line's `config.other_mode_leg`) ----
    other_leg = None
    if want_other_leg and steps > 0:
        other = "table" if par_mode == "tree" else "tree"
        k_leg = max(3, min(steps, 10))
        try:
            other_leg = run_mode_leg(other, world, rank, local_rank, n, count_bits, batch_all, m_total, k_leg, use_dist, balance)
        except Exception as e:                   # a leg must never take the line down; every rank fails the same way or not at all
            if use_dist and world > 1:
                raise
            other_leg = {"mode": other, "error": str(e)[:200]}

    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    # ---- the JSON line ----------------------------------------------------------------------------------------
    # units of one step over all ranks: tree-sharded = every rank's trees x all quartets; table-sharded = all trees x
    # the quartets of every rank's shard (with one shard per rank that is all quartets)
    if shards > 1:
        owned = [ranks.n_quartets(hi_) - ranks.n_quartets(lo_) for lo_, hi_ in
                 (distributed.shard_of_largest_id(n, shards, (args.shard_index if args.shard_index >= 0 else r), by=balance) for r in range(world))]
        units_per_step = m * sum(owned)
    else:
        units_per_step = (m_total if split else m * world) * nq_all
    value = units_per_step * steps / elapsed
    bytes_per_unit = 2 * (count_bits // 8)
    units_per_launch = m * nq / launches
    launch_ms = count_ms / launches
    import re
    mode, classes, ops32 = parse_variant(variant, m)
    depth_bits = max((b_ for b_, _, _ in classes), default=None)
    import re
    wl_name = (f"configs[{cfg_no}]" if not custom else "custom")
    workload_key = f"n{n}_m{m}_u{count_bits}_shard{d_lo}-{d_hi}_{'ladder' if args.shape == 'ladder' else 'nni' if args.nni else 'random'}" + ("" if binary_full_trees else f"_c{args.collapse}_d{args.dropout}" + ("_mixed" if args.mixed else ""))
    out = {
        "metric": "quartets counted/sec",
        "value": value,
        "unit": "quartets/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "n/a" if world == 1 else ("strong" if (split or (shards > 1 and shards == world and args.shard_index < 0)) else "weak"),
        "vs_baseline": None,
        "dtype": "u32" if count_bits == 32 else "u16",
        "data": "synthetic",
        "config": {
            "workload": (f"{wl_name}: {n} taxa x {m_total} trees" + (f" split over {world} ranks" if split else " (all on every rank)" if (world > 1 and par_mode == "table") else " per rank" if world > 1 else "")
                         + f", u{count_bits} table" + (f" shard d[{d_lo},{d_hi}) of {shards}" if shards > 1 else "")
                         + (", ladder+NNI trees" if args.shape == "ladder" else ", ref+NNI trees" if args.nni else ", random binary trees" if binary_full_trees else (", mixed thirds" if args.mixed else "") + f", collapse {args.collapse} dropout {args.dropout}")
                         + f", seeds {seed_ref}/{seed_set}")[:100],
            "baseline_config": (f"BASELINE.json configs[{cfg_no}] (bench.py --config {cfg_no})" if not custom else "custom (not a BASELINE config)"),
            "mode": par_mode, "mode_decided_by": mode_why, "mode_model": mode_est, "shard_balance": balance if shards > 1 else None,
            "other_mode_leg": other_leg,
            "secondary": secondary,
            "one_rank_same_workload": same_workload_scaling(value, world, m * nq_all, count_only_ms, elapsed / steps * 1e3) if count_only_ms else None,
            "quartets": nq_all,
            "quartets_this_rank": nq,
            "workload_key": workload_key,
            "table_shard": [d_lo, d_hi] if shards > 1 else None,
            "algo": variant,
            "step": (("panel build + count kernel per slice (1st slice stores)" if args.algo == "gather" else "table clear + count kernel")
                     + ((" + " + {"u16": "pack u16 + ", "u16x2": "1 word/tuple wire + ", "u32x2": "2 u32 cells/tuple wire + ", None: ""}[wire_fmt] + ("RCCL reduce-scatter" if reduce_mode == "scatter" else "RCCL all-reduce") + ", async, 2 buffers") if collective_saved else ""))[:100],
            "collective": reduce_mode,
            "collective_input_bytes_per_rank": send_words * 4 if collective_saved else None,
            "collective_alone_ms": coll_alone_ms,
            "parity_reduced_tuple_sums_ok": reduced_ok,
            "parity_tuple_sums_ok": parity,
            "parity_bitslice_equals_swar_impl": impl_match,
            "parity_lookup_equals_bruteforce": lookup_ok,
            "panel_kernels_ms_per_step": panel_ms,
            "count_kernels_ms_per_step": count_ms,
            "count_launches_per_step": launches,
            "kernel_ms_source": kernel_ms_source,
            "prewarm_ms": args.prewarm_ms,
            "count_kernels_ms_last_timed_step": last_step_ms[1] if last_step_ms else None,
            "kernels_of_last_timed_step": [[k_, round(ms_, 3)] for k_, ms_ in last_events][:24],
            "depth_clamp": {"trees_below_own_depth_bits": clamp_info[0], "tree_quartet_corrections": clamp_info[1], "fix_workgroups": clamp_info[2],
                            "fix_kernels_ms_last_timed_step": round(last_fix_ms, 3) if last_fix_ms is not None else None,
                            "note": "trees in a class below their depth bits; clamp_fix_kernel adds the tied quartets (in count ms)"},
            "gpu_ms_per_step_events_over_timed_region": region_gpu_ms,
            "score_mode": score_mode,
            "score_phase_ms": score_ms,
            "score_phase_ms_cold": score_cold_ms,
            "score_phases_ms": score_phases,
            "score_phases_ms_cold": score_phases_cold,
            "score_roofline": score_roofline,
            "input_generation_s": gen_s,
            "box_issue_probe_ns_per_inst": box_probe_ns,
            "box_issue_probe_note": "bare 24 v_bitop3 + 4 v_bcnt slot, 4 waves/SIMD, this device; 1.39-1.40 in r03_valu_yardstick.txt",
        },
    }
    # SURVEY 8(d) defines the phase from "trees resident on host": the same step with qs_batch_upload (validation, class plan, pinned
    # staging, H2D) inside, at top level beside `value` (which the bench contract defines with the inputs resident in HBM)
    if upload_step_ms:
        out["value_upload_inclusive"] = (units_per_step if shards > 1 else (m * nq) * world) / (upload_step_ms * 1e-3)
        out["ms_per_step_upload_inclusive"] = upload_step_ms
        out["config"]["resident_ms_per_step"] = elapsed / steps * 1e3
        out["config"]["value_definition"] = "value: inputs in HBM (bench contract); value_upload_inclusive: host arrays -> table (SURVEY 8d)"
    # e2e: what the product delivers when the inputs are NOT yet resident (never `value`)
    if upload_step_ms or cli_e2e:
        e2e = {"note": "inputs not resident: never `value`"}
        if upload_step_ms:
            e2e["upload_in_step_ms"] = upload_step_ms
            e2e["upload_in_step_quartets_per_s"] = (m * nq) / (upload_step_ms * 1e-3)
            e2e["host_batch_bytes"] = int(batch.leaf_ids.nbytes + batch.adj_depth.nbytes + batch.leaf_off.nbytes)
        if cli_e2e:
            e2e["cli"] = cli_e2e
            if cli_e2e.get("counting_phase_ms"):
                e2e["cli_counting_quartets_per_s"] = m * nq_all / (cli_e2e["counting_phase_ms"] * 1e-3)
        out["e2e"] = e2e
    kname = (("count_bitslice3_fused_kernel" if "/fused:" in variant else "count_bitslice3_kernel") if "bitslice" in variant else "count_gather_kernel") if args.algo == "gather" else "count_scatter_kernel"
    hbm_ratio = (units_per_launch * bytes_per_unit) / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    table_bytes = ctx.table_bytes
    panel_bytes = ((m + 31) // 32) * (n * (n - 1) // 2) * (max(depth_bits or 4, 4) * 4) if depth_bits else None
    if ops32:
        achieved = units_per_launch * ops32 / 32.0 / (launch_ms * 1e-3) / 1e12
        # what the instruction mix itself allows (profiles/r03_valu_yardstick.txt): v_bcnt_u32_b32 issues at half rate, so
        # the minimal chain of a class with B depth bits takes 2 x (ops - pops) + 4 x pops cycles, not 2 x ops
        cyc_min = sum((2 * (ops_of(b_, mo_) - POPS[mo_]) + 4 * POPS[mo_]) * cnt_ for b_, cnt_, mo_ in classes) / float(sum(cnt_ for _, cnt_, _ in classes))
        mix_ceiling = 2.0 * ops32 / cyc_min
        roof = {"bound": "valu_issue", "achieved": achieved, "peak": VALU_PEAK_TLOPS, "unit": "Tlane-op/s", "frac": achieved / VALU_PEAK_TLOPS,
                "issue_model": {"bcnt_half_rate": True, "min_cycles_per_unit32": cyc_min, "mix_ceiling_frac": mix_ceiling,
                                "frac_of_mix_ceiling": achieved / VALU_PEAK_TLOPS / mix_ceiling, "source": "profiles/r03_valu_yardstick.txt"},
                "algorithmic_ops_per_unit": ops32 / 32.0,
                "algorithmic_ops_note": f"{ops32:.3f} wave-instr per (quartet, 32 trees), classes (B, trees, mode): {classes}"[:100]}
    else:                                   # scatter / SWAR paths: priced against HBM with SURVEY 8(d)'s bytes
        achieved = (units_per_launch * bytes_per_unit) / (launch_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
    roof.update({"traffic": None, "kernel": kname, "units_per_launch": units_per_launch, "avg_launch_ms": launch_ms,
                 "launches_per_step": launches, "hbm_algorithmic_ratio": hbm_ratio,
                 "hbm_algorithmic_note": "SURVEY 8(d) RMW bytes / time / 8 TB/s: speed-up over a perfect scatter, not a fraction",
                 "hbm_model_bytes_per_step": (table_bytes * (2 * launches - 1) + 2 * panel_bytes) if panel_bytes else None,
                 "hbm_model_note": "gather minimum: table 1 write + (launches-1) RMW, panel written+read once",
                 "issued": None})
    pmc = pmc_for(workload_key, variant)
    if pmc:
        roof["traffic"] = pmc.get("hbm_bytes_per_launch")
        if pmc.get("valu_insts_per_launch"):
            per_simd = pmc["valu_insts_per_launch"] / 1024.0
            roof["issued"] = {"valu_wave_insts_per_launch": pmc["valu_insts_per_launch"],
                              "cycles_per_valu_inst_at_2.4GHz": launch_ms * 1e-3 * 2.4e9 / per_simd,
                              "frac_of_2_cycle_issue": 2.0 * per_simd / (launch_ms * 1e-3 * 2.4e9),
                              "minimal_share": units_per_launch * ops32 / 32.0 / 64.0 / pmc["valu_insts_per_launch"] if ops32 else None}
        roof["pmc_source"] = {k_: pmc.get(k_) for k_ in ("file", "collected", "kernel_source_sha", "fetch_size_kb", "write_size_kb", "l2_hit")}
        if roof["traffic"]:
            # the axis BASELINE.json north_star names: achieved HBM GB/s of the dominant kernel against the 8 TB/s roofline. traffic = the
            # PMC bytes of one launch (FETCH_SIZE x 2 + WRITE_SIZE, profiles/<pmc file>) / THIS run's event-timed launch duration;
            # over_model = traffic / the bytes the gather formulation has to move (table once, panel written + read once)
            gbps = roof["traffic"] / (launch_ms * 1e-3) / 1e9
            model_b = roof.get("hbm_model_bytes_per_step")
            roof["hbm"] = {"achieved_gbps": round(gbps, 1), "frac_of_8tbps": round(gbps / HBM_PEAK_GBS, 4),
                           "over_model": round(roof["traffic"] * launches / model_b, 2) if model_b else None,
                           "note": "fabric traffic of the count kernel (PMC, builder-collected, same kernel source); the kernel is VALU-bound"}
    else:
        roof["pmc_source"] = f"none under profiles/ for kernel source {kernel_source_sha()}"
    out["roofline"] = roof

    if p2p_leg is not None:
        out["config"]["p2p_leg"] = p2p_leg
    if cpu_baseline is not None:                     # rank 0's host (N = 1)
        out["cpu_baseline"] = cpu_baseline
    if comm is not None:
        comm["collective_alone_ms"] = coll_alone_ms
        comm["table_collective"] = reduce_mode
        out["collective"] = comm

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # the JSON line must be the LAST line on stdout: RCCL prints a version banner through C stdio, which is
    # block-buffered on a pipe and would otherwise be flushed after Python's line at exit
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(out))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
