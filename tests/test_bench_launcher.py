"""bench.py's multi-GPU launch path: `python bench.py --gpus N` with N > 1 must run by itself -- the parent
starts ONE fresh `python -m torch.distributed.run` child before anything touches a GPU, relays rank 0's JSON line and the
child's exit code. CPU tests: the command it builds, the loud failure without GPUs, the relay. The GPU side
(`--gpus 1 --via-launcher` against the plain N = 1 line) is tests/test_gpu_bench_launcher.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_dry_launch_builds_the_drivers_command():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "7", "--warmup", "3", "--config", "4", "--dry-launch"],
                       capture_output=True, text=True, env=clean_env(), timeout=300)
    assert p.returncode == 0, p.stderr
    doc = json.loads(p.stdout.strip().splitlines()[-1])
    cmd = doc["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(BENCH)
    # the child gets this run's own arguments, minus the launcher-only flags
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "7", "--warmup", "3", "--config", "4"]
    assert doc["n_ranks"] == 2


def test_launch_fails_loudly_without_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this host has the GPUs")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=clean_env(), timeout=300)
    assert p.returncode != 0
    assert "needs 2 GPUs" in p.stderr and "nothing was launched" in p.stderr
    assert '{"metric"' not in p.stdout          # no line, no fallback


def test_parent_relays_the_childs_line_and_exit_code(monkeypatch, capsys):
    sys.path.insert(0, ROOT)
    import bench
    line = {"metric": "quartets counted/sec", "value": 1.0, "n_gpus": 2, "config": {}}
    child = "import sys; print('RCCL banner'); print(%r); sys.exit(%%d)" % json.dumps(line)

    class A:
        gpus, dry_launch, via_launcher = 2, False, False

    monkeypatch.setattr(bench, "visible_gpus", lambda: 2)
    monkeypatch.setattr(bench, "child_command", lambda n, argv, port=None: [sys.executable, "-c", child % 0])
    assert bench.launch(A, ["--gpus", "2"]) == 0
    out = capsys.readouterr()
    got = json.loads(out.out.strip().splitlines()[-1])
    assert got["value"] == 1.0 and got["config"]["launcher"]["ranks"] == 2
    assert "RCCL banner" in out.err and "RCCL banner" not in out.out      # the JSON line is the only thing on stdout
    # a failed child: its code comes back, no line is printed, nothing is retried
    monkeypatch.setattr(bench, "child_command", lambda n, argv, port=None: [sys.executable, "-c", child % 3])
    assert bench.launch(A, ["--gpus", "2"]) == 3
    assert '{"metric"' not in capsys.readouterr().out
    # a child that exits 0 without a line is a failure too
    monkeypatch.setattr(bench, "child_command", lambda n, argv, port=None: [sys.executable, "-c", "print('nothing')"])
    assert bench.launch(A, ["--gpus", "2"]) == 1


def test_a_rank_refuses_a_world_size_that_contradicts_gpus():
    env = clean_env()
    env.update({"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_the_roofline_prices_every_class_with_its_own_mode():
    """bench.parse_variant: the kernel variant string -> (depth bits, trees, mode) classes and the mean minimal instruction count
    per (quartet, 32 trees): 2(B+1)+2 for full binary trees, 2(B+1)+5 with missing taxa, 3(B+1)+3 / +7 for multifurcating trees."""
    sys.path.insert(0, ROOT)
    import bench
    mode, classes, ops = bench.parse_variant("gather/binary_full/bitslice_b4x2:2428+bitslice_b5x2:7572/count_u32", 10000)
    assert mode == "binary_full" and classes == [(4, 2428, "binary_full"), (5, 7572, "binary_full")]
    assert abs(ops - (12 * 2428 + 14 * 7572) / 10000) < 1e-12
    mode, classes, ops = bench.parse_variant("gather/binary_partial/bitslice_b5x2/count_u32", 1500)
    assert mode == "binary_partial" and classes == [(5, 1500, "binary_partial")] and ops == 17
    mode, classes, ops = bench.parse_variant("gather/mixed/binary_partial.bitslice_b5x2:1000+general_full.bitslice_b5x2:500/count_u32", 1500)
    assert mode == "mixed" and classes == [(5, 1000, "binary_partial"), (5, 500, "general_full")]
    assert abs(ops - (17 * 1000 + 21 * 500) / 1500) < 1e-12
    assert bench.parse_variant("gather/partial/bitslice_b4x2:1418+bitslice_b5x2:82/count_u16", 1500)[2] == (22 * 1418 + 25 * 82) / 1500
    assert bench.parse_variant("gather/partial/depth_u16/count_u32", 40)[2] is None          # byte-SWAR kernel: priced against HBM
    assert bench.parse_variant("scatter/atomic/count_u32", 40) == (None, [], None)


def test_scaling_fields_and_cli_phase_parsing():
    """The pieces of the N > 1 line that need no GPU: scaling efficiency from the same-workload count-only step, the phases of
    a (multi-GPU, --trace) CLI run from its stdout protocol and trace stamps, the GPU count from the KFD topology narrowed by
    *_VISIBLE_DEVICES, and the NUMA node the cpu_baseline child pins itself to."""
    sys.path.insert(0, ROOT)
    import bench
    f = bench.same_workload_scaling(value=7.2e14, world=8, units_per_rank_per_step=2.0e12, count_only_ms=20.0, ms_per_step=22.5)
    assert abs(f["rate_quartets_per_s"] - 1.0e14) < 1 and abs(f["scaling_efficiency"] - 0.9) < 1e-12 and f["collective_exposed_ms"] == 2.5
    out = "There are 7 evaluation trees.\nFinished counting quartets.\nIt took: 446000 microseconds.\nFinished computing scores.\nIt took: 30000 microseconds.\n"
    err = ("[trace] +     1.0 ms  main: arguments parsed\n[trace] +   400.0 ms  main: all GPUs counted\n"
           "[trace] +   400.1 ms  main: no communicator (peer access)\n[trace] +   416.5 ms  main: tables reduced\n")
    ph = bench.cli_phases(out, err)
    assert ph == {"counting_phase_ms": 446.0, "scoring_phase_ms": 30.0, "all_gpus_counted_at_ms": 400.0, "table_reduction_ms": 16.5}
    assert bench.cli_phases("It took: 5 microseconds.\n", "") is None
    assert bench.cli_phases(out, "")["counting_phase_ms"] == 446.0 and "table_reduction_ms" not in bench.cli_phases(out, "")
    have = bench.visible_gpus()
    assert have >= 0
    old = os.environ.get("HIP_VISIBLE_DEVICES")
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    try:
        assert bench.visible_gpus() == 0
    finally:
        if old is None:
            del os.environ["HIP_VISIBLE_DEVICES"]
        else:
            os.environ["HIP_VISIBLE_DEVICES"] = old
    cpus = bench.numa_node0_cpus()
    assert cpus is None or (len(cpus) >= 1 and set(cpus) <= set(os.sched_getaffinity(0)))


def test_auto_mode_follows_the_one_gpu_scaling_model():
    """bench.py --mode auto: the table-sharded mode where the table collective would cost more than table-sharding adds (configs[2]:
    34 GB of table per rank against a replicated 1.8 ms panel build), the tree-sharded mode where it does not (configs[3]: 100 000
    trees, a 2.1 GB table). When profiles/r06_scaling_model.json is there, the choice must agree with its measured per-rank steps."""
    sys.path.insert(0, ROOT)
    import bench
    for world in (2, 4, 8):
        mode, est = bench.auto_mode(512, 10000, world)
        assert mode == "table" and est["table_extra_ms"] < est["tree_collective_ms_ring_bound"], (world, est)
    mode, est = bench.auto_mode(256, 100000, 8)
    assert mode == "tree", est
    path = os.path.join(ROOT, "profiles", "r06_scaling_model.json")
    if os.path.exists(path):
        with open(path) as f:
            doc = json.load(f)
        for cfg_no, entry in doc["configs"].items():
            for n_ranks, tab in entry["table"].items():
                tree = entry["tree"][n_ranks]
                # measured per-rank steps + the ring-bound collective of the model
                better = "table" if tab["max_ms"] < tree["count_max_ms"] + tree["collective_model"]["ring_ms"] else "tree"
                got, _ = bench.auto_mode(entry["taxa"], entry["trees"], int(n_ranks))
                assert got == better, (cfg_no, n_ranks, got, tab["max_ms"], tree["count_max_ms"], tree["collective_model"]["ring_ms"])


def test_secondary_workload_children_and_the_fraction_helper(tmp_path):
    """config.secondary of the default line: the generator children write exactly the batch `bench.py --taxa n --trees m --collapse /
    --dropout / --mixed` would count (same generator, same seeds), and valu_frac prices a step like roofline.frac does."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from quartetscores_amd import flatten, synth
    n, m, seed = 16, 9, 2001
    ref_nw = synth.reference_tree(n, 2000)
    d, jobs = bench.start_secondary_generators(ref_nw, n, m, seed)
    ref = flatten.flatten_reference(ref_nw)
    assert [w["name"] for w, *_ in jobs] == ["collapse0.2", "dropout0.1", "mixed"]
    for w, p, sp, out in jobs:
        _o, err = p.communicate(timeout=300)
        assert p.returncode == 0, err
        z = np.load(out)
        if w["name"] == "mixed":
            sets = [synth.tree_set(n, 3, seed), synth.tree_set(n, 3, seed + 1, dropout=0.1), synth.tree_set(n, 3, seed + 2, collapse=0.2)]
            trees = [sets[i % 3][i // 3] for i in range(9)]
        else:
            trees = synth.tree_set(n, m, seed, **w["kw"])
        want = flatten.flatten_eval_trees(trees, ref.name_to_id)
        assert int(z["n_trees"][0]) == want.n_trees == m
        assert (z["leaf_off"] == want.leaf_off).all() and (z["leaf_ids"] == want.leaf_ids).all() and (z["adj_depth"] == want.adj_depth).all()
        assert len(w["label"]) <= 100
    # 1500 trees x C(512,4) quartets in 60 ms of general_full at 4 bits: 18 instructions per (quartet, 32 trees)
    units = 1500 * 2829877120
    frac, ops = bench.valu_frac("gather/general_full/bitslice_b4x2/count_u32/clamp:150", 1500, units, 60.0)
    assert ops == 18 and abs(frac - units * 18 / 32 / 0.060 / 1e12 / bench.VALU_PEAK_TLOPS) < 1e-12
    frac, ops = bench.valu_frac("gather/mixed/binary_partial.bitslice_b4x2:1000+general_full.bitslice_b4x2:500/count_u32/fused:1", 1500, units, 60.0)
    assert abs(ops - (15 * 1000 + 18 * 500) / 1500) < 1e-12 and frac > 0
    assert bench.valu_frac("gather/partial/depth_u8/count_u32", 10, 10, 1.0) == (None, None)
