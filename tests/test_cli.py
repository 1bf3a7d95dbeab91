"""The C++ host + CLI (quartetscores_amd/bin/QuartetScores): reference CLI surface
(QuartetScores.cpp:48-85) on CPU, end-to-end output against the oracle on GPU."""
import os
import re
import subprocess

import pytest

from oracle_api import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "quartetscores_amd", "bin", "QuartetScores")


def run(*args):
    return subprocess.run([BIN, *args], capture_output=True, text=True, timeout=300)


def cli_comments(path):
    """sorted per-edge annotations of an annotated Newick file"""
    return sorted(re.findall(r"\[([^\]]*)\]", open(path).read()))


def oracle_comments(ref_nw, eval_text):
    """what the annotations must be: the oracle's scores, formatted like quartet_newick_writer.hpp:164-187"""
    o = Oracle(ref_nw)
    o.count(eval_text)
    o.score()
    out = []
    inf = float("inf")
    for lq, qp, eqp in o.scores_by_bipartition().values():
        parts = []
        if qp is not None and lq != inf:          # the writer's qp-ic guard tests the LQ value (quirk Q6)
            parts.append("qp-ic:%f" % qp)
        if lq != inf:
            parts.append("lq-ic:%f" % lq)
        if eqp is not None and eqp != inf:
            parts.append("eqp-ic:%f" % eqp)
        if parts:
            out.append(";".join(parts))
    o.close()
    return sorted(out)


@pytest.fixture()
def d1_files(tmp_path, golden):
    r, e = tmp_path / "ref.nwk", tmp_path / "eval.nwk"
    r.write_text(golden["D1"]["ref"] + "\n")
    e.write_text("\n".join(golden["D1"]["eval"]) + "\n")
    return str(r), str(e)


def test_cli_argument_errors(d1_files, tmp_path):
    assert os.path.exists(BIN), "build the host first (__graft_entry__.build())"
    r, e = d1_files
    p = run("-r", r, "-e", e)
    assert p.returncode == 1 and "ERROR" in p.stderr and "-o" in p.stderr
    p = run("-r", r, "-e", e, "-o")
    assert p.returncode == 1 and "ERROR" in p.stderr
    existing = tmp_path / "exists.nwk"
    existing.write_text("x")
    p = run("-r", r, "-e", e, "-o", str(existing))
    assert p.returncode == 1 and "ERROR: The specified output file already exists." in p.stdout
    p = run("--version")
    assert p.returncode == 0 and "1.0.1" in p.stdout
    # malformed numeric values end like the other argument errors, not in std::terminate
    for flag in ("-t", "--device"):
        p = run("-r", r, "-e", e, "-o", str(tmp_path / "o.nwk"), flag, "four")
        assert p.returncode == 1 and "ERROR" in p.stderr and flag in p.stderr, (flag, p.returncode, p.stderr)


def test_cpp_host_flattening_matches_python_host(tmp_path):
    """quartetscores_amd/csrc/host (newick.hpp, flatten.hpp, ingest.hpp) hands the C-ABI the same arrays as
    the Python host, with 1 and with many ingest threads, for binary / multifurcating / partial / rooted trees,
    quoted labels, branch lengths and comments."""
    import numpy as np
    from quartetscores_amd import flatten, synth
    dump = os.path.join(ROOT, "quartetscores_amd", "bin", "flatten_dump")
    assert os.path.exists(dump)
    n = 23
    ref_nw = synth.reference_tree(n, 5)
    trees = (synth.tree_set(n, 150, 6) + synth.tree_set(n, 60, 7, collapse=0.3, dropout=0.2)
             + synth.tree_set(n, 40, 8, rooted=True) + ["((t0:0.1,'t1':2e-3)x:1,[c](t2,t3)[d],t4);", "(t5,t6);", "t7;"])
    r, e = tmp_path / "r.nwk", tmp_path / "e.nwk"
    r.write_text(ref_nw + "\n")
    e.write_text("\n".join(trees) + "\n\n")
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    outs = []
    for threads in ("1", "7"):
        p = subprocess.run([dump, str(r), str(e), threads], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr
        d = {ln.split(" ", 1)[0]: ln.split(" ")[1:] for ln in p.stdout.strip().split("\n")}
        outs.append(p.stdout)
        assert d["names"] == ref.names
        assert [int(x) for x in d["parent"]] == list(ref.parent)
        assert [int(x) for x in d["leaf_node"]] == list(ref.leaf_node)
        assert int(d["n_trees"][0]) == batch.n_trees == len(trees)
        for key, arr in (("leaf_off", batch.leaf_off), ("leaf_ids", batch.leaf_ids), ("adj_depth", batch.adj_depth),
                         ("node_off", batch.node_off), ("rng_off", batch.rng_off), ("ranges", batch.ranges)):
            assert [int(x) for x in d.get(key, [])] == [int(x) for x in arr], key
    assert outs[0] == outs[1]  # the threaded ingest keeps file order
    bad = tmp_path / "bad.nwk"
    bad.write_text("\n".join(trees[:70]) + "\n((t0,t1),(t2,zzz),(t3,t4));\n")
    p = subprocess.run([dump, str(r), str(bad), "4"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "unknown taxon 'zzz'" in p.stderr and "tree 70" in p.stderr


def test_native_ingest_on_deep_and_decorated_trees_matches_python_host():
    """The C++ ingest finds the centre of a tree with two linear sweeps over its pre-order arrays instead of two BFS passes,
    and scans labels / branch lengths through a character table: the arrays must stay those of the Python flattening (which
    runs the BFS) on random trees of several sizes, ladders (the longest possible paths, centre ties), ladder + NNI trees,
    multifurcating / partial / rooted trees, and with branch lengths, comments, quoted labels and white space in the text."""
    import sys
    import numpy as np
    from quartetscores_amd import flatten, native_ingest, synth
    sys.setrecursionlimit(100000)
    for n, m, kw in ((5, 120, {}), (8, 120, {}), (19, 150, {}), (19, 80, dict(collapse=0.3, dropout=0.2)), (19, 40, dict(rooted=True)),
                     (64, 80, {}), (150, 60, {}), (150, 30, dict(collapse=0.4)), (150, 30, dict(dropout=0.3)), (300, 12, {})):
        ref_nw = synth.reference_tree(n, 300 + n)
        trees = synth.tree_set(n, m, 400 + n, **kw)
        if 19 <= n <= 150:
            lad = f"(t{n - 2},t{n - 1})"
            for i in range(n - 3, -1, -1):
                lad = f"(t{i},{lad})"
            trees.append(lad + ";")
            trees += list(synth.nni_tree_set(lad + ";", 15, 5))
        rng = np.random.default_rng(n)

        def deco(t):
            t = "".join(ch + (" " if ch in ",)" and rng.random() < 0.3 else "") for ch in t)
            if rng.random() < 0.5:
                t = t.replace(")", "):0.%d" % rng.integers(1, 99999), 3)
            if rng.random() < 0.3:
                t = t.replace("t1,", "'t1'[a comment],", 1)
            return t
        trees = [deco(t) for t in trees]
        ref = flatten.flatten_reference(ref_nw)
        want = flatten.flatten_eval_trees(trees, ref.name_to_id)
        got, _ = native_ingest.ingest_text(ref_nw, ("\n".join(trees) + "\n").encode(), threads=3)
        for f in ("leaf_off", "leaf_ids", "adj_depth", "node_off", "rng_off", "ranges"):
            assert np.array_equal(getattr(got, f), getattr(want, f)), (n, kw, f)


def test_native_ingest_binding_matches_python_host(tmp_path):
    """quartetscores_amd.native_ingest (libquartetscores_host.so = the C++ host's ingest behind a C interface) gives
    the Python flattening's arrays, for any tree range and thread count; errors carry the tree index."""
    import numpy as np
    from quartetscores_amd import flatten, native_ingest, synth
    assert native_ingest.available(), "build the host first (__graft_entry__.build())"
    n = 19
    ref_nw = synth.reference_tree(n, 15)
    trees = synth.tree_set(n, 130, 16) + synth.tree_set(n, 50, 17, collapse=0.3, dropout=0.2) + synth.tree_set(n, 20, 18, rooted=True)
    r, e = tmp_path / "r.nwk", tmp_path / "e.nwk"
    r.write_text(ref_nw + "\n")
    e.write_text("\n".join(trees) + "\n")
    ref = flatten.flatten_reference(ref_nw)
    want = flatten.flatten_eval_trees(trees, ref.name_to_id)
    for lo, hi, th in ((0, native_ingest.ALL, 1), (0, len(trees), 5), (37, 141, 3), (199, 200, 2), (200, 500, 2), (10, 10, 1)):
        got, total = native_ingest.ingest(str(r), str(e), lo, hi, th)
        assert total == len(trees)
        exp = want.slice(min(lo, len(trees)), min(hi, len(trees)))
        assert got.n_trees == exp.n_trees
        for f in ("leaf_off", "leaf_ids", "adj_depth", "node_off", "rng_off", "ranges"):
            assert np.array_equal(getattr(got, f), getattr(exp, f)), (f, lo, hi, th)
    lean, _ = native_ingest.ingest(str(r), str(e), 5, 60, 2, want_ranges=False)   # what the gather path needs
    exp = want.slice(5, 60)
    assert np.array_equal(lean.leaf_ids, exp.leaf_ids) and np.array_equal(lean.adj_depth, exp.adj_depth)
    assert lean.ranges.size == 0 and not lean.node_off.any()
    bad = tmp_path / "bad.nwk"
    bad.write_text("\n".join(trees[:70]) + "\n((t0,t1),(t2,zzz),(t3,t4));\n")
    with pytest.raises(native_ingest.IngestError) as ei:
        native_ingest.ingest(str(r), str(bad), 0, native_ingest.ALL, 4)
    assert "zzz" in str(ei.value) and "tree 70" in str(ei.value)
    with pytest.raises(native_ingest.IngestError):
        native_ingest.ingest(str(r), str(tmp_path / "missing.nwk"))


def test_native_ingest_on_decorated_newick(tmp_path):
    """Randomly decorated input (branch lengths, comments -- also with ';' and quotes inside --, quoted labels, blank
    lines, line breaks inside a tree, a last tree without ';'): the allocation-free C++ parser, its memchr tree
    splitter and the Python host agree on every array."""
    import random
    import re
    import numpy as np
    from quartetscores_amd import flatten, native_ingest, synth
    rng = random.Random(4242)
    n = 17
    ref_nw = synth.reference_tree(n, 25)
    ref = flatten.flatten_reference(ref_nw)
    plain = synth.tree_set(n, 120, 26, collapse=0.2, dropout=0.15) + synth.tree_set(n, 30, 27, rooted=True)

    def decorate(t):
        out = []
        for tok in re.split(r"([(),;])", t):
            if re.fullmatch(r"t\d+", tok or ""):
                if rng.random() < 0.3:
                    tok = "'" + tok + "'"
                if rng.random() < 0.5:
                    tok += ":%g" % rng.uniform(0, 2)
                if rng.random() < 0.2:
                    tok += rng.choice(["[c]", "[x;y]", "[it's]", "[&&NHX:S=a]"])
            elif tok == ")" and rng.random() < 0.3:
                tok = ")" + rng.choice(["", "n1", "'inner node'"]) + (":%g" % rng.uniform(0, 1) if rng.random() < 0.5 else "")
            elif tok == "," and rng.random() < 0.1:
                tok = ",\n  "
            out.append(tok or "")
        return "".join(out)

    trees = [decorate(t) for t in plain]
    text = "\n\n".join(trees)
    text = text.rstrip()
    assert text.endswith(";")
    text = text[:-1] + "\n"          # the last tree lacks its ';'
    r, e = tmp_path / "r.nwk", tmp_path / "e.nwk"
    r.write_text(ref_nw + "\n")
    e.write_text(text)
    want = flatten.flatten_eval_trees(trees, ref.name_to_id)
    for th in (1, 4):
        got, total = native_ingest.ingest(str(r), str(e), 0, native_ingest.ALL, th)
        assert total == len(trees) == got.n_trees
        for f in ("leaf_off", "leaf_ids", "adj_depth", "node_off", "rng_off", "ranges"):
            assert np.array_equal(getattr(got, f), getattr(want, f)), f


def test_apostrophes_inside_unquoted_labels_do_not_merge_trees(tmp_path):
    """A quote opens a quoted label only where a label can start; O'Brien is an ordinary unquoted label, so the ';'
    behind it still ends its tree (the splitter once swallowed it and merged the trees)."""
    import numpy as np
    from quartetscores_amd import flatten, native_ingest
    ref_nw = "((O'Brien,b),(c,d),(e,(f,'g;h')));"
    trees = ["((O'Brien,c),(b,d),(e,(f,'g;h')))", "(('g;h',b),(c,O'Brien),(e,(f,d)))", "((e,b),(c,d),(O'Brien,(f,'g;h')))"]
    text = ";\n".join(trees) + ";\n"
    ref = flatten.flatten_reference(ref_nw)
    assert "O'Brien" in ref.names and "g;h" in ref.names
    want = flatten.flatten_eval_trees([t + ";" for t in trees], ref.name_to_id)
    got, total = native_ingest.ingest_text(ref_nw, text)
    assert total == 3 and got.n_trees == 3
    for f in ("leaf_off", "leaf_ids", "adj_depth", "node_off", "rng_off", "ranges"):
        assert np.array_equal(getattr(got, f), getattr(want, f)), f


def test_cli_fails_loudly_without_gpu(d1_files, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r, e = d1_files
    p = run("-r", r, "-e", e, "-o", str(tmp_path / "o.nwk"))
    assert p.returncode == 1 and "no CPU fallback" in p.stderr
    assert not (tmp_path / "o.nwk").exists()


def test_multi_gpu_driver_fails_loudly_without_gpu(d1_files, tmp_path):
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r, e = d1_files
    q = subprocess.run([sys.executable, "-m", "quartetscores_amd.dist_cli", "-r", r, "-e", e, "-o", str(tmp_path / "o.nwk")],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert q.returncode == 1 and "no CPU fallback" in q.stderr
    assert not (tmp_path / "o.nwk").exists()


def test_multi_gpu_driver_annotation_format():
    """The comment layout of dist_cli equals the C++ CLI's (QuartetScores.cpp in csrc/host): parts omitted at +inf,
    the qp-ic guard on the LQ value, std::to_string's %f."""
    import math
    from quartetscores_amd import dist_cli, flatten
    ref = flatten.flatten_reference("((a,b)x,(c,d),e);")
    n = ref.n_nodes
    lq = [math.inf] * n; qp = [math.inf] * n; eqp = [math.inf] * n
    inner = [i for i, nd in enumerate(ref.nodes) if nd.children and i != 0]
    lq[inner[0]], qp[inner[0]], eqp[inner[0]] = 0.5, 0.25, -0.125
    lq[inner[1]] = 1.0   # qp/eqp left at +inf: eqp omitted, qp printed because the guard looks at lq
    text = dist_cli._annotate(ref, lq, qp, eqp, True)
    assert "[qp-ic:0.250000;lq-ic:0.500000;eqp-ic:-0.125000]" in text
    assert "[qp-ic:inf;lq-ic:1.000000]" in text
    assert text.count("[") == 2
    text2 = dist_cli._annotate(ref, lq, qp, eqp, False)   # multifurcating reference: LQ-IC only
    assert "[lq-ic:0.500000]" in text2 and "qp-ic" not in text2


@pytest.mark.gpu
def test_cli_end_to_end_matches_oracle(d1_files, tmp_path, golden):
    r, e = d1_files
    out, raw = tmp_path / "o.nwk", tmp_path / "raw.txt"
    p = run("-r", r, "-e", e, "-o", str(out), "-q", str(raw), "-v", "-s")
    assert p.returncode == 0, p.stderr
    for line in ("There are 20 evaluation trees.", "The reference tree has 8 taxa.", "Using memory-efficient Lookup table",
                 "lookup table size in bytes: 420", "Finished counting quartets.", "The reference tree is bifurcating.",
                 "Finished computing scores.", "Elapsed time:"):
        assert line in p.stdout, line
    text = out.read_text()
    comments = re.findall(r"\[([^\]]*)\]", text)
    o = Oracle(golden["D1"]["ref"])
    o.count("\n".join(golden["D1"]["eval"]))
    o.score()
    want = sorted("qp-ic:%f;lq-ic:%f;eqp-ic:%f" % (v[1], v[0], v[2]) for v in o.scores_by_bipartition().values())
    assert sorted(comments) == want
    # the tree itself round-trips (names and topology untouched)
    assert re.sub(r"\[[^\]]*\]", "", text).strip() == golden["D1"]["ref"]
    # raw QIC dump: one line per quartet of the 8 taxa (binary reference -> all 70 resolved)
    ora_raw = tmp_path / "ora.txt"
    o.raw_qic(str(ora_raw))

    def canon(path):
        d = {}
        for line in open(path):
            lab, val = line.strip().split("): ")
            l, rr = lab[1:].split("|")
            d[frozenset([frozenset(l.split(",")), frozenset(rr.split(","))])] = val
        return d
    assert canon(str(raw)) == canon(str(ora_raw)) and len(canon(str(raw))) == 70
    assert raw.read_text() == ora_raw.read_text()      # byte-identical: the reference's line and label order
    raw2 = tmp_path / "raw_rank.txt"                      # table-rank order: the same lines in another order
    p = run("-r", r, "-e", e, "-o", str(tmp_path / "o_rank.nwk"), "-q", str(raw2), "--qic-rank-order")
    assert p.returncode == 0, p.stderr
    assert sorted(raw2.read_text().splitlines()) == sorted(raw.read_text().splitlines()) and raw2.read_text() != raw.read_text()
    # binary sidecar of the raw QIC dump: same quartets, same topologies, QIC equal to the printed %g value
    import struct
    import numpy as np
    rawb = tmp_path / "raw.bin"
    p = run("-r", r, "-e", e, "-o", str(tmp_path / "ob.nwk"), "--qic-binary", str(rawb))
    assert p.returncode == 0, p.stderr
    blob = rawb.read_bytes()
    assert blob[:8] == b"QSQIC01\0"
    n_taxa, _, nq = struct.unpack_from("<IIQ", blob, 8)
    assert (n_taxa, nq) == (8, 70)
    off, names = 24, []
    for _ in range(n_taxa):
        (ln,) = struct.unpack_from("<I", blob, off)
        names.append(blob[off + 4:off + 4 + ln].decode()); off += 4 + ln
    topo = np.frombuffer(blob, dtype=np.uint8, count=nq, offset=off)
    qic = np.frombuffer(blob[off + nq:off + nq + 8 * nq], dtype="<f8")
    text_vals = canon(str(raw))
    from itertools import combinations
    quads = sorted(combinations(range(n_taxa), 4), key=lambda q: (q[3], q[2], q[1], q[0]))
    assert len(quads) == nq and set(topo.tolist()) <= {0, 2}
    for rk, (s0, s1, s2, s3) in enumerate(quads):
        A, B, C_, D = names[s0], names[s1], names[s2], names[s3]
        key = frozenset([frozenset([A, B]), frozenset([C_, D])]) if topo[rk] == 0 else frozenset([frozenset([A, D]), frozenset([B, C_])])
        assert key in text_vals and "%g" % qic[rk] == text_vals[key]
    # table persistence: save after counting, reload instead of counting -> the same annotated tree
    tab = tmp_path / "table.bin"
    out2, out3 = tmp_path / "o2.nwk", tmp_path / "o3.nwk"
    p = run("-r", r, "-e", e, "-o", str(out2), "-t", "3", "--save-table", str(tab))
    assert p.returncode == 0 and tab.stat().st_size == 70 * 3 * 2
    p = run("-r", r, "-e", e, "-o", str(out3), "--load-table", str(tab))
    assert p.returncode == 0 and out3.read_text() == out2.read_text() == text
    # unknown taxon -> error exit, like the reference's uncaught std::out_of_range
    bad = tmp_path / "bad.nwk"
    bad.write_text(golden["D6"]["bad_tree"] + "\n")
    p = run("-r", r, "-e", str(bad), "-o", str(tmp_path / "o4.nwk"))
    assert p.returncode == 1 and "unknown taxon" in p.stderr


@pytest.mark.gpu
def test_multi_gpu_driver_single_process_matches_the_cli(d1_files, tmp_path):
    """quartetscores_amd.dist_cli (tree-sharded count + reduce-scatter + sharded scoring; here one process, no
    launcher) writes the same annotated tree as the single-GPU C++ CLI, for every wire format."""
    import sys
    r, e = d1_files
    ref_out = tmp_path / "cli.nwk"
    p = run("-r", r, "-e", e, "-o", str(ref_out))
    assert p.returncode == 0, p.stderr
    want = ref_out.read_text()
    for wire in ("auto", "u16x2", "u16", "u32"):
        out = tmp_path / f"dist_{wire}.nwk"
        q = subprocess.run([sys.executable, "-m", "quartetscores_amd.dist_cli", "-r", r, "-e", e, "-o", str(out), "-v", "--wire", wire],
                           capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert q.returncode == 0, q.stderr
        assert "There are 20 evaluation trees." in q.stdout and "Finished computing scores." in q.stdout
        assert out.read_text() == want, wire
    # existing output file -> refused like the CLI
    q = subprocess.run([sys.executable, "-m", "quartetscores_amd.dist_cli", "-r", r, "-e", e, "-o", str(ref_out)],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert q.returncode == 1 and "already exists" in q.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("trees,rooted", [(300, False), (70000, False), (120, True)])
def test_cpp_cli_gpus_path_matches_single_gpu(tmp_path, trees, rooted):
    """QuartetScores --gpus 1 runs the multi-GPU host (multi_gpu.hpp: per-GPU threads, table in attached memory padded
    to N chunks, ncclCommInitAll + reduce-scatter / all-reduce over RCCL, sharded scoring reduced on the host) on one
    device: annotated tree and -q file identical to the single-GPU path; u16 (packed words) and u32 tables."""
    import numpy as np
    from quartetscores_amd import native_ingest, synth
    n = 14
    ref_nw = synth.random_tree(n, np.random.default_rng(901), rooted=rooted)
    text = native_ingest.synth_trees(n, trees, 902)
    r, e = tmp_path / "r.nwk", tmp_path / "e.nwk"
    r.write_text(ref_nw + "\n")
    e.write_bytes(text)
    o1, o2, o3 = tmp_path / "o1.nwk", tmp_path / "o2.nwk", tmp_path / "o3.nwk"
    q1, q3 = tmp_path / "q1.txt", tmp_path / "q3.txt"
    p = run("-r", str(r), "-e", str(e), "-o", str(o1), "-q", str(q1))
    assert p.returncode == 0, p.stderr
    p = run("-r", str(r), "-e", str(e), "-o", str(o2), "--gpus", "1")                      # reduce-scatter + sharded scoring
    assert p.returncode == 0, p.stderr
    assert "reduce-scatter" in p.stdout and o2.read_text() == o1.read_text()
    p = run("-r", str(r), "-e", str(e), "-o", str(o3), "--gpus", "1", "-q", str(q3))        # all-reduce, GPU 0 scores and dumps
    assert p.returncode == 0, p.stderr
    assert "all-reduce" in p.stdout and o3.read_text() == o1.read_text() and q3.read_text() == q1.read_text()
    p = run("-r", str(r), "-e", str(e), "-o", str(tmp_path / "o4.nwk"), "--gpus", "9")
    assert p.returncode == 1 and "device(s) visible" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("driver", ["single", "gpus", "shards"])
def test_cli_savemem_with_a_rooted_reference_ends_like_the_reference(tmp_path, driver):
    """`-s` and a rooted reference tree: the reference's compact table throws std::runtime_error ("id = ..., but
    quartet_lookup_.size() = ...", quartet_lookup_table.hpp:79-85) during the scoring of the root's node pairs and its run ends
    without an output file. All three drivers of the CLI end the same way (the text = the committed fixture = the oracle's);
    without -s the run succeeds, and -s with --root-as-edge too. No note about unreproduced behaviour is printed any more."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "rooted_compact.json")) as f:
        fx = json.load(f)["rooted24"]
    r, e = tmp_path / "r.nwk", tmp_path / "e.nwk"
    r.write_text(fx["ref"] + "\n")
    e.write_text("\n".join(fx["eval"]) + "\n")
    extra = {"single": [], "gpus": ["--gpus", "1"], "shards": ["--gpus", "1", "--table-shards", "3"]}[driver]
    out = tmp_path / "o.nwk"
    p = run("-r", str(r), "-e", str(e), "-o", str(out), "-s", *extra)
    assert p.returncode == 1 and not out.exists()
    assert "ERROR: " + fx["reference_throws"] in p.stderr
    assert "Finished counting quartets." in p.stdout           # the reference counts first, then dies in the scoring loop
    assert "Note: -s with a rooted reference tree" in p.stderr  # round 5: known from the tree alone (qs_score_check), said before the counting
    if driver == "single":                                      # --fail-fast ends the run at the note, with the same error text
        pf = run("-r", str(r), "-e", str(e), "-o", str(out), "-s", "--fail-fast")
        assert pf.returncode == 1 and not out.exists() and "ERROR: " + fx["reference_throws"] in pf.stderr
        assert "Finished counting quartets." not in pf.stdout and "Counting" not in pf.stdout
    ok = tmp_path / "ok.nwk"
    p = run("-r", str(r), "-e", str(e), "-o", str(ok), *extra)
    assert p.returncode == 0 and ok.exists() and "NOT reproduced" not in p.stderr and "Note:" not in p.stderr
    ok2 = tmp_path / "ok2.nwk"
    p = run("-r", str(r), "-e", str(e), "-o", str(ok2), "-s", "--root-as-edge", *extra)
    assert p.returncode == 0 and ok2.exists()


@pytest.mark.gpu
@pytest.mark.parametrize("trees,kind", [(300, "binary"), (70000, "binary"), (200, "mixed")])
def test_cpp_cli_peer_access_reduce_matches_single_gpu(tmp_path, trees, kind):
    """`QuartetScores --gpus N --reduce p2p`: the tree-sharded tables are reduced without a communicator -- every GPU sums
    its chunk of every peer's table with plain loads (qs_sum_words over peer-mapped memory). On a 1-GPU box the N-way
    partition runs as N contexts on one device (--gpus-on-one-device): 3 "GPUs", reduce-scatter of the three-cell u16 table
    (300 trees), the two-cell u32 wire words (70000 binary full trees) and a table of mixed trees; with -q the reduce to GPU 0
    (the whole table on one device). Output byte-identical to the single-GPU CLI."""
    import numpy as np
    from quartetscores_amd import native_ingest, synth
    n = 15
    ref_nw = synth.random_tree(n, np.random.default_rng(911))
    if kind == "binary":
        text = native_ingest.synth_trees(n, trees, 912)
    else:
        text = ("\n".join(synth.tree_set(n, trees // 2, 913, collapse=0.2, dropout=0.1) + synth.tree_set(n, trees - trees // 2, 914)) + "\n").encode()
    r, e = tmp_path / "r.nwk", tmp_path / "e.nwk"
    r.write_text(ref_nw + "\n")
    e.write_bytes(text)
    o1, q1 = tmp_path / "o1.nwk", tmp_path / "q1.txt"
    p = run("-r", str(r), "-e", str(e), "-o", str(o1), "-q", str(q1))
    assert p.returncode == 0, p.stderr
    for gpus in ("1", "3"):
        o2 = tmp_path / f"o2_{gpus}.nwk"
        p = run("-r", str(r), "-e", str(e), "-o", str(o2), "--gpus", gpus, "--reduce", "p2p", "--gpus-on-one-device")
        assert p.returncode == 0, p.stderr
        assert "peer-access reduce-scatter" in p.stdout and o2.read_text() == o1.read_text(), gpus
        if kind == "binary" and trees >= 65536 and gpus == "3":
            assert "two u32 cells per tuple" in p.stdout
        o3, q3 = tmp_path / f"o3_{gpus}.nwk", tmp_path / f"q3_{gpus}.txt"
        p = run("-r", str(r), "-e", str(e), "-o", str(o3), "--gpus", gpus, "--reduce", "p2p", "--gpus-on-one-device", "-q", str(q3))
        assert p.returncode == 0, p.stderr
        assert "peer-access all-reduce" in p.stdout and o3.read_text() == o1.read_text() and q3.read_text() == q1.read_text(), gpus
    # RCCL refuses two ranks on one device: the test hook says so instead of hanging
    p = run("-r", str(r), "-e", str(e), "-o", str(tmp_path / "o4.nwk"), "--gpus", "2", "--gpus-on-one-device")
    assert p.returncode == 1 and "needs --reduce p2p" in p.stderr


@pytest.mark.gpu
def test_insufficient_memory_is_reported_like_the_reference(tmp_path):
    """A count table that does not fit the device: the library reports QS_ERR_OOM with the reference's message
    ("Insufficient memory!", QuartetScoreComputer.hpp:724-745 throws it as a runtime_error) instead of crashing, the
    context stays usable, and the CLI ends with that message and a non-zero status. 1200 taxa = 3 * C(1200,4) * 2 B = 515 GB."""
    from quartetscores_amd import engine, synth
    ctx = engine.Context(1200, 16)
    with pytest.raises(engine.QSError) as ei:
        ctx.table_alloc()
    assert ei.value.code == -3 and "Insufficient memory!" in str(ei.value)
    ctx.close()
    small = engine.Context(12, 16)      # the device is fine afterwards
    small.table_alloc()
    small.close()
    n = 1200
    ref = tmp_path / "big_ref.nwk"
    ref.write_text(synth.reference_tree(n, 31) + "\n")
    ev = tmp_path / "big_eval.nwk"
    ev.write_text(synth.reference_tree(n, 32) + "\n")
    out = tmp_path / "big_out.nwk"
    p = run("-r", str(ref), "-e", str(ev), "-o", str(out))
    assert p.returncode != 0 and "Insufficient memory!" in (p.stderr + p.stdout)
    assert not out.exists()


@pytest.mark.gpu
@pytest.mark.parametrize("trees,rooted,kind", [(90, False, "mixed"), (70000, False, "binary"), (150, True, "binary")])
def test_cpp_cli_table_shards_match_the_whole_table_run(tmp_path, trees, rooted, kind):
    """QuartetScores --table-shards K (table_shards.hpp): the count table passes through ONE GPU in K shards by largest
    taxon id -- count all trees into the shard, score pass 1, spill the shard to host memory or drop it and count it
    again, score pass 2 against the global minima -- and the annotated tree is identical to the whole-table run: u16 and
    u32 tables, partial / multifurcating evaluation trees, a rooted reference (the (root, v) pair sums add up over the
    shards), more shards than largest ids can fill (empty ones are dropped), --table-shards 0 (as many as needed: one)."""
    import numpy as np
    from quartetscores_amd import native_ingest, synth
    n = 19
    ref_nw = synth.random_tree(n, np.random.default_rng(911), rooted=rooted)
    r, e = tmp_path / "r.nwk", tmp_path / "e.nwk"
    r.write_text(ref_nw + "\n")
    if kind == "binary":
        e.write_bytes(native_ingest.synth_trees(n, trees, 912))
    else:
        e.write_text("\n".join(synth.tree_set(n, trees, 913, collapse=0.2, dropout=0.15)) + "\n")
    o1 = tmp_path / "o1.nwk"
    p = run("-r", str(r), "-e", str(e), "-o", str(o1))
    assert p.returncode == 0, p.stderr
    want = o1.read_text()
    # the whole-table run itself against the oracle (not only CLI against CLI); the rooted case goes through the
    # reference's degree-2-root handling, which the oracle restates too
    if trees <= 1000:
        assert cli_comments(o1) == oracle_comments(ref_nw, e.read_text())
    for i, extra in enumerate((["--table-shards", "3", "--spill", "host"], ["--table-shards", "3", "--spill", "recount"],
                               ["--table-shards", "40"], ["--table-shards", "1"], ["--table-shards", "0"],
                               ["--table-shards", "5", "--gpus", "1"])):
        o = tmp_path / f"s{i}.nwk"
        p = run("-r", str(r), "-e", str(e), "-o", str(o), *extra)
        assert p.returncode == 0, (extra, p.stderr)
        assert o.read_text() == want, extra
        if extra[1] not in ("0",):
            assert "table shard(s) by largest taxon id" in p.stdout and "Finished computing scores." in p.stdout
            if extra[1] == "1":
                assert "every shard stays on its GPU" in p.stdout
            elif "--spill" in extra:
                assert ("kept in host memory" in p.stdout) == ("host" in extra)
                assert ("counted again" in p.stdout) == ("recount" in extra)
            else:   # automatic: whichever is cheaper (a host round trip of the table against counting the trees again)
                assert ("kept in host memory" in p.stdout) != ("counted again" in p.stdout)
    p = run("-r", str(r), "-e", str(e), "-o", str(tmp_path / "x.nwk"), "--table-shards", "2", "-q", str(tmp_path / "x.txt"))
    assert p.returncode == 1 and "need the whole table" in p.stderr
    p = run("-r", str(r), "-e", str(e), "-o", str(tmp_path / "y.nwk"), "--table-shards", "2", "--spill", "disk")
    assert p.returncode == 1 and "--spill takes" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ref_kind", ["binary", "multifurcating", "rooted"])
def test_cpp_cli_table_shards_on_gpus_match_the_oracle(tmp_path, ref_kind):
    """QuartetScores --gpus N --table-shards K (BASELINE configs[4]'s mode; here N = 1 device, K = 8 shards, shard s on
    GPU s mod N): every shard counts ALL trees, no table collective, pass-1 sums / minima and the pass-2 candidates are
    combined over the shards, qs_score_finish once -- and the annotated tree carries the ORACLE's scores, for evaluation
    trees with collapsed edges and missing taxa, on a binary, a multifurcating and a rooted reference tree; spill to
    host memory, recount, and --gpus 1 alone with a forced shard count agree."""
    import numpy as np
    from quartetscores_amd import synth
    n = 41
    rng = np.random.default_rng(4100)
    ref_nw = synth.random_tree(n, rng, rooted=(ref_kind == "rooted"))
    if ref_kind == "multifurcating":
        ref_nw = synth.tree_set(n, 1, 4101, collapse=0.3)[0]
    trees = synth.tree_set(n, 120, 4102) + synth.tree_set(n, 150, 4103, collapse=0.25, dropout=0.1)
    r, e = tmp_path / "r.nwk", tmp_path / "e.nwk"
    r.write_text(ref_nw + "\n")
    e.write_text("\n".join(trees) + "\n")
    want = oracle_comments(ref_nw, e.read_text())
    assert len(want) >= 10
    for i, extra in enumerate((["--gpus", "1", "--table-shards", "8"], ["--gpus", "1", "--table-shards", "8", "--spill", "recount"],
                               ["--table-shards", "8", "--spill", "host"], ["--gpus", "1", "--table-shards", "1"])):
        o = tmp_path / f"g{i}.nwk"
        p = run("-r", str(r), "-e", str(e), "-o", str(o), *extra)
        assert p.returncode == 0, (extra, p.stderr)
        assert "no table collective" in p.stdout and "Finished computing scores." in p.stdout
        assert cli_comments(o) == want, extra
    # more GPUs than the box has: a clean error before any counting
    p = run("-r", str(r), "-e", str(e), "-o", str(tmp_path / "z.nwk"), "--gpus", "64", "--table-shards", "64")
    assert p.returncode == 1 and "device(s) visible" in p.stderr


@pytest.mark.gpu
def test_cpp_cli_gpus_mode_tree_and_table_agree_with_the_oracle(tmp_path):
    """`QuartetScores --gpus 3 --mode tree | table | auto` (the 3 "GPUs" are 3 host threads with their own contexts on
    the one device of the box: --gpus-on-one-device): the tree-sharded route (trees / 3 per GPU, peer-access reduce-scatter,
    sharded scoring) and the table-sharded route (all trees per GPU into its cost-balanced shard by largest taxon id, no table
    collective) write the SAME annotated tree = the oracle's scores; -q needs the whole table on one device, so `--mode table`
    with -q says so and runs tree-sharded."""
    import numpy as np
    from quartetscores_amd import synth
    n = 37
    ref_nw = synth.random_tree(n, np.random.default_rng(6100))
    trees = synth.tree_set(n, 90, 6101) + synth.tree_set(n, 60, 6102, collapse=0.2, dropout=0.1)
    r, e = tmp_path / "r.nwk", tmp_path / "e.nwk"
    r.write_text(ref_nw + "\n")
    e.write_text("\n".join(trees) + "\n")
    want = oracle_comments(ref_nw, e.read_text())
    assert len(want) >= 10
    common = ["--gpus", "3", "--reduce", "p2p", "--gpus-on-one-device"]
    outs = {}
    for mode in ("tree", "table"):
        o = tmp_path / f"{mode}.nwk"
        p = run("-r", str(r), "-e", str(e), "-o", str(o), *common, "--mode", mode)
        assert p.returncode == 0, (mode, p.stderr)
        assert ("no table collective" in p.stdout) == (mode == "table") and ("peer-access reduce-scatter" in p.stdout) == (mode == "tree"), p.stdout
        assert cli_comments(o) == want, mode
        outs[mode] = o.read_text()
    assert outs["tree"] == outs["table"]
    q = tmp_path / "q.txt"
    p = run("-r", str(r), "-e", str(e), "-o", str(tmp_path / "q.nwk"), *common, "--mode", "table", "-q", str(q))
    assert p.returncode == 0 and "tree-sharded mode instead" in p.stdout and "peer-access all-reduce" in p.stdout, p.stdout
    assert (tmp_path / "q.nwk").read_text() == outs["tree"] and q.stat().st_size > 0
    p = run("-r", str(r), "-e", str(e), "-o", str(tmp_path / "bad.nwk"), "--gpus", "2", "--mode", "sideways")
    assert p.returncode == 1 and "--mode takes" in p.stderr


@pytest.mark.gpu
def test_python_multi_gpu_driver_with_two_ranks_on_one_gpu_both_modes(tmp_path):
    """quartetscores_amd.dist_cli under `python -m torch.distributed.run --nproc-per-node 2` with QS_DIST_BACKEND=gloo: two REAL ranks share
    cuda:0 (collectives staged through the host, quartetscores_amd/collectives.py). `--mode tree` (trees / 2 per rank, reduce-scatter of
    the wire words, sharded scoring), `--mode table` (all trees per rank into its cost-balanced shard, no table collective) and `--mode
    auto` write the annotated tree the single-GPU C++ CLI writes, for binary trees and for a batch of mixed shapes."""
    import socket
    import sys
    import numpy as np
    from quartetscores_amd import native_ingest, synth
    n = 33
    ref_nw = synth.random_tree(n, np.random.default_rng(7100))
    r = tmp_path / "r.nwk"
    r.write_text(ref_nw + "\n")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["QS_DIST_BACKEND"] = "gloo"
    for kind in ("binary", "mixed"):
        e = tmp_path / f"e_{kind}.nwk"
        if kind == "binary":
            e.write_bytes(native_ingest.synth_trees(n, 500, 7101))
        else:
            e.write_text("\n".join(synth.tree_set(n, 70, 7102) + synth.tree_set(n, 90, 7103, collapse=0.2, dropout=0.1)) + "\n")
        ref_out = tmp_path / f"cli_{kind}.nwk"
        p = run("-r", str(r), "-e", str(e), "-o", str(ref_out))
        assert p.returncode == 0, p.stderr
        for mode in ("tree", "table", "auto"):
            out = tmp_path / f"dist_{kind}_{mode}.nwk"
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
            s.close()
            q = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                                "--master-port", str(port), "-m", "quartetscores_amd.dist_cli", "-r", str(r), "-e", str(e), "-o", str(out), "--mode", mode],
                               capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
            assert q.returncode == 0, (kind, mode, q.stderr[-1500:])
            assert ("no table collective" in q.stdout) == (mode != "tree"), (mode, q.stdout)    # (auto: a one-shot run pays for RCCL's communicators on the tree route)
            assert out.read_text() == ref_out.read_text(), (kind, mode)
