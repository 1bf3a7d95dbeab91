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


def test_cli_fails_loudly_without_gpu(d1_files, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r, e = d1_files
    p = run("-r", r, "-e", e, "-o", str(tmp_path / "o.nwk"))
    assert p.returncode == 1 and "no CPU fallback" in p.stderr
    assert not (tmp_path / "o.nwk").exists()


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
    # unknown taxon -> error exit, like the reference's uncaught std::out_of_range
    bad = tmp_path / "bad.nwk"
    bad.write_text(golden["D6"]["bad_tree"] + "\n")
    p = run("-r", r, "-e", str(bad), "-o", str(tmp_path / "o2.nwk"))
    assert p.returncode == 1 and "unknown taxon" in p.stderr
