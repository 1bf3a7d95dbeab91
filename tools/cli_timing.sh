#!/bin/bash
# End-to-end timing of the CLIs on the GPU box: tools/cli_timing.sh [taxa] [trees]
set -u
N=${1:-256}; M=${2:-20000}
D=$(mktemp -d)
python3 - "$N" "$M" "$D" <<'PY'
import subprocess, sys, time
sys.path.insert(0, ".")
from quartetscores_amd import synth
n, m, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
open(d + "/ref.nwk", "w").write(synth.reference_tree(n, 4000) + "\n")
base = synth.tree_set(n, min(m, 2000), 4001)
with open(d + "/eval.nwk", "w") as f:
    for i in range(m):
        f.write(base[i % len(base)] + "\n")
import os
print("eval file bytes:", os.path.getsize(d + "/eval.nwk"))

def timed(label, cmd, out):
    if os.path.exists(out):
        os.remove(out)
    # (a process that starts right after another one has released a 17-34 GB table can wait 0.3-0.8 s inside its own hipMalloc
    #  while the driver reclaims that memory -- seen as counting phases of 0.7-1.2 s on some boxes of the pool; not the product's time)
    time.sleep(3.0)
    t0 = time.perf_counter()
    p = subprocess.run(cmd, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    lines = [l for l in p.stdout.split("\n") if "took" in l or "Elapsed" in l]
    print(f"{label}: {dt:.3f} s wall, rc {p.returncode} | " + " ".join(lines))
    if "--trace" in cmd:
        print("".join("      " + l + "\n" for l in p.stderr.split("\n") if "rccl" in l or "main:" in l), end="")

for t in ("1", "8", "64", "0"):
    timed(f"QuartetScores -t {t}", ["quartetscores_amd/bin/QuartetScores", "-r", d + "/ref.nwk", "-e", d + "/eval.nwk", "-o", d + "/out.nwk", "-t", t], d + "/out.nwk")
base_cmd = ["quartetscores_amd/bin/QuartetScores", "-r", d + "/ref.nwk", "-e", d + "/eval.nwk", "-o", d + "/out3.nwk", "-t", "8", "--gpus", "1"]
for label, extra in (("--gpus 1, RCCL, counting beside ncclCommInitAll (--comm-overlap 1: A/B only)", ["--comm-overlap", "1"]),
                     ("--gpus 1, RCCL, first launch waits for the communicators (--comm-overlap 0: the default)", ["--comm-overlap", "0"]),
                     ("--gpus 1 --reduce p2p (peer access, no communicator)", ["--reduce", "p2p"]),
                     ("--gpus 3 --reduce p2p --gpus-on-one-device (3 contexts, reduce-scatter by qs_sum_words)", ["--gpus", "3", "--reduce", "p2p", "--gpus-on-one-device"])):
    for rep in range(2):
        timed("QuartetScores " + label, base_cmd + extra + (["--trace"] if rep == 1 else []), d + "/out3.nwk")
    print("   output identical to the single-GPU CLI:", open(d + "/out.nwk").read() == open(d + "/out3.nwk").read())
timed("dist_cli, 1 process", [sys.executable, "-m", "quartetscores_amd.dist_cli", "-r", d + "/ref.nwk", "-e", d + "/eval.nwk", "-o", d + "/out2.nwk"], d + "/out2.nwk")
print("outputs identical:", open(d + "/out.nwk").read() == open(d + "/out2.nwk").read())
PY
rm -rf $D
