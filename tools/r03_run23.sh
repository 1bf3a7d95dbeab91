#!/bin/bash
# round-3 GPU session 23: qs_score_prepare behind the enqueued counts (CLI scoring phase), RCCL init time with / without MSCCL
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3z; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_cli.py -m gpu -x -q -k "prepare or single_read or automatic_scoring or cli" > "$OUT/pytest.log" 2>&1; echo "pytest rc $?" | tee "$OUT/summary.txt"
tail -4 "$OUT/pytest.log"
python3 - "$OUT" <<'PY'
import subprocess, sys, os, re
sys.path.insert(0, ".")
from quartetscores_amd import native_ingest
out = sys.argv[1]
d = "/tmp/qs_trace"; os.makedirs(d, exist_ok=True)
open(d + "/r.nwk", "wb").write(native_ingest.synth_trees(512, 1, 2000))
open(d + "/e.nwk", "wb").write(native_ingest.synth_trees(512, 10000, 2001))
log = []
for extra, env in (([], {}), ([], {}), (["--gpus", "1"], {}), (["--gpus", "1"], {"RCCL_MSCCL_ENABLE": "0", "RCCL_MSCCLPP_ENABLE": "0"}),
                   (["--gpus", "1"], {"RCCL_MSCCL_ENABLE": "0", "RCCL_MSCCLPP_ENABLE": "0", "NCCL_IB_DISABLE": "1", "NCCL_NET_GDR_LEVEL": "0"})):
    o = d + "/o.nwk"
    if os.path.exists(o): os.remove(o)
    e = dict(os.environ); e.update(env)
    p = subprocess.run(["quartetscores_amd/bin/QuartetScores", "-r", d + "/r.nwk", "-e", d + "/e.nwk", "-o", o, "-t", "8", "--trace"] + extra, capture_output=True, text=True, env=e)
    took = [int(x) for x in re.findall(r"It took: (\d+) microseconds", p.stdout)]
    log.append(f"== {' '.join(extra) or 'one GPU'} {env}: rc {p.returncode} counting {took[0] / 1e3:.1f} ms, scoring {took[1] / 1e3:.1f} ms\n" + p.stderr)
open(out + "/cli_trace.txt", "w").write("\n".join(log))
print("\n".join(log)[:9000])
PY
