"""CLI --trace several times in a row; prints the full trace of the slowest and the fastest counting phase (which step of the set-up
waits when a process starts right after another one has released its table?).   python tools/trace_probe.py [runs] [pause_s]"""
import os, re, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quartetscores_amd import native_ingest
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pause = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
d = "/tmp/qs_trace"; os.makedirs(d, exist_ok=True)
open(d + "/r.nwk", "wb").write(native_ingest.synth_trees(512, 1, 2000))
open(d + "/e.nwk", "wb").write(native_ingest.synth_trees(512, 10000, 2001))
res = []
for _ in range(runs):
    o = d + "/o.nwk"
    if os.path.exists(o): os.remove(o)
    time.sleep(pause)
    p = subprocess.run(["quartetscores_amd/bin/QuartetScores", "-r", d + "/r.nwk", "-e", d + "/e.nwk", "-o", o, "-t", "8", "--trace"], capture_output=True, text=True)
    took = [int(x) for x in re.findall(r"It took: (\d+) microseconds", p.stdout)]
    res.append((took[0], p.stderr))
print("counting phases (ms):", [round(t / 1e3, 1) for t, _ in res])
res.sort()
print("== fastest\n" + res[0][1]); print("== slowest\n" + res[-1][1])
