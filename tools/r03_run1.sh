#!/bin/bash
# round-3 GPU session 1: the VALU yardstick, then the count kernel A/B (product vs experiment builds) in one process,
# then two counter passes (wave-time breakdown) of the product and of exp1
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3a; mkdir -p "$OUT"; export TMPDIR=/tmp
rocminfo | grep -m1 -i "gfx950" > "$OUT/box.txt" 2>&1; lscpu | grep -i "model name" >> "$OUT/box.txt"
timeout -k 10 300 tools/bin/valu_yardstick 250 > "$OUT/valu_yardstick.txt" 2>&1 || echo "yardstick rc $?" >> "$OUT/errors.txt"
LIBS="tools/bin/libqs_exp0.so tools/bin/libqs_exp1.so tools/bin/libqs_exp3.so tools/bin/libqs_exp17.so tools/bin/libqs_exp0.so tools/bin/libqs_exp1.so"
timeout -k 10 300 tools/bin/count_bench 512 10000 32 5 $LIBS > "$OUT/cb_512.txt" 2>&1 || echo "cb512 rc $?" >> "$OUT/errors.txt"
timeout -k 10 120 tools/bin/count_bench 256 12500 32 5 $LIBS > "$OUT/cb_256.txt" 2>&1 || echo "cb256 rc $?" >> "$OUT/errors.txt"
timeout -k 10 120 tools/bin/count_bench 128 1000 32 20 $LIBS > "$OUT/cb_128.txt" 2>&1 || echo "cb128 rc $?" >> "$OUT/errors.txt"
CB_NNI=1 timeout -k 10 300 tools/bin/count_bench 512 10000 32 3 tools/bin/libqs_exp0.so tools/bin/libqs_exp1.so > "$OUT/cb_512_nni.txt" 2>&1 || echo "cb512nni rc $?" >> "$OUT/errors.txt"
cd /tmp
for lib in exp0 exp1; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_$lib" -o run -- "$ROOT/tools/bin/count_bench" 512 10000 32 2 "$ROOT/tools/bin/libqs_$lib.so" > "$OUT/pmc_$lib.log" 2>&1 || echo "pmc $lib rc $?" >> "$OUT/errors.txt"
done
cd "$ROOT"
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for lib in ("exp0", "exp1"):
    for f in glob.glob(f"{out}/pmc_{lib}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-60:]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(f"{out}/pmc_{lib}_summary.txt", "w") as o:
            for k, d in acc.items():
                if "count_bitslice3" not in k: continue
                o.write(k + "\n")
                for c, v in sorted(d.items()):
                    o.write(f"  {c}: n={len(v)} avg={sum(v)/len(v):.6g}\n")
PY
find "$OUT" -name "*counter_collection.csv" -size +1M -delete
cat "$OUT/valu_yardstick.txt" "$OUT"/cb_*.txt
