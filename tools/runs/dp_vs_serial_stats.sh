#!/bin/bash
# the grower's counters over a complete 500-tree run: serial learner against the data-parallel path with one RCCL rank
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/dp_vs_serial
mkdir -p "$OUT"
cd /tmp
LGA_ROUNDS_IN_FLIGHT=1 timeout -k 10 200 python $ROOT/bench.py --steps 500 --warmup 5 --stats > $OUT/serial.json 2>> $OUT/err.txt
LGA_BENCH_FORCE_DIST=1 timeout -k 10 200 python $ROOT/bench.py --steps 500 --warmup 5 --stats > $OUT/dp1.json 2>> $OUT/err.txt
for w in 130 300; do
LGA_ROUNDS_IN_FLIGHT=1 timeout -k 10 200 python $ROOT/bench.py --steps 100 --warmup $w --no-full-run > $OUT/serial_w$w.json 2>> $OUT/err.txt
LGA_BENCH_FORCE_DIST=1 timeout -k 10 200 python $ROOT/bench.py --steps 100 --warmup $w --no-full-run > $OUT/dp1_w$w.json 2>> $OUT/err.txt
done
python - <<PY
import json
for f in ["serial","dp1","serial_w130","dp1_w130","serial_w300","dp1_w300"]:
    d=json.load(open("$OUT/%s.json"%f))
    print(f, d["value"], d["ms_per_step"], d["config"].get("profile"), d["config"].get("last_tree"))
PY
