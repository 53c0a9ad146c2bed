#!/bin/bash
# The records of a round on one MI355X, written under gpurun_out/r3/final/ (copy what is to be kept into profiles/):
#   tools/runs/records.sh            everything (about 4 minutes)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3/final
mkdir -p "$OUT"
cd "$ROOT"
echo "== 500-tree benchmark, both precisions"
timeout -k 10 300 python bench.py --steps 500 --warmup 5 > "$OUT/bench_1gpu_500trees.json"
cut -c1-220 "$OUT/bench_1gpu_500trees.json"
echo "== kernel statistics (32 trees after 100), full precision / 16-bit"
bash tools/runs/kstats.sh final_full --no-other-precision > /dev/null
bash tools/runs/kstats.sh final_q16 --precision q16 --no-other-precision > /dev/null
bash tools/runs/kstats.sh final_shard_full --rows 1312500 --no-other-precision > /dev/null
for t in final_full final_q16 final_shard_full; do
  mkdir -p "$OUT/kstats_$t" && cp gpurun_out/r3/kstats_$t/* "$OUT/kstats_$t/"
  echo "-- $t: $(cut -c1-200 gpurun_out/r3/kstats_$t/bench.json | grep -o '"ms_per_step": [0-9.]*')"; head -8 gpurun_out/r3/kstats_$t/kernel_stats.csv
done
echo "== LDS counters of the growth kernels"
bash tools/runs/pmc_lds.sh final_full --no-other-precision > /dev/null
cp gpurun_out/r3/pmc_final_full/pmc_sq_lds.csv "$OUT/pmc_sq_lds_full.csv"
grep -E "grow_kernel.*(LDS_BANK_CONFLICT|ACTIVE_INST_LDS|INSTS_LDS)" "$OUT/pmc_sq_lds_full.csv" | head -8
echo "== upstream's GPU-benchmark shapes"
timeout -k 10 400 python tools/runs/wide_shapes.py --precision full > "$OUT/wide_shapes_full.txt"
timeout -k 10 400 python tools/runs/wide_shapes.py --precision q16 > "$OUT/wide_shapes_q16.txt"
cat "$OUT/wide_shapes_full.txt" "$OUT/wide_shapes_q16.txt" | cut -c1-110
