#!/bin/bash
# Collects the round's evidence on an MI355X box: the 500-tree benchmark, a per-kernel trace of 32 trees, three
# hardware-counter passes over 7 trees, the predictor and the learning-to-rank benchmarks.  Summaries land in gpurun_out/final/ (the raw
# rocprofv3 databases stay in /tmp: they exceed what is copied back).
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 200 python $ROOT/bench.py --steps 500 --warmup 5 --stats | tail -1 > "$OUT/bench_500trees.json"
echo "bench done"
timeout -k 10 200 python $ROOT/bench/predict_bench.py | tail -1 > "$OUT/predict_bench.json"
echo "predict bench done"
timeout -k 10 300 python $ROOT/bench/rank_bench.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/rank_bench.txt"
echo "ranking bench done"
rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $ROOT/bench.py --steps 32 --warmup 3 --no-full-run > /tmp/kt.log 2>&1
python $ROOT/tools/rocpd_stats.py "$(find /tmp/kt -name '*.db' | sort | tail -1)" "$OUT/kernel_stats_32trees.csv" > /dev/null
echo "kernel trace done"
pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -s KILL 240 rocprofv3 --pmc "$@" -d /tmp/pmc_$name -o p -- python $ROOT/bench.py --steps 7 --warmup 1 --no-full-run > /tmp/pmc_$name.log 2>&1
  python $ROOT/tools/rocpd_pmc.py "$(find /tmp/pmc_$name -name '*.db' | sort | tail -1)" "$OUT/pmc_$name.csv" > /dev/null
  echo "pmc $name done"
}
pass sq_lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES
pass fetch_occupancy FETCH_SIZE OccupancyPercent SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY
pass write_atomics WRITE_SIZE TCC_ATOMIC_sum TCC_HIT_sum SQ_INSTS_VMEM_WR
echo "all done"
