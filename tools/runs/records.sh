#!/bin/bash
# The records named by the benchmark configurations, one GPU: bagging vs boosting on 1 M x 50, the Epsilon-shaped wide
# data (serial, with a per-kernel trace), and the collection of tools/collect_profiles.sh.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-rec}
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 python -u $ROOT/examples/rf_vs_gbdt_1m.py > "$OUT/rf_vs_gbdt_1m.log" 2>&1 || { tail -20 "$OUT/rf_vs_gbdt_1m.log"; exit 1; }
grep -v amdgpu.ids "$OUT/rf_vs_gbdt_1m.log" | tail -6
timeout -k 10 400 python -u $ROOT/examples/epsilon_feature_parallel.py > "$OUT/epsilon_serial.log" 2>&1 || { tail -20 "$OUT/epsilon_serial.log"; exit 1; }
grep -v amdgpu.ids "$OUT/epsilon_serial.log" | tail -6
rm -rf /tmp/ktE && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/ktE -o kt -- python $ROOT/examples/epsilon_feature_parallel.py --trees 8 > /tmp/ktE.log 2>&1 || { tail -20 /tmp/ktE.log; exit 1; }
python $ROOT/tools/rocpd_stats.py "$(find /tmp/ktE -name '*.db' | sort | tail -1)" "$OUT/kernel_stats_epsilon_8trees.csv" > /dev/null
echo "epsilon kernel trace done"
head -12 "$OUT/kernel_stats_epsilon_8trees.csv"
