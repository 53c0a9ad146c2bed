set -o pipefail
O=gpurun_out/r2m; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 100 --timeout-method thread -k "kernel or parity or same or identical" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit $rc
run() { # name rows env...
  local name=$1 rows=$2; shift 2
  env "$@" timeout -k 10 120 python bench.py --steps 500 --warmup 5 --stats --rows $rows > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); p=d["config"]["profile"]
print(sys.argv[2], "ms/tree %.3f rounds/tree %.1f tasks/tree %.0f part_rows/tree %.1fM" % (d["ms_per_step"], p["rounds"]/p["trees"], p["split_tasks"]/p["trees"], p["partition_rows"]/p["trees"]/1e6), flush=True)
PY
}
for rows in 10500000 1312500; do
run base_$rows $rows LGA_X=1 &&
run solo2k_$rows $rows LGA_SOLO_ROWS=2048 &&
run solo4k_$rows $rows LGA_SOLO_ROWS=4096 &&
run solo16k_$rows $rows LGA_SOLO_ROWS=16384 &&
run units320_$rows $rows LGA_TARGET_UNITS=320 &&
run units1280_$rows $rows LGA_TARGET_UNITS=1280 &&
run flight1_$rows $rows LGA_ROUNDS_IN_FLIGHT=1 || exit 1
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && LGA_ROUNDS_IN_FLIGHT=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 3 > /tmp/kt.log 2>&1 || { tail -5 /tmp/kt.log; exit 1; }
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py "$(find /tmp/kt -name '*.db' | sort | tail -1)" "$GRAFT_REPO_ROOT/$O/dispatches.csv" 0.9 300 > /dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$(find /tmp/kt -name '*.db' | sort | tail -1)" "$GRAFT_REPO_ROOT/$O/kernel_stats.csv" | head -8
