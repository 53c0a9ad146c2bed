set -o pipefail
O=gpurun_out/r2j; mkdir -p $O
run() { # name rows env...
  local name=$1 rows=$2; shift 2
  env "$@" timeout -k 10 120 python bench.py --steps 200 --warmup 5 --stats --rows $rows > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); p=d["config"]["profile"]
print(sys.argv[2], "ms/tree %.3f rounds/tree %.1f tasks/tree %.0f part_rows/tree %.1fM" % (d["ms_per_step"], p["rounds"]/p["trees"], p["split_tasks"]/p["trees"], p["partition_rows"]/p["trees"]/1e6), flush=True)
PY
}
for rows in 10500000 1312500; do
run fused_$rows $rows LGA_X=1 &&
run split_$rows $rows LGA_FUSED_TAIL=0 &&
run split_flight1_$rows $rows LGA_FUSED_TAIL=0 LGA_ROUNDS_IN_FLIGHT=1 &&
run fused_flight1_$rows $rows LGA_ROUNDS_IN_FLIGHT=1 || exit 1
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && LGA_FUSED_TAIL=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 3 > /tmp/kt.log 2>&1 || { tail -5 /tmp/kt.log; exit 1; }
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$(find /tmp/kt -name '*.db' | sort | tail -1)" "$GRAFT_REPO_ROOT/$O/kernel_stats_split.csv" > /dev/null
head -8 $GRAFT_REPO_ROOT/$O/kernel_stats_split.csv
