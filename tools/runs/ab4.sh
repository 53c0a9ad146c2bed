set -o pipefail
O=gpurun_out/r2n; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 100 --timeout-method thread > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit $rc
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
run solo8k_$rows $rows LGA_SOLO_ROWS=8192 &&
run spec075_$rows $rows LGA_SPEC_FLOOR=0.75 &&
run spec_lo25_$rows $rows LGA_SPEC_LO=0.25 &&
run spec_lo25_hi90_$rows $rows LGA_SPEC_LO=0.25 LGA_SPEC_HI=0.9 &&
run spec_256k_4m_$rows $rows LGA_SPEC_ROWS_LO=262144 LGA_SPEC_ROWS_HI=4194304 &&
run flight1_$rows $rows LGA_ROUNDS_IN_FLIGHT=1 || exit 1
done
