set -o pipefail
O=gpurun_out/r2c; mkdir -p $O
run() { # name rows env...
  local name=$1 rows=$2; shift 2
  env "$@" LGA_GROW_TRACE=1 timeout -k 10 120 python bench.py --steps 500 --warmup 5 --stats --rows $rows > $O/$name.json 2> $O/$name.err || return 1
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); p=d["config"]["profile"]
print(sys.argv[2], "ms/tree %.3f rounds/tree %.1f tasks/tree %.0f part_rows/tree %.1fM" % (d["ms_per_step"], p["rounds"]/p["trees"], p["split_tasks"]/p["trees"], p["partition_rows"]/p["trees"]/1e6), flush=True)
PY
  grep "grow trace" $O/$name.err | sed -e 's/.*rows partitioned/   rows partitioned/'
}
for rows in 10500000 1312500; do
run base_$rows $rows LGA_SPEC_RATIO=0 &&
run ratio_$rows $rows LGA_X=1 &&
run ratio_f100_$rows $rows LGA_SPEC_FLOOR=1.0 &&
run ratio_lo50_hi100_$rows $rows LGA_SPEC_LO=0.5 LGA_SPEC_HI=1.0 &&
run ratio_lo50_hi90_$rows $rows LGA_SPEC_LO=0.5 LGA_SPEC_HI=0.9 &&
run ratio_lo75_hi110_$rows $rows LGA_SPEC_LO=0.75 LGA_SPEC_HI=1.1  || exit 1
done
