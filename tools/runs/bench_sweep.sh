set -o pipefail
O=gpurun_out/r2g; mkdir -p $O
run() { # name rows env...
  local name=$1 rows=$2; shift 2
  env "$@" LGA_GROW_TRACE=1 timeout -k 10 120 python bench.py --steps 500 --warmup 5 --stats --rows $rows > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); p=d["config"]["profile"]
print(sys.argv[2], "ms/tree %.3f rounds/tree %.1f tasks/tree %.0f part_rows/tree %.1fM" % (d["ms_per_step"], p["rounds"]/p["trees"], p["split_tasks"]/p["trees"], p["partition_rows"]/p["trees"]/1e6), flush=True)
PY
}
for rows in 10500000 1312500; do
run base_$rows $rows LGA_X=1 &&
run solo4k_$rows $rows LGA_SOLO_ROWS=4096 &&
run solo16k_$rows $rows LGA_SOLO_ROWS=16384 &&
run units320_$rows $rows LGA_TARGET_UNITS=320 &&
run units1280_$rows $rows LGA_TARGET_UNITS=1280 &&
run flight1_$rows $rows LGA_ROUNDS_IN_FLIGHT=1 &&
run spec075_$rows $rows LGA_SPEC_FLOOR=0.75 &&
run spec_lo25_$rows $rows LGA_SPEC_LO=0.25 LGA_SPEC_HI=1.0 &&
run spec_64k_1m_$rows $rows LGA_SPEC_ROWS_LO=65536 LGA_SPEC_ROWS_HI=1048576 || exit 1
done
