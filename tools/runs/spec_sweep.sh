set -o pipefail
O=gpurun_out/r2b; mkdir -p $O
run() { # name rows env...
  local name=$1 rows=$2; shift 2
  env "$@" LGA_GROW_TRACE=1 timeout -k 10 120 python bench.py --steps 100 --warmup 10 --stats --rows $rows > $O/$name.json 2> $O/$name.err || return 1
  python - $O/$name.json $O/$name.err $name <<'PY'
import json,sys,re
d=json.load(open(sys.argv[1])); p=d["config"]["profile"]
t=[l for l in open(sys.argv[2]) if "grow trace" in l]
m=re.search(r"rows partitioned (\d+), of used splits (\d+)", t[-1]) if t else None
print(sys.argv[3], "ms/tree %.3f rounds/tree %.1f tasks/tree %.0f part_rows/tree %.1fM used %.1fM" % (d["ms_per_step"], p["rounds"]/p["trees"], p["split_tasks"]/p["trees"], p["partition_rows"]/p["trees"]/1e6, (int(m.group(2))/1e6 if m else -1)), flush=True)
PY
}
for rows in 10500000 1312500; do
run base_$rows $rows LGA_X=1 &&
run f100_$rows $rows LGA_SPEC_FLOOR=1.0 &&
run f125_$rows $rows LGA_SPEC_FLOOR=1.25 &&
run lo50_hi100_$rows $rows LGA_SPEC_LO=0.5 LGA_SPEC_HI=1.0 &&
run lo50_hi125_$rows $rows LGA_SPEC_LO=0.5 LGA_SPEC_HI=1.25 &&
run lo25_hi150_$rows $rows LGA_SPEC_LO=0.25 LGA_SPEC_HI=1.5 &&
run lo50_hi200_$rows $rows LGA_SPEC_LO=0.5 LGA_SPEC_HI=2.0 &&
run lo50_hi125_r16k_256k_$rows $rows LGA_SPEC_LO=0.5 LGA_SPEC_HI=1.25 LGA_SPEC_ROWS_LO=16384 LGA_SPEC_ROWS_HI=262144 || exit 1
done
