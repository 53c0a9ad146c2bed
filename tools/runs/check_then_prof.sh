set -o pipefail
O=gpurun_out/r2h; mkdir -p $O
LGA_DEBUG_LIB=1 timeout -k 10 200 python bench.py --steps 120 --warmup 5 --stats > $O/debug_bench.json 2> $O/debug_bench.err || { tail -5 $O/debug_bench.err; exit 1; }
echo "debug build ok"; cat $O/debug_bench.json | cut -c1-200
LGA_DEBUG_LIB=1 timeout -k 10 200 python bench.py --steps 300 --warmup 5 --stats --rows 1312500 --extra '{"num_leaves":255}' > $O/debug_bench2.json 2> $O/debug_bench2.err || { tail -5 $O/debug_bench2.err; exit 1; }
echo "debug build shard ok"
LGA_GROW_TRACE=1 timeout -k 10 200 python bench.py --steps 500 --warmup 5 --stats > $O/bench500.json 2> $O/bench500.err || { tail -5 $O/bench500.err; exit 1; }
cat $O/bench500.json | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 3 > /tmp/kt.log 2>&1 || { tail -5 /tmp/kt.log; exit 1; }
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$(find /tmp/kt -name '*.db' | sort | tail -1)" "$GRAFT_REPO_ROOT/$O/kernel_stats_32trees.csv" > /dev/null
head -12 $GRAFT_REPO_ROOT/$O/kernel_stats_32trees.csv
