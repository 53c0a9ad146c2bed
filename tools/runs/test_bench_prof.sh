set -o pipefail
O=gpurun_out/${1:-r2i}; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_tests.log 2>&1; rc=$?; tail -4 $O/gpu_tests.log; [ $rc -eq 0 ] || exit $rc
LGA_GROW_TRACE=1 timeout -k 10 200 python bench.py --steps 500 --warmup 5 --stats > $O/bench500.json 2> $O/bench500.err || { tail -5 $O/bench500.err; exit 1; }
cat $O/bench500.json | cut -c1-250
timeout -k 10 200 python bench.py --steps 500 --warmup 5 --stats --rows 1312500 > $O/bench_shard.json 2> $O/bench_shard.err || { tail -5 $O/bench_shard.err; exit 1; }
cat $O/bench_shard.json | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 3 --no-full-run > /tmp/kt.log 2>&1 || { tail -5 /tmp/kt.log; exit 1; }
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$(find /tmp/kt -name '*.db' | sort | tail -1)" "$GRAFT_REPO_ROOT/$O/kernel_stats_32trees.csv" > /dev/null
head -8 $GRAFT_REPO_ROOT/$O/kernel_stats_32trees.csv
