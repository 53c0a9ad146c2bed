set -o pipefail
O=gpurun_out/r2a; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_tests.log 2>&1 && echo tests ok &&
LGA_GROW_TRACE=1 timeout -k 10 200 python bench.py --steps 500 --warmup 5 --stats > $O/bench500.json 2> $O/bench500.err && echo bench ok &&
timeout -k 10 200 python bench.py --steps 100 --warmup 5 --stats --rows 1312500 > $O/bench_shard.json 2> $O/bench_shard.err && echo shard ok &&
(cd /tmp && timeout -k 10 600 python $GRAFT_REPO_ROOT/examples/gridsearch_diamonds.py --full --checkpoint /tmp/pg.jsonl > $GRAFT_REPO_ROOT/$O/grid_full.log 2>&1) && echo grid ok &&
timeout -k 10 300 python examples/rf_vs_gbdt_1m.py > $O/rf_vs_gbdt.log 2>&1 && echo rf ok &&
timeout -k 10 300 python examples/epsilon_feature_parallel.py > $O/epsilon.log 2>&1 && echo eps ok
