set -o pipefail
O=gpurun_out/r2o; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 150 --timeout-method thread -k "parallel or rccl or multi_process or fallback" > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || exit $rc
LGA_BENCH_FORCE_DIST=1 timeout -k 10 200 python bench.py --steps 200 --warmup 5 --stats > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err || { tail -5 $O/bench_rccl_1rank.err; exit 1; }
cut -c1-300 $O/bench_rccl_1rank.json
timeout -k 10 200 python bench.py --steps 200 --warmup 5 --stats > $O/bench_serial.json 2> $O/bench_serial.err || exit 1
cut -c1-200 $O/bench_serial.json
timeout -k 10 300 python examples/epsilon_feature_parallel.py > $O/epsilon.log 2>&1 || { tail -5 $O/epsilon.log; exit 1; }
tail -2 $O/epsilon.log
timeout -k 10 300 python examples/rf_vs_gbdt_1m.py > $O/rf_vs_gbdt.log 2>&1 || { tail -5 $O/rf_vs_gbdt.log; exit 1; }
tail -3 $O/rf_vs_gbdt.log
