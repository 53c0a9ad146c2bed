set -o pipefail
O=gpurun_out/r2p; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $O/gpu_tests.log 2>&1; rc=$?; tail -4 $O/gpu_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_default_driver_args.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cat $O/bench_default_driver_args.json
timeout -k 10 900 python -u bench/auc_parity.py --rows 1000000 --trees 200 > $O/auc_parity.txt 2> $O/auc_parity.err || { tail -5 $O/auc_parity.err; exit 1; }
cat $O/auc_parity.txt | head -8
