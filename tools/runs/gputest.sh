set -o pipefail
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu.py -m gpu -x -q --timeout 100 --timeout-method thread -k "kernel" > $O/kernels.log 2>&1; rc=$?; tail -5 $O/kernels.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_tests.log 2>&1; rc=$?; tail -15 $O/gpu_tests.log; [ $rc -eq 0 ] || exit $rc
