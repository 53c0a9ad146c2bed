set -o pipefail
O=gpurun_out/r2k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && LGA_FUSED_TAIL=0 LGA_ROUNDS_IN_FLIGHT=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 3 > /tmp/kt.log 2>&1 || { tail -5 /tmp/kt.log; exit 1; }
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py "$(find /tmp/kt -name '*.db' | sort | tail -1)" "$GRAFT_REPO_ROOT/$O/dispatches.csv" 0.9 300
