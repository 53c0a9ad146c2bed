#!/bin/bash
# the multi-rank path of bench.py on a one-GPU box: (1) one rank through the RCCL communicator, (2) two and four
# ranks sharing the GPU, reduced through host memory under gloo (timings of (2) mean nothing; the point is that the
# driver's command line runs end to end), (3) the per-tree cost of a 1/8 shard on one GPU
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/dp_rehearsal
mkdir -p "$OUT"
cd $ROOT
LGA_BENCH_FORCE_DIST=1 timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-full-run --stats > $OUT/rccl_1rank.json 2> $OUT/rccl_1rank.err || { tail -5 $OUT/rccl_1rank.err; exit 1; }
cut -c1-200 $OUT/rccl_1rank.json
for P in 2 4; do
  LGA_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port 2954$P \
    bench.py --gpus $P --steps 10 --warmup 2 --rows 2000000 --no-full-run > $OUT/gloo_${P}ranks.json 2> $OUT/gloo_${P}ranks.err || { tail -5 $OUT/gloo_${P}ranks.err; exit 1; }
  cut -c1-200 $OUT/gloo_${P}ranks.json
done
timeout -k 10 200 python bench.py --steps 100 --warmup 30 --rows 1312500 --no-full-run > $OUT/shard_1of8.json 2> $OUT/shard_1of8.err || exit 1
cut -c1-200 $OUT/shard_1of8.json
