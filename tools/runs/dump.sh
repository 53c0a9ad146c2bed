set -o pipefail
O=gpurun_out/r2d; mkdir -p $O
LGA_GROW_DUMP=450 LGA_GROW_TRACE=1 timeout -k 10 120 python bench.py --steps 460 --warmup 5 --stats > $O/dump.json 2> $O/dump.err
grep "grow dump" $O/dump.err > $O/dump.txt; wc -l $O/dump.txt
