for t in 60 120 250; do
  LGA_GROW_TRACE=1 LGA_GROW_DUMP=$t timeout -k 10 120 python bench.py --steps $((t+2)) --warmup 0 --no-full-run --no-other-precision 2> gpurun_out/dump_$t.txt > /dev/null || exit 1
  grep "grow dump\] round" gpurun_out/dump_$t.txt > gpurun_out/dump_rounds_$t.txt
done
