#!/bin/bash
# split-K sweep of the step's products (bf16x6), uncapped and capped
for pad in 0 50000; do
  for s in 0 1 2 3 4 5 6 8 10; do
    if [ $s = 0 ]; then unset AMS_GEMM_SPLITS; else export AMS_GEMM_SPLITS=$s; fi
    python tools/gemm_x6_bench.py --modes 1 --reps 60 --warm 30 --pad $pad --only "dense dX,dense dW,lstm dX,lstm dWx,lstm dU,proj" 2>/dev/null | awk -v s=$s -v p=$pad '{printf "pad=%s splits=%s %s\n", p, s, $0}' | cut -c1-110
  done
done
