#!/bin/bash
# fp16x3 products of the step alone under each tile configuration (AMS_GEMM_X6CFG 0 = 128x128, 3 = 128x256, unset = the rule),
# uncapped and with the side-stream pad; optional forced split-K counts ("SPLITS=2,4 tools/probes/cfg_sweep.sh").  -> gpurun_out/cfg_sweep.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/cfg_sweep.txt
: > $O
for pad in 0 50000; do
  only=""; [ $pad != 0 ] && only="--only dW,dU"
  for cfg in rule 0 3; do
    for s in 0 $(echo ${SPLITS:-} | tr ',' ' '); do
      if [ $cfg = rule ] && [ $s != 0 ]; then continue; fi
      e1=""; [ $cfg != rule ] && e1="AMS_GEMM_X6CFG=$cfg"
      e2=""; [ $s != 0 ] && e2="AMS_GEMM_SPLITS=$s"
      env $e1 $e2 python $R/tools/gemm_x6_bench.py --modes 2 --pad $pad --reps 100 --warm 50 $only 2>&1 | grep fp16x3 | sed "s/^/pad=$pad cfg=$cfg splits=$s /" >> $O
    done
  done
done
cut -c1-150 $O
