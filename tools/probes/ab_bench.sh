#!/bin/bash
# A/B of the default bench under environment variants, interleaved (boxes and processes differ by a few %):
#   tools/probes/ab_bench.sh 2 "AMS_RING_ARENA=0" "" "AMS_GEMM_X6WASTE=1.3"      -> gpurun_out/ab_bench.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_bench.txt
rounds=$1; shift
: > $O
for r in $(seq $rounds); do
  for v in "$@"; do
    ms=$(env $v python $R/bench.py --no-cpu-baseline --no-secondary --no-native-f32 --roofline-steps 0 --quiet 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "round=$r [$v] ms_per_step,value = $ms" >> $O
  done
done
cat $O
