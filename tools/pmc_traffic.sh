#!/bin/bash
# HBM-side traffic of the bench workload: two separate rocprofv3 PMC passes (never combined with other trace domains).
# usage (on the GPU box): bash tools/pmc_traffic.sh   -> gpurun_out/hbm_traffic.{txt,json}
R=${GRAFT_REPO_ROOT:-/root/repo}
export AMS_COMMIT=${AMS_COMMIT:-$(cat $R/.ams_commit 2>/dev/null || echo unknown)}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --no-secondary --no-native-f32 --graph 0 --steps 3 --warmup 2 --roofline-steps 1 --quiet"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o run -- $CMD > $R/gpurun_out/pmc_$C.log 2>&1
done
F=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
python $R/tools/pmc_summary.py $F $W $R/gpurun_out/hbm_traffic.txt $R/gpurun_out/hbm_traffic.json "bench.py --no-cpu-baseline --no-secondary --no-native-f32 --graph 0 --steps 3 --warmup 2 --roofline-steps 1"
