#!/bin/bash
# usage: prof_cfg.sh <workload name> [pmc]  -> gpurun_out/cfg_<name>_{stats,timeline}.txt (+ cfg_<name>_hbm_traffic.{txt,json} with `pmc`:
# two separate --pmc passes of the EAGER twin of the workload, kernel trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
export AMS_COMMIT=${AMS_COMMIT:-$(cat $R/.ams_commit 2>/dev/null || echo unknown)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profc && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/profc -o run -- python $R/tools/bench_configs.py --only $1 --steps 10 --warmup 3 > $R/gpurun_out/cfg_$1.log 2>&1
DB=$(find /tmp/profc -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB $R/gpurun_out/cfg_$1_stats.txt "tools/bench_configs.py --only $1 --steps 10 --warmup 3 (commit $AMS_COMMIT)"
python $R/tools/step_timeline.py $DB 2 > $R/gpurun_out/cfg_$1_timeline.txt 2>&1
if [ "$2" = "pmc" ]; then
  E=${1%_graph}
  CMD="python $R/tools/bench_configs.py --only $E --steps 2 --warmup 2"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcc_$C
    timeout 900 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmcc_$C -o run -- $CMD > $R/gpurun_out/cfg_${E}_pmc_$C.log 2>&1
  done
  F=$(find /tmp/pmcc_FETCH_SIZE -name "*.db" | head -1); W=$(find /tmp/pmcc_WRITE_SIZE -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $F $W $R/gpurun_out/cfg_${E}_hbm_traffic.txt $R/gpurun_out/cfg_${E}_hbm_traffic.json "tools/bench_configs.py --only $E --steps 2 --warmup 2 (eager)" > /dev/null
fi
