#!/bin/bash
# usage: prof_cfg.sh <workload name>  -> gpurun_out/cfg_<name>_stats.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profc && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/profc -o run -- python $R/tools/bench_configs.py --only $1 --steps 10 --warmup 3 > $R/gpurun_out/cfg_$1.log 2>&1
DB=$(find /tmp/profc -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB $R/gpurun_out/cfg_$1_stats.txt "tools/bench_configs.py --only $1 --steps 10 --warmup 3"
