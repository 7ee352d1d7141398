#!/bin/bash
# kernel trace of the default bench + per-step timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $R/bench.py --no-cpu-baseline --no-secondary --no-native-f32 --quiet > $R/gpurun_out/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
for k in 10 12 14 16 18; do python $R/tools/step_timeline.py $DB $k > $R/gpurun_out/step_timeline_$k.txt 2>&1; done; cp $R/gpurun_out/step_timeline_12.txt $R/gpurun_out/step_timeline.txt
AMS_PROF_JSON=$R/gpurun_out/replay_kernels.json python $R/tools/prof_summary.py $DB $R/gpurun_out/kernel_stats.txt "python bench.py --no-cpu-baseline --no-secondary (hipGraph replay)"
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-300
