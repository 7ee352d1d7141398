#!/bin/bash
# kernel trace of the default bench + per-step timeline.  TAG=<suffix> names the outputs (gpurun_out/kernel_stats<TAG>.txt, ...);
# environment variables of the caller (AMS_*) reach the bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
export AMS_COMMIT=${AMS_COMMIT:-$(cat $R/.ams_commit 2>/dev/null || echo unknown)}
T=${TAG:-}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $R/bench.py --no-cpu-baseline --no-secondary --no-native-f32 --quiet > $R/gpurun_out/prof_bench$T.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
for k in 20 21; do python $R/tools/step_timeline.py $DB $k > $R/gpurun_out/step_timeline${T}_$k.txt 2>&1; done
AMS_PROF_JSON=$R/gpurun_out/replay_kernels$T.json python $R/tools/prof_summary.py $DB $R/gpurun_out/kernel_stats$T.txt "python bench.py --no-cpu-baseline --no-secondary (hipGraph replay) $T"
tail -1 $R/gpurun_out/prof_bench$T.log | cut -c1-300
