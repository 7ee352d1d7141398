#!/bin/bash
# Matrix-pipe utilisation of the bench workload from ONE rocprofv3 PMC pass (kernel trace only, no other trace domain):
#   MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs), MFMA ops by input type.
# usage (on the GPU box): bash tools/pmc_mfma.sh   -> gpurun_out/mfma_util.{txt,json}
R=${GRAFT_REPO_ROOT:-/root/repo}
export AMS_COMMIT=${AMS_COMMIT:-$(cat $R/.ams_commit 2>/dev/null || echo unknown)}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --no-secondary --no-native-f32 --graph 0 --steps 3 --warmup 2 --roofline-steps 1 --quiet"
rm -rf /tmp/pmc_mfma
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_mfma -o run -- $CMD > $R/gpurun_out/pmc_mfma.log 2>&1
DB=$(find /tmp/pmc_mfma -name "*.db" | head -1)
python $R/tools/pmc_mfma_summary.py $DB $R/gpurun_out/mfma_util.txt $R/gpurun_out/mfma_util.json
