# PMC anatomy of the k-means passes in the inference config: clock, VALU / SALU / LDS instruction counts and busy cycles (two passes)
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_configs.py --only front_DPCL_inference --steps 2"
i=0
for C in "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/kpmc$i
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/kpmc$i -o run -- $CMD > /tmp/kpmc$i.log 2>&1 || tail -5 /tmp/kpmc$i.log
  DB=$(find /tmp/kpmc$i -name "*.db" | head -1)
  python - "$DB" <<'PY' | tee -a $R/gpurun_out/km_pmc.txt
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
if 'counters_collection' not in tabs:
    print('tables', tabs); sys.exit()
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection where kernel_name like '%kmeans_hard%' or kernel_name like '%kmeans_pass%'"):
    agg[row[0][22:62]][row[1]].append((row[3], row[2]))
for k, d in agg.items():
    out = []
    for n, v in d.items():
        per = collections.defaultdict(float)
        for disp, val in v: per[disp] += val
        vals = sorted(per.values())
        out.append('%s=%.4g' % (n, vals[len(vals) // 2]))
    print(k, ' '.join(out))
PY
done
