#!/bin/bash
# memory-side read traffic and time of one product under tile-order band heights: tools/probes/gm_sweep.sh M,N,K,tA,tB "1 2 4 8 16"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for gm in $2; do
  rm -rf /tmp/gms
  AMS_GEMM_GROUP_M=$gm AMS_GEMM_SK=${SK:-1} timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/gms -o run -- python $R/tools/probes/one_gemm.py $1 5 > /tmp/gms.log 2>&1
  DB=$(find /tmp/gms -name "*.db" | head -1)
  python - "$DB" "$gm" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, sum(value) from counters_collection where counter_name='FETCH_SIZE' group by dispatch_id").fetchall()
v = [r[1] for r in rows if 'gemm_x6' in r[0]]
print('group_m %s: read %.1f MB per launch (2 x FETCH_SIZE, %d launches)' % (sys.argv[2], 2 * sum(v) / max(1, len(v)) * 1024 / 1e6, len(v)))
PY
  AMS_GEMM_GROUP_M=$gm AMS_GEMM_SK=${SK:-1} python $R/tools/probes/one_gemm.py $1 20 2>/dev/null | tail -1
done
