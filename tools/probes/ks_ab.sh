# soft k-means passes in the fine-tuning config, per variant library:  tools/probes/ks_ab.sh base ks1024 ...
R=$PWD; P=$PWD/adaptive-multispeaker-separation_amd/ams_hip
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/kp
  L=$P/libams_hip.so; [ "$v" != base ] && L=$P/libams_hip_$v.so
  AMS_HIP_LIB=$L rocprofv3 --kernel-trace --stats -d /tmp/kp -o run -- python $R/tools/bench_configs.py --only front_DPCL_finetuning_graph --steps 10 > /tmp/kp.log 2>&1
  grep '^{' /tmp/kp.log | cut -c1-60,250-330
  DB=$(find /tmp/kp -name "*.db" | head -1)
  python - "$DB" "$v" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for n, k, a, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels where name like '%kmeans_pass%' or name like '%ks_%' group by name"):
    print(sys.argv[2], n[22:70], k, round(a / 1e3, 1), round(mn / 1e3, 1))
PY
done
