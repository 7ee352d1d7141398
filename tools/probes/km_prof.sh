P=$PWD/adaptive-multispeaker-separation_amd/ams_hip
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in kd0 kd1 kd0 kd1; do
  rm -rf /tmp/kp
  AMS_HIP_LIB=$P/libams_hip_$v.so rocprofv3 --kernel-trace --stats -d /tmp/kp -o run -- python $R/tools/bench_configs.py --only front_DPCL_inference --steps 10 > /tmp/kp.log 2>&1
  DB=$(find /tmp/kp -name "*.db" | head -1)
  python - "$DB" "$v" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for n, k, a in c.execute("select name, count(*), avg(duration) from kernels where name like '%kmeans_pass%' group by name"):
    print(sys.argv[2], n[22:60], k, round(a / 1e3, 1))
PY
done
