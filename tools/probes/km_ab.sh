# k-means pass A/B over variant libraries: tools/probes/km_ab.sh "" ktpad ...   ("" = the shipped library)
R=$PWD; P=$PWD/adaptive-multispeaker-separation_amd/ams_hip
mkdir -p gpurun_out; : > gpurun_out/km_ab.txt
cd /tmp && export TMPDIR=/tmp
for v in "$@" "$@"; do
  rm -rf /tmp/kp
  L=$P/libams_hip.so; [ -n "$v" ] && [ "$v" != base ] && L=$P/libams_hip_$v.so
  AMS_HIP_LIB=$L rocprofv3 --kernel-trace --stats -d /tmp/kp -o run -- python $R/tools/bench_configs.py --only front_DPCL_inference --steps 10 > /tmp/kp.log 2>&1
  tail -1 /tmp/kp.log | cut -c1-150
  DB=$(find /tmp/kp -name "*.db" | head -1)
  python - "$DB" "$v" <<'PY' | tee -a $R/gpurun_out/km_ab.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for n, k, a, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels where name like '%kmeans_hard%' or name like '%kmeans_pass%' group by name"):
    print(sys.argv[2] or 'base', n[22:70], k, round(a / 1e3, 1), round(mn / 1e3, 1))
PY
done
