# k-means: parity tests of the five-tries kernel, then the inference config traced with it on and off (AMS_KM_TRIES=0)
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kmeans_tries.py tests/test_gpu_kernels2.py tests/test_golden.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "kmeans or golden" 2>&1 | tail -15 | tee gpurun_out/km_tries_tests.txt
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/kp
  AMS_KM_TRIES=$v rocprofv3 --kernel-trace --stats -d /tmp/kp -o run -- python $R/tools/bench_configs.py --only front_DPCL_inference --steps 10 > /tmp/kp_$v.log 2>&1
  tail -2 /tmp/kp_$v.log
  DB=$(find /tmp/kp -name "*.db" | head -1)
  python - "$DB" "$v" <<'PY' | tee -a $R/gpurun_out/km_tries_kernels.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for n, k, a, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels where name like '%kmeans%' group by name"):
    print('AMS_KM_TRIES=' + sys.argv[2], n[22:70], k, round(a / 1e3, 1), round(mn / 1e3, 1))
PY
done
