#!/bin/bash
# Sample the shader clock and the power draw (rocm-smi) while one product runs in a loop: is a kernel clock/power limited?
# usage: tools/probes/clock_probe.sh <arith 0|1> [label-substring]   (AMS_HIP_LIB / AMS_GEMM_X6CFG are honoured)
python tools/gemm_x6_bench.py --modes $1 --only "${2:-square}" --reps ${REPS:-12000} > /tmp/clock_probe_bench.log 2>&1 &
PID=$!
sleep ${DELAY:-7}
for i in 1 2 3; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr '\n' ' '; echo
  sleep 0.5
done
wait $PID
grep -v amdgpu.ids /tmp/clock_probe_bench.log | cut -c1-120
