#!/bin/bash
# Shader clock and socket power (rocm-smi) while the headline step replays in a loop, with and without the pre-split forward products:
# is the STEP power-limited, i.e. does a faster forward pass slow the backward pass down?   usage: tools/probes/step_power_probe.sh
for v in 0 1; do
  AMS_GEMM_PRESPLIT=$v python bench.py --steps 6000 --warmup 10 --no-cpu-baseline --no-secondary --no-native-f32 --roofline-steps 0 --quiet > /tmp/spp_$v.log 2>&1 &
  PID=$!
  sleep ${DELAY:-14}
  for i in 1 2 3 4; do
    echo -n "presplit=$v  "; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/GPU\[0\][^:]*: //' | tr '\n' ' '; echo
    sleep 0.7
  done
  wait $PID
  tail -1 /tmp/spp_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('presplit=$v', d['ms_per_step'], 'ms per step')"
done
