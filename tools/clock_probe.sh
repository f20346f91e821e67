#!/bin/bash
# Sample the GPU clocks / power while a workload runs:  tools/clock_probe.sh <label> <command...>
label=$1; shift
"$@" > /dev/null 2>&1 &
pid=$!
sleep 8   # let the workload get past its start-up (import, allocation, warm-up)
echo "== $label"
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.7
done
wait $pid
