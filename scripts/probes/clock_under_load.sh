#!/bin/bash
# samples rocm-smi while a loop runs: $1 = label, rest = command
label=$1; shift
"$@" > /dev/null 2>&1 &
pid=$!
sleep 14
for i in 1 2 3; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';'
  echo " [$label]"
  sleep 1
done
wait $pid
