#!/bin/bash
# dev (GPU box): board power and clocks (rocm-smi samples) while one workload loops.  usage: scripts/dev/power_probe.sh <tag> "<perf_quick args with many iterations>"
TAG=$1; ARGS=$2
python scripts/perf_quick.py $ARGS > /tmp/pp_$TAG.log 2>&1 &
PID=$!
sleep 6   # import + synth + first runs
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|memory)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $PID
grep Gbases /tmp/pp_$TAG.log | sed "s/^/$TAG /"
