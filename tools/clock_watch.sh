#!/bin/bash
# Samples the GPU's engine clock and power while a command runs (rocm-smi, ~10 Hz):  tools/clock_watch.sh <cmd...>
"$@" > /tmp/clock_watch_cmd.log 2>&1 &
PID=$!
sleep "${CLOCK_WATCH_DELAY:-12}"
for i in $(seq 1 "${CLOCK_WATCH_N:-40}"); do
  kill -0 $PID 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' ' | sed 's/  */ /g'
  echo
done
wait $PID
tail -1 /tmp/clock_watch_cmd.log | cut -c1-160
