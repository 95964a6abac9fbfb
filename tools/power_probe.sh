#!/bin/bash
# Clock and package power of the GPU while a kernel of this library runs back to back (is the part power-limited under fp64 VALU load?).
# usage (GPU box): bash tools/power_probe.sh   -> gpurun_out/power_probe.txt
OUT=${1:-gpurun_out/power_probe.txt}
mkdir -p $(dirname $OUT)
{
echo "== idle"; rocm-smi --showpower --showmaxpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | head -8
for what in "closed loop RMCKF:time_methods.py --methods GMCKF --reps 12000" "closed loop KF:time_methods.py --methods KF --reps 12000" "closed loop MCKF:time_methods.py --methods MCKF --reps 12000" "noise generator alpha 1.5:noise_loop.py"; do
  name=${what%%:*}; cmd=${what#*:}
  python3 tools/$cmd > /dev/null 2>&1 &
  pid=$!
  ok=0                                   # bounded wait for the tool's start marker: a tool that dies or never writes it must not hang the lease
  for _ in $(seq 1 90); do
    if [ -s gpurun_out/.probe_started ]; then ok=1; break; fi
    if ! kill -0 $pid 2>/dev/null; then break; fi
    sleep 1
  done
  if [ $ok -ne 1 ]; then echo "== $name: tools/$cmd did not start (exited or wrote no marker within 90 s); skipped"; kill $pid 2>/dev/null; wait $pid 2>/dev/null; continue; fi
  sleep 6
  echo "== under load: $name"
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | head -3; sleep 0.7; done
  kill $pid; wait $pid 2>/dev/null; rm -f gpurun_out/.probe_started
done
} > $OUT 2>&1
